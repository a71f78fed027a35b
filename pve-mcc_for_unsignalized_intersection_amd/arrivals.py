"""Arrival streams: the synthetic generator of BASELINE.md §3 and the loader of the reference's
MATLAB v5 `arvTimeNewVeh` files (main.py:387-389, data/test/*.mat)."""
import numpy as np


def synthetic_arrivals(n_envs, rate, horizon_s, seed=20250213, rows=None, lane_num=12):
    """float64 [n_envs, rows, lane_num]: per env e and lane, cumulative sums of inter-arrival times
    max(1.0, Exponential(3600/rate)) from numpy.random.default_rng(seed + e), padded with +inf
    (the shipped streams have the same clipped-exponential shape)."""
    mean = 3600.0 / float(rate)
    if rows is None:
        rows = 64 + int(np.ceil(horizon_s / mean)) * 2
    out = np.full((n_envs, rows, lane_num), np.inf, dtype=np.float64)
    for e in range(n_envs):
        rng = np.random.default_rng(seed + e)
        dt = np.maximum(1.0, rng.exponential(mean, size=(rows - 1, lane_num)))
        out[e, :rows - 1, :] = np.cumsum(dt, axis=0)
    return out


def synthetic_intentions(n_envs, rows, seed=20250213, lane_num=8):
    """int32 [n_envs, rows, lane_num] of 0/1: the intention draws of the 8-lane layout (the reference's
    random.randint(0, 1) at ref :390, one per arriving vehicle), from numpy.random.default_rng(seed + e)."""
    out = np.zeros((n_envs, rows, lane_num), dtype=np.int32)
    for e in range(n_envs):
        out[e] = np.random.default_rng(seed + 7919 * (e + 1)).integers(0, 2, size=(rows, lane_num))
    return out


def pad_stream(arr, extra_rows=1):
    """Replace the zero padding at the tail of a shipped stream by +inf so that an
    exhausted lane simply stops spawning (the reference would spawn every tick, ref :379)."""
    a = np.array(arr, dtype=np.float64, copy=True)
    for l in range(a.shape[1]):
        col = a[:, l]
        nz = np.flatnonzero(col > 0)
        last = nz[-1] if len(nz) else -1
        col[last + 1:] = np.inf
    if extra_rows:
        a = np.concatenate([a, np.full((extra_rows, a.shape[1]), np.inf)], axis=0)
    return a


# ---------------------------------------------------------------------------------------------
# MATLAB v5 reader for the reference's arrival streams (main.py:387-389 uses scipy.io.loadmat;
# this ~60-line reader removes the SciPy dependency for the one variable the path needs).
_MI_SIZES = {1: ("i1", 1), 2: ("u1", 1), 3: ("i2", 2), 4: ("u2", 2), 5: ("i4", 4), 6: ("u4", 4),
             7: ("f4", 4), 9: ("f8", 8), 12: ("i8", 8), 13: ("u8", 8)}


def _read_tag(buf, off):
    """-> (type, nbytes, data_offset, next_offset) of one MAT-v5 data element (handles the small format)."""
    import struct
    t, n = struct.unpack_from("<II", buf, off)
    if t >> 16:                                    # small data element: size in the upper half-word
        return t & 0xFFFF, t >> 16, off + 4, off + 8
    return t, n, off + 8, off + 8 + (n + 7) // 8 * 8


def load_arrival_mat(path, name="arvTimeNewVeh"):
    """float64 [rows, 12] `arvTimeNewVeh` matrix of a MATLAB 5.0 MAT-file (data/test/*.mat of the reference).
    Supports little-endian files, zlib-compressed (miCOMPRESSED) or plain miMATRIX elements, real numeric
    2-D arrays of any integer / float storage type."""
    import struct
    import zlib
    with open(path, "rb") as f:
        raw = f.read()
    if len(raw) < 128 or not raw.startswith(b"MATLAB 5.0 MAT-file"):
        raise ValueError("%s: not a MATLAB 5.0 MAT-file" % path)
    if raw[126:128] != b"IM":
        raise ValueError("%s: big-endian MAT-files are not supported" % path)
    off = 128
    while off + 8 <= len(raw):
        t, n, d0, nxt = _read_tag(raw, off)
        off = nxt if t != 15 else d0 + n           # compressed elements are not padded
        if t == 15:
            elem = zlib.decompress(raw[d0:d0 + n])
            t, n, d0, _ = _read_tag(elem, 0)
        else:
            elem = raw
        if t != 14:
            continue
        p = d0
        ft, fn, fd, p = _read_tag(elem, p)          # array flags
        flags = struct.unpack_from("<I", elem, fd)[0]
        klass, is_complex = flags & 0xFF, bool(flags & 0x0800)
        dt, dn, dd, p = _read_tag(elem, p)          # dimensions
        dims = np.frombuffer(elem, dtype="<i4", count=dn // 4, offset=dd)
        nt, nn, nd, p = _read_tag(elem, p)          # name
        vname = elem[nd:nd + nn].decode("latin1")
        if vname != name:
            continue
        if is_complex or klass not in (6, 7, 8, 9, 10, 11, 12, 13, 14, 15) or len(dims) != 2:
            raise ValueError("%s: variable %s is not a real numeric 2-D array" % (path, name))
        rt, rn, rd, p = _read_tag(elem, p)          # real part, column-major, possibly a narrower storage type
        if rt not in _MI_SIZES:
            raise ValueError("%s: unsupported storage type %d" % (path, rt))
        code, size = _MI_SIZES[rt]
        data = np.frombuffer(elem, dtype="<" + code, count=rn // size, offset=rd).astype(np.float64)
        return np.ascontiguousarray(data.reshape(int(dims[1]), int(dims[0])).T)
    raise KeyError("%s: variable %s not found" % (path, name))
