"""ctypes wrapper of the CPU oracle (oracle/pve_oracle.c).  TEST INFRASTRUCTURE ONLY.

Allowed importers: tests/, tests/golden/gen_golden.py, __graft_entry__.smoke(),
bench.py's cpu_baseline leg.  The product package must never import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .record import VEH_I_COLS, VEH_F_COLS

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpve_oracle.so")
_lib = None


class PvoParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("deltaT", "vm", "vM", "am", "aM", "v0", "lane_cw", "dis_ctl", "collision_thr")]


def build(force=False):
    src = os.path.join(_HERE, "pve_oracle.c")
    if force or not os.path.isfile(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpve_oracle.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.pvo_default_params.argtypes = [C.POINTER(PvoParams)]
    L.pvo_create.restype = vp
    L.pvo_create.argtypes = [dp, C.c_int, C.POINTER(PvoParams), C.c_int]
    L.pvo_destroy.argtypes = [vp]
    L.pvo_step.argtypes = [vp, C.c_int, C.c_int, C.c_double]
    L.pvo_scene_update.argtypes = [vp]
    L.pvo_delete_vehicle.argtypes = [vp]
    L.pvo_tick_actions.argtypes = [vp, dp]
    for name in ("pvo_n_ctl", "pvo_collisions", "pvo_lock", "pvo_n_jerks", "pvo_n_deleted",
                 "pvo_ref_would_raise", "pvo_n_alive", "pvo_peak_alive"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp]
    for name in ("pvo_ids", "pvo_nbr", "pvo_coll_pv", "pvo_deleted"):
        getattr(L, name).restype = ip
        getattr(L, name).argtypes = [vp]
    for name in ("pvo_reward", "pvo_state", "pvo_jerks"):
        getattr(L, name).restype = dp
        getattr(L, name).argtypes = [vp]
    L.pvo_time.restype = C.c_double
    L.pvo_time.argtypes = [vp]
    L.pvo_lane_counts.argtypes = [vp, ip]
    L.pvo_export_vehicles.argtypes = [vp, ip, dp, dp]
    L.pvo_export_env.argtypes = [vp, ip]
    L.pvo_get_p.argtypes = [vp, C.c_double, C.c_int, dp]
    L.pvo_get_virtual_distance.restype = C.c_int
    L.pvo_get_virtual_distance.argtypes = [vp, C.c_int, C.c_int, C.c_double, dp]
    L.pvo_run.restype = C.c_long
    L.pvo_run.argtypes = [vp, C.c_int, C.c_int, C.c_double, C.c_int, C.POINTER(C.c_long)]
    L.pvo_run_pool.restype = C.c_long
    L.pvo_run_pool.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]
    L.pvo_run_many.restype = C.c_long
    L.pvo_run_many.argtypes = [C.POINTER(vp), ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, dp, C.c_int, C.c_int, C.c_int,
                               C.POINTER(C.c_long)]
    _lib = L
    return L


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros((0,), dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class OracleEnv:
    """Single environment; same call protocol as the reference object
    (ctor warm-up, step, scene_update, delete_vehicle)."""

    def __init__(self, arrive_time, **params):
        L = lib()
        self._L = L
        prm = PvoParams()
        L.pvo_default_params(C.byref(prm))
        for k, v in params.items():
            if not hasattr(prm, k):
                raise TypeError("unknown parameter %s" % k)
            setattr(prm, k, float(v))
        arr = np.ascontiguousarray(arrive_time, dtype=np.float64)
        assert arr.ndim == 2 and arr.shape[1] == 12
        self._arr = arr
        self._h = L.pvo_create(arr.ctypes.data_as(C.POINTER(C.c_double)), arr.shape[0], C.byref(prm), 0)
        self.tick_no = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pvo_destroy(self._h)
            self._h = None

    # -- reference-shaped calls
    def step(self, lane, ind, a):
        self._L.pvo_step(self._h, int(lane), int(ind), float(a))

    def scene_update(self):
        self._L.pvo_scene_update(self._h)

    def delete_vehicle(self):
        self._L.pvo_delete_vehicle(self._h)

    # -- views
    @property
    def n_alive(self):
        return self._L.pvo_n_alive(self._h)

    @property
    def current_time(self):
        return self._L.pvo_time(self._h)

    @property
    def ref_would_raise(self):
        return self._L.pvo_ref_would_raise(self._h)

    @property
    def peak_alive(self):
        """most vehicles alive at the start of a tick of run() (the slots a capacity-bound build would need)"""
        return self._L.pvo_peak_alive(self._h)

    def lane_counts(self):
        out = np.zeros(12, np.int32)
        self._L.pvo_lane_counts(self._h, out.ctypes.data_as(C.POINTER(C.c_int)))
        return out

    def vehicles(self):
        n = self.n_alive
        vi = np.zeros((n, len(VEH_I_COLS)), np.int32)
        vf = np.zeros((n, len(VEH_F_COLS)), np.float64)
        obs0 = np.zeros((n, 28), np.float64)
        if n:
            self._L.pvo_export_vehicles(self._h, vi.ctypes.data_as(C.POINTER(C.c_int)),
                                        vf.ctypes.data_as(C.POINTER(C.c_double)),
                                        obs0.ctypes.data_as(C.POINTER(C.c_double)))
        return vi, vf, obs0

    def alive_view(self):
        vi, _vf, obs0 = self.vehicles()
        return vi[:, 2].astype(np.int64), vi[:, 5].copy(), obs0

    def get_p(self, p, lane):
        out = np.zeros(2, np.float64)
        self._L.pvo_get_p(self._h, float(p), int(lane), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def get_virtual_distance(self, lane1, lane2, p1):
        vd = C.c_double(0.0)
        ok = self._L.pvo_get_virtual_distance(self._h, lane1, lane2, float(p1), C.byref(vd))
        return (vd.value if ok else None)

    # -- one caller-protocol tick -> canonical record (snapshot before delete), then compaction
    def tick(self, actions, want_state=False):
        L, h = self._L, self._h
        actions = np.ascontiguousarray(actions, dtype=np.float64)
        assert actions.shape[0] == self.n_alive
        L.pvo_tick_actions(h, actions.ctypes.data_as(C.POINTER(C.c_double)))
        rec = self.snapshot(want_state)
        L.pvo_delete_vehicle(h)
        self.tick_no += 1
        return rec

    def snapshot(self, want_state=False):
        L, h = self._L, self._h
        Cn = L.pvo_n_ctl(h)
        rec = dict(tick=self.tick_no, time=L.pvo_time(h))
        rec["ids"] = _arr(L.pvo_ids(h), Cn * 2, np.int32).reshape(Cn, 2)
        rec["nbr"] = _arr(L.pvo_nbr(h), Cn * 12, np.int32).reshape(Cn, 6, 2)
        rec["reward"] = _arr(L.pvo_reward(h), Cn, np.float64)
        st = _arr(L.pvo_state(h), Cn * 196, np.float64).reshape(Cn, 7, 28)
        rec["obs0"] = np.ascontiguousarray(st[:, 0, :])
        rec["state"] = st if want_state else None
        rec["act7"] = np.ascontiguousarray(st[:, :, 2]) if want_state else None
        rec["coll_pv"] = _arr(L.pvo_coll_pv(h), Cn, np.int32)
        rec["collisions"] = L.pvo_collisions(h)
        rec["lock"] = L.pvo_lock(h)
        nj = L.pvo_n_jerks(h)
        rec["jerks"] = _arr(L.pvo_jerks(h), nj, np.float64)
        nd = L.pvo_n_deleted(h)
        rec["deleted"] = _arr(L.pvo_deleted(h), nd * 2, np.int32).reshape(nd, 2)
        vi, vf, _ = self.vehicles()
        rec["veh_i"], rec["veh_f"] = vi, vf
        ev = np.zeros(63, np.int32)
        L.pvo_export_env(h, ev.ctypes.data_as(C.POINTER(C.c_int)))
        rec["id_seq"], rec["passed"], rec["passed_step_total"] = int(ev[0]), int(ev[1]), int(ev[2])
        rec["veh_num"] = ev[3:15].copy()
        rec["veh_rec"] = ev[15:27].copy()
        rec["heads"] = ev[27:63].reshape(12, 3).copy()
        return rec

    def run(self, ticks, policy=0, amp=1.0, tick0=0):
        """Timing loop entirely in C (GIL released by ctypes). -> (alive_steps, ctl_steps)"""
        ctl = C.c_long(0)
        alive = self._L.pvo_run(self._h, int(ticks), int(policy), float(amp), int(tick0), C.byref(ctl))
        self.tick_no += ticks
        return int(alive), int(ctl.value)

    def run_pool(self, ticks, pool, tick0=0):
        """Timing loop with a per-slot action pool [n_pool, cap] (see pvo_run_pool)."""
        pool = np.ascontiguousarray(pool, dtype=np.float64)
        ctl = C.c_long(0)
        alive = self._L.pvo_run_pool(self._h, int(ticks), pool.ctypes.data_as(C.POINTER(C.c_double)),
                                     pool.shape[0], pool.shape[1], int(tick0), C.byref(ctl))
        self.tick_no += ticks
        return int(alive), int(ctl.value)


def run_many(envs, env_index, n_threads, ticks, policy=1, amp=1.0, tick0=0, pool=None):
    """bench.py's cpu_baseline on all host cores: `envs` (OracleEnv objects) dealt to n_threads POSIX threads inside ONE C call
    (pvo_run_many) -- no Python, no GIL in the timed loop.  pool: [n_pool, n_envs_total, cap] float64 (env_index[i] = the row of
    envs[i] in it) for the slot-indexed tape, None for policy 0 (zero) / 1 (a = amp sin(0.37 id + 0.05 tick)).
    -> (alive_steps, ctl_steps)"""
    L = lib()
    hs = (C.c_void_p * len(envs))(*[e._h for e in envs])
    idx = np.ascontiguousarray(env_index, dtype=np.int32)
    ctl = C.c_long(0)
    if pool is not None:
        pool = np.ascontiguousarray(pool, dtype=np.float64)
        pp, n_pool, n_tot, cap = pool.ctypes.data_as(C.POINTER(C.c_double)), pool.shape[0], pool.shape[1], pool.shape[2]
    else:
        pp, n_pool, n_tot, cap = None, 0, 0, 0
    alive = L.pvo_run_many(hs, idx.ctypes.data_as(C.POINTER(C.c_int)), len(envs), int(n_threads), int(ticks), int(policy), float(amp),
                           int(tick0), pp, n_pool, cap, n_tot, C.byref(ctl))
    if alive < 0:
        raise MemoryError("pvo_run_many")
    for e in envs:
        e.tick_no += ticks
    return int(alive), int(ctl.value)
