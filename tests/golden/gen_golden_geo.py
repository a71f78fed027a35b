"""Golden vectors of the 4- and 8-lane geometries, made by importing the UNMODIFIED reference.

Run in the build container only:   python tests/golden/gen_golden_geo.py
Outputs (committed): tests/golden/geo_<case>.npz and tests/golden/geometry_geo.npz.

The reference ships no 4-/8-lane arrival streams (data/test holds *_12.mat only), so the streams are synthetic
(seeded Poisson gaps clipped at 1 s, like the shipped ones) and are stored in the fixture together with the
8-lane intention draws that replace the reference's entropy-seeded random.randint(0, 1) (ref :381, :390; see
tests/golden/ref_harness.py:_ChoiceShim).  Same per-case layout as gen_golden.py plus `choice`, `intent`.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle.record import digest, get_policy, DIGEST_I_COLS, DIGEST_F_COLS  # noqa: E402
from tests.golden import ref_harness as rh  # noqa: E402
from tests.golden.gen_golden import dense_tick_set  # noqa: E402

CASES = [
    # name, lane_num, policy, ticks, mean gap [s], seed
    ("g4_zero", 4, "zero", 1000, 2.0, 41),
    ("g4_sin2", 4, "sin2", 1200, 1.5, 42),
    ("g4_sin3", 4, "sin3", 800, 1.2, 43),
    ("g8_zero", 8, "zero", 1000, 2.0, 81),
    ("g8_sin2", 8, "sin2", 1200, 1.5, 82),
    ("g8_sin3", 8, "sin3", 800, 1.2, 83),
]
DENSE_FIELDS = ("ids", "nbr", "reward", "obs0", "coll_pv", "deleted", "jerks", "veh_i", "veh_f",
                "heads", "veh_num", "veh_rec", "intent")


def make_stream(lane_num, rows, mean, seed):
    rng = np.random.default_rng(seed)
    gaps = np.maximum(1.0, rng.exponential(mean, size=(rows, lane_num)))
    return np.cumsum(gaps, axis=0), rng.integers(0, 2, size=(rows, lane_num)).astype(np.int32)


def gen_case(name, lane_num, pol, ticks, mean, seed):
    arr, choice = make_stream(lane_num, 400, mean, seed)
    ref = rh.GeoRefRunner(arr, lane_num, get_policy(pol), choice=choice, want_state=True)
    dense = dense_tick_set(ticks)
    state_ticks = [3, ticks // 2, ticks - 1]
    out = {}
    dig_i = np.zeros((ticks, len(DIGEST_I_COLS)), np.int64)
    dig_f = np.zeros((ticks, len(DIGEST_F_COLS)), np.float64)
    n_coll = n_lock = 0
    for t in range(ticks):
        rec = ref.tick()
        dig_i[t], dig_f[t] = digest(rec)
        n_coll += int((rec["coll_pv"] > 0).sum())
        n_lock += rec["lock"]
        if t in dense or t in state_ticks:
            for f in DENSE_FIELDS:
                out["t%d_%s" % (t, f)] = rec[f]
            out["t%d_scalars" % t] = np.array([rec["collisions"], rec["lock"], rec["id_seq"], rec["passed"],
                                               rec["passed_step_total"]], np.int64)
            out["t%d_time" % t] = np.array(rec["time"], np.float64)
            out["t%d_tape" % t] = np.asarray(ref.tape, np.float64)
        if t in state_ticks:
            out["t%d_state" % t] = rec["state"]
            out["t%d_act7" % t] = rec["act7"]
    ref.close()
    rows = int(np.max(ref.env.veh_rec)) + 2
    out["arrive"] = arr[:rows].copy()
    out["choice"] = choice[:rows].copy()
    out["dig_i"], out["dig_f"] = dig_i, dig_f
    out["dense_ticks"] = np.array(sorted(set(dense) | set(state_ticks)), np.int32)
    out["state_ticks"] = np.array(state_ticks, np.int32)
    out["guard_hits"] = np.array(ref.guard_hits, np.int32)
    out["meta"] = np.array(json.dumps(dict(name=name, lane_num=lane_num, stream="synthetic mean %.1f s seed %d" % (mean, seed),
                                           policy=pol, ticks=ticks, ctor={}, numpy=np.__version__)))
    path = os.path.join(HERE, "geo_" + name + ".npz")
    np.savez_compressed(path, **out)
    print("%-10s lanes %2d ticks %4d alive-steps %6d ctl-steps %6d id_seq %3d passed %3d collided %3d locks %4d "
          "guard %d -> %d KB" % (name, lane_num, ticks, dig_i[:, 0].sum(), dig_i[:, 1].sum(), dig_i[-1, 2],
                                 dig_i[-1, 3], n_coll, n_lock, ref.guard_hits, os.path.getsize(path) // 1024))


def gen_geometry():
    """Known answers of get_p / get_virtual_distance for lane_num 4 and 8, sampled through the reference."""
    tis = rh.import_reference()
    out = {}
    for ln in (4, 8):
        arr, _ = make_stream(ln, 50, 3.0, 7)
        env = tis.TrafficInteraction(arr, 150, rh.default_args(), show_col=False, virtual_l=True, lane_num=ln)
        li = np.array(env.lane_info, np.float64)
        ps = np.concatenate([np.linspace(-140, 170, 125), np.linspace(0, 21, 85),
                             np.array([0.0, 1e-9, li[0][1], li[1][1], li[2][1], li[0][1] + 1e-9, li[2][1] - 1e-9])])
        gp = np.full((ln, 3, len(ps), 2), np.nan)
        for lane in range(ln):
            for m in range(3):
                if env.direction[lane][m] == -1:
                    continue
                for k, p in enumerate(ps):
                    q = env.get_p(float(p), lane, m)
                    gp[lane, m, k] = (q[0], q[1])
        nd = env.direction_num
        vd = np.full((nd, nd, len(ps)), np.nan)
        for ego in range(nd):
            for other in env.lane2lane[ego]:
                for k, p in enumerate(ps):
                    d, ch = env.get_virtual_distance(other, ego, float(p))
                    assert len(d) <= 1
                    if ch:
                        vd[ego, other, k] = d[0]
        out["ps%d" % ln], out["get_p%d" % ln], out["vd%d" % ln], out["lane_info%d" % ln] = ps, gp, vd, li
    np.savez_compressed(os.path.join(HERE, "geometry_geo.npz"), **out)
    print("geometry_geo.npz written")


if __name__ == "__main__":
    only = sys.argv[1:]
    gen_geometry()
    for c in CASES:
        if not only or c[0] in only:
            gen_case(*c)
