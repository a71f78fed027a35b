"""Canonical per-tick record + digest shared by every parity checker.

TEST INFRASTRUCTURE ONLY (see oracle/README.md): imported by tests/, by
tests/golden/gen_golden.py, by __graft_entry__.smoke() and by bench.py's
cpu_baseline leg. The product package never imports anything from oracle/.

A *tick record* is what one caller-protocol tick of the reference produces
(`for lane, ind: step()` -> `scene_update()`; snapshot taken BEFORE
`delete_vehicle()`; reference: traffic_interaction_scene.py:222-376, 1501-1539):

  per controlled vehicle, in processing order (= `ids` order, :291)
    ids      int32 [C,2]    (lane, j), pre-compaction                 (:291)
    nbr      int32 [C,6,2]  six nearest (lane, j) or -1               (:1391-1405)
    reward   f64   [C]                                                (:311-320,346,357)
    obs0     f64   [C,28]   row 0 of the 7x28 state                   (:1336-1337)
    state    f64   [C,7,28] full state (optional)                     (:1325-1337)
    act7     f64   [C,7]    column 2 of the 7 rows (optional)         (:290)
    coll_pv  int32 [C]      collisions_per_veh[:,0]                   (:339-340)
  scalars: collisions (:337), lock (:365-370), time (:223)
  jerks    f64 [F]      jerk_sum of vehicles finishing this tick      (:358)
  deleted  int32 [D,2]  delete_veh in scene order                     (:348)
  per alive vehicle incl. this tick's spawns, (lane, j) order
    veh_i    int32 [N,15]  lane, j, id, seq_in_lane, vnum(id_info[1]), control, finish, done,
                           collision, step, count, lock, lock_a, hdr_lane, hdr_j
    veh_f    f64   [N,7]   p, v, a, jerk, jerk_sum, vir_dis, closer_p
  env: id_seq, passed, passed_step_total, veh_num[12], veh_rec[12],
       heads int32 [12,3] = (len>0, lane, j) of virtual_lane_4[d][0]   (:1517)
"""
import zlib
import numpy as np

VEH_I_COLS = ("lane", "j", "id", "seq", "vnum", "control", "finish", "done", "collision",
              "step", "count", "lock", "lock_a", "hdr_lane", "hdr_j")
VEH_F_COLS = ("p", "v", "a", "jerk", "jerk_sum", "vir_dis", "closer_p")

# integer scalars stored verbatim in the digest table
DIGEST_I_COLS = ("n_alive", "n_ctl", "id_seq", "passed", "passed_step_total", "collisions",
                 "lock", "n_deleted", "n_jerks", "crc")
DIGEST_F_COLS = ("time", "sum_p", "sum_v", "sum_a", "sum_jerk", "sum_jerk_sum", "sum_vir_dis",
                 "sum_closer_p", "sum_reward", "sum_obs0", "sum_abs_obs0", "sum_jerks")


def empty_record():
    return dict(
        tick=0, time=0.0,
        ids=np.zeros((0, 2), np.int32), nbr=np.zeros((0, 6, 2), np.int32),
        reward=np.zeros((0,), np.float64), obs0=np.zeros((0, 28), np.float64),
        state=None, act7=None,
        coll_pv=np.zeros((0,), np.int32), collisions=0, lock=0,
        jerks=np.zeros((0,), np.float64), deleted=np.zeros((0, 2), np.int32),
        veh_i=np.zeros((0, len(VEH_I_COLS)), np.int32), veh_f=np.zeros((0, len(VEH_F_COLS)), np.float64),
        id_seq=0, passed=0, passed_step_total=0,
        veh_num=np.zeros(12, np.int32), veh_rec=np.zeros(12, np.int32),
        heads=np.zeros((12, 3), np.int32),
    )


def int_blob(rec):
    """Canonical int32 vector of every integer-valued field (bit-exact parity domain)."""
    parts = [rec["ids"], rec["nbr"], rec["coll_pv"], rec["deleted"], rec["veh_i"],
             rec["veh_num"], rec["veh_rec"], rec["heads"]]
    return np.concatenate([np.ascontiguousarray(p, dtype=np.int32).ravel() for p in parts])


def digest(rec):
    """-> (int64[len(DIGEST_I_COLS)], f64[len(DIGEST_F_COLS)])"""
    blob = int_blob(rec)
    crc = zlib.crc32(blob.astype("<i4").tobytes()) & 0xFFFFFFFF
    vf = rec["veh_f"]
    di = np.array([rec["veh_i"].shape[0], rec["ids"].shape[0], rec["id_seq"], rec["passed"],
                   rec["passed_step_total"], rec["collisions"], rec["lock"],
                   rec["deleted"].shape[0], rec["jerks"].shape[0], crc], dtype=np.int64)
    sums = [float(vf[:, k].sum()) if vf.shape[0] else 0.0 for k in range(vf.shape[1])]
    df = np.array([rec["time"]] + sums +
                  [float(rec["reward"].sum()), float(rec["obs0"].sum()),
                   float(np.abs(rec["obs0"]).sum()), float(rec["jerks"].sum())], dtype=np.float64)
    return di, df


def close(x, y, tol=1e-5):
    """north_star tolerance: |x-y| <= tol*max(1,|x|) (mixed relative)."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    if x.shape != y.shape:
        return False
    return bool(np.all(np.abs(x - y) <= tol * np.maximum(1.0, np.abs(x))))


def compare_records(a, b, tol=1e-5, check_state=True, label=""):
    """Raise AssertionError naming the first differing field; ints exact, floats within tol."""
    def fail(name, extra=""):
        raise AssertionError("%s tick %s: field %s differs %s" % (label, a.get("tick"), name, extra))
    for name in ("ids", "nbr", "coll_pv", "deleted", "veh_i", "veh_num", "veh_rec", "heads"):
        x = np.asarray(a[name], np.int64)
        y = np.asarray(b[name], np.int64)
        if x.shape != y.shape or not np.array_equal(x, y):
            extra = ""
            if x.shape == y.shape and x.size:
                bad = np.argwhere(x != y)[0]
                extra = "at %s: %s vs %s" % (bad.tolist(), x[tuple(bad)], y[tuple(bad)])
                if name == "veh_i":
                    extra += " (col %s, row %s)" % (VEH_I_COLS[bad[1]], x[bad[0]].tolist())
            else:
                extra = "shape %s vs %s" % (x.shape, y.shape)
            fail(name, extra)
    for name in ("collisions", "lock", "id_seq", "passed", "passed_step_total"):
        if int(a[name]) != int(b[name]):
            fail(name, "%s vs %s" % (a[name], b[name]))
    for name in ("reward", "obs0", "jerks", "veh_f"):
        if not close(a[name], b[name], tol):
            x = np.asarray(a[name], np.float64)
            y = np.asarray(b[name], np.float64)
            extra = "shape %s vs %s" % (x.shape, y.shape)
            if x.shape == y.shape:
                err = np.abs(x - y) / np.maximum(1.0, np.abs(x))
                k = np.unravel_index(np.argmax(err), err.shape)
                extra = "max err %.3e at %s: %r vs %r" % (err[k], k, x[k], y[k])
            fail(name, extra)
    if not close(a["time"], b["time"], 1e-12):
        fail("time", "%r vs %r" % (a["time"], b["time"]))
    if check_state:
        for name in ("state", "act7"):
            if a.get(name) is not None and b.get(name) is not None:
                if not close(a[name], b[name], tol):
                    x = np.asarray(a[name], np.float64)
                    y = np.asarray(b[name], np.float64)
                    err = np.abs(x - y) / np.maximum(1.0, np.abs(x)) if x.shape == y.shape else None
                    extra = "shape %s vs %s" % (x.shape, y.shape) if err is None else \
                        "max err %.3e at %s" % (err.max(), np.unravel_index(np.argmax(err), err.shape))
                    fail(name, extra)


# ---------------------------------------------------------------- policies (action "tapes")
def policy_zero(tick, veh_id, control, obs0=None):
    return np.zeros(len(veh_id), np.float64)


def make_policy_sin(amp):
    """a = float32(A*sin(0.37*id + 0.05*tick)) for controlled vehicles, 0 otherwise.
    Rounded to float32 like a real actor output (model_agent_maddpg.py:15 uses tf.float32), which
    also makes the tape independent of last-bit differences between libm / NumPy sin builds."""
    def pol(tick, veh_id, control, obs0=None):
        a = amp * np.sin(0.37 * np.asarray(veh_id, np.float64) + 0.05 * float(tick))
        a = a.astype(np.float32).astype(np.float64)
        return np.where(np.asarray(control) != 0, a, 0.0)
    return pol


def make_policy_rand(amp):
    """a = float32(A * u), u in [-1, 1) from an integer hash of (vehicle id, tick): a reproducible "random" tape that needs no
    generator state (the same value wherever it is evaluated), for controlled vehicles, 0 otherwise."""
    def pol(tick, veh_id, control, obs0=None):
        x = (np.asarray(veh_id, np.uint64) * np.uint64(2654435761) + np.uint64(int(tick)) * np.uint64(40503) + np.uint64(12345)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(15); x = (x * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(13); x = (x * np.uint64(3266489917)) & np.uint64(0xFFFFFFFF)
        x ^= x >> np.uint64(16)
        u = x.astype(np.float64) / 2147483648.0 - 1.0
        a = (amp * u).astype(np.float32).astype(np.float64)
        return np.where(np.asarray(control) != 0, a, 0.0)
    return pol


def get_policy(name):
    if name == "zero":
        return policy_zero
    if name.startswith("sin"):
        return make_policy_sin(float(name[3:]))
    if name.startswith("rand"):
        return make_policy_rand(float(name[4:]))
    raise KeyError(name)
