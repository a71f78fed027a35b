"""Shared helpers for the parity tests: golden-case loading, digest checks, replay loops."""
import json
import os

import numpy as np

from oracle.record import (DIGEST_F_COLS, DIGEST_I_COLS, compare_records, digest, get_policy)

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASE_NAMES = ["s1000_zero", "s1000_sin1", "s200_sin1", "s1200_sin1", "s1200_zero", "s400_sin2",
              "s1000_sin3", "s1000_sin1_vm6", "s1000_rand_kw", "s1000_actor"]
# 4- / 8-lane geometries: synthetic streams + intention draws stored in the fixture
GEO_CASE_NAMES = ["geo_g4_zero", "geo_g4_sin2", "geo_g4_sin3", "geo_g8_zero", "geo_g8_sin2", "geo_g8_sin3"]
DENSE_FIELDS = ("ids", "nbr", "reward", "obs0", "coll_pv", "deleted", "jerks", "veh_i", "veh_f",
                "heads", "veh_num", "veh_rec")


class GoldenCase:
    def __init__(self, name):
        self.name = name
        self.z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.arrive = np.ascontiguousarray(self.z["arrive"], np.float64)
        self.dig_i = self.z["dig_i"]
        self.dig_f = self.z["dig_f"]
        self.ticks = int(self.meta["ticks"])
        self.ctor = dict(self.meta["ctor"])
        if self.meta["policy"] == "actor_tape":
            # the recorded actions of the pretrained actor on the live reference env (gen_golden.py); the tape,
            # not the network, is what is replayed
            self.tape_vals = self.z["tape_vals"].astype(np.float64)
            self.tape_off = self.z["tape_off"]
            self.aggregates = json.loads(str(self.z["aggregates"]))
            self.policy = self._tape_policy
        else:
            self.policy = get_policy(self.meta["policy"])
        self.dense_ticks = set(int(t) for t in self.z["dense_ticks"])
        self.state_ticks = set(int(t) for t in self.z["state_ticks"])
        self.lane_num = int(self.meta.get("lane_num", 12))
        self.choice = np.ascontiguousarray(self.z["choice"], np.int32) if "choice" in self.z.files else None

    def _tape_policy(self, tick, veh_id, control, obs0=None):
        a = self.tape_vals[self.tape_off[tick]:self.tape_off[tick + 1]]
        assert len(a) == len(veh_id), "%s tick %d: the tape holds %d actions for %d alive vehicles" % (self.name, tick, len(a), len(veh_id))
        return a

    def dense_record(self, t):
        z = self.z
        rec = {f: z["t%d_%s" % (t, f)] for f in DENSE_FIELDS}
        if "t%d_intent" % t in z.files:
            rec["intent"] = z["t%d_intent" % t]
        sc = z["t%d_scalars" % t]
        rec.update(tick=t, time=float(z["t%d_time" % t]), collisions=int(sc[0]), lock=int(sc[1]),
                   id_seq=int(sc[2]), passed=int(sc[3]), passed_step_total=int(sc[4]))
        if t in self.state_ticks:
            rec["state"] = z["t%d_state" % t]
            rec["act7"] = z["t%d_act7" % t]
        else:
            rec["state"] = None
            rec["act7"] = None
        return rec


def check_against_golden(case, t, rec, ftol=1e-9, dtol=1e-9):
    """Digest check on every tick (ints + CRC exact, float sums within ftol relative-to-scale) and a
    field-by-field check on the dense ticks."""
    di, df = digest(rec)
    gi, gf = case.dig_i[t], case.dig_f[t]
    for k, col in enumerate(DIGEST_I_COLS):
        assert int(di[k]) == int(gi[k]), "%s tick %d: digest int %s: %d vs golden %d" % (
            case.name, t, col, di[k], gi[k])
    for k, col in enumerate(DIGEST_F_COLS):
        scale = max(1.0, abs(gf[k]))
        if col in ("sum_obs0",):
            scale = max(scale, abs(gf[DIGEST_F_COLS.index("sum_abs_obs0")]))
        assert abs(df[k] - gf[k]) <= ftol * scale, "%s tick %d: digest float %s: %r vs golden %r" % (
            case.name, t, col, df[k], gf[k])
    if t in case.dense_ticks:
        gold = case.dense_record(t)
        compare_records(gold, rec, tol=dtol, label=case.name + "/dense")
        if "intent" in gold and rec.get("intent") is not None:
            assert np.array_equal(gold["intent"], rec["intent"]), "%s tick %d: intent differs" % (case.name, t)


def replay_case(case, env, ticks=None, ftol=1e-9, dtol=1e-9, want_state=True):
    """env: object with alive_view() -> (ids, ctl, obs0) and tick(actions, want_state) -> record."""
    n = case.ticks if ticks is None else min(ticks, case.ticks)
    for t in range(n):
        vid, ctl, obs0 = env.alive_view()
        acts = case.policy(t, vid, ctl, obs0)
        rec = env.tick(acts, want_state=(want_state and t in case.state_ticks))
        check_against_golden(case, t, rec, ftol, dtol)
    return n
