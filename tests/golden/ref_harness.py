"""Drive the *unmodified* Python reference and emit canonical tick records.

Works ONLY in the build container (needs /root/reference). Nothing here is copied
from the reference: it is imported read-only and driven through its public surface
(`TrafficInteraction(...)`, `.step`, `.scene_update`, `.delete_vehicle`, `veh_info`).

Crash guard: the reference's diagnostics loop at
traffic_interaction_scene.py:371-375 raises IndexError when lane 0 is empty and
`virtual_lane_4[0]` still holds stale indices. Just for the duration of the
`scene_update()` call we substitute a list subclass that iterates as empty (len()/[0]
unchanged, so `step` :1517 sees the real content), and restore the plain list after.
"""
import os
import sys
import types

import numpy as np

REF_DIR = "/root/reference"
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
from oracle.record import VEH_I_COLS, VEH_F_COLS  # noqa: E402


def reference_available():
    return os.path.isfile(os.path.join(REF_DIR, "traffic_interaction_scene.py"))


def import_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    if REF_DIR not in sys.path:
        sys.path.insert(0, REF_DIR)
    import traffic_interaction_scene as tis  # noqa
    if not tis.__file__.startswith(REF_DIR):
        raise RuntimeError("wrong traffic_interaction_scene imported: %s" % tis.__file__)
    return tis


def load_stream(name):
    import scipy.io as scio
    path = os.path.join(REF_DIR, "data", "test", "arvTimeNewVeh_new_%s_12.mat" % name)
    return np.ascontiguousarray(scio.loadmat(path)["arvTimeNewVeh"], dtype=np.float64)


class _SilentIter(list):
    def __iter__(self):
        return iter(())


def default_args():
    return types.SimpleNamespace(collision_thr=2, o_agent_num=6, c_mode="closer")


class ObjRunner:
    """Drives ANY object that has the reference's public surface (the reference itself, or the
    product's drop-in class) with the caller protocol of main.py:398-441; `tick()` returns the
    canonical record (snapshot before delete_vehicle) and then compacts."""

    def __init__(self, env, policy, want_state=False, guard=False):
        self.env = env
        self.policy = policy
        self.want_state = want_state
        self.guard = guard
        self.tick_no = 0
        self.guard_hits = 0
        self.tape = []          # actions fed this tick, (lane, j) order, all alive vehicles
        self._nbr_log = []
        self.nl = int(getattr(env, "lane_num", 12))          # physical lanes
        self.ndir = int(getattr(env, "direction_num", 12))   # virtual-lane lists (routes)
        # lane whose scene_update pass rebuilds virtual_lane_4[0] (ref :235-239)
        self.owner0 = 0
        if hasattr(env, "direction"):
            for lane in range(self.nl):
                if 0 in list(env.direction[lane]):
                    self.owner0 = lane

    def alive_view(self):
        env = self.env
        ids, ctl, obs = [], [], []
        for lane in range(self.nl):
            for veh in env.veh_info[lane]:
                ids.append(veh["id_info"][0])
                ctl.append(1 if veh["control"] else 0)
                obs.append(np.asarray(veh["state"][0], np.float64))
        return np.array(ids, np.int64), np.array(ctl, np.int32), \
            (np.stack(obs) if obs else np.zeros((0, 28)))

    def tick(self, actions=None):
        env = self.env
        veh_id, ctl, obs0 = self.alive_view()
        if actions is None:
            actions = self.policy(self.tick_no, veh_id, ctl, obs0)
        actions = np.asarray(actions, np.float64)
        self.tape = actions
        k = 0
        for lane in range(self.nl):
            for ind, veh in enumerate(env.veh_info[lane]):
                a = float(actions[k]) if veh["control"] else 0
                env.step(lane, ind, a)
                k += 1
        self._nbr_log.clear()
        guarded = False
        plain = None
        if self.guard and len(env.veh_info[self.owner0]) == 0 and len(env.virtual_lane_4[0]) > 0:
            plain = env.virtual_lane_4[0]
            env.virtual_lane_4[0] = _SilentIter(plain)
            guarded = True
            self.guard_hits += 1
        try:
            out = env.scene_update()
        finally:
            if guarded:
                # lane 0 was empty so scene_update did not rebuild it (:234): put the plain list back
                env.virtual_lane_4[0] = plain
        rec = self._snapshot(out)
        env.delete_vehicle()
        self.tick_no += 1
        return rec

    def _snapshot(self, out):
        env = self.env
        ids, re_state, reward, actions, collisions, _estm, cpv, jerks, lock = out
        C = len(ids)
        rec = dict(tick=self.tick_no, time=float(env.current_time))
        rec["ids"] = np.array(ids, np.int32).reshape(C, 2)
        nbr_log = self._nbr_log if self.guard else getattr(env, "last_nbr", [])
        rec["nbr"] = np.array(nbr_log, np.int32).reshape(C, 6, 2)
        rec["reward"] = np.array([float(r) for r in reward], np.float64)
        st = np.array(re_state, np.float64).reshape(C, 7, 28)
        rec["obs0"] = np.ascontiguousarray(st[:, 0, :])
        rec["state"] = st if self.want_state else None
        rec["act7"] = np.array(actions, np.float64).reshape(C, 7) if self.want_state else None
        rec["coll_pv"] = np.array([c[0] for c in cpv], np.int32)
        rec["collisions"] = int(collisions)
        rec["lock"] = int(lock)
        rec["jerks"] = np.array([float(x) for x in jerks], np.float64)
        rec["deleted"] = np.array(env.delete_veh, np.int32).reshape(len(env.delete_veh), 2)
        vi, vf, intent = [], [], []
        for lane in range(self.nl):
            for j, v in enumerate(env.veh_info[lane]):
                d = dict(lane=lane, j=j, id=v["id_info"][0], seq=v["seq_in_lane"], vnum=v["id_info"][1],
                         control=int(bool(v["control"])), finish=int(bool(v["finish"])),
                         done=int(bool(v["Done"])), collision=int(v["collision"]), step=int(v["step"]),
                         count=int(v["count"]), lock=int(bool(v["lock"])), lock_a=int(v["lock_a"]),
                         hdr_lane=int(v["vir_header"][0]), hdr_j=int(v["vir_header"][1]))
                vi.append([d[c] for c in VEH_I_COLS])
                vf.append([float(v[c]) for c in VEH_F_COLS])
                intent.append([int(v.get("intention", lane % 3)), int(v.get("route", lane))])
        rec["veh_i"] = np.array(vi, np.int32).reshape(len(vi), len(VEH_I_COLS))
        rec["veh_f"] = np.array(vf, np.float64).reshape(len(vf), len(VEH_F_COLS))
        rec["intent"] = np.array(intent, np.int32).reshape(len(intent), 2)
        rec["intention_re"] = int(getattr(env, "intention_re", 0))
        rec["id_seq"] = int(env.id_seq)
        rec["passed"] = int(env.passed_veh)
        rec["passed_step_total"] = int(env.passed_veh_step_total)
        rec["veh_num"] = np.array(env.veh_num, np.int32)
        rec["veh_rec"] = np.array(env.veh_rec, np.int32)
        heads = np.zeros((self.ndir, 3), np.int32)
        for d in range(self.ndir):
            vl = env.virtual_lane_4[d]
            if len(vl) > 0:
                heads[d] = (1, vl[0][1], vl[0][2])
            else:
                heads[d] = (0, -1, -1)
        rec["heads"] = heads
        return rec


class RefRunner(ObjRunner):
    """ObjRunner over the unmodified reference class (+ crash guard + neighbour spy)."""

    def __init__(self, arrive_time, policy, want_state=False, **ctor_kw):
        tis = import_reference()
        # the constructor's positional dis_ctl (ref :21; callers pass 150, main.py:394) and args.collision_thr (ref :32)
        ctor_kw = dict(ctor_kw)
        dis_ctl = ctor_kw.pop("dis_ctl", 150)
        args = default_args()
        args.collision_thr = ctor_kw.pop("collision_thr", args.collision_thr)
        env = tis.TrafficInteraction(arrive_time, dis_ctl, args, show_col=False,
                                     virtual_l=True, lane_num=12, **ctor_kw)
        ObjRunner.__init__(self, env, policy, want_state, guard=True)
        orig = env.virtual_lane_search_closer

        def spy(i, j, vl4, mode="front", veh_num=3):
            orig(i, j, vl4, mode=mode, veh_num=veh_num)
            self._nbr_log.append([list(c) for c in env.closer_cars])
        env.virtual_lane_search_closer = spy


class _ChoiceShim:
    """Stands in for the `random` module inside the reference (ref :381, :390): seed() is a no-op and
    randint(0, 1) replays `choice[veh_rec[lane]][lane]` for the lane that is spawning."""

    def __init__(self, choice):
        self.choice = np.asarray(choice, np.int64)
        self.lane = 0
        self.rec = 0
        self.draws = 0

    def seed(self, *a, **k):
        pass

    def randint(self, a, b):
        assert (a, b) == (0, 1)
        self.draws += 1
        return int(self.choice[self.rec][self.lane])


class GeoRefRunner(ObjRunner):
    """ObjRunner over the unmodified reference class with lane_num 4 or 8 (or 12).  A thin subclass tells the
    random shim which lane is spawning; everything else is the reference's own code."""

    def __init__(self, arrive_time, lane_num, policy, choice=None, want_state=False, **ctor_kw):
        tis = import_reference()
        arrive_time = np.asarray(arrive_time, np.float64)
        if choice is None:
            choice = np.zeros(arrive_time.shape, np.int64)
        shim = _ChoiceShim(choice)
        self.shim = shim

        class _Spy(tis.TrafficInteraction):
            def add_new_veh(self_env, i):
                shim.lane = i
                shim.rec = self_env.veh_rec[i]
                return tis.TrafficInteraction.add_new_veh(self_env, i)

        self._tis = tis
        self._saved_random = tis.random
        tis.random = shim
        env = _Spy(arrive_time, 150, default_args(), show_col=False, virtual_l=True, lane_num=lane_num, **ctor_kw)
        ObjRunner.__init__(self, env, policy, want_state, guard=True)
        orig = env.virtual_lane_search_closer

        def spy(i, j, vl4, mode="front", veh_num=3):
            orig(i, j, vl4, mode=mode, veh_num=veh_num)
            self._nbr_log.append([list(c) for c in env.closer_cars])
        env.virtual_lane_search_closer = spy

    def close(self):
        self._tis.random = self._saved_random
