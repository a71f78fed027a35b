"""Drop-in `TrafficInteraction` for the reference's MADDPG loop (main.py:230-311, 394-441, 552-575).

Same constructor, `step(lane, ind, a)`, `scene_update()` 9-tuple, `delete_vehicle()` and
`veh_info[lane][ind]` mapping view as the reference class
(Mingtzge/PVE-MCC_for_unsignalized_intersection, traffic_interaction_scene.py:21, :1501, :222, :435),
but every tick runs in the hand-written HIP kernels of libpveenv.so through the C ABI
(pve_scene_update / pve_compact on a 1-environment batch).  There is no CPU path: without the
library or without an AMD GPU construction raises PveError.

What is NOT provided (out of scope): the matplotlib renderer `Visible`, the
plotting recorders (`virtual_data`, `choose_veh_info`, `veh_info_record` stay empty), the 3-lane layout (broken
upstream) and the reference's crash at :371-375.

lane_num = 12 runs the optimised kernel; lane_num = 4 / 8 (ref :66-145) run the general-geometry kernel.  For
lane_num = 8 the reference draws every new vehicle's intention with an entropy-seeded random.randint(0, 1)
(ref :381, :390); here the draws come from `intentions` ([rows, 8] of 0/1) or, when that is None, from
numpy.random.default_rng(intention_seed) -- random like the reference, but reproducible.
"""
import numpy as np
import torch

from . import _capi
from .batched import BatchedIntersections

_OUTPUTS = ("obs_post", "obs_pre", "state_pre", "reward", "flags", "lanej", "nbr", "new_slot", "env_out")


class TrafficInteraction:
    def __init__(self, arrive_time, dis_ctl, args, deltaT=0.1, vm=5, vM=13, am=-3, aM=3, v0=10, diff_max=220,
                 lane_cw=2.5, loc_con=True, show_col=False, virtual_l=True, lane_num=12,
                 capacity=128, device=None, intentions=None, intention_seed=None, _lib=None):
        if lane_num not in (12, 8, 4):
            raise _capi.PveError("lane_num must be 12, 8 or 4 (the 3-lane branch is broken upstream)")
        if not loc_con:
            raise _capi.PveError("loc_con=False is not supported")
        # attributes main.py / the renderer read (ref :28-45, :148-152, :195-213)
        self.virtual_l = virtual_l
        self.virtual_data = {}
        self.show_col = show_col
        self.loc_con = loc_con
        self.collision_thr = args.collision_thr
        self.vm, self.vM, self.am, self.aM, self.v0 = vm, vM, am, aM, v0
        self.lane_cw = lane_cw
        self.lane_num = lane_num
        self.closer_veh_num = getattr(args, "o_agent_num", 6)
        self.c_mode = getattr(args, "c_mode", "closer")
        if self.closer_veh_num != 6:
            raise _capi.PveError("o_agent_num must be 6 (the reference hard-codes 6 neighbours, ref :1324)")
        half, rl = {12: (6, 7), 8: (4, 5), 4: (2, 3)}[lane_num]           # ref :69-71, :103-105, :149-151
        self.lane_info = [
            [dis_ctl - half * lane_cw, 3.1415 / 2 * rl * lane_cw, -(dis_ctl - half * lane_cw)],
            [dis_ctl - half * lane_cw, 2 * half * lane_cw, -(dis_ctl - half * lane_cw)],
            [dis_ctl - half * lane_cw, 3.1415 / 2 * lane_cw, -(dis_ctl - half * lane_cw)]]
        if lane_num == 4:                                                 # ref :88-93
            self.direction = [[6, 7, 8], [0, 1, 2], [9, 10, 11], [3, 4, 5]]
        elif lane_num == 8:                                               # ref :135-144
            self.direction = [[0, 1, -1], [-1, 2, 3], [4, 5, -1], [-1, 6, 7],
                              [8, 9, -1], [-1, 10, 11], [12, 13, -1], [-1, 14, 15]]
        else:                                                             # ref :168-181
            self.direction = [[i if m == i % 3 else -1 for m in range(3)] for i in range(12)]
        self.direction_num = _capi.DIR_NUM[lane_num]
        self.intention_re = 0
        self.deltaT = deltaT
        self.dis_control = dis_ctl
        self.diff_max = diff_max
        self.arrive_time = arrive_time
        self.choose_veh_info = [[] for _ in range(lane_num)]
        self.veh_info_record = [[] for _ in range(lane_num)]
        self.delete_veh = []
        self.virtual_lane_4 = [[] for _ in range(self.direction_num)]   # only the head entry [vd?, lane, j] is mirrored
        arr = np.ascontiguousarray(np.asarray(arrive_time, dtype=np.float64))
        if lane_num == 8 and intentions is None:
            intentions = np.random.default_rng(intention_seed).integers(0, 2, size=arr.shape).astype(np.int32)
        self._b = BatchedIntersections(1, capacity, arr, device=device, outputs=_OUTPUTS, _lib=_lib,
                                       intentions=intentions if lane_num == 8 else None, lane_num=lane_num,
                                       deltaT=deltaT, vm=vm, vM=vM, am=am, aM=aM, v0=v0, lane_cw=lane_cw,
                                       dis_ctl=dis_ctl, collision_thr=args.collision_thr)
        self._cap = capacity
        self._actions = torch.zeros(1, capacity, dtype=torch.float64)
        self._veh = {}            # vehicle id -> persistent dict (callers mutate "buffer" / "count" in place)
        self._dev_count = {}      # vehicle id -> count as last seen on the device
        self._b.reset()           # constructor warm-up (ref :214-220)
        self._refresh()

    # ------------------------------------------------------------------ view maintenance
    def _refresh(self, states=None):
        """Rebuild veh_info[lane][ind] from the device state (one D2H read of a <= 128-slot env)."""
        b = self._b
        info = b.read_env(0)
        vehs = b.read_vehicles(0)
        self.current_time = info.current_time
        nl = self.lane_num
        self.veh_num = list(info.lane_count)[:nl]
        self.veh_rec = list(info.veh_rec)[:nl]
        self.intention_re = info.intention_re
        self.id_seq = info.id_seq
        self.passed_veh = info.passed_veh
        self.passed_veh_step_total = info.passed_veh_step_total
        for d in range(self.direction_num):
            self.virtual_lane_4[d] = [[None, info.head_lane[d], info.head_j[d]]] if info.head_valid[d] else []
        lanes = [[] for _ in range(nl)]
        seen = set()
        for slot, v in enumerate(vehs):
            vid = v.id
            seen.add(vid)
            d = self._veh.get(vid)
            if d is None:
                d = {"intention": v.intention, "buffer": [], "route": v.route, "count": 0, "action": 0,
                     "lane": v.lane, "header": False, "reward": 10, "dis_front": 50,
                     "seq_in_lane": v.seq_in_lane, "estm_collision": 0, "estm_arrive_time": 0.0,
                     "id_info": [v.id, v.vnum],
                     "state": np.zeros((7, 28))}
                self._veh[vid] = d
                self._dev_count[vid] = 0
            # caller-side adjustment of count (main.py:266 does `count -= 1`) is preserved
            adj = d["count"] - self._dev_count[vid]
            d["count"] = v.count + adj
            self._dev_count[vid] = v.count
            d["Done"] = bool(v.done)
            d["p"], d["v"], d["a"] = v.p, v.v, v.a
            d["jerk"], d["jerk_sum"] = v.jerk, v.jerk_sum
            d["lock_a"], d["lock"] = v.lock_a, bool(v.lock)
            d["vir_header"] = [v.vir_header[0], v.vir_header[1]]
            d["vir_dis"], d["closer_p"] = v.vir_dis, v.closer_p
            d["control"], d["finish"] = bool(v.control), bool(v.finish)
            d["step"], d["collision"] = v.step, v.collision
            if states is not None and vid in states:
                d["state"] = states[vid]
            lanes[v.lane].append(d)
        for vid in [k for k in self._veh if k not in seen]:
            del self._veh[vid]
            del self._dev_count[vid]
        self.veh_info = lanes
        self._lane_start = np.concatenate([[0], np.cumsum(self.veh_num)]).astype(int)

    # ------------------------------------------------------------------ reference API
    def step(self, i, j, eval_a):
        """ref :1501 -- stage the action of vehicle (lane i, index j); the kinematics of all vehicles
        are integrated (in (lane, j) order semantics) at the start of the next scene_update()."""
        self._actions[0, int(self._lane_start[i]) + int(j)] = float(eval_a)

    def scene_update(self):
        """ref :222-376 -> (ids, re_state, reward, actions, collisions, estm_collisions,
        collisions_per_veh, jerks, lock)"""
        b = self._b
        out = b.scene_update(self._actions.to(b.device))
        b.synchronize()
        self._actions.zero_()
        eo = out["env_out"][0].cpu().numpy()
        n_pre = int(eo[0])
        flags = out["flags"][0, :n_pre].cpu().numpy().astype(np.int64)
        lanej = out["lanej"][0, :n_pre].cpu().numpy().astype(np.int64)
        # scene_update reports in processing order: lane, intention, j (ref :233-275); == slot order for 12 lanes
        order = np.lexsort((lanej & 0xFFFF, (flags >> _capi.F_INTENT_SHIFT) & 3, lanej >> 16)) if n_pre else \
            np.zeros(0, np.int64)
        ctl = order[((flags & _capi.F_CTL) != 0)[order]]
        state = out["state_pre"][0, :n_pre].cpu().numpy()
        reward = out["reward"][0, :n_pre].cpu().numpy()
        new_slot = out["new_slot"][0, :n_pre].cpu().numpy()
        ids = [[int(x) >> 16, int(x) & 0xFFFF] for x in lanej[ctl]]
        re_state = [np.array(s) for s in state[ctl]]
        rew = [float(x) for x in reward[ctl]]
        actions = [[float(row[2]) for row in s] for s in state[ctl]]
        cpv = [[int(c), 0] for c in (flags[ctl] >> 8)]
        nbr = out["nbr"][0, :n_pre].cpu().numpy()[ctl]
        self.last_nbr = [[[int(x) >> 16, int(x) & 0xFFFF] if x >= 0 else [-1, -1] for x in row] for row in nbr]
        self.delete_veh = [[int(x) >> 16, int(x) & 0xFFFF]
                           for x in lanej[order[((flags & _capi.F_DELETED) != 0)[order]]]]
        # veh["state"] = deep copy of the new state for the controlled vehicles (ref :288)
        ids_dev = b.state_field("id")[0].cpu().numpy()
        states = {int(ids_dev[new_slot[k]]): np.array(state[k]) for k in ctl}
        self._refresh(states)
        fin = order[((flags & _capi.F_FINISHED) != 0)[order]]
        jerks = [float(self._veh[int(ids_dev[new_slot[k]])]["jerk_sum"]) for k in fin]
        return ids, re_state, rew, actions, int(eo[2]), 0, cpv, jerks, int(eo[3])

    def delete_vehicle(self):
        """ref :435-444"""
        self._b.compact()
        self._b.synchronize()
        self._refresh()
