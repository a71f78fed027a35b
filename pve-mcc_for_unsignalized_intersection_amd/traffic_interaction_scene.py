"""Drop-in `TrafficInteraction` for the reference's MADDPG loop (main.py:230-311, 394-441, 552-575).

Same constructor, `step(lane, ind, a)`, `scene_update()` 9-tuple, `delete_vehicle()` and
`veh_info[lane][ind]` mapping view as the reference class
(Mingtzge/PVE-MCC_for_unsignalized_intersection, traffic_interaction_scene.py:21, :1501, :222, :435),
but every tick runs in the hand-written HIP kernels of libpveenv.so through the C ABI
(pve_scene_update / pve_compact on a 1-environment batch).  There is no CPU path: without the
library or without an AMD GPU construction raises PveError.

What is NOT provided (out of scope): the matplotlib renderer `Visible`, the
plotting recorders (`virtual_data`, `choose_veh_info`, `veh_info_record` stay empty), the 3/4/8-lane
geometries (lane_num must be 12), and the reference's crash at :371-375.
"""
import numpy as np
import torch

from . import _capi
from .batched import BatchedIntersections

_OUTPUTS = ("obs_post", "obs_pre", "state_pre", "reward", "flags", "lanej", "nbr", "new_slot", "env_out")


class TrafficInteraction:
    def __init__(self, arrive_time, dis_ctl, args, deltaT=0.1, vm=5, vM=13, am=-3, aM=3, v0=10, diff_max=220,
                 lane_cw=2.5, loc_con=True, show_col=False, virtual_l=True, lane_num=12,
                 capacity=128, device=None, _lib=None):
        if lane_num != 12:
            raise _capi.PveError("only the 12-lane intersection is implemented (lane_num=12)")
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
        self.lane_info = [
            [dis_ctl - 6 * lane_cw, 3.1415 / 2 * 7 * lane_cw, -(dis_ctl - 6 * lane_cw)],
            [dis_ctl - 6 * lane_cw, 12 * lane_cw, -(dis_ctl - 6 * lane_cw)],
            [dis_ctl - 6 * lane_cw, 3.1415 / 2 * lane_cw, -(dis_ctl - 6 * lane_cw)]]
        self.deltaT = deltaT
        self.dis_control = dis_ctl
        self.diff_max = diff_max
        self.arrive_time = arrive_time
        self.choose_veh_info = [[] for _ in range(lane_num)]
        self.veh_info_record = [[] for _ in range(lane_num)]
        self.delete_veh = []
        self.virtual_lane_4 = [[] for _ in range(lane_num)]   # only the head entry [vd?, lane, j] is mirrored
        arr = np.ascontiguousarray(np.asarray(arrive_time, dtype=np.float64))
        self._b = BatchedIntersections(1, capacity, arr, device=device, outputs=_OUTPUTS, _lib=_lib,
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
        self.veh_num = list(info.lane_count)
        self.veh_rec = list(info.veh_rec)
        self.id_seq = info.id_seq
        self.passed_veh = info.passed_veh
        self.passed_veh_step_total = info.passed_veh_step_total
        for d in range(12):
            self.virtual_lane_4[d] = [[None, info.head_lane[d], info.head_j[d]]] if info.head_valid[d] else []
        lanes = [[] for _ in range(12)]
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
        ctl = (flags & _capi.F_CTL) != 0
        lanej = out["lanej"][0, :n_pre].cpu().numpy()
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
        self.delete_veh = [[int(x) >> 16, int(x) & 0xFFFF] for x in lanej[(flags & _capi.F_DELETED) != 0]]
        # veh["state"] = deep copy of the new state for the controlled vehicles (ref :288)
        ids_dev = b.state_field("id")[0].cpu().numpy()
        states = {int(ids_dev[new_slot[k]]): np.array(state[k]) for k in np.flatnonzero(ctl)}
        self._refresh(states)
        fin = np.flatnonzero((flags & _capi.F_FINISHED) != 0)
        jerks = [float(self._veh[int(ids_dev[new_slot[k]])]["jerk_sum"]) for k in fin]
        return ids, re_state, rew, actions, int(eo[2]), 0, cpv, jerks, int(eo[3])

    def delete_vehicle(self):
        """ref :435-444"""
        self._b.compact()
        self._b.synchronize()
        self._refresh()
