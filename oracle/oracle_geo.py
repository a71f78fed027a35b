"""ctypes wrapper of the general-geometry CPU oracle (oracle/pve_oracle_geo.c; lane_num 4 / 8 / 12).
TEST INFRASTRUCTURE ONLY -- same import rules as oracle/oracle.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from .record import VEH_I_COLS, VEH_F_COLS

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpve_oracle_geo.so")
_lib = None

DIR_NUM = {4: 12, 8: 16, 12: 12}


class PvgParams(C.Structure):
    _fields_ = [(n, C.c_double) for n in
                ("deltaT", "vm", "vM", "am", "aM", "v0", "lane_cw", "dis_ctl", "collision_thr")]


def build(force=False):
    src = os.path.join(_HERE, "pve_oracle_geo.c")
    if force or not os.path.isfile(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libpve_oracle_geo.so"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    L.pvg_default_params.argtypes = [C.POINTER(PvgParams)]
    L.pvg_create.restype = vp
    L.pvg_create.argtypes = [dp, ip, C.c_int, C.c_int, C.POINTER(PvgParams)]
    L.pvg_destroy.argtypes = [vp]
    L.pvg_step.argtypes = [vp, C.c_int, C.c_int, C.c_double]
    L.pvg_scene_update.argtypes = [vp]
    L.pvg_delete_vehicle.argtypes = [vp]
    L.pvg_tick_actions.argtypes = [vp, dp]
    for name in ("pvg_n_ctl", "pvg_collisions", "pvg_lock", "pvg_n_jerks", "pvg_n_deleted",
                 "pvg_ref_would_raise", "pvg_n_alive", "pvg_lane_num", "pvg_dir_num"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp]
    for name in ("pvg_ids", "pvg_nbr", "pvg_coll_pv", "pvg_deleted"):
        getattr(L, name).restype = ip
        getattr(L, name).argtypes = [vp]
    for name in ("pvg_reward", "pvg_state", "pvg_jerks"):
        getattr(L, name).restype = dp
        getattr(L, name).argtypes = [vp]
    L.pvg_time.restype = C.c_double
    L.pvg_time.argtypes = [vp]
    L.pvg_export_vehicles.argtypes = [vp, ip, dp, dp, ip]
    L.pvg_export_env.argtypes = [vp, ip]
    L.pvg_get_p.argtypes = [vp, C.c_double, C.c_int, C.c_int, dp]
    L.pvg_get_virtual_distance.restype = C.c_int
    L.pvg_get_virtual_distance.argtypes = [vp, C.c_int, C.c_int, C.c_double, dp]
    L.pvg_run_pool.restype = C.c_long
    L.pvg_run_pool.argtypes = [vp, C.c_int, dp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_long)]
    _lib = L
    return L


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros((0,), dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


class OracleGeoEnv:
    """Single environment of `lane_num` physical lanes; same call protocol as the reference object.
    `choice` ([rows, lane_num] of 0/1) replaces the reference's random.randint(0, 1) draws (8-lane)."""

    def __init__(self, arrive_time, lane_num, choice=None, **params):
        L = lib()
        self._L = L
        prm = PvgParams()
        L.pvg_default_params(C.byref(prm))
        for k, v in params.items():
            if not hasattr(prm, k):
                raise TypeError("unknown parameter %s" % k)
            setattr(prm, k, float(v))
        arr = np.ascontiguousarray(arrive_time, dtype=np.float64)
        assert arr.ndim == 2 and arr.shape[1] == lane_num
        ch = None
        if choice is not None:
            ch = np.ascontiguousarray(choice, dtype=np.int32)
            assert ch.shape == arr.shape
        self.lane_num = int(lane_num)
        self.dir_num = DIR_NUM[self.lane_num]
        self._h = L.pvg_create(arr.ctypes.data_as(C.POINTER(C.c_double)),
                               ch.ctypes.data_as(C.POINTER(C.c_int)) if ch is not None else None,
                               arr.shape[0], self.lane_num, C.byref(prm))
        if not self._h:
            raise ValueError("lane_num must be 4, 8 or 12")
        self.tick_no = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.pvg_destroy(self._h)
            self._h = None

    def step(self, lane, ind, a):
        self._L.pvg_step(self._h, int(lane), int(ind), float(a))

    def scene_update(self):
        self._L.pvg_scene_update(self._h)

    def delete_vehicle(self):
        self._L.pvg_delete_vehicle(self._h)

    @property
    def n_alive(self):
        return self._L.pvg_n_alive(self._h)

    @property
    def current_time(self):
        return self._L.pvg_time(self._h)

    @property
    def ref_would_raise(self):
        return self._L.pvg_ref_would_raise(self._h)

    def vehicles(self):
        n = self.n_alive
        vi = np.zeros((n, len(VEH_I_COLS)), np.int32)
        vf = np.zeros((n, len(VEH_F_COLS)), np.float64)
        obs0 = np.zeros((n, 28), np.float64)
        intent = np.zeros((n, 2), np.int32)
        if n:
            ip, dp = C.POINTER(C.c_int), C.POINTER(C.c_double)
            self._L.pvg_export_vehicles(self._h, vi.ctypes.data_as(ip), vf.ctypes.data_as(dp),
                                        obs0.ctypes.data_as(dp), intent.ctypes.data_as(ip))
        return vi, vf, obs0, intent

    def alive_view(self):
        vi, _vf, obs0, _ = self.vehicles()
        return vi[:, 2].astype(np.int64), vi[:, 5].copy(), obs0

    def get_p(self, p, lane, intention):
        out = np.zeros(2, np.float64)
        self._L.pvg_get_p(self._h, float(p), int(lane), int(intention), out.ctypes.data_as(C.POINTER(C.c_double)))
        return out

    def get_virtual_distance(self, lane1, lane2, p1):
        vd = C.c_double(0.0)
        ok = self._L.pvg_get_virtual_distance(self._h, lane1, lane2, float(p1), C.byref(vd))
        return (vd.value if ok else None)

    def tick(self, actions, want_state=False):
        L, h = self._L, self._h
        actions = np.ascontiguousarray(actions, dtype=np.float64)
        assert actions.shape[0] == self.n_alive
        L.pvg_tick_actions(h, actions.ctypes.data_as(C.POINTER(C.c_double)))
        rec = self.snapshot(want_state)
        L.pvg_delete_vehicle(h)
        self.tick_no += 1
        return rec

    def snapshot(self, want_state=False):
        L, h = self._L, self._h
        Cn = L.pvg_n_ctl(h)
        nl, nd_ = self.lane_num, self.dir_num
        rec = dict(tick=self.tick_no, time=L.pvg_time(h))
        rec["ids"] = _arr(L.pvg_ids(h), Cn * 2, np.int32).reshape(Cn, 2)
        rec["nbr"] = _arr(L.pvg_nbr(h), Cn * 12, np.int32).reshape(Cn, 6, 2)
        rec["reward"] = _arr(L.pvg_reward(h), Cn, np.float64)
        st = _arr(L.pvg_state(h), Cn * 196, np.float64).reshape(Cn, 7, 28)
        rec["obs0"] = np.ascontiguousarray(st[:, 0, :])
        rec["state"] = st if want_state else None
        rec["act7"] = np.ascontiguousarray(st[:, :, 2]) if want_state else None
        rec["coll_pv"] = _arr(L.pvg_coll_pv(h), Cn, np.int32)
        rec["collisions"] = L.pvg_collisions(h)
        rec["lock"] = L.pvg_lock(h)
        nj = L.pvg_n_jerks(h)
        rec["jerks"] = _arr(L.pvg_jerks(h), nj, np.float64)
        nd = L.pvg_n_deleted(h)
        rec["deleted"] = _arr(L.pvg_deleted(h), nd * 2, np.int32).reshape(nd, 2)
        vi, vf, _, intent = self.vehicles()
        rec["veh_i"], rec["veh_f"], rec["intent"] = vi, vf, intent
        ev = np.zeros(4 + 2 * nl + 3 * nd_, np.int32)
        L.pvg_export_env(h, ev.ctypes.data_as(C.POINTER(C.c_int)))
        rec["id_seq"], rec["passed"], rec["passed_step_total"] = int(ev[0]), int(ev[1]), int(ev[2])
        rec["intention_re"] = int(ev[3])
        rec["veh_num"] = ev[4:4 + nl].copy()
        rec["veh_rec"] = ev[4 + nl:4 + 2 * nl].copy()
        rec["heads"] = ev[4 + 2 * nl:].reshape(nd_, 3).copy()
        return rec

    def run_pool(self, ticks, pool, tick0=0):
        """Timing loop entirely in C with a per-slot action pool [n_pool, cap] (bench.py cpu_baseline).
        -> (alive_steps, ctl_steps)"""
        pool = np.ascontiguousarray(pool, dtype=np.float64)
        ctl = C.c_long(0)
        alive = self._L.pvg_run_pool(self._h, int(ticks), pool.ctypes.data_as(C.POINTER(C.c_double)),
                                     pool.shape[0], pool.shape[1], int(tick0), C.byref(ctl))
        self.tick_no += ticks
        return int(alive), int(ctl.value)
