"""ctypes binding of libpveenv.so (include/pve_env.h).

The library is the hand-written HIP implementation; there is NO CPU fallback: if the shared
object is missing or no AMD GPU is visible the calls fail loudly (PveError).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_NAME = "libpveenv.so"
LIB_PATH = os.path.join(_HERE, LIB_NAME)

PVE_LANES = 12
PVE_MAX_DIRS = 16
DIR_NUM = {4: 12, 8: 16, 12: 12}     # virtual-lane lists per layout (ref :86, :132, :167)
CFG_GENERAL_PATH = 0x1
CFG_OBS_F32 = 0x2
CFG_GEO_SCAN = 0x4
CFG_ACTOR_F32 = 0x8
PVE_OBS_WIDTH = 28
PVE_NBR = 6
PVE_N_METRICS = 12
PVE_ENV_OUT_N = 8
PVE_ACTOR_N_WEIGHTS = 6393
ABI_VERSION = 8
SRC_ZERO, SRC_POOL, SRC_ACTOR, SRC_TABLE = 0, 1, 2, 3

F_ALIVE, F_CTL, F_DONE, F_DELETED, F_FINISHED, F_LOCK = 0x01, 0x02, 0x04, 0x08, 0x10, 0x20
F_INTENT_SHIFT = 6
META_CONTROL, META_FINISH, META_DONE, META_LOCK = 0x1, 0x2, 0x4, 0x8
METRIC_NAMES = ("slot_steps", "alive_steps", "ctl_steps", "spawned", "passed", "collided", "locks",
                "sum_reward", "sum_jerk", "passed_steps", "overflow", "ticks")
ENV_OUT_NAMES = ("n_pre", "n_ctl", "collisions", "lock", "n_deleted", "n_finished", "n_spawned", "n_post")


class PveError(RuntimeError):
    pass


class PveConfig(C.Structure):
    _fields_ = [("deltaT", C.c_double), ("vm", C.c_double), ("vM", C.c_double), ("am", C.c_double),
                ("aM", C.c_double), ("v0", C.c_double), ("lane_cw", C.c_double), ("dis_ctl", C.c_double),
                ("collision_thr", C.c_double), ("lane_num", C.c_int32), ("flags", C.c_int32)]


class PveOutputs(C.Structure):
    _fields_ = [("obs_post", C.c_void_p), ("obs_pre", C.c_void_p), ("state_pre", C.c_void_p),
                ("obs_prev_post", C.c_void_p), ("reward", C.c_void_p), ("flags", C.c_void_p),
                ("lanej", C.c_void_p), ("nbr", C.c_void_p), ("new_slot", C.c_void_p), ("env_out", C.c_void_p)]


class PveRollout(C.Structure):
    _fields_ = [("n_ticks", C.c_int32), ("source", C.c_int32), ("pool", C.c_void_p), ("n_pool", C.c_int32),
                ("pool_tick0", C.c_int32), ("actor_weights", C.c_void_p), ("actor_obs", C.c_void_p),
                ("actor_actions", C.c_void_p), ("trajectory", C.c_int32), ("table_ids", C.c_int32), ("chunk_ticks", C.c_int32),
                ("persistent", C.c_int32)]


class PveVehicle(C.Structure):
    _fields_ = [(n, C.c_double) for n in ("p", "v", "a", "jerk", "jerk_sum", "vir_dis", "closer_p")] + \
               [(n, C.c_int32) for n in ("lane", "j", "id", "vnum", "seq_in_lane", "control", "finish", "done",
                                         "collision", "step", "count", "lock", "lock_a")] + \
               [("vir_header", C.c_int32 * 2), ("intention", C.c_int32), ("route", C.c_int32)]


class PveEnvInfo(C.Structure):
    _fields_ = [("current_time", C.c_double), ("n_alive", C.c_int32), ("lane_count", C.c_int32 * 12),
                ("veh_rec", C.c_int32 * 12), ("id_seq", C.c_int32), ("passed_veh", C.c_int32),
                ("passed_veh_step_total", C.c_int32), ("head_valid", C.c_int32 * 16),
                ("head_lane", C.c_int32 * 16), ("head_j", C.c_int32 * 16), ("overflow", C.c_int32),
                ("intention_re", C.c_int32)]


EXPORTS = ("pve_abi_version", "pve_last_error", "pve_default_config", "pve_workspace_bytes", "pve_create",
           "pve_destroy", "pve_set_stream", "pve_set_arrivals", "pve_reset", "pve_step_all",
           "pve_scene_update", "pve_compact", "pve_read_env", "pve_read_vehicles", "pve_get_metrics",
           "pve_state_field", "pve_synchronize", "pve_debug_phase_cycles", "pve_actor_forward",
           "pve_step_all_actor", "pve_debug_traffic_probe", "pve_set_intentions", "pve_step_many", "pve_set_actor",
           "pve_debug_stop_phase", "pve_debug_last_launch", "pve_debug_item_schedule")
LAUNCH_NONE, LAUNCH_TICK, LAUNCH_RESIDENT, LAUNCH_PERSISTENT = 0, 1, 2, 3      # pve_debug_last_launch


def _declare(L):
    vp = C.c_void_p
    L.pve_abi_version.restype = C.c_int
    L.pve_last_error.restype = C.c_char_p
    L.pve_default_config.argtypes = [C.POINTER(PveConfig)]
    L.pve_workspace_bytes.restype = C.c_size_t
    L.pve_workspace_bytes.argtypes = [C.c_int, C.c_int]
    L.pve_create.argtypes = [C.POINTER(PveConfig), C.c_int, C.c_int, C.c_int, vp, vp, C.POINTER(vp)]
    L.pve_destroy.argtypes = [vp]
    L.pve_set_stream.argtypes = [vp, vp]
    L.pve_set_arrivals.argtypes = [vp, vp, C.c_int, C.c_int]
    L.pve_set_intentions.argtypes = [vp, vp, C.c_int, C.c_int]
    L.pve_reset.argtypes = [vp]
    L.pve_step_all.argtypes = [vp, vp, C.POINTER(PveOutputs)]
    L.pve_scene_update.argtypes = [vp, vp, C.POINTER(PveOutputs)]
    L.pve_compact.argtypes = [vp, vp]
    L.pve_read_env.argtypes = [vp, C.c_int, C.POINTER(PveEnvInfo)]
    L.pve_read_vehicles.argtypes = [vp, C.c_int, C.POINTER(PveVehicle), C.c_int, C.POINTER(C.c_int)]
    L.pve_get_metrics.argtypes = [vp, C.POINTER(C.c_double)]
    L.pve_state_field.argtypes = [vp, C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int)]
    L.pve_synchronize.argtypes = [vp]
    L.pve_debug_phase_cycles.argtypes = [vp, vp]
    L.pve_debug_stop_phase.argtypes = [vp, C.c_int]
    L.pve_debug_last_launch.argtypes = [vp]
    L.pve_debug_item_schedule.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32)]
    L.pve_debug_traffic_probe.argtypes = [vp, vp]
    L.pve_set_actor.argtypes = [vp, vp]
    L.pve_actor_forward.argtypes = [vp, vp, vp, vp]
    L.pve_step_all_actor.argtypes = [vp, vp, vp, vp, C.POINTER(PveOutputs)]
    L.pve_step_many.argtypes = [vp, C.POINTER(PveRollout), C.POINTER(PveOutputs)]
    for name in EXPORTS:
        if name not in ("pve_last_error", "pve_workspace_bytes", "pve_default_config"):
            getattr(L, name).restype = C.c_int
    L.pve_default_config.restype = None
    return L


_cached = None


def load_library(path=None):
    """Load libpveenv.so (built in-tree by __graft_entry__.build() / `make -C csrc`)."""
    global _cached
    if path is None and _cached is not None:
        return _cached
    p = path or os.environ.get("PVE_LIBRARY_PATH") or LIB_PATH   # env override: profiling builds
    if not os.path.isfile(p):
        raise PveError("%s not found: build the HIP library first (python -c 'import __graft_entry__ as g; "
                       "g.build()' or make -C %s/csrc). There is no CPU fallback." % (p, _HERE))
    L = _declare(C.CDLL(p))
    if L.pve_abi_version() != ABI_VERSION:
        raise PveError("ABI mismatch: library %d, binding %d" % (L.pve_abi_version(), ABI_VERSION))
    if path is None:
        _cached = L
    return L


def check(L, rc, what=""):
    if rc != 0:
        msg = L.pve_last_error()
        raise PveError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
