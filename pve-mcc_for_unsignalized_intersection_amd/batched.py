"""BatchedIntersections: thousands of independent intersections per GPU (12-lane fast path; 4- / 8-lane layouts
through the general-geometry kernel).

Python host code over the C ABI (include/pve_env.h); torch is used only for device memory and
the stream.  One fused HIP kernel launch per tick = the reference's
`for lane, ind: env.step(lane, ind, a)` + `env.scene_update()` + `env.delete_vehicle()`
(reference traffic_interaction_scene.py:1501, :222, :435; caller protocol main.py:398-441).
"""
import ctypes as C

import numpy as np
import torch

from . import _capi
from ._capi import PveConfig, PveEnvInfo, PveOutputs, PveRollout, PveVehicle, PveError, check

ALL_OUTPUTS = ("obs_post", "obs_pre", "reward", "flags", "lanej", "nbr", "new_slot", "env_out", "state_pre")
DEFAULT_OUTPUTS = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out")


def make_config(lib, **kw):
    cfg = PveConfig()
    lib.pve_default_config(C.byref(cfg))
    for k, v in kw.items():
        if not hasattr(cfg, k):
            raise TypeError("unknown config field %r" % k)
        setattr(cfg, k, type(getattr(cfg, k))(v))
    return cfg


class BatchedIntersections:
    """n_envs environments x `capacity` vehicle slots, state resident in HBM (struct of arrays).

    arrivals: float64 array/tensor [rows, lane_num] (shared by all envs) or [n_envs, rows, lane_num]; the
              reference's `arrive_time` matrix (main.py:388-389). Pad with +inf.
    outputs:  names of per-tick output buffers to allocate (see include/pve_env.h `pve_outputs`).
    intentions: lane_num = 8 only -- int array [rows, 8] / [n_envs, rows, 8] of 0/1: the reference's
              random.randint(0, 1) draws (ref :390) as an input stream (None = zeros).
    config:   reference constructor arguments (lane_num = 12 | 8 | 4, vm, ...); general_path=True runs the
              general-geometry kernel for lane_num = 12 too (cross-checks).
    obs_dtype: torch.float64 (reference parity layout) or torch.float32 (the actor's input type: the largest output
              shrinks by half; fused ticks only, no obs_pre / state_pre).
    """

    def __init__(self, n_envs, capacity, arrivals, device=None, outputs=DEFAULT_OUTPUTS, stream=None,
                 intentions=None, obs_dtype=torch.float64, _lib=None, **config):
        if device is None:
            device = "cuda"
        self.device = torch.device(device)
        if _lib is None:
            if self.device.type != "cuda":
                raise PveError("BatchedIntersections runs on an AMD GPU only (device=%s); there is no CPU path"
                               % self.device)
            _lib = _capi.load_library()
        self.lib = _lib
        self.n_envs, self.capacity = int(n_envs), int(capacity)
        if config.pop("general_path", False):
            config["flags"] = int(config.get("flags", 0)) | _capi.CFG_GENERAL_PATH
        if config.pop("geo_scan", False):
            config["flags"] = int(config.get("flags", 0)) | _capi.CFG_GEO_SCAN
        if config.pop("actor_f32", False):
            config["flags"] = int(config.get("flags", 0)) | _capi.CFG_ACTOR_F32
        if obs_dtype not in (torch.float64, torch.float32):
            raise TypeError("obs_dtype must be torch.float64 or torch.float32")
        self.obs_dtype = obs_dtype
        if obs_dtype == torch.float32:
            config["flags"] = int(config.get("flags", 0)) | _capi.CFG_OBS_F32
        self.cfg = make_config(self.lib, **config)
        self.lane_num = int(self.cfg.lane_num)
        self.dir_num = _capi.DIR_NUM.get(self.lane_num, 12)
        nbytes = self.lib.pve_workspace_bytes(self.n_envs, self.capacity)
        if nbytes == 0:
            raise PveError("invalid n_envs/capacity (capacity must be 64 or 128)")
        self.workspace = torch.zeros(nbytes, dtype=torch.uint8, device=self.device)
        dev_index = self.device.index if self.device.index is not None else (
            torch.cuda.current_device() if self.device.type == "cuda" else 0)
        self._stream_obj = stream
        sptr = self._stream_ptr()
        h = C.c_void_p()
        check(self.lib, self.lib.pve_create(C.byref(self.cfg), self.n_envs, self.capacity, dev_index,
                                            C.c_void_p(self.workspace.data_ptr()), sptr, C.byref(h)),
              "pve_create")
        self._h = h
        self.set_arrivals(arrivals)
        self.intentions = None
        if intentions is not None:
            self.set_intentions(intentions)
        E, K = self.n_envs, self.capacity
        dev = self.device
        self.out = {}
        names = set(outputs)
        if "state_pre" in names:
            names.update(("obs_pre", "obs_post"))
        self._obs = None
        self._obs_cur = 0
        if "obs_post" in names:
            # state_pre reads the rows the previous tick stored (stale neighbour rows, ref :1332): ping-pong pair.
            # Otherwise ONE buffer: the tick never reads observations, and 117 MB less working set per 4096 envs
            # keeps more of the tick's traffic in the 256 MB Infinity Cache (57.2 vs 59.1 us per tick).
            self._obs = [torch.zeros(E, K, 28, dtype=obs_dtype, device=dev)
                         for _ in range(2 if "state_pre" in names else 1)]
            self._obs_cur = 0
        shapes = dict(obs_pre=((E, K, 28), obs_dtype), state_pre=((E, K, 7, 28), obs_dtype),
                      reward=((E, K), torch.float64), flags=((E, K), torch.int32), lanej=((E, K), torch.int32),
                      nbr=((E, K, 6), torch.int32), new_slot=((E, K), torch.int32), env_out=((E, 8), torch.int32))
        for n in names:
            if n == "obs_post":
                continue
            if n not in shapes:
                raise TypeError("unknown output %r" % n)
            shp, dt = shapes[n]
            self.out[n] = torch.zeros(shp, dtype=dt, device=dev)
        self._zero_actions = torch.zeros(E, K, dtype=torch.float64, device=dev)
        self.ticks = 0
        self._is_reset = False

    # ------------------------------------------------------------------ plumbing
    def _stream_ptr(self):
        if self._stream_obj is not None:
            return C.c_void_p(self._stream_obj.cuda_stream)
        if self.device.type == "cuda":
            return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        return C.c_void_p(0)

    def close(self):
        if getattr(self, "_h", None) is not None and self._h:
            self.lib.pve_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_arrivals(self, arrivals):
        a = torch.as_tensor(np.asarray(arrivals) if not torch.is_tensor(arrivals) else arrivals,
                            dtype=torch.float64)
        rows, stride_rows = self._stream_shape(a, "arrivals")
        self.arrivals = a.contiguous().to(self.device)
        check(self.lib, self.lib.pve_set_arrivals(self._h, C.c_void_p(self.arrivals.data_ptr()), rows, stride_rows),
              "pve_set_arrivals")
        self._is_reset = False

    def _stream_shape(self, a, what):
        ln = self.lane_num
        if a.dim() == 2:
            assert a.shape[1] == ln, "%s must be [rows, %d]" % (what, ln)
            return a.shape[0], 0
        assert a.dim() == 3 and a.shape[0] == self.n_envs and a.shape[2] == ln, \
            "%s must be [rows,%d] or [n_envs,rows,%d]" % (what, ln, ln)
        return a.shape[1], a.shape[1]

    def set_intentions(self, choice):
        """lane_num = 8: the 0/1 intention draws (ref :390), same shape as the arrival stream."""
        c = torch.as_tensor(np.asarray(choice) if not torch.is_tensor(choice) else choice, dtype=torch.int32)
        rows, stride_rows = self._stream_shape(c, "intentions")
        self.intentions = c.contiguous().to(self.device)
        check(self.lib, self.lib.pve_set_intentions(self._h, C.c_void_p(self.intentions.data_ptr()), rows, stride_rows),
              "pve_set_intentions")
        self._is_reset = False

    def state_field(self, name):
        """Zero-copy [n_envs, capacity] view of a persistent per-slot field (see pve_state_field)."""
        p = C.c_void_p()
        eb = C.c_int()
        check(self.lib, self.lib.pve_state_field(self._h, name.encode(), C.byref(p), C.byref(eb)), "pve_state_field")
        off = p.value - self.workspace.data_ptr()
        n = self.n_envs * self.capacity * eb.value
        dt = torch.float64 if eb.value == 8 else torch.int32
        return self.workspace[off:off + n].view(dt).view(self.n_envs, self.capacity)

    @property
    def obs(self):
        """[n_envs, capacity, 28] observation (state row 0) of the vehicle now in each slot."""
        return self._obs[self._obs_cur]

    def control_mask(self):
        return (self.state_field("meta") & _capi.META_CONTROL) != 0

    # ------------------------------------------------------------------ reference-shaped calls
    def reset(self):
        """New episode: the reference builds a new TrafficInteraction (main.py:394) whose constructor
        warms up until the first vehicle exists (ref :214-220)."""
        self.lib.pve_set_stream(self._h, self._stream_ptr())
        check(self.lib, self.lib.pve_reset(self._h), "pve_reset")
        if self._obs is not None:
            for o in self._obs:
                o.zero_()
        for o in self.out.values():       # per-slot outputs of empty slots are not rewritten by the ticks
            o.zero_()
        self.ticks = 0
        self._is_reset = True

    def _outputs_struct(self, flip_obs):
        """pve_outputs for this call (cached per observation buffer: the host side of a tick is ~10 us of Python,
        which matters once sub-batches are pipelined on several streams)."""
        if self._obs is not None and len(self._obs) == 2 and flip_obs:
            self._obs_cur ^= 1
        cache = self.__dict__.setdefault("_out_structs", {})
        o = cache.get(self._obs_cur)
        if o is None:
            o = PveOutputs()
            if self._obs is not None:
                if len(self._obs) == 2:
                    o.obs_prev_post = self._obs[self._obs_cur ^ 1].data_ptr()
                o.obs_post = self._obs[self._obs_cur].data_ptr()
            for n, tns in self.out.items():
                setattr(o, n, tns.data_ptr())
            d = dict(self.out)
            if self._obs is not None:
                d["obs_post"] = self._obs[self._obs_cur]
            cache[self._obs_cur] = o
            self.__dict__.setdefault("_out_dicts", {})[self._obs_cur] = d
        return o

    def _bind_stream(self):
        """Kernels follow torch's current stream unless the batch was created with its own `stream`."""
        if self._stream_obj is None:
            self.lib.pve_set_stream(self._h, self._stream_ptr())

    def _own_stream(self):
        """Context in which torch's allocations / fills / copies are ordered with this handle's kernels: the handle's own
        stream when it has one (a torch side stream is non-blocking: nothing else orders a `torch.zeros` or `copy_` on
        the current stream against kernels launched on it), else a no-op (the kernels follow the current stream)."""
        import contextlib
        if self._stream_obj is not None and self.device.type == "cuda":
            return torch.cuda.stream(self._stream_obj)
        return contextlib.nullcontext()

    def step(self, actions=None):
        """One fused tick for every env. actions: float64 [n_envs, capacity] indexed by current slot
        (None = zeros). Returns the dict of output tensors (views, overwritten by the next call)."""
        a = self._zero_actions if actions is None else actions
        assert a.dtype == torch.float64 and a.is_contiguous() and a.shape == (self.n_envs, self.capacity)
        self._bind_stream()
        o = self._outputs_struct(flip_obs=True)
        rc = self.lib.pve_step_all(self._h, a.data_ptr(), C.byref(o))
        if rc != 0:
            check(self.lib, rc, "pve_step_all")
        self.ticks += 1
        return self._out_dicts[self._obs_cur]

    # ------------------------------------------------------------------ on-device MADDPG actor
    ACTOR_KEYS = ("ln0_gamma", "ln0_beta", "w1", "b1", "ln1_gamma", "ln1_beta", "w2", "b2", "ln2_gamma",
                  "ln2_beta", "w3", "b3")

    def set_actor(self, weights):
        """Pin the actor weights on the device. weights: flat float32[6393] tensor/array, or a dict with the
        TF variables LayerNorm{,_1,_2}/{gamma,beta} as ln{0,1,2}_{gamma,beta} and dense{,_1,_2}/{kernel,bias}
        as w{1,2,3}/b{1,2,3} (kernels in TF layout [in][out])."""
        if isinstance(weights, dict):
            weights = np.concatenate([np.asarray(weights[k], np.float32).ravel() for k in self.ACTOR_KEYS])
        w = torch.as_tensor(weights, dtype=torch.float32).contiguous()
        if w.numel() != _capi.PVE_ACTOR_N_WEIGHTS:
            raise PveError("actor needs %d float32 weights, got %d" % (_capi.PVE_ACTOR_N_WEIGHTS, w.numel()))
        with self._own_stream():
            self._actor_w = w.to(self.device)
            self._actor_actions = torch.zeros(self.n_envs, self.capacity, dtype=torch.float64, device=self.device)
            self._bind_stream()
            # copied into the handle's workspace (flat + packed for the matrix cores) on the handle's stream
            check(self.lib, self.lib.pve_set_actor(self._h, C.c_void_p(self._actor_w.data_ptr())), "pve_set_actor")

    def act(self):
        """actions [n_envs, capacity] = actor(obs) for the controlled slots, 0 elsewhere (device tensor)."""
        self.lib.pve_set_stream(self._h, self._stream_ptr())
        check(self.lib, self.lib.pve_actor_forward(self._h, None, C.c_void_p(self.obs.data_ptr()),
                                                   C.c_void_p(self._actor_actions.data_ptr())), "pve_actor_forward")
        return self._actor_actions

    def step_with_actor(self):
        """One closed-loop tick entirely on the device: actor(obs) -> fused tick (two launches, no host
        round trip; main.py:398-441 with the actor of model_agent_maddpg.py)."""
        self._bind_stream()
        obs_in = self._obs[self._obs_cur]
        o = self._outputs_struct(flip_obs=True)
        check(self.lib, self.lib.pve_step_all_actor(self._h, None, C.c_void_p(obs_in.data_ptr()),
                                                    C.c_void_p(self._actor_actions.data_ptr()), C.byref(o)),
              "pve_step_all_actor")
        self.ticks += 1
        return self.outputs()

    # ------------------------------------------------------------------ many ticks per call (pve_step_many)
    def set_action_pool(self, pool):
        """Action tape resident on the device: float64 [n_pool, n_envs, capacity]; tick number k since reset() uses
        pool[k % n_pool] (what `step(pool[k % n_pool])` would pass)."""
        p = torch.as_tensor(pool, dtype=torch.float64).contiguous().to(self.device)
        if p.dim() != 3 or tuple(p.shape[1:]) != (self.n_envs, self.capacity):
            raise PveError("action pool must be [n_pool, %d, %d]" % (self.n_envs, self.capacity))
        self._pool = p

    def set_action_table(self, table):
        """A policy that is a function of (tick, vehicle id), resident on the device: float64 [n_rows, n_ids]; at tick
        number k since reset() the vehicle with id v gets table[k % n_rows, min(v, n_ids - 1)] (controlled vehicles; the
        others 0, main.py:401) -- e.g. the sin tape of BASELINE.md 3, float32(sin(0.37 id + 0.05 tick)).
        step_many(source="table") gathers it inside the resident kernel; actions_from_table() gives the same per tick."""
        t = torch.as_tensor(table, dtype=torch.float64).contiguous().to(self.device)
        if t.dim() != 2:
            raise PveError("action table must be [n_rows, n_ids]")
        self._table = t

    def actions_from_table(self):
        """[n_envs, capacity] actions of the CURRENT tick from the table (what step_many(source="table") applies next)."""
        tab = self._table
        ids = self.state_field("id").long().clamp(0, tab.shape[1] - 1)
        return tab[self.ticks % tab.shape[0]][ids].contiguous()

    _TRAJ_SHAPES = dict(reward=(), flags=(), lanej=(), new_slot=(), nbr=(6,))

    def alloc_trajectory(self, n_ticks):
        """Reusable [n_ticks, ...] output buffers for step_many(trajectory=<this dict>): a trainer that collects
        roll-outs chunk by chunk hands the same buffers (or a ring of them) to every call instead of paying an allocation
        and a zero-fill of several GB per chunk.  Slots that hold no vehicle are marked by flags == 0; their other
        per-slot outputs keep whatever an earlier tick left there."""
        E, K, dev = self.n_envs, self.capacity, self.device
        traj = {}
        with self._own_stream():
            if self._obs is not None:
                traj["obs_post"] = torch.zeros(n_ticks, E, K, 28, dtype=self.obs_dtype, device=dev)
            for n, tns in self.out.items():
                traj[n] = torch.zeros((n_ticks,) + tuple(tns.shape), dtype=tns.dtype, device=dev)
        return traj

    def step_many(self, n_ticks, actor=False, source=None, trajectory=False, chunk=0, update_views=True, persistent=False):
        """n_ticks fused ticks in ONE call, the action source on the device (the reference's episode loop
        main.py:397-441 without the host in it): source = "pool" (set_action_pool), "actor" (set_actor; closed loop)
        or "zero".  Bit-identical to n_ticks step() / step_with_actor() calls.
        trajectory=False: returns the usual output dict holding the LAST tick's outputs.
        trajectory=True: returns a dict of freshly allocated [n_ticks, ...] tensors with every tick's outputs (a roll-out).
        trajectory=<dict from alloc_trajectory(m), m >= n_ticks>: the same into the caller's buffers (blocks 0 .. n_ticks-1).
        update_views=False (trajectory roll-outs only): skip the copy of the last tick into the handle's single-tick views
        (`obs`, `out`) -- not with source="actor", whose next call reads `obs`.
        persistent=True (with 0 < chunk < n_ticks): the call is ONE launch whose workgroups pull (intersection, chunk) items
        from a queue -- for a batch of at least twice as many intersections as the chip holds workgroups (4096 x 128 slots on
        one MI355X); same results.  Eligible: lane_num 12 with every source (not the exact-float32 actor), trajectory
        roll-outs and the training outputs included; lane_num 4 / 8 with every source, the training outputs with "pool" / "zero"
        (with "actor" they run in the resident kernel as chunked launches); anything else runs as chunked launches (last_launch() tells which).  With source="actor" the handle's
        `_actor_actions` scratch is the hand-off buffer between the items of an intersection."""
        n_ticks = int(n_ticks)
        if source is None:
            source = "actor" if actor else ("pool" if getattr(self, "_pool", None) is not None else "zero")
        self._bind_stream()
        ro = PveRollout()
        ro.n_ticks = n_ticks
        ro.trajectory = 1 if trajectory else 0
        ro.chunk_ticks = int(chunk)
        ro.persistent = 1 if persistent else 0
        if source == "pool":
            if getattr(self, "_pool", None) is None:
                raise PveError("step_many(source='pool'): call set_action_pool first")
            ro.source, ro.pool, ro.n_pool = _capi.SRC_POOL, self._pool.data_ptr(), self._pool.shape[0]
            ro.pool_tick0 = self.ticks % self._pool.shape[0]
        elif source == "actor":
            if getattr(self, "_actor_w", None) is None:
                raise PveError("step_many(source='actor'): call set_actor first")
            ro.source, ro.actor_weights = _capi.SRC_ACTOR, None          # (installed by set_actor)
            ro.actor_obs, ro.actor_actions = self._obs[self._obs_cur].data_ptr(), self._actor_actions.data_ptr()
            if not update_views:
                raise PveError("step_many(source='actor') reads the handle's observation view: update_views must stay True")
        elif source == "zero":
            ro.source = _capi.SRC_ZERO
        elif source == "table":
            if getattr(self, "_table", None) is None:
                raise PveError("step_many(source='table'): call set_action_table first")
            ro.source, ro.pool, ro.n_pool = _capi.SRC_TABLE, self._table.data_ptr(), self._table.shape[0]
            ro.table_ids, ro.pool_tick0 = self._table.shape[1], self.ticks % self._table.shape[0]
        else:
            raise PveError("unknown action source %r" % (source,))
        if "state_pre" in self.out and not trajectory:
            raise PveError("step_many: state_pre needs a trajectory roll-out (every tick reads the rows the previous one stored)")
        if not trajectory:
            o = self._outputs_struct(flip_obs=False)
            check(self.lib, self.lib.pve_step_many(self._h, C.byref(ro), C.byref(o)), "pve_step_many")
            self.ticks += n_ticks
            return self._out_dicts[self._obs_cur]
        o = PveOutputs()
        # the zero-fill of fresh trajectory buffers, the roll-out and the copy-back of the last tick are one ordered
        # sequence on the stream the kernels run on (the returned tensors belong to that stream: consume them there or
        # after synchronize())
        with self._own_stream():
            if isinstance(trajectory, dict):
                traj = trajectory
                want = set(self.out) | ({"obs_post"} if self._obs is not None else set())
                if set(traj) != want or any(t.shape[0] < n_ticks for t in traj.values()):
                    raise PveError("step_many: trajectory buffers must come from alloc_trajectory(m) of this batch, m >= n_ticks")
            else:
                traj = self.alloc_trajectory(n_ticks)
            for n, tns in traj.items():
                setattr(o, n, tns.data_ptr())
            if "state_pre" in self.out:      # the stale neighbour rows of the first tick: what the previous tick stored
                o.obs_prev_post = self._obs[self._obs_cur].data_ptr()
                if not update_views:
                    raise PveError("step_many with state_pre reads the handle's observation view: update_views must stay True")
            check(self.lib, self.lib.pve_step_many(self._h, C.byref(ro), C.byref(o)), "pve_step_many")
            self.ticks += n_ticks
            if n_ticks > 0 and update_views:     # the handle's single-tick views keep showing the latest tick
                if self._obs is not None:
                    self._obs[self._obs_cur].copy_(traj["obs_post"][n_ticks - 1])
                for n, tns in self.out.items():
                    tns.copy_(traj[n][n_ticks - 1])
        return traj

    def prepare_step_many(self, n_ticks, source=None, chunk=0, persistent=False):
        """A prepared pve_step_many call (trajectory=False): everything the call needs is built once; the returned callable
        re-issues it (only the position in the action pool advances) at the cost of one ctypes call -- for loops whose host
        side is measured in microseconds (RL inner loops, bench.py's timed region).  The prepared call keeps the pool / table
        tensor it was built on alive and refuses to run (PveError) once set_action_pool / set_action_table has replaced
        the source of ITS kind; a batch without a stream of its own re-binds torch's current stream on every call."""
        n_ticks = int(n_ticks)
        if source is None:
            source = "pool" if getattr(self, "_pool", None) is not None else "zero"
        if source == "actor" or "state_pre" in self.out:
            raise PveError("prepare_step_many: pool / zero sources without state_pre (use step_many)")
        ro = PveRollout()
        ro.n_ticks, ro.trajectory, ro.chunk_ticks, ro.persistent = n_ticks, 0, int(chunk), (1 if persistent else 0)
        if source == "pool":
            if getattr(self, "_pool", None) is None:
                raise PveError("prepare_step_many(source='pool'): call set_action_pool first")
            ro.source, ro.pool, ro.n_pool = _capi.SRC_POOL, self._pool.data_ptr(), self._pool.shape[0]
        elif source == "zero":
            ro.source = _capi.SRC_ZERO
        elif source == "table":
            if getattr(self, "_table", None) is None:
                raise PveError("prepare_step_many(source='table'): call set_action_table first")
            ro.source, ro.pool, ro.n_pool = _capi.SRC_TABLE, self._table.data_ptr(), self._table.shape[0]
            ro.table_ids = self._table.shape[1]
        else:
            raise PveError("unknown action source %r" % (source,))
        self._bind_stream()
        o = self._outputs_struct(flip_obs=False)
        fn, h, pro, po, n_pool = self.lib.pve_step_many, self._h, C.byref(ro), C.byref(o), max(1, int(ro.n_pool))
        src_attr = {"pool": "_pool", "table": "_table"}.get(source)
        src_tensor = getattr(self, src_attr) if src_attr else None
        follow_stream = self._stream_obj is None

        def call():
            if src_attr is not None and getattr(self, src_attr) is not src_tensor:
                raise PveError("prepared step_many call is stale: the action %s was replaced after prepare_step_many"
                               % source)
            if follow_stream:
                self._bind_stream()
            ro.pool_tick0 = self.ticks % n_pool
            rc = fn(h, pro, po)
            if rc != 0:
                check(self.lib, rc, "pve_step_many")
            self.ticks += n_ticks
        call._keep = (ro, o, src_tensor)
        return call

    def scene_update(self, actions=None):
        """Split protocol, part 1: all step() calls + scene_update(); Done vehicles stay in place."""
        a = self._zero_actions if actions is None else actions
        assert a.dtype == torch.float64 and a.is_contiguous() and a.shape == (self.n_envs, self.capacity)
        self.lib.pve_set_stream(self._h, self._stream_ptr())
        o = self._outputs_struct(flip_obs=True)
        check(self.lib, self.lib.pve_scene_update(self._h, C.c_void_p(a.data_ptr()), C.byref(o)), "pve_scene_update")
        self.ticks += 1
        return self.outputs()

    def compact(self):
        """Split protocol, part 2: delete_vehicle() (ref :435)."""
        self.lib.pve_set_stream(self._h, self._stream_ptr())
        obs = C.c_void_p(self._obs[self._obs_cur].data_ptr()) if self._obs is not None else C.c_void_p(0)
        check(self.lib, self.lib.pve_compact(self._h, obs), "pve_compact")

    def outputs(self):
        self._outputs_struct(flip_obs=False)
        return self._out_dicts[self._obs_cur]

    def synchronize(self):
        check(self.lib, self.lib.pve_synchronize(self._h), "pve_synchronize")

    def last_launch(self):
        """What the last stepping call launched: "tick" (one launch per tick), "resident" (one k_rollout launch per chunk),
        "persistent" (one launch for the call, items pulled from the work queue) or "none" (pve_debug_last_launch)."""
        k = self.lib.pve_debug_last_launch(self._h)
        if k < 0 or k > 3:                              # (a negative PVE_ERR_* must not index the tuple from its end)
            check(self.lib, k if k < 0 else -1, "pve_debug_last_launch")
        return ("none", "tick", "resident", "persistent")[k]

    # ------------------------------------------------------------------ host read-back
    def read_env(self, env=0):
        info = PveEnvInfo()
        check(self.lib, self.lib.pve_read_env(self._h, env, C.byref(info)), "pve_read_env")
        return info

    def read_vehicles(self, env=0):
        buf = (PveVehicle * self.capacity)()
        n = C.c_int()
        check(self.lib, self.lib.pve_read_vehicles(self._h, env, buf, self.capacity, C.byref(n)), "pve_read_vehicles")
        return [buf[i] for i in range(min(n.value, self.capacity))]

    def metrics(self):
        """dict of the 12 metric sums over this handle's envs since reset (SURVEY §8e vector)."""
        out = (C.c_double * _capi.PVE_N_METRICS)()
        check(self.lib, self.lib.pve_get_metrics(self._h, out), "pve_get_metrics")
        return dict(zip(_capi.METRIC_NAMES, [float(x) for x in out]))



class PipelinedIntersections:
    """`n_envs` environments as `n_sub` sub-batches, each a BatchedIntersections on its own HIP stream.

    The environments are independent, so the sub-batches never synchronise with each other: tick t+1 of one is in
    flight while tick t of another still runs.  One launch over all envs runs its workgroups in lock-step (every
    workgroup loads, computes and stores at the same time, and 4096 envs are exactly two rounds of the 2048 resident
    workgroups); two free-running populations interleave instead, the chip-wide LOAD / FIN bursts of one overlap the
    compute phases of the other: 4096 x 128 slots take 39 us per tick of ALL envs instead of 54 us (MI355X).

    Ordering is per sub-batch: `step()` enqueues on `streams[k]`; produce the actions of sub-batch k on that stream (or
    call `wait_stream()` after producing them elsewhere) and consume its outputs on that stream or after
    `synchronize()`.  This is the usual two-batch pipelining of an RL loop: policy inference for one half while the
    environment steps the other.
    """

    def __init__(self, n_envs, capacity, arrivals, n_sub=2, device=None, outputs=DEFAULT_OUTPUTS, intentions=None,
                 _lib=None, **config):
        if n_sub < 1 or n_sub > n_envs:
            raise PveError("n_sub must be in 1 .. n_envs")
        self.device = torch.device("cuda" if device is None else device)
        self.n_envs, self.capacity, self.n_sub = int(n_envs), int(capacity), int(n_sub)
        base, rem = divmod(self.n_envs, self.n_sub)
        sizes = [base + (1 if k < rem else 0) for k in range(self.n_sub)]
        self.bounds = [0]
        for z in sizes:
            self.bounds.append(self.bounds[-1] + z)
        on_gpu = self.device.type == "cuda"
        self.streams = [torch.cuda.Stream(self.device) if on_gpu else None for _ in range(self.n_sub)]

        def part(x, k):
            if x is None:
                return None
            x = torch.as_tensor(np.asarray(x) if not torch.is_tensor(x) else x)
            return x if x.dim() == 2 else x[self.bounds[k]:self.bounds[k + 1]]
        self.subs = []
        for k in range(self.n_sub):
            with self._on(k):
                self.subs.append(BatchedIntersections(sizes[k], capacity, part(arrivals, k), device=self.device,
                                                      outputs=outputs, stream=self.streams[k],
                                                      intentions=part(intentions, k), _lib=_lib, **config))
        self.synchronize()

    def _on(self, k):
        import contextlib
        return torch.cuda.stream(self.streams[k]) if self.streams[k] is not None else contextlib.nullcontext()

    def _parts(self, x):
        if x is None:
            return [None] * self.n_sub
        if isinstance(x, (list, tuple)):
            return list(x)
        return [x[self.bounds[k]:self.bounds[k + 1]] for k in range(self.n_sub)]     # contiguous row ranges

    def reset(self):
        for k, sub in enumerate(self.subs):
            with self._on(k):
                sub.reset()

    def wait_stream(self, stream=None):
        """Every sub-batch stream waits for what has been enqueued on `stream` (default: torch's current stream)."""
        if self.device.type != "cuda":
            return
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        for s in self.streams:
            s.wait_stream(stream)

    def step(self, actions=None, wait=True):
        """One fused tick of every sub-batch, each on its own stream.  actions: [n_envs, capacity] float64 (or a list of
        per-sub-batch tensors, or None).  Returns the list of per-sub-batch output dicts.
        wait=True orders every sub-batch stream after torch's current stream (where `actions` are normally produced);
        pass wait=False when the actions are known to be complete (e.g. a tape uploaded and synchronised earlier)."""
        outs = []
        for k, (sub, a) in enumerate(zip(self.subs, self._parts(actions))):
            if a is not None and self.streams[k] is not None:
                # device-produced actions (a policy forward on torch's current stream) must be complete before the
                # tick of sub-batch k reads them: an event wait, no host synchronisation
                if wait:
                    self.streams[k].wait_stream(torch.cuda.current_stream(self.device))
                a.record_stream(self.streams[k])       # the caching allocator must not recycle it under the kernel
            outs.append(sub.step(a))
        return outs

    def set_actor(self, weights):
        for k, sub in enumerate(self.subs):
            with self._on(k):
                sub.set_actor(weights)

    def step_with_actor(self):
        return [sub.step_with_actor() for sub in self.subs]

    def set_action_table(self, table):
        for k, sub in enumerate(self.subs):
            with self._on(k):
                sub.set_action_table(table)
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def set_action_pool(self, pool):
        pool = torch.as_tensor(pool, dtype=torch.float64)
        for k, sub in enumerate(self.subs):
            with self._on(k):
                sub.set_action_pool(pool[:, self.bounds[k]:self.bounds[k + 1]].contiguous())
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def alloc_trajectory(self, n_ticks):
        return [sub.alloc_trajectory(n_ticks) for sub in self.subs]

    def step_many(self, n_ticks, actor=False, source=None, trajectory=False, chunk=0, update_views=True, persistent=False):
        """n_ticks of every sub-batch, one pve_step_many call each (on its own stream).  chunk > 0 splits every call into
        launches of `chunk` ticks: a launch lasts as long as its slowest intersection, and the other sub-batches'
        workgroups fill the slots its fast ones free, so short launches keep the chip full.
        trajectory: False / True / the list alloc_trajectory() returned (one dict per sub-batch)."""
        tr = trajectory if isinstance(trajectory, (list, tuple)) else [trajectory] * self.n_sub
        return [sub.step_many(n_ticks, actor=actor, source=source, trajectory=tr[k], chunk=chunk, update_views=update_views,
                              persistent=persistent)
                for k, sub in enumerate(self.subs)]

    def prepare_step_many(self, n_ticks, source=None, chunk=0, persistent=False):
        """One prepared call per sub-batch (BatchedIntersections.prepare_step_many); the returned callable issues them all."""
        calls = [sub.prepare_step_many(n_ticks, source=source, chunk=chunk, persistent=persistent) for sub in self.subs]

        def call():
            for c in calls:
                c()
        return call

    def synchronize(self):
        for sub in self.subs:
            sub.synchronize()

    def metrics(self):
        tot = None
        for sub in self.subs:
            m = sub.metrics()
            tot = m if tot is None else {k: tot[k] + m[k] for k in m}
        return tot

    def sub_of(self, env):
        """(sub-batch index, local env index) of a global env index."""
        for k in range(self.n_sub):
            if env < self.bounds[k + 1]:
                return k, env - self.bounds[k]
        raise IndexError(env)
