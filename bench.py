#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused HIP tick at BASELINE.json's headline config
(4096 parallel 12-lane intersections x 128 vehicle slots per GPU, synthetic Poisson arrivals at
1100 veh/h/lane), weak-scaled env-parallel over N GPUs (one process per GPU, no data-path collective;
one RCCL all-gather of the metrics vector after the timed region).

  python bench.py [--gpus N] [--steps K] [--warmup W]

N > 1 without an outer launcher: bench.py starts `python -m torch.distributed.run --nproc-per-node N bench.py ...`
as a child process (before anything touches the GPU) and forwards rank 0's JSON line and the children's exit code.

Protocol (reference loop main.py:397): the intersections are first filled to steady state by an
UN-TIMED prefill (>= 300 ticks, continued until the mean population moves < 1 % over 50 ticks) whatever --warmup
says; then W un-timed warm-up steps, then EXACTLY K timed steps between barrier + synchronize pairs.  A "step" =
one fused tick (all step() calls + scene_update() + delete_vehicle()) of every env of the rank.  Inputs (arrival
streams, action pool) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

After the timed region (outside it) the line certifies itself: 8 of the envs this rank timed are replayed from reset by
the CPU oracle and compared (final state + the last tick's outputs): "verified": true / false, exit code != 0 on false.
`roofline` carries three HBM fractions (bytes the measured mode must move / SURVEY 8d's nominal 380 B / PMC counter bytes)
and `binding`, the roofline that actually limits the kernel (vector-instruction issue, from the SQ counter passes of the
same build); `retained_outputs` is the same workload with every tick's outputs kept (second timed region).
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

B_ALG_FP64 = 380.0      # algorithmic bytes per vehicle-slot-step, FP64 layout
B_ALG_OBS_F32 = 268.0   # the same with float32 observation rows (--obs-f32): 380 - 28 x 4 (SURVEY.md §8d, FP32 output)
# of which the persistent state: read 56 B (everything but the action), written 68 B (everything but reward / row /
# neighbour ids / done mask).  pve_step_many keeps the state on the chip: a launch of T ticks moves it once, not T times.
B_STATE_IN, B_STATE_OUT = 56.0, 68.0
N_SIMD, VALU_CYCLES = 1024, 4      # MI355X: 256 CUs x 4 SIMD16; one wave64 VALU instruction occupies its SIMD for 4 cycles
VERIFY_ENVS, VERIFY_TOL = 8, 1e-9
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak (guides/MI355X_MICROARCH.md)
N_POOL = 16
PREFILL_MIN, PREFILL_CHUNK, PREFILL_MAX = 300, 50, 1000
CPU_WARM, CPU_TICKS = 300, 600           # cpu_baseline sample: fixed, independent of --steps / --warmup (~1 s on 256 threads)


def action_pool(n_envs, cap, seed):
    """Synthetic action tape resident in HBM: pool[k][env][slot] = float32(sin(0.37*u + 0.05*k*7))-like
    values in [-1, 1] (the SURVEY §8d 'sin, A=1' pattern, indexed by slot so that it needs no feedback)."""
    rng = np.random.default_rng(seed)
    phase = rng.uniform(0, 2 * np.pi, size=(1, n_envs, cap))
    k = np.arange(N_POOL, dtype=np.float64)[:, None, None]
    a = np.sin(phase + 0.37 * np.arange(cap)[None, None, :] + 0.05 * 7 * k)
    return a.astype(np.float32).astype(np.float64)


def id_sin_table(rows_t):
    """BASELINE.md 3's tape as a [tick][vehicle id] table: a = float32(sin(0.37 id + 0.05 tick)); one column per id a 12-lane
    intersection can hand out in rows_t ticks (headways >= 1 s)."""
    cols_t = int(12 * (rows_t * 0.1 + 4)) + 64
    t = np.sin(0.37 * np.arange(cols_t, dtype=np.float64)[None, :] + 0.05 * np.arange(rows_t, dtype=np.float64)[:, None])
    return t.astype(np.float32).astype(np.float64)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(arr, pool, cap, lane_num=12, choice=None, id_sin=False):
    """The CPU oracle (oracle/pve_oracle.c, a plain-C port of the reference algorithm) timed on this host's cores on a
    BOUNDED sample of the same workload: the first n envs of the same arrival tensor with the same action tape, 300
    warm-up ticks (steady population) + 600 timed ticks each, independent of --steps / --warmup.  Two figures: every
    hardware thread busy (envs statically dealt to POSIX threads inside ONE C call, pvo_run_many: no Python in the timed
    loop; lane_num 4 / 8: Python threads around one C call per env), and one thread alone."""
    from oracle.oracle import OracleEnv
    cores = os.cpu_count() or 1

    def make(e):
        if lane_num == 12:
            return OracleEnv(arr[e])
        from oracle.oracle_geo import OracleGeoEnv
        return OracleGeoEnv(arr[e], lane_num, choice=None if choice is None else choice[e])

    def timed(envs, n_threads):
        if lane_num == 12:
            # ONE C call per region: the environments are dealt to POSIX threads inside the oracle library (pvo_run_many), no
            # Python and no GIL in the timed loop
            from oracle.oracle import run_many
            objs, idx = [o for _, o in envs], [e for e, _ in envs]
            kw = dict(policy=1, amp=1.0) if id_sin else dict(pool=pool)
            run_many(objs, idx, n_threads, CPU_WARM, tick0=0, **kw)
            best = None
            for rep_ in range(2):                          # (two timed regions of CPU_TICKS ticks, the faster one counts: host noise)
                t = time.perf_counter()
                alive, _ = run_many(objs, idx, n_threads, CPU_TICKS, tick0=CPU_WARM + rep_ * CPU_TICKS, **kw)
                dt = time.perf_counter() - t
                if best is None or dt < best[0]:
                    best = (dt, alive)
            return best
        res = [None] * len(envs)

        def worker(k, nt, t0):
            for i in range(k, len(envs), n_threads):
                res[i] = envs[i][1].run(nt, 1, 1.0, t0) if id_sin else envs[i][1].run_pool(nt, pool[:, envs[i][0], :], t0)

        def run(nt, t0):
            ths = [threading.Thread(target=worker, args=(k, nt, t0)) for k in range(n_threads)]
            t = time.perf_counter()
            [x.start() for x in ths]
            [x.join() for x in ths]
            return time.perf_counter() - t

        run(CPU_WARM, 0)
        dt = run(CPU_TICKS, CPU_WARM)
        return dt, sum(r[0] for r in res)

    n = min(arr.shape[0], max(cores * 4, 16))
    dt, alive = timed([(e, make(e)) for e in range(n)], min(cores, n))
    n1 = min(arr.shape[0], 32)
    dt1, alive1 = timed([(e, make(e)) for e in range(n1)], 1)
    return dict(value=n * cap * CPU_TICKS / dt, unit="env-steps/s", cores=min(cores, n), kind="port",
                cpu_model=cpu_model(), alive_steps_per_s=alive / dt,
                mean_alive_per_env=alive / float(n * CPU_TICKS),
                single_thread=dict(value=n1 * cap * CPU_TICKS / dt1, alive_steps_per_s=alive1 / dt1,
                                   sample="%d envs x %d ticks, 1 thread, %.2f s wall" % (n1, CPU_TICKS, dt1)),
                sample="%d envs x %d timed ticks (after %d un-timed warm-up ticks: steady population) of the same synthetic "
                       "workload and action pool, %d threads, %.2f s wall" % (n, CPU_TICKS, CPU_WARM, min(cores, n), dt))


# (launch-shape knobs exist in the `make knobs` build of the library only; a run with one of them set measures another kernel)
KNOBS = ("PVE_NO_ROLLOUT_KERNEL", "PVE_NO_ROLLOUT_ACTOR", "PVE_ROLLOUT_GEO_WPE5", "PVE_ACTOR_GRID",
         "PVE_LIBRARY_PATH", "PVE_NO_PERSISTENT", "PVE_TAPER_TAIL", "PVE_PERSISTENT_GRID", "PVE_ROLLOUT_WPE5")


def csrc_sha():
    """Fingerprint of the BUILD the committed counter figures apply to: the kernel sources, the Makefile (= the default
    compiler flags) and the flags the library next to the package was actually built with (csrc/Makefile records them in
    libpveenv.flags; `make EXTRA=...` variants therefore get another fingerprint).  None when a launch / library knob of
    the A-B tooling is set in the environment: such a run measures another kernel than the profiled one."""
    if any(os.environ.get(k) for k in KNOBS):
        return None
    pkg = os.path.join(ROOT, "pve-mcc_for_unsignalized_intersection_amd")
    d = os.path.join(pkg, "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".inc")) or f == "Makefile":
            h.update(open(os.path.join(d, f), "rb").read())
    flags = os.path.join(pkg, "libpveenv.flags")
    if os.path.isfile(flags):
        h.update(b"flags:" + open(flags, "rb").read().strip())
    return h.hexdigest()[:16]


def _latest_profile(pattern):
    """The newest profiles/rNN_<pattern> (files whose name does not start with rNN_ are ignored)."""
    import glob
    import re
    files = [(int(m.group(1)), f) for f in glob.glob(os.path.join(ROOT, "profiles", "r*_" + pattern))
             for m in [re.match(r"r(\d+)_", os.path.basename(f))] if m]
    return max(files)[1] if files else None


def pmc_traffic(envs_per_launch, cap, outputs, mode, ticks_per_launch, other, pkey=None):
    """HBM bytes per launch of the dominant kernel from the PMC counters (FETCH_SIZE x gfx950 correction + WRITE_SIZE).
    Counters cannot be read from inside the process, so they come from the committed rocprofv3 passes of this very
    command (tools/collect_profiles.sh -> profiles/r*_traffic.json); null unless kernel sources, config and launch
    shape are the profiled ones.  pkey: the profile of the persistent launch ("persist": items of 10 ticks, "persist_short":
    the 20-tick call in items of 6), whose bytes scale with the ticks of the call."""
    default_outputs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out")
    sha = csrc_sha()
    if other or cap != 128 or tuple(outputs) != default_outputs or sha is None:
        return None, None
    f = _latest_profile("traffic.json")
    if not f:
        return None, None
    j = json.load(open(f))
    if pkey:
        t = j.get(pkey)
        if not t or int(t.get("envs_per_launch", 0)) != envs_per_launch or t.get("csrc_sha") != sha:
            return None, None
        t = dict(t, hbm_bytes_per_launch=t["hbm_bytes_per_tick"] * ticks_per_launch, profiled_ticks_per_launch=t["ticks_per_launch"])
        return t, os.path.relpath(f, ROOT)
    t = j.get("%s%d" % (mode, ticks_per_launch)) or j.get(mode)      # (a launch shape of its own, e.g. rollout5, else the mode's)
    if not t or int(t.get("envs_per_launch", 4096)) != envs_per_launch or t.get("csrc_sha") != sha \
            or int(t.get("ticks_per_launch", 1)) != ticks_per_launch:
        return None, None
    return t, os.path.relpath(f, ROOT)


def binding_profile(mode, ticks_per_launch, cap, other, pkey=None):
    """What actually binds the dominant kernel (it is not bandwidth): instruction counts per wave and tick, LDS bank
    conflicts and wait share from the committed SQ counter passes of this very build (profiles/r*_binding.json,
    tools/collect_profiles.sh); null unless the kernel sources are the profiled ones."""
    sha = csrc_sha()
    if other or cap != 128 or sha is None:
        return None, None
    f = _latest_profile("binding.json")
    if not f:
        return None, None
    j = json.load(open(f))
    t = j.get(pkey) if pkey else (j.get("%s%d" % (mode, ticks_per_launch)) or j.get(mode))
    if not t or t.get("csrc_sha") != sha or (not pkey and int(t.get("ticks_per_launch", 1)) != ticks_per_launch):
        return None, None
    return t, os.path.relpath(f, ROOT)


def sample_envs(n_envs, n_sample, seed):
    """The envs a verification leg replays: drawn afresh for every run from `seed` (printed in the line; --verify-seed pins it),
    always with the rank's first and last env."""
    rng = np.random.default_rng(seed)
    n = min(n_sample, n_envs)
    pick = set([0, n_envs - 1][:n])
    for e in rng.permutation(n_envs):
        if len(pick) >= n:
            break
        pick.add(int(e))
    return sorted(pick)


def verify_against_oracle(locate, last_outputs, arr, pool_np, total_ticks, lane_num, choice, n_sample=VERIFY_ENVS, table_np=None,
                          skip_overflowed=False, seed=0):
    """Outside the timed region: `n_sample` of the envs this rank just timed are replayed from reset by the CPU oracle
    (the checker; oracle/README.md) on the same arrival stream and the same action pool for the same number of ticks;
    the final persistent state (ints exact, floats 1e-9) and the last tick's outputs (controlled set, rewards, collision /
    dead-lock counters) must agree.  -> dict(verified=bool, ...)."""
    from oracle.oracle import OracleEnv
    from oracle.record import close
    n_envs = arr.shape[0]
    sample = sample_envs(n_envs, n_sample, seed)
    if skip_overflowed:
        # a leg that runs ON a rate at which some intersections fill up: compare n_sample envs that never deferred a spawn --
        # candidates in the seed's order until enough are found (at most 8 x n_sample looked at)
        sample = [int(e) for e in np.random.default_rng(seed).permutation(n_envs)[:8 * n_sample]]
    res = dict(verified=True, envs=sample, seed=int(seed), ticks_replayed=int(total_ticks), tol=VERIFY_TOL, checker="oracle (CPU restatement)",
               compared="final state of every vehicle (13 int fields exact, p v a jerk jerk_sum at tol) + last tick: "
                        "controlled set, rewards, collision and dead-lock counters")
    if total_ticks < 1:
        res.update(verified=None, reason="no tick executed")
        return res
    skipped, compared = [], []
    try:
        for e in sample:
            if skip_overflowed:
                # a full intersection defers its spawns (the build's one documented deviation; the reference has no capacity):
                # such an env has left the oracle's trajectory and is reported, not compared
                if len(compared) >= n_sample:
                    break
                b0, le0 = locate(e)
                if b0.read_env(le0).overflow > 0:
                    skipped.append(e)
                    continue
            compared.append(e)
            if lane_num == 12:
                o = OracleEnv(arr[e])
            else:
                from oracle.oracle_geo import OracleGeoEnv
                o = OracleGeoEnv(arr[e], lane_num, choice=None if choice is None else choice[e])
            if table_np is not None:                           # actions by (tick, vehicle id): the same table the device gathers from
                for tk in range(total_ticks - 1):
                    vi = o.vehicles()[0]
                    o.tick(np.where(vi[:, 5] != 0, table_np[tk % table_np.shape[0], np.minimum(vi[:, 2], table_np.shape[1] - 1)], 0.0))
                vi = o.vehicles()[0]
                n = vi.shape[0]
                acts = np.where(vi[:, 5] != 0, table_np[(total_ticks - 1) % table_np.shape[0], np.minimum(vi[:, 2], table_np.shape[1] - 1)], 0.0)
            else:
                if total_ticks > 1:
                    o.run_pool(total_ticks - 1, pool_np[:, e, :], 0)
                vi = o.vehicles()[0]
                n = vi.shape[0]
                acts = np.where(vi[:, 5] != 0, pool_np[(total_ticks - 1) % N_POOL, e, :n], 0.0)
            rec = o.tick(acts)
            slot_of = {(int(l), int(j)): k for k, (l, j) in enumerate(vi[:, :2])}
            batch, le = locate(e)
            out = last_outputs(e)
            eo, flags = out["env_out"], out["flags"].astype(np.int64)
            assert int(eo[0]) == n, "env %d: %d vehicles before the last tick, oracle %d" % (e, int(eo[0]), n)
            want = [slot_of[(int(l), int(j))] for l, j in rec["ids"]]
            got = np.nonzero(flags[:n] & 2)[0].tolist()
            assert sorted(want) == got, "env %d: controlled set of the last tick differs" % e
            assert close(rec["reward"], out["reward"][want], VERIFY_TOL), "env %d: rewards of the last tick differ" % e
            assert int(eo[2]) == rec["collisions"] and int(eo[3]) == rec["lock"], "env %d: collision / lock counters differ" % e
            vs = batch.read_vehicles(le)
            ovi, ovf = o.vehicles()[:2]
            assert len(vs) == ovi.shape[0], "env %d: %d vehicles at the end, oracle %d" % (e, len(vs), ovi.shape[0])
            gi = np.array([[v.lane, v.j, v.id, v.seq_in_lane, v.vnum, v.control, v.finish, v.done, v.collision, v.step,
                            v.count, v.lock, v.lock_a] for v in vs], np.int64).reshape(len(vs), 13)
            gf = np.array([[v.p, v.v, v.a, v.jerk, v.jerk_sum] for v in vs], np.float64).reshape(len(vs), 5)
            assert np.array_equal(gi, ovi[:, :13].astype(np.int64)), "env %d: integer state differs" % e
            assert close(ovf[:, :5], gf, VERIFY_TOL), "env %d: float state differs" % e
    except AssertionError as ex:
        res.update(verified=False, mismatch=str(ex))
    if skip_overflowed:
        res.update(envs=compared, envs_compared=len(compared), envs_skipped_overflowed=skipped)
        if len(compared) < n_sample and res["verified"]:
            # (a parity statement must rest on the full sample: fewer comparable envs than asked for is "not verified", not "verified")
            res.update(verified=None, reason="only %d of the %d envs looked at never deferred a spawn (%d wanted)" % (len(compared), len(sample), n_sample))
    return res


ACTION_TOL = 5e-4       # |a_device - a_numpy| on actions in [-3, 3]: two float32 evaluation orders (tests/actor_scenarios.py)
STATE_FIELDS = ("p", "v", "a", "jerk", "jerk_sum", "vir_dis", "closer_p", "id", "seq", "vnum", "step", "count", "meta", "hdr")


def verify_closed_loop(torch, dev, locate, arr, total_ticks, cap, obs_dtype, weights, n_sample=2 * VERIFY_ENVS, lane_num=12, choice=None, seed=0):
    """Closed loop (BASELINE config 5), outside the timed region: `n_sample` of the envs this rank just timed are replayed
    from reset on the GPU as ONE small batch in the two-launch form (step_with_actor: actor kernel + tick kernel per tick,
    whose tick is the kernel the oracle certifies) for the same number of ticks.  The timed path (the actor inside the
    resident kernel) and the replay run the same float32 actor routine on the same rows, so the persistent state of every
    live slot and the observation rows of the controlled vehicles must be BIT-equal; then the actions the device computes
    on those final rows are held against the NumPy restatement of the actor (oracle/actor_np.py, the checker) at the
    action-level bar.  A trajectory-level oracle does not exist for the closed loop: the NumPy actor's float32 round-off
    (1e-7) is amplified ~270x over 400 ticks.  -> dict(verified=bool, ...)."""
    import pve_mcc_amd
    from oracle.actor_np import actor_forward
    n_envs = arr.shape[0]
    sample = sample_envs(n_envs, n_sample, seed)
    res = dict(verified=True, envs=sample, seed=int(seed), ticks_replayed=int(total_ticks), action_tol=ACTION_TOL,
               checker="the same streams as a %d-env batch in the two-launch form (k_actor_h + k_tick per tick) on the GPU, bit for "
                       "bit; actions on the final rows vs oracle/actor_np.py" % len(sample),
               compared="every persistent field of every live slot + observation rows of the controlled vehicles (exact); "
                        "actor actions on those rows (|da| <= %g)" % ACTION_TOL)
    if total_ticks < 1:
        res.update(verified=None, reason="no tick executed")
        return res
    geo = dict(lane_num=lane_num, intentions=None if choice is None else choice[sample]) if lane_num != 12 else {}
    small = pve_mcc_amd.BatchedIntersections(len(sample), cap, arr[sample], device=dev, obs_dtype=obs_dtype,
                                             outputs=("obs_post", "reward", "flags", "env_out"), **geo)
    small.reset()
    small.set_actor(weights)
    for _ in range(int(total_ticks)):
        small.step_with_actor()
    torch.cuda.synchronize(dev)
    try:
        meta_s = small.state_field("meta")
        for i, e in enumerate(sample):
            b, le = locate(e)
            live = meta_s[i] != 0
            assert torch.equal(b.state_field("meta")[le] != 0, live), "env %d: alive slots differ from the two-launch replay" % e
            for f in STATE_FIELDS:
                assert torch.equal(b.state_field(f)[le][live], small.state_field(f)[i][live]), \
                    "env %d: field %s differs from the two-launch replay" % (e, f)
            ctl = live & ((meta_s[i] & 1) != 0)
            assert torch.equal(b.obs[le][ctl], small.obs[i][ctl]), "env %d: observation rows differ from the two-launch replay" % e
        a_dev = small.act().cpu().numpy()
        ctl = (meta_s.cpu().numpy() & 1) != 0
        a_np = actor_forward(weights, small.obs.cpu().numpy()).astype(np.float64)
        worst = float(np.abs(a_dev - a_np)[ctl].max()) if ctl.any() else 0.0
        res.update(controlled_rows=int(ctl.sum()), max_action_diff=worst)
        assert ctl.any(), "no controlled vehicle in the sample"
        assert worst <= ACTION_TOL, "actor actions differ from the NumPy restatement by %.3e" % worst
    except AssertionError as ex:
        res.update(verified=False, mismatch=str(ex))
    small.close()
    return res


def actor_weights():
    z = np.load(os.path.join(ROOT, "tests", "golden", "actor_66.npz"))
    return {k: z[k] for k in z.files}       # the reference's pretrained actor (model_data/baseline/66.cptk)


def run_companion(torch, dev, kind, K, W, rank, n_envs, verify=True, vseed=0):
    """A second BASELINE configuration timed OUTSIDE the headline region, on envs of its own, with the headline's protocol
    in small (300 un-timed prefill ticks, W warm-up ticks, exactly K timed ticks between synchronisations, the launch
    shape the headline would use at this K) -- so that the driver's single `bench.py` line carries driver-timed numbers
    for them as well:
      closed_loop: BASELINE config 5 -- 4096 x 128, the MADDPG actor inside the resident kernel (pve_step_many(PVE_SRC_ACTOR)),
                   float32 rows, 1000 veh/h/lane; verified by verify_closed_loop;
      cap64:       BASELINE config 2 -- 4096 x 64, the slot-indexed sin pool, 350 veh/h/lane (500 overflows 64 slots);
                   verified by the oracle replay (verify_against_oracle)."""
    import pve_mcc_amd
    from pve_mcc_amd.arrivals import synthetic_arrivals
    closed = kind == "closed_loop"
    on_spec = kind == "cap64_on_spec"
    full_rate = kind == "config2_rate_128slots"           # config 2's STATED rate with room for every vehicle: no deviation
    cap, rate = (128, 1000.0) if closed else ((128, 500.0) if full_rate else (64, 500.0 if on_spec else 350.0))
    prefill = PREFILL_MIN
    seed = 20250213 + (104729 if closed else (15485863 if (on_spec or full_rate) else 1299709)) + rank * n_envs   # (full_rate: cap64_on_spec's streams)
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=(prefill + W + K) * 0.1 + 20.0, seed=seed, lane_num=12)
    obs_dtype = torch.float32 if closed else torch.float64
    outputs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out")
    n_sub, chunk, pers = launch_shape(cap, K, 12, closed, table=not closed, n_envs=n_envs)
    if n_sub == 1:
        env = pve_mcc_amd.BatchedIntersections(n_envs, cap, arr, device=dev, outputs=outputs, obs_dtype=obs_dtype)
    else:
        env = pve_mcc_amd.PipelinedIntersections(n_envs, cap, arr, n_sub=n_sub, device=dev, outputs=outputs, obs_dtype=obs_dtype)
    env.reset()
    table_np, w = None, None
    if closed:
        w = actor_weights()
        env.set_actor(w)

        def run(n):
            if n > 0:
                env.step_many(n, actor=True, chunk=chunk, persistent=pers)
    else:
        table_np = id_sin_table(prefill + W + K + 8)
        env.set_action_table(torch.as_tensor(table_np))
        calls = {n: env.prepare_step_many(n, source="table", chunk=chunk, persistent=pers) for n in {prefill, W, K} if n > 0}

        def run(n):
            if n > 0:
                calls[n]()
    run(prefill)
    run(W)
    torch.cuda.synchronize(dev)
    m0 = env.metrics()
    t0 = time.perf_counter()
    run(K)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    m1 = env.metrics()
    total = prefill + W + K

    def locate(e):
        if n_sub == 1:
            return env, e
        k, le = env.sub_of(e)
        return env.subs[k], le
    if not verify:
        ver = dict(verified=None, reason="skipped (--no-verify)")
    elif closed:
        ver = verify_closed_loop(torch, dev, locate, arr, total, cap, obs_dtype, w, seed=vseed)
    else:
        ver = verify_against_oracle(locate, lambda e: {n: locate(e)[0].out[n][locate(e)[1]].cpu().numpy() for n in ("flags", "reward", "env_out")},
                                    arr, None, total, 12, None, table_np=table_np, n_sample=(2 * VERIFY_ENVS if on_spec else VERIFY_ENVS),
                                    skip_overflowed=on_spec, seed=vseed)
    b_nom = B_ALG_OBS_F32 if closed else B_ALG_FP64
    # ticks per launch / per queue item: the state crosses HBM once per that many ticks
    tpl = (K / float(persistent_items(K, chunk))) if pers else (chunk if chunk > 0 else K)
    b_alg = b_nom - (B_STATE_IN + B_STATE_OUT) * (1.0 - 1.0 / tpl)
    res = {"what": ("BASELINE config 5: %d x %d, 1000 veh/h/lane, MADDPG actor (pretrained 66.cptk weights) inside the resident "
                    "kernel, float32 rows, pve_step_many(PVE_SRC_ACTOR)" if closed else
                    ("BASELINE config 2 ON its stated rate: %d x %d, BASELINE.md 3's tape by vehicle id, 500 veh/h/lane; some of the 4096 "
                     "intersections fill all 64 slots and defer spawns (`overflow`, the build's documented deviation from the "
                     "reference, which has no capacity): throughput is reported as measured, parity on the sampled envs that never "
                     "overflowed" if on_spec else
                     "BASELINE config 2's workload WITHOUT a deviation: %d intersections at its stated 500 veh/h/lane (cap64_on_spec's very "
                     "streams) in %d-slot intersections -- the oracle, which like the reference has no capacity, peaks at 74 vehicles over "
                     "4096 envs x 2300 ticks of these streams (tools/peak_population.py: 189 envs need more than 64 slots, none more than "
                     "96), so no spawn is ever deferred (overflow 0) and every sampled env is held to the oracle" if full_rate else
                     "BASELINE config 2: %d x %d, BASELINE.md 3's tape a = float32(sin(0.37 id + 0.05 tick)) by vehicle id, 350 veh/h/lane "
                     "(BASELINE.md's 500 overflows 64 slots in 4096 envs: the `cap64_on_spec` leg)")) % (n_envs, cap),
           "ms_per_step": dt / K * 1e3, "value": float(cap) * n_envs * K / dt, "unit": "env-steps/s", "steps": K, "warmup": W,
           "prefill_ticks": prefill, "ticks_per_launch": K if (pers or chunk == 0) else chunk, "ticks_per_item": tpl,
           "sub_batches": n_sub, "launch": "persistent work queue" if pers else "one launch per chunk and sub-batch",
           "verified": ver["verified"], "envs_compared": ver.get("envs_compared", len(ver.get("envs", []))),
           "envs_skipped_overflowed": len(ver.get("envs_skipped_overflowed", [])), "verification": ver,
           "overflow": m1["overflow"], "mean_alive_per_env": (m1["alive_steps"] - m0["alive_steps"]) / float(K * n_envs),
           "mean_ctl_per_env": (m1["ctl_steps"] - m0["ctl_steps"]) / float(K * n_envs),
           "hbm_frac": b_alg * cap * n_envs / (dt / K) / 1e9 / HBM_PEAK_GBS}
    del env
    torch.cuda.empty_cache()
    return res


def run_config4(torch, dist, dev, K, W, rank, world, n_envs, env_factory=None, verify=True, vseed=0):
    """BASELINE config 4 beside the weak-scaled headline of a multi-rank run: 64-slot intersections sharded env-parallel,
    n_envs per rank -- with the default 4096 per rank and 8 ranks that is exactly 32 768 x 64 over 8 GPUs, rank k owning the
    global envs shard_range(32768, k, 8) with arrival seeds 20250213 + 32452843 + global env index.  Same protocol as the
    headline: un-timed prefill, W warm-up ticks, exactly K timed ticks between barrier + synchronize pairs, MAX over ranks,
    ONE all-gather of the metrics vectors (the rank's first env and wall-clock ride along); BASELINE.md 3's id-indexed tape
    at 350 veh/h/lane (500 overflows 64 slots, `cap64_on_spec`); every rank's sampled envs replayed by the oracle.
    env_factory: the CPU tests inject the emulator (gloo); the product run never passes it."""
    import pve_mcc_amd
    from pve_mcc_amd.arrivals import synthetic_arrivals
    from pve_mcc_amd.distributed import gather_metrics, shard_range
    emu = env_factory is not None
    cap, rate = 64, 350.0
    lo, hi = shard_range(n_envs * world, rank, world)
    assert hi - lo == n_envs and lo == rank * n_envs
    prefill = PREFILL_MIN if not emu else 0
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=(prefill + W + K) * 0.1 + 20.0, seed=20250213 + 32452843 + lo, lane_num=12)
    outputs = ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out")
    n_sub, chunk, pers = launch_shape(cap, K, 12, False, table=True, n_envs=n_envs)

    def sync():
        if not emu:
            torch.cuda.synchronize(dev)
    # set-up, prefill and warm-up are local; a rank that cannot set the leg up must not leave the others waiting in a
    # collective (nor cost the run its headline): every rank learns whether ALL ranks are ready before the timed region
    env, table_np, calls, problem = None, None, None, None
    try:
        if emu:
            env, n_sub = env_factory(n_envs, cap, arr, outputs), 1
        elif n_sub == 1:
            env = pve_mcc_amd.BatchedIntersections(n_envs, cap, arr, device=dev, outputs=outputs)
        else:
            env = pve_mcc_amd.PipelinedIntersections(n_envs, cap, arr, n_sub=n_sub, device=dev, outputs=outputs)
        env.reset()
        table_np = id_sin_table(prefill + W + K + 8)
        env.set_action_table(torch.as_tensor(table_np))
        calls = {n: env.prepare_step_many(n, source="table", chunk=chunk, persistent=pers) for n in {prefill, W, K} if n > 0}
        for n in (prefill, W):
            if n > 0:
                calls[n]()
        sync()
    except Exception as ex:                               # noqa: BLE001 (reported in the line, the headline stands)
        problem = "%s: %s" % (type(ex).__name__, ex)
    ready = torch.tensor([0.0 if problem else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(ready, op=dist.ReduceOp.MIN)
    if float(ready.item()) < 1.0:
        return {"what": "BASELINE config 4 (64-slot intersections sharded over %d ranks)" % world, "skipped": True,
                "reason": problem or "another rank could not set the leg up", "verified": None}
    dist.barrier()
    sync()
    m0 = env.metrics()
    t0 = time.perf_counter()
    calls[K]()
    sync()
    dt = time.perf_counter() - t0                         # (this rank's K ticks; the closing barrier follows, MAX over ranks below)
    dist.barrier()
    sync()
    tw = torch.tensor([dt], dtype=torch.float64, device=dev)
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    dt_max = float(tw.item())
    m1 = env.metrics()
    per_rank, tot = gather_metrics({k: m1[k] - m0[k] for k in m1}, dev, extra=(dt * 1e3, lo))
    NM = len(pve_mcc_amd._capi.METRIC_NAMES)

    def locate(e):
        if n_sub == 1:
            return env, e
        k, le = env.sub_of(e)
        return env.subs[k], le
    if verify:
        sync()
        try:
            ver = verify_against_oracle(locate, lambda e: {n: locate(e)[0].out[n][locate(e)[1]].cpu().numpy() for n in ("flags", "reward", "env_out")},
                                        arr, None, prefill + W + K, 12, None, table_np=table_np, n_sample=min(4, n_envs), seed=vseed + rank)
        except Exception as ex:                           # noqa: BLE001 (every rank must reach the collective below)
            ver = dict(verified=False, mismatch="%s: %s" % (type(ex).__name__, ex))
    else:
        ver = dict(verified=None, reason="skipped (--no-verify)")
    ok = torch.tensor([0.0 if ver["verified"] is False else 1.0], dtype=torch.float64, device=dev)
    dist.all_reduce(ok, op=dist.ReduceOp.MIN)
    res = {"what": "BASELINE config 4: %d intersections x %d slots sharded env-parallel over %d ranks (%d per rank; global env index = "
                   "shard_range(%d, rank, %d)), BASELINE.md 3's tape by vehicle id, 350 veh/h/lane; one all-gather of the metrics vectors"
                   % (n_envs * world, cap, world, n_envs, n_envs * world, world),
           "ms_per_step": dt_max / K * 1e3, "value": float(cap) * n_envs * world * K / dt_max, "unit": "env-steps/s", "steps": K, "warmup": W,
           "n_gpus": world, "scaling": "weak", "prefill_ticks": prefill,
           "ranks": {"seen": int(per_rank.shape[0]), "ms": [float(x) for x in per_rank[:, NM]],
                     "first_env": [int(x) for x in per_rank[:, NM + 1]], "envs_per_rank": n_envs,
                     "arrival_seeds": "20250213 + 32452843 + first_env + e"},
           "verified": bool(float(ok.item()) >= 1.0) if verify else None, "verification_rank0": ver,
           "overflow": tot["overflow"], "mean_alive_per_env": tot["alive_steps"] / float(K * n_envs * world)}
    del env
    if not emu:
        torch.cuda.empty_cache()
    return res


def item_schedule(K, T, lib=None):
    """The items per intersection of a persistent call of K ticks with items of at most T ticks, as pve_step_many lays them out
    (pve_debug_item_schedule: the library's own schedule function -- host only, no device needed) -> list of item lengths."""
    import ctypes as C
    if lib is None:
        from pve_mcc_amd import _capi
        lib = _capi.load_library()
    out = (C.c_int32 * 12)()
    if lib.pve_debug_item_schedule(int(K), int(T), out) != 0:
        raise ValueError("no persistent schedule for K = %d, T = %d" % (K, T))
    return [out[0]] * out[1] + [out[3 + k] for k in range(out[2])]


def persistent_items(K, T, lib=None):
    return len(item_schedule(K, T, lib))


def launch_shape(cap, K, lane_num, actor, table=False, trajectory=False, n_envs=4096):
    """Default launch shape of a roll-out of K ticks: (sub-batches, ticks per launch or queue item, persistent).
    Measured on MI355X with 4096 envs (tools/ab_launch_shapes.py, same process, medians):
      12 lanes x 128 slots, pool / zero actions: ONE batch and ONE persistent launch per call whose workgroups pull
        (intersection, <= 10 ticks) items from a queue -- 621 us for 20 ticks (items of 9, 8, 3; 6, 6, 5, 3: 627) against 669 us for two
        stream-pipelined sub-batches in launches of 5; 26.4 against 27.5 us per tick in a 1000-tick region (launches of 25);
      12 lanes x 64 slots: every intersection is resident at once (16 one-wave workgroups per CU): ONE launch of all of them
        for the whole call (323 against 341 us for 20 ticks with two sub-batches; 13.7 us per tick in a 1000-tick region, the
        queue with T = 10: 13.6);
      closed loop: the queue (34.6 vs 35.9 us per tick in a long call, 41.7 vs 43.6 for 20 ticks); 4 / 8 lanes x 128 slots: the queue
        for long calls only (37.0 vs 38.2, 36.5 vs 41.7 us per tick; no gain for 20 ticks); small batches: two stream-pipelined sub-batches (a batch below twice the chip's resident workgroups
        gives the queue nothing to balance)."""
    if lane_num == 12 and not actor and n_envs >= 4096:     # (the queue balances a batch of >= 2x the workgroups the chip holds)
        if cap == 128:
            return 1, (12 if K < 100 else 10), True      # (a 20-tick call becomes items of 9, 7 and 4 ticks)
        return 1, 0, False
    if lane_num == 12 and actor and cap == 128 and n_envs >= 4096:
        return 1, (25 if K >= 100 else 12), True     # closed loop: 34.6 against 35.9 us per tick (long call), 41.7 against 43.6 (20 ticks)
    if lane_num != 12 and cap == 128 and not table and n_envs >= 4096 and K >= 100:
        # 8 lanes: 37.0 against 38.2 us per tick, 4 lanes x 128: 36.5 against 41.7; 20 ticks: no gain (2 streams).  Closed loop (round 5,
        # the actor inside k_rollout_geo): 4 lanes x 128 38.4 against 41.8, 8 lanes 41.3 against 41.5; 20 ticks: 49.0 (2 streams) against 54-57
        return 1, 10, True
    return 2, ((25 if K >= 100 else 5) if cap == 128 else 0), False


def self_launch(args_list, n):
    """--gpus N without an outer torchrun: start the N ranks as a child process group (this process has not touched the
    GPU), forward stdout (rank 0's JSON line) and return the children's exit code."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(sys.argv[0])] + list(args_list)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    for l in (lines[-1:] if lines else p.stdout.splitlines()):
        print(l, flush=True)
    return p.returncode


def measured_copy_peak(torch, dev):
    """Device-to-device copy of 1 GiB (read + write counted): the bandwidth a streaming kernel reaches on this chip."""
    n = 1 << 30
    a = torch.empty(n, dtype=torch.uint8, device=dev)
    b = torch.empty(n, dtype=torch.uint8, device=dev)
    a.zero_()
    best = None
    for _ in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        b.copy_(a)
        e1.record()
        torch.cuda.synchronize(dev)
        ms = e0.elapsed_time(e1)
        best = ms if best is None else min(best, ms)
    del a, b
    torch.cuda.empty_cache()
    return 2.0 * n / (best * 1e-3) / 1e9


def main(argv=None, env_factory=None):
    """env_factory: tests inject a factory (device, backend, builder) to exercise the rank plumbing and the
    JSON contract without a GPU; the product run never passes it."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=300)
    ap.add_argument("--envs", type=int, default=4096, help="environments per GPU")
    ap.add_argument("--capacity", type=int, default=128)
    ap.add_argument("--rate", type=float, default=None, help="veh/h/lane (default 1100 at cap 128, 350 at cap 64)")
    ap.add_argument("--prefill", type=int, default=PREFILL_MIN,
                    help="un-timed ticks that fill the intersections before warm-up (continued in chunks of 50 until the "
                         "mean population moves < 1 %%); < 300 marks the line population=cold")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-copy-peak", action="store_true")
    ap.add_argument("--outputs", default="obs_post,reward,flags,nbr,new_slot,env_out")
    ap.add_argument("--lane-num", type=int, default=12, choices=(12, 8, 4),
                    help="intersection layout; 12 = BASELINE metric (k_tick), 4 / 8 = SURVEY 8 f4 (k_tick_geo)")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="free-running sub-batches per GPU, each on its own HIP stream (PipelinedIntersections); 1 = one "
                         "launch over all envs per step")
    ap.add_argument("--mode", default=None, choices=("step", "rollout"),
                    help="step: one pve_step_all launch per tick and sub-batch, enqueued from Python (the reference's "
                         "caller protocol, main.py:397-441); rollout: pve_step_many -- the K ticks of the timed region are "
                         "one call per sub-batch with the action source (pool / actor) on the device")
    ap.add_argument("--persistent", type=int, default=None, choices=(0, 1),
                    help="rollout mode, lane_num 12, action pool: 1 = ONE batch of all envs and ONE persistent launch per call; its "
                         "workgroups pull (intersection, --chunk ticks) items from a queue (pve_rollout.persistent); 0 = "
                         "stream-pipelined sub-batches, one launch per chunk")
    ap.add_argument("--chunk", type=int, default=None,
                    help="rollout mode: ticks per kernel launch / per item of the persistent launch (0 = the whole call in one "
                         "launch; default: launch_shape())")
    ap.add_argument("--trajectory", type=int, default=None, choices=(0, 1),
                    help="rollout mode: 1 = every tick's outputs are RETAINED (trajectory roll-outs into a ring of two "
                         "chunk buffers per sub-batch, what a trainer consumes, main.py:397-441); 0 = each tick overwrites "
                         "the previous tick's outputs")
    ap.add_argument("--tape", default="id-sin", choices=("id-sin", "pool"),
                    help="id-sin (default, the stated workload): BASELINE.md 3's tape a = float32(sin(0.37 id + 0.05 tick)) by VEHICLE "
                         "id, gathered on the device from a [tick][id] table (pve_step_many(PVE_SRC_TABLE); lane_num 12, rollout mode -- "
                         "elsewhere the pool is used and named); pool: 16 slot-indexed sin entries (any layout / mode; what rounds 1-4 "
                         "measured; the default line times K ticks of it behind the headline region: `tape_slot_pool`)")
    ap.add_argument("--no-companion", action="store_true",
                    help="skip the second timed region (per-tick outputs retained) behind the headline roll-out")
    ap.add_argument("--verify-seed", type=int, default=None,
                    help="seed of the envs the self-check replays (default: drawn afresh every run and printed as verification.seed)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the oracle replay of %d sampled envs after the timed region" % VERIFY_ENVS)
    ap.add_argument("--obs-f64", action="store_true",
                    help="with --actor: keep float64 observation rows (default there: float32, the type the actor consumes, "
                         "model_agent_maddpg.py:15; identical trajectories, tests/actor_scenarios.py)")
    ap.add_argument("--obs-f32", action="store_true",
                    help="float32 observation rows (PVE_CFG_OBS_F32; SURVEY 8d's FP32-output variant, 268 B algorithmic); "
                         "the headline / BASELINE metric is the float64 parity layout (380 B)")
    ap.add_argument("--actor", action="store_true",
                    help="BASELINE config 5: close the loop on the device instead of the action pool -- rollout mode: the actor "
                         "runs inside the resident kernel (pve_step_many(PVE_SRC_ACTOR)); step mode: actor launch + tick launch")
    args = ap.parse_args(argv)
    if args.actor and not args.obs_f64:
        args.obs_f32 = True

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # nothing has touched the GPU yet (torch is imported below): children, never exec
        sys.exit(self_launch(sys.argv[1:] if argv is None else argv, args.gpus))
    if args.gpus != world:
        sys.exit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    import torch
    import torch.distributed as dist
    emu = env_factory is not None
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if emu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
    dev = torch.device("cpu") if emu else torch.device("cuda", local_rank)
    if not emu:
        torch.cuda.set_device(dev)

    def sync():
        if not emu:
            torch.cuda.synchronize(dev)

    import pve_mcc_amd
    from pve_mcc_amd.arrivals import synthetic_arrivals, synthetic_intentions
    from pve_mcc_amd.distributed import gather_metrics

    cap, n_envs, lane_num = args.capacity, args.envs, args.lane_num
    vseed = args.verify_seed if args.verify_seed is not None else int.from_bytes(os.urandom(4), "little")
    # capacity 64: 350 veh/h/lane keeps the peak population of 4096 envs x 2300 ticks at 57 of 64 slots (oracle run): no deferred
    # spawn ever enters a timed run (400 peaks at 62, 450 at 64+, 500 overflowed 4539 times in round 1)
    # closed loop: the pretrained actor keeps ~7 % more vehicles in the box than the sin tape; 1000 veh/h/lane keeps every
    # one of 4096 envs under 128 slots (1100 deferred 59 spawns in 300 ticks)
    rate = args.rate or {12: ((1000.0 if args.actor else 1100.0) if cap == 128 else 350.0), 8: 1500.0,
                         4: (1800.0 if cap == 128 else 1200.0)}[lane_num]
    K, W = args.steps, args.warmup
    prefill_min = max(0, args.prefill)
    prefill_cap = max(prefill_min, PREFILL_MAX) if prefill_min >= PREFILL_MIN else prefill_min
    horizon = max((K + W + prefill_cap) * 0.1 + 20.0, (CPU_WARM + 2 * CPU_TICKS) * 0.1 + 10.0)   # (the CPU sample replays the same streams)
    # weak scaling: every rank owns its own n_envs environments (global env index = rank*n_envs + e)
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=horizon, seed=20250213 + rank * n_envs, lane_num=lane_num)
    choice = synthetic_intentions(n_envs, arr.shape[1], seed=20250213 + rank * n_envs) if lane_num == 8 else None
    pool_np = action_pool(n_envs, cap, seed=99 + rank)
    # BASELINE.md 3's tape, by vehicle id: one row per tick of the whole run (incl. the companion region), one column per id a
    # lane can hand out in that time (headways >= 1 s)
    id_sin = (args.tape == "id-sin") and lane_num == 12 and not args.actor and not emu
    table_np = None
    if id_sin:
        table_np = id_sin_table(prefill_cap + W + 4 * K + 256)     # (headline + the retained-outputs companion behind it)
    outputs = tuple(x for x in args.outputs.split(",") if x)
    # launch shape: launch_shape()'s measured default unless --pipeline / --chunk / --persistent say otherwise
    rollout_like = (args.mode or ("rollout" if not emu else "step")) == "rollout"
    d_sub, d_chunk, d_pers = launch_shape(cap, K, lane_num, args.actor, table=id_sin, n_envs=n_envs)
    if not rollout_like:
        d_sub, d_chunk, d_pers = 2, 0, False
    can_pers = rollout_like and not emu
    pers = can_pers and args.chunk != 0 and \
        (bool(args.persistent) if args.persistent is not None else (d_pers and args.pipeline in (None, 1)))
    if args.pipeline is None:
        args.pipeline = 1 if pers else (d_sub if args.persistent is None else 2)
    if args.chunk is None:
        args.chunk = d_chunk if (pers == d_pers and args.pipeline == d_sub) else ((25 if K >= 100 else 5) if (cap == 128 or pers) else 0)
    n_sub = 1 if pers else max(1, min(args.pipeline, n_envs))
    obs_dtype = torch.float32 if args.obs_f32 else torch.float64
    if emu:
        env = env_factory(n_envs, cap, arr, outputs)
        n_sub = getattr(env, "n_sub", 1)
    elif n_sub == 1:
        env = pve_mcc_amd.BatchedIntersections(n_envs, cap, arr, device=dev, outputs=outputs, lane_num=lane_num,
                                               intentions=choice, obs_dtype=obs_dtype)
    else:
        # the envs are independent: n_sub free-running sub-batches on their own streams pipeline the ticks (the chip-wide
        # LOAD / FIN bursts of one sub-batch overlap the compute phases of the other), DESIGN.md 5
        env = pve_mcc_amd.PipelinedIntersections(n_envs, cap, arr, n_sub=n_sub, device=dev, outputs=outputs,
                                                 lane_num=lane_num, intentions=choice, obs_dtype=obs_dtype)
    # auto: pve_step_many (state resident on the chip, host out of the loop) is the faster product path; launch_shape() above
    mode = args.mode or ("rollout" if not emu else "step")
    if id_sin and mode != "rollout":
        id_sin, table_np = False, None                   # (one launch per tick: the slot-indexed pool)
    src = "table" if id_sin else None
    if mode == "rollout" and not hasattr(env, "step_many"):
        sys.exit("--mode rollout: this build has no pve_step_many")
    traj_on = bool(args.trajectory) if args.trajectory is not None else False
    if traj_on and (mode != "rollout" or args.actor):
        sys.exit("--trajectory 1 needs --mode rollout with the action pool")
    pool = torch.as_tensor(pool_np, device=dev)
    sync()                                            # the pool upload precedes every sub-batch stream's first launch
    env.reset()
    sub_streams = getattr(env, "streams", None) if not emu else None
    if args.actor:
        env.set_actor(actor_weights())
    if mode == "rollout" and not args.actor:
        if id_sin:
            env.set_action_table(torch.as_tensor(table_np))
        else:
            env.set_action_pool(pool)
    tick = [0]
    step_kw = {"wait": False} if sub_streams else {}     # the pool upload was synchronised above
    # retained per-tick outputs: a ring of two chunk buffers per sub-batch (the consumer reads one while the next fills)
    traj_len = ((25 if pers else args.chunk) if args.chunk > 0 else 25) if traj_on else 0      # ticks per call = per ring buffer
    ring = [env.alloc_trajectory(traj_len) for _ in range(2)] if traj_on else None
    ring_pos = [0, None, 0]                              # next buffer, (buffer, ticks) of the last call

    prepared = {}
    pers_kw = {"persistent": True} if pers else {}

    launch_seq = {}
    if os.environ.get("PVE_BENCH_CHUNKS") and mode == "rollout" and hasattr(env, "prepare_step_many"):
        seq = [int(x) for x in os.environ["PVE_BENCH_CHUNKS"].split(",")]
        launch_seq[sum(seq)] = seq
        for m in set(seq):
            prepared[("one", m)] = env.prepare_step_many(m, source=src, chunk=0)

    def run_ticks(n):
        """n ticks of every env of this rank, enqueued (not synchronised)."""
        if n <= 0:
            return
        if mode == "rollout" and not traj_on and not args.actor and hasattr(env, "prepare_step_many"):
            # prepared calls: the host side of the timed region is one ctypes call per sub-batch
            seq = launch_seq.get(n)
            if seq:                                       # a sequence of launches of different lengths (see below)
                for m in seq:
                    prepared[("one", m)]()
                tick[0] += n
                return
            if n not in prepared:
                prepared[n] = env.prepare_step_many(n, source=src, chunk=args.chunk, **pers_kw)
            prepared[n]()
            tick[0] += n
            return
        if traj_on:
            for c0 in range(0, n, traj_len):
                m = min(traj_len, n - c0)
                env.step_many(m, source=src, trajectory=ring[ring_pos[0]], update_views=False,
                              **({"chunk": args.chunk, "persistent": True} if pers else {}))
                ring_pos[1], ring_pos[2] = ring_pos[0], m
                ring_pos[0] ^= 1
        elif mode == "rollout":
            env.step_many(n, actor=args.actor, source=src, chunk=args.chunk, **pers_kw)
        elif args.actor:
            for _ in range(n):
                env.step_with_actor()
        else:
            for t in range(tick[0], tick[0] + n):
                env.step(pool[t % N_POOL], **step_kw)
        tick[0] += n

    # ---- un-timed prefill to steady state (vehicle lifetime ~270 ticks): never part of --warmup
    def alive_steps():
        sync()
        return env.metrics()["alive_steps"]

    prefill, drift = 0, None
    a_prev = alive_steps()
    chunk_means = []
    while True:
        if prefill >= prefill_min and (prefill_min < PREFILL_MIN or prefill >= prefill_cap or
                                       (drift is not None and drift < 0.01)):
            break
        n = min(PREFILL_CHUNK, prefill_min - prefill) if prefill < prefill_min else PREFILL_CHUNK
        run_ticks(n)
        prefill += n
        a_now = alive_steps()
        chunk_means.append((a_now - a_prev) / float(n * n_envs))
        a_prev = a_now
        if len(chunk_means) >= 2 and chunk_means[-1] > 0:
            drift = abs(chunk_means[-1] - chunk_means[-2]) / chunk_means[-1]
    steady = prefill >= PREFILL_MIN and drift is not None and drift < 0.01

    run_ticks(W)
    if mode == "rollout" and not traj_on and not args.actor and hasattr(env, "prepare_step_many") and K > 0 and K not in prepared:
        prepared[K] = env.prepare_step_many(K, source=src, chunk=args.chunk, **pers_kw)      # (built outside the timed region)
    sync()
    if world > 1:
        dist.barrier()
    sync()
    m0 = env.metrics()
    # HIP events on the streams the kernels are launched on: torch's current stream for one batch, every sub-batch
    # stream for the pipelined form (start / end of the K ticks of that stream)
    if not emu:
        ev_streams = sub_streams if sub_streams else [torch.cuda.current_stream(dev)]
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in ev_streams]
    # (the start events are instrumentation, not work: recorded on the idle streams before the clock starts -- two
    #  hipEventRecord calls are 13-25 us of host time, 2-3 % of the driver's 20-tick region)
    if not emu:
        for (e0, _), st in zip(evs, ev_streams):
            e0.record(st)
    t0 = time.perf_counter()
    t_rec = t0
    run_ticks(K)
    t_enq = time.perf_counter()
    if not emu:
        for (_, e1), st in zip(evs, ev_streams):
            e1.record(st)
    sync()
    # this rank's K ticks, from the first enqueue to their completion.  The closing bracket (barrier + synchronize) follows; the
    # job's time is the MAX over the ranks' own times -- the env-parallel path has no collective, so the latency of the
    # bracketing barrier itself (an RCCL kernel) is not part of any rank's K ticks
    wall_local = time.perf_counter() - t0
    if os.environ.get("PVE_BENCH_TIMELINE") and not emu:     # host side of the timed region, us (diagnostics, stderr)
        t_end = time.perf_counter()
        sys.stderr.write("timeline us: record %.1f  enqueue %.1f  wait %.1f  total %.1f | per-stream event spans %s\n" % (
            (t_rec - t0) * 1e6, (t_enq - t_rec) * 1e6, (t_end - t_enq) * 1e6, (t_end - t0) * 1e6,
            ["%.1f" % (e0.elapsed_time(e1) * 1e3) for e0, e1 in evs]))
    if world > 1:
        dist.barrier()
    sync()
    wall = wall_local
    # average duration of one tick of one sub-batch on its stream (K back-to-back ticks per stream)
    gpu_ms = (sum(e0.elapsed_time(e1) for e0, e1 in evs) / len(evs)) if not emu else wall * 1e3
    if world > 1:
        tw = torch.tensor([wall_local], dtype=torch.float64, device=dev)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    m1 = env.metrics()
    delta = {k: m1[k] - m0[k] for k in m1}
    # the single RCCL all-gather (metrics only; the rank's own wall-clock and first global env index ride along)
    per_rank, tot = gather_metrics(delta, dev, extra=(wall_local * 1e3, rank * n_envs))
    NM = len(pve_mcc_amd._capi.METRIC_NAMES)
    rank_ms = [float(x) for x in per_rank[:, NM]]
    rank_env0 = [int(x) for x in per_rank[:, NM + 1]]
    rank_ticks = [float(x) / n_envs for x in per_rank[:, pve_mcc_amd._capi.METRIC_NAMES.index("ticks")]]

    # ---- self-check, outside the timed region: the envs that were timed against the CPU oracle
    verify = dict(verified=None, reason="skipped (--no-verify)")
    subs = getattr(env, "subs", None)

    def locate(e):
        if subs is None:
            return env, e
        k, le = env.sub_of(e)
        return subs[k], le
    if args.actor and not args.no_verify and not emu:
        sync()
        verify = verify_closed_loop(torch, dev, locate, arr, tick[0], cap, obs_dtype, actor_weights(), lane_num=lane_num, choice=choice, seed=vseed)
    elif args.actor:
        verify = dict(verified=None, reason="closed loop: skipped (--no-verify) or no GPU replay available")
    elif not args.no_verify:

        def last_outputs(e):
            b, le = locate(e)
            if traj_on:
                k = env.sub_of(e)[0] if subs is not None else 0
                tr = ring[ring_pos[1]]
                tr = tr[k] if subs is not None else tr
                return {n: tr[n][ring_pos[2] - 1, le].cpu().numpy() for n in ("flags", "reward", "env_out")}
            return {n: b.out[n][le].cpu().numpy() for n in ("flags", "reward", "env_out")}
        if all(n in outputs for n in ("flags", "reward", "env_out")):
            sync()
            verify = verify_against_oracle(locate, last_outputs, arr, pool_np, tick[0], lane_num, choice, table_np=table_np, seed=vseed + rank)
        else:
            verify = dict(verified=None, reason="needs the flags, reward and env_out outputs")
    # ---- companion figure: the headline roll-out overwrites each tick's outputs with the next tick's; the same K
    # ticks with every tick's outputs RETAINED (trajectory roll-outs into a ring of two chunk buffers per sub-batch: what a
    # trainer consumes) are timed right behind it, outside the headline region
    companion = None
    if mode == "rollout" and not traj_on and not args.actor and not emu and world == 1 and not args.no_companion:
        # calls of `tl` ticks, each into one buffer of the ring (with the persistent launch: one launch per call, items of
        # args.chunk ticks, every tick writing its own block)
        tl = (min(K, 25) if pers else (args.chunk if args.chunk > 0 else 25))
        ring2 = [env.alloc_trajectory(tl) for _ in range(2)]
        def traj_ticks(n):
            for i, c0 in enumerate(range(0, n, tl)):
                env.step_many(min(tl, n - c0), source=src, trajectory=ring2[i & 1], update_views=False,
                              **({"chunk": args.chunk, "persistent": True} if pers else {}))
        traj_ticks(2 * tl)
        sync()
        tc0 = time.perf_counter()
        traj_ticks(K)
        sync()
        tc = time.perf_counter() - tc0
        tick[0] += 2 * tl + K
        companion = {"what": "the same workload with every tick's outputs retained (pve_step_many trajectory = 1, calls of %d "
                             "ticks into a ring of two buffers per sub-batch)" % tl,
                     "ms_per_step": tc / K * 1e3, "value": float(cap) * n_envs * K / tc, "unit": "env-steps/s", "steps": K}
        # ... and certified like the headline: the sampled envs replayed from reset by the oracle through ALL the ticks so far,
        # the last tick's outputs taken from the trajectory block that holds them
        if not args.no_verify and all(n in outputs for n in ("flags", "reward", "env_out")) and n_sub == 1:
            n_calls = -(-K // tl)
            last_buf, last_k = ring2[(n_calls - 1) & 1], (K - (n_calls - 1) * tl) - 1
            cv = verify_against_oracle(locate, lambda e: {n: last_buf[n][last_k, locate(e)[1]].cpu().numpy() for n in ("flags", "reward", "env_out")},
                                       arr, pool_np, tick[0], lane_num, choice, table_np=table_np, seed=vseed + 1)
            companion.update(verified=cv["verified"], verification=cv)
        del ring2
    # ---- the slot-indexed sin pool (the headline of rounds 1-4) beside BASELINE.md 3's id-indexed tape, or the other way round
    # with --tape pool: K more ticks of the same envs under the other tape, timed outside the headline (and verified) region
    tape_id_sin = tape_slot_pool = None
    if mode == "rollout" and not traj_on and not args.actor and not emu and world == 1 and lane_num == 12 and not args.no_companion:
        if id_sin:
            env.set_action_pool(pool)
            other_src = "pool"
        else:
            env.set_action_table(torch.as_tensor(id_sin_table(tick[0] + 2 * K + 64)))
            other_src = "table"
        call = env.prepare_step_many(K, source=other_src, chunk=args.chunk, **pers_kw)
        warm = env.prepare_step_many(min(K, 50), source=other_src, chunk=args.chunk, **pers_kw)
        warm(); sync()
        ti0 = time.perf_counter()
        call()
        sync()
        ti = time.perf_counter() - ti0
        tick[0] += K + min(K, 50)
        comp = {"what": ("the same envs continued for %d ticks under %s; not part of the verified region" % (K, (
                         "the slot-indexed sin pool of rounds 1-4 (pool[k][env][slot], 16 entries, pve_step_many(PVE_SRC_POOL))" if id_sin else
                         "BASELINE.md 3's tape a = float32(sin(0.37 id + 0.05 tick)) by vehicle id (pve_step_many(PVE_SRC_TABLE))"))),
                "ms_per_step": ti / K * 1e3, "value": float(cap) * n_envs * K / ti, "unit": "env-steps/s", "steps": K}
        if id_sin:
            tape_slot_pool = comp
        else:
            tape_id_sin = comp
    # ---- BASELINE configs 5 and 2 beside the headline (config 3), each on envs of its own, timed outside the headline region
    closed_loop = cap64 = cap64_on_spec = config2_rate_128slots = None
    if mode == "rollout" and not traj_on and not args.actor and not emu and world == 1 and lane_num == 12 \
            and cap == 128 and not args.no_companion and K > 0:
        del env
        torch.cuda.empty_cache()
        closed_loop = run_companion(torch, dev, "closed_loop", K, W, rank, n_envs, verify=not args.no_verify, vseed=vseed + 2)
        cap64 = run_companion(torch, dev, "cap64", K, W, rank, n_envs, verify=not args.no_verify, vseed=vseed + 3)
        cap64_on_spec = run_companion(torch, dev, "cap64_on_spec", K, W, rank, n_envs, verify=not args.no_verify, vseed=vseed + 4)
        config2_rate_128slots = run_companion(torch, dev, "config2_rate_128slots", K, W, rank, n_envs, verify=not args.no_verify, vseed=vseed + 6)
    # ---- multi-rank runs carry BASELINE config 4 (64-slot intersections sharded over the ranks) beside the weak-scaled headline
    config4 = None
    if world > 1 and lane_num == 12 and not args.no_companion and K > 0 and (mode == "rollout" or emu):
        del env
        config4 = run_config4(torch, dist, dev, K, W, rank, world, n_envs, env_factory=env_factory, verify=not args.no_verify, vseed=vseed + 5)
    ok_flag = 0.0 if (verify["verified"] is False or any(c and c.get("verified") is False for c in (closed_loop, cap64, cap64_on_spec, config2_rate_128slots, config4, companion))) else 1.0
    if world > 1:
        tv = torch.tensor([ok_flag], dtype=torch.float64, device=dev)
        dist.all_reduce(tv, op=dist.ReduceOp.MIN)
        if float(tv.item()) < 1.0 and verify["verified"]:
            verify = dict(verify, verified=False, mismatch="another rank's envs differ from the oracle")
        ok_flag = float(tv.item())

    if rank == 0:
        slot_steps = float(cap) * n_envs * K * world
        value = slot_steps / wall
        envs_per_launch = n_envs / float(n_sub)
        tpl = (traj_len if traj_on else (args.chunk if args.chunk > 0 else K)) if mode == "rollout" else 1      # ticks per kernel launch
        if pers:
            tpl = K                                       # one launch for the whole call ...
        # ... whose items move an intersection's state across HBM once each
        t_state = (K / float(persistent_items(K, args.chunk))) if pers else tpl
        # ---- algorithmic bytes.  SURVEY 8d's per-unit figure (380 B per vehicle-slot-step, FP64 layout) assumes the
        # persistent state is read and written every tick.  pve_step_many keeps it on the chip: a launch of T ticks
        # moves the state once, so per slot-step it must move 380 - 124 + 124 / T bytes.  `achieved` / `frac` charge what
        # the measured mode has to move; `nominal` is the 8d figure whatever the mode (an "equivalent" rate).
        b_nom = B_ALG_OBS_F32 if args.obs_f32 else B_ALG_FP64
        b_alg = b_nom if mode != "rollout" else b_nom - (B_STATE_IN + B_STATE_OUT) * (1.0 - 1.0 / t_state)
        # per GPU: algorithmic bytes of one tick of all the rank's envs / wall-clock per tick (NOT per-launch x launches)
        achieved = b_alg * cap * n_envs / (wall / K) / 1e9
        nominal = b_nom * cap * n_envs / (wall / K) / 1e9
        kern_s = gpu_ms * 1e-3 / K                      # one tick of one sub-batch (n_envs / n_sub envs) on its stream
        per_launch = b_alg * cap * envs_per_launch / kern_s / 1e9
        kname = (("k_rollout<%d>" if mode == "rollout" else "k_tick<%d>") if lane_num == 12 else
                 ("k_rollout_geo<%d>" if mode == "rollout" else "k_tick_geo<%d>")) % cap
        if pers:
            kname += " persistent (PERS: work queue)"
            if lane_num == 12 and cap == 128 and not args.actor and not traj_on and not any(n in outputs for n in ("obs_pre", "state_pre")):
                kname = "k_rollout<128, 5, ..> persistent (PERS: work queue; HOME: carried per-slot fields in LDS homes, 10 workgroups per CU)"
        if args.actor:
            kname += " with the actor inside (ACT)" if mode == "rollout" else " + k_actor_h"
        # (another kernel variant than the profiled one: the committed counter passes are those of the default = id-sin command)
        other = args.actor or lane_num != 12 or args.obs_f32 or traj_on or (mode == "rollout" and not id_sin)
        pkey = (("persist" if K >= 100 else "persist_short") if pers else None)
        tr, traffic_src = (None, None) if (emu or not steady) else pmc_traffic(int(envs_per_launch), cap, outputs, mode, tpl, other, pkey)
        traffic = tr["hbm_bytes_per_launch"] if tr else None
        counter_rate = (traffic / tpl * n_sub / (wall / K) / 1e9) if traffic else None
        peak_meas = None if (emu or args.no_copy_peak) else measured_copy_peak(torch, dev)
        mean_alive = tot["alive_steps"] / (K * n_envs * world)
        # ---- what binds: VALU issue (SQ counters of this build), not bandwidth
        std_out = tuple(outputs) == ("obs_post", "reward", "flags", "nbr", "new_slot", "env_out")
        if args.actor and mode == "rollout" and lane_num == 12 and args.obs_f32 and not traj_on:
            bp, bind_src = (None, None) if emu else binding_profile("actor_rollout", tpl, cap, not std_out,
                                                                    "actor_persist" if pers else None)   # the closed loop's own SQ pass
        else:
            bp, bind_src = (None, None) if emu else binding_profile(mode, tpl, cap, other or not std_out, pkey)
        binding = None
        if bp:
            waves_per_tick = n_envs * cap / 64.0
            issue_us = waves_per_tick * bp["valu_per_wave_tick"] * VALU_CYCLES / (N_SIMD * bp["shader_clock_ghz"] * 1e3)
            binding = {"kind": "valu-issue", "frac": issue_us / (wall / K * 1e6), "valu_issue_us_per_tick": issue_us,
                       "valu_per_wave_tick": bp["valu_per_wave_tick"], "salu_per_wave_tick": bp["salu_per_wave_tick"],
                       "lds_per_wave_tick": bp["lds_per_wave_tick"],
                       "lds_bank_conflict_frac": bp["lds_bank_conflict_frac"], "wait_frac": bp["wait_frac"],
                       "lds_bank_conflict_cycles_per_wave_tick": bp.get("lds_bank_conflict_cycles_per_wave_tick"),
                       "shader_clock_ghz": bp["shader_clock_ghz"], "profiled_in": bind_src,
                       "definition": "frac = (waves per tick x VALU instructions per wave and tick x 4 cycles / 1024 SIMDs / "
                                     "shader clock) / measured time per tick: the share of the tick during which the vector "
                                     "pipes HAVE to be busy; lds_bank_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_ACTIVE_INST_LDS; "
                                     "wait_frac = SQ_WAIT_ANY / SQ_WAVE_CYCLES"}
        pool_desc = ("sin action tape by VEHICLE id, a = float32(sin(0.37 id + 0.05 tick)) (BASELINE.md 3), gathered on the device "
                     "from a [tick][id] table (pve_step_many(PVE_SRC_TABLE))") if id_sin else ("sin action pool indexed by SLOT, pool[k][env][slot] = float32(sin(phase_env + 0.37 slot + 0.35 k)), 16 "
                     "entries (BASELINE.md 3 indexes its sin tape by vehicle id: sin(0.37 id + 0.05 tick); a slot-indexed "
                     "tape needs no feedback from the device and costs the same per tick)")
        line = {
            "metric": "env-steps/sec (vehicles x envs x steps/s) at 128 veh x 4096 envs",
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": wall / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic" if not emu else "synthetic (injected test environment: timings meaningless)",
            "verified": verify["verified"], "verification": verify,
            "retained_outputs": companion, "tape_id_sin": tape_id_sin, "tape_slot_pool": tape_slot_pool,
            "closed_loop": closed_loop, "cap64": cap64, "cap64_on_spec": cap64_on_spec, "config2_rate_128slots": config2_rate_128slots,
            "config4": config4,
            "config": {"workload": "%d parallel %d-lane intersections x %d vehicle slots per GPU, synthetic Poisson "
                                   "arrivals %.0f veh/h/lane, %s, fused step+scene_update+delete tick"
                                   % (n_envs, lane_num, cap, rate, "on-device MADDPG actor (pretrained 66.cptk weights) closing the loop"
                                      if args.actor else pool_desc),
                       "envs_per_gpu": n_envs, "capacity": cap, "mode": mode, "ticks_per_launch": tpl, "ticks_per_state_move": t_state,
                       "launch": ("persistent: one launch per call, %d-tick (intersection, chunk) items pulled from a work queue"
                                  % args.chunk) if pers else "one launch per chunk and sub-batch",
                       "per_tick_outputs": ("every tick's outputs written to their own block (trajectory roll-out, ring of 2 "
                                            "chunk buffers per sub-batch)" if traj_on else
                                            ("overwritten by the next tick of the same launch (only the last tick of a call "
                                             "can be read; --trajectory 1 retains them)" if mode == "rollout" else
                                             "readable after every tick (one launch per tick)")),
                       "parallelism": "env-parallel x%d" % world + (", %d stream-pipelined sub-batches of %d envs per GPU"
                                                                   % (n_sub, int(envs_per_launch)) if n_sub > 1 else ""),
                       "outputs": list(outputs), "obs_dtype": "f32" if args.obs_f32 else "f64"},
            "ranks": {"seen": len(rank_ms), "ms": rank_ms, "ticks": rank_ticks, "first_env": rank_env0,
                      "envs_per_rank": n_envs, "arrival_seeds": "20250213 + first_env + e"},
            "population": "steady" if steady else "cold",
            "prefill_ticks": prefill, "prefill_drift": drift,
            "alive_steps_per_s": tot["alive_steps"] / wall,
            "ctl_steps_per_s": tot["ctl_steps"] / wall,
            "mean_alive_per_env": mean_alive,
            "mean_ctl_per_env": tot["ctl_steps"] / (K * n_envs * world),
            "overflow": tot["overflow"],
            "roofline": {"bound": "hbm", "limited_by": "vector-instruction issue and LDS round-trip latency, not bandwidth (`binding`; "
                                                        "`bound` names the roofline SURVEY 8d prices this scan / element-wise path against)",
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "alg_bytes_per_slot_step": b_alg,
                         "nominal": {"alg_bytes_per_slot_step": b_nom, "achieved": nominal, "frac": nominal / HBM_PEAK_GBS,
                                     "note": "SURVEY 8d's figure charged every tick whatever the mode (state re-read and "
                                             "re-written per tick); equals `achieved` in step mode"},
                         "traffic": traffic, "traffic_unit": "bytes/launch", "traffic_profiled_in": traffic_src,
                         "achieved_counter_bytes": counter_rate,
                         "frac_counter_bytes": (counter_rate / HBM_PEAK_GBS) if counter_rate else None,
                         "peak_measured": peak_meas,
                         "frac_of_peak_measured": (achieved / peak_meas) if peak_meas else None,
                         "binding": binding,
                         "kernel": kname, "kernel_ms": kern_s * 1e3,
                         "launch_ms": (gpu_ms if pers else kern_s * 1e3 * tpl),     # duration of one launch of the dominant kernel (HIP events)
                         "envs_per_launch": int(envs_per_launch),
                         "concurrent_launches": n_sub, "per_launch_achieved": per_launch,
                         "definition": "achieved = algorithmic bytes the measured mode must move (step: 380 B per slot-step, "
                                       "SURVEY 8d; pve_step_many launch of T ticks: 380 - 124 (1 - 1/T), the persistent state "
                                       "crosses HBM once per launch -- once per queue ITEM of the persistent launch) x capacity x envs of the GPU, every slot counted / "
                                       "wall-clock per tick; per_launch_achieved = the bytes of one sub-batch / "
                                       "its tick time on its own stream (HIP events), sub-batches overlap; kernel_ms = launch_ms / ticks per "
                                       "launch (a persistent launch = the whole timed call: rocprofv3's LAST dispatch of the kernel); "
                                       "achieved_counter_bytes = HBM bytes the PMC counters saw (traffic, profiled on this "
                                       "very build and config, else null) / wall-clock per tick; peak_measured = 1 GiB "
                                       "device copy, read + write; binding = the roofline that actually limits the kernel.  "
                                       "TWO READINGS of north_star's `>= 40 % HBM roofline`: `frac` answers SURVEY 8d's accounting "
                                       "(algorithmic bytes the mode must move / time / 8 TB/s: what a byte-minimal implementation "
                                       "of the same interface would have to stream); `frac_counter_bytes` answers the literal one "
                                       "(`rocprof achieved-HBM-GB/s against the chip's peak`: bytes the memory system actually "
                                       "moved, ~half the algorithmic ones because empty slots, uncontrolled rows and unchanged "
                                       "fields are never written) -- the kernel is bound by `binding`, not by either"},
        }
        if args.actor:
            line["roofline"]["note"] = "closed loop: actor + tick per step; achieved uses the tick's algorithmic bytes only"
            line["metric"] += " (closed loop: MADDPG actor inference per controlled vehicle inside the step)"
        if peak_meas and achieved > peak_meas:
            line["roofline"]["note_above_copy_peak"] = ("algorithmic bytes exceed what a copy kernel moves in the same time: "
                                                        "empty slots are counted but not moved")
        if not args.no_cpu_baseline and not args.actor and world == 1:      # reported at N=1 only
            line["cpu_baseline"] = cpu_baseline(arr, pool_np, cap, lane_num, choice, id_sin=id_sin)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()                # every rank leaves together (rank 0 may still be timing the CPU baseline)
        dist.destroy_process_group()
    if ok_flag < 1.0:
        bad = [v.get("mismatch") for v in [verify] + [c.get("verification") or c.get("verification_rank0") or {}
                                                      for c in (closed_loop, cap64, cap64_on_spec, config2_rate_128slots, config4, companion) if c] if v.get("verified") is False]
        sys.exit("bench.py: the timed environments do NOT match the checker (%s)" % "; ".join(str(b) for b in bad))


if __name__ == "__main__":
    main()
