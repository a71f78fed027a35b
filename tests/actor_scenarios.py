"""Actor-inference parity scenarios, shared by the CPU (emulator) and GPU test files.
Parity is ACTION-level (float32, tolerance on `a`), never trajectory-level: the closed loop amplifies a
1e-7 perturbation ~270x over 400 ticks."""
import numpy as np
import torch

from oracle.actor_np import actor_forward, flat_weights, load_weights
from oracle.oracle import OracleEnv
from pve_mcc_amd.arrivals import load_arrival_mat
from tests.hip_adapter import _np, make_batch
from tests.parity_util import GOLDEN_DIR
import os

# |a_dev - a_numpy| on a in [-3, 3].  Both sides are float32 with different accumulation orders; the three
# LayerNorms amplify round-off: the NumPy float32 restatement itself is up to 3.1e-4 away from a float64
# evaluation of the same network on these states, so two float32 implementations agree to ~5e-4, not 1e-6.
ACTION_TOL = 5e-4


def stream_1000():
    return load_arrival_mat(os.path.join(GOLDEN_DIR, "streams", "arvTimeNewVeh_new_1000_12.mat"))


def check_actions_on_oracle_states(backend, ticks=300):
    """Drive the ORACLE env with the numpy actor (the pinned closed loop); at every tick feed the very same
    observations to the device actor and compare the actions."""
    w = load_weights()
    arr = stream_1000()
    orc = OracleEnv(arr)
    b = make_batch(arr, 1, 128, backend, outputs=("obs_post", "reward", "flags", "env_out"))
    b.reset()
    b.set_actor(flat_weights(w))
    worst = 0.0
    for t in range(ticks):
        vid, ctl, obs0 = orc.alive_view()
        n = len(vid)
        a_np = np.where(ctl != 0, actor_forward(w, obs0).astype(np.float64), 0.0)
        # mirror the oracle's state into the device buffers: same ids / control flags by construction
        assert np.array_equal(_np(b.state_field("id")[0, :n]), vid), "state diverged at tick %d" % t
        b.obs[0, :n] = torch.as_tensor(obs0).to(b.device)
        a_dev = _np(b.act())[0]
        assert np.all(a_dev[n:] == 0)
        assert np.all(a_dev[:n][ctl == 0] == 0), "uncontrolled slots must get 0 (main.py:401)"
        err = np.abs(a_dev[:n] - a_np).max() if n else 0.0
        worst = max(worst, err)
        assert err <= ACTION_TOL, "tick %d: actor output differs by %.3e" % (t, err)
        # advance both with the ORACLE's actions (tape discipline)
        orc.tick(a_np)
        acts = torch.zeros(1, 128, dtype=torch.float64)
        acts[0, :n] = torch.as_tensor(a_np)
        b.step(acts.to(b.device))
    return worst


def check_closed_loop_on_device(backend, ticks=1000):
    """Closed loop entirely behind the C ABI (actor -> tick, no host round trip): the pretrained policy must
    drive the 1000 stream collision-free with the throughput of SURVEY App. D (281 passed of 323 spawned)."""
    w = load_weights()
    b = make_batch(stream_1000(), 1, 128, backend, outputs=("obs_post", "reward", "flags", "env_out"))
    b.reset()
    b.set_actor(flat_weights(w))
    for t in range(ticks):
        b.step_with_actor()
    m = b.metrics()
    assert m["spawned"] == 323
    assert m["collided"] == 0, "the pretrained actor must not collide"
    assert abs(m["passed"] - 281) <= 3 and m["overflow"] == 0
    pt_m = m["passed_steps"] / (m["passed"] + 1e-4) * 0.1
    assert abs(pt_m - 12.294) < 0.15, pt_m
    return m


def check_closed_loop_f32_obs_equals_f64(backend, ticks=300, n_envs=4):
    """PVE_CFG_OBS_F32: the actor casts its input to float32 anyway (model_agent_maddpg.py:15), so the closed loop on
    float32 observation rows is bit-identical to the closed loop on float64 rows."""
    from pve_mcc_amd.arrivals import synthetic_arrivals
    w = flat_weights(load_weights())
    arr = synthetic_arrivals(n_envs, rate=1000.0, horizon_s=ticks * 0.1 + 30, seed=3)
    outs = ("obs_post", "reward", "flags", "env_out")
    b64 = make_batch(arr, n_envs, 128, backend, outputs=outs)
    b32 = make_batch(arr, n_envs, 128, backend, outputs=outs, obs_dtype=torch.float32)
    for b in (b64, b32):
        b.reset()
        b.set_actor(w)
    for t in range(ticks):
        o64, o32 = b64.step_with_actor(), b32.step_with_actor()
        assert np.array_equal(_np(o64["reward"]), _np(o32["reward"])), "tick %d" % t
    for k in ("p", "v", "a", "id", "meta", "step"):
        assert np.array_equal(_np(b64.state_field(k)), _np(b32.state_field(k))), k
    assert b64.metrics() == b32.metrics()


def check_split_actor_long_horizon(backend, ticks=1000, n_envs=16, seed=9):
    """Long-horizon closed loop, split-half actor (default: f16 matrix instructions on hi / lo operand halves, pre-centered
    kernels, v_rsq / v_exp) against the exact float32 chain (PVE_CFG_ACTOR_F32) -- ADVICE r2: the default no longer
    reproduces the float32 actor's trajectory bit for bit, so the drift is measured and bounded here.
    The two actors agree to ~4e-5 per action; the closed loop amplifies a perturbation (~270x over 400 ticks),
    so individual trajectories separate after a few hundred ticks while every AGGREGATE of the evaluation protocol
    (main.py:523-526) stays put.  Returns (first tick at which any env's state differs, relative differences)."""
    from pve_mcc_amd.arrivals import synthetic_arrivals
    w = flat_weights(load_weights())
    arr = synthetic_arrivals(n_envs, rate=1000.0, horizon_s=ticks * 0.1 + 30, seed=seed)
    outs = ("obs_post", "reward", "flags", "env_out")
    bs = make_batch(arr, n_envs, 128, backend, outputs=outs)
    be = make_batch(arr, n_envs, 128, backend, outputs=outs, actor_f32=True)
    for b in (bs, be):
        b.reset()
        b.set_actor(w)
    first, worst_a = None, 0.0
    for t in range(ticks):
        if first is None:
            same = np.array_equal(_np(bs.state_field("p")), _np(be.state_field("p")))
            if same:
                a_s, a_e = _np(bs.act()), _np(be.act())
                worst_a = max(worst_a, float(np.abs(a_s - a_e).max()))
            else:
                first = t
        bs.step_with_actor(); be.step_with_actor()
    ms, me = bs.metrics(), be.metrics()
    # (synthetic Poisson streams are harsher than the shipped 1000 stream: a handful of the ~5000 vehicles do collide, in
    #  either form; the shipped stream's zero-collision run is check_closed_loop_on_device)
    assert abs(ms["collided"] - me["collided"]) <= max(6.0, 0.5 * me["collided"]), (ms["collided"], me["collided"])
    assert ms["spawned"] == me["spawned"] and ms["overflow"] == 0 and me["overflow"] == 0
    rel = {k: abs(ms[k] - me[k]) / max(1.0, abs(me[k])) for k in ("passed", "passed_steps", "alive_steps", "ctl_steps", "sum_reward", "sum_jerk", "locks")}
    assert worst_a <= 1e-4, "actions on identical states: %.3e" % worst_a
    assert rel["passed"] <= 0.01 and rel["passed_steps"] <= 0.01 and rel["alive_steps"] <= 0.01 and rel["sum_reward"] <= 0.02, rel
    rel["collided_split"], rel["collided_exact"], rel["spawned"] = ms["collided"], me["collided"], ms["spawned"]
    return first, worst_a, rel


# ------------------------------------------------------------------ the graph-level pin (round 5)
def load_graph_golden():
    """tests/golden/actor_graph.npz: (row, action) vectors produced by evaluating the reference's OWN MetaGraphDef
    (model_data/baseline/66.cptk.meta) op by op -- tests/golden/gen_actor_golden.py.  Returns (rows f32 [N,28], kinds,
    actions_f32, actions_f64, meta dict, well) where `well` masks out the constant non-zero rows: there the graph's
    x*inv - mean*inv form cancels two numbers of magnitude 1e6 in float32 (result good to ~0.1 only, in ANY float32
    evaluation order; such rows cannot occur: a row holds a position, a speed and a route)."""
    import json
    z = np.load(os.path.join(GOLDEN_DIR, "actor_graph.npz"))
    rows = z["rows"]
    const_nonzero = (np.ptp(rows, axis=1) == 0) & (rows[:, 0] != 0)
    return rows, z["kinds"], z["actions_f32"], z["actions_f64"], json.loads(str(z["meta"])), ~const_nonzero


def planted_batch(backend, n_envs, ticks=150, rate=1000.0, seed=31, **cfg):
    """A batch in a filled state (closed loop with the installed actor for `ticks` ticks) whose controlled slots the
    callers overwrite with golden rows.  Returns (batch, [(env, slot)] of the controlled slots)."""
    from pve_mcc_amd.arrivals import synthetic_arrivals
    arr = synthetic_arrivals(n_envs, rate=rate, horizon_s=ticks * 0.1 + 40, seed=seed)
    b = make_batch(arr, n_envs, 128, backend, outputs=("obs_post", "reward", "flags", "env_out", "new_slot"), **cfg)
    b.reset()
    b.set_actor(flat_weights(load_weights()))
    for _ in range(ticks):
        b.step_with_actor()
    ctl = _np(b.control_mask())
    return b, np.argwhere(ctl)


def actions_of_planted_rows(b, slots, rows):
    """Feed `rows` through the batch's actor entry point (pve_actor_forward), len(slots) rows per call, each row in the
    observation cell of a controlled slot; returns the actions in row order."""
    out = np.zeros(len(rows))
    for lo in range(0, len(rows), len(slots)):
        part = rows[lo:lo + len(slots)]
        sl = slots[:len(part)]
        obs = np.zeros((b.n_envs, b.capacity, 28), np.float32)
        obs[sl[:, 0], sl[:, 1]] = part
        b.obs.copy_(torch.as_tensor(obs).to(b.obs.dtype).to(b.device))
        a = _np(b.act())
        ctl = np.zeros(a.shape, bool)
        ctl[slots[:, 0], slots[:, 1]] = True
        assert np.all(a[~ctl] == 0), "uncontrolled slots must get 0 (main.py:401)"
        out[lo:lo + len(part)] = a[sl[:, 0], sl[:, 1]]
    return out


def check_actor_entry_point_vs_graph(backend, n_envs=8, **cfg):
    """pve_actor_forward (k_actor_h by default, k_actor_t with actor_f32=True, `actor_canonical` in the emulator) on the
    golden rows against the actions of the reference's graph: |a - a_graph_f32| and |a - a_graph_f64| <= ACTION_TOL."""
    rows, kinds, a32, a64, meta, well = load_graph_golden()
    b, slots = planted_batch(backend, n_envs, **cfg)
    assert len(slots) >= 20 * n_envs, len(slots)
    a = actions_of_planted_rows(b, slots, rows)
    e32, e64 = np.abs(a - a32), np.abs(a - a64)
    worst = {}
    for k in np.unique(kinds):
        m = (kinds == k) & well
        worst[str(k)] = (float(e32[m].max()), float(e64[m].max()))
    bad = np.flatnonzero(well & ((e32 > ACTION_TOL) | (e64 > ACTION_TOL)))
    assert len(bad) == 0, "rows %s: %s vs graph %s" % (bad[:8], a[bad[:8]], a32[bad[:8]])
    assert np.all(np.abs(a[~well] - a64[~well]) <= 0.2)        # the ill-conditioned constant rows: sane, not pinned
    zero = np.flatnonzero((np.abs(rows).max(axis=1) == 0))
    assert len(zero) >= 1 and e32[zero].max() <= 1e-5, "the all-zero row of a fresh vehicle (ref :380, :420)"
    return worst


def check_fused_rollout_actor_vs_graph(backend, n_envs=64, **cfg):
    """The actor INSIDE the resident kernel (k_rollout<.., ACT>, pve_step_many(PVE_SRC_ACTOR)) held to the graph's actions:
    two batches in the same state; batch G takes one plain tick with the graph's float32 actions as the tape, batch R
    gets the golden ROWS planted in its observation buffer and takes the same tick through step_many(1, source="actor").
    The accelerations the step applied must agree to ACTION_TOL (clip is 1-Lipschitz, the overrides of ref :1503-1520 do
    not depend on the action), everything discrete must be identical."""
    rows, kinds, a32, a64, meta, well = load_graph_golden()
    g, slots = planted_batch(backend, n_envs, **cfg)
    r, slots_r = planted_batch(backend, n_envs, **cfg)
    assert np.array_equal(slots, slots_r) and len(slots) >= len(rows), (len(slots), len(rows))
    for k in ("p", "v", "a", "id", "meta"):
        assert np.array_equal(_np(g.state_field(k)), _np(r.state_field(k))), k
    use = np.flatnonzero(well)
    sl = slots[:len(use)]
    obs = np.zeros((n_envs, 128, 28), np.float32)
    obs[sl[:, 0], sl[:, 1]] = rows[use]
    tape = np.zeros((n_envs, 128))
    tape[sl[:, 0], sl[:, 1]] = a32[use].astype(np.float64)
    # controlled slots beyond the golden rows: the action of the all-zero row (what the planted zero rows yield)
    zero_a = float(a32[np.flatnonzero(np.abs(rows).max(axis=1) == 0)[0]])
    rest = slots[len(use):]
    tape[rest[:, 0], rest[:, 1]] = zero_a
    a_before = _np(g.state_field("a")).copy()
    out_g = g.step(torch.as_tensor(tape).to(g.device))
    new_slot = _np(out_g["new_slot"]).copy()
    r.obs.copy_(torch.as_tensor(obs).to(r.obs.dtype).to(r.device))
    r.step_many(1, source="actor")
    for k in ("id", "meta", "step", "count"):
        assert np.array_equal(_np(g.state_field(k)), _np(r.state_field(k))), k
    a_g, a_r = _np(g.state_field("a")), _np(r.state_field("a"))
    worst = float(np.abs(a_g - a_r).max())
    assert worst <= ACTION_TOL, worst
    # how many of the planted actions the comparison actually saw (no override, no deletion)
    ns = new_slot[sl[:, 0], sl[:, 1]]
    alive = ns >= 0
    applied = a_g[sl[alive, 0], ns[alive]]
    seen = int((applied == np.clip(a32[use][alive].astype(np.float64), -3, 3)).sum())
    assert seen >= 0.5 * len(use), (seen, len(use))
    del a_before
    return worst, seen
