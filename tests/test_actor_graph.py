"""f1, graph-level pin (round 5): the MADDPG actor is pinned against the reference's OWN shipped graph,
model_data/baseline/66.cptk.meta (MetaGraphDef), decoded and evaluated op by op without TensorFlow
(tests/golden/gen_actor_golden.py -> tests/golden/actor_graph.npz).  What is asserted here:
  * the decoded file's facts: op chain, operand order, reduction axes, the three variance epsilons, the gain;
  * oracle/actor_np.py == the graph evaluation (float32) on every golden row;
  * the closed loop driven by the graph reproduces SURVEY App. D's pretrained-actor row;
  * the C-ABI actor entry point (emulated kernels: csrc/pve_actor.h `actor_canonical`) against the graph's actions;
  * (build container) a fresh decode of the live file == the fixture."""
import numpy as np
import pytest

from oracle.actor_np import actor_forward, load_weights
from tests import actor_scenarios as A

LN = ("agent1actor/LayerNorm", "agent1actor/LayerNorm_1", "agent1actor/LayerNorm_2")


def ulps(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return np.abs(a - b) / np.spacing(np.maximum(np.abs(a), np.abs(b)).astype(np.float32))


def test_fixture_holds_what_the_verdict_asked_for():
    rows, kinds, a32, a64, meta, well = A.load_graph_golden()
    assert rows.shape[0] >= 2000 and rows.shape[1] == 28 and rows.dtype == np.float32
    assert (kinds == "closed_loop").sum() >= 1000
    for k in ("random_1", "random_30", "random_150"):
        assert (kinds == k).sum() == 300
    assert (~well).sum() == 2 and np.all(np.abs(a32) <= 3.0) and np.all(np.abs(a64) <= 3.0)
    # float32 round-off of the graph itself on real states: the reason the parity bar is 5e-4 and not 1e-6
    cl = kinds == "closed_loop"
    assert 1e-5 < np.abs(a32[cl] - a64[cl]).max() < A.ACTION_TOL


def test_decoded_graph_facts():
    """model_agent_maddpg.py:23-49 as the MetaGraphDef spells it out."""
    meta = A.load_graph_golden()[4]
    assert meta["output"] == "agent1actor/Mul" and meta["placeholder"] == "Placeholder" and meta["gain"] == 3.0
    # tc.layers.layer_norm: variance_epsilon 1e-12 (as a float32 constant), moments over the LAST axis, keep_dims
    assert sorted(meta["epsilon"]) == sorted(l + "/batchnorm/add/y" for l in LN)
    assert all(np.float32(v) == np.float32(1e-12) for v in meta["epsilon"].values())
    assert len(meta["reduction_axes"]) == 6 and all(v == [1] for v in meta["reduction_axes"].values())
    assert len(meta["keep_dims"]) == 6 and all(meta["keep_dims"].values())
    assert all(v == [False, False] for v in meta["matmul_transpose"].values()) and len(meta["matmul_transpose"]) == 3
    assert meta["variables"] == {
        "agent1actor/LayerNorm/beta": [28], "agent1actor/LayerNorm/gamma": [28],
        "agent1actor/dense/kernel": [28, 64], "agent1actor/dense/bias": [64],
        "agent1actor/LayerNorm_1/beta": [64], "agent1actor/LayerNorm_1/gamma": [64],
        "agent1actor/dense_1/kernel": [64, 64], "agent1actor/dense_1/bias": [64],
        "agent1actor/LayerNorm_2/beta": [64], "agent1actor/LayerNorm_2/gamma": [64],
        "agent1actor/dense_2/kernel": [64, 1], "agent1actor/dense_2/bias": [1]}
    node = {n: (op, inp) for n, op, inp in meta["chain"]}
    # the data path: LN -> dense -> LN_1 -> Relu -> dense_1 -> LN_2 -> Relu_1 -> dense_2 -> Tanh -> * 3
    path = [("agent1actor/Mul", "Mul", ["agent1actor/Tanh", "agent1actor/Const"]),
            ("agent1actor/Tanh", "Tanh", ["agent1actor/dense_2/BiasAdd"]),
            ("agent1actor/dense_2/BiasAdd", "BiasAdd", ["agent1actor/dense_2/MatMul", "agent1actor/dense_2/bias/read"]),
            ("agent1actor/dense_2/MatMul", "MatMul", ["agent1actor/Relu_1", "agent1actor/dense_2/kernel/read"]),
            ("agent1actor/Relu_1", "Relu", ["agent1actor/LayerNorm_2/batchnorm/add_1"]),
            ("agent1actor/dense_1/MatMul", "MatMul", ["agent1actor/Relu", "agent1actor/dense_1/kernel/read"]),
            ("agent1actor/Relu", "Relu", ["agent1actor/LayerNorm_1/batchnorm/add_1"]),
            ("agent1actor/dense/MatMul", "MatMul", ["agent1actor/LayerNorm/batchnorm/add_1", "agent1actor/dense/kernel/read"])]
    for n, op, inp in path:
        assert node[n] == (op, inp), (n, node[n])
    src = {LN[0]: "Placeholder", LN[1]: "agent1actor/dense/BiasAdd", LN[2]: "agent1actor/dense_1/BiasAdd"}
    for l in LN:      # y = x * (rsqrt(var + eps) * gamma) + (beta - mean * (rsqrt(var + eps) * gamma)), biased variance
        x = src[l]
        assert node[l + "/moments/mean"] == ("Mean", [x, l + "/moments/mean/reduction_indices"])
        assert node[l + "/moments/SquaredDifference"] == ("SquaredDifference", [x, l + "/moments/StopGradient"])
        assert node[l + "/moments/variance"] == ("Mean", [l + "/moments/SquaredDifference", l + "/moments/variance/reduction_indices"])
        assert node[l + "/batchnorm/add"] == ("Add", [l + "/moments/variance", l + "/batchnorm/add/y"])
        assert node[l + "/batchnorm/Rsqrt"] == ("Rsqrt", [l + "/batchnorm/add"])
        assert node[l + "/batchnorm/mul"] == ("Mul", [l + "/batchnorm/Rsqrt", l + "/gamma/read"])
        assert node[l + "/batchnorm/mul_1"] == ("Mul", [x, l + "/batchnorm/mul"])
        assert node[l + "/batchnorm/mul_2"] == ("Mul", [l + "/moments/mean", l + "/batchnorm/mul"])
        assert node[l + "/batchnorm/sub"] == ("Sub", [l + "/beta/read", l + "/batchnorm/mul_2"])
        assert node[l + "/batchnorm/add_1"] == ("Add", [l + "/batchnorm/mul_1", l + "/batchnorm/sub"])
    assert len(meta["chain"]) == 78


def test_numpy_restatement_equals_the_graph_evaluation():
    """oracle/actor_np.py (what every closed-loop test of the HIP actor is certified against) vs the graph's actions:
    <= 2 ulp float32 on EVERY golden row, the ill-conditioned ones included (same NumPy primitives, the restatement's
    own composition of them)."""
    rows, kinds, a32, a64, meta, well = A.load_graph_golden()
    a = actor_forward(load_weights(), rows)
    assert a.dtype == np.float32
    assert ulps(a, a32).max() <= 2.0


def test_graph_driven_closed_loop_is_survey_appendix_d():
    agg = A.load_graph_golden()[4]["closed_loop_aggregates"]
    assert (agg["alive_steps"], agg["ctl_steps"], agg["id_seq"], agg["passed"], agg["collided"], agg["locks"]) == \
        (72416, 37295, 323, 281, 0, 548)
    assert abs(agg["pT_m"] - 12.294) < 1e-3 and abs(agg["reward_mean"] - 1.30294) < 1e-4


def test_weight_fixture_is_what_the_graph_reads():
    meta = A.load_graph_golden()[4]
    assert sorted(meta["fixture_names"].values()) == sorted(meta["variables"])
    assert sorted(meta["fixture_names"]) == sorted(load_weights())


def test_actor_entry_point_vs_graph_emulated():
    worst = A.check_actor_entry_point_vs_graph("emu", n_envs=6, ticks=120)
    print("emulated actor vs graph (|a - f32|, |a - f64|) per kind:", worst)


@pytest.mark.reference
def test_live_decode_of_the_reference_graph_equals_the_fixture():
    """Build container only: decode /root/reference/model_data/baseline/66.cptk.meta again, evaluate, compare."""
    from tests.golden.gen_actor_golden import GraphActor
    rows, kinds, a32, a64, meta, well = A.load_graph_golden()
    g = GraphActor()
    d = g.description()
    for k in ("output", "placeholder", "epsilon", "reduction_axes", "keep_dims", "matmul_transpose", "gain", "variables"):
        assert d[k] == meta[k], k
    assert [list(c) for c in d["chain"]] == meta["chain"]
    assert np.array_equal(g.run(rows, np.float32), a32)
    assert np.allclose(g.run(rows, np.float64), a64, rtol=0, atol=1e-12)
    # a wrong epsilon WOULD show: the rows of variance ~1e-12 move by far more than the parity bar
    eps_nodes = [n for n in g.order if n.endswith("batchnorm/add/y")]
    saved = {n: g.nodes[n]["attrs"]["value"] for n in eps_nodes}
    import struct
    for n in eps_nodes:       # patch the TensorProto's float_val to 1e-5 (the Keras default, a plausible wrong guess)
        g.nodes[n]["attrs"]["value"] = saved[n].replace(struct.pack("<f", 1e-12), struct.pack("<f", 1e-5))
    moved = np.abs(g.run(rows, np.float32) - a32)
    assert moved[kinds == "degenerate"].max() > 10 * A.ACTION_TOL
