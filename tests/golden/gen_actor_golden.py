"""Pin the MADDPG actor against the reference's OWN shipped graph -- without TensorFlow.

Run in the build container only:   python tests/golden/gen_actor_golden.py
Output (committed):                tests/golden/actor_graph.npz

What the reference runs (main.py:36-45 -> model_agent_maddpg.py:23-49 `actor_network`, restored by
main.py:380-384 `saver.restore`) is stored next to the checkpoint as a MetaGraphDef:
/root/reference/model_data/baseline/66.cptk.meta.  This script

  1. decodes MetaGraphDef -> GraphDef -> NodeDef (name, op, inputs, attrs) with a minimal protobuf wire reader
     (tools/extract_actor.py:parse_proto),
  2. finds the action output `agent1actor/Mul` and walks the sub-graph back to the state placeholder along the
     graph's OWN edges (nothing about the network is assumed: ops, operand order, reduction axes, the three
     variance epsilons and the final gain are what the file says),
  3. evaluates that sub-graph with an interpreter of primitive ops only (Placeholder, Const, VariableV2, Identity,
     Mean, StopGradient, SquaredDifference, Add, Sub, Mul, Rsqrt, MatMul, BiasAdd, Relu, Tanh) -- once in float32
     (the graph's dtype, NumPy reductions) and once in float64 (the real-valued semantics of the same graph on the
     same float32 weights: every float32 evaluation order, TensorFlow's Eigen kernels included, lies within float32
     round-off of it),
  4. reads the variables from the checkpoint bundle (tools/extract_actor.py:load_bundle),
  5. emits rows -> actions: closed-loop states of the CPU oracle driven by THIS graph evaluation on the reference's
     1000 stream (main.py:test() protocol; the run's aggregates are stored too: SURVEY App. D), random rows at scales
     1 / 30 / 150, and the degenerate rows where the epsilon decides (all-zero row of a freshly spawned vehicle,
     ref :380/:420; constant rows; rows of variance ~1e-12).

The fixture holds numbers and a JSON description of the decoded op chain -- no reference source.
"""
import json
import os
import struct
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from extract_actor import load_bundle, parse_proto  # noqa: E402

META = "/root/reference/model_data/baseline/66.cptk.meta"
CKPT = "/root/reference/model_data/baseline/66.cptk"
OUT = os.path.join(HERE, "actor_graph.npz")
OUTPUT_NODE = "agent1actor/Mul"        # model_agent_maddpg.py:49 `tf.multiply(tf.nn.tanh(x), w_)` of agent "agent1" (main.py:372)

DT_FLOAT, DT_INT32 = 1, 3


# ------------------------------------------------------------------------------------------------ GraphDef decoding
def _packed_varints(b):
    out, p = [], 0
    while p < len(b):
        r, s = 0, 0
        while True:
            c = b[p]
            p += 1
            r |= (c & 0x7F) << s
            if not c & 0x80:
                break
            s += 7
        out.append(r)
    return out


def _sint(v, bits=64):
    return v - (1 << bits) if v >= 1 << (bits - 1) else v


def decode_shape(msg):
    """TensorShapeProto: field 2 = repeated Dim {1: size}"""
    dims = []
    for d in parse_proto(msg).get(2, []):
        dims.append(_sint(parse_proto(d).get(1, [0])[0]))
    return dims


def decode_tensor(msg):
    """TensorProto -> ndarray (float32 / int32 only: all the actor sub-graph holds)"""
    t = parse_proto(msg)
    dtype = t.get(1, [0])[0]
    shape = decode_shape(t[2][0]) if 2 in t else []
    n = int(np.prod(shape)) if shape else 1
    if dtype == DT_FLOAT:
        if 4 in t and len(t[4][0]):
            a = np.frombuffer(t[4][0], "<f4").copy()
        else:
            vals = []
            for v in t.get(5, []):             # float_val: packed (length-delimited) or single fixed32
                vals += list(np.frombuffer(v, "<f4"))
            a = np.array(vals, np.float32)
        if a.size == 1 and n > 1:
            a = np.full(n, a[0], np.float32)
        return a.reshape(shape)
    if dtype == DT_INT32:
        if 4 in t and len(t[4][0]):
            a = np.frombuffer(t[4][0], "<i4").copy()
        else:
            vals = []
            for v in t.get(7, []):             # int_val: packed varints or a single varint
                vals += [_sint(x) for x in _packed_varints(v)] if isinstance(v, (bytes, bytearray)) else [_sint(v)]
            a = np.array(vals, np.int32)
        if a.size == 1 and n > 1:
            a = np.full(n, a[0], np.int32)
        return a.reshape(shape)
    raise ValueError("tensor dtype %d not handled" % dtype)


def decode_attr(msg):
    """AttrValue -> python value (only the kinds the sub-graph uses)"""
    a = parse_proto(msg)
    if 8 in a:
        return decode_tensor(a[8][0])
    if 5 in a:
        return bool(a[5][0])
    if 3 in a:
        return _sint(a[3][0])
    if 4 in a:
        return struct.unpack("<f", a[4][0])[0]
    if 6 in a:
        return ("type", a[6][0])
    if 7 in a:
        return ("shape", decode_shape(a[7][0]))
    if 2 in a:
        return a[2][0].decode("latin1")
    return None


def load_graph(path=META):
    """{node name: dict(op, inputs, attrs)} of the MetaGraphDef's graph_def (field 2; GraphDef.node = field 1;
    NodeDef: 1 name, 2 op, 3 input, 5 attr map entries {1 key, 2 AttrValue})"""
    meta = parse_proto(open(path, "rb").read())
    gd = parse_proto(meta[2][0])
    nodes = {}
    for raw in gd[1]:
        n = parse_proto(raw)
        name = n[1][0].decode()
        attrs = {}
        for e in n.get(5, []):
            kv = parse_proto(e)
            attrs[kv[1][0].decode()] = kv[2][0]          # decoded lazily (most nodes are the trainer's)
        nodes[name] = dict(op=n[2][0].decode(), inputs=[i.decode() for i in n.get(3, [])], attrs=attrs)
    return nodes


def subgraph(nodes, out=OUTPUT_NODE):
    """Nodes reachable from `out` along data edges, in topological order (inputs first)."""
    order, seen = [], set()

    def visit(name):
        assert not name.startswith("^"), "control edge in the actor sub-graph"
        base = name.split(":")[0]
        if base in seen:
            return
        seen.add(base)
        for i in nodes[base]["inputs"]:
            visit(i)
        order.append(base)
    visit(out)
    return order


# ------------------------------------------------------------------------------------------------ interpreter
class GraphActor:
    """The `agent1actor` sub-graph as decoded from the reference's MetaGraphDef, evaluated op by op."""

    PRIMITIVES = ("Placeholder", "Const", "VariableV2", "Identity", "Mean", "StopGradient", "SquaredDifference",
                  "Add", "Sub", "Mul", "Rsqrt", "MatMul", "BiasAdd", "Relu", "Tanh")

    def __init__(self, meta_path=META, ckpt_prefix=CKPT):
        self.nodes = load_graph(meta_path)
        self.order = subgraph(self.nodes)
        self.variables = load_bundle(ckpt_prefix)
        ops = set(self.nodes[n]["op"] for n in self.order)
        assert ops <= set(self.PRIMITIVES), "unexpected ops %s" % (ops - set(self.PRIMITIVES))
        ph = [n for n in self.order if self.nodes[n]["op"] == "Placeholder"]
        assert len(ph) == 1, ph
        self.placeholder = ph[0]
        shp = decode_attr(self.nodes[ph[0]]["attrs"]["shape"])
        assert shp == ("shape", [-1, 28]), shp
        assert decode_attr(self.nodes[ph[0]]["attrs"]["dtype"]) == ("type", DT_FLOAT)

    def attr(self, node, key, default=None):
        a = self.nodes[node]["attrs"]
        return decode_attr(a[key]) if key in a else default

    def run(self, rows, dtype=np.float32, keep=False):
        """rows [N, 28] -> actions [N] (and every intermediate with keep=True).  dtype float32 = the graph's own type;
        float64 = the same op chain in double precision on the same float32 constants / variables / inputs."""
        F = dtype
        val = {}
        for name in self.order:
            nd = self.nodes[name]
            op = nd["op"]
            x = [val[i.split(":")[0]] for i in nd["inputs"]]
            if op == "Placeholder":
                v = np.asarray(rows, np.float32).astype(F)
            elif op == "Const":
                c = self.attr(name, "value")
                v = c.astype(F) if c.dtype == np.float32 else c
            elif op == "VariableV2":
                v = self.variables[name].astype(F)
            elif op in ("Identity", "StopGradient"):
                v = x[0]
            elif op == "Mean":
                axes = tuple(int(a) for a in np.atleast_1d(x[1]))
                v = np.mean(x[0], axis=axes, keepdims=bool(self.attr(name, "keep_dims", False)), dtype=F)
            elif op == "SquaredDifference":
                d = (x[0] - x[1]).astype(F)
                v = (d * d).astype(F)
            elif op == "Add":
                v = (x[0] + x[1]).astype(F)
            elif op == "Sub":
                v = (x[0] - x[1]).astype(F)
            elif op == "Mul":
                v = (x[0] * x[1]).astype(F)
            elif op == "Rsqrt":
                v = (F(1.0) / np.sqrt(x[0], dtype=F)).astype(F)
            elif op == "MatMul":
                a = x[0].T if self.attr(name, "transpose_a", False) else x[0]
                b = x[1].T if self.attr(name, "transpose_b", False) else x[1]
                v = (a @ b).astype(F)
            elif op == "BiasAdd":
                assert self.attr(name, "data_format", "NHWC") == "NHWC"
                v = (x[0] + x[1]).astype(F)
            elif op == "Relu":
                v = np.maximum(x[0], F(0))
            elif op == "Tanh":
                v = np.tanh(x[0], dtype=F)
            else:
                raise AssertionError(op)
            val[name] = v
        out = val[OUTPUT_NODE]
        assert out.shape == (len(rows), 1)
        return (out[:, 0], val) if keep else out[:, 0]

    def description(self):
        """What the decoded file says, for the fixture (asserted again by tests/test_actor_graph.py)."""
        chain = [[n, self.nodes[n]["op"], self.nodes[n]["inputs"]] for n in self.order]
        eps = {n: float(self.attr(n, "value")) for n in self.order if n.endswith("batchnorm/add/y")}
        axes = {n: [int(a) for a in np.atleast_1d(self.attr(n, "value"))] for n in self.order
                if n.endswith("reduction_indices")}
        keep = {n: bool(self.attr(n, "keep_dims", False)) for n in self.order if self.nodes[n]["op"] == "Mean"}
        mm = {n: [bool(self.attr(n, "transpose_a", False)), bool(self.attr(n, "transpose_b", False))]
              for n in self.order if self.nodes[n]["op"] == "MatMul"}
        gain = float(self.attr("agent1actor/Const", "value"))
        shapes = {n: list(self.variables[n].shape) for n in self.order if self.nodes[n]["op"] == "VariableV2"}
        return dict(output=OUTPUT_NODE, placeholder=self.placeholder, chain=chain, epsilon=eps, reduction_axes=axes,
                    keep_dims=keep, matmul_transpose=mm, gain=gain, variables=shapes)


# ------------------------------------------------------------------------------------------------ rows
def closed_loop_rows(actor, ticks=1000, every=25):
    """main.py:test() protocol on the reference's 1000 stream with the GRAPH as the policy, environment = CPU oracle
    (pinned to the reference separately).  Returns the sampled observation rows and the run's aggregates."""
    from oracle.oracle import OracleEnv
    from pve_mcc_amd.arrivals import load_arrival_mat
    arr = load_arrival_mat(os.path.join(HERE, "streams", "arvTimeNewVeh_new_1000_12.mat"))
    env = OracleEnv(arr)
    rows, alive, ctl, coll, locks, rew, k = [], 0, 0, 0, 0, [], 0
    for t in range(ticks):
        vid, c, obs0 = env.alive_view()
        a = np.zeros(len(vid))
        if c.any():
            r = obs0[c != 0]
            a[c != 0] = actor.run(r).astype(np.float64)
            for row in r:
                if k % every == 0:
                    rows.append(row)
                k += 1
        alive += len(vid)
        ctl += int(c.sum())
        rec = env.tick(a)
        coll += int((rec["coll_pv"] > 0).sum())
        locks += rec["lock"]
        rew += list(rec["reward"])
    agg = dict(alive_steps=alive, ctl_steps=ctl, id_seq=int(rec["id_seq"]), passed=int(rec["passed"]), collided=coll,
               locks=locks, pT_m=float(rec["passed_step_total"] / (rec["passed"] + 1e-4) * 0.1),
               reward_mean=float(np.mean(rew)))
    return np.asarray(rows, np.float64), agg


def make_rows(actor):
    rng = np.random.default_rng(66)
    cl, agg = closed_loop_rows(actor)
    parts, kinds = [cl.astype(np.float32)], ["closed_loop"] * len(cl)
    for scale in (1.0, 30.0, 150.0):
        parts.append((rng.standard_normal((300, 28)) * scale).astype(np.float32))
        kinds += ["random_%g" % scale] * 300
    # observation-shaped random rows: [p, v, a, route] x 7 in the env's value ranges, some neighbours absent (zeros)
    shaped = np.zeros((300, 7, 4), np.float32)
    shaped[..., 0] = rng.uniform(-5, 165, (300, 7))
    shaped[..., 1] = rng.uniform(5, 13, (300, 7))
    shaped[..., 2] = rng.uniform(-3, 3, (300, 7))
    shaped[..., 3] = rng.integers(0, 12, (300, 7))
    n_nb = rng.integers(0, 7, 300)
    for i in range(300):
        shaped[i, 1 + n_nb[i]:] = 0
    parts.append(shaped.reshape(300, 28))
    kinds += ["shaped"] * 300
    # where the variance epsilon decides: the all-zero row of a freshly spawned vehicle (ref :380, :420), constant rows,
    # rows whose variance is of the order of the epsilon itself
    deg = [np.zeros(28, np.float32), np.full(28, 1.0, np.float32), np.full(28, -7.5, np.float32)]
    for s in (1e-7, 3e-7, 1e-6, 3e-6, 1e-5, 1e-4):
        for _ in range(8):
            deg.append((rng.standard_normal(28) * s).astype(np.float32))
    parts.append(np.asarray(deg, np.float32))
    kinds += ["degenerate"] * len(deg)
    return np.concatenate(parts).astype(np.float32), np.asarray(kinds), agg


def main():
    actor = GraphActor()
    desc = actor.description()
    print("sub-graph: %d nodes; ops:" % len(actor.order), sorted(set(actor.nodes[n]["op"] for n in actor.order)))
    print("epsilon:", desc["epsilon"])
    print("axes:", desc["reduction_axes"], "gain:", desc["gain"])
    rows, kinds, agg = make_rows(actor)
    print("closed loop driven by the decoded graph:", agg)
    a32 = actor.run(rows, np.float32)
    a64 = actor.run(rows, np.float64)
    print("%d rows; |f32 - f64| max %.3e" % (len(rows), np.abs(a32 - a64).max()))
    # the variables of the sub-graph ARE the committed weight fixture
    z = np.load(os.path.join(HERE, "actor_66.npz"))
    names = {"ln0_beta": "agent1actor/LayerNorm/beta", "ln0_gamma": "agent1actor/LayerNorm/gamma",
             "w1": "agent1actor/dense/kernel", "b1": "agent1actor/dense/bias",
             "ln1_beta": "agent1actor/LayerNorm_1/beta", "ln1_gamma": "agent1actor/LayerNorm_1/gamma",
             "w2": "agent1actor/dense_1/kernel", "b2": "agent1actor/dense_1/bias",
             "ln2_beta": "agent1actor/LayerNorm_2/beta", "ln2_gamma": "agent1actor/LayerNorm_2/gamma",
             "w3": "agent1actor/dense_2/kernel", "b3": "agent1actor/dense_2/bias"}
    for k, v in names.items():
        assert np.array_equal(z[k], actor.variables[v]), k
    assert sorted(names.values()) == sorted(desc["variables"]), "the sub-graph reads exactly the 12 fixture tensors"
    desc["closed_loop_aggregates"] = agg
    desc["fixture_names"] = names
    desc["numpy"] = np.__version__
    np.savez_compressed(OUT, rows=rows, kinds=kinds, actions_f32=a32.astype(np.float32), actions_f64=a64.astype(np.float64),
                        meta=np.array(json.dumps(desc)))
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
