"""NumPy float32 restatement of the reference's MADDPG actor forward pass. TEST INFRASTRUCTURE ONLY.

model_agent_maddpg.py:23-49: x(28) -> LayerNorm -> Dense64 -> LayerNorm -> ReLU -> Dense64 -> LayerNorm -> ReLU
-> Dense1 -> 3*tanh, all float32 (placeholder dtype, :15).  `tc.layers.layer_norm(center=True, scale=True)`
normalises over the last axis with the biased variance and variance_epsilon = 1e-12 (the Const nodes of the graph).

Parity status: PINNED against the reference's own graph (model_data/baseline/66.cptk.meta, a MetaGraphDef decoded and
evaluated op by op by tests/golden/gen_actor_golden.py): tests/test_actor_graph.py holds this restatement to <= 2 ulp of the
graph's float32 evaluation on 2 743 rows, and asserts the op chain / reduction axes / variance_epsilon (float32(1e-12)) from
the decoded file.  The closed-loop known answers of SURVEY.md App. D (tests/test_actor.py) are a second, indirect pin."""
import os

import numpy as np

F = np.float32


def load_weights(path=None):
    if path is None:
        path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "actor_66.npz")
    z = np.load(path)
    return {k: z[k].astype(np.float32) for k in z.files}


def layer_norm(x, gamma, beta):
    mean = x.mean(axis=-1, keepdims=True, dtype=F)
    var = np.mean(np.square(x - mean, dtype=F), axis=-1, keepdims=True, dtype=F)
    inv = (F(1.0) / np.sqrt(var + F(1e-12), dtype=F)) * gamma
    return (x * inv + (beta - mean * inv)).astype(F)


def actor_forward(w, obs):
    """obs [..., 28] (any float) -> action [...] float32 in [-3, 3]"""
    x = np.asarray(obs).astype(F)
    x = layer_norm(x, w["ln0_gamma"], w["ln0_beta"])
    x = (x @ w["w1"] + w["b1"]).astype(F)
    x = np.maximum(layer_norm(x, w["ln1_gamma"], w["ln1_beta"]), F(0))
    x = (x @ w["w2"] + w["b2"]).astype(F)
    x = np.maximum(layer_norm(x, w["ln2_gamma"], w["ln2_beta"]), F(0))
    y = (x @ w["w3"] + w["b3"]).astype(F)[..., 0]
    return (np.tanh(y, dtype=F) * F(3.0)).astype(F)


def flat_weights(w):
    """The 6393 float32 weights in the order the HIP kernel expects (see include/pve_env.h pve_actor_forward)."""
    order = ("ln0_gamma", "ln0_beta", "w1", "b1", "ln1_gamma", "ln1_beta", "w2", "b2", "ln2_gamma", "ln2_beta",
             "w3", "b3")
    return np.concatenate([np.asarray(w[k], np.float32).ravel() for k in order])
