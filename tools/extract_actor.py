#!/usr/bin/env python3
"""Extract the pretrained actor of the reference (model_data/baseline/66.cptk.*) WITHOUT TensorFlow and
store it as a small data fixture (tests/golden/actor_66.npz, 12 float32 tensors, 6393 weights).

The checkpoint is a TF "tensor bundle": `66.cptk.index` is a LevelDB-style SSTable (48-byte footer with
magic 0xdb4775248b80fb57, uncompressed blocks, prefix-compressed keys) whose values are BundleEntryProto
messages (field 1 dtype, 2 shape, 4 offset, 5 size); `66.cptk.data-00000-of-00001` holds the raw
little-endian tensors.  Build container only (needs /root/reference)."""
import os
import struct

import numpy as np

CK = "/root/reference/model_data/baseline/66.cptk"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "actor_66.npz")


def varint(buf, p):
    r, s = 0, 0
    while True:
        b = buf[p]
        p += 1
        r |= (b & 0x7F) << s
        if not b & 0x80:
            return r, p
        s += 7


def read_block(buf, off, size):
    blk = buf[off:off + size]
    n_restarts = struct.unpack_from("<I", blk, len(blk) - 4)[0]
    end = len(blk) - 4 - 4 * n_restarts
    p, key, out = 0, b"", []
    while p < end:
        shared, p = varint(blk, p)
        non_shared, p = varint(blk, p)
        vlen, p = varint(blk, p)
        key = key[:shared] + blk[p:p + non_shared]
        p += non_shared
        out.append((key, blk[p:p + vlen]))
        p += vlen
    return out


def parse_proto(msg):
    """minimal protobuf wire parser -> {field: [values]} (varints and length-delimited only)"""
    p, out = 0, {}
    while p < len(msg):
        tag, p = varint(msg, p)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, p = varint(msg, p)
        elif wt == 2:
            n, p = varint(msg, p)
            v = msg[p:p + n]
            p += n
        elif wt == 5:
            v = msg[p:p + 4]
            p += 4
        elif wt == 1:
            v = msg[p:p + 8]
            p += 8
        else:
            raise ValueError("wire type %d" % wt)
        out.setdefault(f, []).append(v)
    return out


def load_bundle(prefix):
    idx = open(prefix + ".index", "rb").read()
    data = open(prefix + ".data-00000-of-00001", "rb").read()
    footer = idx[-48:]
    assert struct.unpack("<Q", footer[40:])[0] == 0xdb4775248b80fb57, "bad SSTable magic"
    p = 0
    _mi_off, p = varint(footer, p)
    _mi_size, p = varint(footer, p)
    ix_off, p = varint(footer, p)
    ix_size, p = varint(footer, p)
    tensors = {}
    for _k, handle in read_block(idx, ix_off, ix_size):
        boff, q = varint(handle, 0)
        bsize, q = varint(handle, q)
        for key, val in read_block(idx, boff, bsize):
            if not key:
                continue                                   # BundleHeaderProto
            e = parse_proto(val)
            dtype = e.get(1, [0])[0]
            shape = []
            if 2 in e:
                for dim in parse_proto(e[2][0]).get(2, []):
                    shape.append(parse_proto(dim).get(1, [0])[0])
            off = e.get(4, [0])[0]
            size = e.get(5, [0])[0]
            if dtype != 1:
                continue                                   # float32 only
            tensors[key.decode()] = np.frombuffer(data, dtype="<f4", count=size // 4, offset=off).reshape(shape).copy()
    return tensors


def main():
    t = load_bundle(CK)
    names = {"ln0_beta": "agent1actor/LayerNorm/beta", "ln0_gamma": "agent1actor/LayerNorm/gamma",
             "w1": "agent1actor/dense/kernel", "b1": "agent1actor/dense/bias",
             "ln1_beta": "agent1actor/LayerNorm_1/beta", "ln1_gamma": "agent1actor/LayerNorm_1/gamma",
             "w2": "agent1actor/dense_1/kernel", "b2": "agent1actor/dense_1/bias",
             "ln2_beta": "agent1actor/LayerNorm_2/beta", "ln2_gamma": "agent1actor/LayerNorm_2/gamma",
             "w3": "agent1actor/dense_2/kernel", "b3": "agent1actor/dense_2/bias"}
    out = {k: t[v] for k, v in names.items()}
    shapes = {k: v.shape for k, v in out.items()}
    print(len(t), "tensors in the bundle; actor:", shapes, "total", sum(v.size for v in out.values()))
    assert shapes["w1"] == (28, 64) and shapes["w2"] == (64, 64) and shapes["w3"] == (64, 1)
    np.savez_compressed(OUT, **out)
    print("wrote", os.path.abspath(OUT), os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
