#!/usr/bin/env python3
"""tools/attn_f32_bench.py -- the f32 attention chain of the encoders' flash-attention-off graphs (MUL_MAT K.Q -> SOFT_MAX_EXT -> MUL_MAT V^T.P -> PERMUTE + CONT: vision.cpp:670-690 at
SigLip2's shape: head size 72, 16 heads, 1024 patches; env HD / NQ / NKV / NH) as one k_attn_f32 launch per chain: 24 chains over 4 Q / K / V sets in one graph, hipGraph replay,
HIP events; result of the first chain against float64."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402


def main():
    pkg = load_pkg()
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F32, Context
    be = pkg.backend(0)
    D = int(os.environ.get("HD", "72"))
    nq, nkv, nh = int(os.environ.get("NQ", "1024")), int(os.environ.get("NKV", "1024")), int(os.environ.get("NH", "16"))
    c = Context(be)
    sets = [(c.new_tensor(GGML_TYPE_F32, D, nq, nh), c.new_tensor(GGML_TYPE_F32, D, nkv, nh), c.new_tensor(GGML_TYPE_F32, nkv, D, nh)) for _ in range(4)]
    scale = 1.0 / np.sqrt(D)
    outs = []
    nodes = 24
    for i in range(nodes):
        q, k, v = sets[i % 4]
        kq = c.soft_max_ext(c.mul_mat(k, q), None, scale, 0.0)
        kqv = c.mul_mat(v, kq)
        outs.append(c.cont(c.permute(kqv, 0, 2, 1, 3), D * nh, nq))
    c.alloc()
    rng = np.random.default_rng(0)
    vals = []
    for q, k, v in sets:
        a = [rng.standard_normal(t.nelements()).astype(np.float32) for t in (q, k, v)]
        for t, x in zip((q, k, v), a):
            be.tensor_set(t, x)
        vals.append(a)
    g = c.graph()
    for _ in range(3):
        be.graph_compute(g)
    be.synchronize()
    print("launches in the graph:", be.get_stat("kernels_last_graph"), "for", nodes, "chains")
    best = 1e9
    for _ in range(5):
        be.synchronize(); t0 = time.perf_counter()
        for _ in range(4):
            be.graph_compute(g)
        be.synchronize(); best = min(best, (time.perf_counter() - t0) / 4)
    us = best * 1e6 / nodes
    fl = 4.0 * D * nq * nkv * nh
    print(f"D={D} nq={nq} nkv={nkv} heads={nh}: {us:.1f} us per chain  ({fl / us / 1e6:.1f} TFLOP/s f32)")
    got = be.tensor_get(outs[0]).reshape(nq, nh, D)
    qv, kv, vv = vals[0]
    Q = qv.reshape(nh, nq, D).astype(np.float64); K = kv.reshape(nh, nkv, D).astype(np.float64); V = vv.reshape(nh, D, nkv).astype(np.float64)
    S = np.einsum("hqd,hkd->hqk", Q, K) * scale
    P = np.exp(S - S.max(-1, keepdims=True)); P /= P.sum(-1, keepdims=True)
    O = np.einsum("hqk,hdk->qhd", P, V)
    print("nmse vs float64: %.2e" % (float(((got - O) ** 2).sum() / (O ** 2).sum())))


if __name__ == "__main__":
    main()
