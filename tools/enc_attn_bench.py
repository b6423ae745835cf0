#!/usr/bin/env python3
"""tools/enc_attn_bench.py -- the encoders' FLASH_ATTN_EXT shape (head size 64, F16 K / V, no mask, NQ = NKV = 1500 x 16 heads = Whisper-medium over 30 s; env NQ / NKV / NH)
in-graph: 24 nodes over 4 K / V sets, hipGraph replay, HIP events.  The launch-choice switches of fattn_mma.hip apply (MI355X_FA_NO_KVSPLIT, MI355X_FA_KS, MI355X_FA_SQ,
MI355X_FA_BIG_MIN_WGS)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402


def main():
    pkg = load_pkg()
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
    be = pkg.backend(0)
    D = int(os.environ.get("HD", "64"))
    nq, nkv, nh = int(os.environ.get("NQ", "1500")), int(os.environ.get("NKV", "1500")), int(os.environ.get("NH", "16"))
    nhkv = int(os.environ.get("NHKV", str(nh)))                  # GQA: KV heads
    causal = os.environ.get("MASK", "") == "causal"              # a prefill ubatch's mask (the last nq cells of the view are the ubatch's own)
    c = Context(be)
    q = c.new_tensor(GGML_TYPE_F32, D, nq, nh)
    m = c.new_tensor(GGML_TYPE_F16, nkv, nq) if causal else None
    sets = [(c.new_tensor(GGML_TYPE_F16, D, nkv, nhkv), c.new_tensor(GGML_TYPE_F16, D, nkv, nhkv)) for _ in range(4)]
    nodes = 24
    for i in range(nodes):
        k, v = sets[i % 4]
        c.flash_attn_ext(q, k, v, m, 1.0 / np.sqrt(D))
    c.alloc()
    if causal:
        mk = np.zeros((nq, nkv), np.float16)
        for i in range(nq):
            mk[i, nkv - nq + i + 1:] = -np.inf
        be.tensor_set(m, mk)
    rng = np.random.default_rng(0)
    be.tensor_set(q, rng.standard_normal(q.nelements()).astype(np.float32))
    for k, v in sets:
        be.tensor_set(k, (rng.standard_normal(k.nelements()) * 0.5).astype(np.float16))
        be.tensor_set(v, (rng.standard_normal(v.nelements()) * 0.5).astype(np.float16))
    g = c.graph()
    for _ in range(3):
        be.graph_compute(g)
    be.synchronize()
    best = 1e9
    for _ in range(7):
        a, b = be.timed_event(), be.timed_event()
        be.record(a); be.graph_compute(g); be.record(b)
        be.synchronize()
        best = min(best, be.elapsed_ms(a, b))
    us = best * 1e3 / nodes
    print(f"D={D} nq={nq} nkv={nkv} heads={nh}/{nhkv}{' causal' if causal else ''}: {us:.2f} us per node, {4.0 * D * nq * nkv * nh / us * 1e-6:.1f} TFLOP/s ({int(be.get_stat('kernels_last_graph'))} launches / {nodes})", flush=True)
    c.free()


if __name__ == "__main__":
    main()
