#!/usr/bin/env python3
"""tools/gemm_f32_bench.py -- in-graph time of the small f32 x f32 MUL_MATs of the reference's Token2Wav DiT (512 / 2048-wide projections over 50..56 frames x batch 2,
the attention products over 16 head-batches): R nodes with R different weight tensors (so the weights come from HBM, as in the window graph where 450 MB of them
cycle through) over one activation, submitted as one cgraph, hipGraph replay, HIP events.  Env toggles of gemm_any.hip apply (MI355X_NO_GEMM_F32_KSPLIT, ...)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402

SHAPES = [  # name, M, K, N, B (activation [K, N, B]), weight batch (1: broadcast)
    ("qkvo 512x512", 512, 512, 50, 2, 1), ("mlp1 2048x512", 2048, 512, 50, 2, 1), ("mlp2 512x2048", 512, 2048, 50, 2, 1), ("conv-like 512x1536", 512, 1536, 50, 2, 1),
    ("adaLN 4608x512 n1", 4608, 512, 1, 2, 1), ("attn KQ 200x64", 200, 64, 50, 16, 16), ("attn PV 64x200", 64, 200, 50, 16, 16), ("qkvo N=56", 512, 512, 56, 2, 1), ("mlp2 N=56", 512, 2048, 56, 2, 1)]


def main():
    pkg = load_pkg()
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F32, Context
    be = pkg.backend(0)
    R = 48
    rng = np.random.default_rng(0)
    only = sys.argv[1] if len(sys.argv) > 1 else None
    for name, M, K, N, B, WB in SHAPES:
        if only and only not in name:
            continue
        c = Context(be)
        x = c.new_tensor(GGML_TYPE_F32, K, N, B)
        ws = [c.new_tensor(GGML_TYPE_F32, K, M, WB) for _ in range(R)]
        ys = [c.mul_mat(w, x) for w in ws]
        c.alloc()
        be.tensor_set(x, rng.standard_normal(K * N * B).astype(np.float32))
        wv = rng.standard_normal(K * M * WB).astype(np.float32)
        for w in ws:
            be.tensor_set(w, wv)
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
        nk = be.get_stat("kernels_last_graph")
        flops = 2.0 * M * K * N * B
        print(f"{name:22s} M={M:5d} K={K:5d} N={N:3d} B={B:2d}: {best * 1e3 / R:7.2f} us per product ({int(nk)} launches / {R}), {flops / (best * 1e-3 / R) * 1e-12:6.2f} TFLOP/s, weights {M * K * WB * 4 / (best * 1e-3 / R) * 1e-9:7.1f} GB/s", flush=True)
        c.free()


if __name__ == "__main__":
    main()
