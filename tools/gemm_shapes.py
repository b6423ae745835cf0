#!/usr/bin/env python3
"""tools/gemm_shapes.py -- the four GEMM shapes of a Qwen3-8B F16 prefill layer at ubatch 2048 / 512 (C3), per tile variant
(MI355X_GEMM_256 / MI355X_GEMM_BM are read when the library loads: run once per setting)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0); be.set_option("graphs", 0)
rng = np.random.default_rng(0); REP = 10
tag = "256=%s BM=%s" % (os.environ.get("MI355X_GEMM_256", "-"), os.environ.get("MI355X_GEMM_BM", "-"))
for N in (2048, 512):
    for name, M, K in (("qkv", 6144, 4096), ("wo", 4096, 4096), ("gate+up", 24576, 4096), ("down", 4096, 12288)):
        c = Context(be)
        w = c.new_tensor(GGML_TYPE_F16, K, M); x = c.new_tensor(GGML_TYPE_F32, K, N)
        ys = [c.mul_mat(w, x) for _ in range(REP)]
        c.alloc()
        be.tensor_set(w, (rng.standard_normal(M * K, dtype=np.float32) * 0.05).astype(np.float16)); be.tensor_set(x, rng.standard_normal(K * N, dtype=np.float32))
        g = c.graph()
        for _ in range(2): be.graph_compute(g)
        be.synchronize(); best = 1e9
        for _ in range(5):
            a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); best = min(best, be.elapsed_ms(a, b))
        print(f"[{tag}] N={N:5d} {name:8s} M={M:6d} K={K:6d}  {best * 1e3 / REP:8.1f} us  {2.0 * M * K * N / (best * 1e-3 / REP) / 1e12:7.1f} TFLOP/s", flush=True)
        c.free()
