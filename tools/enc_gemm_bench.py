#!/usr/bin/env python3
"""tools/enc_gemm_bench.py -- the F16-weight GEMM shapes of the omni encoders (Whisper-medium: 1500 frames x 1024 / 4096; SigLip2: 1024 patches x 1152 / 4304) in-graph:
12 different weight tensors per shape over one activation, hipGraph replay off / on as the executor decides, HIP events.  MI355X_GEMM_BM, MI355X_GEMM_256,
MI355X_GEMM_SPLIT_TARGET and the short-K split rule are read when the library loads: run once per setting."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0)
rng = np.random.default_rng(0); REP = 12
for name, N, M, K in (("whisper q/k/v/o", 1500, 1024, 1024), ("whisper fc1", 1500, 4096, 1024), ("whisper fc2", 1500, 1024, 4096),
                      ("siglip q/k/v/o", 1024, 1152, 1152), ("siglip fc1", 1024, 4304, 1152), ("siglip fc2", 1024, 1152, 4304), ("whisper chunk q", 100, 1024, 1024)):
    c = Context(be)
    x = c.new_tensor(GGML_TYPE_F32, K, N)
    ws = [c.new_tensor(GGML_TYPE_F16, K, M) for _ in range(REP)]
    ys = [c.mul_mat(w, x) for w in ws]
    c.alloc()
    wv = (rng.standard_normal(M * K, dtype=np.float32) * 0.05).astype(np.float16)
    for w in ws:
        be.tensor_set(w, wv)
    be.tensor_set(x, rng.standard_normal(K * N, dtype=np.float32))
    g = c.graph()
    for _ in range(3): be.graph_compute(g)
    be.synchronize(); best = 1e9
    for _ in range(5):
        a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); be.synchronize(); best = min(best, be.elapsed_ms(a, b))
    print(f"{name:18s} N={N:5d} M={M:5d} K={K:5d}  {best * 1e3 / REP:8.1f} us  {2.0 * M * K * N / (best * 1e-3 / REP) / 1e12:7.1f} TFLOP/s  ({int(be.get_stat('kernels_last_graph'))} launches / {REP})", flush=True)
    c.free()
