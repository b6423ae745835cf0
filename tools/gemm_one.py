#!/usr/bin/env python3
"""tools/gemm_one.py M K N [REP] -- one GEMM shape through the C-ABI, TFLOP/s (HIP events around REP nodes)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0); be.set_option("graphs", 0)
M, K, N = (int(v) for v in sys.argv[1:4]); REP = int(sys.argv[4]) if len(sys.argv) > 4 else 10
rng = np.random.default_rng(0)
c = Context(be)
w = c.new_tensor(GGML_TYPE_F16, K, M); x = c.new_tensor(GGML_TYPE_F32, K, N)
ys = [c.mul_mat(w, x) for _ in range(REP)]
c.alloc()
if os.environ.get("GEMM_ZEROS"):                                 # DVFS check: zero operands draw less power per MFMA, the same schedule then runs at a higher clock
    be.tensor_set(w, np.zeros(M * K, np.float16)); be.tensor_set(x, np.zeros(K * N, np.float32))
else:
    be.tensor_set(w, (rng.standard_normal(M * K, dtype=np.float32) * 0.05).astype(np.float16)); be.tensor_set(x, rng.standard_normal(K * N, dtype=np.float32))
g = c.graph()
for _ in range(2): be.graph_compute(g)
be.synchronize(); best = 1e9
for _ in range(5):
    a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); best = min(best, be.elapsed_ms(a, b))
tag = " ".join(f"{k[7:]}={v}" for k, v in os.environ.items() if k.startswith("MI355X_GEMM")) + (" zeros" if os.environ.get("GEMM_ZEROS") else "")
print(f"[{tag}] M={M} K={K} N={N}: {best * 1e3 / REP:8.1f} us  {2.0 * M * K * N / (best * 1e-3 / REP) / 1e12:7.1f} TFLOP/s", flush=True)
