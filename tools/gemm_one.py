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
COLD = bool(os.environ.get("GEMM_COLD"))                        # a weight tensor of its own per node (REP x the bytes: past the Infinity Cache when REP * M * K * 2 > 256 MB)
ws = [c.new_tensor(GGML_TYPE_F16, K, M) for _ in range(REP if COLD else 1)]; w = ws[0]; x = c.new_tensor(GGML_TYPE_F32, K, N)
ys = [c.mul_mat(ws[i % len(ws)], x) for i in range(REP)]
c.alloc()
if os.environ.get("GEMM_ZEROS"):                                 # DVFS check: zero operands draw less power per MFMA, the same schedule then runs at a higher clock
    be.tensor_set(w, np.zeros(M * K, np.float16)); be.tensor_set(x, np.zeros(K * N, np.float32))
else:
    wv = (rng.standard_normal(M * K, dtype=np.float32) * 0.05).astype(np.float16)
    for t in ws: be.tensor_set(t, wv)
    be.tensor_set(x, rng.standard_normal(K * N, dtype=np.float32))
g = c.graph()
for _ in range(2): be.graph_compute(g)
be.synchronize(); best = 1e9
for _ in range(5):
    a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); best = min(best, be.elapsed_ms(a, b))
tag = " ".join(f"{k[7:]}={v}" for k, v in os.environ.items() if k.startswith("MI355X_GEMM")) + (" zeros" if os.environ.get("GEMM_ZEROS") else "") + (" cold" if COLD else "")
print(f"[{tag}] M={M} K={K} N={N}: {best * 1e3 / REP:8.1f} us  {2.0 * M * K * N / (best * 1e-3 / REP) / 1e12:7.1f} TFLOP/s", flush=True)
