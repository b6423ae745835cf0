#!/usr/bin/env python3
"""tools/gemm_sk_check.py -- the stream-K GEMM (gemm_sk.hip) against numpy on shapes that exercise tile / segment boundaries, grouped launches and the residual
epilogue; prints max relative error per shape and the number of stream-K launches the library counted."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0); be.set_option("graphs", 0); be.set_option("gemm_sk", 1)
rng = np.random.default_rng(0)
shapes = [((4096,), 4096, 512, True), ((4096,), 12288, 512, True), ((4096, 1024, 1024), 4096, 512, False), ((1000, 520), 1024, 300, False), ((4096,), 4096, 2048, True),
          ((384,), 8192, 129, False), ((128,), 64 * 300, 256, True)]
if len(sys.argv) > 1: shapes = shapes[:int(sys.argv[1])]
worst = 0.0
for Ms, K, N, resid in shapes:
    c = Context(be)
    x = c.new_tensor(GGML_TYPE_F32, K, N)
    ws = [c.new_tensor(GGML_TYPE_F16, K, M) for M in Ms]
    ys = [c.mul_mat(w, x) for w in ws]
    rs = []
    if resid:
        rs = [c.new_tensor(GGML_TYPE_F32, M, N) for M in Ms]
        ys = [c.add(y, r) for y, r in zip(ys, rs)]
    c.alloc()
    xv = rng.standard_normal((N, K), dtype=np.float32)
    be.tensor_set(x, xv.ravel())
    wv = [(rng.standard_normal((M, K), dtype=np.float32) * 0.05).astype(np.float16) for M in Ms]
    for w, v in zip(ws, wv): be.tensor_set(w, v.ravel())
    rv = [rng.standard_normal((N, M), dtype=np.float32) for M in Ms] if resid else []
    for r, v in zip(rs, rv): be.tensor_set(r, v.ravel())
    g = c.graph()
    be.graph_compute(g); be.synchronize()
    got1 = [be.tensor_get(y).copy().reshape(N, -1) for y in ys]
    be.graph_compute(g); be.synchronize()
    got2 = [be.tensor_get(y).copy().reshape(N, -1) for y in ys]
    xh = xv.astype(np.float16).astype(np.float64)
    for i, M in enumerate(Ms):
        ref = xh @ wv[i].astype(np.float64).T + (rv[i] if resid else 0.0)
        err = float(np.abs(got1[i] - ref).max() / np.abs(ref).max())
        same = bool((got1[i] == got2[i]).all())
        worst = max(worst, err)
        print(f"M={M} K={K} N={N} resid={resid}: max rel err {err:.2e}  repeat identical {same}", flush=True)
        assert err < 2e-5 and same
print("stream-K launches:", be.get_stat("gemm_sk_launches"), " worst", worst)
