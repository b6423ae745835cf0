import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0); be.set_option("graphs", 0)
rng = np.random.default_rng(1)
for (M, N, K) in [(4096, 4096, 1088), (4000, 4200, 192), (4096, 4096, 64), (8192, 2048, 4096)]:
    wv = (rng.standard_normal((M, K)) * 0.05).astype(np.float16); xv = rng.standard_normal((N, K)).astype(np.float32)
    c = Context(be)
    w = c.new_tensor(GGML_TYPE_F16, K, M); x = c.new_tensor(GGML_TYPE_F32, K, N)
    y = c.mul_mat(w, x)
    c.alloc()
    be.tensor_set(w, wv); be.tensor_set(x, xv)
    b0 = be.get_stat("gemm256_launches")
    outs = []
    for it in range(6):
        be.graph_compute(c.graph()); outs.append(be.tensor_get(y).copy().reshape(N, M))
    sel = be.get_stat("gemm256_launches") - b0
    want = xv.astype(np.float16).astype(np.float32) @ wv.astype(np.float32).T
    err = np.abs(outs[0] - want).max(); den = np.abs(want).max()
    same = all(np.array_equal(outs[0], o) for o in outs[1:])
    print(f"M={M} N={N} K={K}: 256-tile launches {sel}, max abs err {err:.3e} (max |want| {den:.2f}), 6 runs identical: {same}", flush=True)
    c.free()
