#!/usr/bin/env python3
"""tools/gemm_rf_check.py -- the register-ring GEMM (option "gemm_rf") against the LDS-DMA form on the pp512 shapes: same bits, us per node hot and cold."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0); be.set_option("graphs", 0)
rng = np.random.default_rng(0)
REP = 12
for (M, K, N) in [(4096, 4096, 512), (4096, 12288, 512), (6144, 4096, 512), (4096, 4096, 2048), (4096, 4096, 160), (1024, 4096, 512)]:
    res = {}
    for cold in (False, True):
        c = Context(be)
        ws = [c.new_tensor(GGML_TYPE_F16, K, M) for _ in range(REP if cold else 1)]; x = c.new_tensor(GGML_TYPE_F32, K, N)
        ys = [c.mul_mat(ws[i % len(ws)], x) for i in range(REP)]
        c.alloc()
        wv = (rng.standard_normal(M * K, dtype=np.float32) * 0.05).astype(np.float16)
        for t in ws: be.tensor_set(t, wv)
        be.tensor_set(x, rng.standard_normal(K * N, dtype=np.float32))
        g = c.graph()
        line = f"M={M} K={K} N={N} {'cold' if cold else 'hot '}:"
        for mode in (0, 4, 2):
            be.set_option("gemm_rf", mode)
            n0 = be.get_stat("gemm_rf_launches")
            for _ in range(2): be.graph_compute(g)
            be.synchronize(); best = 1e9
            for _ in range(5):
                a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); best = min(best, be.elapsed_ms(a, b))
            out = be.tensor_get(ys[-1]).copy()
            res[mode] = out
            line += f"   rf={mode}: {best * 1e3 / REP:6.1f} us ({int(be.get_stat('gemm_rf_launches') - n0)} rf launches)"
        same = all(np.array_equal(res[0].view(np.uint32), res[m].view(np.uint32)) for m in (2, 4))
        print(line + f"   bits {'identical' if same else 'DIFFER max ' + str(np.abs(res[0] - res[4]).max())}", flush=True)
        c.free()
be.set_option("gemm_rf", -1)
