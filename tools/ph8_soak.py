#!/usr/bin/env python3
"""tools/ph8_soak.py -- race screen of the eight-phase GEMM: the same product many times, every result compared bit for bit with the first
(the schedule's RAW / WAR distances are counted, not timed -- a wrong count shows up as rare differing tiles under load)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
be = pkg.backend(0); be.set_option("graphs", 0)
rng = np.random.default_rng(2)
REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for (M, N, K, glu) in [(24576, 2048, 4096, False), (8190, 4090, 320, False), (4096, 16384, 4096, False), (12288, 2048, 4096, True)]:
    c = Context(be)
    x = c.new_tensor(GGML_TYPE_F32, K, N)
    if glu:
        wg = c.new_tensor(GGML_TYPE_F16, K, M); wu = c.new_tensor(GGML_TYPE_F16, K, M); wd = c.new_tensor(GGML_TYPE_F16, M, 256)
        y = c.mul_mat(wd, c.swiglu_split(c.mul_mat(wg, x), c.mul_mat(wu, x)))
        ws = [wg, wu, wd]
    else:
        w = c.new_tensor(GGML_TYPE_F16, K, M); y = c.mul_mat(w, x); ws = [w]
    c.alloc()
    for t in ws:
        be.tensor_set(t, (rng.standard_normal(t.nelements()) * 0.05).astype(np.float16))
    be.tensor_set(x, rng.standard_normal(K * N).astype(np.float32))
    g = c.graph()
    b0 = be.get_stat("gemm256_launches") + be.get_stat("gemm_glu_launches")
    be.graph_compute(g); first = be.tensor_get(y).copy()
    bad = 0
    for r in range(REPS):
        for _ in range(4):
            be.graph_compute(g)
        if not np.array_equal(first.view(np.uint32), be.tensor_get(y).view(np.uint32)): bad += 1
    n = be.get_stat("gemm256_launches") + be.get_stat("gemm_glu_launches") - b0
    print(f"M={M} N={N} K={K} glu={glu}: {int(n)} eight-phase launches, {REPS} checks, differing results: {bad}", flush=True)
    c.free()
