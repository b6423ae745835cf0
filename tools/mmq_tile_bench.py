#!/usr/bin/env python3
"""tools/mmq_tile_bench.py [M K N ...] -- Q4_K x f32 MUL_MAT nodes of a prefill ubatch through the C-ABI: the tiled int8-MFMA kernel (mmq_tile.hip, option mmq_tile = 1)
against the F16-image GEMM (mmq_tile = 0), us per node (HIP events around REP nodes, a weight tensor per node so the stream comes from HBM) and the NMSE of each
against the C oracle on a row subset.  Includes the activation conversion once per graph (quantise / f32 -> f16)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd.ggml import GGML_TYPE_F32, GGML_TYPE_Q4_K, Context
from llama_cpp_omni_amd import qwen3
from oracle import oracle_py as orc
be = pkg.backend(0); be.set_option("graphs", 0)
args = [int(v) for v in sys.argv[1:] if v.isdigit()]
shapes = [tuple(args[i:i + 3]) for i in range(0, len(args) - 2, 3)] or [(4096, 4096, 512), (6144, 4096, 512), (24576, 4096, 512), (4096, 12288, 512), (4096, 4096, 2048), (24576, 4096, 2048)]
REP = int(os.environ.get("REP", "8"))
rng = np.random.default_rng(0)
for (M, K, N) in shapes:
    wv = qwen3.random_blocks(rng, GGML_TYPE_Q4_K, M, K, std=0.05)
    xv = rng.standard_normal((N, K)).astype(np.float32)
    want = orc.mul_mat(GGML_TYPE_Q4_K, wv.view(np.uint8).reshape(M, -1)[:64], xv)
    line = f"M={M:6d} K={K:6d} N={N:5d}:"
    for mode in (1, 0):
        be.set_option("mmq_tile", mode)
        c = Context(be)
        ws = [c.new_tensor(GGML_TYPE_Q4_K, K, M) for _ in range(REP)]; x = c.new_tensor(GGML_TYPE_F32, K, N)
        ys = [c.mul_mat(w, x) for w in ws]
        c.alloc()
        for t in ws: be.tensor_set(t, wv)
        be.tensor_set(x, xv)
        g = c.graph()
        for _ in range(2): be.graph_compute(g)
        be.synchronize(); best = 1e9
        for _ in range(5):
            a, b = be.timed_event(), be.timed_event(); be.record(a); be.graph_compute(g); be.record(b); best = min(best, be.elapsed_ms(a, b))
        got = be.tensor_get(ys[-1]).reshape(N, M)
        err = float(((got[:, :64] - want) ** 2).sum() / (want ** 2).sum())
        us = best * 1e3 / REP
        line += f"   {'mmq_tile' if mode else 'f16 image'} {us:8.1f} us {2.0 * M * K * N / us / 1e6:7.1f} T(FL)OP/s nmse {err:.1e}"
        c.free()
    print(line, flush=True)
be.set_option("mmq_tile", -1)
