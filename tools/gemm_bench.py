#!/usr/bin/env python3
"""tools/gemm_bench.py -- prefill-path micro-benchmarks through the backend C-ABI: the MFMA GEMM (F16 weights x f32 activations,
includes the activation f32->f16 conversion only once per graph) and the MFMA flash-attention, timed with HIP events around a
cgraph of REP identical nodes.  Prints TFLOP/s per shape."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402


def main():
    pkg = load_pkg()
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
    be = pkg.backend(0)
    be.set_option("graphs", 0)
    rng = np.random.default_rng(0)
    REP = 10

    def time_graph(c, flops):
        g = c.graph()
        for _ in range(2):
            be.graph_compute(g)
        be.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = be.timed_event(), be.timed_event()
            be.record(a); be.graph_compute(g); be.record(b)
            best = min(best, be.elapsed_ms(a, b))
        return best * 1e3 / REP, flops / (best * 1e-3 / REP) / 1e12

    shapes = [(4096, 4096, 512), (1024, 4096, 512), (12288, 4096, 512), (4096, 12288, 512), (4096, 4096, 2048), (12288, 4096, 2048),
              (4096, 12288, 2048), (4096, 4096, 4096), (8192, 8192, 8192)]
    if len(sys.argv) > 1 and sys.argv[1] == "--quick":
        shapes = shapes[:4]
    if "--attn-only" in sys.argv:
        shapes = []
    for (M, K, N) in shapes:
        c = Context(be)
        w = c.new_tensor(GGML_TYPE_F16, K, M)
        x = c.new_tensor(GGML_TYPE_F32, K, N)
        ys = [c.mul_mat(w, x) for _ in range(REP)]
        c.alloc()
        be.tensor_set(w, (rng.standard_normal(M * K, dtype=np.float32) * 0.05).astype(np.float16))
        be.tensor_set(x, rng.standard_normal(K * N, dtype=np.float32))
        us, tf = time_graph(c, 2.0 * M * K * N)
        print(f"gemm  M={M:6d} K={K:6d} N={N:5d}   {us:9.1f} us   {tf:7.1f} TFLOP/s", flush=True)
        c.free()

    for (nq, nkv, nh, nhkv, ns) in [(512, 512, 32, 8, 1), (512, 2048, 32, 8, 1), (2048, 2048, 32, 8, 1), (512, 2048, 32, 8, 8)]:
        D = 128
        c = Context(be)
        q = c.new_tensor(GGML_TYPE_F32, D, nq, nh, ns)
        k = c.new_tensor(GGML_TYPE_F16, D, nkv, nhkv, ns)
        v = c.new_tensor(GGML_TYPE_F16, D, nkv, nhkv, ns)
        m = c.new_tensor(GGML_TYPE_F16, nkv, nq)
        ys = [c.flash_attn_ext(q, k, v, m, 1.0 / np.sqrt(D)) for _ in range(REP)]
        c.alloc()
        be.tensor_set(q, rng.standard_normal(q.nelements(), dtype=np.float32))
        be.tensor_set(k, rng.standard_normal(k.nelements(), dtype=np.float32).astype(np.float16))
        be.tensor_set(v, rng.standard_normal(v.nelements(), dtype=np.float32).astype(np.float16))
        mask = np.zeros((nq, nkv), np.float16)
        off = nkv - nq
        for i in range(nq):
            mask[i, off + i + 1:] = -np.inf
        be.tensor_set(m, mask)
        live = float((mask == 0).sum())
        us, tf = time_graph(c, 4.0 * live * D * nh * ns)
        print(f"fattn nq={nq:5d} nkv={nkv:5d} nh={nh} ns={ns}   {us:9.1f} us   {tf:7.1f} TFLOP/s (unmasked cells only)", flush=True)
        c.free()


if __name__ == "__main__":
    main()
