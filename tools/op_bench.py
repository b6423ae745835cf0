#!/usr/bin/env python3
"""tools/op_bench.py -- per-node cost of the small (non-matmul) decode ops in hipGraph-replay and eager mode.
A chain of N identical nodes is submitted as one cgraph through the backend C-ABI and timed with HIP events."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402


def main():
    pkg = load_pkg()
    from llama_cpp_omni_amd.ggml import GGML_ROPE_TYPE_NEOX, GGML_TYPE_F16, GGML_TYPE_F32, GGML_TYPE_I32, GGML_TYPE_I64, Context
    be = pkg.backend(0)
    N = 400

    def chain(name, build):
        for graphs in (1, 0):
            be.set_option("graphs", graphs)
            c = Context(be)
            build(c)
            c.alloc()
            for t in c.tensors:
                if t.t.view_src:
                    continue
                if t.type == GGML_TYPE_I32 or t.type == GGML_TYPE_I64:
                    be.tensor_set(t, np.zeros(t.nelements(), np.int32 if t.type == GGML_TYPE_I32 else np.int64))
                elif t.type == GGML_TYPE_F32:
                    be.tensor_set(t, np.ones(t.nelements(), np.float32))
                elif t.type == GGML_TYPE_F16 and t.ne[1] == 64:          # KQ mask: 72 live cells (tg128 mid-run depth), rest -inf
                    m = np.full((64, t.ne[0]), -np.inf, np.float16)
                    m[:, :72] = 0
                    be.tensor_set(t, m)
                elif t.type == GGML_TYPE_F16:
                    be.tensor_set(t, (np.random.default_rng(0).standard_normal(t.nelements()) * 0.5).astype(np.float16))
            g = c.graph()
            for _ in range(3):
                be.graph_compute(g)
            be.synchronize()
            best = 1e9
            for _ in range(5):
                a, b = be.timed_event(), be.timed_event()
                be.record(a); be.graph_compute(g); be.record(b)
                best = min(best, be.elapsed_ms(a, b))
            n_k = be.get_stat("kernels_last_graph")
            print(f"{name:28s} graphs={graphs}  {best * 1e3 / n_k:7.2f} us/kernel  ({int(n_k)} kernels)", flush=True)
            c.free()
        be.set_option("graphs", 1)

    def b_add(c):
        x = c.new_tensor(GGML_TYPE_F32, 4096)
        y = c.new_tensor(GGML_TYPE_F32, 4096)
        for _ in range(N):
            x = c.add(x, y)

    def b_scale(c):
        x = c.new_tensor(GGML_TYPE_F32, 4096)
        for _ in range(N):
            x = c.scale(x, 1.0001)

    def b_rms(c):
        x = c.new_tensor(GGML_TYPE_F32, 4096)
        w = c.new_tensor(GGML_TYPE_F32, 4096)
        for _ in range(N):
            x = c.mul(c.rms_norm(x, 1e-6), w)

    def b_rope(c):
        x = c.new_tensor(GGML_TYPE_F32, 128, 32, 1)
        p = c.new_tensor(GGML_TYPE_I32, 1)
        for _ in range(N):
            x = c.rope_ext(x, p, None, 128, GGML_ROPE_TYPE_NEOX, 40960, 1e6, 1.0, 0.0, 1.0, 32.0, 1.0)

    def b_glu(c):
        a = c.new_tensor(GGML_TYPE_F32, 12288)
        b = c.new_tensor(GGML_TYPE_F32, 12288)
        for _ in range(N):
            a = c.swiglu_split(a, b)

    def b_setrows(c):
        tab = c.new_tensor(GGML_TYPE_F16, 1024, 256)
        src = c.new_tensor(GGML_TYPE_F32, 1024, 1)
        idx = c.new_tensor(GGML_TYPE_I64, 1)
        for _ in range(N):
            c.set_rows(tab, src, idx)

    def b_fattn(c):
        q = c.new_tensor(GGML_TYPE_F32, 128, 1, 32)
        k = c.new_tensor(GGML_TYPE_F16, 128, 256, 8)
        v = c.new_tensor(GGML_TYPE_F16, 128, 256, 8)
        m = c.new_tensor(GGML_TYPE_F16, 256, 64)
        for _ in range(100):
            c.flash_attn_ext(q, k, v, m, 0.088)

    for name, fn in (("add 4096", b_add), ("scale 4096", b_scale), ("rms_norm+mul 4096", b_rms), ("rope 128x32", b_rope), ("swiglu 12288", b_glu),
                     ("set_rows 1024 f16", b_setrows), ("fattn decode nkv=256", b_fattn)):
        chain(name, fn)


if __name__ == "__main__":
    main()
