#!/usr/bin/env python3
"""tools/fattn_bench.py -- decode attention (one query token, Qwen3-8B heads: 32 q / 8 kv, D=128) at KV depth N: us per node and the
K+V bytes/s it streams.  Nodes rotate over 4 distinct K/V sets so the larger depths come from HBM, not from the 256 MB cache."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402


def main():
    pkg = load_pkg()
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F16, GGML_TYPE_F32, Context
    be = pkg.backend(0)
    nq = int(os.environ.get("NQ", "1"))
    for nkv in [int(x) for x in os.environ.get("NKV", "256,2048,8192,32768").split(",")]:
        c = Context(be)
        q = c.new_tensor(GGML_TYPE_F32, 128, nq, 32)
        m = c.new_tensor(GGML_TYPE_F16, nkv, 64)
        sets = [(c.new_tensor(GGML_TYPE_F16, 128, nkv, 8), c.new_tensor(GGML_TYPE_F16, 128, nkv, 8)) for _ in range(4)]
        nodes = 36
        for i in range(nodes):
            k, v = sets[i % 4]
            c.flash_attn_ext(q, k, v, m, 0.088)
        c.alloc()
        rng = np.random.default_rng(0)
        be.tensor_set(q, rng.standard_normal(q.nelements()).astype(np.float32))
        mk = np.zeros((64, nkv), np.float16)
        live = int(os.environ.get("LIVE", "0"))                       # 0: every cell live; N: the first N cells (a padded cache view)
        if live:
            mk[:, live:] = -np.inf
        be.tensor_set(m, mk)
        for k, v in sets:
            be.tensor_set(k, (rng.standard_normal(k.nelements()) * 0.5).astype(np.float16))
            be.tensor_set(v, (rng.standard_normal(v.nelements()) * 0.5).astype(np.float16))
        g = c.graph()
        for _ in range(3):
            be.graph_compute(g)
        be.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = be.timed_event(), be.timed_event()
            be.record(a); be.graph_compute(g); be.record(b)
            best = min(best, be.elapsed_ms(a, b))
        us = best * 1e3 / nodes
        byt = 2 * nkv * 8 * 128 * 2
        print(f"fattn decode nq={nq} nkv={nkv:6d}: {us:8.2f} us/node  {byt / us / 1e3:8.1f} GB/s  ({int(be.get_stat('kernels_last_graph'))} kernels)", flush=True)
        c.free()


if __name__ == "__main__":
    main()
