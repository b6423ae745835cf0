#!/usr/bin/env python3
"""tools/mmv_bench.py -- kernel-level microbenchmark of the weight-streaming mat-vec kernels through the backend
C-ABI.  One cgraph holds N MUL_MAT nodes over N *distinct* weight tensors (N x bytes > 512 MiB, so the 256 MiB
Infinity Cache cannot serve re-reads) that share one activation vector; the graph is replayed a few times and timed
with HIP events on the backend's stream.  Reports us per mat-vec and algorithmic GB/s (weight bytes / time).

usage: python tools/mmv_bench.py [--types q4_K,q6_K] [--shapes 4096x4096,...] [--reps 5] [--ncols 1]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg  # noqa: E402

TYPES = {"q4_K": 12, "q6_K": 14, "q8_0": 8, "f16": 1, "q4_0": 2, "q5_K": 13}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--types", default="q4_K,q6_K")
    ap.add_argument("--shapes", default="1024x4096,4096x4096,12288x4096,4096x12288,151936x4096")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--ncols", type=int, default=1)
    ap.add_argument("--min-mib", type=int, default=768)
    ap.add_argument("--pair", action="store_true")
    args = ap.parse_args()
    pkg = load_pkg()
    from llama_cpp_omni_amd import qwen3
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F32, Context, row_size
    be = pkg.backend(0)
    rng = np.random.default_rng(0)
    for tname in args.types.split(","):
        ty = TYPES[tname]
        for shp in args.shapes.split(","):
            M, K = (int(v) for v in shp.split("x"))
            wbytes = M * row_size(ty, K)
            n = max(2, min(512, (args.min_mib << 20) // wbytes + 1))
            c = Context(be)
            x = c.new_tensor(GGML_TYPE_F32, K, args.ncols)
            ws = [c.new_tensor(ty, K, M) for _ in range(n)]
            if args.pair:                                            # ffn_gate + ffn_up + SWIGLU launches (k_mmv_pair)
                n -= n % 2
                ws = ws[:n]
                ys = []
                for i in range(0, n, 2):
                    up, gate = c.mul_mat(ws[i], x), c.mul_mat(ws[i + 1], x)
                    ys.append(c.swiglu_split(gate, up))
            else:
                ys = [c.mul_mat(w, x) for w in ws]
            c.alloc()
            host = qwen3.random_blocks(rng, ty, min(M, 4096), K)
            reps_rows = (M + host.shape[0] - 1) // host.shape[0]
            full = np.tile(host, (reps_rows, 1))[:M]
            for w in ws:
                be.tensor_set(w, full)
            be.tensor_set(x, rng.standard_normal((args.ncols, K)).astype(np.float32))
            g = c.graph()
            for _ in range(3):
                be.graph_compute(g)           # eager, capture, first replay
            be.synchronize()
            best = 1e30
            for _ in range(args.reps):
                a, b = be.timed_event(), be.timed_event()
                be.record(a)
                be.graph_compute(g)
                be.record(b)
                ms = be.elapsed_ms(a, b)
                best = min(best, ms)
            us = best * 1e3 / n
            print(f"{tname:5s} {M:6d}x{K:<6d} ncols={args.ncols} n={n:3d}  {us:9.2f} us/matvec  {wbytes / us / 1e3:8.1f} GB/s  ({wbytes / 1e6:.1f} MB)", flush=True)
            c.free()


if __name__ == "__main__":
    main()
