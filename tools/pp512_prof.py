#!/usr/bin/env python3
"""tools/pp512_prof.py [--no-fa] [--tokens N] [--reps R] [--f16] -- ONLY the prefill leg of bench.py (one ubatch of N tokens at depth 0, Qwen3-8B shapes, Q4_K_M or
all-F16 weights), for use under rocprofv3 --kernel-trace --stats: every kernel in the trace belongs to the prefill.  Prints tok/s and, with MI355X_BENCH_PROFILE=1,
the per-class event profile."""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

ap = argparse.ArgumentParser()
ap.add_argument("--no-fa", action="store_true"); ap.add_argument("--tokens", type=int, default=512); ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--f16", action="store_true"); ap.add_argument("--layers", type=int, default=0)
a = ap.parse_args()
pkg = bench.load_pkg()
from llama_cpp_omni_amd import qwen3
be = pkg.backend(0)
cfg = dict(qwen3.QWEN3_8B)
if a.layers: cfg["n_layer"] = a.layers
types = qwen3.uniform_types(cfg, pkg.GGML_TYPE_F16) if a.f16 else qwen3.q4_k_m_types(cfg)
n_ctx = max(512, (a.tokens + 255) // 256 * 256)
model = qwen3.Model(be, cfg, types, n_ctx=n_ctx, seed=1234, share_layer_bytes=True, flash_attn=not a.no_fa)
pp, ok = bench.prefill_tok_s(pkg, be, model, n_tokens=a.tokens, reps=a.reps)
print(json.dumps({"pp_tok_s": round(pp, 1), "tokens": a.tokens, "ms": round(a.tokens / pp * 1e3, 3), "flash_attn": not a.no_fa, "finite": ok, "layers": cfg["n_layer"]}))
