#!/usr/bin/env python3
"""tools/pp_small.py [n_tokens ...] -- prefill of short prompts (one ubatch of N tokens, Qwen3-8B Q4_K_M) through the C-ABI: tok/s and the per-class profile"""
import os, sys
os.environ["MI355X_BENCH_PROFILE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = bench.load_pkg()
from llama_cpp_omni_amd import qwen3
be = pkg.backend(0)
cfg = qwen3.QWEN3_8B
model = qwen3.Model(be, cfg, qwen3.q4_k_m_types(cfg), n_ctx=1024, seed=1, share_layer_bytes=True, flash_attn=True)
for n in [int(a) for a in sys.argv[1:]] or [33, 64, 100, 128, 256]:
    tps, ok = bench.prefill_tok_s(pkg, be, model, n_tokens=n)
    print(f"pp{n}: {tps:9.1f} tok/s  ({n / tps * 1e3:.2f} ms)  ok={ok}", flush=True)
