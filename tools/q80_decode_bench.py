#!/usr/bin/env python3
"""tools/q80_decode_bench.py -- decode step of the all-Q8_0 Qwen3-8B (BASELINE configs[4]'s LLM) and of the Q8_0 TTS decoder through bench.Decoder: tok/s, three repeats.
A/B of the engine's waves per workgroup: MI355X_MV2_Q80_NW=16|12|10, MI355X_MV2_NW16=1."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pkg = bench.load_pkg()
from llama_cpp_omni_amd import qwen3
be = pkg.backend(0)
for name, cfg in (("qwen3_8b_q8_0", qwen3.QWEN3_8B), ("tts_q8_0", qwen3.TTS)):
    dec = bench.Decoder(pkg, be, cfg, qwen3.uniform_types(cfg, pkg.GGML_TYPE_Q8_0), n_ctx=256, n_kv=256, flash_attn=True, seed=4321)
    for p in range(8):
        dec.step(p)
    best = 0.0
    for r in range(3):
        be.synchronize(); t0 = time.perf_counter()
        for p in range(8, 72):
            dec.step(p)
        be.synchronize(); best = max(best, 64 / (time.perf_counter() - t0))
    print(f"{name}: {best:.1f} tok/s ({1e3 / best:.4f} ms per step)")
    dec.g.free(); dec.model.wctx.free()
