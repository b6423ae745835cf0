#!/usr/bin/env python3
"""tools/omni_prof.py -- per-class launch counts / times (HIP events around every launch) of the omni encoder and Token2Wav graphs at their real shapes"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import load_pkg
pkg = load_pkg()
from llama_cpp_omni_amd import encoders as E, token2wav as T
be = pkg.backend(0); be.set_option("graphs", 0)
rng = np.random.default_rng(1)
CLS = ("gemm_f16", "gemm_any_f16", "gemm_any_f32", "gemm_reduce", "act_convert", "rms_norm_mul", "rms_norm", "norm", "norm_rope", "rope", "fattn", "set_rows", "get_rows", "bin", "glu", "cpy", "soft_max", "dequant_f16",
       "mmv_f16", "mmv_f32", "mmv_q80", "mmv_q4k", "mmv_q6k", "unary", "scale", "im2col", "pool", "math", "concat", "repeat", "pad", "pad_reflect", "sum_rows", "conv_transpose_1d", "timestep_embedding", "arange", "empty")


def flat(W):
    out = [v for k, v in W.items() if k != "layers" and hasattr(v, "nelements")]
    for L in W.get("layers", []):
        out += list(L.values())
    return out


def run(name, c, tensors):
    c.alloc()
    for t in tensors:
        n = t.nelements(); v = (rng.standard_normal(n) * 0.05).astype(np.float32)
        be.tensor_set(t, v.astype(np.float16) if t.type == 1 else ((np.abs(v) + 0.5 if n <= 4096 else v) if t.type == 0 else np.zeros(n, np.int32)))
    g = c.graph()
    be.graph_compute(g); be.synchronize()
    be.set_option("profile", 1); be.set_option("reset_stats", 1)
    be.graph_compute(g); be.synchronize()
    tot = 0.0
    print("==", name, "kernels", int(be.get_stat("kernels_last_graph")))
    for cls in CLS:
        u, k = be.get_stat(f"prof_{cls}_us"), be.get_stat(f"prof_{cls}_n")
        if k > 0:
            print(f"   {cls:20s} n={int(k):5d}  total {u:10.1f} us  avg {u / k:8.2f}")
            tot += u
    print(f"   sum of classes {tot:.1f} us")
    be.set_option("profile", 0)
    c.free()


which = sys.argv[1:] or ["whisper", "siglip2", "dit", "hift"]
if "whisper" in which:
    c = pkg.Context(be); W = E.whisper_weights(c, E.WHISPER, 1); inp, _ = E.whisper(c, E.WHISPER, W, 3000); run("whisper", c, flat(W) + [inp])
if "siglip2" in which:
    c = pkg.Context(be); W = E.siglip2_weights(c, E.SIGLIP2, 1); inp, vit = E.siglip2(c, E.SIGLIP2, W)
    Wr = E.resampler_weights(c, E.RESAMPLER); pe, _ = E.resampler(c, E.RESAMPLER, Wr, vit, 1024); run("siglip2+resampler", c, flat(W) + flat(Wr) + [inp, pe])
if "dit" in which:
    c = pkg.Context(be); W = T.dit_weights(c, T.DIT); x, cond, _ = T.dit_block(c, T.DIT, W, 200); run("dit_block", c, flat(W) + [x, cond])
if "hift" in which:
    c = pkg.Context(be); W = T.hift_weights(c, T.HIFT); x, _ = T.hift_upsample_stage(c, T.HIFT, W, 120); run("hift_stage", c, flat(W) + [x])
