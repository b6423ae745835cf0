#!/usr/bin/env python3
"""oracle/make_golden.py -- TEST INFRASTRUCTURE.  Generates the committed fixtures under tests/golden/ by running the
REAL reference (oracle/_ref/libggml-ref.so, compiled from /root/reference by oracle/Makefile.ref) in this container.

  quant.npz      block bytes / de-quantised values / activation quantisation / vec_dot scalars from the reference's
                 own functions on the reference's own test signal 0.1 + 2*cos(i + off) (tests/test-quantize-fns.cpp:31-35)
  ops.npz        per-op input/output captured by executing single-op graphs on the reference CPU backend
  tiny_model.npz a 2-layer Qwen3-shaped Q4_K_M model: weights, greedy token ids for 32 steps and final logits from the
                 reference CPU backend

Fixtures are data only (inputs + expected outputs).  Run:  python oracle/make_golden.py
"""
import ctypes as C
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def load_pkg():
    name = "llama_cpp_omni_amd"
    d = os.path.join(ROOT, "llama.cpp-omni_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


def ref_signal(n, off):
    i = np.arange(n, dtype=np.float32)
    return (0.1 + 2.0 * np.cos(i + np.float32(off))).astype(np.float32)


def gen_quant(ref):
    out = {}
    ref.ggml_quantize_chunk.restype = C.c_size_t
    ref.ggml_quantize_chunk.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
    ref.ggml_quantize_init.argtypes = [C.c_int]
    types = {"q4_K": (12, 256, 144, ref.dequantize_row_q4_K), "q6_K": (14, 256, 210, ref.dequantize_row_q6_K), "q8_0": (8, 32, 34, ref.dequantize_row_q8_0)}
    vdot = {"q4_K": ref.ggml_vec_dot_q4_K_q8_K, "q6_K": ref.ggml_vec_dot_q6_K_q8_K, "q8_0": ref.ggml_vec_dot_q8_0_q8_0}
    K = 12288
    x = ref_signal(K, 0.0)
    y = ref_signal(K, 1.0)
    out["x"], out["y"] = x, y
    # activation quantisers exactly as the x86 CPU backend runs them
    q8k = np.zeros(K // 256 * 292, np.uint8)
    ref.quantize_row_q8_K(y.ctypes.data_as(C.c_void_p), q8k.ctypes.data_as(C.c_void_p), C.c_int64(K))
    q80 = np.zeros(K // 32 * 34, np.uint8)
    ref.quantize_row_q8_0(y.ctypes.data_as(C.c_void_p), q80.ctypes.data_as(C.c_void_p), C.c_int64(K))
    out["y_q8_K"], out["y_q8_0"] = q8k, q80
    # edge rows for the Q8_K tie / zero rules
    edge = np.zeros(1024, np.float32)
    edge[256:512] = ref_signal(256, 3.0)
    edge[256 + 7] = 5.0
    edge[256 + 100] = -5.0            # |x| tie: the first (positive) one defines the sign of iscale
    edge[512:768] = -edge[256:512]
    edge[768:1024] = np.linspace(-1, 1, 256, dtype=np.float32) * 1e-30
    e8k = np.zeros(4 * 292, np.uint8)
    ref.quantize_row_q8_K(edge.ctypes.data_as(C.c_void_p), e8k.ctypes.data_as(C.c_void_p), C.c_int64(1024))
    e8k.reshape(4, 292)[0, 260:] = 0   # all-zero block: the reference leaves bsums untouched (stale); they are multiplied by d = 0
    out["edge"], out["edge_q8_K"] = edge, e8k
    for name, (ty, blck, bsz, deq) in types.items():
        ref.ggml_quantize_init(ty)
        blocks = np.zeros(K // blck * bsz, np.uint8)
        ref.ggml_quantize_chunk(ty, x.ctypes.data, blocks.ctypes.data, 0, 1, K, None)
        d = np.zeros(K, np.float32)
        deq.argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        deq(blocks.ctypes.data, d.ctypes.data, K)
        out[f"{name}_blocks"], out[f"{name}_deq"] = blocks, d
        f = vdot[name]
        f.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_int]
        act = q80 if name == "q8_0" else q8k
        dots = []
        for k in (256, 4096, 12288):
            s = C.c_float(0)
            f(k, C.byref(s), 0, blocks.ctypes.data, 0, act.ctypes.data, 0, 1)
            dots.append(s.value)
        out[f"{name}_dots"] = np.array(dots, np.float32)
    np.savez_compressed(os.path.join(OUT, "quant.npz"), **out)
    print("quant.npz:", {k: v.shape for k, v in out.items()})


def gen_ops(pkg, be):
    from llama_cpp_omni_amd import qwen3
    from llama_cpp_omni_amd.ggml import (GGML_ROPE_TYPE_NEOX, GGML_ROPE_TYPE_NORMAL, GGML_TYPE_F16, GGML_TYPE_F32, GGML_TYPE_I32, GGML_TYPE_I64,
                                         GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, GGML_TYPE_Q8_0, Context)
    rng = np.random.default_rng(42)
    out = {}

    def run(ctx, node, feeds):
        ctx.alloc()
        for t, v in feeds:
            be.tensor_set(t, v)
        be.graph_compute(ctx.graph())
        r = be.tensor_get(node).copy()
        ctx.free()
        return r

    # RMS_NORM + MUL (layer norm 4096-wide rows; q/k-norm 128-wide rows)
    for tag, (n, rows) in {"rms4096": (4096, 3), "rms128": (128, 40)}.items():
        c = Context(be)
        x = c.new_tensor(GGML_TYPE_F32, n, rows)
        w = c.new_tensor(GGML_TYPE_F32, n)
        y = c.mul(c.rms_norm(x, 1e-6), w)
        xv = (rng.standard_normal((rows, n)) * 3).astype(np.float32)
        wv = (1 + 0.1 * rng.standard_normal(n)).astype(np.float32)
        out[f"{tag}_x"], out[f"{tag}_w"] = xv, wv
        out[f"{tag}_y"] = run(c, y, [(x, xv), (w, wv)]).reshape(rows, n)
    # ROPE neox theta 1e6 at positions {0,1,2047}, and normal mode theta 1e4
    for tag, mode, base in (("rope_neox", GGML_ROPE_TYPE_NEOX, 1e6), ("rope_norm", GGML_ROPE_TYPE_NORMAL, 1e4)):
        c = Context(be)
        x = c.new_tensor(GGML_TYPE_F32, 128, 8, 3)
        p = c.new_tensor(GGML_TYPE_I32, 3)
        y = c.rope_ext(x, p, None, 128, mode, 40960, base, 1.0, 0.0, 1.0, 32.0, 1.0)
        xv = rng.standard_normal((3, 8, 128)).astype(np.float32)
        pv = np.array([0, 1, 2047], np.int32)
        out[f"{tag}_x"], out[f"{tag}_pos"] = xv, pv
        out[f"{tag}_y"] = run(c, y, [(x, xv), (p, pv)]).reshape(3, 8, 128)
    # SOFT_MAX with f16 mask
    c = Context(be)
    x = c.new_tensor(GGML_TYPE_F32, 256, 4, 8)
    m = c.new_tensor(GGML_TYPE_F16, 256, 64)
    y = c.soft_max_ext(x, m, 0.0883883, 0.0)
    xv = (rng.standard_normal((8, 4, 256)) * 4).astype(np.float32)
    mv = np.full((64, 256), -np.inf, np.float16)
    for i in range(4):
        mv[i, : 100 + i] = 0
    out["softmax_x"], out["softmax_mask"] = xv, mv
    out["softmax_y"] = run(c, y, [(x, xv), (m, mv)]).reshape(8, 4, 256)
    # SWIGLU (split)
    c = Context(be)
    a = c.new_tensor(GGML_TYPE_F32, 1024, 2)
    b = c.new_tensor(GGML_TYPE_F32, 1024, 2)
    y = c.swiglu_split(a, b)
    av, bv = (rng.standard_normal((2, 1024)) * 3).astype(np.float32), rng.standard_normal((2, 1024)).astype(np.float32)
    out["swiglu_a"], out["swiglu_b"] = av, bv
    out["swiglu_y"] = run(c, y, [(a, av), (b, bv)]).reshape(2, 1024)
    # SET_ROWS f32 -> f16 by i64 index
    c = Context(be)
    tab = c.new_tensor(GGML_TYPE_F16, 1024, 16)
    src = c.new_tensor(GGML_TYPE_F32, 1024, 3)
    idx = c.new_tensor(GGML_TYPE_I64, 3)
    y = c.set_rows(tab, src, idx)
    sv = (rng.standard_normal((3, 1024)) * 10).astype(np.float32)
    iv = np.array([5, 0, 15], np.int64)
    out["setrows_src"], out["setrows_idx"] = sv, iv
    out["setrows_tab"] = run(c, y, [(tab, np.zeros((16, 1024), np.float16)), (src, sv), (idx, iv)]).view(np.uint16).reshape(16, 1024)
    # FLASH_ATTN_EXT D=128, GQA 4, kv in {256, 2048}
    for nkv in (256, 2048):
        c = Context(be)
        q = c.new_tensor(GGML_TYPE_F32, 128, 2, 8)
        k = c.new_tensor(GGML_TYPE_F16, 128, nkv, 2)
        v = c.new_tensor(GGML_TYPE_F16, 128, nkv, 2)
        m = c.new_tensor(GGML_TYPE_F16, nkv, 64)
        y = c.flash_attn_ext(q, k, v, m, 1.0 / np.sqrt(128.0))
        qv = rng.standard_normal((8, 2, 128)).astype(np.float32)
        kv = rng.standard_normal((2, nkv, 128)).astype(np.float16)
        vv = rng.standard_normal((2, nkv, 128)).astype(np.float16)
        mv = np.full((64, nkv), -np.inf, np.float16)
        mv[0, : nkv - 37] = 0
        mv[1, : nkv - 36] = 0
        out[f"fa{nkv}_q"], out[f"fa{nkv}_k"], out[f"fa{nkv}_v"], out[f"fa{nkv}_mask"] = qv, kv.view(np.uint16), vv.view(np.uint16), mv.view(np.uint16)
        out[f"fa{nkv}_y"] = run(c, y, [(q, qv), (k, kv), (v, vv), (m, mv)]).reshape(2, 8, 128)
    # MUL_MAT m in {16..64} x k, n in {1, 3} for every weight type on the path
    for name, ty in (("q4_K", GGML_TYPE_Q4_K), ("q6_K", GGML_TYPE_Q6_K), ("q8_0", GGML_TYPE_Q8_0), ("f16", GGML_TYPE_F16)):
        for (M, K, N) in ((16, 256, 1), (64, 1024, 3)):
            c = Context(be)
            w = c.new_tensor(ty, K, M)
            x = c.new_tensor(GGML_TYPE_F32, K, N)
            y = c.mul_mat(w, x)
            wv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
            xv = rng.standard_normal((N, K)).astype(np.float32)
            tag = f"mm_{name}_{M}x{K}x{N}"
            out[tag + "_w"], out[tag + "_x"] = wv, xv
            out[tag + "_y"] = run(c, y, [(w, wv), (x, xv)]).reshape(N, M)
    np.savez_compressed(os.path.join(OUT, "ops.npz"), **out)
    print("ops.npz:", len(out), "arrays")


def gen_tiny_model(pkg, be):
    from llama_cpp_omni_amd import qwen3
    cfg = qwen3.TINY
    best = None
    for seed in range(11, 19):                                   # pick the seed with the widest top-2 logit gap
        mdl = qwen3.Model(be, cfg, qwen3.q4_k_m_types(cfg), n_ctx=256, seed=seed, flash_attn=True, host_copy=True)
        rng = np.random.default_rng(seed)
        table = (rng.standard_normal((cfg["n_vocab"], cfg["n_embd"]))).astype(np.float16)
        g, I, logits = mdl.build(1, 256)
        gr = g.graph()
        tok, toks, gap = 1, [], 1e9
        for step in range(32):
            mdl.set_inputs(I, table[tok].astype(np.float32)[None, :], step, 256)
            be.graph_compute(gr)
            l = be.tensor_get(logits).copy()
            s = np.sort(l)
            gap = min(gap, float(s[-1] - s[-2]))
            tok = int(np.argmax(l))
            toks.append(tok)
        if best is None or gap > best[0]:
            best = (gap, seed, dict(mdl.host), table, toks, l)
        g.free()
        mdl.wctx.free()
    gap, seed, host, table, toks, l = best
    out = {"seed": np.array(seed), "min_top2_gap": np.array(gap), "table": table.view(np.uint16), "tokens": np.array(toks, np.int32), "final_logits": l}
    for (il, name), v in host.items():
        out[f"w_{il}_{name}"] = v
    np.savez_compressed(os.path.join(OUT, "tiny_model.npz"), **out)
    print("tiny_model.npz: seed", seed, "min top-2 gap", gap, "tokens", toks[:8], "...")


if __name__ == "__main__":
    from oracle.ref_backend import make_ref_cpu_backend, ref_lib
    os.makedirs(OUT, exist_ok=True)
    pkg = load_pkg()
    gen_quant(ref_lib())
    be = make_ref_cpu_backend(pkg, 8)
    gen_ops(pkg, be)
    gen_tiny_model(pkg, be)
