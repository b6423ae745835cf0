"""oracle/oracle_py.py -- TEST INFRASTRUCTURE ONLY: numpy-friendly ctypes bindings of oracle/liboracle.so
(the C restatement in omni_oracle.c) plus a few small numpy helpers built on it."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

Q4_K, Q6_K, Q8_0, F16, F32 = 12, 14, 8, 1, 0
BLOCK = {Q4_K: (256, 144), Q6_K: (256, 210), Q8_0: (32, 34), F16: (1, 2), F32: (1, 4)}


def lib():
    global _LIB
    if _LIB is None:
        p = os.path.join(HERE, "liboracle.so")
        if not os.path.exists(p):
            raise RuntimeError(f"{p} missing: run `make -C oracle`")
        L = C.CDLL(p)
        L.orc_h2f.restype = C.c_float
        L.orc_h2f.argtypes = [C.c_uint16]
        L.orc_f2h.restype = C.c_uint16
        L.orc_f2h.argtypes = [C.c_float]
        for f in ("orc_vec_dot_q4_K_q8_K", "orc_vec_dot_q6_K_q8_K", "orc_vec_dot_q8_0_q8_0", "orc_vec_dot_f16"):
            getattr(L, f).restype = C.c_float
            getattr(L, f).argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        L.orc_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
        L.orc_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
        L.orc_rope_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_float] * 6 + [C.c_void_p]
        L.orc_soft_max_row.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float]
        L.orc_swiglu.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]
        L.orc_flash_attn_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_void_p]
        for f in ("orc_dequantize_row_q4_K", "orc_dequantize_row_q6_K", "orc_dequantize_row_q8_0", "orc_quantize_row_q8_K", "orc_quantize_row_q8_0",
                  "orc_f32_to_f16_row", "orc_f16_to_f32_row"):
            getattr(L, f).argtypes = [C.c_void_p, C.c_void_p, C.c_int64]
        _LIB = L
    return _LIB


def _p(a):
    return a.ctypes.data


def dequantize(ty, blocks, k):
    """blocks: uint8 array holding k/blck blocks -> float32[k]"""
    blocks = np.ascontiguousarray(blocks, dtype=np.uint8)
    y = np.empty(k, np.float32)
    {Q4_K: lib().orc_dequantize_row_q4_K, Q6_K: lib().orc_dequantize_row_q6_K, Q8_0: lib().orc_dequantize_row_q8_0}[ty](_p(blocks), _p(y), k)
    return y


def quantize_q8_K(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros(x.size // 256 * 292, np.uint8)
    lib().orc_quantize_row_q8_K(_p(x), _p(y), x.size)
    return y


def quantize_q8_0(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.zeros(x.size // 32 * 34, np.uint8)
    lib().orc_quantize_row_q8_0(_p(x), _p(y), x.size)
    return y


def f32_to_f16(x):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty(x.size, np.uint16)
    lib().orc_f32_to_f16_row(_p(x), _p(y), x.size)
    return y.reshape(x.shape)


def q8k_image(x):
    """The device-side activation layout (csrc/common.hpp): [qs K][bsums K/16 i16][d K/256 f32][pad16], from block_q8_K."""
    b = quantize_q8_K(x).reshape(-1, 292)
    qs = b[:, 4:260].reshape(-1)
    bs = b[:, 260:292].reshape(-1)
    d = b[:, 0:4].reshape(-1)
    img = np.concatenate([qs, bs, d])
    pad = (-img.size) % 16
    return np.concatenate([img, np.zeros(pad, np.uint8)])


def vec_dot(ty, k, w_row, act_q):
    f = {Q4_K: lib().orc_vec_dot_q4_K_q8_K, Q6_K: lib().orc_vec_dot_q6_K_q8_K, Q8_0: lib().orc_vec_dot_q8_0_q8_0, F16: lib().orc_vec_dot_f16}[ty]
    w_row = np.ascontiguousarray(w_row)
    act_q = np.ascontiguousarray(act_q)
    return float(f(k, _p(w_row), _p(act_q)))


def mul_mat(ty, W, X):
    """W: uint8 [M, row_bytes] (or f16/f32 viewed as bytes), X: float32 [N, K] -> float32 [N, M] (dst column-major like ggml)."""
    W = np.ascontiguousarray(W)
    X = np.ascontiguousarray(X, np.float32)
    N, K = X.shape
    M = W.shape[0]
    out = np.empty((N, M), np.float32)
    lib().orc_mul_mat(ty, _p(W), W.strides[0], _p(X), K, M, N, _p(out))
    return out


def rms_norm(x, eps):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    for r in range(x.reshape(-1, x.shape[-1]).shape[0]):
        xr = x.reshape(-1, x.shape[-1])[r]
        yr = y.reshape(-1, x.shape[-1])[r]
        lib().orc_rms_norm(_p(xr), _p(yr), xr.size, eps)
    return y


def rope(x, pos, n_dims, mode, n_ctx_orig=4096, freq_base=10000.0, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0, ff=None):
    """x: [n_tok, n_head, ne0] f32, pos: [n_tok]"""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    ffp = _p(np.ascontiguousarray(ff, np.float32)) if ff is not None else None
    for t in range(x.shape[0]):
        for h in range(x.shape[1]):
            lib().orc_rope_row(_p(x[t, h]), _p(y[t, h]), x.shape[2], n_dims, mode, int(pos[t]), n_ctx_orig, freq_base, freq_scale, ext_factor,
                               attn_factor, beta_fast, beta_slow, ffp)
    return y


def soft_max(x, mask, scale):
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    x2, y2 = x.reshape(-1, x.shape[-1]), y.reshape(-1, x.shape[-1])
    for r in range(x2.shape[0]):
        m = None if mask is None else np.ascontiguousarray(mask.reshape(-1, x.shape[-1])[r % mask.reshape(-1, x.shape[-1]).shape[0]], np.float32)
        lib().orc_soft_max_row(_p(x2[r]), _p(m) if m is not None else None, _p(y2[r]), x2.shape[1], scale)
    return y


def swiglu(x, g):
    x = np.ascontiguousarray(x, np.float32)
    g = np.ascontiguousarray(g, np.float32)
    y = np.empty_like(x)
    lib().orc_swiglu(_p(x), _p(g), _p(y), x.size)
    return y


def flash_attn_row(q, K16, V16, mask16, scale):
    """q f32[D]; K16,V16 uint16 [nkv, D]; mask16 uint16[nkv] or None"""
    q = np.ascontiguousarray(q, np.float32)
    K16 = np.ascontiguousarray(K16)
    V16 = np.ascontiguousarray(V16)
    out = np.empty(q.size, np.float32)
    m = np.ascontiguousarray(mask16) if mask16 is not None else None
    lib().orc_flash_attn_row(_p(q), _p(K16), K16.strides[0], _p(V16), V16.strides[0], _p(m) if m is not None else None, K16.shape[0], q.size, scale, _p(out))
    return out


def check_mul_mat_q4k(be, pkg, M=64, K=1024, N=2, seed=5):
    """Run one Q4_K MUL_MAT graph on backend `be` and compare with the C restatement (used by smoke() when oracle/_ref is absent)."""
    from llama_cpp_omni_amd import qwen3
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F32, GGML_TYPE_Q4_K, Context
    rng = np.random.default_rng(seed)
    c = Context(be)
    w = c.new_tensor(GGML_TYPE_Q4_K, K, M)
    x = c.new_tensor(GGML_TYPE_F32, K, N)
    y = c.mul_mat(w, x)
    c.alloc()
    wv = qwen3.random_blocks(rng, GGML_TYPE_Q4_K, M, K, std=0.05)
    xv = rng.standard_normal((N, K)).astype(np.float32)
    be.tensor_set(w, wv)
    be.tensor_set(x, xv)
    be.graph_compute(c.graph())
    got = be.tensor_get(y).reshape(N, M)
    ref = mul_mat(Q4_K, wv, xv)
    err = float(np.abs(got - ref).max() / max(1e-30, np.abs(ref).max()))
    assert err < 1e-5, f"Q4_K mul_mat vs C oracle: rel err {err}"
    c.free()
    return err
