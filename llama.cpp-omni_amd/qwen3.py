"""Host-side mirror of the reference's Qwen3 graph (llm_build_qwen3, reference src/llama-model.cpp:9287-9406,
attention glue src/llama-graph.cpp:1303-1590, KV cache views src/llama-kv-cache.cpp:969-1110) built with the
ctypes graph builder, plus the synthetic-weight generator of BASELINE.md section 3 / SURVEY.md 8(d).

It produces exactly the node sequence the backend's graph_compute receives from libllama for one ubatch, so
the standalone bench (bench.py) and the parity tests exercise the same graphs llama-bench would submit.
No arithmetic is done here.
"""
import math

import numpy as np

from .ggml import (GGML_BACKEND_BUFFER_USAGE_WEIGHTS, GGML_ROPE_TYPE_NEOX, GGML_TYPE_F16, GGML_TYPE_F32, GGML_TYPE_I32, GGML_TYPE_I64, GGML_TYPE_Q4_K, GGML_TYPE_Q6_K,
                   GGML_TYPE_Q8_0, Context, row_size)

QWEN3_8B = dict(n_embd=4096, n_layer=36, n_head=32, n_head_kv=8, head_dim=128, n_ff=12288, n_vocab=151936,
                rms_eps=1e-6, rope_base=1e6, n_ctx_orig=40960)
TTS = dict(arch="llama", n_embd=768, n_layer=20, n_head=12, n_head_kv=12, head_dim=64, n_ff=3072, n_vocab=32000,
           rms_eps=1e-5, rope_base=1e4, n_ctx_orig=4096)           # the omni TTS decoder (reference tools/omni/convert/tts.txt; tools/make_synth_gguf.py --config tts)
TINY = dict(n_embd=256, n_layer=2, n_head=4, n_head_kv=2, head_dim=64, n_ff=512, n_vocab=512,
            rms_eps=1e-6, rope_base=1e6, n_ctx_orig=4096)


def use_more_bits(i, n):
    """llama-quant.cpp:185-187"""
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def q4_k_m_types(cfg):
    """Tensor-type map of a Q4_K_M file (llama-quant.cpp:224-226,302-303,358-364)."""
    n = cfg["n_layer"]
    t = {"output": GGML_TYPE_Q6_K}
    for i in range(n):
        hi = GGML_TYPE_Q6_K if use_more_bits(i, n) else GGML_TYPE_Q4_K
        t[i] = dict(attn_q=GGML_TYPE_Q4_K, attn_k=GGML_TYPE_Q4_K, attn_v=hi, attn_output=GGML_TYPE_Q4_K,
                    ffn_gate=GGML_TYPE_Q4_K, ffn_up=GGML_TYPE_Q4_K, ffn_down=hi)
    return t


def uniform_types(cfg, ty):
    n = cfg["n_layer"]
    t = {"output": ty}
    for i in range(n):
        t[i] = {k: ty for k in ("attn_q", "attn_k", "attn_v", "attn_output", "ffn_gate", "ffn_up", "ffn_down")}
    return t


# ------------------------------------------------------------------------------------------------ synthetic weights
def _f16_bits(x):
    return np.asarray(x, dtype=np.float16).view(np.uint16)


def random_blocks(rng, ty, nrows, K, std=0.02):
    """Random *valid* quantised rows emitted directly in block format (no f32 master copy).

    Scales are chosen so the de-quantised weights have roughly `std` standard deviation and zero mean,
    which keeps a 36-layer random network numerically tame.
    """
    if ty == GGML_TYPE_F32:
        return (rng.standard_normal((nrows, K), dtype=np.float32) * std).view(np.uint8).reshape(nrows, -1)
    if ty == GGML_TYPE_F16:
        return (rng.standard_normal((nrows, K), dtype=np.float32) * std).astype(np.float16).view(np.uint8).reshape(nrows, -1)
    if ty == GGML_TYPE_Q4_K:
        nb = K // 256
        blk = np.empty((nrows, nb, 144), dtype=np.uint8)
        # value = d*sc*q - dmin*m, sc,m in [0,63], q in [0,15]: E[sc*q] = 236.25, E[m] = 31.5, std(sc*q) ~ 200
        d = (std / 200.0) * rng.uniform(0.5, 1.5, size=(nrows, nb)).astype(np.float32)
        dmin = d * 7.5
        blk[..., 0:2] = _f16_bits(d)[..., None].view(np.uint8).reshape(nrows, nb, 2)
        blk[..., 2:4] = _f16_bits(dmin)[..., None].view(np.uint8).reshape(nrows, nb, 2)
        blk[..., 4:] = rng.integers(0, 256, size=(nrows, nb, 140), dtype=np.uint8)
        return blk.reshape(nrows, nb * 144)
    if ty == GGML_TYPE_Q6_K:
        nb = K // 256
        blk = np.empty((nrows, nb, 210), dtype=np.uint8)
        blk[..., :208] = rng.integers(0, 256, size=(nrows, nb, 208), dtype=np.uint8)        # ql, qh, int8 scales
        # value = d*sc*(q-32), sc in [-128,127], q-32 in [-32,31]: std ~ 74*18.5 = 1370
        d = (std / 1370.0) * rng.uniform(0.5, 1.5, size=(nrows, nb)).astype(np.float32)
        blk[..., 208:210] = _f16_bits(d)[..., None].view(np.uint8).reshape(nrows, nb, 2)
        return blk.reshape(nrows, nb * 210)
    if ty == GGML_TYPE_Q8_0:
        nb = K // 32
        blk = np.empty((nrows, nb, 34), dtype=np.uint8)
        d = (std / 73.0) * rng.uniform(0.5, 1.5, size=(nrows, nb)).astype(np.float32)
        blk[..., 0:2] = _f16_bits(d)[..., None].view(np.uint8).reshape(nrows, nb, 2)
        blk[..., 2:] = rng.integers(0, 256, size=(nrows, nb, 32), dtype=np.uint8)
        return blk.reshape(nrows, nb * 34)
    if ty == 2:                                              # Q4_0: value = (q - 8) * d, q in [0, 15]: std ~ 4.6 d
        nb = K // 32
        blk = np.empty((nrows, nb, 18), dtype=np.uint8)
        d = (std / 4.6) * rng.uniform(0.5, 1.5, size=(nrows, nb)).astype(np.float32)
        blk[..., 0:2] = _f16_bits(d)[..., None].view(np.uint8).reshape(nrows, nb, 2)
        blk[..., 2:] = rng.integers(0, 256, size=(nrows, nb, 16), dtype=np.uint8)
        return blk.reshape(nrows, nb * 18)
    if ty == 13:                                             # Q5_K: value = d*sc*q - dmin*m, q in [0, 31]: E[sc*q] = 488, std ~ 420
        nb = K // 256
        blk = np.empty((nrows, nb, 176), dtype=np.uint8)
        d = (std / 420.0) * rng.uniform(0.5, 1.5, size=(nrows, nb)).astype(np.float32)
        blk[..., 0:2] = _f16_bits(d)[..., None].view(np.uint8).reshape(nrows, nb, 2)
        blk[..., 2:4] = _f16_bits(d * 15.5)[..., None].view(np.uint8).reshape(nrows, nb, 2)
        blk[..., 4:] = rng.integers(0, 256, size=(nrows, nb, 172), dtype=np.uint8)
        return blk.reshape(nrows, nb * 176)
    raise ValueError(ty)


class Model:
    """Weights + KV cache resident in HBM, and graph builders for one ubatch."""

    def __init__(self, be, cfg, types, n_ctx=512, seed=1234, share_layer_bytes=False, flash_attn=True, host_copy=False, weights=None, v_trans=None):
        self.be, self.cfg, self.types, self.n_ctx, self.fa = be, cfg, types, n_ctx, flash_attn
        # llama_kv_cache: the V cache is TRANSPOSED whenever flash-attention is off (llama-kv-cache.cpp: v_trans = !cparams.flash_attn);
        # v_trans=False with flash_attn=False builds build_attn_mha's "avoid this branch" form (row cache + CONT(TRANSPOSE(v))) instead
        self.v_trans = (not flash_attn) if v_trans is None else bool(v_trans)
        c = cfg
        self.wctx = Context(be)
        w = self.wctx
        E, H, HK, D, F, V = c["n_embd"], c["n_head"], c["n_head_kv"], c["head_dim"], c["n_ff"], c["n_vocab"]
        self.layers = []
        for il in range(c["n_layer"]):
            t = types[il]
            L = dict(
                attn_norm=w.new_tensor(GGML_TYPE_F32, E), attn_q=w.new_tensor(t["attn_q"], E, H * D), attn_k=w.new_tensor(t["attn_k"], E, HK * D),
                attn_v=w.new_tensor(t["attn_v"], E, HK * D), attn_q_norm=w.new_tensor(GGML_TYPE_F32, D), attn_k_norm=w.new_tensor(GGML_TYPE_F32, D),
                attn_output=w.new_tensor(t["attn_output"], H * D, E), ffn_norm=w.new_tensor(GGML_TYPE_F32, E),
                ffn_gate=w.new_tensor(t["ffn_gate"], E, F), ffn_up=w.new_tensor(t["ffn_up"], E, F), ffn_down=w.new_tensor(t["ffn_down"], F, E),
                # KV cache, layout of llama_kv_cache: K [n_embd_k_gqa, kv_size]; V the same with FA, transposed without
                k_cache=w.new_tensor(GGML_TYPE_F16, HK * D, n_ctx), v_cache=w.new_tensor(GGML_TYPE_F16, HK * D, n_ctx),
            )
            self.layers.append(L)
        self.output_norm = w.new_tensor(GGML_TYPE_F32, E)
        self.output = w.new_tensor(types["output"], E, V)
        w.alloc(usage=GGML_BACKEND_BUFFER_USAGE_WEIGHTS)
        self.host = {} if host_copy else None
        rng = np.random.default_rng(seed)
        cache = {}

        def fill(T, name, il):
            ty, K, n = T.type, T.ne[0], T.ne[1]
            if weights is not None:                                   # explicit bytes (fixtures): no RNG involved
                be.tensor_set(T, weights[(il, name)])
                if self.host is not None:
                    self.host[(il, name)] = weights[(il, name)]
                return
            key = (name, ty) if share_layer_bytes else (name, ty, il)
            if key not in cache:
                cache.clear() if not share_layer_bytes else None
                cache[key] = random_blocks(rng, ty, n, K)
            be.tensor_set(T, cache[key])
            if self.host is not None:
                self.host[(il, name)] = cache[key].copy()

        for il, L in enumerate(self.layers):
            for name in ("attn_q", "attn_k", "attn_v", "attn_output", "ffn_gate", "ffn_up", "ffn_down"):
                fill(L[name], name, il)
            for name in ("attn_norm", "ffn_norm", "attn_q_norm", "attn_k_norm"):
                if weights is not None:
                    g = weights[(il, name)]
                else:
                    g = (1.0 + 0.1 * rng.standard_normal(L[name].ne[0])).astype(np.float32) if host_copy else np.ones(L[name].ne[0], np.float32)
                be.tensor_set(L[name], g)
                if self.host is not None:
                    self.host[(il, name)] = g
            be.tensor_set(L["k_cache"], np.zeros(HK * D * n_ctx, np.float16))
            be.tensor_set(L["v_cache"], np.zeros(HK * D * n_ctx, np.float16))
        g = weights[(-1, "output_norm")] if weights is not None else np.ones(E, np.float32)
        be.tensor_set(self.output_norm, g)
        fill(self.output, "output", -1)
        if self.host is not None:
            self.host[(-1, "output_norm")] = g

    def weight_bytes(self):
        """Algorithmic bytes streamed per decoded token: every matrix once."""
        n = self.output.nbytes()
        for L in self.layers:
            n += sum(L[k].nbytes() for k in ("attn_q", "attn_k", "attn_v", "attn_output", "ffn_gate", "ffn_up", "ffn_down"))
        return n

    # ---------------------------------------------------------------------------------- graph
    def build(self, n_tokens, n_kv, n_outputs=None):
        """Graph for one ubatch of `n_tokens` new tokens attending to `n_kv` cache cells (llm_build_qwen3).

        Inputs (set by the caller before graph_compute): inp_embd [n_embd, n_tokens] f32, inp_pos i32 [n_tokens],
        kq_mask f16 [n_kv, pad(n_tokens, 64)], k_idxs/v_idxs i64 [n_tokens], out_ids i32 [n_outputs].
        """
        c, be = self.cfg, self.be
        E, H, HK, D = c["n_embd"], c["n_head"], c["n_head_kv"], c["head_dim"]
        g = Context(be)
        roots = []                                                   # ggml_build_forward_expand() call order of build_attn / llm_build_qwen3
        I = dict(
            inp_embd=g.new_tensor(GGML_TYPE_F32, E, n_tokens), inp_pos=g.new_tensor(GGML_TYPE_I32, n_tokens),
            kq_mask=g.new_tensor(GGML_TYPE_F16 if self.fa else GGML_TYPE_F32, n_kv, (n_tokens + 63) // 64 * 64),
            k_idxs=g.new_tensor(GGML_TYPE_I64, n_tokens), v_idxs=g.new_tensor(GGML_TYPE_I64, n_tokens * (HK * D if self.v_trans else 1)),
        )
        if n_outputs is not None and n_outputs != n_tokens:
            I["out_ids"] = g.new_tensor(GGML_TYPE_I32, n_outputs)
        kq_scale = 1.0 / math.sqrt(D)
        llama_arch = c.get("arch") == "llama"                          # llm_build_llama (the omni TTS decoder): no q / k norm, RoPE NORM
        rope = dict(n_dims=D, mode=0 if llama_arch else GGML_ROPE_TYPE_NEOX, n_ctx_orig=c["n_ctx_orig"], freq_base=c["rope_base"], freq_scale=1.0,
                    ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0)
        f16 = 2
        inpL = I["inp_embd"]
        n_layer = c["n_layer"]
        for il, L in enumerate(self.layers):
            inpSA = inpL
            cur = g.mul(g.rms_norm(inpL, c["rms_eps"]), self._w(g, L["attn_norm"]))
            Q = g.mul_mat(self._w(g, L["attn_q"]), cur)
            K = g.mul_mat(self._w(g, L["attn_k"]), cur)
            V = g.mul_mat(self._w(g, L["attn_v"]), cur)
            Q = g.reshape(Q, D, H, n_tokens)
            K = g.reshape(K, D, HK, n_tokens)
            V = g.reshape(V, D, HK, n_tokens)
            if not llama_arch:
                Q = g.mul(g.rms_norm(Q, c["rms_eps"]), self._w(g, L["attn_q_norm"]))
            Q = g.rope_ext(Q, I["inp_pos"], None, **rope)
            if not llama_arch:
                K = g.mul(g.rms_norm(K, c["rms_eps"]), self._w(g, L["attn_k_norm"]))
            K = g.rope_ext(K, I["inp_pos"], None, **rope)
            # store into the cache (llama_kv_cache::cpy_k / cpy_v, FA layout)
            kc, vc = self._w(g, L["k_cache"]), self._w(g, L["v_cache"])
            roots += [Q, K, V]
            roots.append(g.set_rows(kc, g.view_2d(K, HK * D, n_tokens, K.nb[2], 0), I["k_idxs"]))
            if self.v_trans:                                            # cpy_v, transposed cache: one element per row index (llama-kv-cache.cpp:1091-1109)
                roots.append(g.set_rows(g.reshape(vc, 1, HK * D * self.n_ctx), g.reshape(g.reshape(V, HK * D, n_tokens), 1, HK * D * n_tokens), I["v_idxs"]))
            else:
                roots.append(g.set_rows(vc, g.view_2d(V, HK * D, n_tokens, V.nb[2], 0), I["v_idxs"]))
            # attention over the first n_kv cells (get_k / get_v views, build_attn_mha)
            k = g.view_4d(kc, D, HK, n_kv, 1, D * f16, HK * D * f16, HK * D * f16 * self.n_ctx, 0)
            if self.v_trans:                                            # get_v: [n_kv, HK, D] over the [kv_size, n_embd_v_gqa] cache (llama-kv-cache.cpp:1012-1018)
                v = g.view_4d(vc, n_kv, HK, D, 1, self.n_ctx * D * f16, self.n_ctx * f16, self.n_ctx * HK * D * f16, 0)
            else:
                v = g.view_4d(vc, D, HK, n_kv, 1, D * f16, HK * D * f16, HK * D * f16 * self.n_ctx, 0)
            q = g.permute(Q, 0, 2, 1, 3)
            k = g.permute(k, 0, 2, 1, 3)
            v = g.permute(v, 0, 2, 1, 3)
            if self.fa:
                cur = g.flash_attn_ext(q, k, v, I["kq_mask"], kq_scale)
                cur = g.reshape(cur, H * D, n_tokens)
            else:
                kq = g.mul_mat(k, q)
                kqs = g.soft_max_ext(kq, I["kq_mask"], kq_scale, 0.0)
                vt = v if self.v_trans else g.cont(g.transpose(v))    # (row cache: the "avoid this branch" path of build_attn_mha)
                kqv = g.mul_mat(vt, kqs)
                cur = g.cont(g.permute(kqv, 0, 2, 1, 3), H * D, n_tokens)
                if getattr(self, "taps", None) is not None and il == 0:   # debugging aid: keep layer 0's attention intermediates alive
                    self.taps.update(Q=Q, K=K, V=V, kq=kq, kqs=kqs, vt=vt, kqv=kqv, attn=cur)
                    roots += [kq, kqs, vt, kqv, cur]
            cur = g.mul_mat(self._w(g, L["attn_output"]), cur)
            if il == n_layer - 1 and "out_ids" in I:
                cur = g.get_rows(cur, I["out_ids"])
                inpSA = g.get_rows(inpSA, I["out_ids"])
            ffn_inp = g.add(cur, inpSA)
            cur = g.mul(g.rms_norm(ffn_inp, c["rms_eps"]), self._w(g, L["ffn_norm"]))
            up = g.mul_mat(self._w(g, L["ffn_up"]), cur)
            gate = g.mul_mat(self._w(g, L["ffn_gate"]), cur)
            cur = g.swiglu_split(gate, up)
            cur = g.mul_mat(self._w(g, L["ffn_down"]), cur)
            inpL = g.add(cur, ffn_inp)
        cur = g.mul(g.rms_norm(inpL, c["rms_eps"]), self._w(g, self.output_norm))
        self.hidden_out = None
        if getattr(self, "tap_hidden", False):                         # result_norm as llama_get_embeddings exposes it to the TTS module (omni.cpp:256-270):
            self.hidden_out = g.scale(cur, 1.0)                          # a second reader keeps the rows materialised next to the fused lm-head launch
            roots.append(self.hidden_out)
        logits = g.mul_mat(self._w(g, self.output), cur)
        roots.append(logits)
        g.roots = roots
        g.alloc()
        return g, I, logits

    def _w(self, g, real):
        """Leaf tensor inside graph context `g` that aliases a weight / cache tensor of the model buffer
        (a zero-offset view, so Context.alloc() resolves its data pointer into the model's buffer)."""
        T = g._new(real.type, real.ne, view_src=real, view_offs=0)
        for i in range(4):
            T.t.nb[i] = real.t.nb[i]
        return T

    # ---------------------------------------------------------------------------------- inputs
    def set_inputs(self, I, embd, pos0, n_kv, n_seq=1):
        """Causal decode/prefill inputs for tokens at positions pos0 .. pos0+n-1 written to cache cells of the same index.
        n_seq > 1: the ubatch holds n_seq equal-length sequences back to back (unified KV cache, llama_kv_cache::set_input_kq_mask with
        several seq_ids): token i of sequence s sits at position i in cell s * len + i and sees only its own sequence's earlier cells."""
        be = self.be
        n = embd.shape[0]
        be.tensor_set(I["inp_embd"], embd.astype(np.float32))
        if n_seq > 1:
            assert pos0 == 0 and n % n_seq == 0 and not self.v_trans
            ln = n // n_seq
            be.tensor_set(I["inp_pos"], np.tile(np.arange(ln, dtype=np.int32), n_seq))
            cells = np.arange(n, dtype=np.int64)
            be.tensor_set(I["k_idxs"], cells); be.tensor_set(I["v_idxs"], cells)
            npad = I["kq_mask"].ne[1]
            m = np.full((npad, n_kv), -np.inf, dtype=np.float16 if self.fa else np.float32)
            blk = np.where(np.tril(np.ones((ln, ln), bool)), 0.0, -np.inf).astype(m.dtype)
            for s in range(n_seq):
                m[s * ln:(s + 1) * ln, s * ln:(s + 1) * ln] = blk
            be.tensor_set(I["kq_mask"], m)
            return
        pos = np.arange(pos0, pos0 + n, dtype=np.int32)
        be.tensor_set(I["inp_pos"], pos)
        be.tensor_set(I["k_idxs"], pos.astype(np.int64))
        if self.v_trans:                                                # set_input_v_idxs: j * kv_size + cell for every element j of the row (llama-kv-cache.cpp:1169-1183)
            nv = self.cfg["n_head_kv"] * self.cfg["head_dim"]
            be.tensor_set(I["v_idxs"], (np.arange(nv, dtype=np.int64)[None, :] * self.n_ctx + pos.astype(np.int64)[:, None]).ravel())
        else:
            be.tensor_set(I["v_idxs"], pos.astype(np.int64))
        npad = I["kq_mask"].ne[1]
        m = np.full((npad, n_kv), -np.inf, dtype=np.float32)
        for i in range(n):
            m[i, : pos0 + i + 1] = 0.0                                 # llama_kv_cache::set_input_kq_mask (causal)
        be.tensor_set(I["kq_mask"], m.astype(np.float16) if self.fa else m)
