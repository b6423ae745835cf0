"""encoders.py -- host mirror of the omni encoder graphs, node for node as the reference emits them:

  whisper(...)  : APM, `build_whisper` (reference tools/omni/audition.cpp:341-715, the branch without the streaming KV cache):
                  conv1d_ph + bias + GELU (x2, the second with stride 2), learned positions, per layer LayerNorm -> q/k/v (+bias) ->
                  K / V cast to F16 -> KQ -> soft_max_ext (no mask) -> KQV -> out proj (+bias) -> residual -> LayerNorm -> MLP(GELU) ->
                  residual; final LayerNorm, two projections with ReLU, avg-pool(5) over the tokens.
  resampler(...): VPM projector, the second half of `build_minicpmv` (vision.cpp:292-377): 64 learned queries cross-attend over the patches
  siglip2(...)  : VPM, `build_inp` + `build_vit` (reference tools/omni/vision.cpp:394-705): ggml_conv_2d patch embedding (+bias), learned
                  positions, per layer LayerNorm -> q/k/v (+bias) -> f32 KQ -> soft_max_ext -> KQV -> out proj (+bias) -> residual ->
                  LayerNorm -> FFN(GELU, biases) -> residual; post LayerNorm.

Test / bench harness only: no arithmetic happens here, the graphs run on whatever backend the Context belongs to (the MI355X plug-in or
the reference CPU backend), which is how tests/ compare the two.
"""
import numpy as np

from .ggml import GGML_TYPE_F16, GGML_TYPE_F32, UNARY

WHISPER = dict(n_mels=80, n_state=1024, n_head=16, n_ctx=1500, eps=1e-5, d_proj=4096)      # MiniCPM-o APM: Whisper-medium encoder widths
SIGLIP2 = dict(image=448, patch=14, n_embd=1152, n_head=16, n_ff=4304, eps=1e-6)           # VPM: SigLip2-so400m widths (head_dim 72)
RESAMPLER = dict(n_embd=1152, n_out=4096, n_query=64, d_head=128, eps=1e-6)               # MiniCPM-V resampler projector: 64 queries x 4096 over the patches


def conv_1d_ph(c, kernel, x, s, d):
    """ggml_conv_1d_ph (ggml.c): "half" padding kernel / 2"""
    return c.conv_1d(kernel, x, s, kernel.ne[0] // 2, d)


def conv_2d(c, kernel, x, s0, s1, p0, p1, d0, d1):
    """ggml_conv_2d (ggml.c): im2col in the kernel's type, one MUL_MAT, reshape, permute, cont"""
    col = c.im2col(kernel, x, s0, s1, p0, p1, d0, d1, True, kernel.type)
    r = c.mul_mat(c.reshape(col, col.ne[0], col.ne[3] * col.ne[2] * col.ne[1]), c.reshape(kernel, kernel.ne[0] * kernel.ne[1] * kernel.ne[2], kernel.ne[3]))
    r = c.reshape(r, col.ne[1], col.ne[2], col.ne[3], kernel.ne[3])
    return c.cont(c.permute(r, 0, 1, 3, 2))


def layer_norm(c, x, w, b, eps):
    y = c.norm(x, eps)
    if w is not None:
        y = c.mul(y, w)
    if b is not None:
        y = c.add(y, b)
    return y


def whisper_weights(c, hp, n_layer, wtype=GGML_TYPE_F16):
    """tensor set of the encoder (names as audition.cpp's model struct); matrices `wtype`, biases / norms / positions f32"""
    S, M = hp["n_state"], hp["n_mels"]
    f32 = GGML_TYPE_F32
    W = dict(conv_1_w=c.new_tensor(GGML_TYPE_F16, 3, M, S), conv_1_b=c.new_tensor(f32, 1, S), conv_2_w=c.new_tensor(GGML_TYPE_F16, 3, S, S), conv_2_b=c.new_tensor(f32, 1, S),
             pe=c.new_tensor(f32, S, hp["n_ctx"]), ln_w=c.new_tensor(f32, S), ln_b=c.new_tensor(f32, S),
             proj_1_w=c.new_tensor(wtype, S, hp["d_proj"]), proj_1_b=c.new_tensor(f32, hp["d_proj"]),
             proj_2_w=c.new_tensor(wtype, hp["d_proj"], hp["d_proj"]), proj_2_b=c.new_tensor(f32, hp["d_proj"]), layers=[])
    for _ in range(n_layer):
        W["layers"].append(dict(ln0_w=c.new_tensor(f32, S), ln0_b=c.new_tensor(f32, S), q_w=c.new_tensor(wtype, S, S), q_b=c.new_tensor(f32, S),
                                k_w=c.new_tensor(wtype, S, S), v_w=c.new_tensor(wtype, S, S), v_b=c.new_tensor(f32, S),
                                o_w=c.new_tensor(wtype, S, S), o_b=c.new_tensor(f32, S), ln1_w=c.new_tensor(f32, S), ln1_b=c.new_tensor(f32, S),
                                m0_w=c.new_tensor(wtype, S, 4 * S), m0_b=c.new_tensor(f32, 4 * S), m1_w=c.new_tensor(wtype, 4 * S, S), m1_b=c.new_tensor(f32, S)))
    return W


def whisper(c, hp, W, n_frames):
    """returns (inp_raw, out): inp_raw f32 [n_frames, n_mels]; out f32 [d_proj, n_frames / 2 / 5]"""
    S, H = hp["n_state"], hp["n_head"]
    D = S // H
    inp = c.new_tensor(GGML_TYPE_F32, n_frames, hp["n_mels"])
    cur = c.unary(c.add(conv_1d_ph(c, W["conv_1_w"], inp, 1, 1), W["conv_1_b"]), UNARY.GELU)
    cur = c.unary(c.add(conv_1d_ph(c, W["conv_2_w"], cur, 2, 1), W["conv_2_b"]), UNARY.GELU)
    n_tok = cur.ne[0]
    pe = c.view_2d(W["pe"], S, n_tok, S * 4, 0)
    cur = c.add(pe, c.cont(c.transpose(c.reshape(cur, cur.ne[0], cur.ne[1]))))
    inpL = cur
    scale = 1.0 / np.sqrt(float(D))
    for L in W["layers"]:
        cur = layer_norm(c, inpL, L["ln0_w"], L["ln0_b"], hp["eps"])
        Q = c.add(c.mul_mat(L["q_w"], cur), L["q_b"])
        K = c.mul_mat(L["k_w"], cur)
        V = c.add(c.mul_mat(L["v_w"], cur), L["v_b"])
        Q = c.permute(c.reshape(Q, D, H, n_tok), 0, 2, 1, 3)
        K = c.permute(c.cast(c.reshape(K, D, H, n_tok), GGML_TYPE_F16), 0, 2, 1, 3)
        V = c.cast(c.permute(c.reshape(V, D, H, n_tok), 1, 2, 0, 3), GGML_TYPE_F16)
        KQ = c.soft_max_ext(c.mul_mat(K, Q), None, scale, 0.0)
        KQV = c.mul_mat(V, KQ)
        cur = c.cont(c.permute(KQV, 0, 2, 1, 3), S, n_tok)
        cur = c.add(c.add(c.mul_mat(L["o_w"], cur), L["o_b"]), inpL)
        inpFF = cur
        cur = layer_norm(c, inpFF, L["ln1_w"], L["ln1_b"], hp["eps"])
        cur = c.unary(c.add(c.mul_mat(L["m0_w"], cur), L["m0_b"]), UNARY.GELU)
        cur = c.add(c.mul_mat(L["m1_w"], cur), L["m1_b"])
        inpL = c.add(cur, inpFF)
    cur = layer_norm(c, inpL, W["ln_w"], W["ln_b"], hp["eps"])
    cur = c.unary(c.add(c.mul_mat(W["proj_1_w"], cur), W["proj_1_b"]), UNARY.RELU)
    cur = c.add(c.mul_mat(W["proj_2_w"], cur), W["proj_2_b"])
    cur = c.cpy(c.permute(cur, 1, 0, 2, 3), c.new_tensor(GGML_TYPE_F32, cur.ne[1], cur.ne[0]))
    cur = c.pool_1d(cur, 1, 5, 5, 0)                                  # GGML_OP_POOL_AVG over the tokens
    cur = c.cpy(c.permute(cur, 1, 0, 2, 3), c.new_tensor(GGML_TYPE_F32, cur.ne[1], cur.ne[0]))
    return inp, cur


def siglip2_weights(c, hp, n_layer, wtype=GGML_TYPE_F16):
    E, F, P = hp["n_embd"], hp["n_ff"], hp["patch"]
    n_pos = (hp["image"] // P) ** 2
    f32 = GGML_TYPE_F32
    W = dict(patch_w=c.new_tensor(GGML_TYPE_F16, P, P, 3, E), patch_b=c.new_tensor(f32, E), pos=c.new_tensor(f32, E, n_pos),
             post_ln_w=c.new_tensor(f32, E), post_ln_b=c.new_tensor(f32, E), layers=[])
    for _ in range(n_layer):
        W["layers"].append(dict(ln1_w=c.new_tensor(f32, E), ln1_b=c.new_tensor(f32, E), ln2_w=c.new_tensor(f32, E), ln2_b=c.new_tensor(f32, E),
                                q_w=c.new_tensor(wtype, E, E), q_b=c.new_tensor(f32, E), k_w=c.new_tensor(wtype, E, E), k_b=c.new_tensor(f32, E),
                                v_w=c.new_tensor(wtype, E, E), v_b=c.new_tensor(f32, E), o_w=c.new_tensor(wtype, E, E), o_b=c.new_tensor(f32, E),
                                up_w=c.new_tensor(wtype, E, F), up_b=c.new_tensor(f32, F), down_w=c.new_tensor(wtype, F, E), down_b=c.new_tensor(f32, E)))
    return W


def siglip2(c, hp, W):
    """returns (inp_raw, out): inp_raw f32 [image, image, 3]; out f32 [n_embd, n_patches]"""
    E, H, P = hp["n_embd"], hp["n_head"], hp["patch"]
    D = E // H
    n_pos = (hp["image"] // P) ** 2
    inp = c.new_tensor(GGML_TYPE_F32, hp["image"], hp["image"], 3)
    x = conv_2d(c, W["patch_w"], inp, P, P, 0, 0, 1, 1)
    x = c.cont(c.transpose(c.reshape(x, n_pos, E)))
    x = c.add(x, W["patch_b"])
    inpL = c.add(x, W["pos"])
    scale = 1.0 / np.sqrt(float(D))
    for L in W["layers"]:
        cur = layer_norm(c, inpL, L["ln1_w"], L["ln1_b"], hp["eps"])
        Q = c.reshape(c.add(c.mul_mat(L["q_w"], cur), L["q_b"]), D, H, n_pos)
        K = c.reshape(c.add(c.mul_mat(L["k_w"], cur), L["k_b"]), D, H, n_pos)
        V = c.reshape(c.add(c.mul_mat(L["v_w"], cur), L["v_b"]), D, H, n_pos)
        q, k = c.permute(Q, 0, 2, 1, 3), c.permute(K, 0, 2, 1, 3)
        v = c.cont(c.permute(V, 1, 2, 0, 3))
        kq = c.soft_max_ext(c.mul_mat(k, q), None, scale, 0.0)
        kqv = c.mul_mat(v, kq)
        cur = c.cont(c.permute(kqv, 0, 2, 1, 3), E, n_pos)
        cur = c.add(c.add(c.mul_mat(L["o_w"], cur), L["o_b"]), inpL)
        inpL = cur
        cur = layer_norm(c, cur, L["ln2_w"], L["ln2_b"], hp["eps"])
        cur = c.unary(c.add(c.mul_mat(L["up_w"], cur), L["up_b"]), UNARY.GELU)
        cur = c.add(c.mul_mat(L["down_w"], cur), L["down_b"])
        inpL = c.add(inpL, cur)
    return inp, layer_norm(c, inpL, W["post_ln_w"], W["post_ln_b"], hp["eps"])


def resampler_weights(c, rp, wtype=GGML_TYPE_F16):
    E, O, NQ = rp["n_embd"], rp["n_out"], rp["n_query"]
    f32 = GGML_TYPE_F32
    t = c.new_tensor
    return dict(query=t(f32, O, NQ), kv_proj_w=t(wtype, E, O), ln_q_w=t(f32, O), ln_q_b=t(f32, O), ln_kv_w=t(f32, O), ln_kv_b=t(f32, O),
                q_w=t(wtype, O, O), q_b=t(f32, O), k_w=t(wtype, O, O), k_b=t(f32, O), v_w=t(wtype, O, O), v_b=t(f32, O), o_w=t(wtype, O, O), o_b=t(f32, O),
                ln_post_w=t(f32, O), ln_post_b=t(f32, O), proj_w=t(wtype, O, O))


def resampler(c, rp, W, embeddings, n_pos):
    """the projector half of build_minicpmv (reference tools/omni/vision.cpp:292-377): kv projection of the ViT output, LayerNorms, k = v +
    pos_embed, cross-attention of the learned queries over the patches through build_attn (:648-703: permutes, CONT of V^T, K.Q^T,
    soft_max_ext, V product, CONT), post LayerNorm, output projection.  returns (pos_embed input [n_out, n_pos], out [n_out, n_query])"""
    O, NQ, D = rp["n_out"], rp["n_query"], rp["d_head"]
    H = O // D
    pos_embed = c.new_tensor(GGML_TYPE_F32, O, n_pos)
    q = layer_norm(c, W["query"], W["ln_q_w"], W["ln_q_b"], rp["eps"])
    v = layer_norm(c, c.mul_mat(W["kv_proj_w"], embeddings), W["ln_kv_w"], W["ln_kv_b"], rp["eps"])
    k = c.add(v, pos_embed)
    Q = c.reshape(c.add(c.mul_mat(W["q_w"], q), W["q_b"]), D, H, NQ)
    K = c.reshape(c.add(c.mul_mat(W["k_w"], k), W["k_b"]), D, H, n_pos)
    V = c.reshape(c.add(c.mul_mat(W["v_w"], v), W["v_b"]), D, H, n_pos)
    qp, kp = c.permute(Q, 0, 2, 1, 3), c.permute(K, 0, 2, 1, 3)
    vt = c.cont(c.permute(V, 1, 2, 0, 3))
    kq = c.soft_max_ext(c.mul_mat(kp, qp), None, 1.0 / np.sqrt(float(D)), 0.0)
    kqv = c.mul_mat(vt, kq)
    cur = c.cont(c.permute(kqv, 0, 2, 1, 3), D * H, NQ)
    cur = c.add(c.mul_mat(W["o_w"], cur), W["o_b"])
    cur = layer_norm(c, cur, W["ln_post_w"], W["ln_post_b"], rp["eps"])
    return pos_embed, c.mul_mat(W["proj_w"], cur)


def declined_nodes(backend, graph_ctx):
    """nodes of the context's graph the backend's supports_op refuses (the scheduler would leave them on the CPU)"""
    bad = []
    for t in graph_ctx.nodes:
        if not backend.supports_op(t):
            bad.append((int(t.t.op), int(t.t.type), tuple(t.ne)))
    return bad
