"""token2wav.py -- host mirror of pieces of the Token2Wav graphs, node for node as the reference emits them
(reference tools/omni/token2wav/token2wav-impl.cpp; all weights f32 there, convolutions as im2col(F32) + MUL_MAT):

  dit_block(...)       : flow-matching DiT block, fmDiTBlock::build_forward_graph (:1451-1487): adaLN (SiLU -> linear -> nine strided chunk
                         views, :1373-1380), LayerNorm + modulate (:1121-1164), attention with q / k LayerNorm over head_dim, batched
                         K.Q^T -> scale -> soft_max_ext -> V^T product (:406-439 and the permute / cont helpers :245-291), the causal
                         conv block (pad_ext left + im2col + MUL_MAT, LayerNorm, Mish spelled with sub / exp / add / log / tanh / mul,
                         :866-932, :1027-1050, :1138-1150), the GELU MLP, gates and residuals.  hidden 512, 8 heads x 64, mlp 2048 (:695-702)
  timestep_embedder    : timestep_embedding(256, 10000) -> linear -> SiLU -> linear (:2452 and around)
  hift_upsample_stage  : HiFT generator stage: leaky_relu -> conv_transpose_1d (+ trim, bias through repeat) -> snake -> dilated conv1d ->
                         snake -> conv1d -> residual (:4955-4990, :5925-5941, :5136-5235)
  istft_head           : exp / clamp magnitude, sin / cos of sin(phase), overlap-add as conv_transpose_1d with hop 4, window-sum normalisation
                         with clamp + div (:5254-5275, :5406-5421)
  speaker_norm         : sqr -> sum_rows -> add eps -> sqrt -> div (:78-88)
  length_mask          : arange -> repeat_4d -> sub -> step -> cast to i32 (:2740-2753)

Test / bench harness only: no arithmetic happens here; the graphs run on whatever backend the Context belongs to.
"""
import numpy as np

from .ggml import GGML_TYPE_F32, GGML_TYPE_I32, UNARY

DIT = dict(hidden=512, n_head=8, head_dim=64, mlp=2048, eps=1e-5, freq_dim=256)
HIFT = dict(ch_in=512, ch_out=256, up_k=16, up_s=8, lrelu=0.1, rb_k=7, rb_dil=3, n_fft=16, hop=4)


def linear(c, x, w, b):
    y = c.mul_mat(w, x)
    return c.add(y, b) if b is not None else y


def layer_norm(c, x, w, b, eps):
    y = c.norm(x, eps)
    if w is not None:
        y = c.mul(y, w)
    if b is not None:
        y = c.add(y, b)
    return y


def mish(c, x):
    zeros = c.sub(x, x)
    ones = c.unary(zeros, UNARY.EXP)
    sp = c.log(c.add(c.unary(x, UNARY.EXP), ones))
    return c.mul(x, c.unary(sp, UNARY.TANH))


def modulate(c, x, shift, scale):
    return c.add(c.add(x, c.mul(x, scale)), shift)


def causal_conv1d(c, x_ctb, w, b):
    """x [C, T, 1] -> [Cout, T, 1]; w [K, Cin, Cout] f32 (fmCausalConv1d::build_forward_graph, B == 1)"""
    K, Cin, Cout = w.ne[0], w.ne[1], w.ne[2]
    x_tcb = c.cont(c.permute(x_ctb, 1, 0, 2, 3))
    x_pad = c.pad_ext(x_tcb, K - 1, 0, 0, 0, 0, 0, 0, 0)
    col = c.im2col(w, x_pad, 1, 0, 0, 0, 1, 0, False, GGML_TYPE_F32)
    mm = c.mul_mat(c.reshape(col, col.ne[0], col.ne[2] * col.ne[1]), c.reshape(w, K * Cin, Cout))
    y = c.cont(c.permute(c.reshape(mm, col.ne[1], Cout, col.ne[2]), 1, 0, 2, 3))
    return c.add(y, c.reshape(b, Cout, 1, 1)) if b is not None else y


def causal_conv1d_chunk(c, x_ctb, cache_ctb, w, b, want_cache=True):
    """Streaming form: x [C, dt, B] behind the cached K - 1 frames [C, K - 1, B] -> (y [Cout, dt, B], new cache [C, K - 1, B]);
    fmCausalConv1d::build_forward_chunk_graph (token2wav-impl.cpp:925-1000), node for node with a cache given: the cache and x transposed
    to [T, C, B] copies, CONCAT on the time axis + CONT, per batch element VIEW -> IM2COL(F32) -> MUL_MAT, CONCAT over the batch,
    PERMUTE + CONT back, ADD of the bias; the new cache is the tail of CONT(CONCAT(cache, x) on dim 1).  The plug-in runs the y branch
    as one dense concat + one any-shape GEMM over overlapping rows (graph_exec_t2w.cpp exec_causal_conv)."""
    K, Cin, Cout = w.ne[0], w.ne[1], w.ne[2]
    dt, B = x_ctb.ne[1], x_ctb.ne[2]
    cache_in = c.cont(cache_ctb)
    cache_tcb = c.cont(c.permute(cache_in, 1, 0, 2, 3))
    x_tcb = c.cont(c.permute(x_ctb, 1, 0, 2, 3))
    x_cat = c.cont(c.concat(cache_tcb, x_tcb, 0))
    y_tcb = None
    for bi in range(B):
        xb = c.view_3d(x_cat, x_cat.ne[0], x_cat.ne[1], 1, x_cat.nb[1], x_cat.nb[2], x_cat.nb[2] * bi)
        col = c.im2col(w, xb, 1, 0, 0, 0, 1, 0, False, GGML_TYPE_F32)
        mm = c.mul_mat(c.reshape(col, col.ne[0], col.ne[2] * col.ne[1]), c.reshape(w, K * Cin, Cout))
        yb = c.reshape(mm, col.ne[1], Cout, col.ne[2])
        y_tcb = yb if y_tcb is None else c.concat(y_tcb, yb, 2)
    y = c.cont(c.permute(y_tcb, 1, 0, 2, 3))
    if b is not None:
        y = c.add(y, c.reshape(b, Cout, 1, 1))
    new_cache = None
    if want_cache:
        x_cont = x_ctb if x_ctb.t.op in (0, 35) else c.cont(x_ctb)      # (NONE / RESHAPE are taken as they are)
        cat = c.cont(c.concat(cache_in, x_cont, 1))
        new_cache = c.cont(c.view_3d(cat, Cin, K - 1, B, cat.nb[1], cat.nb[2], cat.nb[1] * dt))
    return y, new_cache


def dit_weights(c, hp):
    E, D, M = hp["hidden"], hp["head_dim"], hp["mlp"]
    f = GGML_TYPE_F32
    t = c.new_tensor
    return dict(ada_w=t(f, E, 9 * E), ada_b=t(f, 9 * E), n1_w=t(f, E), n1_b=t(f, E), n2_w=t(f, E), n2_b=t(f, E), n3_w=t(f, E), n3_b=t(f, E),
                q_w=t(f, E, E), q_b=t(f, E), k_w=t(f, E, E), k_b=t(f, E), v_w=t(f, E, E), v_b=t(f, E), qn_w=t(f, D), qn_b=t(f, D), kn_w=t(f, D), kn_b=t(f, D),
                o_w=t(f, E, E), o_b=t(f, E), c1_w=t(f, 3, E, E), c1_b=t(f, E), cln_w=t(f, E), cln_b=t(f, E), c2_w=t(f, 3, E, E), c2_b=t(f, E),
                m1_w=t(f, E, M), m1_b=t(f, M), m2_w=t(f, M, E), m2_b=t(f, E))


def dit_block(c, hp, W, T):
    """returns (x_in [hidden, T, 1], cond [hidden, 1, 1], out [hidden, T, 1])"""
    E, H, D, eps = hp["hidden"], hp["n_head"], hp["head_dim"], hp["eps"]
    x = c.new_tensor(GGML_TYPE_F32, E, T, 1)
    cond = c.new_tensor(GGML_TYPE_F32, E, 1, 1)
    ada = linear(c, c.unary(cond, UNARY.SILU), W["ada_w"], W["ada_b"])                   # [9 E, 1, 1]
    ch = [c.view_3d(ada, E, ada.ne[1], ada.ne[2], ada.nb[1], ada.nb[2], i * E * 4) for i in range(9)]
    shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp, shift_conv, scale_conv, gate_conv = ch
    # attention
    h = modulate(c, layer_norm(c, x, W["n1_w"], W["n1_b"], eps), shift_msa, scale_msa)
    q = c.reshape(linear(c, h, W["q_w"], W["q_b"]), D, H, T, 1)
    k = c.reshape(linear(c, h, W["k_w"], W["k_b"]), D, H, T, 1)
    v = c.reshape(linear(c, h, W["v_w"], W["v_b"]), D, H, T, 1)
    q = layer_norm(c, q, W["qn_w"], W["qn_b"], eps)
    k = layer_norm(c, k, W["kn_w"], W["kn_b"], eps)
    flat = lambda t: c.reshape(c.cont(c.permute(t, 0, 2, 1, 3)), D, T, H)                 # noqa: E731
    qf, kf = flat(q), flat(k)
    vf = c.reshape(c.cont(c.permute(flat(v), 1, 0, 2, 3)), T, D, H)
    scores = c.scale(c.mul_mat(kf, qf), 1.0 / np.sqrt(float(D)))
    probs = c.soft_max_ext(scores, None, 1.0, 0.0)
    ctxv = c.mul_mat(vf, probs)                                                           # [D, T, H]
    merged = c.reshape(c.cont(c.permute(c.reshape(ctxv, D, T, H, 1), 0, 2, 1, 3)), D * H, T, 1)
    x1 = c.add(x, c.mul(linear(c, merged, W["o_w"], W["o_b"]), gate_msa))
    # causal conv block
    h = modulate(c, layer_norm(c, x1, W["n3_w"], W["n3_b"], eps), shift_conv, scale_conv)
    h = causal_conv1d(c, h, W["c1_w"], W["c1_b"])
    h = mish(c, layer_norm(c, h, W["cln_w"], W["cln_b"], 1e-5))
    h = causal_conv1d(c, h, W["c2_w"], W["c2_b"])
    x2 = c.add(x1, c.mul(h, gate_conv))
    # MLP
    h = modulate(c, layer_norm(c, x2, W["n2_w"], W["n2_b"], eps), shift_mlp, scale_mlp)
    h = linear(c, c.unary(linear(c, h, W["m1_w"], W["m1_b"]), UNARY.GELU), W["m2_w"], W["m2_b"])
    return x, cond, c.add(x2, c.mul(h, gate_mlp))


def timestep_embedder(c, hp, w1, b1, w2, b2, n):
    """t [n] -> [hidden, n]"""
    t = c.new_tensor(GGML_TYPE_F32, n)
    emb = c.timestep_embedding(c.scale(t, 1000.0), hp["freq_dim"], 10000)
    return t, linear(c, c.unary(linear(c, emb, w1, b1), UNARY.SILU), w2, b2)


def snake(c, x_tcb, alpha):
    """hg2_snake_build_graph: x + sin(alpha x)^2 / (alpha + 1e-9)"""
    C = x_tcb.ne[1]
    a = c.repeat(c.reshape(alpha, 1, C, 1), x_tcb)
    s = c.sin(c.mul(x_tcb, a))
    return c.add(x_tcb, c.div(c.mul(s, s), c.scale(a, 1.0, 1e-9)))


def conv1d_same(c, x_tcb, w, b, dilation):
    """x [T, Cin, 1], w [K, Cin, Cout] f32 -> [T, Cout, 1] ('same' padding, im2col F32 + MUL_MAT, bias through repeat)"""
    K, Cin, Cout = w.ne[0], w.ne[1], w.ne[2]
    pad = (K - 1) * dilation // 2
    col = c.im2col(w, x_tcb, 1, 0, pad, 0, dilation, 0, False, GGML_TYPE_F32)
    mm = c.mul_mat(c.reshape(col, col.ne[0], col.ne[2] * col.ne[1]), c.reshape(w, K * Cin, Cout))
    y = c.reshape(mm, col.ne[1], Cout, col.ne[2])
    return c.add(y, c.repeat(c.reshape(b, 1, Cout, 1), y))


def hift_weights(c, hp):
    f = GGML_TYPE_F32
    t = c.new_tensor
    Ci, Co, K = hp["ch_in"], hp["ch_out"], hp["rb_k"]
    return dict(up_w=t(f, hp["up_k"], Co, Ci), up_b=t(f, Co), a1=t(f, Co), a2=t(f, Co), c1_w=t(f, K, Co, Co), c1_b=t(f, Co), c2_w=t(f, K, Co, Co), c2_b=t(f, Co))


def hift_upsample_stage(c, hp, W, T):
    """x [T, ch_in, 1] -> [T * up_s, ch_out, 1]"""
    Ci, Co, s, K = hp["ch_in"], hp["ch_out"], hp["up_s"], hp["up_k"]
    x = c.new_tensor(GGML_TYPE_F32, T, Ci, 1)
    h = c.leaky_relu(x, hp["lrelu"])
    full = c.cont(c.conv_transpose_1d(W["up_w"], c.cont(c.reshape(h, T, Ci)), s))         # [(T - 1) s + K, Co]
    pad = (K - s) // 2
    L = full.ne[0] - 2 * pad
    y = c.cont(c.view_2d(full, L, Co, full.nb[1], pad * 4))
    y = c.reshape(c.add(y, c.repeat(c.cont(c.reshape(W["up_b"], 1, Co)), y)), L, Co, 1)
    r = conv1d_same(c, snake(c, y, W["a1"]), W["c1_w"], W["c1_b"], hp["rb_dil"])
    r = conv1d_same(c, snake(c, r, W["a2"]), W["c2_w"], W["c2_b"], 1)
    return x, c.add(y, r)


def istft_head(c, hp, T):
    """mag_log [T, F], raw_phase [T, F] (F = n_fft / 2 + 1) -> (real, imag) spectra and the overlap-added, normalised wave of a
    [n_fft, 1, n_fft] synthesis kernel applied to a [T, n_fft] frame matrix (the OLA + window-sum division of :5406-5421)"""
    F, N, hop = hp["n_fft"] // 2 + 1, hp["n_fft"], hp["hop"]
    mag_log = c.new_tensor(GGML_TYPE_F32, T, F)
    raw_phase = c.new_tensor(GGML_TYPE_F32, T, F)
    mag = c.clamp(c.unary(mag_log, UNARY.EXP), -1e30, 1e2)
    phase = c.sin(raw_phase)
    real, imag = c.mul(mag, c.cos(phase)), c.mul(mag, c.sin(phase))
    frames = c.new_tensor(GGML_TYPE_F32, T, N)
    wsq = c.new_tensor(GGML_TYPE_F32, T, N)
    ola = c.new_tensor(GGML_TYPE_F32, N, 1, N)
    y = c.conv_transpose_1d(ola, frames, hop)
    wsum = c.clamp(c.conv_transpose_1d(ola, wsq, hop), 1e-8, 1e30)
    wave = c.clamp(c.div(y, wsum), -0.99, 0.99)
    return dict(mag_log=mag_log, raw_phase=raw_phase, frames=frames, wsq=wsq, ola=ola), [real, imag, wave]


def speaker_norm(c, C, B):
    x = c.new_tensor(GGML_TYPE_F32, C, B)
    eps = c.new_tensor(GGML_TYPE_F32, 1, B)
    den = c.sqrt(c.add(c.sum_rows(c.sqr(x)), eps))
    return x, eps, c.div(x, c.repeat(den, x))


def length_mask(c, max_len, B):
    """lengths [1, B] f32 -> (valid f32 [max_len, B], pad mask i32 [max_len, B])"""
    lengths = c.new_tensor(GGML_TYPE_F32, 1, B)
    rng = c.repeat_4d(c.arange(0.0, float(max_len), 1.0), max_len, B, 1, 1)
    valid = c.unary(c.sub(c.repeat_4d(lengths, max_len, B, 1, 1), rng), UNARY.STEP)
    one = c.arange(1.0, 2.0, 1.0)
    padf = c.sub(c.repeat_4d(one, max_len, B, 1, 1), valid)
    return lengths, valid, c.cast(padf, GGML_TYPE_I32)
