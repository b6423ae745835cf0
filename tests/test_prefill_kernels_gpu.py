"""Round-2 prefill kernels, each against the REFERENCE CPU backend on the same graph (oracle/_ref) or numpy:
the eight-phase 256 x 256 GEMM at ragged shapes and odd K-step counts, the wave-per-row RMS_NORM / SOFT_MAX (incl. the f16 activation image they
emit for the next MUL_MAT), the transposing single-element SET_ROWS with arbitrary indices, and the 16-lanes-per-row norm / rope kernel."""
import numpy as np
import pytest

from conftest import nmse

pytestmark = pytest.mark.gpu


def _both(pkg, be, ref_be, build, feeds):
    res = []
    for backend in (be, ref_be):
        c = pkg.Context(backend)
        ins, outs = build(c)
        c.alloc()
        for name, t in ins.items():
            backend.tensor_set(t, feeds[name])
        backend.graph_compute(c.graph())
        res.append([backend.tensor_get(o).copy() for o in outs])
        c.free()
    return res


@pytest.mark.parametrize("M,N,K", [(8190, 4090, 320), (4096, 4096, 1088), (8192, 2048, 64)])
def test_eight_phase_gemm_full_result_ragged_and_odd_ksteps(pkg, be, M, N, K):
    """k_gemm_f16_ph8: every output element (not a sample) against numpy on f16-rounded activations, with tile rows / columns past M / N
    (clamped DMA sources, guarded stores), an odd number of K-steps (the unrolled pair loop's tail) and a single K-step (prologue only);
    six repetitions must be bit-identical (the schedule's RAW / WAR distances do not depend on timing)."""
    from test_gpu_parity import run_graph  # noqa: F401
    rng = np.random.default_rng(M + N + K)
    wv = (rng.standard_normal((M, K)) * 0.05).astype(np.float16)
    xv = rng.standard_normal((N, K)).astype(np.float32)
    c = pkg.Context(be)
    w = c.new_tensor(pkg.GGML_TYPE_F16, K, M); x = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
    y = c.mul_mat(w, x)
    c.alloc()
    be.tensor_set(w, wv); be.tensor_set(x, xv)
    before = be.get_stat("gemm256_launches")
    outs = []
    for _ in range(6):
        be.graph_compute(c.graph())
        outs.append(be.tensor_get(y).copy().reshape(N, M))
    assert be.get_stat("gemm256_launches") - before == 6, "the shape did not select the 256 x 256 kernel"
    c.free()
    want = xv.astype(np.float16).astype(np.float32) @ wv.astype(np.float32).T
    assert np.isfinite(outs[0]).all()
    assert np.abs(outs[0] - want).max() <= 2e-5 * max(1.0, np.abs(want).max()) * np.sqrt(K / 64)
    for o in outs[1:]:
        assert np.array_equal(outs[0], o)


@pytest.mark.parametrize("n,rows", [(4096, 700), (1152, 333), (8192, 65), (2048, 64)])
def test_rms_norm_rows_vs_reference_backend(pkg, be, ref_be, n, rows):
    rng = np.random.default_rng(n)

    def build(c):
        x = c.new_tensor(pkg.GGML_TYPE_F32, n, rows); w = c.new_tensor(pkg.GGML_TYPE_F32, n)
        return dict(x=x, w=w), [c.mul(c.rms_norm(x, 1e-6), w), c.rms_norm(x, 1e-5)]
    feeds = dict(x=(rng.standard_normal(n * rows) * 3).astype(np.float32), w=(1 + 0.1 * rng.standard_normal(n)).astype(np.float32))
    got, want = _both(pkg, be, ref_be, build, feeds)
    for g, w in zip(got, want):
        assert nmse(g, w) < 1e-13, nmse(g, w)


@pytest.mark.parametrize("n,mask_type", [(512, "f32"), (1000, "f16"), (2048, "f32"), (256, None), (2560, "f32"), (8192, "f16")])
def test_soft_max_rows_and_its_f16_image_vs_reference_backend(pkg, be, ref_be, n, mask_type):
    """SOFT_MAX over [n, 96, 6] with a causal-style mask (rows with -inf tails), alone (f32 result) and followed by the per-head MUL_MAT that
    makes the kernel emit the f16 activation image instead of the f32 block (V^T . P of the flash-attention-off graph)."""
    rng = np.random.default_rng(n)
    T, H, D = 96, 6, 64
    F32, F16 = pkg.GGML_TYPE_F32, pkg.GGML_TYPE_F16

    def build(c):
        x = c.new_tensor(F32, n, T, H)
        ins = dict(x=x)
        m = None
        if mask_type:
            m = c.new_tensor(F16 if mask_type == "f16" else F32, n, T)
            ins["m"] = m
        vt = c.new_tensor(F16, n, D, H)
        ins["vt"] = vt
        p = c.soft_max_ext(x, m, 0.125, 0.0)
        p2 = c.soft_max_ext(c.scale(x, 1.0), m, 0.125, 0.0)
        return ins, [p, c.mul_mat(vt, p2)]
    mask = np.zeros((T, n), np.float32)
    for t in range(T):
        mask[t, 1 + (t * 7) % n:] = -np.inf
    feeds = dict(x=(rng.standard_normal(n * T * H) * 4).astype(np.float32), vt=rng.standard_normal(n * D * H).astype(np.float16))
    if mask_type:
        feeds["m"] = mask.astype(np.float16 if mask_type == "f16" else np.float32).ravel()
    got, want = _both(pkg, be, ref_be, build, feeds)
    assert np.isfinite(got[0]).all() and np.isfinite(got[1]).all()
    assert nmse(got[0], want[0]) < 1e-12, nmse(got[0], want[0])
    assert nmse(got[1], want[1]) < 1e-6, nmse(got[1], want[1])          # f16-rounded probabilities, f32 accumulate: the reference's MUL_MAT bar is 5e-4


@pytest.mark.parametrize("structured", [True, False])
def test_set_rows_single_element_rows_any_indices(pkg, be, structured):
    """SET_ROWS with one-element rows (the transposed V cache scatter, llama-kv-cache.cpp:1091-1109): the transposing kernel orders its work
    by a period hint taken from the reshape chain; the result must not depend on it -- the reference's index pattern (d * kv_size + cell)
    and a random permutation, with and without a reshape chain to take the hint from."""
    rng = np.random.default_rng(5)
    nv, T, kv = 1024, 40, 96
    R = nv * T
    src = rng.standard_normal(R).astype(np.float32)
    if structured:
        cells = rng.permutation(kv)[:T].astype(np.int64)
        idx = (np.arange(nv, dtype=np.int64)[None, :] * kv + cells[:, None]).ravel()
    else:
        idx = rng.permutation(nv * kv)[:R].astype(np.int64)
    for chain in (True, False):
        c = pkg.Context(be)
        cache = c.new_tensor(pkg.GGML_TYPE_F16, 1, nv * kv)
        if chain:
            v = c.new_tensor(pkg.GGML_TYPE_F32, nv, T)
            s1 = c.reshape(v, 1, R)
        else:
            v = c.new_tensor(pkg.GGML_TYPE_F32, 1, R)
            s1 = v
        ix = c.new_tensor(pkg.GGML_TYPE_I64, R)
        out = c.set_rows(cache, s1, ix)
        c.alloc()
        be.tensor_set(cache, np.zeros(nv * kv, np.float16)); be.tensor_set(v, src); be.tensor_set(ix, idx)
        be.graph_compute(c.graph())
        got = be.tensor_get(out).copy().ravel()
        c.free()
        want = np.zeros(nv * kv, np.float16)
        want[idx] = src.astype(np.float16)
        assert np.array_equal(got.view(np.uint16), want.view(np.uint16)), (structured, chain)


@pytest.mark.parametrize("T,H,HK", [(64, 32, 8), (19, 8, 4)])
def test_norm_rope_prefill_chains_vs_reference_backend(pkg, be, ref_be, T, H, HK):
    """The q chain (RMS_NORM -> MUL -> ROPE neox), the k chain + SET_ROWS into an f16 cache and the plain v SET_ROWS of a prefill ubatch at head
    size 128 (one k_norm_rope_v4 launch: 16 lanes per row) against the reference CPU backend: rope output and both cache tables."""
    rng = np.random.default_rng(T)
    D, n_ctx = 128, 256
    F32, F16, I32, I64 = pkg.GGML_TYPE_F32, pkg.GGML_TYPE_F16, pkg.GGML_TYPE_I32, pkg.GGML_TYPE_I64
    rope = dict(n_dims=D, mode=2, n_ctx_orig=40960, freq_base=1e6, freq_scale=1.0, ext_factor=0.0, attn_factor=1.0, beta_fast=32.0, beta_slow=1.0)

    def build(c):
        q = c.new_tensor(F32, D, H, T); k = c.new_tensor(F32, D, HK, T); v = c.new_tensor(F32, D, HK, T)
        qw = c.new_tensor(F32, D); kw = c.new_tensor(F32, D); pos = c.new_tensor(I32, T); idx = c.new_tensor(I64, T)
        kc = c.new_tensor(F16, HK * D, n_ctx); vc = c.new_tensor(F16, HK * D, n_ctx)
        Q = c.rope_ext(c.mul(c.rms_norm(q, 1e-6), qw), pos, None, **rope)
        K = c.rope_ext(c.mul(c.rms_norm(k, 1e-6), kw), pos, None, **rope)
        ks = c.set_rows(kc, c.view_2d(K, HK * D, T, K.nb[2], 0), idx)
        vs = c.set_rows(vc, c.view_2d(v, HK * D, T, v.nb[2], 0), idx)
        return dict(q=q, k=k, v=v, qw=qw, kw=kw, pos=pos, idx=idx, kc=kc, vc=vc), [Q, ks, vs]
    cells = rng.permutation(n_ctx)[:T].astype(np.int64)
    feeds = dict(q=rng.standard_normal(D * H * T).astype(np.float32), k=rng.standard_normal(D * HK * T).astype(np.float32), v=rng.standard_normal(D * HK * T).astype(np.float32),
                 qw=(1 + 0.1 * rng.standard_normal(D)).astype(np.float32), kw=(1 + 0.1 * rng.standard_normal(D)).astype(np.float32),
                 pos=(np.arange(T) * 3 + 5).astype(np.int32), idx=cells, kc=np.zeros(HK * D * n_ctx, np.float16), vc=np.zeros(HK * D * n_ctx, np.float16))
    got, want = _both(pkg, be, ref_be, build, feeds)
    assert nmse(got[0], want[0]) < 1e-10, nmse(got[0], want[0])
    assert nmse(got[1].astype(np.float32), want[1].astype(np.float32)) < 1e-6                     # f16 cache rows: a last-bit angle difference may flip a rounding
    assert np.array_equal(got[2].view(np.uint16), want[2].view(np.uint16))                          # v: a pure f32 -> f16 store


@pytest.mark.parametrize("wtype,xtype,M,N,K,H,HK", [("f32", "f32", 200, 130, 72, 4, 4), ("f32", "f32", 512, 200, 1000, 1, 1), ("f16", "f32", 64, 150, 1500, 6, 2),
                                                    ("f16", "f16", 300, 600, 588, 1, 1), ("f16", "f32", 1152, 100, 4304, 1, 1),
                                                    ("f16", "f32", 70, 300, 333, 4, 2), ("f16", "f16", 65, 200, 77, 3, 1), ("f16", "f32", 64, 1500, 1500, 2, 2)])
def test_any_shape_gemm_vs_reference_backend(pkg, be, ref_be, wtype, xtype, M, N, K, H, HK):
    """gemm_any.hip: MUL_MAT with more than 8 columns for F32 weights (Token2Wav, SigLip2's f32 K . Q^T with K = 72), F16 weights with a
    contraction length that is not a multiple of 32 (Whisper's V^T . P over 1500 frames, with a GQA-style broadcast), F16 x F16 (im2col
    columns of the patch embedding, K = 588) and the split form (SigLip2's n_ff 4304: F16 MFMA GEMM over 4288 columns + accumulated tail),
    against the reference CPU backend.  f32 fused multiply-adds on both sides: only the summation order differs.  (F16 weights run on the f16
    matrix cores -- k_gemm_any_h: exact f16 x f16 products, f32 accumulate; odd K / 2-byte aligned rows take its single-element loads.)"""
    rng = np.random.default_rng(M + N + K)
    F = dict(f32=pkg.GGML_TYPE_F32, f16=pkg.GGML_TYPE_F16)
    npt = dict(f32=np.float32, f16=np.float16)

    def build(c):
        w = c.new_tensor(F[wtype], K, M, HK); x = c.new_tensor(F[xtype], K, N, H)
        return dict(w=w, x=x), [c.mul_mat(w, x)]
    feeds = dict(w=(rng.standard_normal(K * M * HK) / np.sqrt(K)).astype(npt[wtype]), x=rng.standard_normal(K * N * H).astype(npt[xtype]))
    before = be.get_stat("kernels_last_graph")
    got, want = _both(pkg, be, ref_be, build, feeds)
    assert be.get_stat("kernels_last_graph") <= 3, "one launch (two for the split form, plus the activation image), not a mat-vec per 8 columns"
    assert np.isfinite(got[0]).all()
    assert nmse(got[0], want[0]) < 1e-10, nmse(got[0], want[0])


@pytest.mark.parametrize("name,M,K,N", [("q4_K", 4096, 4096, 100), ("q6_K", 1024, 4096, 128), ("q4_K", 12288, 4096, 256), ("q6_K", 4096, 12288, 72), ("q4_K", 200, 512, 65)])
def test_kquant_blocks_dequantised_in_the_gemm_staging_bit_identical_to_the_f16_image(pkg, be, name, M, K, N):
    """k_gemm_kq_glds (65 .. 256 columns against Q4_K / Q6_K weights): the blocks are de-quantised inside the GEMM's LDS staging instead of being
    read from a resident F16 image (the form taken when no image is resident; slower than the image path otherwise).  The same product with the oracle's dequantize_row_* output rounded to f16 as an F16 weight tensor goes
    through k_gemm_f16_glds with the same tiles, split and MFMA order: the two results must be BIT-identical (this pins the in-kernel nibble /
    scale arithmetic to the reference's de-quantiser), and the launch counter confirms the path."""
    import oracle.oracle_py as orc_
    from llama_cpp_omni_amd import qwen3
    rng = np.random.default_rng(M + K + N)
    ty = dict(q4_K=pkg.GGML_TYPE_Q4_K, q6_K=pkg.GGML_TYPE_Q6_K)[name]
    blocks = qwen3.random_blocks(rng, ty, M, K, std=0.05)
    w16 = orc_.dequantize(ty, blocks.view(np.uint8).reshape(M, -1), M * K).reshape(M, K).astype(np.float16)
    xv = rng.standard_normal((N, K)).astype(np.float32)
    outs = []
    be.set_option("kq_staging", 1)                              # (by default this form is the fallback for weights without a resident F16 image)
    for quant in (True, False):
        c = pkg.Context(be)
        w = c.new_tensor(ty if quant else pkg.GGML_TYPE_F16, K, M); x = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
        y = c.mul_mat(w, x)
        c.alloc()
        be.tensor_set(w, blocks if quant else w16); be.tensor_set(x, xv)
        before = be.get_stat("gemm_kq_launches")
        be.graph_compute(c.graph())
        assert (be.get_stat("gemm_kq_launches") - before == 1) == quant
        outs.append(be.tensor_get(y).copy())
        c.free()
    be.set_option("kq_staging", 0)
    assert np.isfinite(outs[0]).all()
    assert np.array_equal(outs[0].view(np.uint32), outs[1].view(np.uint32))


@pytest.mark.parametrize("n,rows", [(1024, 1500), (1152, 1024), (512, 200), (4096, 64)])
def test_layer_norm_rows_vs_reference_backend(pkg, be, ref_be, n, rows):
    """NORM (LayerNorm without affine part; the encoders' and the DiT's normalisation) through the wave-per-row kernel against the reference CPU
    backend: mean and variance in double on both sides."""
    rng = np.random.default_rng(n + rows)

    def build(c):
        x = c.new_tensor(pkg.GGML_TYPE_F32, n, rows)
        return dict(x=x), [c.norm(x, 1e-5)]
    got, want = _both(pkg, be, ref_be, build, dict(x=(rng.standard_normal(n * rows) * 2 + 0.3).astype(np.float32)))
    assert nmse(got[0], want[0]) < 1e-12, nmse(got[0], want[0])


@pytest.mark.parametrize("n,rows,bias,consumer", [(1024, 1500, True, "gemm"), (1152, 1024, True, "gemm3"), (1024, 300, False, "gemm"), (512, 200, True, "none"), (1024, 100, True, "gemm+add")])
def test_layer_norm_with_weight_and_bias_folded_in_vs_reference_backend(pkg, be, ref_be, n, rows, bias, consumer):
    """The encoders' LayerNorm as the graphs spell it -- NORM, MUL by the weight, ADD of the bias -- runs as one launch of the wave-per-row kernel
    (three f32 roundings, as the separate ops), which also leaves the f16 activation image for the MFMA GEMMs behind it: one reader (fc1: the
    f32 rows are not written at all), three readers (wq / wk / wv), no GEMM at all, and a second non-GEMM reader (f32 rows needed)."""
    rng = np.random.default_rng(n + rows)
    M = 256

    def build(c):
        x = c.new_tensor(pkg.GGML_TYPE_F32, n, rows); w = c.new_tensor(pkg.GGML_TYPE_F32, n); b = c.new_tensor(pkg.GGML_TYPE_F32, n)
        ws = [c.new_tensor(pkg.GGML_TYPE_F16, n, M) for _ in range(3)]
        y = c.mul(c.norm(x, 1e-5), w)
        if bias:
            y = c.add(y, b)
        if consumer == "none":
            outs = [y]
        elif consumer == "gemm":
            outs = [c.mul_mat(ws[0], y)]
        elif consumer == "gemm3":
            outs = [c.mul_mat(wi, y) for wi in ws]
        else:
            outs = [c.mul_mat(ws[0], y), c.add(y, x)]
        return dict(x=x, w=w, b=b, w0=ws[0], w1=ws[1], w2=ws[2]), outs
    feeds = dict(x=(rng.standard_normal(n * rows) * 2 + 0.3).astype(np.float32), w=(1 + 0.2 * rng.standard_normal(n)).astype(np.float32), b=(0.1 * rng.standard_normal(n)).astype(np.float32))
    for k in ("w0", "w1", "w2"):
        feeds[k] = (rng.standard_normal(n * M) / np.sqrt(n)).astype(np.float16)
    got, want = _both(pkg, be, ref_be, build, feeds)
    launches = be.get_stat("kernels_last_graph")
    assert launches <= {"none": 1, "gemm": 2, "gemm3": 2, "gemm+add": 3}[consumer] + (1 if rows < 128 else 0), launches       # (norm [+ f16 image] + the GEMM launch)
    for g_, w_ in zip(got, want):
        assert np.isfinite(g_).all()
        assert nmse(g_, w_) < 1e-10, nmse(g_, w_)


@pytest.mark.parametrize("op", ["gelu", "gelu_quick"])
def test_gelu_follows_the_reference_f16_tables(pkg, be, ref_be, op):
    """The reference evaluates GELU / GELU_QUICK of f32 values through f16 tables (vec.h GGML_GELU_FP16: argument rounded to f16, result an f16 value;
    GELU passes x >= 10 through and returns 0 for x <= -10): the device evaluates the same roundings instead of the f32 formula -- bit-identical to the
    reference CPU backend where the formula is well conditioned (|x| <= 2.5: > 99.9 % of the values), within an f16 ulp or two elsewhere."""
    from llama_cpp_omni_amd.ggml import UNARY
    rng = np.random.default_rng(5)
    n = 1 << 18
    xv = (rng.standard_normal(n) * 4.0).astype(np.float32)
    xv[:8] = [-10.0, 10.0, -10.000001, 9.999999, 0.0, -0.0, 70000.0, -70000.0]

    def build(c):
        x = c.new_tensor(pkg.GGML_TYPE_F32, n)
        return dict(x=x), [c.unary(x, getattr(UNARY, op.upper()))]
    got, want = _both(pkg, be, ref_be, build, dict(x=xv))
    g_, w_ = got[0].ravel(), want[0].ravel()
    fin = np.isfinite(w_)
    assert np.array_equal(np.isfinite(g_), fin)
    same = g_.view(np.uint32) == w_.view(np.uint32)
    core = fin & (np.abs(xv) <= 2.5)
    assert same[core].mean() > 0.999, same[core].mean()
    # far on the negative side 1 + tanh(u) cancels: the last f32 bit of tanhf (device libm vs glibc) decides between neighbouring tiny f16 values
    assert np.all(np.abs(g_[fin] - w_[fin]) <= 3e-7 + np.abs(w_[fin]) * 2.0 ** -9)
    assert nmse(g_[fin], w_[fin]) < 1e-12


@pytest.mark.parametrize("op,n,rows,second_reader", [("gelu", 1024, 300, False), ("gelu_quick", 4096, 130, False), ("silu", 512, 100, True), ("gelu", 1020, 70, False)])
def test_unary_between_two_gemms_emits_the_f16_image_vs_reference_backend(pkg, be, ref_be, op, n, rows, second_reader):
    """fc1 -> GELU -> fc2 of an encoder block: the unary op writes the f16 activation image of fc2's GEMM itself (and no f32 block when fc2 is its only
    reader and the next launch); a row length that is not a multiple of 8 and a second reader keep the f32 path.  Against the reference CPU backend."""
    rng = np.random.default_rng(n + rows)
    M = 192

    def build(c):
        x = c.new_tensor(pkg.GGML_TYPE_F32, n, rows); w = c.new_tensor(pkg.GGML_TYPE_F16, n, M)
        from llama_cpp_omni_amd.ggml import UNARY
        y = c.unary(x, getattr(UNARY, op.upper()))
        outs = [c.mul_mat(w, y)] + ([c.add(y, x)] if second_reader else [])
        return dict(x=x, w=w), outs
    feeds = dict(x=(rng.standard_normal(n * rows) * 1.5).astype(np.float32), w=(rng.standard_normal(n * M) / np.sqrt(n)).astype(np.float16))
    got, want = _both(pkg, be, ref_be, build, feeds)
    if n % 8 == 0 and not second_reader:
        assert be.get_stat("kernels_last_graph") <= 2 + (1 if n > 2048 else 0), be.get_stat("kernels_last_graph")      # (+ the split-K reduce of a long, lone GEMM)
    for g_, w_ in zip(got, want):
        assert np.isfinite(g_).all()
        assert nmse(g_, w_) < 1e-10, nmse(g_, w_)


@pytest.mark.parametrize("wtype,M,N,K", [("f32", 512, 200, 512), ("f32", 80, 120, 1000), ("f16", 1152, 100, 4304), ("f16", 300, 70, 333)])
def test_bias_add_folded_into_the_any_shape_gemm_vs_reference_backend(pkg, be, ref_be, wtype, M, N, K):
    """Token2Wav's linear layers: MUL_MAT with F32 weights (or F16 with an odd K) followed by the ADD of a [M] bias row -- the bias goes into the any-shape
    GEMM's epilogue (one more f32 rounding, as the separate op; the small split-K variant, the plain tiles, the f16 kernel and the split form with its
    accumulating tail).  A second product on the same x keeps its own bias; against the reference CPU backend."""
    rng = np.random.default_rng(M + N + K)
    F = dict(f32=pkg.GGML_TYPE_F32, f16=pkg.GGML_TYPE_F16); npt = dict(f32=np.float32, f16=np.float16)

    def build(c):
        w = c.new_tensor(F[wtype], K, M); w2 = c.new_tensor(F[wtype], K, M); x = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
        b = c.new_tensor(pkg.GGML_TYPE_F32, M); b2 = c.new_tensor(pkg.GGML_TYPE_F32, M)
        return dict(w=w, w2=w2, x=x, b=b, b2=b2), [c.add(c.mul_mat(w, x), b), c.add(c.mul_mat(w2, x), b2)]
    feeds = dict(w=(rng.standard_normal(K * M) / np.sqrt(K)).astype(npt[wtype]), w2=(rng.standard_normal(K * M) / np.sqrt(K)).astype(npt[wtype]),
                 x=rng.standard_normal(K * N).astype(np.float32), b=rng.standard_normal(M).astype(np.float32), b2=rng.standard_normal(M).astype(np.float32))
    got, want = _both(pkg, be, ref_be, build, feeds)
    launches = be.get_stat("kernels_last_graph")
    assert launches <= (5 if K == 4304 else 2), launches          # (split form: image + MFMA part [+ reduce] + tail)... no ADD launches
    for g_, w_ in zip(got, want):
        assert np.isfinite(g_).all()
        assert nmse(g_, w_) < 1e-10, nmse(g_, w_)


@pytest.mark.parametrize("kv_type,D,Dv,nq,nkv,H,HK", [("f16", 96, 96, 5, 70, 4, 2), ("f16", 192, 128, 3, 113, 4, 4), ("q8_0", 128, 128, 4, 96, 8, 2), ("q4_0", 64, 64, 35, 130, 4, 4),
                                                      ("bf16", 80, 80, 2, 64, 2, 1), ("f32", 40, 40, 7, 50, 2, 2), ("q8_0", 256, 256, 1, 300, 4, 1)])
def test_flash_attn_other_head_sizes_and_cache_types_vs_reference_backend(pkg, be, ref_be, kv_type, D, Dv, nq, nkv, H, HK):
    """fattn_any.hip: FLASH_ATTN_EXT at head sizes other than 64 / 128 (incl. K and V heads of different size) and with F32 / BF16 / Q8_0 / Q4_0
    K / V (a quantised KV cache), causal mask with a -inf tail, GQA broadcast, against the reference CPU backend.  The node must be accepted by
    supports_op (no hand-off to the CPU backend under the scheduler)."""
    from llama_cpp_omni_amd import encoders as E, qwen3
    rng = np.random.default_rng(D + nq + nkv)
    T = dict(f16=pkg.GGML_TYPE_F16, f32=pkg.GGML_TYPE_F32, bf16=30, q8_0=pkg.GGML_TYPE_Q8_0, q4_0=2)[kv_type]
    kf = rng.standard_normal((HK * nkv, D)).astype(np.float32); vf = rng.standard_normal((HK * nkv, Dv)).astype(np.float32)

    def enc(x):
        if kv_type == "f16":
            return x.astype(np.float16)
        if kv_type == "f32":
            return x
        if kv_type == "bf16":
            u = x.view(np.uint32); return ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)
        rows, n = x.shape
        b = x.reshape(rows, n // 32, 32)
        if kv_type == "q8_0":
            d = (np.abs(b).max(-1) / 127.0).astype(np.float16)
            q = np.rint(b / np.where(d == 0, 1, d).astype(np.float32)[..., None]).clip(-127, 127).astype(np.int8)
            out = np.zeros((rows, n // 32, 34), np.uint8)
            out[..., :2] = d[..., None].view(np.uint8).reshape(rows, n // 32, 2); out[..., 2:] = q.view(np.uint8)
            return out
        amax_i = np.abs(b).argmax(-1)
        mx = np.take_along_axis(b, amax_i[..., None], -1)[..., 0]
        d = (mx / -8.0).astype(np.float16)
        q = np.clip(np.floor(b / np.where(d == 0, 1, d).astype(np.float32)[..., None] + 8.5), 0, 15).astype(np.uint8)
        out = np.zeros((rows, n // 32, 18), np.uint8)
        out[..., :2] = d[..., None].view(np.uint8).reshape(rows, n // 32, 2); out[..., 2:] = q[..., :16] | (q[..., 16:] << 4)
        return out
    mask = np.zeros((((nq + 63) // 64) * 64, nkv), np.float32)
    for i in range(nq):
        mask[i, nkv - nq + i + 1:] = -np.inf
    res = []
    for backend in (be, ref_be):
        c = pkg.Context(backend)
        q = c.new_tensor(pkg.GGML_TYPE_F32, D, nq, H); k = c.new_tensor(T, D, nkv, HK); v = c.new_tensor(T, Dv, nkv, HK)
        m = c.new_tensor(pkg.GGML_TYPE_F16, nkv, mask.shape[0])
        y = c.flash_attn_ext(q, k, v, m, 1.0 / np.sqrt(D))
        if backend is be:
            assert not E.declined_nodes(backend, c)
        c.alloc()
        backend.tensor_set(q, rng.standard_normal(D * nq * H).astype(np.float32) if backend is be else qv)
        if backend is be:
            qv = backend.tensor_get(q).copy()
        backend.tensor_set(k, enc(kf)); backend.tensor_set(v, enc(vf)); backend.tensor_set(m, mask.astype(np.float16))
        backend.graph_compute(c.graph())
        res.append(backend.tensor_get(y).copy())
        c.free()
    assert np.isfinite(res[0]).all()
    e = nmse(res[0], res[1])
    assert e < (1e-5 if kv_type == "f16" else 1e-9), (kv_type, e)          # F16 V: the reference accumulates V in f16 (ops.cpp:8069-8083), this kernel in f32


@pytest.mark.parametrize("tname", ["q8_0", "q4_0", "bf16"])
def test_set_rows_into_a_quantised_cache_bit_exact_vs_reference_backend(pkg, be, ref_be, tname):
    """SET_ROWS f32 -> Q8_0 / Q4_0 / BF16 rows (a KV cache with -ctk / -ctv): the table bytes must equal the reference CPU backend's -- the
    from_float of each type (quantize_row_q8_0 as compiled for x86, quantize_row_q4_0_ref, fp32 -> bf16 round-to-nearest-even), incl. an all-zero
    block, ties of the largest magnitude and values that round at .5."""
    rng = np.random.default_rng(7)
    T = dict(q8_0=pkg.GGML_TYPE_Q8_0, q4_0=2, bf16=30)[tname]
    n, rows, table = 256, 37, 64
    x = rng.standard_normal((rows, n)).astype(np.float32)
    x[3, :32] = 0.0                                               # an all-zero block
    x[5, 32:64] = np.tile(np.array([1.5, -1.5], np.float32), 16)  # ties of the largest magnitude (first one decides Q4_0's sign)
    x[7, :32] = (np.arange(32, dtype=np.float32) - 16) * 0.5      # halves after scaling
    ids = rng.permutation(table)[:rows].astype(np.int64)
    res = []
    for backend in (be, ref_be):
        c = pkg.Context(backend)
        tab = c.new_tensor(T, n, table); src = c.new_tensor(pkg.GGML_TYPE_F32, n, rows); ix = c.new_tensor(pkg.GGML_TYPE_I64, rows)
        out = c.set_rows(tab, src, ix)
        c.alloc()
        nb = tab.nbytes()
        backend.tensor_set(tab, np.zeros(nb // 2, np.uint16) if tname == "bf16" else np.zeros(nb, np.uint8))
        backend.tensor_set(src, x); backend.tensor_set(ix, ids)
        backend.graph_compute(c.graph())
        res.append(np.asarray(backend.tensor_get(out)).copy().view(np.uint8))
        c.free()
    assert np.array_equal(res[0], res[1]), tname


@pytest.mark.parametrize("M,N,K,H", [(300, 1, 512, 1), (1024, 5, 4096, 1), (256, 70, 1000, 3)])
def test_bf16_weights_vs_reference_backend(pkg, be, ref_be, M, N, K, H):
    """MUL_MAT with BF16 weights (bf16 GGUFs) at one, a few and many columns: gemm_any.hip with the weights widened exactly and the activations rounded
    to bf16 (ggml_vec_dot_bf16's vec_dot_type), f32 accumulate -- against the reference CPU backend."""
    rng = np.random.default_rng(M + N + K)
    wf = (rng.standard_normal(K * M * H) / np.sqrt(K)).astype(np.float32)
    u = wf.view(np.uint32); wb = ((u + 0x7fff + ((u >> 16) & 1)) >> 16).astype(np.uint16)

    def build(c):
        w = c.new_tensor(30, K, M, H); x = c.new_tensor(pkg.GGML_TYPE_F32, K, N, H)
        return dict(w=w, x=x), [c.mul_mat(w, x)]
    got, want = _both(pkg, be, ref_be, build, dict(w=wb, x=rng.standard_normal(K * N * H).astype(np.float32)))
    assert np.isfinite(got[0]).all()
    assert nmse(got[0], want[0]) < 1e-9, nmse(got[0], want[0])
