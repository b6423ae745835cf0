"""Token2Wav row: the ops its graphs add to the decoder's set, each against the REFERENCE CPU backend on the same
graph (oracle/_ref), and blocks of the flow-matching DiT / HiFT vocoder at their real shapes, node for node as
tools/omni/token2wav/token2wav-impl.cpp emits them (llama.cpp-omni_amd/token2wav.py).  Token2Wav drives one backend with
ggml_backend_graph_compute directly (no scheduler), so every node must be accepted by supports_op -- asserted here."""
import numpy as np
import pytest

from conftest import nmse

pytestmark = pytest.mark.gpu


def _run_both(pkg, be, ref_be, build, feeds_fn, check_declined=True):
    """build(c) -> (inputs dict name -> tensor, outputs list); feeds_fn(rng, name, tensor) -> numpy array"""
    from llama_cpp_omni_amd import encoders as E
    res = []
    for backend in (be, ref_be):
        c = pkg.Context(backend)
        ins, outs = build(c)
        if backend is be and check_declined:
            bad = E.declined_nodes(backend, c)
            assert not bad, f"supports_op declined: {bad}"
        c.alloc()
        rng = np.random.default_rng(5)
        for name, t in ins.items():
            backend.tensor_set(t, feeds_fn(rng, name, t))
        backend.graph_compute(c.graph())
        res.append([backend.tensor_get(o).copy() for o in outs])
        c.free()
    return res


def _randn(rng, name, t, scale=1.0):
    n = t.nelements()
    if t.type == 26:                                           # i32
        return rng.integers(-50, 50, n).astype(np.int32)
    if t.type == 1:
        return (rng.standard_normal(n) * scale).astype(np.float16)
    return (rng.standard_normal(n) * scale).astype(np.float32)


OPS = ["sqr", "sqrt", "log", "sin", "cos", "clamp", "leaky_relu", "sum_rows", "repeat", "repeat_f16", "concat0", "concat1", "concat2", "concat3", "concat_i32",
       "concat_strided", "pad", "pad_reflect", "arange", "timestep_even", "timestep_odd", "conv_transpose_f16", "conv_transpose_f32", "cast_i32", "cast_f32", "scale_bias"]


@pytest.mark.parametrize("op", OPS)
def test_t2w_op_vs_reference_backend(pkg, be, ref_be, op):
    F32, F16, I32 = pkg.GGML_TYPE_F32, pkg.GGML_TYPE_F16, pkg.GGML_TYPE_I32
    exact = True

    def build(c):
        nonlocal exact
        if op in ("sqr", "sqrt", "log", "sin", "cos", "leaky_relu"):
            x = c.new_tensor(F32, 77, 5, 3)
            fn = dict(sqr=c.sqr, sqrt=c.sqrt, log=c.log, sin=c.sin, cos=c.cos, leaky_relu=lambda t: c.leaky_relu(t, 0.1))[op]
            exact = op in ("sqr", "sqrt", "leaky_relu")        # libm vs device transcendentals differ in the last bits
            return {"x_pos" if op in ("sqrt", "log") else "x": x}, [fn(x)]
        if op == "clamp":
            x = c.new_tensor(F32, 100, 7)
            return {"x": x}, [c.clamp(c.scale(x, 2.0), -1.5, 0.75)]
        if op == "scale_bias":
            x = c.new_tensor(F32, 100, 7)
            return {"x": x}, [c.scale(x, 1.0, -0.25)]
        if op == "sum_rows":
            x = c.new_tensor(F32, 1000, 6, 2)
            xt = c.permute(x, 0, 2, 1, 3)                      # rows with permuted outer strides
            return {"x": x}, [c.sum_rows(x), c.sum_rows(xt)]
        if op == "repeat":
            x = c.new_tensor(F32, 3, 1, 2)
            return {"x": x}, [c.repeat_4d(x, 9, 5, 4, 2)]
        if op == "repeat_f16":
            x = c.new_tensor(F16, 4, 3)
            return {"x": x}, [c.repeat_4d(x, 8, 6, 2, 1)]
        if op.startswith("concat") and op[-1].isdigit():
            d = int(op[-1])
            ne_a, ne_b = [5, 4, 3, 2], [5, 4, 3, 2]
            ne_b[d] = 7
            a, b = c.new_tensor(F32, *ne_a), c.new_tensor(F32, *ne_b)
            return {"a": a, "b": b}, [c.concat(a, b, d)]
        if op == "concat_i32":
            a, b = c.new_tensor(I32, 6, 3), c.new_tensor(I32, 6, 2)
            return {"a": a, "b": b}, [c.concat(a, b, 1)]
        if op == "concat_strided":
            a, b = c.new_tensor(F32, 6, 5, 2), c.new_tensor(F32, 5, 4, 2)
            return {"a": a, "b": b}, [c.concat(c.permute(a, 1, 0, 2, 3), b, 1)]      # [5, 6, 2] (non-contiguous) ++ [5, 4, 2]
        if op == "pad":
            x = c.new_tensor(F32, 11, 5, 3, 2)
            return {"x": x}, [c.pad_ext(x, 2, 0, 0, 0, 0, 0, 0, 0), c.pad_ext(x, 1, 3, 0, 2, 1, 0, 0, 1), c.pad_ext(c.permute(x, 0, 2, 1, 3), 0, 1, 0, 1, 0, 0, 0, 0)]
        if op == "pad_reflect":
            x = c.new_tensor(F32, 40, 3, 2)
            return {"x": x}, [c.pad_reflect_1d(x, 8, 8), c.pad_reflect_1d(x, 1, 0), c.pad_reflect_1d(x, 0, 5)]
        if op == "arange":
            return {}, [c.arange(0.0, 37.0, 1.0), c.arange(1.0, 2.0, 1.0), c.arange(-3.0, 4.1, 0.7)]
        if op.startswith("timestep"):
            exact = False
            t = c.new_tensor(F32, 5)
            return {"t": t}, [c.timestep_embedding(c.scale(t, 100.0), 256 if op.endswith("even") else 33, 10000)]
        if op.startswith("conv_transpose"):
            exact = False
            w = c.new_tensor(F16 if op.endswith("f16") else F32, 16, 24, 40)          # [K, Cout, Cin]
            x = c.new_tensor(F32, 50, 40)
            w2 = c.new_tensor(F16 if op.endswith("f16") else F32, 3, 5, 40)
            return {"w": w, "x": x, "w2": w2}, [c.conv_transpose_1d(w, x, 8), c.conv_transpose_1d(w2, x, 1), c.conv_transpose_1d(w2, x, 5)]
        if op == "cast_i32":
            x = c.new_tensor(F32, 33, 4)
            return {"x": x}, [c.cast(c.scale(x, 10.0), I32)]
        if op == "cast_f32":
            x = c.new_tensor(I32, 33, 4)
            return {"x": x}, [c.cast(x, F32)]
        raise AssertionError(op)

    def feeds(rng, name, t):
        v = _randn(rng, name, t)
        if name == "x_pos":
            v = np.abs(v) + 0.01
        return v

    got, want = _run_both(pkg, be, ref_be, build, feeds)
    for g, w in zip(got, want):
        if op.startswith("timestep_odd"):                      # the reference leaves the pad element of an odd width unwritten
            g, w = g.reshape(5, 34)[:, :33], w.reshape(5, 34)[:, :33]
        assert np.isfinite(np.asarray(g, np.float64)).all()
        if exact:
            assert np.array_equal(g.view(np.uint8), w.view(np.uint8)), op
        else:
            assert nmse(g, w) < 1e-10, (op, nmse(g, w))


def _fill_scaled(rng, name, t):
    n = t.nelements()
    ne = [d for d in t.ne]
    if name.endswith("_w") and len([d for d in ne if d > 1]) >= 2:       # matrices / conv kernels ~ 1 / sqrt(fan_in)
        fan = ne[0] * (ne[1] if len([d for d in ne if d > 1]) == 3 else 1)
        return (rng.standard_normal(n) / np.sqrt(fan)).astype(np.float32)
    if name.endswith("_w"):
        return (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)   # norm gains
    if name.endswith("_b"):
        return (0.1 * rng.standard_normal(n)).astype(np.float32)
    if name in ("a1", "a2"):
        return (0.5 + np.abs(rng.standard_normal(n))).astype(np.float32) # snake alphas
    return rng.standard_normal(n).astype(np.float32)


@pytest.mark.parametrize("which", ["dit_block", "timestep", "hift_stage", "istft", "speaker_norm", "length_mask"])
def test_t2w_blocks_at_real_shape_vs_reference_backend(pkg, be, ref_be, which):
    """DiT block at hidden 512 / 8 x 64 heads / 200 frames, HiFT upsample stage 512 -> 256 channels x8 over 120 frames with its snake
    residual block, the ISTFT head (hop 4, n_fft 16) over 2000 frames, speaker normalisation, length masks."""
    from llama_cpp_omni_amd import token2wav as T
    F32 = pkg.GGML_TYPE_F32

    def build(c):
        if which == "dit_block":
            W = T.dit_weights(c, T.DIT)
            x, cond, out = T.dit_block(c, T.DIT, W, 200)
            return dict(W, x=x, cond=cond), [out]
        if which == "timestep":
            hp = T.DIT
            W = dict(t1_w=c.new_tensor(F32, hp["freq_dim"], hp["hidden"]), t1_b=c.new_tensor(F32, hp["hidden"]), t2_w=c.new_tensor(F32, hp["hidden"], hp["hidden"]), t2_b=c.new_tensor(F32, hp["hidden"]))
            t, out = T.timestep_embedder(c, hp, W["t1_w"], W["t1_b"], W["t2_w"], W["t2_b"], 10)
            return dict(W, t01=t), [out]
        if which == "hift_stage":
            W = T.hift_weights(c, T.HIFT)
            x, out = T.hift_upsample_stage(c, T.HIFT, W, 120)
            return dict(W, x=x), [out]
        if which == "istft":
            ins, outs = T.istft_head(c, T.HIFT, 2000)
            return ins, outs
        if which == "speaker_norm":
            x, eps, out = T.speaker_norm(c, 192, 3)
            return dict(x=x, eps_pos=eps), [out]
        if which == "length_mask":
            lengths, valid, pad = T.length_mask(c, 300, 4)
            return dict(lengths=lengths), [valid, pad]
        raise AssertionError(which)

    def feeds(rng, name, t):
        if name == "t01":
            return rng.random(t.nelements()).astype(np.float32)
        if name == "eps_pos":
            return np.full(t.nelements(), 1e-12, np.float32)
        if name == "lengths":
            return np.array([300, 17, 0, 123], np.float32)
        if name == "wsq":
            return (0.2 + rng.random(t.nelements())).astype(np.float32)
        if name == "up_w":                                     # [K, Cout, Cin]: fan-in = Cin * K / stride
            return (rng.standard_normal(t.nelements()) / np.sqrt(t.ne[2] * t.ne[0] / 8)).astype(np.float32)
        return _fill_scaled(rng, name, t)

    got, want = _run_both(pkg, be, ref_be, build, feeds)
    for g, w in zip(got, want):
        assert np.isfinite(np.asarray(g, np.float64)).all()
        if g.dtype == np.int32 or which == "length_mask":
            assert np.array_equal(g, w)
        else:
            e = nmse(g, w)
            print(which, "NMSE vs the reference CPU backend:", e)
            # f32 weights: > 8 columns run the MFMA GEMM on f16-rounded operands (the reference's own MUL_MAT bar: 5e-4)
            assert e < (5e-4 if which in ("dit_block", "hift_stage", "timestep") else 1e-10), (which, e)


def test_dit_block_graph_replay_matches_the_eager_run(pkg, be):
    """A Token2Wav DiT block is launch-bound (59 launches of a few microseconds), so the backend replays it as a hipGraph from the third submission: the
    replayed results -- with the fused LayerNorm / bias epilogues / f16 images decided at capture time -- are bit-identical to the first, eager run."""
    from llama_cpp_omni_amd import token2wav as T
    c = pkg.Context(be)
    W = T.dit_weights(c, T.DIT)
    x, cond, out = T.dit_block(c, T.DIT, W, 200)
    ins = dict(W, x=x, cond=cond)
    c.alloc()
    rng = np.random.default_rng(5)
    for name, t in ins.items():
        be.tensor_set(t, _fill_scaled(rng, name, t))
    g = c.graph()
    before = be.get_stat("graph_replays")
    runs = []
    for _ in range(5):
        be.graph_compute(g)
        runs.append(be.tensor_get(out).copy())
    assert be.get_stat("graph_replays") > before, "the block did not replay"
    for r in runs[1:]:
        assert np.array_equal(r.view(np.uint32), runs[0].view(np.uint32))
    c.free()


@pytest.mark.parametrize("C,T,B", [(512, 56, 2), (64, 7, 1)])
def test_lazy_cache_copies_become_real_when_somebody_else_reads_them(pkg, be, ref_be, C, T, B):
    """graph_exec_t2w.cpp lazy_try_register: the causal convolution's two copies of the cached frames (cache_in = CONT(view of the packed cache), cache_tcb =
    CONT(PERMUTE(cache_in)); token2wav-impl.cpp:952-957) are not run when they are met.  This graph has the shape that makes them lazy but NOT the convolution
    behind it, so every reader is an ordinary node: the CONCAT with the transposed x must get the transposed frames (read in place through swapped strides), and a late reader of
    cache_in -- behind a CPY that overwrites those very frames in the cache -- must still see the OLD frames (materialised at the deadline, before the CPY runs).
    Copies only: bit-exact against the reference CPU backend."""
    F32 = pkg.GGML_TYPE_F32
    P, slots, slot = 2, 5, 3

    def build(c):
        cache = c.new_tensor(F32, C, P * slots, B)
        x = c.new_tensor(F32, C, T, B)
        nb1, nb2 = C * 4, C * P * slots * 4
        cv = c.view_3d(cache, C, P, B, nb1, nb2, slot * P * nb1)
        cc = c.cont(cv)                                         # lazy (A)
        ct = c.cont(c.permute(cc, 1, 0, 2, 3))                  # lazy (B)
        xt = c.cont(c.permute(x, 1, 0, 2, 3))
        cat = c.concat(ct, xt, 0)                               # ordinary reader of cache_tcb
        y1 = c.cont(cat)
        upd = c.cpy(c.view_3d(x, C, P, B, x.t.nb[1], x.t.nb[2], 0), c.view_3d(cache, C, P, B, nb1, nb2, slot * P * nb1))     # the cache slot is overwritten ...
        y2 = c.cont(c.concat(cc, x, 1))                         # ... before cache_in's other reader runs
        return {"cache": cache, "x": x}, [y1, y2, upd]

    n0, m0 = be.get_stat("lazy_conts"), be.get_stat("lazy_conts_materialised")
    got, want = _run_both(pkg, be, ref_be, build, lambda rng, name, t: _randn(rng, name, t), check_declined=False)
    # both copies were deferred; the CONCAT reads cache_tcb's source through swapped strides (compute_node, CONCAT), cache_in is made real at the deadline
    assert be.get_stat("lazy_conts") - n0 == 2 and be.get_stat("lazy_conts_materialised") - m0 == 1
    for g, w in zip(got, want):
        assert np.array_equal(g, w)


@pytest.mark.parametrize("Tq,Tk", [(56, 206), (50, 200)])
def test_attention_chain_reads_its_operands_through_the_permuted_views(pkg, be, ref_be, Tq, Tk):
    """graph_exec_t2w.cpp lazy_try_register, case C: the DiT attention as token2wav-impl.cpp:245-291 / :470-497 spells it -- Q, K, V [D, H, T, B] each flattened by CONT(PERMUTE),
    V transposed by a second CONT(PERMUTE), K.Q -> SCALE -> SOFT_MAX -> V^T.P -> merge.  The four copies are not run: k_attn_f32 reads the permuted 4-D views (two-level
    head-batch strides; V itself, keys a row apart).  206 keys: rows of V^T would be 824 bytes, the odd-length path.  Bar: the fused chain's f32 arithmetic against the reference
    CPU backend's separate nodes, NMSE <= 1e-10; the copies must have been deferred and only the merge's reader-less CONT may be real."""
    F32 = pkg.GGML_TYPE_F32
    D, H, B = 64, 8, 2

    def build(c):
        q, k, v = c.new_tensor(F32, D, H, Tq, B), c.new_tensor(F32, D, H, Tk, B), c.new_tensor(F32, D, H, Tk, B)

        def flat(t, T):
            return c.reshape(c.cont(c.permute(t, 0, 2, 1, 3)), D, T, H * B)
        qf, kf = flat(q, Tq), flat(k, Tk)
        vf = c.reshape(c.cont(c.permute(flat(v, Tk), 1, 0, 2, 3)), Tk, D, H * B)
        scores = c.scale(c.mul_mat(kf, qf), 1.0 / 8.0)
        probs = c.soft_max_ext(scores, None, 1.0, 0.0)
        ctx = c.mul_mat(vf, probs)                                                        # [D, Tq, H B]
        merged = c.cont(c.permute(c.reshape(ctx, D, Tq, H, B), 0, 2, 1, 3))               # [D, H, Tq, B]
        return {"q": q, "k": k, "v": v}, [merged]

    n0, m0 = be.get_stat("lazy_conts"), be.get_stat("lazy_conts_materialised")
    got, want = _run_both(pkg, be, ref_be, build, lambda rng, name, t: _randn(rng, name, t))
    assert be.get_stat("lazy_conts") - n0 == 4 and be.get_stat("lazy_conts_materialised") - m0 == 0, (be.get_stat("lazy_conts") - n0, be.get_stat("lazy_conts_materialised") - m0)
    e = nmse(got[0], want[0])
    print("attention chain through lazy operands, NMSE vs the reference CPU backend:", e)
    assert e <= 1e-10, e
