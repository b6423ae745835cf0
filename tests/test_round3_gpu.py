"""Round-3 GPU parity tests (through the C-ABI, `-m gpu`): the LDS-DMA loader / consumer engine of the batch-1 decode mat-vec (mmv2.hip) in every
launch form the decode step uses -- against the C oracle (the reference's integer dot over its own Q8_K image) and against the register-load
family (mmv1.hip) it replaces -- and the per-node identity check in front of every hipGraph replay.  Nothing here reads /root/reference."""
import numpy as np
import pytest

from conftest import nmse
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def _run(be, c, outs, feeds):
    from test_gpu_parity import run_graph
    return run_graph(be, c, outs, feeds)


def _silu(x):
    return x / (1.0 + np.exp(-x))


def _act(rng, K):
    x = (rng.standard_normal((1, K)) * 2.0).astype(np.float32)
    x[0, 256:512] = 0.0                                       # an all-zero Q8_K block
    x[0, 700] = -x[0, 701]                                    # a +/- tie candidate for the block maximum
    x[0, K - 256:K - 128] = 0.0
    return x


@pytest.mark.parametrize("name", ["q4_K", "q6_K"])
@pytest.mark.parametrize("K,M", [(4096, 4096), (12288, 4096), (4096, 19200)])
def test_engine_single_matrix_vs_oracle_and_register_family(pkg, be, name, K, M):
    """One matrix, plain f32 activation + residual epilogue (the wo / ffn_down launches; M = 19200: 75 rows per workgroup, more steps than the
    LDS ring holds, so slots are re-used behind the consumers).  The engine and mmv1 form the same integer sums; only the order of the f32
    partial sums over super-blocks differs."""
    from llama_cpp_omni_amd import qwen3
    ty = {"q4_K": pkg.GGML_TYPE_Q4_K, "q6_K": pkg.GGML_TYPE_Q6_K}[name]
    rng = np.random.default_rng(K + M)
    x = _act(rng, K)
    r = rng.standard_normal((1, M)).astype(np.float32)
    wv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
    wb = wv.view(np.uint8).reshape(M, -1)
    res = {}
    for mv2 in (1, 0):
        be.set_option("mv2", mv2)
        try:
            c = pkg.Context(be)
            w = c.new_tensor(ty, K, M); xt = c.new_tensor(pkg.GGML_TYPE_F32, K, 1); rt = c.new_tensor(pkg.GGML_TYPE_F32, M, 1)
            y = c.add(c.mul_mat(w, xt), rt)
            (res[mv2],) = _run(be, c, [y], [(w, wv), (xt, x), (rt, r)])
            assert be.get_stat("kernels_last_graph") == 1
        finally:
            be.set_option("mv2", 1)
    want = orc.mul_mat(ty, wb, x) + r
    assert nmse(res[1], want) < 1e-9, nmse(res[1], want)
    assert nmse(res[1], res[0]) < 1e-12, nmse(res[1], res[0])


def test_engine_gate_up_pair_with_norm_and_swiglu(pkg, be):
    """The dominant launch of the decode step: RMS_NORM + MUL folded into the prologue (the sum of squares in double, the scale computed once per
    workgroup), ffn_gate and ffn_up streamed as one pair of rings, SwiGLU in the epilogue -- 12288 x 4096 Q4_K twice, one launch."""
    from llama_cpp_omni_amd import qwen3
    K, M = 4096, 12288
    rng = np.random.default_rng(5)
    x = _act(rng, K)
    nw = rng.standard_normal(K).astype(np.float32)
    ty = pkg.GGML_TYPE_Q4_K
    gv = qwen3.random_blocks(rng, ty, M, K, std=0.05); uv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
    res = {}
    for mv2 in (1, 0):
        be.set_option("mv2", mv2)
        try:
            c = pkg.Context(be)
            wg = c.new_tensor(ty, K, M); wu = c.new_tensor(ty, K, M); xt = c.new_tensor(pkg.GGML_TYPE_F32, K, 1); nt = c.new_tensor(pkg.GGML_TYPE_F32, K)
            xn = c.mul(c.rms_norm(xt, 1e-6), nt)
            up = c.mul_mat(wu, xn); gate = c.mul_mat(wg, xn)
            y = c.swiglu_split(gate, up)
            (res[mv2],) = _run(be, c, [y], [(wg, gv), (wu, uv), (xt, x), (nt, nw)])
            assert be.get_stat("kernels_last_graph") == 1
        finally:
            be.set_option("mv2", 1)
    xn_ref = (orc.rms_norm(x, 1e-6) * nw).astype(np.float32)
    g = orc.mul_mat(ty, gv.view(np.uint8).reshape(M, -1), xn_ref); u = orc.mul_mat(ty, uv.view(np.uint8).reshape(M, -1), xn_ref)
    want = (_silu(g.astype(np.float64)) * u.astype(np.float64)).astype(np.float32)
    assert nmse(res[1], want) < 1e-9, nmse(res[1], want)
    assert nmse(res[1], res[0]) < 1e-12, nmse(res[1], res[0])


@pytest.mark.parametrize("vtype", ["q4_K", "q6_K"])
def test_engine_grouped_qkv_launch(pkg, be, vtype):
    """wq / wk / wv on one normalised activation: three matrices in one launch, workgroups split between them by bytes; with the Q4_K_M map
    attn_v is Q6_K on half the layers, so one launch mixes the two weight formats (and the two ring geometries)."""
    from llama_cpp_omni_amd import qwen3
    K = 4096
    rows = [4096, 1024, 1024]
    types = [pkg.GGML_TYPE_Q4_K, pkg.GGML_TYPE_Q4_K, pkg.GGML_TYPE_Q6_K if vtype == "q6_K" else pkg.GGML_TYPE_Q4_K]
    rng = np.random.default_rng(9)
    x = _act(rng, K)
    nw = rng.standard_normal(K).astype(np.float32)
    wvs = [qwen3.random_blocks(rng, t, m, K, std=0.05) for t, m in zip(types, rows)]
    res = {}
    for mv2 in (1, 0):
        be.set_option("mv2", mv2)
        try:
            c = pkg.Context(be)
            ws = [c.new_tensor(t, K, m) for t, m in zip(types, rows)]
            xt = c.new_tensor(pkg.GGML_TYPE_F32, K, 1); nt = c.new_tensor(pkg.GGML_TYPE_F32, K)
            xn = c.mul(c.rms_norm(xt, 1e-6), nt)
            ys = [c.mul_mat(w, xn) for w in ws]
            res[mv2] = _run(be, c, ys, list(zip(ws, wvs)) + [(xt, x), (nt, nw)])
            assert be.get_stat("kernels_last_graph") == 1
        finally:
            be.set_option("mv2", 1)
    xn_ref = (orc.rms_norm(x, 1e-6) * nw).astype(np.float32)
    for i, (t, m) in enumerate(zip(types, rows)):
        want = orc.mul_mat(t, wvs[i].view(np.uint8).reshape(m, -1), xn_ref)
        assert nmse(res[1][i], want) < 1e-9, (i, nmse(res[1][i], want))
        assert nmse(res[1][i], res[0][i]) < 1e-12, i


def test_replay_is_guarded_by_the_per_node_record_not_by_the_fingerprint(pkg, be):
    """Every cgraph is given the SAME fingerprint (test hook).  Two graphs of identical structure but different tensors are alive together; A is
    submitted until it is captured and replays, then B is submitted: a fingerprint-keyed replay would launch A's capture (and write A's
    output); the per-node record (data pointers, op, type, shape, strides, op_params, source pointers, readers outside the graph) refuses it,
    B runs and is captured on its own, and both keep giving their own results."""
    rng = np.random.default_rng(1)
    n = 1024

    def chain(scale):
        c = pkg.Context(be)
        x = c.new_tensor(pkg.GGML_TYPE_F32, n, 1); b = c.new_tensor(pkg.GGML_TYPE_F32, n, 1)
        y = x
        for _ in range(12):
            y = c.add(y, b)
        c.alloc()
        xv = rng.standard_normal((1, n)).astype(np.float32); bv = (rng.standard_normal((1, n)) * scale).astype(np.float32)
        be.tensor_set(x, xv); be.tensor_set(b, bv)
        want = xv.copy()
        for _ in range(12):
            want = want + bv
        return c, c.graph(), y, want

    be.set_option("graphs", 1)
    be.set_option("fp_collide", 1)
    try:
        ca, ga, ya, wa = chain(1.0)
        cb, gb, yb, wb = chain(3.0)
        mis0, rep0 = be.get_stat("graph_fp_mismatch"), be.get_stat("graph_replays")
        for _ in range(4):
            be.graph_compute(ga)
        assert np.array_equal(be.tensor_get(ya).ravel(), wa.ravel())
        assert be.get_stat("graph_replays") > rep0                       # A is captured and replays
        for _ in range(4):
            be.graph_compute(gb)
        assert np.array_equal(be.tensor_get(yb).ravel(), wb.ravel())                     # B ran as B
        assert be.get_stat("graph_fp_mismatch") > mis0                   # ... after A's capture was refused for it
        be.graph_compute(ga)
        assert np.array_equal(be.tensor_get(ya).ravel(), wa.ravel())
        ca.free(); cb.free()
    finally:
        be.set_option("fp_collide", 0)


def test_batched_small_uploads_keep_stream_order(pkg, be):
    """set_tensor_async of small inputs is staged and written by one launch in front of the next stream work: later writes win over earlier
    ones whatever their size (staged or direct copy), reads see them, and the result equals the one-copy-per-call path."""
    rng = np.random.default_rng(2)
    n_big = 64 * 1024                                            # 256 KB: above the staging limit -> direct copy
    c = pkg.Context(be)
    t = c.new_tensor(pkg.GGML_TYPE_F32, n_big, 1)
    s = c.new_tensor(pkg.GGML_TYPE_F32, 1000, 1)
    y = c.add(s, s)
    c.alloc()
    small = [rng.standard_normal(1000).astype(np.float32) for _ in range(3)]
    big = [rng.standard_normal(n_big).astype(np.float32) for _ in range(2)]
    out = np.empty(n_big, np.float32)
    for batch in (1, 0):
        be.set_option("batch_uploads", batch)
        try:
            be.tensor_set_async(t, big[0])
            be.tensor_set_async(t, small[0])                    # staged, after the direct copy: overwrites the first 1000 values
            be.tensor_get_async(t, out); be.synchronize()
            assert np.array_equal(out[:1000], small[0]) and np.array_equal(out[1000:], big[0][1000:])
            be.tensor_set_async(t, small[1])                    # staged ...
            be.tensor_set_async(t, big[1])                      # ... then a direct copy over it: the direct copy must come second
            be.tensor_get_async(t, out); be.synchronize()
            assert np.array_equal(out, big[1])
            for k in range(70):                                 # more entries than one staging table holds
                be.tensor_set_async(s, small[k % 3])
            be.graph_compute(c.graph())
            assert np.array_equal(be.tensor_get(y).ravel(), small[69 % 3] * 2)
        finally:
            be.set_option("batch_uploads", 1)
    c.free()
