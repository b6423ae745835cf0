"""Round-5 GPU tests (`-m gpu`), through the C-ABI.  Nothing here reads /root/reference.

* mmq_tile.hip: MUL_MAT of Q4_K weights against a prefill ubatch (> 64 columns) on the int8 matrix cores from the blocks themselves, activations quantised
  to Q8_K exactly like the reference CPU backend does (ggml_vec_dot_q4_K_q8_K, ggml-cpu/quants.c:550-623; quantize_row_q8_K, ggml-quants.c:2555-2592):
  the oracle's integer sums, f32 re-association across the 256-blocks only -- so the bar is the mat-vec bar (NMSE 1e-9), not the F16-image GEMM's 5e-4."""
import numpy as np
import pytest

from conftest import nmse
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def _run(be, c, outs, feeds):
    from test_gpu_parity import run_graph
    return run_graph(be, c, outs, feeds)


def _x(rng, N, K, scale):
    xv = (rng.standard_normal((N, K)) * scale).astype(np.float32)
    xv[N // 2, : K // 2] = 0.0                                 # all-zero Q8_K blocks: d = 0
    xv[1, 3] = -xv[1, 2]                                       # a +/- tie inside one block (the first of the two decides the sign of the scale)
    xv[N - 1, :] *= 1e-3                                       # a quiet last row
    return xv


@pytest.mark.parametrize("M,K,N", [(128, 256, 65), (130, 512, 128), (257, 1024, 129), (96, 2304, 200), (1000, 768, 300), (4096, 4096, 512), (1024, 12288, 257), (12288, 4096, 96)])
def test_mmq_tile_vs_oracle(pkg, be, M, K, N):
    """ragged row / token tiles (M, N not multiples of 128), one block (K = 256) up to 48, with and without the split along K"""
    from llama_cpp_omni_amd import qwen3
    rng = np.random.default_rng(M * 7 + K + N)
    ty = pkg.GGML_TYPE_Q4_K
    wv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
    xv = _x(rng, N, K, rng.choice([0.1, 1.0, 10.0]))
    n0 = be.get_stat("mmq_tile_launches")
    c = pkg.Context(be)
    w = c.new_tensor(ty, K, M)
    x = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
    y = c.mul_mat(w, x)
    (got,) = _run(be, c, [y], [(w, wv), (x, xv)])
    assert be.get_stat("mmq_tile_launches") - n0 >= 1, "the node did not take the tiled int8 kernel"
    want = orc.mul_mat(ty, wv.view(np.uint8).reshape(M, -1), xv)
    assert np.isfinite(got).all()
    assert nmse(got, want) < 1e-9, (M, K, N, nmse(got, want))


def test_mmq_tile_integer_sums_are_exact(pkg, be):
    """activations that quantise to THEMSELVES (integers in [-127, 127] with 127 present per block -> iscale = -1, d = -1 ... i.e. q = x, d = 1 up to sign) and
    unit block scales make the reference expression an integer: the kernel must return exactly that integer -- which pins the hi / lo scale split, the f16 mins
    MFMA and the 8 * hi + lo recombination, not only their NMSE."""
    from llama_cpp_omni_amd import qwen3
    rng = np.random.default_rng(11)
    M, K, N = 160, 1024, 96
    ty = pkg.GGML_TYPE_Q4_K
    wv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
    raw = wv.view(np.uint8).reshape(M, K // 256, 144).copy()
    raw[:, :, 0:2] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)          # d = 1
    raw[:, :, 2:4] = np.frombuffer(np.float16(1.0).tobytes(), np.uint8)          # dmin = 1
    raw[:, :, 4:16] = rng.integers(0, 256, (M, K // 256, 12), dtype=np.uint8)      # every 6-bit scale / min value
    raw[0, :, 4:16] = 0xff                                                         # the largest scales and mins (63) ...
    raw[0, :, 16:144] = 0xff                                                       # ... against the largest nibbles (15)
    wv = raw.reshape(M, -1)
    xv = rng.integers(-127, 128, (N, K)).astype(np.float32)
    xv[:, ::256] = 127.0                                                           # max of every block = +127 at its first element: iscale = -1 -> q = -x, d = -1
    xv[0, :] = 64.0; xv[0, ::256] = 127.0                                         # ... and a loud row against row 0's maximal weights (a block's integer stays below 2^24)
    c = pkg.Context(be)
    w = c.new_tensor(ty, K, M)
    x = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
    y = c.mul_mat(w, x)
    (got,) = _run(be, c, [y], [(w, wv), (x, xv)])
    # exact integer reference in int64: sum_blocks sum_j sc_j (q4 . x)_j - sum_j m_j bsum_j  (d = dmin = 1, yd = -1 and q = -x cancel)
    sc = np.zeros((M, K // 256, 8), np.int64); mn = np.zeros_like(sc)
    s = raw[:, :, 4:16].astype(np.int64)
    for j in range(4):
        sc[:, :, j] = s[:, :, j] & 63; mn[:, :, j] = s[:, :, 4 + j] & 63
        sc[:, :, 4 + j] = (s[:, :, 8 + j] & 15) | ((s[:, :, j] >> 6) << 4); mn[:, :, 4 + j] = (s[:, :, 8 + j] >> 4) | ((s[:, :, 4 + j] >> 6) << 4)
    qs = raw[:, :, 16:144].astype(np.int64).reshape(M, K // 256, 4, 32)
    q4 = np.stack([qs & 15, qs >> 4], axis=3).reshape(M, K // 256, 8, 32)          # sub-block 2q = low nibbles of chunk q, 2q + 1 = high
    xi = xv.astype(np.int64).reshape(N, K // 256, 8, 32)
    dots = np.einsum("mbjl,nbjl->nmbj", q4, xi)
    bs = xi.sum(axis=3)
    per_block = (dots * sc[None]).sum(axis=3) - np.einsum("mbj,nbj->nmb", mn, bs)   # [N, M, blocks], exact integers
    assert np.abs(per_block).max() < 2 ** 24                                       # each block's integer is exact in f32 ...
    want = np.zeros((N, M), np.float32)
    for b in range(K // 256):                                                      # ... and the kernel adds the blocks in order, one f32 rounding per add (no split along K at this shape)
        want = want + per_block[:, :, b].astype(np.float32)
    got = got.reshape(N, M)
    assert np.array_equal(got, want), np.abs(got - want).max()

def test_mmq_tile_grouped_and_residual(pkg, be):
    """wq / wk / wv sharing the activation (one launch of three matrices) and a wo-like matrix with the residual ADD behind it"""
    from llama_cpp_omni_amd import qwen3
    rng = np.random.default_rng(5)
    K, N = 2048, 200
    ty = pkg.GGML_TYPE_Q4_K
    Ms = (2048, 512, 512)
    ws = [qwen3.random_blocks(rng, ty, M, K, std=0.05) for M in Ms]
    wo = qwen3.random_blocks(rng, ty, K, K, std=0.05)
    xv = _x(rng, N, K, 1.0)
    rv = rng.standard_normal((N, K)).astype(np.float32)
    c = pkg.Context(be)
    x = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
    r = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
    wt = [c.new_tensor(ty, K, M) for M in Ms]
    wot = c.new_tensor(ty, K, K)
    ys = [c.mul_mat(w, x) for w in wt]
    yo = c.add(c.mul_mat(wot, x), r)
    got = _run(be, c, ys + [yo], [(x, xv), (r, rv), (wot, wo)] + list(zip(wt, ws)))
    for q in range(3):
        want = orc.mul_mat(ty, ws[q].view(np.uint8).reshape(Ms[q], -1), xv)
        assert nmse(got[q], want) < 1e-9, q
    want = orc.mul_mat(ty, wo.view(np.uint8).reshape(K, -1), xv) + rv
    assert nmse(got[3], want) < 1e-9


def test_mmq_tile_switch_gives_the_f16_image_path(pkg, be):
    """option mmq_tile = 0: the same node on the F16-image GEMM (the round-4 path) -- inside the reference's MUL_MAT bar, and the tiled kernel strictly closer to the oracle"""
    from llama_cpp_omni_amd import qwen3
    rng = np.random.default_rng(9)
    M, K, N = 512, 4096, 256
    ty = pkg.GGML_TYPE_Q4_K
    wv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
    xv = _x(rng, N, K, 1.0)
    want = orc.mul_mat(ty, wv.view(np.uint8).reshape(M, -1), xv)
    errs = []
    try:
        for mode in (1, 0):
            be.set_option("mmq_tile", mode)
            c = pkg.Context(be)
            w = c.new_tensor(ty, K, M)
            x = c.new_tensor(pkg.GGML_TYPE_F32, K, N)
            y = c.mul_mat(w, x)
            (got,) = _run(be, c, [y], [(w, wv), (x, xv)])
            errs.append(nmse(got, want))
    finally:
        be.set_option("mmq_tile", -1)
    assert errs[0] < 1e-9 and errs[1] < 5e-4 and errs[0] < errs[1], errs
