"""Round-4 GPU tests (`-m gpu`).  Nothing here reads /root/reference.

* the persistent role-split decode engine LAB (tools/mmv3_engine.hip, tools/mmv3_lab.hip: the four mat-vec stages of a decode layer as ONE launch
  whose weight stream runs across the stage boundaries) produces every stage's output bit for bit as the product's four k_mv2 launches do --
  the evidence behind profiles/r04_engine_lab.txt.  The lab is compiled on the box (hipcc is part of the image) because it is not part of the
  product library: it measured slower than the launch form ("Round 4")."""
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_lab(tmp):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    exe = os.path.join(tmp, "mmv3_lab")
    # (-DMV2_BLOCK_LANE=0: the round-4 engine shares the sub-block-pair Q4_K consumer with the launch form it is compared with bit for bit; the product's launches
    #  moved to the block-per-lane consumer in round 6 -- same integers, another order of the float sums, parity with the oracle in test_round3_gpu.py)
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-mllvm", "-amdgpu-kernarg-preload-count=14", "-Wno-inline-asm", "-DMV2_BLOCK_LANE=0",
           os.path.join(ROOT, "tools", "mmv3_lab.hip"), "-o", exe]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:]
    return exe


@pytest.mark.parametrize("stages", [(0, 3), (1, 2)])
def test_persistent_engine_lab_is_bit_identical_to_the_launch_form(tmp_path, stages):
    """wo + resid -> [norm] gate / up + SwiGLU -> down + resid -> [norm] wq / wk / wv, Q4_K and Q6_K layer variants, three weight sets each: every
    element of x2, h, x3, q, k, v equal between one k_mv3 launch and the k_mv2 launches; no bounded wait of the engine may give up."""
    exe = _build_lab(str(tmp_path))
    r = subprocess.run([exe, str(stages[0]), str(stages[1])], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    out = r.stdout
    assert r.returncode == 0, out[-3000:]
    assert "GAVE UP" not in out and "MISMATCH" not in out and "RESULTS DIFFER" not in out, out[-3000:]
    assert out.count("results identical") == 2, out[-3000:]                   # both layer variants
    assert out.count(" identical (0 /") == 2 * 3 * (stages[1] - stages[0] + 1), out[-3000:]


# ------------------------------------------------------------------------------------------------ Q8_0 on the LDS-DMA engine (the 8B LLM of BASELINE configs[4])
import numpy as np

from conftest import nmse
from oracle import oracle_py as orc


def _run(be, c, outs, feeds):
    from test_gpu_parity import run_graph
    return run_graph(be, c, outs, feeds)


def _act(rng, K):
    x = (rng.standard_normal((1, K)) * 2.0).astype(np.float32)
    x[0, 256:320] = 0.0                                       # two all-zero Q8_0 blocks (amax = 0: d = 0, id = 0)
    x[0, 700] = -x[0, 701]
    return x


def _silu(x):
    return x / (1.0 + np.exp(-x))


@pytest.mark.parametrize("K,M", [(4096, 4096), (12288, 4096), (4096, 70000)])
def test_q8_0_engine_single_matrix_vs_oracle_and_row_kernels(pkg, be, K, M):
    """One Q8_0 matrix + residual epilogue (wo / ffn_down of the Q8_0 8B model; 70000 rows: more than k_mv1q takes, ring slots re-used many times).
    mv2 = 1: k_mv2<4, NIT, false>; mv2 = 0: the round-1 / round-2 Q8_0 kernels.  Both form the reference's integer sums (ggml_vec_dot_q8_0_q8_0,
    ggml-cpu/quants.c:305-333) over the reference's Q8_0 activation blocks; only the order of the f32 sums differs."""
    from llama_cpp_omni_amd import qwen3
    ty = pkg.GGML_TYPE_Q8_0
    rng = np.random.default_rng(K + M)
    x = _act(rng, K)
    r = rng.standard_normal((1, M)).astype(np.float32)
    wv = qwen3.random_blocks(rng, ty, M, K, std=0.05)
    res = {}
    for mv2 in (1, 0):
        be.set_option("mv2", mv2)
        try:
            c = pkg.Context(be)
            w = c.new_tensor(ty, K, M); xt = c.new_tensor(pkg.GGML_TYPE_F32, K, 1); rt = c.new_tensor(pkg.GGML_TYPE_F32, M, 1)
            y = c.add(c.mul_mat(w, xt), rt)
            (res[mv2],) = _run(be, c, [y], [(w, wv), (xt, x), (rt, r)])
            if mv2:                                                     # (the engine stages at most 256 residual rows per workgroup: beyond 65536 rows the ADD stays its own launch)
                assert be.get_stat("kernels_last_graph") == (1 if M <= 65536 else 2)
        finally:
            be.set_option("mv2", 1)
    want = orc.mul_mat(ty, wv.view(np.uint8).reshape(M, -1), x) + r
    assert nmse(res[1], want) < 1e-9, nmse(res[1], want)
    assert nmse(res[1], res[0]) < 1e-11, nmse(res[1], res[0])


def test_q8_0_engine_gate_up_pair_with_norm_and_swiglu(pkg, be):
    from llama_cpp_omni_amd import qwen3
    K, M = 4096, 12288
    ty = pkg.GGML_TYPE_Q8_0
    rng = np.random.default_rng(15)
    x = _act(rng, K)
    nw = rng.standard_normal(K).astype(np.float32)
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
    assert nmse(res[1], res[0]) < 1e-11, nmse(res[1], res[0])


def test_q8_0_engine_grouped_qkv_launch(pkg, be):
    from llama_cpp_omni_amd import qwen3
    K = 4096
    rows = [4096, 1024, 1024]
    ty = pkg.GGML_TYPE_Q8_0
    rng = np.random.default_rng(19)
    x = _act(rng, K)
    nw = rng.standard_normal(K).astype(np.float32)
    wvs = [qwen3.random_blocks(rng, ty, m, K, std=0.05) for m in rows]
    res = {}
    for mv2 in (1, 0):
        be.set_option("mv2", mv2)
        try:
            c = pkg.Context(be)
            ws = [c.new_tensor(ty, K, m) for m in rows]
            xt = c.new_tensor(pkg.GGML_TYPE_F32, K, 1); nt = c.new_tensor(pkg.GGML_TYPE_F32, K)
            xn = c.mul(c.rms_norm(xt, 1e-6), nt)
            ys = [c.mul_mat(w, xn) for w in ws]
            res[mv2] = _run(be, c, ys, list(zip(ws, wvs)) + [(xt, x), (nt, nw)])
            assert be.get_stat("kernels_last_graph") == 1
        finally:
            be.set_option("mv2", 1)
    xn_ref = (orc.rms_norm(x, 1e-6) * nw).astype(np.float32)
    for i, m in enumerate(rows):
        want = orc.mul_mat(ty, wvs[i].view(np.uint8).reshape(m, -1), xn_ref)
        assert nmse(res[1][i], want) < 1e-9, (i, nmse(res[1][i], want))
        assert nmse(res[1][i], res[0][i]) < 1e-11, i


def test_8b_q8_0_greedy_ids_identical_through_libllama(tmp_path):
    """BASELINE configs[4]: the omni pipeline ships its 8B LLM as Q8_0 (tools/omni/convert/run_convert.sh:67-70).  The 36-layer Qwen3-8B GGUF with EVERY matrix
    Q8_0 (8.7 GB), separated-logits fixture as in test_round3_gpu.py, decoded by the reference's libllama on the plug-in and on the reference CPU backend:
    128 / 128 free-running greedy ids identical, flash-attention off and on; wq / wk / wv, wo, ffn_gate / ffn_up, ffn_down (K = 12288) and the 151936-row
    lm-head all run on the LDS-DMA engine's Q8_0 body (k_mv2<4, ..>)."""
    import sys
    import shutil
    import test_round3_gpu as r3
    r3._need_ref(tmp_path)
    if shutil.disk_usage(str(tmp_path)).free < 12e9:
        pytest.skip("needs 9 GB of scratch disk for the synthetic Q8_0 8B GGUF")
    gguf = str(tmp_path / "q8b_q80.gguf")
    gen = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_gguf.py"), "--config", "8b", "--types", "q8_0", "-o", gguf, "--n-ctx", "4096",
                          "--separated", "160"], check=True, timeout=1800, capture_output=True, text=True)
    start = int(gen.stdout.split("start token")[1].split()[0])
    threads = max(4, min(48, len(os.sched_getaffinity(0)) // 2))
    n = 128
    try:
        for fa in (0, 1):
            ids_cpu = r3._bench_min(gguf, 0, fa, n, threads, ["--start-token", str(start), "--dump-all-logits", str(tmp_path / "c.bin")], False)
            ids_gpu = r3._bench_min(gguf, 99, fa, n, threads, ["--start-token", str(start), "--dump-all-logits", str(tmp_path / "g.bin")], True)
            assert len(set(ids_cpu)) == n, "the fixture's cycle is longer than the run"
            assert ids_gpu == ids_cpu, (fa, [i for i in range(n) if ids_gpu[i] != ids_cpu[i]])
            lc = np.fromfile(str(tmp_path / "c.bin"), np.float32).reshape(n, -1); lg = np.fromfile(str(tmp_path / "g.bin"), np.float32).reshape(n, -1)
            worst = max(float(((lg[t] - lc[t]) ** 2).sum() / (lc[t] ** 2).sum()) for t in range(n))
            print(f"Q8_0 8B, fa={fa}: {n}/{n} greedy ids identical, worst logits NMSE {worst:.2e}")
            assert worst < 1e-3, worst
    finally:
        os.remove(gguf)


# ---- the LDS-DMA ring form of the prefill FLASH_ATTN_EXT kernel (fattn_mma.hip k_fattn_dma128: K / V tiles by global_load_lds, V^T fragments by
# ds_read_b64_tr_b16, two heads of a KV head per workgroup, deferred running maximum): taken for head size 128 from 512 workgroups on.  Against the
# float64 restatement of ggml_compute_forward_flash_attn_ext_f16 (ops.cpp:7912-8148), bar = the reference's NMSE 5e-4 for this op, and against the
# register-staged kernel it replaces (option fattn_dma 0).
@pytest.mark.parametrize("D,nq,nh,nhkv,nkv,ns,kind", [
    (128, 256, 32, 8, 300, 8, "causal"),        # ragged last tile, diagonal tiles, dead tiles; pairs of heads per workgroup
    (128, 384, 16, 16, 777, 11, "none"),        # no mask, no GQA (one head per workgroup), odd number of live tiles
    (128, 256, 32, 8, 1024, 8, "padded"),       # the second half of the cache view unused
    (128, 256, 32, 4, 512, 8, "sinks"),
    (128, 256, 32, 8, 640, 8, "softcap"),
    (128, 128, 64, 8, 4096, 8, "spike"),        # one key far above the rest late in the row: the deferred maximum must follow it
    # head size 64 (round 5: the encoders' shape; rows of 128 bytes, one head per workgroup, taken from 192 workgroups on)
    (64, 1500, 16, 16, 1500, 1, "none"),        # Whisper-medium over 30 s (tools/omni/audition.cpp:596-640): 1500 x 1500 x 16 heads, ragged last query and key tiles
    (64, 256, 16, 16, 300, 6, "causal"),
    (64, 200, 20, 20, 777, 5, "none"),
    (64, 256, 16, 4, 512, 6, "sinks"),
    (64, 256, 16, 16, 640, 6, "softcap"),
    (64, 256, 16, 8, 1024, 6, "padded"),
    (64, 128, 32, 8, 2048, 6, "spike")])
def test_flash_attn_prefill_dma_ring(pkg, be, D, nq, nh, nhkv, nkv, ns, kind):
    import numpy as np
    from conftest import nmse
    from test_gpu_parity import _attn_f64, run_graph
    rng = np.random.default_rng(nq + nkv + nh)
    qv = rng.standard_normal((ns, nh, nq, D)).astype(np.float32)
    kv = rng.standard_normal((ns, nhkv, nkv, D)).astype(np.float16)
    vv = rng.standard_normal((ns, nhkv, nkv, D)).astype(np.float16)
    if kind == "spike":                      # q . k of one late key ~ 40 x the typical score: forces the rescale long after the first tiles
        kv[:, :, nkv - 37, :] = (qv[:, ::nh // nhkv, nq // 2, :] * 3.0).astype(np.float16)
    mask = None
    if kind not in ("none",):
        mask = np.zeros((nq, nkv), np.float16)
        off = nkv - nq if kind != "padded" else nkv // 2 - nq
        for i in range(nq):
            mask[i, max(0, min(nkv, off + i + 1)):] = -np.inf
    sinks = rng.standard_normal(nh).astype(np.float32) if kind == "sinks" else None
    softcap = 7.0 if kind == "softcap" else 0.0
    scale = 1.0 / np.sqrt(D)
    outs = []
    for dma in (1, 0):
        be.set_option("fattn_dma", dma)
        try:
            c = pkg.Context(be)
            q = c.new_tensor(pkg.GGML_TYPE_F32, D, nq, nh, ns)
            k = c.new_tensor(pkg.GGML_TYPE_F16, D, nkv, nhkv, ns)
            v = c.new_tensor(pkg.GGML_TYPE_F16, D, nkv, nhkv, ns)
            feeds = [(q, qv), (k, kv), (v, vv)]
            m = sk = None
            if mask is not None:
                m = c.new_tensor(pkg.GGML_TYPE_F16, nkv, nq); feeds.append((m, mask))
            if sinks is not None:
                sk = c.new_tensor(pkg.GGML_TYPE_F32, nh); feeds.append((sk, sinks))
            y = c.flash_attn_ext(q, k, v, m, scale, 0.0, softcap, sk)
            before = be.get_stat("fattn_dma_launches")
            (got,) = run_graph(be, c, [y], feeds)
            assert (be.get_stat("fattn_dma_launches") - before >= 1) == (dma == 1), "the shape did not select the kernel under test"
            outs.append(got.astype(np.float64).reshape(ns, nq, nh, D))
        finally:
            be.set_option("fattn_dma", -1)
    want = _attn_f64(qv, kv, vv, mask, scale, softcap, sinks)
    assert np.isfinite(outs[0]).all()
    assert nmse(outs[0], want) < 2e-6, nmse(outs[0], want)           # (bar 5e-4; P is rounded to f16: ~1e-7)
    assert nmse(outs[0], outs[1]) < 2e-6


# ---- flash-attention OFF, batches of query rows: MUL_MAT(k, q) -> SOFT_MAX_EXT -> MUL_MAT(v^T, p) -> PERMUTE -> CONT runs as ONE flash-attention launch that reads V^T as it
# lies (graph.cpp exec_attn_sm_prefill, fattn_mma.hip V^T staging).  Against the REFERENCE CPU backend running the same five nodes (ops.cpp:5072-5182 between two
# ggml_compute_forward_mul_mat): q and p rounded to f16 on both sides, f32 sums -- bar 2e-6 NMSE (the reference's own MUL_MAT / SOFT_MAX bars are 5e-4 / 1e-6).
@pytest.mark.parametrize("D,nq,H,HK,nkv,n_ctx,mask,kind", [
    (128, 77, 8, 2, 256, 512, "f32", "llama"),        # the llama -fa 0 graph: K rows and V^T rows are views of caches with room for n_ctx cells, f32 causal mask padded to 96 rows
    (128, 512, 32, 8, 512, 512, "f32", "llama"),      # pp512 at the 8B head counts
    (64, 150, 4, 4, 150, 150, None, "whisper"),       # encoder: no mask, K a strided permutation, V^T rows of 300 bytes (4-byte aligned only)
    (64, 151, 4, 4, 151, 151, None, "whisper"),       # ... 302 bytes (2-byte aligned only), ragged last tile
    (128, 40, 4, 1, 1000, 1024, "f16", "llama")])     # an f16 mask, 4 query heads per KV head
def test_soft_max_attention_of_a_batch_is_one_launch_vs_reference_backend(pkg, be, ref_be, D, nq, H, HK, nkv, n_ctx, mask, kind):
    import numpy as np
    from conftest import nmse
    F32, F16 = pkg.GGML_TYPE_F32, pkg.GGML_TYPE_F16
    rng = np.random.default_rng(D + nq + nkv)
    nq_pad = (nq + 31) // 32 * 32

    def build(c):
        ins = {}
        if kind == "llama":
            qc = c.new_tensor(F32, D, H, nq)                                   # [D, H, nq] as the rope leaves it -> permuted view [D, nq, H]
            kc = c.new_tensor(F16, D * HK, n_ctx)                              # K cache rows [HK * D] per cell
            vc = c.new_tensor(F16, n_ctx, D * HK)                              # transposed V cache: a row per (kv head, d)
            ins.update(q=qc, k=kc, v=vc)
            q = c.permute(qc, 0, 2, 1, 3)
            k = c.view_3d(kc, D, nkv, HK, D * HK * 2, D * 2, 0)
            v = c.view_3d(vc, nkv, D, HK, n_ctx * 2, n_ctx * 2 * D, 0)
        else:
            qc = c.new_tensor(F32, D, H, nq); kc = c.new_tensor(F32, D, H, nkv); vc = c.new_tensor(F32, D, H, nkv)
            ins.update(q=qc, k=kc, v=vc)
            q = c.permute(qc, 0, 2, 1, 3)
            k = c.permute(c.cast(kc, F16), 0, 2, 1, 3)
            v = c.cast(c.permute(vc, 1, 2, 0, 3), F16)
        m = None
        if mask:
            m = c.new_tensor(F16 if mask == "f16" else F32, nkv, nq_pad); ins["m"] = m
        kq = c.mul_mat(k, q)
        p = c.soft_max_ext(kq, m, 1.0 / np.sqrt(D), 0.0)
        kqv = c.mul_mat(v, p)
        out = c.cont(c.permute(kqv, 0, 2, 1, 3), D * H, nq)
        return ins, [out]

    feeds = {}
    if kind == "llama":
        feeds["q"] = rng.standard_normal(D * H * nq).astype(np.float32)
        feeds["k"] = rng.standard_normal(D * HK * n_ctx).astype(np.float16)
        feeds["v"] = rng.standard_normal(n_ctx * D * HK).astype(np.float16)
    else:
        feeds["q"] = rng.standard_normal(D * H * nq).astype(np.float32)
        feeds["k"] = rng.standard_normal(D * H * nkv).astype(np.float32)
        feeds["v"] = rng.standard_normal(D * H * nkv).astype(np.float32)
    if mask:
        mv = np.zeros((nq_pad, nkv), np.float32)
        for i in range(nq_pad):
            mv[i, max(1, min(nkv, nkv - nq + min(i, nq - 1) + 1)):] = -np.inf
        feeds["m"] = mv.astype(np.float16 if mask == "f16" else np.float32).ravel()
    res = []
    for backend in (be, ref_be):
        c = pkg.Context(backend)
        ins, outs = build(c)
        c.alloc()
        for name, t in ins.items():
            backend.tensor_set(t, feeds[name])
        backend.graph_compute(c.graph())
        if backend is be:
            launches = be.get_stat("kernels_last_graph")
        res.append(backend.tensor_get(outs[0]).copy())
        c.free()
    got, want = res
    assert np.isfinite(got).all()
    assert nmse(got, want) < 2e-6, nmse(got, want)
    assert launches <= (5 if kind == "whisper" else 3), launches       # the attention itself is one launch (+ mask cast / tile map, or the encoder's two casts and their copies)


# ---- BASELINE C3 as a MODEL, not per kernel: an 8B-width F16 prefill (n_embd 4096, 32 / 8 heads of 128, n_ff 12288; 2 layers, a 4096-row lm-head), one 512-token ubatch,
# every token's logits against the reference CPU backend running the same graph -- north_star's "within 1e-3 for F16 logits" -- with FLASH_ATTN_EXT and with the
# flash-attention-off node chain (llama-bench's default), through the GEMM tiles, the grouped q/k/v launch, the SWIGLU epilogue, the norm / rope row kernels, the prefill
# attention kernel (and its V^T form), the split-K reductions fused into the norms.
@pytest.mark.parametrize("flash_attn", [True, False])
def test_8b_width_f16_prefill_512_logits_nmse_1e6_and_within_1e3_on_all_but_1e5_of_the_entries(pkg, be, ref_be, flash_attn):
    import numpy as np
    from llama_cpp_omni_amd import qwen3
    cfg = dict(qwen3.QWEN3_8B, n_layer=2, n_vocab=4096)
    n_tok = 512
    rng = np.random.default_rng(512)
    embd = rng.standard_normal((n_tok, cfg["n_embd"])).astype(np.float32)
    outs = []
    for backend in (be, ref_be):
        mdl = qwen3.Model(backend, cfg, qwen3.uniform_types(cfg, pkg.GGML_TYPE_F16), n_ctx=512, seed=21, flash_attn=flash_attn)
        g, I, logits = mdl.build(n_tok, 512)
        mdl.set_inputs(I, embd, 0, 512)
        backend.graph_compute(g.graph())
        outs.append(backend.tensor_get(logits).copy().reshape(n_tok, cfg["n_vocab"]))
        g.free(); mdl.wctx.free()
    got, want = outs
    assert np.isfinite(got).all()
    # 2 M logits of magnitude up to ~6.5: both backends round the SAME f32 activations to f16 before every mat-mul, but activations that differ in the 7th digit (another
    # f32 summation order) land on different f16 neighbours now and then -- a 2^-11 step of that activation on one side only.  So the bar is stated on the distribution:
    # NMSE, the 1e-3 bound (relative to the largest logit) on all but 1e-5 of the entries, and a hard cap of 1.5e-3 on every entry.  Measured (round 6, MI355X): FA on
    # NMSE 9.7e-7, 2 of 2 097 152 entries above 1e-3, max 1.04e-3; FA off NMSE 3.7e-7, none above, max 6.7e-4 -- DESIGN.md section 4 carries these as the stated deviation
    # from north_star's "within 1e-3" at 8B width (strict 1e-3 on every entry holds on the tiny model: test_gpu_parity.py::test_f16_model_logits_within_1e3).
    from conftest import nmse
    scale = max(1.0, float(np.abs(want).max()))
    d = np.abs(got - want)
    stats = (nmse(got, want), float((d > 1e-3 * scale).mean()), float(d.max()) / scale)
    print("8B-width F16 prefill: NMSE %.2e, fraction above 1e-3 x max logit %.2e, max |d| / max logit %.2e" % stats)
    assert stats[0] < 1e-6, stats
    assert stats[1] < 1e-5, stats
    assert stats[2] < 1.5e-3, stats
    assert (got.argmax(1) == want.argmax(1)).mean() > 0.99                     # (near-ties among 4096 random logits may flip)


# ---- small asynchronous uploads sit in host staging until the backend's next entry point; the buffer-level paths settle them first now
def test_staged_async_upload_is_ordered_with_buffer_level_writes_and_survives_a_free(pkg, be):
    import numpy as np
    F32 = pkg.GGML_TYPE_F32
    c = pkg.Context(be)
    x = c.new_tensor(F32, 1024)
    y = c.scale(x, 2.0)
    c.alloc()
    a = np.full(1024, 1.0, np.float32); b = np.full(1024, 3.0, np.float32)
    be.tensor_set(x, np.zeros(1024, np.float32))
    be.tensor_set_async(x, a)                        # staged (4 KB)
    be.tensor_set(x, b)                              # blocking, buffer level, same bytes, LATER: must win
    be.graph_compute(c.graph())                      # (a backend entry point: would flush anything still staged)
    assert np.array_equal(be.tensor_get(y), 2.0 * b)
    # partial overlap: the staged write covers the first half only
    be.tensor_set_async(x, a[:512])
    be.tensor_set(x, b[:256], offset=1024)           # bytes 1024 .. 2047 = elements 256 .. 511
    be.graph_compute(c.graph())
    want = b.copy(); want[:256] = 1.0
    assert np.array_equal(be.tensor_get(y), 2.0 * want)
    # a buffer released with a write still staged for it: the write is settled (or dropped) before the free, later work is unaffected
    be.tensor_set_async(x, a)
    c.free()
    c2 = pkg.Context(be)
    z = c2.new_tensor(F32, 1024); w = c2.scale(z, 0.5)
    c2.alloc()
    be.tensor_set(z, b)
    be.graph_compute(c2.graph())
    assert np.array_equal(be.tensor_get(w), 0.5 * b)
    c2.free()


@pytest.mark.gpu
@pytest.mark.parametrize("F", [8448, 8320])
def test_gate_up_swiglu_96_row_tiles_vs_oracle(pkg, be, F):
    """ffn_gate / ffn_up + SWIGLU in one launch with 96-row tiles (k_gemm_f16_ph8<.., GLU, R96>): chosen when 128-row tiles leave a quarter of the CUs without a
    workgroup (n_ff 12288 x 512 tokens: 192 tiles on 256 CUs).  F = 8448 is a whole number of 96-row tiles (88), F = 8320 ends in a tile of 64 rows (clamped
    source rows, rows past F not stored).  The launch counter confirms the variant; ffn_down's output against the oracle's chain on sampled tokens and against the
    same graph with the fusion broken by a second reader of the GLU result."""
    from test_gpu_parity import run_graph
    from conftest import nmse
    from oracle import oracle_py as orc
    rng = np.random.default_rng(96 + F)
    E, N = 256, 512
    ty = pkg.GGML_TYPE_F16
    wg = (rng.standard_normal((F, E)) * 0.05).astype(np.float16); wu = (rng.standard_normal((F, E)) * 0.05).astype(np.float16)
    wd = (rng.standard_normal((E, F)) * 0.05).astype(np.float16)
    xv = rng.standard_normal((N, E)).astype(np.float32)
    outs = []
    for second_reader in (False, True):
        c = pkg.Context(be)
        x = c.new_tensor(pkg.GGML_TYPE_F32, E, N)
        tg, tu, td_ = c.new_tensor(ty, E, F), c.new_tensor(ty, E, F), c.new_tensor(ty, F, E)
        up = c.mul_mat(tu, x); gate = c.mul_mat(tg, x)
        act = c.swiglu_split(gate, up)
        y = c.mul_mat(td_, act)
        roots = [y] + ([c.scale(act, 1.0)] if second_reader else [])
        before = be.get_stat("gemm_glu96_launches")
        got = run_graph(be, c, roots, [(x, xv), (tg, wg), (tu, wu), (td_, wd)])
        fused = be.get_stat("gemm_glu96_launches") - before
        assert (fused == 0) if second_reader else (fused == 1), fused
        outs.append(got[0].reshape(N, E))
    cols = rng.choice(N, 48, replace=False)
    g_ = orc.mul_mat(ty, wg.view(np.uint8).reshape(F, -1), xv[cols]); u_ = orc.mul_mat(ty, wu.view(np.uint8).reshape(F, -1), xv[cols])
    a_ = (g_ / (1.0 + np.exp(-g_.astype(np.float64)))).astype(np.float32) * u_
    want = orc.mul_mat(ty, wd.view(np.uint8).reshape(E, -1), a_.astype(np.float32))
    assert np.isfinite(outs[0]).all()
    assert nmse(outs[0][cols], want) < 1e-6, nmse(outs[0][cols], want)
    assert nmse(outs[0], outs[1]) < 1e-9, nmse(outs[0], outs[1])


@pytest.mark.parametrize("E,F,N,two_addends", [(1024, 4096, 50, True), (1024, 1024, 100, True), (1152, 4304 - 16, 64, True), (1024, 4096, 50, False)])
def test_split_k_reduction_inside_the_layer_norm_vs_reference_backend(pkg, be, ref_be, E, F, N, two_addends):
    """An encoder layer's tail as the graphs spell it: x1 = W . h + bias + x0 (fc2 / wo of a streaming audio chunk: few columns, so the mat-mul is split along K),
    y = LayerNorm(x1) * w + b, z = W2 . y, r = z + x1.  The split-K slabs, the bias and the residual are summed by the LayerNorm launch itself (k_norm_rows with a
    norm_split_src: no reduction launch in between), x1 is still written for the later residual.  Against the reference CPU backend, and bit-identical to the same graph
    with the fold switched off (MI355X_NO_REDUCE_IN_LAYER_NORM is read once per process: the un-folded arithmetic is reached here through a second reader of x1 BEFORE the
    norm, which forces the reduction launch)."""
    from conftest import nmse
    rng = np.random.default_rng(E + F + N)

    def make(force_materialise):
        def build(c):
            h = c.new_tensor(pkg.GGML_TYPE_F32, F, N); x0 = c.new_tensor(pkg.GGML_TYPE_F32, E, N)
            W = c.new_tensor(pkg.GGML_TYPE_F16, F, E); bias = c.new_tensor(pkg.GGML_TYPE_F32, E)
            lw = c.new_tensor(pkg.GGML_TYPE_F32, E); lb = c.new_tensor(pkg.GGML_TYPE_F32, E); W2 = c.new_tensor(pkg.GGML_TYPE_F16, E, 256)
            x1 = c.add(c.mul_mat(W, h), bias)
            if two_addends:
                x1 = c.add(x1, x0)
            outs = []
            if force_materialise:
                outs.append(c.scale(x1, 1.0))                         # a reader of x1 in front of the norm
            y = c.add(c.mul(c.norm(x1, 1e-5), lw), lb)
            z = c.mul_mat(W2, y)
            outs = [z, c.scale(x1, 2.0)] + outs
            return dict(h=h, x0=x0, W=W, bias=bias, lw=lw, lb=lb, W2=W2), outs
        return build
    feeds = dict(h=rng.standard_normal(F * N).astype(np.float32), x0=rng.standard_normal(E * N).astype(np.float32),
                 W=(rng.standard_normal(E * F) / np.sqrt(F)).astype(np.float16), bias=(0.1 * rng.standard_normal(E)).astype(np.float32),
                 lw=(1 + 0.2 * rng.standard_normal(E)).astype(np.float32), lb=(0.1 * rng.standard_normal(E)).astype(np.float32),
                 W2=(rng.standard_normal(E * 256) / np.sqrt(E)).astype(np.float16))
    from test_prefill_kernels_gpu import _both
    f0 = be.get_stat("norm_from_split_launches")
    got, want = _both(pkg, be, ref_be, make(False), feeds)
    assert be.get_stat("norm_from_split_launches") == f0 + 1            # the LayerNorm launch took the slabs
    got2, _ = _both(pkg, be, be, make(True), feeds)
    assert be.get_stat("norm_from_split_launches") == f0 + 1            # ... and did not when x1 has a reader in front of it
    for g_, w_ in zip(got, want):
        assert np.isfinite(g_).all()
        assert nmse(g_, w_) < 1e-9, nmse(g_, w_)
    assert np.array_equal(got[0].view(np.uint32), got2[0].view(np.uint32))       # same bits with and without the fold
    assert np.array_equal(got[1].view(np.uint32), got2[1].view(np.uint32))


@pytest.mark.parametrize("E,F,N,op", [(1024, 4096, 50, "gelu"), (1152, 4304, 64, "gelu"), (1024, 4096, 100, "gelu_quick"), (1024, 512, 7, "gelu")])
def test_gelu_in_the_split_k_reduction_vs_reference_backend(pkg, be, ref_be, E, F, N, op):
    """An encoder's MLP on a streaming chunk: fc1 . y + bias -> GELU -> fc2.  fc1 has few columns, so it is split along K; the reduction launch adds the bias, applies the GELU
    (the reference's f16-table form) and writes the f16 activation image fc2 reads -- no unary launch, no f32 block.  Against the reference CPU backend, and bit-identical
    to the same graph with a second reader of the bias ADD (which keeps the GELU a launch of its own)."""
    from conftest import nmse
    from llama_cpp_omni_amd.ggml import UNARY
    from test_prefill_kernels_gpu import _both
    rng = np.random.default_rng(E + F + N)
    uop = UNARY.GELU if op == "gelu" else UNARY.GELU_QUICK

    def make(second_reader):
        def build(c):
            y = c.new_tensor(pkg.GGML_TYPE_F32, E, N); W1 = c.new_tensor(pkg.GGML_TYPE_F16, E, F); b1 = c.new_tensor(pkg.GGML_TYPE_F32, F); W2 = c.new_tensor(pkg.GGML_TYPE_F16, F, E)
            pre = c.add(c.mul_mat(W1, y), b1)
            z = c.mul_mat(W2, c.unary(pre, uop))
            return dict(y=y, W1=W1, b1=b1, W2=W2), [z] + ([c.scale(pre, 2.0)] if second_reader else [])
        return build
    feeds = dict(y=rng.standard_normal(E * N).astype(np.float32), W1=(rng.standard_normal(E * F) / np.sqrt(E) * 2).astype(np.float16), b1=(0.2 * rng.standard_normal(F)).astype(np.float32),
                 W2=(rng.standard_normal(E * F) / np.sqrt(F)).astype(np.float16))
    got, want = _both(pkg, be, ref_be, make(False), feeds)
    n_folded = be.get_stat("kernels_last_graph")
    got2, _ = _both(pkg, be, be, make(True), feeds)
    n_plain = be.get_stat("kernels_last_graph")
    if F % 64 == 0:                                                    # (fc2 with K = 4304 is not an MFMA-only reader: the GELU stays a launch, the results must still agree)
        assert n_folded <= n_plain - 2, (n_folded, n_plain)              # (the second reader's launch and the GELU launch)
    assert np.isfinite(got[0]).all()
    assert nmse(got[0], want[0]) < 1e-9, nmse(got[0], want[0])
    assert np.array_equal(got[0].view(np.uint32), got2[0].view(np.uint32))


@pytest.mark.parametrize("it,n_tok,kv_size", [(0, 50, 1500), (3, 50, 1500), (1, 100, 750)])
def test_streaming_encoder_attention_block_vs_reference_backend(pkg, be, ref_be, it, n_tok, kv_size):
    """One attention block of the streaming Whisper graph as the reference spells it (audition.cpp:455-636), chunk `it` of a stream: q / k / v projections of the chunk's tokens
    (k without bias), Kcur -> CPY into a contiguous run of the f16 K cache, Vcur -> TRANSPOSE -> CPY into a [n_tok, n_state] view of the TRANSPOSED f16 V cache (rows a cache pitch
    apart), then K = a view of the cache window, V = CAST(PERMUTE(RESHAPE(CONT(TRANSPOSE(view of the V cache window))))), soft-max attention over the window, wo + bias + residual,
    LayerNorm.  On the device: the two cache stores are written by the q / k / v reduction launch (no CPY launches, no f32 K / V rows), the CONT + CAST copies of the V window are
    skipped (the fused attention reads V^T in the cache), the wo reduction runs inside the LayerNorm launch.  Compared with the reference CPU backend: the attention output, the
    normalised rows and BOTH caches afterwards (cells of earlier chunks untouched, the new cells equal)."""
    from conftest import nmse
    from test_prefill_kernels_gpu import _both
    E, H, D = 1024, 16, 64
    rng = np.random.default_rng(100 + it + n_tok)
    tot = (it + 1) * n_tok

    def build(c):
        F32, F16 = pkg.GGML_TYPE_F32, pkg.GGML_TYPE_F16
        x = c.new_tensor(F32, E, n_tok); res = c.new_tensor(F32, E, n_tok)
        W = {k: c.new_tensor(F16, E, E) for k in ("q", "k", "v", "o")}
        B = {k: c.new_tensor(F32, E) for k in ("q", "v", "o", "lw", "lb")}
        kc = c.new_tensor(F16, E * kv_size); vc = c.new_tensor(F16, E * kv_size)
        Q = c.reshape(c.add(c.mul_mat(W["q"], x), B["q"]), D, H, n_tok)
        Kc = c.reshape(c.mul_mat(W["k"], x), D, H, n_tok)
        Vc = c.reshape(c.add(c.mul_mat(W["v"], x), B["v"]), D, H, n_tok)
        st_k = c.cpy(Kc, c.view_1d(kc, n_tok * E, 2 * E * it * n_tok) if hasattr(c, "view_1d") else c.view_2d(kc, n_tok * E, 1, 2 * n_tok * E, 2 * E * it * n_tok))
        st_v = c.cpy(c.transpose(c.reshape(Vc, E, n_tok)), c.view_2d(vc, n_tok, E, 2 * kv_size, 2 * it * n_tok))
        K = c.view_3d(kc, D, tot, H, 2 * E, 2 * D, 0)
        V2t = c.cont(c.transpose(c.view_2d(vc, tot, E, 2 * kv_size, 0)))
        V = c.cast(c.permute(c.reshape(V2t, D, H, tot), 1, 2, 0, 3), F16)
        KQ = c.soft_max_ext(c.mul_mat(K, c.permute(Q, 0, 2, 1, 3)), None, 1.0 / 8.0, 0.0)
        att = c.cont(c.permute(c.mul_mat(V, KQ), 0, 2, 1, 3), E, n_tok)
        x1 = c.add(c.add(c.mul_mat(W["o"], att), B["o"]), res)
        y = c.add(c.mul(c.norm(x1, 1e-5), B["lw"]), B["lb"])
        ins = dict(x=x, res=res, kc=kc, vc=vc, **{"W" + k: v for k, v in W.items()}, **{"B" + k: v for k, v in B.items()})
        # the stores are graph roots (ggml_build_forward_expand on the CPY nodes) in front of their readers
        return ins, [st_k, st_v, c.scale(att, 1.0), y, c.scale(x1, 1.0)]      # (a reader each for att and x1: rows only GEMMs / the norm read are not written in f32)
    feeds = dict(x=rng.standard_normal(E * n_tok).astype(np.float32), res=rng.standard_normal(E * n_tok).astype(np.float32),
                 kc=(rng.standard_normal(E * kv_size) * 0.5).astype(np.float16), vc=(rng.standard_normal(E * kv_size) * 0.5).astype(np.float16))
    for k in ("q", "k", "v", "o"):
        feeds["W" + k] = (rng.standard_normal(E * E) / np.sqrt(E)).astype(np.float16)
    for k in ("q", "v", "o", "lb"):
        feeds["B" + k] = (0.1 * rng.standard_normal(E)).astype(np.float32)
    feeds["Blw"] = (1 + 0.2 * rng.standard_normal(E)).astype(np.float32)
    f0 = be.get_stat("norm_from_split_launches")
    got, want = _both(pkg, be, ref_be, build, feeds)
    launches = be.get_stat("kernels_last_graph")
    assert be.get_stat("norm_from_split_launches") == f0 + 1
    assert launches <= 8, launches              # x image, q/k/v GEMM (+ reduction with the two cache stores), attention, wo GEMM, LayerNorm (+ reduction), the two extra readers (+ slack 1)
    for name, g_, w_ in zip(("k cache run", "v cache view", "attention", "layer norm", "x1"), got, want):
        g32, w32 = g_.astype(np.float32), w_.astype(np.float32)
        assert np.isfinite(g32).all(), name
        assert nmse(g32, w32) < 2e-6, (name, nmse(g32, w32))


def test_grouped_split_k_slabs_are_summed_by_the_norm_rope_launch(pkg, be, ref_be):
    """Prefill ubatch of 512 tokens at the 8B widths, one layer: wq / wk / wv run as ONE grouped launch with two K halves; the q / k norm + rope + K / V store launch behind it sums
    the slabs itself (k_norm_rope_v4 with slab sources) -- k_gemm_reduce_multi does not run for them.  Logits of the last 64 tokens against the reference CPU backend on the same graph; the launch counter
    confirms the path."""
    from conftest import nmse
    from llama_cpp_omni_amd import qwen3
    cfg = dict(qwen3.QWEN3_8B, n_layer=1, n_vocab=4096)
    T, n_kv = 512, 512
    rng = np.random.default_rng(12)
    embd = rng.standard_normal((T, cfg["n_embd"])).astype(np.float32)
    outs = []; launches = None
    for backend in (be, ref_be):
        mdl = qwen3.Model(backend, cfg, qwen3.uniform_types(cfg, pkg.GGML_TYPE_F16), n_ctx=n_kv, seed=5, flash_attn=True)
        g, I, logits = mdl.build(T, n_kv, n_outputs=64)
        mdl.set_inputs(I, embd, 0, n_kv)
        backend.tensor_set(I["out_ids"], np.arange(T - 64, T, dtype=np.int32))
        gr = g.graph()
        if backend is be:
            launches = backend.get_stat("norm_rope_split_launches")
        backend.graph_compute(gr); backend.synchronize()
        if backend is be:
            launches = backend.get_stat("norm_rope_split_launches") - launches
        outs.append(backend.tensor_get(logits).copy())
        g.free(); mdl.wctx.free()
    assert np.isfinite(outs[0]).all()
    assert nmse(outs[0], outs[1]) < 1e-6, nmse(outs[0], outs[1])
    assert launches == 1, launches                                     # the layer's norm + rope + store launch took the slabs
