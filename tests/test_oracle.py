"""CPU tests: the C restatement (oracle/omni_oracle.c) is pinned against golden vectors produced by the REAL
reference (tests/golden/*.npz, generator oracle/make_golden.py) -- bit-exact for every integer / byte stage,
tight tolerances for float stages -- and, when oracle/_ref is present, against the live reference library."""
import ctypes as C

import numpy as np
import pytest

from conftest import nmse
from oracle import oracle_py as orc

TYPES = {"q4_K": orc.Q4_K, "q6_K": orc.Q6_K, "q8_0": orc.Q8_0}


@pytest.mark.parametrize("name", list(TYPES))
def test_dequant_bit_exact_vs_reference(golden, name):
    g = golden["quant"]
    got = orc.dequantize(TYPES[name], g[f"{name}_blocks"], 12288)
    assert np.array_equal(got.view(np.uint32), g[f"{name}_deq"].view(np.uint32))


def test_quantize_q8_K_bit_exact_vs_reference(golden):
    g = golden["quant"]
    assert np.array_equal(orc.quantize_q8_K(g["y"]), g["y_q8_K"])
    # edge rows: all-zero block, |x| tie with opposite signs (first one wins), mirrored block, denormal-scale block
    assert np.array_equal(orc.quantize_q8_K(g["edge"]), g["edge_q8_K"])


def test_quantize_q8_0_bit_exact_vs_reference_x86_path(golden):
    g = golden["quant"]
    assert np.array_equal(orc.quantize_q8_0(g["y"]), g["y_q8_0"])


@pytest.mark.parametrize("name", list(TYPES))
def test_vec_dot_vs_reference(golden, name):
    g = golden["quant"]
    act = g["y_q8_0"] if name == "q8_0" else g["y_q8_K"]
    for k, want in zip((256, 4096, 12288), g[f"{name}_dots"]):
        got = orc.vec_dot(TYPES[name], k, g[f"{name}_blocks"], act)
        # integer sub-sums are exact; only the order of the f32 additions differs (generic C vs AVX2 lanes)
        assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (k, got, want)


def test_f16_conversion_exhaustive():
    h = np.arange(65536, dtype=np.uint16)
    f = h.view(np.float16).astype(np.float32)
    ok = ~np.isnan(f)
    back = orc.f32_to_f16(f)
    assert np.array_equal(back[ok], h[ok])
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000) * 10.0 ** rng.integers(-9, 6, 200000), [65504, 65519.9, 65520, 1e-8, 6e-8, -0.0]]).astype(np.float32)
    with np.errstate(over="ignore"):
        want = x.astype(np.float16).view(np.uint16)
    assert np.array_equal(orc.f32_to_f16(x), want)


def test_ops_vs_reference_cpu_backend(golden):
    g = golden["ops"]
    for tag in ("rms4096", "rms128"):
        got = orc.rms_norm(g[f"{tag}_x"], 1e-6) * g[f"{tag}_w"]
        assert np.abs(got - g[f"{tag}_y"]).max() <= 2e-6 * np.abs(g[f"{tag}_y"]).max()
    assert nmse(orc.rope(g["rope_neox_x"], g["rope_neox_pos"], 128, 2, 40960, 1e6), g["rope_neox_y"]) < 1e-10
    assert nmse(orc.rope(g["rope_norm_x"], g["rope_norm_pos"], 128, 0, 40960, 1e4), g["rope_norm_y"]) < 1e-10
    m = g["softmax_mask"].astype(np.float32)
    x = g["softmax_x"]
    got = np.stack([orc.soft_max(x[b], m[:4], 0.0883883) for b in range(x.shape[0])])
    assert nmse(got, g["softmax_y"]) < 1e-10
    assert nmse(orc.swiglu(g["swiglu_a"], g["swiglu_b"]), g["swiglu_y"]) < 1e-10
    exp = np.zeros((16, 1024), np.uint16)
    exp[g["setrows_idx"]] = orc.f32_to_f16(g["setrows_src"])
    assert np.array_equal(exp, g["setrows_tab"])
    for name, ty in (("q4_K", orc.Q4_K), ("q6_K", orc.Q6_K), ("q8_0", orc.Q8_0), ("f16", orc.F16)):
        for (M, K, N) in ((16, 256, 1), (64, 1024, 3)):
            tag = f"mm_{name}_{M}x{K}x{N}"
            w = g[tag + "_w"]
            W = w.view(np.uint8).reshape(M, -1)
            got = orc.mul_mat(ty, W, g[tag + "_x"])
            assert nmse(got, g[tag + "_y"]) < 1e-9, tag


def test_flash_attn_vs_reference_cpu_backend(golden):
    g = golden["ops"]
    for nkv in (256, 2048):
        q, k, v, m, y = (g[f"fa{nkv}_{n}"] for n in ("q", "k", "v", "mask", "y"))
        for h in range(8):
            for iq in range(2):
                got = orc.flash_attn_row(q[h, iq], k[h // 4], v[h // 4], m[iq], 1.0 / np.sqrt(128.0))
                assert nmse(got, y[iq, h]) < 1e-6, (nkv, h, iq)


def test_oracle_vs_live_reference_library(golden):
    """Where oracle/_ref exists, run the reference's own functions on fresh random data and compare."""
    from oracle.ref_backend import ref_available, ref_lib
    if not ref_available():
        pytest.skip("oracle/_ref not built")
    ref = ref_lib()
    rng = np.random.default_rng(99)
    x = (rng.standard_normal(8192) * rng.choice([0.01, 1.0, 100.0], 8192)).astype(np.float32)
    x[256:512] = 0
    out = np.zeros(8192 // 256 * 292, np.uint8)
    ref.quantize_row_q8_K(x.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p), C.c_int64(8192))
    mine = orc.quantize_q8_K(x)
    o, m = out.reshape(-1, 292).copy(), mine.reshape(-1, 292).copy()
    o[1, 260:] = 0          # zero block: reference leaves bsums stale
    assert np.array_equal(o, m)
    out0 = np.zeros(8192 // 32 * 34, np.uint8)
    ref.quantize_row_q8_0(x.ctypes.data_as(C.c_void_p), out0.ctypes.data_as(C.c_void_p), C.c_int64(8192))
    assert np.array_equal(out0, orc.quantize_q8_0(x))


def test_reference_decorrelates_under_a_1e6_perturbation():
    """Control for the end-to-end logit bars of the GPU tests: the reference CPU backend against ITSELF.  An 8-layer Q4_K_M model
    (n_embd 1024) decodes one token twice, the second time with the input embedding perturbed by 1e-6 relative -- the size of an f32
    summation-order difference.  Every mat-vec re-quantises its input to Q8_K; once one rounding flips, the +-1 step is a 1e-2
    perturbation for everything downstream, so the logits differ by ~1e-3 NMSE although both runs are "the reference".  Any two
    implementations that add the same exact integer block sums in a different f32 order sit on this noise floor."""
    import numpy as np
    from conftest import load_pkg, nmse
    from oracle.ref_backend import make_ref_cpu_backend, ref_available
    if not ref_available():
        pytest.skip("oracle/_ref/libggml-ref.so not built")
    pkg = load_pkg()
    from llama_cpp_omni_amd import qwen3
    be = make_ref_cpu_backend(pkg, 8)
    cfg = dict(n_embd=1024, n_layer=8, n_head=8, n_head_kv=2, head_dim=128, n_ff=3072, n_vocab=2048, rms_eps=1e-6, rope_base=1e6, n_ctx_orig=4096)
    types = qwen3.q4_k_m_types(cfg)
    rng = np.random.default_rng(3)
    embd = rng.standard_normal((1, cfg["n_embd"])).astype(np.float32)
    outs = []
    for eps in (0.0, 1e-7, 1e-6):
        mdl = qwen3.Model(be, cfg, types, n_ctx=256, seed=11, flash_attn=True)
        g, I, logits = mdl.build(1, 256)
        e = (embd * (1.0 + eps * np.sign(rng.standard_normal(embd.shape)))).astype(np.float32)
        mdl.set_inputs(I, e, 0, 256)
        be.graph_compute(g.graph())
        outs.append(be.tensor_get(logits).copy())
        g.free(); mdl.wctx.free()
    be.close()
    assert nmse(outs[1], outs[0]) < 1e-9            # (1e-7: no rounding happened to flip in this small model)
    assert 1e-5 < nmse(outs[2], outs[0]) < 1e-1     # 1e-6: one flipped, the outputs decorrelate to the rounding-noise floor


def test_reference_build_flags_give_the_same_bytes(tmp_path):
    """oracle/_ref is the reference built with `-std=c11 -ffp-contract=off -march=x86-64-v3` (oracle/Makefile.ref); the reference's OWN default build is GGML_NATIVE +
    gnu11 (ggml/CMakeLists.txt:265), under which gcc contracts `iscale * x + 12582912.f` inside nearest_int (ggml-quants.c) into one fma.  Both builds (the second:
    `make -f oracle/Makefile.ref native`, oracle/_ref/native/) run oracle/ref_flags_probe.py in a process of their own:
      * on the golden signal of tests/golden/quant.npz every byte the parity tests are pinned to -- quantize_row_q8_K / _q8_0 of y and of the tie / zero edge rows, the
        Q4_K / Q6_K / Q8_0 blocks of x -- and the three vec_dot scalars per type are IDENTICAL under both flag sets (and equal to the fixture);
      * on 2^20 seeded normal values at block scales 2^-20 .. 2^20 the Q8_0 image is identical, every Q8_K block scale d is identical, and a handful of Q8_K quants
        (measured: 3 of 1 048 576) differ by one step -- exact ties of the magic-number rounding, which the contracted fma resolves on the unrounded product.
    "Bit-exact with the reference CPU backend" therefore means: with either build on the fixtures; with the c11 build (the stricter reading of the source) at ties."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    libs = [os.path.join(root, "oracle", "_ref", "libggml-ref.so"), os.path.join(root, "oracle", "_ref", "native", "libggml-ref.so")]
    if not all(os.path.exists(l) for l in libs):
        pytest.skip("needs oracle/_ref and oracle/_ref/native (make -f oracle/Makefile.ref all native; the second is tied to the build container's CPU)")
    outs, arrs = [], []
    for i, l in enumerate(libs):
        npz = str(tmp_path / f"b{i}.npz")
        r = subprocess.run([sys.executable, os.path.join(root, "oracle", "ref_flags_probe.py"), l, npz], capture_output=True, text=True, timeout=300)
        if r.returncode != 0:
            pytest.skip(f"{l} does not run on this CPU: {r.stderr[-200:]}")
        outs.append(json.loads(r.stdout.strip().splitlines()[-1])); arrs.append(np.load(npz))
    a, b = outs
    for k in a:
        if k.endswith("_diff_bytes_vs_golden"):
            assert a[k] == 0 and b[k] == 0, (k, a[k], b[k])
        elif k.endswith("_dots_rel_vs_golden"):
            assert max(a[k]) == 0.0 and max(b[k]) == 0.0, (k, a[k], b[k])
        elif not k.startswith("rand_"):
            assert a[k] == b[k], k
    assert a["rand_q8_0"] == b["rand_q8_0"]
    qa, qb = arrs[0]["rand_q8_K"].reshape(-1, 292), arrs[1]["rand_q8_K"].reshape(-1, 292)
    assert (qa[:, :4] == qb[:, :4]).all()                               # block scales d
    ia, ib = qa[:, 4:260].view(np.int8).astype(np.int32), qb[:, 4:260].view(np.int8).astype(np.int32)
    n_diff = int((ia != ib).sum())
    print(f"Q8_K quants differing between the c11 / no-contraction and the gnu11 / native build: {n_diff} of {ia.size}")
    assert np.abs(ia - ib).max() <= 1 and n_diff <= ia.size // 100000, n_diff
    assert (arrs[0]["rand"] == arrs[1]["rand"]).all()
