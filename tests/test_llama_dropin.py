"""True drop-in run: the REFERENCE's libllama + ggml_backend_sched (oracle/_ref, built from the reference
sources by oracle/Makefile.ref `llama`) load this repo's backend as a plug-in from GGML_BACKEND_PATH and decode a synthetic
2-layer Q4_K_M GGUF written by tools/make_synth_gguf.py.  Greedy token ids must equal the reference CPU backend's, logits within
the reference's MUL_MAT / FLASH_ATTN_EXT bar (NMSE 5e-4).  The binaries travel with the snapshot; nothing here reads /root/reference."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "llama-bench-min")
LIB = os.path.join(ROOT, "llama.cpp-omni_amd", "lib", "libggml-mi355x.so")


def _greedy(gguf, ngl, fa, dump, env_extra=None):
    env = dict(os.environ)
    env.pop("GGML_BACKEND_PATH", None)
    if env_extra:
        env.update(env_extra)
    out = subprocess.run([BIN, "-m", gguf, "-ngl", str(ngl), "-fa", str(fa), "--greedy", "24", "-t", "4", "--dump-logits", dump],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads(out.stdout.strip().splitlines()[-1])["greedy_ids"], np.fromfile(dump, np.float32), out.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("config,types,fa", [("tiny", "q4_k_m", 1), ("tiny", "q4_k_m", 0), ("tiny", "q5_k", 1),
                                             # the omni TTS decoder's family: arch llama (RoPE NORM, no q/k-norm), Q8_0 / F16 weights
                                             ("tts-tiny", "q8_0", 1), ("tts-tiny", "f16", 1), ("tts-tiny", "q8_0", 0)])
def test_reference_libllama_drives_the_plugin(tmp_path, config, types, fa):
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/llama-bench-min not built (make -f oracle/Makefile.ref llama)")
    gguf = str(tmp_path / "tiny.gguf")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_gguf.py"), "--config", config, "--types", types, "-o", gguf,
                    "--distinct-layers"], check=True, timeout=300)
    ids_cpu, l_cpu, _ = _greedy(gguf, 0, fa, str(tmp_path / "cpu.bin"))
    ids_gpu, l_gpu, err = _greedy(gguf, 99, fa, str(tmp_path / "gpu.bin"), {"GGML_BACKEND_PATH": LIB})
    assert "MI355X0" in err and "offloaded 3/3 layers to GPU" in err          # the plug-in really ran the layers
    assert ids_gpu == ids_cpu
    nm = float(((l_cpu - l_gpu) ** 2).sum() / (l_cpu ** 2).sum())
    assert nm < 5e-4


@pytest.mark.gpu
@pytest.mark.parametrize("types", ["q4_0"])
def test_image_quant_models_stay_on_the_gpu(tmp_path, types):
    """Q4_0 models (no integer-dot kernels here: MUL_MAT on the resident F16 image of the blocks, GET_ROWS de-quantising): every
    layer offloaded, logits inside the reference's bar against the CPU backend's integer arithmetic; greedy ids may flip on a near-tie
    of this random-weight toy model (f16-rounded activations vs Q8 activations), so 90 % agreement is asked for"""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/llama-bench-min not built")
    gguf = str(tmp_path / "tiny.gguf")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_gguf.py"), "--config", "tiny", "--types", types, "-o", gguf,
                    "--distinct-layers"], check=True, timeout=300)
    ids_cpu, l_cpu, _ = _greedy(gguf, 0, 1, str(tmp_path / "cpu.bin"))
    ids_gpu, l_gpu, err = _greedy(gguf, 99, 1, str(tmp_path / "gpu.bin"), {"GGML_BACKEND_PATH": LIB})
    assert "MI355X0" in err and "offloaded 3/3 layers to GPU" in err and "graph splits = 2" in err
    agree = sum(a == b for a, b in zip(ids_gpu, ids_cpu))
    assert agree >= 0.9 * len(ids_cpu), (ids_gpu, ids_cpu)


@pytest.mark.gpu
@pytest.mark.parametrize("n_par,fa,unified", [(3, 1, 1), (8, 0, 1), (4, 0, 0), (5, 0, 1), (8, 1, 1), (4, 1, 0), (16, 0, 1), (20, 1, 0), (24, 1, 1), (40, 0, 0)])
def test_parallel_sequences_through_libllama(tmp_path, n_par, fa, unified):
    """Several sequences decoded together (one token each per llama_decode): 2..5-column mat-vecs, the int8-MFMA kernel from 6 columns
    (16 / 24 / 40 sequences: two to four token groups, two passes above 32), attention with one mask row per
    sequence (unified KV) or one KV stream per sequence.  Without flash-attention both backends do the same arithmetic and every
    greedy id must be equal; with it the CPU accumulates V in f16 and this backend in f32 (logits NMSE ~4e-5), which on this
    random-weight toy model can flip a near-tie, so a small number of differing positions is tolerated there."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/llama-bench-min not built")
    gguf = str(tmp_path / "tiny.gguf")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_gguf.py"), "--config", "tiny", "--types", "q4_k_m", "-o", gguf,
                    "--distinct-layers"], check=True, timeout=300)

    def run(ngl, env_extra):
        env = dict(os.environ)
        env.pop("GGML_BACKEND_PATH", None)
        env.update(env_extra)
        out = subprocess.run([BIN, "-m", gguf, "-ngl", str(ngl), "-fa", str(fa), "--greedy", "16", "--parallel", str(n_par), "--kv-unified", str(unified), "-t", "4"],
                             env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stderr[-2000:]
        return json.loads(out.stdout.strip().splitlines()[-1])["greedy_ids"]

    gpu, cpu = run(99, {"GGML_BACKEND_PATH": LIB}), run(0, {})
    assert len(gpu) == len(cpu) == n_par
    if fa == 0:
        assert gpu == cpu
    else:
        same = sum(a == b for sa, sb in zip(gpu, cpu) for a, b in zip(sa, sb))
        assert same >= 0.9 * 16 * n_par, (same, gpu, cpu)


@pytest.mark.gpu
def test_graph_optimize_hook_under_the_scheduler(tmp_path):
    """ggml_backend_sched calls the plug-in's graph_optimize before ggml-alloc: sibling mat-muls get grouped, so the decode graph needs
    fewer launches -- and the tokens do not change either way."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/llama-bench-min not built")
    import re
    gguf = str(tmp_path / "tiny.gguf")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_gguf.py"), "--config", "tiny", "--types", "q4_k_m", "-o", gguf,
                    "--distinct-layers"], check=True, timeout=300)
    res = {}
    for tag, extra in (("opt", {}), ("noopt", {"MI355X_NO_GRAPH_OPTIMIZE": "1"})):
        env = {"GGML_BACKEND_PATH": LIB, "MI355X_LOG_STATS": "1"}
        env.update(extra)
        ids, _, err = _greedy(gguf, 99, 1, str(tmp_path / f"{tag}.bin"), env)
        m = re.search(r"kernels in last graph=(\d+)", err)
        assert m, err[-1500:]
        res[tag] = (ids, int(m.group(1)))
    assert res["opt"][0] == res["noopt"][0]
    assert res["opt"][1] < res["noopt"][1]


@pytest.mark.parametrize("types", ["q4_k_m", "q4_0", "q5_k", "q8_0"])
def test_synthetic_gguf_loads_on_reference_cpu(tmp_path, types):
    """CPU-only: the file format written by tools/make_synth_gguf.py is accepted by the reference loader and decodes."""
    if not os.path.exists(BIN):
        pytest.skip("oracle/_ref/llama-bench-min not built")
    gguf = str(tmp_path / "tiny.gguf")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_gguf.py"), "--config", "tiny", "--types", types, "-o", gguf,
                    "--distinct-layers"], check=True, timeout=300)
    ids, logits, _ = _greedy(gguf, 0, 1, str(tmp_path / "cpu.bin"))
    assert len(ids) == 24 and np.isfinite(logits).all() and logits.size == 512
