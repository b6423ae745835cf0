"""Round-3 host-side pieces that need no GPU: the synthetic omni module files against the reference's own loaders (CPU backend, where oracle/_ref was built),
bench.py's CPU placement order, the graph-dump diff tool."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENC = os.path.join(ROOT, "oracle", "_ref", "omni-enc-min")


def test_cpu_order_lists_every_allowed_cpu_once_physical_cores_first():
    sys.path.insert(0, ROOT)
    import bench
    allowed = sorted(os.sched_getaffinity(0))
    order, topo = bench._cpu_order(allowed)
    assert sorted(order) == allowed and len(set(order)) == len(order)
    assert 1 <= topo["physical_cores"] <= len(allowed) and topo["nodes"] >= 1
    # the first `physical_cores` entries are one logical CPU per (package, core)
    seen = set()
    for c in order[:topo["physical_cores"]]:
        p = f"/sys/devices/system/cpu/cpu{c}/topology"
        try:
            key = (open(p + "/physical_package_id").read().strip(), open(p + "/core_id").read().strip())
        except OSError:
            key = c
        assert key not in seen
        seen.add(key)


@pytest.mark.parametrize("module,layers,extra,n_tok", [("apm", 2, ["--chunks", "2", "--frames", "100"], 20), ("vpm", 1, ["--chunks", "1", "--size", "224x224"], 64)])
def test_synthetic_module_files_load_in_the_reference_encoders(tmp_path, module, layers, extra, n_tok):
    """tools/make_synth_omni_gguf.py writes what tools/omni/audition.cpp / vision.cpp load: the reference's own loader + graph builder run it on the CPU backend."""
    if not os.path.exists(ENC):
        pytest.skip("oracle/_ref/omni-enc-min not built (make -f oracle/Makefile.ref omni)")
    g = str(tmp_path / "m.gguf")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_omni_gguf.py"), "--module", module, "--layers", str(layers), "-o", g], check=True, timeout=300)
    env = dict(os.environ); env.pop("GGML_BACKEND_PATH", None)
    outs = []
    for run in range(2):
        o = str(tmp_path / f"o{run}.bin")
        r = subprocess.run([ENC, module, g, o, "--threads", "4"] + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
        j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        assert j["tokens"] == n_tok and j["n_embd"] == 4096
        outs.append(np.fromfile(o, np.float32))
    assert outs[0].size == n_tok * 4096 and np.isfinite(outs[0]).all() and float(outs[0].std()) > 0.01
    assert np.array_equal(outs[0], outs[1])                       # deterministic inputs, deterministic reference


def test_graph_diff_tool_counts_compute_nodes_and_ignores_layout_nodes(tmp_path):
    a, b = tmp_path / "a.dump", tmp_path / "b.dump"
    a.write_text("graph 0 nodes 4\n28 -1 t0 [8,4,1,1] | t1 [16,8,1,1] | t0 [16,4,1,1]\n35 -1 t0 [32,1,1,1] | t0 [8,4,1,1]\n2 -1 t0 [8,4,1,1] | t0 [8,4,1,1] | t0 [8,1,1,1]\n34 -1 t0 [8,4,1,1] | t0 [8,4,1,1]\n")
    b.write_text("graph 0 nodes 2\n28 -1 t0 [8,4,1,1] | t1 [16,8,1,1] | t0 [16,4,1,1]\n2 -1 t0 [8,4,1,1] | t0 [8,4,1,1] | t0 [8,1,1,1]\n")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "graph_diff.py"), str(a), str(b)], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "A: 3 compute nodes" in r.stdout and "B: 2 compute nodes" in r.stdout
    assert "identical (op, types, shapes) nodes: 2; only in A: 1; only in B: 0" in r.stdout and "CONT" in r.stdout


def test_synthetic_token2wav_set_runs_in_the_reference_module_on_cpu(tmp_path):
    """tools/make_synth_omni_gguf.py --module t2w writes what tools/omni/token2wav/token2wav-impl.cpp binds (every tensor name and layout of the conformer encoder, the DiT,
    the flow extras and the HiFT vocoder, plus the prompt bundle): the reference's own Token2WavSession sets up the prompt caches and produces one window of audio on the CPU backend."""
    t2w = os.path.join(ROOT, "oracle", "_ref", "t2w-min")
    if not os.path.exists(t2w):
        pytest.skip("oracle/_ref/t2w-min not built (make -f oracle/Makefile.ref omni)")
    d = str(tmp_path / "t2w")
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_omni_gguf.py"), "--module", "t2w", "--prompt-tokens", "28", "-o", d], check=True, timeout=600)
    env = dict(os.environ); env.pop("GGML_BACKEND_PATH", None)
    o = str(tmp_path / "w.f32")
    r = subprocess.run([t2w, d, o, "cpu", "--windows", "1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    w = np.fromfile(o, np.float32)
    assert j["samples"] == w.size and w.size >= 24000 and np.isfinite(w).all() and float(w.std()) > 0.01 and float(np.abs(w).max()) <= 1.1
