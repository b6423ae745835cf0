"""CPU tests of the host-side mirror (ctypes graph builder + Qwen3 graph): the graphs it emits are executed by the
REAL reference CPU backend (oracle/_ref) and must reproduce the committed golden vectors -- which validates the
mirror, the fixtures and the generator script at once.  Plus the world_size-2 gloo test of bench.py's replica logic."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT, nmse


def tiny_weights(tm):
    w = {}
    for k in tm.files:
        if k.startswith("w_"):
            _, il, name = k.split("_", 2)
            w[(int(il), name)] = tm[k]
    return w


def run_tiny(pkg, backend, tm, steps, flash_attn=True):
    from llama_cpp_omni_amd import qwen3
    cfg = qwen3.TINY
    mdl = qwen3.Model(backend, cfg, qwen3.q4_k_m_types(cfg), n_ctx=256, flash_attn=flash_attn, weights=tiny_weights(tm))
    table = tm["table"].view(np.float16)
    g, I, logits = mdl.build(1, 256)
    gr = g.graph()
    tok, toks, l = 1, [], None
    for step in range(steps):
        mdl.set_inputs(I, table[tok].astype(np.float32)[None, :], step, 256)
        backend.graph_compute(gr)
        l = backend.tensor_get(logits).copy()
        tok = int(np.argmax(l))
        toks.append(tok)
    g.free()
    mdl.wctx.free()
    return toks, l


def test_tiny_model_fixture_reproduces_on_reference_cpu(pkg, ref_be, golden):
    tm = golden["tiny_model"]
    toks, l = run_tiny(pkg, ref_be, tm, 32)
    assert toks == list(tm["tokens"])
    assert nmse(l, tm["final_logits"]) < 1e-10


def test_graph_node_sequence_matches_llm_build_qwen3(pkg, ref_be):
    """Per layer the reference emits: norm,mul, 3 mul_mat, (q) norm,mul,rope, (k) norm,mul,rope, 2 set_rows, fattn, mul_mat, add,
    norm,mul, 2 mul_mat, glu, mul_mat, add   -- plus views/reshapes/permutes."""
    from llama_cpp_omni_amd import qwen3
    OP = pkg.OP
    cfg = qwen3.TINY
    mdl = qwen3.Model(ref_be, cfg, qwen3.q4_k_m_types(cfg), n_ctx=256)
    g, I, logits = mdl.build(1, 256)
    # node order = ggml_build_forward_expand() on q_cur, k_cur, v_cur, cpy_k, cpy_v (build_attn, llama-graph.cpp:1559-1575), then the result:
    # each chain is visited depth-first, so wk's MUL_MAT comes after the whole q chain
    ops = [n.t.op for n in g.graph().nodes if n.t.op not in (OP.VIEW, OP.RESHAPE, OP.PERMUTE, OP.TRANSPOSE, OP.NONE)]
    layer = [OP.RMS_NORM, OP.MUL, OP.MUL_MAT, OP.RMS_NORM, OP.MUL, OP.ROPE, OP.MUL_MAT, OP.RMS_NORM, OP.MUL, OP.ROPE, OP.MUL_MAT,
             OP.SET_ROWS, OP.SET_ROWS, OP.FLASH_ATTN_EXT, OP.MUL_MAT, OP.ADD, OP.RMS_NORM, OP.MUL, OP.MUL_MAT, OP.MUL_MAT, OP.GLU, OP.MUL_MAT, OP.ADD]
    assert ops == layer * cfg["n_layer"] + [OP.RMS_NORM, OP.MUL, OP.MUL_MAT]
    assert len(layer) == 23
    g.free()
    mdl.wctx.free()


def test_q4_k_m_type_map():
    from conftest import load_pkg
    load_pkg()
    from llama_cpp_omni_amd import qwen3
    t = qwen3.q4_k_m_types(qwen3.QWEN3_8B)
    hi = [i for i in range(36) if t[i]["ffn_down"] == 14]
    assert hi == [0, 1, 2, 3, 6, 9, 12, 15, 18, 21, 24, 27, 30, 31, 32, 33, 34, 35]
    assert all(t[i]["attn_v"] == t[i]["ffn_down"] for i in range(36)) and t["output"] == 14


def test_bench_replica_logic_world2_gloo(tmp_path):
    """bench.py's N > 1 path is its `Replicas` class: rank / device from the launcher's env, barrier + synchronize on both sides of exactly K
    timed steps, MAX over ranks of the elapsed time, value = units of all ranks / that time.  Two gloo ranks on CPU run THAT code (imported
    from bench.py) around a stand-in step of different length per rank; the GPU form of the same launch is
    tests/test_round2_gpu.py::test_bench_runs_as_two_ranks_through_its_gloo_hooks."""
    script = tmp_path / "w2.py"
    script.write_text(
        "import os, sys, time\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "rep = bench.Replicas()\n"
        "assert rep.world == 2 and rep.backend == 'gloo' and rep.dev_index == 0\n"
        "calls, syncs = [], []\n"
        "def step(pos): calls.append(pos); time.sleep(0.02 * (rep.rank + 1))\n"
        "dt, pos = rep.timed(step, 5, 2, lambda: syncs.append(1))\n"
        "assert calls == list(range(7)) and pos == 7 and len(syncs) == 3        # barrier + synchronize on both sides, + the rank's own synchronize that stamps its per-rank time\n"
        "ev = rep.evidence(5, rep.dt_local)\n"
        "assert ev['rccl_world'] == 2 and [r['rank'] for r in ev['per_rank']] == [0, 1] and len({r['pid'] for r in ev['per_rank']}) == 2\n"
        "assert ev['per_rank'][0]['tok_s'] > ev['per_rank'][1]['tok_s'] > 0        # each rank's own rate over the timed region: rank 1's steps are twice as long\n"
        "if rep.rank == 0: print('VALUE', rep.aggregate(5, dt), dt)\n"
        "rep.finish()\n")
    env = dict(os.environ, MI355X_BENCH_DIST_BACKEND="gloo", MI355X_BENCH_SHARE_GPU="1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29533", str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("VALUE")][0].split()
    dt = float(line[2])
    assert 0.2 <= dt < 0.5                                        # MAX over ranks: the slow rank's 5 x 0.04 s, not rank 0's 5 x 0.02 s
    assert abs(float(line[1]) - 2 * 5 / dt) < 1e-6                # whole-job aggregate = units of all ranks / max time


def test_omni_runtime_device_map_uses_the_reference_knobs():
    """tools/omni_runtime.sh --map ... --print: one module per GPU translated to the reference runtime's own knobs and nothing else -- common_params.main_gpu / split_mode
    (common/arg.cpp:2906, 2955) for the LLM, MTMD_BACKEND_DEVICE (audition.cpp:241), Omni_BACKEND_DEVICE (vision.cpp:201), omni_init's token2wav_device
    (omni.cpp:3775); a TTS device that differs from the LLM's is refused with a pointer to the maintainer patch (omni.cpp:3457 loads it with the LLM's params)."""
    import os, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sh = os.path.join(root, "tools", "omni_runtime.sh")
    r = subprocess.run(["bash", sh, "--map", "llm=0,tts=1,t2w=2,apm=3,vpm=4", "--print", "16", "2", "omni"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    line = r.stdout.strip().splitlines()[-1]
    assert " -mg 0 -sm none " in line and "--t2w-device gpu:2" in line and "MTMD_BACKEND_DEVICE=MI355X3" in line and "Omni_BACKEND_DEVICE=MI355X4" in line, line
    assert "--test case/audio_ 2" in line and "--max-tgt 16" in line and line.rstrip().endswith("--omni"), line
    assert "GGML_BACKEND_PATH=" in line and "libggml-mi355x.so" in line
    assert "tts=1 ignored" in r.stderr and "omni.cpp:3457" in r.stderr
    r = subprocess.run(["bash", sh, "--print"], capture_output=True, text=True, timeout=60)                # no map: the one-GPU form, unchanged
    line = r.stdout.strip().splitlines()[-1]
    assert "-mg" not in line and "BACKEND_DEVICE" not in line and "--t2w-device gpu:0" in line, line
    r = subprocess.run(["bash", sh, "--map", "dsp=1", "--print"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 2 and "unknown module" in r.stderr
