#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X ggml backend (contract: see the repo prompt / DESIGN.md).

Metric (BASELINE.json): llama-bench tg128-style decode tokens/s, Qwen3-8B Q4_K_M shapes, batch 1, per GPU
replica.  A "step" = one decoded token = one `graph_compute` of the llm_build_qwen3 decode graph (36 layers +
lm head, 4670.5 MB of quantised weights streamed from HBM) submitted through the ggml backend C-ABI of
libggml-mi355x.so, plus the per-token input upload and logits read-back that llama_decode performs.
Weights / KV cache are resident in HBM before the timed region.  N > 1: N independent replicas (one
process per GPU, no data-path collective -- decode of one sequence does not shard).

Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
MFMA_F16_PEAK_TFLOPS = 2500.0    # dense F16/BF16 MFMA peak (MI355X_MICROARCH.md: ~2.5 PF dense; 2:1-sparsity figures are not used)
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def load_pkg():
    name = "llama_cpp_omni_amd"
    if name in sys.modules:
        return sys.modules[name]
    d = os.path.join(ROOT, "llama.cpp-omni_amd")
    spec = importlib.util.spec_from_file_location(name, os.path.join(d, "__init__.py"), submodule_search_locations=[d])
    m = importlib.util.module_from_spec(spec)
    sys.modules[name] = m
    spec.loader.exec_module(m)
    return m


class Decoder:
    """llama-bench test_gen analogue: n_gen x { decode(1 token); synchronize } at growing depth."""

    def __init__(self, pkg, be, cfg, types, n_ctx, n_kv, flash_attn=True, seed=1234, share=True, host_copy=False, pinned=True):
        from llama_cpp_omni_amd import qwen3
        self.be, self.n_kv = be, n_kv
        self.model = qwen3.Model(be, cfg, types, n_ctx=n_ctx, seed=seed, share_layer_bytes=share, flash_attn=flash_attn, host_copy=host_copy)
        self.g, self.I, self.logits = self.model.build(1, n_kv)
        self.graph = self.g.graph()
        E, V = cfg["n_embd"], cfg["n_vocab"]
        self.fa = flash_attn
        alloc = (lambda n: be.host_array(n)) if pinned else (lambda n: np.empty(n, np.uint8))
        self.h_embd = alloc(E * 4).view(np.float32)
        self.h_pos = alloc(4).view(np.int32)
        self.h_idx = alloc(8).view(np.int64)
        self.nv = cfg["n_head_kv"] * cfg["head_dim"] if self.model.v_trans else 0       # flash-attention off: the v store scatters single elements
        self.n_ctx = n_ctx
        self.h_vidx = alloc(8 * max(self.nv, 1)).view(np.int64)
        self.v_base = np.arange(max(self.nv, 1), dtype=np.int64) * n_ctx
        msz = 2 if flash_attn else 4
        self.h_mask = alloc(n_kv * msz).view(np.float16 if flash_attn else np.float32)   # row 0 of the padded mask
        self.h_logits = alloc(V * 4).view(np.float32)
        self.mask_upto = -1
        rng = np.random.default_rng(seed + 1)
        self.embd_pool = (rng.standard_normal((64, E)) * 1.0).astype(np.float32)
        # rows 1..63 of the padded mask stay -inf for the whole run
        npad = self.I["kq_mask"].ne[1]
        full = np.full((npad, n_kv), -np.inf, dtype=np.float16 if flash_attn else np.float32)
        be.tensor_set(self.I["kq_mask"], full)

    def step(self, pos, fetch_logits=True):
        be, I = self.be, self.I
        self.h_embd[:] = self.embd_pool[pos % 64]
        self.h_pos[0] = pos
        self.h_idx[0] = pos
        if pos == self.mask_upto:                                # causal mask row of the new token: one more visible cell than the last step's
            self.h_mask[pos] = 0.0
        else:
            self.h_mask[:] = -np.inf
            self.h_mask[: pos + 1] = 0.0
        self.mask_upto = pos + 1
        be.tensor_set_async(I["inp_embd"], self.h_embd)
        be.tensor_set_async(I["inp_pos"], self.h_pos)
        be.tensor_set_async(I["k_idxs"], self.h_idx)
        if self.nv:
            self.h_vidx[:] = self.v_base + pos
            be.tensor_set_async(I["v_idxs"], self.h_vidx)
        else:
            be.tensor_set_async(I["v_idxs"], self.h_idx)
        be.tensor_set_async(I["kq_mask"], self.h_mask)
        be.graph_compute(self.graph)
        if fetch_logits:
            be.tensor_get_async(self.logits, self.h_logits)
        be.synchronize()


def _cpu_order(allowed):
    """The allowed logical CPUs in the order threads should be added: one per physical core first, the NUMA node this thread runs on (where its
    allocations are first-touched) before the other nodes, round-robin over the node's L3 domains (a CCD's link to memory carries ~60 GB/s: eight
    threads on one CCD get less than eight threads on eight), hyper-thread siblings last."""
    import ctypes
    import glob
    try:
        here = ctypes.CDLL(None).sched_getcpu()
    except Exception:
        here = allowed[0]
    info = {}
    for c in allowed:
        base = f"/sys/devices/system/cpu/cpu{c}"
        try:
            pkg_id = int(open(base + "/topology/physical_package_id").read()); core = int(open(base + "/topology/core_id").read())
        except Exception:
            pkg_id, core = 0, c
        nodes = glob.glob(base + "/node*")
        node = int(os.path.basename(nodes[0])[4:]) if nodes else pkg_id
        try:
            l3 = open(base + "/cache/index3/shared_cpu_list").read().strip()
        except Exception:
            l3 = str(pkg_id)
        info[c] = (node, pkg_id, core, l3)
    home = info.get(here, info[allowed[0]])
    first, rest, seen = [], [], set()
    node_list = sorted({v[0] for v in info.values()}, key=lambda n: (n != home[0], n))
    for n in node_list:
        groups = {}
        for c in sorted(allowed):
            if info[c][0] == n:
                groups.setdefault(info[c][3], []).append(c)
        cores, sibs = [], []
        for g in groups.values():                                  # per L3 domain: one CPU per physical core, then its siblings
            gc, gs = [], []
            for c in g:
                key = info[c][1:3]
                (gs if key in seen else gc).append(c)
                seen.add(key)
            cores.append(gc); sibs.append(gs)
        for lst, dst in ((cores, first), (sibs, rest)):
            i = 0
            while any(lst):
                if lst[i % len(lst)]:
                    dst.append(lst[i % len(lst)].pop(0))
                i += 1
    return first + rest, {"home_node": home[0], "nodes": len(node_list), "l3_domains": len({v[3] for v in info.values()}), "physical_cores": len(first)}


def _cpu_baseline_worker(pkg, cfg, types, n_kv, th, seconds):
    """One thread count, in a process of its own (started by cpu_baseline with OMP_PLACES / OMP_PROC_BIND in the environment, so the reference
    backend's OpenMP team is pinned one thread per listed CPU from its first parallel region)."""
    sys.path.insert(0, ROOT)
    from oracle.ref_backend import make_ref_cpu_backend, ref_variant
    be = make_ref_cpu_backend(pkg, th)
    dec = Decoder(pkg, be, cfg, types, n_ctx=n_kv, n_kv=n_kv, flash_attn=True, pinned=False)
    wbytes = dec.model.weight_bytes()
    t0 = time.perf_counter()
    dec.step(0)                                                  # warm-up (page-in, team start)
    pilot = time.perf_counter() - t0
    t0 = time.perf_counter()
    dec.step(1)
    pilot = min(pilot, time.perf_counter() - t0)
    n, pos = 0, 2
    t0 = time.perf_counter()
    while n < 48 and (time.perf_counter() - t0) < seconds and (n < 2 or pilot < 2.0):
        dec.step(pos); pos += 1; n += 1
    dt = time.perf_counter() - t0
    be.close()
    return {"tok_s": n / dt if n else 1.0 / pilot, "steps": n, "wbytes": wbytes, "build": ref_variant()[0]}


def _cgroup_cpu_quota():
    """CPUs' worth of time the container's cgroup grants this process (cpu.max of cgroup v2, cfs_quota / cfs_period of v1), None when unlimited or unreadable:
    a team wider than the quota is throttled, whatever sched_getaffinity says."""
    try:
        here = "/sys/fs/cgroup"
        if os.path.exists(here + "/cpu.max"):
            q, per = open(here + "/cpu.max").read().split()[:2]
            return None if q == "max" else round(int(q) / int(per), 2)
        q = int(open(here + "/cpu/cpu.cfs_quota_us").read()); per = int(open(here + "/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else round(q / per, 2)
    except Exception:
        return None


def cpu_baseline(pkg, cfg, types, n_kv, seconds_budget=24.0, tiny=False):
    """Reference CPU backend (oracle/_ref, built from /root/reference sources) on the SAME decode graph, timed on this box's host
    cores for a bounded sample.  Thread counts come from the cores this process may actually run on (sched_getaffinity: a cgroup /
    affinity mask smaller than the machine would otherwise be oversubscribed); an ascending sweep up to that count is timed and the best is
    reported together with the bandwidth it implies.  Each count runs in its own process with the OpenMP team pinned one thread per CPU
    (OMP_PLACES lists the CPUs in _cpu_order: one per physical core, this process's NUMA node first, round-robin over its L3 domains):
    a team floating over all allowed CPUs peaked at 16 threads / 175-195 GB/s on the 2 x 64-core box and lost half of that at 32."""
    import subprocess
    try:
        sys.path.insert(0, ROOT)
        from oracle.ref_backend import ref_available
        if not ref_available():
            return None
        allowed = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
        n_aff = len(allowed)
        order, topo = _cpu_order(allowed)
        n_phys = topo["physical_cores"]
        per_node = max(1, n_phys // max(1, topo["nodes"]))
        pin = os.environ.get("MI355X_CPU_BASELINE_NO_PIN") is None
        quota = _cgroup_cpu_quota()
        cands = sorted({c for c in (8, 16, 24, 32, 48, 64, per_node, n_phys) if 1 <= c <= n_phys})
        if quota is not None:                                      # the sweep still runs past the quota once (the line shows the throttled point), not beyond
            lim = max(1, int(quota))
            cands = sorted({c for c in cands if c <= lim} | {min(lim, n_phys)} | ({min(c for c in cands if c > lim)} if any(c > lim for c in cands) else set()))
        results, build, wbytes = [], None, None
        t_start = time.perf_counter()
        for th in cands:
            env = dict(os.environ)
            env["ORACLE_REF_VARIANT"] = "v4"                      # the AVX-512 build of the same sources when this host has the level (oracle/ref_backend.py)
            env["OMP_NUM_THREADS"] = str(th)
            if pin:
                env["OMP_PLACES"] = ",".join("{%d}" % c for c in order[:th]); env["OMP_PROC_BIND"] = "true"
            cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(th), "--cpu-baseline-seconds", str(seconds_budget / 4)] + (["--tiny"] if tiny else [])
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                results.append((0.0, th, 0)); break
            d = json.loads(r.stdout.strip().splitlines()[-1])
            build, wbytes = d["build"], d["wbytes"]
            results.append((d["tok_s"], th, d["steps"]))
            # ascending sweep through 32 / 48 / 64 threads too (the reference's per-node barrier makes them slower than 16 on this box: the line carries the numbers),
            # stopped only on a collapse (an oversubscribed team of spinning workers can be 100x slower) or when the budget is used up
            if len(results) >= 2 and results[-1][0] < 0.4 * max(results[:-1])[0]:
                break
            if time.perf_counter() - t_start > 4 * seconds_budget:
                break
        best = max(results)
        v3 = None
        if build == "x86-64-v4":                                   # the same point on the AVX2 build (what the parity tests run): both figures are reported
            env = dict(os.environ); env["ORACLE_REF_VARIANT"] = "v3"; env["OMP_NUM_THREADS"] = str(best[1])
            if pin:
                env["OMP_PLACES"] = ",".join("{%d}" % c for c in order[:best[1]]); env["OMP_PROC_BIND"] = "true"
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-worker", str(best[1]), "--cpu-baseline-seconds", str(seconds_budget / 4)] + (["--tiny"] if tiny else []),
                               env=env, capture_output=True, text=True, timeout=300)
            if r.returncode == 0:
                v3 = round(json.loads(r.stdout.strip().splitlines()[-1])["tok_s"], 3)
        return {"value": round(best[0], 3), "unit": "tok/s", "cores": best[1], "kind": "reference", "value_x86_64_v3_build": v3,
                "gb_per_s": round(best[0] * wbytes / 1e9, 1) if wbytes else None,
                "allowed_cpus": n_aff, "physical_cores_allowed": n_phys, "cgroup_cpu_quota": quota,
                "cgroup_note": ("no cgroup CPU quota: the thread sweep is limited by the affinity mask only" if quota is None else
                                f"the cgroup grants {quota} CPUs' worth of time: teams wider than that are throttled (the sweep's falling tail), so this baseline is the container's, not the host's"),
                "thread_sweep_tok_s": {str(th): round(v, 3) for v, th, _ in results},
                "build": build, "placement": ("one pinned thread per CPU (OMP_PLACES): one per physical core, home NUMA node first, round-robin over its L3 domains" if pin else "unpinned"), "topology": topo,
                "sample": f"{best[2]} decode steps of the same Qwen3-8B Q4_K_M graph on the reference ggml CPU backend (oracle/_ref, {build} build, "
                          f"its own OpenMP team), best of the thread sweep over the {n_phys} physical cores this process is allowed on"}
    except Exception as e:  # the baseline is a reported extra, never a reason to lose the GPU number
        return {"value": None, "unit": "tok/s", "cores": 0, "kind": "reference", "sample": f"failed: {e!r}"}


def kernel_roofline(pkg, be, model, reps=5):
    """Live roofline of the dominant kernel, mi::k_mv2<1,1,true,true> (mmv2.hip, the LDS-DMA loader / consumer engine: RMS norm + Q8_K image in
    the prologue, ffn_gate + ffn_up + SWIGLU: 2 x 12288 x 4096 Q4_K rows = 56.6 MB per launch, 36 launches and 2.04 of the 4.67 GB of every decoded token;
    MI355X_MV2=0 runs the register-load form mi::k_mv1<8,2,2,1,1,true,...> in its place).  One cgraph holding the 36 launches of one
    decode step -- the real layers' weights, 2 GB, 8x the Infinity Cache -- is replayed as a hipGraph and bracketed by two HIP
    events on the backend's stream; avg launch = elapsed / 36 (so it includes the launch-to-launch boundary, like the
    per-dispatch duration rocprofv3 --kernel-trace reports; profiles/).  achieved = algorithmic weight bytes / time."""
    from llama_cpp_omni_amd.ggml import GGML_TYPE_F32, GGML_TYPE_Q4_K, Context
    cfg = model.cfg
    c = Context(be)
    x = c.new_tensor(GGML_TYPE_F32, cfg["n_embd"], 1)
    nbytes, launches = 0, 0
    for L in model.layers:
        if L["ffn_up"].type != GGML_TYPE_Q4_K or L["ffn_gate"].type != GGML_TYPE_Q4_K:
            continue
        xn = c.mul(c.rms_norm(x, cfg["rms_eps"]), model._w(c, L["ffn_norm"]))     # the launch as the step runs it: ffn_norm folded into its prologue
        up = c.mul_mat(model._w(c, L["ffn_up"]), xn)
        gate = c.mul_mat(model._w(c, L["ffn_gate"]), xn)
        c.swiglu_split(gate, up)
        nbytes += L["ffn_up"].nbytes() + L["ffn_gate"].nbytes()
        launches += 1
    if launches == 0:
        return None
    c.alloc()
    be.tensor_set(x, np.random.default_rng(0).standard_normal(cfg["n_embd"]).astype(np.float32))
    g = c.graph()
    for _ in range(3):
        be.graph_compute(g)                      # eager, capture, first replay
    be.synchronize()
    kern = int(be.get_stat("kernels_last_graph"))
    assert kern == launches, (kern, launches)           # exactly the pair launches: norm, quantisation and SwiGLU are inside them
    best = 1e30
    for _ in range(reps):
        a, b = be.timed_event(), be.timed_event()
        be.record(a); be.graph_compute(g); be.record(b)
        best = min(best, be.elapsed_ms(a, b))
    c.free()
    us = best * 1e3
    ach = nbytes / us / 1e3
    # HBM traffic per launch: PMC counters cannot be read from inside this process.  The figure comes from the committed rocprofv3
    # --pmc FETCH_SIZE pass of this same command (tools/profile_round.sh -> profiles/<round>_pmc_fetch_size.json, corrected x2 per
    # MI355X_MICROARCH.md); it is used only when that file names THIS kernel, and the file / kernel symbol are printed beside it
    traffic, traffic_src = None, None
    mv2 = os.environ.get("MI355X_MV2", "1") != "0"
    try:
        cands = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_fetch_size.json"))
        for fn in reversed(cands):
            pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))
            for k, v in pmc["kernels"].items():
                if ("k_mv2<1, 1, true" in k) if mv2 else ("k_mv1<8, 2, 2, 1, 1, true" in k):
                    traffic = v["hbm_bytes_per_dispatch_corrected"]
                    traffic_src = {"file": "profiles/" + fn, "kernel": k[:80], "commit": pmc.get("commit")}
            if traffic is not None:
                break
    except Exception:
        pass
    kname = "mi::k_mv2<1,1,true,true,16> (LDS-DMA engine, 16 waves: " if mv2 else "mi::k_mv1<8,2,2,1,1,true,false,false> ("
    return {"bound": "hbm", "kernel": kname + "RMS norm + Q8_K image prologue, Q4_K ffn_gate+ffn_up mat-vec, SWIGLU epilogue)",
            "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
            "bytes_per_launch": nbytes // launches, "avg_launch_us": round(us / launches, 3), "launches": launches,
            "method": "hipGraph replay of the step's 36 launches of this kernel, two HIP events on the backend stream, best of 5"}


def via_libllama(threads=8, reps=5):
    """The same metric through the REFERENCE's own libllama + ggml_backend_sched with this backend loaded as a plug-in (GGML_BACKEND_PATH):
    oracle/_ref/llama-bench-min (the reference's llama-bench measurement loops on the public llama.h API, tools/llama_bench_min.cpp) on a
    synthetic Qwen3-8B Q4_K_M GGUF written by tools/make_synth_gguf.py -- pp512 and tg128, flash-attention off (llama-bench's default) and on.
    Only when oracle/_ref travelled with the snapshot (checker-side binaries; the product library is what is measured).  The difference to
    `value` is host work outside the plug-in that llama_decode + llama_synchronize serialise with the device: graph build / scheduling, the CPU
    split that looks the token up (token_embd stays on the CPU), and six blocking input copies per step (ggml_backend_tensor_copy)."""
    import shutil
    import subprocess
    import tempfile
    binp = os.path.join(ROOT, "oracle", "_ref", "llama-bench-min")
    lib = os.path.join(ROOT, "llama.cpp-omni_amd", "lib", "libggml-mi355x.so")
    if not os.path.exists(binp):
        return None
    tmp = tempfile.mkdtemp(prefix="mi355x_bench_")
    try:
        if shutil.disk_usage(tmp).free < 7e9:
            return {"error": "needs 5 GB of scratch disk for the synthetic GGUF"}
        gguf = os.path.join(tmp, "q8b.gguf")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_gguf.py"), "--config", "8b", "--types", "q4_k_m", "-o", gguf], check=True, timeout=900,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        env = dict(os.environ); env["GGML_BACKEND_PATH"] = lib
        out = {"harness": "oracle/_ref/llama-bench-min (reference libllama + ggml_backend_sched, plug-in from GGML_BACKEND_PATH), -ngl 99 -p 512 -n 128 "
                          f"-r {reps} -t {threads}, synthetic Qwen3-8B Q4_K_M GGUF"}
        for fa in (0, 1):
            r = subprocess.run([binp, "-m", gguf, "-ngl", "99", "-fa", str(fa), "-p", "512", "-n", "128", "-r", str(reps), "-t", str(threads)],
                               env=env, capture_output=True, text=True, timeout=900)
            if r.returncode != 0:
                out[f"fa{fa}"] = {"error": r.stderr[-300:]}
                continue
            res = {}
            for line in r.stdout.strip().splitlines():
                j = json.loads(line)
                res[j["test"] + "_tok_s"] = j["avg_ts"]; res[j["test"] + "_stddev"] = j["stddev_ts"]
            splits = [ln for ln in r.stderr.splitlines() if "graph splits" in ln]
            res["graph_splits"] = int(splits[-1].split("=")[-1]) if splits else None
            out[f"fa{fa}"] = res
        # where the difference to `value` sits: the plug-in's own host-time account of a tg128-only run at llama-bench's default (-fa 0), from MI355X_LOG_STATS
        try:
            import re
            env2 = dict(env); env2["MI355X_LOG_STATS"] = "1"
            r = subprocess.run([binp, "-m", gguf, "-ngl", "99", "-fa", "0", "-p", "0", "-n", "128", "-r", "3", "-t", str(threads)], env=env2, capture_output=True, text=True, timeout=600)
            acc = {}
            for ln in r.stderr.splitlines():
                if "host time inside the backend" in ln:
                    for key, name in (("graph_compute", "graph_compute"), ("set_tensor_async", "set_tensor_async"), ("get_tensor_async", "get_tensor_async"), ("synchronize", "synchronize")):
                        m = re.search(name + r" (\d+) calls ([0-9.]+) us avg", ln)
                        if m:
                            acc[key] = {"calls": int(m.group(1)), "us_avg": float(m.group(2))}
                if "blocking buffer transfers" in ln:
                    m = re.search(r"tensor_set (\d+) calls ([0-9.]+) MB ([0-9.]+) ms, tensor_get (\d+) calls ([0-9.]+) MB ([0-9.]+) ms", ln)
                    if m:
                        acc["blocking_tensor_set"] = {"calls": int(m.group(1)), "ms_total": float(m.group(3))}; acc["blocking_tensor_get"] = {"calls": int(m.group(4)), "ms_total": float(m.group(6))}
            ng = (acc.get("graph_compute") or {}).get("calls")
            if ng:
                # one graph_compute per decoded token (+ warm-up): per-token host microseconds inside the plug-in; `synchronize` is mostly waiting for the device
                acc["per_token_us"] = {k: round(v["calls"] * v["us_avg"] / ng, 1) for k, v in acc.items() if isinstance(v, dict) and "us_avg" in v}
                tg = [json.loads(x) for x in r.stdout.strip().splitlines() if x.startswith("{")]
                if tg:
                    acc["tg128_tok_s_this_run"] = tg[-1].get("avg_ts")
            out["plugin_host_account_tg128_fa0"] = acc or None
        except Exception as e:
            out["plugin_host_account_tg128_fa0"] = {"error": repr(e)}
        return out
    except Exception as e:
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def via_reference_omni_modules():
    """BASELINE configs[3] / [4] legs through the REFERENCE's own module code with this backend loaded as a plug-in: oracle/_ref/omni-enc-min (tools/omni/audition.cpp,
    vision.cpp) and oracle/_ref/t2w-min (tools/omni/token2wav/token2wav-impl.cpp) on the full-size synthetic module files of tools/make_synth_omni_gguf.py.  Wall time as the
    caller of audition_audio_encode / vision_image_encode / Token2WavSession::feed_window sees it (host + device): one streaming second of audio through Whisper-medium,
    one 448 x 448 slice through SigLip2 + resampler, and the Token2Wav real-time factor (seconds of compute per second of 24 kHz audio, windows after the first).
    Checker-side binaries, present only when oracle/_ref travelled with the snapshot; the product library is what they run on."""
    import shutil
    import subprocess
    import tempfile
    enc, t2w = os.path.join(ROOT, "oracle", "_ref", "omni-enc-min"), os.path.join(ROOT, "oracle", "_ref", "t2w-min")
    lib = os.path.join(ROOT, "llama.cpp-omni_amd", "lib", "libggml-mi355x.so")
    if not (os.path.exists(enc) and os.path.exists(t2w)):
        return None
    tmp = tempfile.mkdtemp(prefix="mi355x_omni_")
    gen = os.path.join(ROOT, "tools", "make_synth_omni_gguf.py")
    env = dict(os.environ); env["GGML_BACKEND_PATH"] = lib; env["MTMD_BACKEND_DEVICE"] = "MI355X0"
    out = {"harness": "oracle/_ref/omni-enc-min + oracle/_ref/t2w-min (reference audition.cpp / vision.cpp / token2wav-impl.cpp, plug-in from GGML_BACKEND_PATH), synthetic full-size module files"}

    def last_json(r):
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    try:
        for mod, args, key in (("apm", ["--chunks", "12", "--frames", "100"], "apm_whisper_1s_stream_chunk_ms"), ("vpm", ["--chunks", "4"], "vpm_siglip2_resampler_slice_ms")):
            g = os.path.join(tmp, mod + ".gguf")
            subprocess.run([sys.executable, gen, "--module", mod, "-o", g], check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            r = subprocess.run([enc, mod, g, os.path.join(tmp, mod + ".bin"), "--gpu"] + args, env=env, capture_output=True, text=True, timeout=600)
            out[key] = round(last_json(r)["ms_last_chunk"], 3) if r.returncode == 0 else {"error": r.stderr[-300:]}
            os.remove(g)
        d = os.path.join(tmp, "t2w")
        subprocess.run([sys.executable, gen, "--module", "t2w", "-o", d], check=True, timeout=600, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        r = subprocess.run([t2w, d, os.path.join(tmp, "w.f32"), "gpu", "--windows", "8"], env=env, capture_output=True, text=True, timeout=900)
        if r.returncode == 0:
            j = last_json(r)
            steady = j["ms_windows"][2:-1]                          # (window 0 carries the first-touch costs, window 1 the hipGraph capture of the window graph, the last one is the longer final window)
            out["t2w_rtf"] = round(sum(steady) / len(steady) / 1e3, 5); out["t2w_ms_per_1s_window"] = steady; out["t2w_rtf_all_windows"] = round(j["rtf"], 5)
        else:
            out["t2w_rtf"] = {"error": r.stderr[-300:]}
        return out
    except Exception as e:
        out["error"] = repr(e)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def via_reference_omni_runtime():
    """BASELINE configs[3] (C4 TTFT) through the REFERENCE's omni runtime itself: oracle/_ref/omni-min = tools/omni/omni.cpp (omni_init, stream_prefill,
    stream_decode and its LLM / TTS / Token2Wav threads) + audition.cpp + token2wav-impl.cpp + libllama, with this backend loaded from GGML_BACKEND_PATH, over the synthetic
    full-size module set of tools/make_synth_omni_set.py (one 2 s user turn after the system prompt with a 3 s reference voice).  Every number is the reference's own
    timestamp (tests/test_omni_runtime_gpu.py summarise()); the HiFT vocoder runs on the host CPU because omni.cpp:3779 pins it there.  Checker-side binary; absent -> None."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(ROOT, "oracle", "_ref", "omni-min")
    if not os.path.exists(exe):
        return None
    tmp = tempfile.mkdtemp(prefix="mi355x_omnirt_")
    out = {"harness": "oracle/_ref/omni-min (reference tools/omni/omni.cpp + modules + libllama, plug-in from GGML_BACKEND_PATH), synthetic full-size module set"}
    try:
        if shutil.disk_usage(tmp).free < 9e9:
            return {"skipped": "needs 7 GB of scratch disk"}
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import test_omni_runtime_gpu as T
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_synth_omni_set.py"), "-o", tmp], check=True, timeout=900, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        j = None
        for _ in range(2):                                         # (the first run pages the files in and pays the one-off image builds; the second is reported)
            shutil.rmtree(os.path.join(tmp, "out"), ignore_errors=True)
            r, log = T.run_omni_min(tmp, os.path.join(tmp, "out"), max_tgt=24, timeout=600)
            if r.returncode != 0:
                return dict(out, error=log[-300:])
            j = T.summarise(log)
        out.update({"llm_layers_on_plugin": "37/37" if "offloaded 37/37 layers to GPU" in log else None, "tts_layers_on_plugin": "21/21" if "offloaded 21/21 layers to GPU" in log else None,
                    "t2w_flow_backend": "MI355X0" if "init_backend device=gpu:0, gpu_idx=0, backend=MI355X0" in log else None,
                    "system_prompt_prefill_s": j["prefill_each_s"][0], "n_past_after_prefill": j["n_past_after_prefill"],
                    "first_audio_ms_reference_timestamp": j["reference_first_audio_ms"], "llm_decode_tok_s_in_runtime": round(j["llm_decode_tok_s"], 1) if j["llm_decode_tok_s"] else None,
                    "llm_prompt_tok_s_in_runtime": round(j["llm_prompt_tok_s"], 1) if j["llm_prompt_tok_s"] else None,
                    "t2w_token2mel_ms_per_window_on_plugin": j["t2w_token2mel_ms_median"], "t2w_vocoder_ms_per_window_on_host_cpu": j["t2w_vocoder_ms_median"], "wav_windows": j["n_wav"]})
        return out
    except Exception as e:
        out["error"] = repr(e)
        return out
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def prefill_tok_s(pkg, be, model, n_tokens=512, reps=3):
    """llama-bench pp512 analogue: one ubatch of 512 tokens at depth 0 through the same backend (MFMA GEMM path for the mat-muls)."""
    g, I, logits = model.build(n_tokens, n_tokens, n_outputs=1)
    gr = g.graph()
    rng = np.random.default_rng(5)
    model.set_inputs(I, rng.standard_normal((n_tokens, model.cfg["n_embd"])).astype(np.float32), 0, n_tokens)
    be.tensor_set(I["out_ids"], np.array([n_tokens - 1], np.int32))
    for _ in range(2):
        be.graph_compute(gr)
    be.synchronize()
    best = 1e30
    for _ in range(reps):
        t0 = time.perf_counter()
        be.graph_compute(gr)
        be.synchronize()
        best = min(best, time.perf_counter() - t0)
    ok = bool(np.isfinite(be.tensor_get(logits)).all())
    if os.environ.get("MI355X_BENCH_PROFILE"):
        be.set_option("profile", 1)
        be.set_option("reset_stats", 1)
        be.graph_compute(gr)
        be.synchronize()
        prof = {}
        for cls in ("gemm_f16", "dequant_f16", "mmv_q4k", "mmv_q6k", "act_convert", "rms_norm_mul_quant", "rms_norm_mul", "rms_norm", "norm_rope", "rope",
                    "fattn", "set_rows", "get_rows", "bin", "glu", "cpy", "soft_max", "dequant_f16", "empty"):
            u, k = be.get_stat(f"prof_{cls}_us"), be.get_stat(f"prof_{cls}_n")
            if k > 0:
                prof[cls] = {"n": int(k), "total_us": round(u, 1), "avg_event_to_event_us": round(u / k, 2)}
        be.set_option("profile", 0)
        sys.stderr.write(f"prefill {n_tokens} tokens, eager per-class profile: " + json.dumps(prof) + "\n")
    g.free()
    return n_tokens / best, ok


def c3_prefill(pkg, be, n_seq=8, n_prompt=2048, n_ubatch=512, tiny=False, one_ubatch=False):
    """BASELINE.json configs[2]: Qwen3-8B F16 prefill, 8 sequences x 2048 tokens (MFMA GEMM + flash-attn), llama-bench style:
    each prompt is fed as n_prompt / n_ubatch ubatches at growing KV depth; inputs are resident before the timed region."""
    from llama_cpp_omni_amd import qwen3
    cfg = qwen3.TINY if tiny else qwen3.QWEN3_8B
    if tiny:
        n_prompt, n_ubatch = 256, 64
    if one_ubatch:
        n_ubatch = n_seq * n_prompt
    model = qwen3.Model(be, cfg, qwen3.uniform_types(cfg, pkg.GGML_TYPE_F16), n_ctx=n_ubatch if one_ubatch else n_prompt, seed=77, share_layer_bytes=True, flash_attn=True)
    rng = np.random.default_rng(6)
    chunks = []
    if one_ubatch:
        # all sequences in ONE ubatch (llama-batched-bench -npp 2048 -npl 8 -ub 16384, unified KV cache): cells s * n_prompt + i, positions i,
        # block-diagonal causal mask -- the GEMMs see 16384 columns, the attention only the live blocks
        n_tok = n_seq * n_prompt
        g, I, logits = model.build(n_tok, n_tok, n_outputs=n_seq)
        model.set_inputs(I, rng.standard_normal((n_tok, cfg["n_embd"])).astype(np.float32), 0, n_tok, n_seq=n_seq)
        be.tensor_set(I["out_ids"], (np.arange(n_seq, dtype=np.int32) + 1) * n_prompt - 1)
        chunks.append((g, g.graph(), logits))
        reps = 1
    else:
        reps = n_seq
    for c in range(n_prompt // n_ubatch if reps == n_seq else 0):
        n_kv = (c + 1) * n_ubatch
        g, I, logits = model.build(n_ubatch, n_kv, n_outputs=1)
        model.set_inputs(I, rng.standard_normal((n_ubatch, cfg["n_embd"])).astype(np.float32), c * n_ubatch, n_kv)
        be.tensor_set(I["out_ids"], np.array([n_ubatch - 1], np.int32))
        chunks.append((g, g.graph(), logits))
    for _ in range(2):                                             # eager pass, then the hipGraph-capturing pass
        for _, gr, _ in chunks:
            be.graph_compute(gr)
    be.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for _, gr, _ in chunks:
            be.graph_compute(gr)
    be.synchronize()
    dt = time.perf_counter() - t0
    ok = bool(np.isfinite(be.tensor_get(chunks[-1][2])).all())
    if os.environ.get("MI355X_BENCH_PROFILE"):
        be.set_option("profile", 1)
        be.set_option("reset_stats", 1)
        for _, gr, _ in chunks:
            be.graph_compute(gr)
        be.synchronize()
        prof = {}
        for cls in ("gemm_f16", "gemm_reduce", "act_convert", "rms_norm_mul", "rms_norm", "norm_rope", "rope", "fattn", "set_rows", "get_rows", "bin", "glu", "cpy", "soft_max", "dequant_f16",
                    "mmv_f16", "empty"):
            u, k = be.get_stat(f"prof_{cls}_us"), be.get_stat(f"prof_{cls}_n")
            if k > 0:
                prof[cls] = {"n": int(k), "total_us": round(u, 1), "avg_event_to_event_us": round(u / k, 2)}
        be.set_option("profile", 0)
        sys.stderr.write(f"C3 one sequence ({n_prompt} tokens, ubatch {n_ubatch}), eager per-class profile: " + json.dumps(prof) + "\n")
    n_tok = n_seq * n_prompt
    # FLOPs as BASELINE.md row C3: 2 x layer weights per token + causal attention 4 * D * n_head * (n^2 / 2) per layer and sequence
    E, F, L = cfg["n_embd"], cfg["n_ff"], cfg["n_layer"]
    hd, nh, nkvh = cfg["head_dim"], cfg["n_head"], cfg["n_head_kv"]
    w_layer = E * hd * nh + 2 * E * hd * nkvh + hd * nh * E + 3 * E * F
    flops = n_tok * 2.0 * L * w_layer + n_seq * L * 4.0 * hd * nh * (n_prompt * (n_prompt + 1) / 2)
    for g, _, _ in chunks:
        g.free()
    model.wctx.free()
    return {"tok_s": round(n_tok / dt, 1) if ok else None, "ms_total": round(dt * 1e3, 2), "tflops": round(flops / dt / 1e12, 1),
            "frac_of_dense_f16_peak": round(flops / dt / 1e12 / MFMA_F16_PEAK_TFLOPS, 4), "n_seq": n_seq, "n_prompt": n_prompt, "n_ubatch": n_ubatch}


def extra_legs(pkg, be, headline_no_fa):
    """Measured beside the headline (same timing loop, 64 steps each; not part of `value`): the decode step with flash-attention OFF --
    llama-bench's default, libllama's transposed-V-cache graph -- and the omni TTS decoder at its real shape (Q8_0,
    llama architecture: 8(f) rank 2), both through the same C-ABI."""
    from llama_cpp_omni_amd import qwen3
    res = {}
    try:
        legs = [("tts_q8_0_decode", qwen3.TTS, qwen3.uniform_types(qwen3.TTS, pkg.GGML_TYPE_Q8_0), True)]
        if not headline_no_fa:
            legs.insert(0, ("qwen3_8b_q4_k_m_decode_no_fa", qwen3.QWEN3_8B, qwen3.q4_k_m_types(qwen3.QWEN3_8B), False))
        # BASELINE configs[4] ships the 8B LLM as Q8_0 (tools/omni/convert/run_convert.sh:67-70): the same decode step on all-Q8_0 weights (8.7 GB per token)
        legs.append(("qwen3_8b_q8_0_decode", qwen3.QWEN3_8B, qwen3.uniform_types(qwen3.QWEN3_8B, pkg.GGML_TYPE_Q8_0), True))
        for name, cfg, types, fa in legs:
            dec = Decoder(pkg, be, cfg, types, n_ctx=256, n_kv=256, flash_attn=fa, seed=4321)
            for p in range(8):
                dec.step(p)
            be.synchronize()
            t0 = time.perf_counter()
            for p in range(8, 72):
                dec.step(p)
            be.synchronize()
            dt = time.perf_counter() - t0
            ok = bool(np.isfinite(dec.h_logits).all())
            res[name] = {"tok_s": round(64 / dt, 1) if ok else None, "ms_per_step": round(dt / 64 * 1e3, 4), "kernels_per_token": be.get_stat("kernels_last_graph"),
                         "n_layer": cfg["n_layer"], "n_embd": cfg["n_embd"]}
            if name == "qwen3_8b_q8_0_decode" and ok:
                wb = dec.model.weight_bytes()
                res[name]["weight_bytes_per_token"] = wb
                res[name]["hbm_frac_whole_step"] = round(wb * (64 / dt) / 1e9 / HBM_PEAK_GBS, 4)
            dec.g.free(); dec.model.wctx.free()
    except Exception as e:  # extras never cost the headline
        res["error"] = repr(e)
    try:
        res["omni_modules"] = omni_module_legs(pkg, be)
    except Exception as e:
        res["omni_modules_error"] = repr(e)
    return res


def omni_module_legs(pkg, be):
    """SURVEY.md 8(f) ranks 3 / 4 measured beside the headline: one layer of each omni encoder and the Token2Wav blocks at their real shapes
    (the graphs of llama.cpp-omni_amd/encoders.py / token2wav.py, random weights), best of 5 submissions between two HIP events; the
    per-module figure is that layer time x the module's layer count (front ends / tails included once)."""
    from llama_cpp_omni_amd import encoders as E, token2wav as T
    rng = np.random.default_rng(99)

    def timed(c, tensors):
        c.alloc()
        for t in tensors:
            n = t.nelements()
            v = (rng.standard_normal(n) * 0.05).astype(np.float32)
            be.tensor_set(t, v.astype(np.float16) if t.type == 1 else (np.abs(v) + 0.5 if t.type == 0 and n <= 4096 else v) if t.type == 0 else np.zeros(n, np.int32))
        g = c.graph()
        for _ in range(2):
            be.graph_compute(g)
        be.synchronize()
        best = 1e9
        for _ in range(5):
            a, b = be.timed_event(), be.timed_event()
            be.record(a); be.graph_compute(g); be.record(b)
            best = min(best, be.elapsed_ms(a, b))
        k = be.get_stat("kernels_last_graph")
        c.free()
        return round(best, 4), int(k)

    def flat(W):
        out = [v for k, v in W.items() if k != "layers" and hasattr(v, "nelements")]
        for L in W.get("layers", []):
            out += list(L.values())
        return out
    out = {}
    for nl in (1, 24):                                             # Whisper-medium encoder: 24 layers (30 s of audio: 3000 mel frames -> 1500 tokens -> 300 embeddings)
        c = pkg.Context(be)
        W = E.whisper_weights(c, E.WHISPER, nl); inp, _ = E.whisper(c, E.WHISPER, W, 3000)
        ms, k = timed(c, flat(W) + [inp])
        out[f"whisper_apm_{nl}_layer{'s' if nl > 1 else ''}_30s_audio_ms"] = ms; out[f"whisper_{nl}l_kernels"] = k
    for nl in (1, 27):                                             # SigLip2-so400m: 27 layers, one 448 x 448 slice -> 1024 patches -> 64 resampled queries
        c = pkg.Context(be)
        W = E.siglip2_weights(c, E.SIGLIP2, nl); inp, vit = E.siglip2(c, E.SIGLIP2, W)
        Wr = E.resampler_weights(c, E.RESAMPLER); pe, _ = E.resampler(c, E.RESAMPLER, Wr, vit, (E.SIGLIP2["image"] // E.SIGLIP2["patch"]) ** 2)
        ms, k = timed(c, flat(W) + flat(Wr) + [inp, pe])
        out[f"siglip2_vpm_{nl}_layer{'s' if nl > 1 else ''}_resampler_one_slice_ms"] = ms; out[f"siglip2_{nl}l_kernels"] = k
    c = pkg.Context(be)
    W = T.dit_weights(c, T.DIT); x, cond, _ = T.dit_block(c, T.DIT, W, 200)
    ms, k = timed(c, flat(W) + [x, cond])
    out["token2wav_dit_block_200_frames_ms"] = ms; out["dit_kernels"] = k
    c = pkg.Context(be)
    W = T.hift_weights(c, T.HIFT); x, _ = T.hift_upsample_stage(c, T.HIFT, W, 120)
    ms, k = timed(c, flat(W) + [x])
    out["token2wav_hift_stage_120_frames_ms"] = ms; out["hift_kernels"] = k
    return out


def omni_pinned(pkg, be, llm_model):
    """`--omni-pinned` (BASELINE configs[3] / [4] on the module map): every omni module on the backend mi355x_module_device() pins it to -- distinct
    GPUs when several are visible, otherwise several streams of device 0 -- with the embeddings moved by real mi355x_handoff calls (RCCL send / recv
    between devices, device copy + event on one device; csrc/handoff.cpp) and no host copy or host synchronisation between the modules.
      C4 stream_prefill TTFT: APM (Whisper-medium, 30 s of audio -> 300 embeddings) and VPM (SigLip2 + resampler, one slice -> 64) are submitted
        back to back on their own backends, both hand their rows into the LLM's input, the LLM prefills 300 + 64 + 30 rows; wall time from the first
        submission to the LLM's logits, best of 5.
      C5 chunk: the LLM prefills a 26-row chunk, result_norm rows are handed to the TTS backend, projector (4096 -> 768, F16) + the 768-wide Q8_0 TTS
        decoder over the 26 rows; wall time from the LLM submission to the TTS logits."""
    from llama_cpp_omni_amd import encoders as E, qwen3
    lib = be.lib
    mods = {m: int(lib.mi355x_module_device(m.encode())) for m in ("apm", "vpm", "llm", "tts")}
    n_dev = int(be.reg.contents.iface.get_device_count(be.reg))
    rng = np.random.default_rng(123)
    made = []

    def backend_for(m):
        b = pkg.Backend(mods[m]); made.append(b); return b

    def fill(b, tensors):
        for t in tensors:
            n = t.nelements()
            v = (rng.standard_normal(n) * 0.05).astype(np.float32)
            b.tensor_set(t, v.astype(np.float16) if t.type == 1 else (np.abs(v) + 0.5 if t.type == 0 and n <= 4096 else v) if t.type == 0 else np.zeros(n, np.int32))

    def flat(W):
        out = [v for k, v in W.items() if k != "layers" and hasattr(v, "nelements")]
        for L in W.get("layers", []):
            out += list(L.values())
        return out
    res = {"module_devices": mods, "devices_visible": n_dev}
    try:
        b_apm, b_vpm, b_tts = backend_for("apm"), backend_for("vpm"), backend_for("tts")
        if mods["llm"] == 0:
            b_llm, llm = be, llm_model                             # the benchmark's own 8B Q4_K_M weights (already resident on device 0)
        else:
            b_llm = backend_for("llm")
            llm = qwen3.Model(b_llm, qwen3.QWEN3_8B, qwen3.q4_k_m_types(qwen3.QWEN3_8B), n_ctx=512, seed=1234, flash_attn=True)
        E_ = qwen3.QWEN3_8B["n_embd"]
        # ---- APM / VPM graphs on their own backends
        ca = pkg.Context(b_apm)
        Wa = E.whisper_weights(ca, E.WHISPER, 24); mel, aud = E.whisper(ca, E.WHISPER, Wa, 3000)
        aud_out = ca.scale(aud, 1.0); ca.alloc(); fill(b_apm, flat(Wa) + [mel]); ga = ca.graph()
        cv = pkg.Context(b_vpm)
        Wv = E.siglip2_weights(cv, E.SIGLIP2, 27); img, vit = E.siglip2(cv, E.SIGLIP2, Wv)
        Wr = E.resampler_weights(cv, E.RESAMPLER); pe, vis = E.resampler(cv, E.RESAMPLER, Wr, vit, (E.SIGLIP2["image"] // E.SIGLIP2["patch"]) ** 2)
        vis_out = cv.scale(vis, 1.0); cv.alloc(); fill(b_vpm, flat(Wv) + flat(Wr) + [img, pe]); gv = cv.graph()
        n_a, n_v, n_t = int(aud_out.ne[1]), int(vis_out.ne[1]), 30
        assert aud_out.ne[0] == E_ and vis_out.ne[0] == E_, (aud_out.ne, vis_out.ne)
        n_llm = n_a + n_v + n_t
        # ---- LLM prefill graph over the handed-over rows
        gl, Il, logits = llm.build(n_llm, n_llm, n_outputs=1)
        llm.set_inputs(Il, (rng.standard_normal((n_llm, E_)) * 0.05).astype(np.float32), 0, n_llm)       # (the 30 text rows stay; the module rows are overwritten by the hand-offs)
        b_llm.tensor_set(Il["out_ids"], np.array([n_llm - 1], np.int32))
        glr = gl.graph()
        dst = Il["inp_embd"].t.data
        row = E_ * 4

        def c4_once():
            t0 = time.perf_counter()
            b_apm.graph_compute(ga)
            b_vpm.graph_compute(gv)
            r1 = lib.mi355x_handoff(b_apm.be, aud_out.t.data, b_llm.be, dst, n_a * row)
            r2 = lib.mi355x_handoff(b_vpm.be, vis_out.t.data, b_llm.be, dst + n_a * row, n_v * row)
            b_llm.graph_compute(glr)
            b_llm.synchronize()
            return time.perf_counter() - t0, (r1, r2)
        for _ in range(2):
            c4_once()
        best, kinds = min(c4_once() for _ in range(5))
        ok = bool(np.isfinite(b_llm.tensor_get(logits)).all())
        # the legs alone (same graphs, each synchronised): what the composed figure of the default run adds up
        legs = {}
        for name, b, g in (("apm_ms", b_apm, ga), ("vpm_ms", b_vpm, gv), ("llm_prefill_ms", b_llm, glr)):
            b.synchronize(); t = 1e9
            for _ in range(3):
                t0 = time.perf_counter(); b.graph_compute(g); b.synchronize(); t = min(t, time.perf_counter() - t0)
            legs[name] = round(t * 1e3, 3)
        res["c4_stream_prefill_ttft"] = {"measured_ttft_ms": round(best * 1e3, 3) if ok else None, "llm_prefill_tokens": n_llm, "handoff_kinds": {"apm": kinds[0], "vpm": kinds[1]},
                                         **legs, "sum_of_legs_ms": round(sum(legs.values()), 3),
                                         "note": "hand-off kind 1 = RCCL send/recv, 2 = device copy + event; APM and VPM run concurrently (own devices or own streams), the LLM waits on both hand-off events"}
        gl.free(); ca.free(); cv.free()
        # ---- C5: LLM chunk -> TTS
        n_c = 26
        TTS = qwen3.TTS
        llm.tap_hidden = True
        gl, Il, _ = llm.build(n_c, 256)
        hid = llm.hidden_out
        llm.tap_hidden = False
        llm.set_inputs(Il, (rng.standard_normal((n_c, E_)) * 0.05).astype(np.float32), 0, 256)
        glr = gl.graph()
        tts = qwen3.Model(b_tts, TTS, qwen3.uniform_types(TTS, pkg.GGML_TYPE_Q8_0), n_ctx=256, seed=9, flash_attn=True)
        cp = pkg.Context(b_tts)
        hin = cp.new_tensor(pkg.GGML_TYPE_F32, E_, n_c); pw = cp.new_tensor(pkg.GGML_TYPE_F16, E_, TTS["n_embd"]); proj = cp.mul_mat(pw, hin)
        cp.alloc(); b_tts.tensor_set(pw, (rng.standard_normal(pw.nelements()) / 64.0).astype(np.float16)); gp = cp.graph()
        gt, It, tl = tts.build(n_c, 256)
        tts.set_inputs(It, np.zeros((n_c, TTS["n_embd"]), np.float32), 0, 256)
        gtr = gt.graph()

        def c5_once():
            t0 = time.perf_counter()
            b_llm.graph_compute(glr)
            k1 = b_llm.handoff_tensor(hid, b_tts, hin)
            b_tts.graph_compute(gp)
            k2 = b_tts.handoff_tensor(proj, b_tts, It["inp_embd"])
            b_tts.graph_compute(gtr)
            b_tts.synchronize()
            return time.perf_counter() - t0, (k1, k2)
        for _ in range(2):
            c5_once()
        best5, kinds5 = min(c5_once() for _ in range(5))
        ok5 = bool(np.isfinite(b_tts.tensor_get(tl)).all())
        res["c5_llm_to_tts_chunk"] = {"measured_ms": round(best5 * 1e3, 3) if ok5 else None, "chunk_rows": n_c, "handoff_kinds": {"llm_to_tts": kinds5[0], "projector_to_decoder": kinds5[1]}}
        gl.free(); cp.free(); gt.free(); tts.wctx.free()
    except Exception as e:
        res["error"] = repr(e)
    finally:
        for b in made:
            try:
                b.close()
            except Exception:
                pass
    return res


class Replicas:
    """N > 1: one process per GPU (torch.distributed.run), each a whole-model replica; no data-path collective (decode of one sequence does
    not shard).  The only communication is the contract's: barrier + synchronize on both sides of the timed region and the
    MAX over ranks of the elapsed time; `value` = units of all ranks / that time.
    Test hooks (1-GPU box / CPU smoke of this very code): MI355X_BENCH_DIST_BACKEND=gloo -> CPU rendezvous instead of RCCL;
    MI355X_BENCH_SHARE_GPU=1 -> every rank on device 0.  The driver's real N > 1 runs use neither."""

    def __init__(self):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.backend = os.environ.get("MI355X_BENCH_DIST_BACKEND", "nccl")
        self.dev_index = 0 if os.environ.get("MI355X_BENCH_SHARE_GPU") else self.local_rank
        self.dist = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            if self.backend == "nccl":
                torch.cuda.set_device(self.dev_index)
                dist.init_process_group(backend="nccl", device_id=torch.device("cuda", self.dev_index))
            else:
                dist.init_process_group(backend=self.backend)
            self.dist = dist

    def barrier(self, device_sync):
        device_sync()
        if self.dist is not None:
            self.dist.barrier()
            if self.backend == "nccl":
                import torch
                torch.cuda.synchronize()

    def timed(self, step, steps, warmup, device_sync):
        """W untimed warm-up steps, then EXACTLY `steps` steps bracketed by barrier + synchronize; returns the MAX over ranks of the seconds"""
        pos = 0
        for _ in range(warmup):
            step(pos); pos += 1
        self.barrier(device_sync)
        t0 = time.perf_counter()
        for _ in range(steps):
            step(pos); pos += 1
        device_sync()
        self.dt_local = time.perf_counter() - t0                     # this rank's own time for its K steps (before it waits for the others): per-rank tok/s in `per_rank`
        self.barrier(device_sync)
        dt = time.perf_counter() - t0
        if self.dist is not None:
            import torch
            t = torch.tensor([dt], dtype=torch.float64, device="cuda" if self.backend == "nccl" else "cpu")
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt, pos

    def aggregate(self, steps, dt):
        return self.world * steps / dt                               # whole-job units per second (weak scaling: per-rank work is fixed)

    def evidence(self, steps, dt_local):
        """N > 1: what proves the line came from N ranks on N devices -- the world size the process group reports, and per rank its device ordinal, the
        device's PCI bus id (all-gathered) and its own tok/s over the timed region.  None at N = 1."""
        if self.dist is None:
            return None
        pci = None
        if self.backend == "nccl":
            import torch
            p = torch.cuda.get_device_properties(self.dev_index)
            if hasattr(p, "pci_bus_id"):
                pci = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, getattr(p, "pci_device_id", 0))
            else:
                pci = str(getattr(p, "uuid", ""))
        mine = {"rank": self.rank, "local_rank": self.local_rank, "device": self.dev_index, "pci": pci, "host": os.uname().nodename, "pid": os.getpid(),
                "tok_s": round(steps / dt_local, 2)}
        allr = [None] * self.world
        self.dist.all_gather_object(allr, mine)
        return {"rccl_world": int(self.dist.get_world_size()), "dist_backend": "rccl (torch.distributed nccl)" if self.backend == "nccl" else self.backend, "per_rank": allr}

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()
            self.dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="print only the cpu_baseline object (no GPU work)")
    ap.add_argument("--cpu-baseline-worker", type=int, default=0, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-baseline-seconds", type=float, default=6.0, help=argparse.SUPPRESS)
    ap.add_argument("--tiny", action="store_true", help="tiny shapes (plumbing check)")
    ap.add_argument("--no-fa", action="store_true")
    ap.add_argument("--c3", action="store_true", help="(default on at N = 1) BASELINE configs[2]: Qwen3-8B F16 prefill 8 x 2048 tokens (the `c3_f16_prefill` object)")
    ap.add_argument("--no-c3", action="store_true", help="skip the C3 leg")
    ap.add_argument("--omni-pinned", action="store_true", help="add `omni_pinned`: C4 TTFT / C5 chunk measured with every module on its mi355x_module_device() backend and real hand-offs")
    ap.add_argument("--no-omni-runtime", action="store_true", help="skip the via_reference_omni_runtime leg (the reference's omni.cpp driving the plug-in; ~60 s)")
    ap.add_argument("--no-libllama", action="store_true", help="skip the via_libllama leg (the metric through the reference's libllama with this plug-in)")
    args = ap.parse_args()

    if args.cpu_baseline_only or args.cpu_baseline_worker:
        pkg = load_pkg()
        from llama_cpp_omni_amd import qwen3
        cfg = qwen3.TINY if args.tiny else qwen3.QWEN3_8B
        if args.cpu_baseline_worker:
            print(json.dumps(_cpu_baseline_worker(pkg, cfg, qwen3.q4_k_m_types(cfg), 256, args.cpu_baseline_worker, args.cpu_baseline_seconds)), flush=True)
        else:
            print(json.dumps(cpu_baseline(pkg, cfg, qwen3.q4_k_m_types(cfg), 256, tiny=args.tiny)), flush=True)
        return
    rep = Replicas()
    world, rank = rep.world, rep.rank
    pkg = load_pkg()
    from llama_cpp_omni_amd import qwen3
    be = pkg.backend(rep.dev_index if world > 1 else 0)
    cfg = qwen3.TINY if args.tiny else qwen3.QWEN3_8B
    types = qwen3.q4_k_m_types(cfg)
    n_kv = 256                                                     # llama pads the visible KV length to 256 with FA
    n_ctx = max(512, ((args.steps + args.warmup + 8 + 255) // 256) * 256)
    n_kv = min(n_ctx, ((args.steps + args.warmup + 8 + 255) // 256) * 256)
    dec = Decoder(pkg, be, cfg, types, n_ctx=n_ctx, n_kv=n_kv, flash_attn=not args.no_fa)
    wbytes = dec.model.weight_bytes()

    dt, pos = rep.timed(dec.step, args.steps, args.warmup, be.synchronize)
    evidence = rep.evidence(args.steps, rep.dt_local)              # (a collective: every rank takes part)
    replays = be.get_stat("graph_replays")
    kernels = be.get_stat("kernels_last_graph")

    # ---- dominant kernel, measured live: replay of exactly the step's Q4_K mat-vec launches with HIP events at both ends
    roof = None
    if rank == 0 and not args.tiny:                               # (the tiny plumbing config is below the batch-1 kernels' shapes: no roofline)
        roof = kernel_roofline(pkg, be, dec.model)
        if os.environ.get("MI355X_BENCH_PROFILE"):
            be.set_option("profile", 1)
            be.set_option("reset_stats", 1)
            for _ in range(4):
                dec.step(pos); pos += 1
            e_us, e_n = be.get_stat("prof_empty_us"), be.get_stat("prof_empty_n")
            bracket = e_us / e_n if e_n > 0 else 0.0             # cost of an empty hipEvent pair on the stream (calibration)
            prof = {"_event_bracket_us": round(bracket, 3)}
            for cls in ("mmv_q4k", "mmv_q6k", "act_convert", "rms_norm_mul_quant", "rms_norm_mul", "rms_norm", "norm_rope", "rope", "fattn", "set_rows",
                        "get_rows", "bin", "glu", "cpy", "scale", "soft_max", "unary"):
                u, k = be.get_stat(f"prof_{cls}_us"), be.get_stat(f"prof_{cls}_n")
                if k > 0:
                    prof[cls] = {"launches_per_step": k / 4, "us_per_step": round(u / 4, 1), "avg_event_to_event_us": round(u / k, 2)}
            be.set_option("profile", 0)
            sys.stderr.write("eager per-class profile (HIP events around every launch): " + json.dumps(prof) + "\n")

    if rank == 0:
        tok_s = rep.aggregate(args.steps, dt)
        out = {
            "metric": "llama-bench tg128 tok/s (decode, batch 1), Qwen3-8B Q4_K_M" if not args.tiny else "decode tok/s (tiny plumbing config)",
            "value": round(tok_s, 2), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "q4_K/q6_K weights x q8_K activations (int8 dot, f32 accumulate)", "data": "synthetic",
            "config": {"workload": f"Qwen3-8B Q4_K_M text-only decode, batch=1 per GPU, {world}xMI355X (mul_mat_vec_q path)" if not args.tiny else "tiny",
                       "n_layer": cfg["n_layer"], "n_embd": cfg["n_embd"], "n_ff": cfg["n_ff"], "n_vocab": cfg["n_vocab"], "n_kv": n_kv,
                       "flash_attn": not args.no_fa, "weight_bytes_per_token": wbytes, "parallelism": f"replicas x{world} (no collective)",
                       "graph_replays": replays, "kernels_per_token": kernels},
            "hbm_frac_whole_step": round(wbytes * (args.steps / dt) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": roof,
        }
        if evidence:
            out.update(evidence)
        if world == 1 and not args.tiny and not os.environ.get("MI355X_BENCH_NO_EXTRAS"):
            out["extras"] = extra_legs(pkg, be, args.no_fa)        # (before the prefill legs: they leave 15 GB of F16 weight images behind)
        if world == 1 and not os.environ.get("MI355X_BENCH_NO_PP"):
            try:                                                   # second half of the headline metric: pp512 (reported, not `value`)
                pp, ok = prefill_tok_s(pkg, be, dec.model)
                out["pp512_tok_s"] = round(pp, 1) if ok else None
                out["ttft_ms_pp512"] = round(512.0 / pp * 1e3, 2) if ok else None      # time to first token of a 512-token prompt: one ubatch through the same graphs
                om = (out.get("extras") or {}).get("omni_modules") or {}
                apm, vpm = om.get("whisper_apm_24_layers_30s_audio_ms"), om.get("siglip2_vpm_27_layers_resampler_one_slice_ms")
                if ok and apm and vpm and not args.tiny:
                    # BASELINE configs[3] (omni stream_prefill) composed from legs measured in THIS run on one GPU: one APM pass (30 s of audio ->
                    # 300 embeddings), one VPM pass (one 448 x 448 slice -> 64 queries), then the LLM prefill of those 364 embeddings + 30 text tokens
                    n_llm = 300 + 64 + 30
                    ppn, okn = prefill_tok_s(pkg, be, dec.model, n_tokens=n_llm)
                    llm_ms = n_llm / ppn * 1e3
                    out["c4_stream_prefill_ttft"] = {"apm_ms": apm, "vpm_ms": vpm, "llm_prefill_tokens": n_llm, "llm_prefill_ms": round(llm_ms, 2),
                                                     "one_gpu_ms": round(apm + vpm + llm_ms, 2), "modules_on_own_gpus_ms": round(max(apm, vpm) + llm_ms, 2),
                                                     "note": "legs measured on this GPU; the pinned form (APM / VPM / LLM on three GPUs, module map) adds the two embedding hand-offs (<= 4.7 MB over xGMI)"} if okn else None
                if ok and not args.tiny:
                    # the two round-5 arithmetic options of the prefill GEMMs, same graph: off by default because they are slower
                    var = {}
                    for name, opts in (("prefill_q8k", {"prefill_q8k": 1}), ("prefill_q8k_mmq_tile", {"prefill_q8k": 1, "mmq_tile": 1})):
                        try:
                            for k, v in opts.items(): be.set_option(k, v)
                            ppv, okv = prefill_tok_s(pkg, be, dec.model, reps=2)
                            var[name] = round(ppv, 1) if okv else None
                        finally:
                            for k in opts: be.set_option(k, -1)
                    out["pp512_arithmetic_options_tok_s"] = dict(var, note="prefill_q8k: K-quant GEMMs take the Q8_K-quantised activations (per-op NMSE vs the oracle 2e-6 instead of 5e-5); mmq_tile: Q4_K on the int8 matrix cores, the oracle's exact integers")
            except Exception as e:
                out["pp512_tok_s"] = None
                out["pp512_error"] = repr(e)
        if (args.c3 or not (args.no_c3 or args.tiny)) and world == 1:
            try:
                out["c3_f16_prefill"] = {"ub512": c3_prefill(pkg, be, n_ubatch=512, tiny=args.tiny), "ub2048": c3_prefill(pkg, be, n_ubatch=2048, tiny=args.tiny),
                                        "ub16384": c3_prefill(pkg, be, tiny=args.tiny, one_ubatch=True)}
            except Exception as e:
                out["c3_f16_prefill"] = {"error": repr(e)}
        if world == 1 and not args.tiny and not args.no_libllama and not os.environ.get("MI355X_BENCH_NO_EXTRAS"):
            be.synchronize()
            out["via_reference_omni_modules"] = via_reference_omni_modules()
            if not args.no_omni_runtime:
                out["via_reference_omni_runtime"] = via_reference_omni_runtime()
            out["via_libllama"] = via_libllama()
            v = out["via_libllama"]
            if v and isinstance(v.get("fa1"), dict) and v["fa1"].get("tg128_tok_s"):
                v["tg128_fa1_over_value"] = round(v["fa1"]["tg128_tok_s"] / out["value"], 3)
            # which number is llama-bench's: `value` is this repo's harness on the plug-in's C-ABI; the figures below come from the reference's own libllama +
            # scheduler at llama-bench's defaults (-fa 0), the closest thing to the llama-bench binary that builds here (tools/llama_bench_min.cpp)
            if v and isinstance(v.get("fa0"), dict):
                out["value_via_libllama"] = v["fa0"].get("tg128_tok_s")
                out["pp512_via_libllama"] = v["fa0"].get("pp512_tok_s")
        if args.omni_pinned and world == 1 and not args.tiny:
            be.synchronize()
            out["omni_pinned"] = omni_pinned(pkg, be, dec.model)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(pkg, cfg, types, n_kv, tiny=args.tiny)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    rep.finish()


if __name__ == "__main__":
    main()
