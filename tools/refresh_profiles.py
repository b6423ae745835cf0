#!/usr/bin/env python3
"""tools/refresh_profiles.py <gpurun_out/prof_TAG> [TAG] [COMMIT] -- copy the summaries of a tools/profile_round.sh run (and the latest
default bench / drop-in outputs in gpurun_out/) into profiles/ under the round's names (TAG, default r02)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

P = sys.argv[1]
TAG = sys.argv[2] if len(sys.argv) > 2 else "r02"
COMMIT = sys.argv[3] if len(sys.argv) > 3 else None
stats = max(glob.glob(P + "/trace/**/*kernel_stats.csv", recursive=True), key=os.path.getmtime)      # (gpurun merges every run into the same local directory: newest)
pmc = max(glob.glob(P + "/pmc/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
shutil.copy(stats, "profiles/" + TAG + "_kernel_stats.csv")
shutil.copy(P + "/bench_under_rocprof.json", "profiles/" + TAG + "_bench_under_rocprof.json")
if os.path.exists("gpurun_out/bench_default.json"):
    shutil.copy("gpurun_out/bench_default.json", "profiles/" + TAG + "_bench.json")
rows = list(csv.DictReader(open(stats)))
with open("profiles/" + TAG + "_kernel_stats_summary.txt", "w") as f:
    f.write("rocprofv3 --kernel-trace --stats --output-format csv -- MI355X_GRAPHS=0 python bench.py --steps 32 --warmup 8 --no-cpu-baseline\n")
    f.write("(decode steps + the roofline replay + the extras [no-FA decode, TTS decoder, omni encoder / Token2Wav graphs] + the pp512 leg; graph replay off under the profiler)  tools/profile_round.sh\n\n")
    f.write("%-100s %8s %12s %8s\n" % ("kernel", "calls", "avg_ns", "pct"))
    for r in rows:
        f.write("%-100s %8s %12.1f %8s\n" % (r["Name"][:100], r["Calls"], float(r["AverageNs"]), r["Percentage"]))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(pmc)):
    if r.get("Counter_Name") != "FETCH_SIZE":
        continue
    agg[r["Kernel_Name"]][0] += 1
    agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
out = {"command": "MI355X_GRAPHS=0 MI355X_BENCH_NO_PP=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline",
       "unit_note": "FETCH_SIZE is reported in KiB and counts 64 B per 128-B request for wide coalesced streaming reads on gfx950: bytes = KiB * 1024 * 2 (MI355X_MICROARCH.md, HBM section)",
       "commit": COMMIT, "kernels": {}}
for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    out["kernels"][k] = {"dispatches": n, "fetch_size_kib_avg": round(v / n, 1), "hbm_bytes_per_dispatch_corrected": int(v / n * 1024 * 2)}
json.dump(out, open("profiles/" + TAG + "_pmc_fetch_size.json", "w"), indent=1)
if os.path.exists("gpurun_out/dropin_8b_fa1.txt") and TAG == "r01":
    with open("profiles/" + TAG + "_llama_dropin.txt", "w") as f:
        f.write("# tools/run_llama_dropin.sh all  (MI355X box; reference libllama + ggml_backend_sched from oracle/_ref, plug-in from GGML_BACKEND_PATH)\n")
        f.write("# synthetic Qwen3-8B Q4_K_M GGUF (tools/make_synth_gguf.py), llama-bench loops (tools/llama_bench_min.cpp), -ngl 99 -t 8 -r 5\n\n## 8B, -fa 1\n")
        f.write(open("gpurun_out/dropin_8b_fa1.txt").read() + "\n## 8B, -fa 0 (llama-bench default)\n" + open("gpurun_out/dropin_8b_fa0.txt").read())
        f.write("\n## tiny 2-layer Q4_K_M GGUF: greedy ids + final logits, CPU backend (-ngl 0) vs MI355X (-ngl 99)\nfa=1:\n" + open("gpurun_out/dropin_tiny_parity_fa1.txt").read())
        f.write("fa=0:\n" + open("gpurun_out/dropin_tiny_parity_fa0.txt").read())
b = json.load(open("profiles/" + TAG + "_bench.json"))
print("decode", b["value"], "ms", b["ms_per_step"], "frac_step", b["hbm_frac_whole_step"], "roof", b["roofline"]["achieved"], b["roofline"]["frac"], b["roofline"]["avg_launch_us"],
      "pp512", b.get("pp512_tok_s"), "c3", b.get("c3_f16_prefill"), "cpu", b["cpu_baseline"]["value"] if b.get("cpu_baseline") else None)
for r in rows[:3]:
    print(r["Name"][:60], r["AverageNs"])
