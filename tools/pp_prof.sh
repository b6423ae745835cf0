#!/bin/bash
# tools/pp_prof.sh [fa] -- rocprofv3 --kernel-trace --stats of pp512 as the reference's libllama submits it (llama-bench-min -p 512 -n 0), flash-attention off (0, llama-bench's default) or on (1)
FA=${1:-0}
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/pp_prof_fa$FA; mkdir -p $OUT
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
cd /tmp; export TMPDIR=/tmp
GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o pp -- $ROOT/oracle/_ref/llama-bench-min -m /tmp/q8b.gguf -ngl 99 -fa $FA -p 512 -n 0 -r 5 -t 8 > $OUT/run.txt 2>&1
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total device time %.2f ms" % (tot / 1e6))
for r in rows[:16]:
    print("%8.3f ms %6s calls %9.2f us avg %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:120]))
PY
grep '"test"' $OUT/run.txt | cut -c1-120
rm -f /tmp/q8b.gguf
