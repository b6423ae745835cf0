#!/bin/bash
# tools/depth_trace.sh DEPTH -- rocprofv3 kernel durations of the decode step at a KV depth through the reference libllama (graph replay off)
D=${1:-2048}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
OUT=gpurun_out/prof_depth_$D
rm -rf "$OUT"; mkdir -p "$OUT"
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
MI355X_GRAPHS=0 GGML_BACKEND_PATH=$LIB timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 0 -n 8 -d $D -r 1 -t 8 > "$OUT/run.txt" 2> "$OUT/err.txt" < /dev/null
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    # the last decode step: kernels after the last lm-head-sized gap; simply take the last 230 kernels
    last = rows[-222:]
    agg = collections.defaultdict(list)
    for r in last:
        agg[r["Kernel_Name"][:80]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    tot = 0
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print("%-82s n=%4d avg %8.2f us total %8.1f us" % (k, len(v), sum(v) / len(v) / 1e3, sum(v) / 1e3)); tot += sum(v)
    print("sum of kernel durations of the last step: %.1f us; span %.1f us" % (tot / 1e3, (int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])) / 1e3))
PY
rm -f /tmp/q8b.gguf
