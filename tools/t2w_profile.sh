#!/bin/bash
# tools/t2w_profile.sh <tag> -- the reference's Token2Wav (oracle/_ref/t2w-min) on the plug-in under rocprofv3 --kernel-trace --stats (run on the GPU box)
set -e
TAG=${1:-t2w}
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so MI355X_LOG_STATS=1
oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 8 > $OUT/plain.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
MI355X_GRAPHS=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t2w -- $ROOT/oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 4 > $OUT/rocprof.txt 2>&1 || true
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
tot = sum(float(r["TotalDurationNs"]) for r in rows); n = sum(int(r["Calls"]) for r in rows)
with open("$OUT/kernels.txt", "w") as o:
    o.write("total device time %.3f ms, %d launches (prompt set-up + 4 windows)\n" % (tot / 1e6, n))
    for r in rows[:40]:
        o.write("%8.3f ms %6s calls %9.2f us avg %5.1f%%  %s\n" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:150]))
print(open("$OUT/kernels.txt").read())
PY
grep -E "^\{|mi355x" $OUT/plain.txt | tail -6
