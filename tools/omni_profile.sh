#!/bin/bash
# tools/omni_profile.sh <module apm|vpm> <tag> -- the reference's encoder code (oracle/_ref/omni-enc-min) on the plug-in under rocprofv3 --kernel-trace --stats:
# wall time per chunk (host + device, the number a user of audition_audio_encode / vision_image_encode sees) beside the device time per kernel.
set -e
MOD=$1; TAG=${2:-omni}
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT
python tools/make_synth_omni_gguf.py --module $MOD -o /tmp/$MOD.gguf
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so MTMD_BACKEND_DEVICE=MI355X0 MI355X_LOG_STATS=1
ARGS="--chunks 8"; [ $MOD = apm ] && ARGS="--chunks 8 --frames 100"
oracle/_ref/omni-enc-min $MOD /tmp/$MOD.gguf /tmp/$MOD.bin --gpu $ARGS > $OUT/${MOD}_plain.txt 2>&1 || true
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${MOD}_prof -o $MOD -- $ROOT/oracle/_ref/omni-enc-min $MOD /tmp/$MOD.gguf /tmp/$MOD.bin --gpu $ARGS > $OUT/${MOD}_rocprof.txt 2>&1 || true
cd $ROOT
python - <<PY
import csv, glob
f = glob.glob("$OUT/${MOD}_prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0]))) if f else []
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("$OUT/${MOD}_kernels.txt", "w") as o:
    o.write("total device time %.3f ms over 8 chunks (+ load/reserve)\n" % (tot / 1e6))
    for r in rows[:25]:
        o.write("%8.3f ms %6s calls %9.2f us avg %5.1f%%  %s\n" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:150]))
print(open("$OUT/${MOD}_kernels.txt").read())
PY
tail -4 $OUT/${MOD}_plain.txt
