#!/bin/bash
# tools/par_tail_ab.sh -- option par_tail (MI355X_PAR_TAIL=S: the layout-only suffix of a captured graph spread over S capture streams) on the reference's Token2Wav:
# (the option lives in tools/lab/par_tail.diff -- measured slower, not in the product; apply the diff to llama.cpp-omni_amd/csrc to re-run this)
# device time per window graph and wall time per window with S = 0 / 2 / 4 / 8, the waveforms compared byte for byte.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
[ -d /tmp/t2w ] || python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so
for S in 0 2 4 8; do
  echo "-- MI355X_PAR_TAIL=$S"
  MI355X_PAR_TAIL=$S MI355X_GRAPH_GPU_TIME=1 MI355X_LOG_STATS=1 timeout 300 oracle/_ref/t2w-min /tmp/t2w /tmp/w$S.f32 gpu --windows 7 2>&1 | grep "device time per graph\|graphs eager\|\"module\"\|capture failed\|rror" | cut -c1-420
  cmp /tmp/w0.f32 /tmp/w$S.f32 && echo "waveform identical to S=0"
done
