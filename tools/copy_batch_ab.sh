#!/bin/bash
# tools/copy_batch_ab.sh -- deferred layout copies (graph_exec.cpp copy_queue / elementwise.hip k_copy_batch) on the reference's Token2Wav: launches and device time per window
# graph, wall time per window, waveforms compared byte for byte against MI355X_NO_COPY_BATCH=1.
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
[ -d /tmp/t2w ] || python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so
for V in off on; do
  echo "-- copy batching: $V"
  unset MI355X_NO_COPY_BATCH
  [ $V = off ] && export MI355X_NO_COPY_BATCH=1
  MI355X_GRAPH_GPU_TIME=1 MI355X_LOG_STATS=1 timeout 300 oracle/_ref/t2w-min /tmp/t2w /tmp/w_$V.f32 gpu --windows 7 2>&1 | grep "device time per graph\|graphs eager\|\"module\"\|capture failed\|rror\|abort" | cut -c1-420
done
cmp /tmp/w_off.f32 /tmp/w_on.f32 && echo "on: waveform identical to off"
