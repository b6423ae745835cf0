#!/bin/bash
# tools/lazy_cont_ab.sh -- the causal convolution's two cache copies left un-run (lazy_try_register, graph_exec.cpp) against MI355X_NO_LAZY_CACHE_CONT=1 on the reference's
# Token2Wav: launches and device time per window graph, wall time per window, waveforms compared byte for byte
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
[ -d /tmp/t2w ] || python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so
for V in off on; do
  echo "-- lazy cache copies $V"
  if [ $V = off ]; then export MI355X_NO_LAZY_CACHE_CONT=1; else unset MI355X_NO_LAZY_CACHE_CONT; fi
  MI355X_GRAPH_GPU_TIME=1 MI355X_LOG_STATS=1 timeout 300 oracle/_ref/t2w-min /tmp/t2w /tmp/w_$V.f32 gpu --windows 7 2>&1 | grep "device time per graph\|graphs eager\|\"module\"\|capture failed\|rror" | cut -c1-420
done
cmp /tmp/w_off.f32 /tmp/w_on.f32 && echo "waveforms identical"
