#!/bin/bash
# tools/lazy_cont_ab.sh -- copies left un-run (lazy_try_register, graph_exec.cpp) on the reference's Token2Wav: launches and device time per window graph, wall time per
# window, waveforms compared byte for byte.  off: MI355X_NO_LAZY_CACHE_CONT=1 MI355X_NO_LAZY_ATTN_CONT=1; conv: the causal convolution's two cache copies lazy;
# all: also the Q / K head-flattening copies in front of the f32 attention chain (attn_f32 reads the permuted views).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD
[ -d /tmp/t2w ] || python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so
for V in off conv all; do  # (MI355X_NO_LAZY_CONCAT_SRC=1 switches the CONCAT-operand form off separately)
  echo "-- lazy copies: $V"
  unset MI355X_NO_LAZY_CACHE_CONT MI355X_NO_LAZY_ATTN_CONT
  [ $V = off ] && export MI355X_NO_LAZY_CACHE_CONT=1 MI355X_NO_LAZY_ATTN_CONT=1
  [ $V = conv ] && export MI355X_NO_LAZY_ATTN_CONT=1
  MI355X_GRAPH_GPU_TIME=1 MI355X_LOG_STATS=1 MI355X_SINK_DEBUG=1 timeout 300 oracle/_ref/t2w-min /tmp/t2w /tmp/w_$V.f32 gpu --windows 7 2>&1 | grep "device time per graph\|graphs eager\|\"module\"\|capture failed\|rror\|lazy_cont\|attn_f32:" | cut -c1-420
done
cmp /tmp/w_off.f32 /tmp/w_conv.f32 && echo "conv: waveform identical to off"
cmp /tmp/w_off.f32 /tmp/w_all.f32 && echo "all: waveform identical to off"
