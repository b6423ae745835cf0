#!/bin/bash
# tools/llama_stats.sh [fa] -- per-class launch counts / times of the decode step as the reference's libllama submits it (MI355X_PROFILE: HIP
# events around every launch, graph replay off), flash-attention on (1, default) or off (0: llama-bench's default)
FA=${1:-1}; DEPTH=${2:-0}
cd "$(dirname "$0")/.."
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
MI355X_LOG_STATS=1 MI355X_VERBOSE=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/q8b.gguf -ngl 99 -fa $FA -p 0 -n 64 -d $DEPTH -r 2 -t 8 2>&1 < /dev/null | grep -E "mi355x|tg128" | tail -5
MI355X_PROFILE=1 MI355X_LOG_STATS=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/q8b.gguf -ngl 99 -fa $FA -p 0 -n 32 -d $DEPTH -r 1 -t 8 2>&1 < /dev/null | grep -E "mi355x|prof" | tail -40
rm -f /tmp/q8b.gguf
