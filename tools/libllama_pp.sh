#!/bin/bash
# tools/libllama_pp.sh [fa] -- pp512 through the reference's libllama (llama-bench-min -p 512 -n 0) with the plug-in's host-time account (MI355X_LOG_STATS)
FA=${1:-1}
cd "$(dirname "$0")/.."
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
MI355X_LOG_STATS=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/q8b.gguf -ngl 99 -fa $FA -p 512 -n 0 -r 10 -t 8 2>&1 < /dev/null | grep -E "mi355x|pp512|avg_ts" | cut -c1-600 | tail -12
rm -f /tmp/q8b.gguf
