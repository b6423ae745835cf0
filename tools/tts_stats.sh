#!/bin/bash
# tools/tts_stats.sh [q8_0|f16] -- the omni TTS decoder (synthetic GGUF at its real shape) through the reference's libllama: tg128 with the
# embedding input path, launches per token, per-class launch times
TY=${1:-q8_0}
cd "$(dirname "$0")/.."
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config tts --types $TY -o /tmp/tts.gguf --n-ctx 4096 >/dev/null || exit 1
MI355X_LOG_STATS=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/tts.gguf -ngl 99 -fa 1 -p 26 -n 128 -r 3 -t 8 --embd 2>&1 < /dev/null | grep -E "mi355x|tg128|pp26" | tail -4
MI355X_PROFILE=1 MI355X_LOG_STATS=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/tts.gguf -ngl 99 -fa 1 -p 0 -n 32 -r 1 -t 8 --embd 2>&1 < /dev/null | grep -E "mi355x" | tail -20
rm -f /tmp/tts.gguf
