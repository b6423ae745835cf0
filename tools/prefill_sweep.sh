#!/bin/bash
# tools/prefill_sweep.sh -- prefill through the reference libllama at other ubatch sizes / depths / prompt lengths than llama-bench's defaults
# (sanity + numbers for profiles/): both attention paths
cd "$(dirname "$0")/.."
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf --n-ctx 8192 >/dev/null || exit 1
for fa in 1 0; do
  for cfg in "-p 2048 -ub 2048 -b 2048" "-p 2048 -ub 512" "-p 512 -d 2048" "-p 100" "-p 1000 -ub 256" "-p 4096 -ub 4096 -b 4096"; do
    GGML_BACKEND_PATH=$LIB timeout 600 $BIN -m /tmp/q8b.gguf -ngl 99 -fa $fa $cfg -n 0 -r 3 -t 8 2>/tmp/sweep.err | grep avg_ts | sed "s/^/[fa=$fa $cfg] /" | cut -c1-220
    grep -i -E "abort|error|assert" /tmp/sweep.err | head -3
  done
done
rm -f /tmp/q8b.gguf
