#!/bin/bash
# tools/depth_bench.sh -- decode at KV depth and several sequences decoded together, through the reference libllama (oracle/_ref/llama-bench-min)
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
for d in ${DEPTHS:-0 512 2048 8192 32768}; do
  GGML_BACKEND_PATH=$LIB timeout 600 $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 0 -n 64 -d $d -r 2 -t 8 2>/dev/null < /dev/null | tail -1
done
for n in ${PARS:-2 4 8}; do
  for u in 1 0; do
    GGML_BACKEND_PATH=$LIB timeout 600 $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 0 -n 64 --parallel $n --kv-unified $u -r 3 -t 8 2>/dev/null < /dev/null | tail -1
  done
done
