#!/bin/bash
# tools/par_bench.sh -- N sequences decoded together through the reference libllama (aggregate tok/s), PARS="2 4 8 ..." KVU="1 0"
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
for n in ${PARS:-2 4 8 16 32 64}; do
  for u in ${KVU:-1 0}; do
    GGML_BACKEND_PATH=$LIB timeout 600 $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 0 -n 64 --parallel $n --kv-unified $u -r 3 -t 8 2>/dev/null < /dev/null | tail -1
  done
done
