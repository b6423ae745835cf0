#!/bin/bash
# tools/par_prof.sh N -- per-class event timing (MI355X_PROFILE, eager) of N sequences decoded together through the reference libllama
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
N=${1:-16}
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
MI355X_PROFILE=1 MI355X_LOG_STATS=1 MI355X_VERBOSE=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 0 -n 16 --parallel $N --kv-unified 0 -r 1 -t 8 > /tmp/pp.log 2>/tmp/pp.err < /dev/null
tail -1 /tmp/pp.log
grep "mi355x" /tmp/pp.err | tail -30 | tee gpurun_out/par_prof_$N.txt
