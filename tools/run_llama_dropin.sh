#!/bin/bash
# tools/run_llama_dropin.sh [tiny|8b|all] -- SURVEY.md 8(f) rank 1: the REAL libllama (oracle/_ref/libllama-ref.so, built from the
# reference sources by oracle/Makefile.ref `llama`) drives this repo's backend through ggml_backend_sched, loaded as a plug-in from
# GGML_BACKEND_PATH.  Runs on the GPU box (gpurun); writes gpurun_out/dropin_*.txt.
#   tiny : greedy token ids + final logits, CPU backend (-ngl 0) vs MI355X (-ngl 99) on a synthetic 2-layer Q4_K_M GGUF
#   8b   : llama-bench's pp512 / tg128 loops (tools/llama_bench_min.cpp) on a synthetic Qwen3-8B Q4_K_M GGUF, -fa 1 and -fa 0
set -u
cd "$(dirname "$0")/.."
WHAT=${1:-all}
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=oracle/_ref/llama-bench-min
mkdir -p gpurun_out
[ -x $BIN ] || { echo "missing $BIN (make -f oracle/Makefile.ref llama)"; exit 1; }

if [ "$WHAT" = tiny ] || [ "$WHAT" = all ]; then
  python tools/make_synth_gguf.py --config tiny --types q4_k_m -o /tmp/tiny.gguf --distinct-layers
  for fa in 1 0; do
    timeout 120 $BIN -m /tmp/tiny.gguf -ngl 0 -fa $fa --greedy 32 -t 4 --dump-logits /tmp/tiny_cpu_$fa.bin 2>/tmp/tiny_cpu.err | tail -1 > gpurun_out/dropin_tiny_cpu_fa$fa.txt
    GGML_BACKEND_PATH=$LIB timeout 120 $BIN -m /tmp/tiny.gguf -ngl 99 -fa $fa --greedy 32 -t 4 --dump-logits /tmp/tiny_gpu_$fa.bin 2>/tmp/tiny_gpu_$fa.err | tail -1 > gpurun_out/dropin_tiny_gpu_fa$fa.txt
    grep -E "^devices|MI355X|offload|buffer size|graph splits" /tmp/tiny_gpu_$fa.err | head -12
    python3 - $fa <<'PY'
import json, sys, numpy as np
fa = sys.argv[1]
a = json.load(open(f"gpurun_out/dropin_tiny_cpu_fa{fa}.txt"))["greedy_ids"]; b = json.load(open(f"gpurun_out/dropin_tiny_gpu_fa{fa}.txt"))["greedy_ids"]
x = np.fromfile(f"/tmp/tiny_cpu_{fa}.bin", np.float32); y = np.fromfile(f"/tmp/tiny_gpu_{fa}.bin", np.float32)
nm = float(((x - y) ** 2).sum() / (x ** 2).sum())
print(f"DROPIN tiny fa={fa}: ids_equal={a == b} n={len(a)} logits_nmse={nm:.3e}")
open(f"gpurun_out/dropin_tiny_parity_fa{fa}.txt", "w").write(f"ids_equal={a == b} n={len(a)} logits_nmse={nm:.3e}\ncpu={a}\ngpu={b}\n")
PY
  done
fi

if [ "$WHAT" = 8b ] || [ "$WHAT" = all ]; then
  df -h /tmp | tail -1
  python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf || exit 1
  for fa in 1 0; do
    GGML_BACKEND_PATH=$LIB timeout 900 $BIN -m /tmp/q8b.gguf -ngl 99 -fa $fa -p 512 -n 128 -r 5 -t 8 2>/tmp/q8b_$fa.err | tee gpurun_out/dropin_8b_fa$fa.txt
    grep -E "^devices|offloaded|MI355X.*buffer size|graph splits|graph nodes" /tmp/q8b_$fa.err | head -8 | tee -a gpurun_out/dropin_8b_fa$fa.txt
    tail -3 /tmp/q8b_$fa.err
  done
  rm -f /tmp/q8b.gguf
fi
