cd /root/repo
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null
export MI355X_LOG_STATS=1 GGML_BACKEND_PATH=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
for args in "-p 2048 -n 64 -fa 1 -ub 512" "-p 2048 -n 0 -fa 1 -ub 2048" "-p 4096 -n 32 -fa 0 -ub 512" "-p 300 -n 300 -fa 1"; do
  echo "== $args"; timeout 600 oracle/_ref/llama-bench-min -m /tmp/q8b.gguf -ngl 99 -r 3 -t 8 $args 2>&1 | grep -E "mi355x\] MI|avg_ts|error|abort|failed"
done
