LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
MI355X_LOG_STATS=1 MI355X_VERBOSE=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 0 -n 128 -r 2 -t 8 2>&1 < /dev/null | grep -E "mi355x|tg128" | tail -5
