# tools/depth_prof.sh DEPTH -- per-class event timing (MI355X_PROFILE) of decode at a KV depth through the reference libllama
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
BIN=$PWD/oracle/_ref/llama-bench-min
D=${1:-32768}
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
MI355X_PROFILE=1 MI355X_LOG_STATS=1 MI355X_VERBOSE=1 GGML_BACKEND_PATH=$LIB timeout 300 $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 0 -n 16 -d $D -r 1 -t 8 > /tmp/dp.log 2>/tmp/dp.err < /dev/null
tail -1 /tmp/dp.log
grep "mi355x" /tmp/dp.err | tail -30 | tee gpurun_out/depth_prof_$D.txt
