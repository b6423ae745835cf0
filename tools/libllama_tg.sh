#!/bin/bash
# tools/libllama_tg.sh -- tg128 / pp512 through the reference's libllama with the plug-in, deferred small blocking uploads on and off (MI355X_NO_DEFERRED_SET), with the plug-in's stats line
cd "$(dirname "$0")/.."
LIB=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so; BIN=$PWD/oracle/_ref/llama-bench-min
python tools/make_synth_gguf.py --config 8b --types q4_k_m -o /tmp/q8b.gguf >/dev/null || exit 1
for v in on off on off; do
  if [ $v = off ]; then export MI355X_NO_DEFERRED_SET=1; else unset MI355X_NO_DEFERRED_SET; fi
  echo "== deferred small blocking uploads: $v"
  MI355X_LOG_STATS=1 GGML_BACKEND_PATH=$LIB timeout 600 $BIN -m /tmp/q8b.gguf -ngl 99 -fa 1 -p 512 -n 128 -r 5 -t 8 2>&1 < /dev/null | grep -E '"test"|blocking buffer|host time' | cut -c1-260
done
rm -f /tmp/q8b.gguf
