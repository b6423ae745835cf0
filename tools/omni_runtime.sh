#!/bin/bash
# tools/omni_runtime.sh -- SURVEY.md 8 row g1 on the GPU box: the reference's omni runtime (oracle/_ref/omni-min = tools/omni/omni.cpp + modules + libllama)
# drives the plug-in over the synthetic full-size module set; log and summary into gpurun_out/ (copy the summary into profiles/).
#   tools/omni_runtime.sh [--map llm=0,t2w=1,apm=2,vpm=2] [--print] [max_tgt] [turns] [omni]
#       (omni: media_type 2 -- every user turn after the first carries a 448 x 448 picture through vision.cpp)
# --map: one module per GPU (BASELINE C4 / C5's pinned form), spelled with the reference runtime's OWN knobs -- nothing of this repo's in the command line but the plug-in path:
#     llm=N  ->  -mg N -sm none              common_params.main_gpu / split_mode (common/arg.cpp:2906, 2955): the LLM's one device
#     tts=N  ->  (must equal llm)            the reference loads the TTS model with the LLM's common_params (omni.cpp:3453-3461 llama_init_tts): a device of its own needs the
#                                            one-line maintainer change of INTEGRATION.md 3a; so does the LLM -> TTS hidden-state hand-off over xGMI (mi355x_handoff, csrc/handoff.cpp)
#     t2w=N  ->  --t2w-device gpu:N          omni_init's token2wav_device (omni.cpp:3775-3778; token2wav-impl.cpp:1891-1960 picks the N-th GPU device of the registry)
#     apm=N  ->  MTMD_BACKEND_DEVICE=MI355XN audition.cpp:241 (ggml_backend_init_by_name)
#     vpm=N  ->  Omni_BACKEND_DEVICE=MI355XN vision.cpp:201
# --print: show the translated command and environment, run nothing (the translation is testable without a GPU: tests/test_host_mirror.py).
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD; SET=${OMNI_SET:-/tmp/omni_set}; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
MAP=""; PRINT=0
while [ $# -gt 0 ]; do
  case "$1" in
    --map) MAP=$2; shift 2;;
    --print) PRINT=1; shift;;
    *) break;;
  esac
done
LLM_DEV=""; TTS_DEV=""; T2W_DEV=0; APM_DEV=""; VPM_DEV=""
IFS=, read -ra KV <<< "$MAP"
for kv in "${KV[@]:-}"; do
  [ -z "$kv" ] && continue
  k=${kv%%=*}; v=${kv#*=}
  case "$v" in ''|*[!0-9]*) echo "omni_runtime.sh: --map $kv: device index expected" >&2; exit 2;; esac
  case "$k" in
    llm) LLM_DEV=$v;; tts) TTS_DEV=$v;; t2w) T2W_DEV=$v;; apm) APM_DEV=$v;; vpm) VPM_DEV=$v;;
    *) echo "omni_runtime.sh: --map: unknown module '$k' (llm, tts, t2w, apm, vpm)" >&2; exit 2;;
  esac
done
if [ -n "$TTS_DEV" ] && [ "$TTS_DEV" != "${LLM_DEV:-0}" ]; then
  echo "omni_runtime.sh: tts=$TTS_DEV ignored -- the reference runtime loads the TTS model on the LLM's device (omni.cpp:3457); a device of its own is INTEGRATION.md 3a's patch" >&2
fi
DEVARGS=(); DEVENV=()
[ -n "$LLM_DEV" ] && DEVARGS+=(-mg "$LLM_DEV" -sm none)
[ -n "$APM_DEV" ] && DEVENV+=("MTMD_BACKEND_DEVICE=MI355X$APM_DEV")
[ -n "$VPM_DEV" ] && DEVENV+=("Omni_BACKEND_DEVICE=MI355X$VPM_DEV")
MAXTGT=${1:-24}; TURNS=${2:-1}; MODE=${3:-audio}
VIS=""; OMNI=""; [ "$MODE" = omni ] && { VIS="--vision"; OMNI="--omni"; }
if [ $PRINT = 1 ]; then
  echo "env GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so ${DEVENV[*]:-} oracle/_ref/omni-min -m gguf/MiniCPM-o-4_5-Q4_K_M.gguf --test case/audio_ $TURNS -ngl 99 ${DEVARGS[*]:-} --t2w-device gpu:$T2W_DEV --max-tgt $MAXTGT -c 4096 $OMNI"
  exit 0
fi
[ -f "$SET/gguf/MiniCPM-o-4_5-Q4_K_M.gguf" ] || python tools/make_synth_omni_set.py -o "$SET" --turns "$TURNS" $VIS > "$OUT/omni_set.log" 2>&1 || { tail -20 "$OUT/omni_set.log"; exit 1; }
cd "$SET"
for pass in 1 2; do           # (pass 1 pages the files in and builds the resident images; pass 2 is the one reported)
  rm -rf "$SET/out"
  env GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so MI355X_LOG_STATS=1 ${DEVENV[@]+"${DEVENV[@]}"} timeout 900 "$ROOT/oracle/_ref/omni-min" -m gguf/MiniCPM-o-4_5-Q4_K_M.gguf \
      --test case/audio_ "$TURNS" -ngl 99 ${DEVARGS[@]+"${DEVARGS[@]}"} --t2w-device "gpu:$T2W_DEV" --max-tgt "$MAXTGT" --out "$SET/out" -c 4096 $OMNI > "$OUT/omni_runtime_pass$pass.log" 2>&1
  echo "pass $pass exit $?"
done
cd "$ROOT"
python - <<PY
import json, sys
sys.path.insert(0, "tests")
import test_omni_runtime_gpu as t
for p in (1, 2):
    log = open("gpurun_out/omni_runtime_pass%d.log" % p, errors="replace").read()
    try:
        j = t.summarise(log); print("pass", p, json.dumps(j))
        if p == 2: open("gpurun_out/omni_runtime_summary.json", "w").write(json.dumps(j, indent=1) + "\n")
    except Exception as e:
        print("pass", p, "no summary:", e)
PY
grep -h "offloaded\|vision using\|vision chunks\|init_backend\|Audio Response\|\[mi355x\] MI355X0: graphs" "$OUT/omni_runtime_pass2.log" | cut -c1-220
