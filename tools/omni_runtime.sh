#!/bin/bash
# tools/omni_runtime.sh -- SURVEY.md 8 row g1 on the GPU box: the reference's omni runtime (oracle/_ref/omni-min = tools/omni/omni.cpp + modules + libllama)
# drives the plug-in over the synthetic full-size module set; log and summary into gpurun_out/ (copy the summary into profiles/).
#   tools/omni_runtime.sh [max_tgt] [turns] [omni]      (omni: media_type 2 -- every user turn after the first carries a 448 x 448 picture through vision.cpp)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD; SET=${OMNI_SET:-/tmp/omni_set}; OUT=$ROOT/gpurun_out; mkdir -p "$OUT"
MAXTGT=${1:-24}; TURNS=${2:-1}; MODE=${3:-audio}
VIS=""; OMNI=""; [ "$MODE" = omni ] && { VIS="--vision"; OMNI="--omni"; }
[ -f "$SET/gguf/MiniCPM-o-4_5-Q4_K_M.gguf" ] || python tools/make_synth_omni_set.py -o "$SET" --turns "$TURNS" $VIS > "$OUT/omni_set.log" 2>&1 || { tail -20 "$OUT/omni_set.log"; exit 1; }
cd "$SET"
for pass in 1 2; do           # (pass 1 pages the files in and builds the resident images; pass 2 is the one reported)
  rm -rf "$SET/out"
  GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so MI355X_LOG_STATS=1 timeout 900 "$ROOT/oracle/_ref/omni-min" -m gguf/MiniCPM-o-4_5-Q4_K_M.gguf \
      --test case/audio_ "$TURNS" -ngl 99 --t2w-device gpu:0 --max-tgt "$MAXTGT" --out "$SET/out" -c 4096 $OMNI > "$OUT/omni_runtime_pass$pass.log" 2>&1
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
