#!/bin/bash
# tools/mmqt_pmc.sh M K N -- SQ / LDS / TCC counters of the tiled int8 prefill kernel (mmq_tile.hip) at one shape: separate --pmc passes, kernel trace only
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355X_GRAPHS=0 REP=4
M=${1:-4096}; K=${2:-4096}; N=${3:-512}
OUT=gpurun_out/mmqt_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python tools/mmq_tile_bench.py $M $K $N > "$OUT/$tag.txt" 2> "$OUT/$tag.err" < /dev/null
  echo "rc=$? $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "mmq_tile" not in k and "quantize_q8k_tile" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "mmq_tile" not in k and "quantize_q8k_tile" not in k: continue
        d = dur[k]; d[0] += 1; d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, d in agg.items():
    print(k, " avg duration %.1f us (n=%d)" % (dur[k][1] / max(dur[k][0], 1) / 1e3, dur[k][0]))
    for c, (n, v) in sorted(d.items()):
        print("   %-32s per dispatch %16.0f   (n=%d)" % (c, v / n, n))
PY
