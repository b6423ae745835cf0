#!/bin/bash
# tools/gemm_pmc.sh M K N -- SQ / LDS counters of the prefill GEMM kernel at one shape (separate --pmc passes, kernel trace only)
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355X_GRAPHS=0
M=${1:-8192}; K=${2:-8192}; N=${3:-8192}
OUT=gpurun_out/gemm_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python tools/gemm_one.py $M $K $N 4 > "$OUT/$tag.txt" 2> "$OUT/$tag.err" < /dev/null
  echo "rc=$? $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
dur = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        d = dur[k]; d[0] += 1; d[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
for k, d in agg.items():
    print(k, " avg duration %.1f us (n=%d)" % (dur[k][1] / max(dur[k][0], 1) / 1e3, dur[k][0]))
    for c, (n, v) in sorted(d.items()):
        print("   %-32s per dispatch %16.0f   (n=%d)" % (c, v / n, n))
PY
