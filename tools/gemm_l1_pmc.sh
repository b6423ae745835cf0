#!/bin/bash
# tools/gemm_l1_pmc.sh M K N -- vector-memory-path counters of the prefill GEMM kernel at one shape (separate --pmc passes, kernel trace only): requests from the CU's L1 to L2,
# L2 hits / misses, HBM bytes -- the evidence behind "the 512-column launches sit at the CU's line-request rate"
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355X_GRAPHS=0
M=${1:-4096}; K=${2:-4096}; N=${3:-512}
OUT=gpurun_out/gemm_l1_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
for grp in "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE GRBM_GUI_ACTIVE" "TCP_PENDING_STALL_CYCLES_sum TA_BUSY_avr"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  GEMM_COLD=1 timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python tools/gemm_one.py $M $K $N 12 > "$OUT/$tag.txt" 2> "$OUT/$tag.err" < /dev/null
  echo "rc=$? $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        if "gemm" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in agg.items():
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-32s per dispatch %14.0f   (n=%d)" % (c, v / n, n))
PY
