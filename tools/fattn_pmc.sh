#!/bin/bash
# tools/fattn_pmc.sh -- SQ / LDS counters of the prefill flash-attention kernel (tools/gemm_bench.py's attention shapes), separate --pmc passes
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355X_GRAPHS=0
OUT=gpurun_out/fattn_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS SQ_INSTS_VMEM"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- python tools/gemm_bench.py --attn-only > "$OUT/$tag.txt" 2> "$OUT/$tag.err" < /dev/null
  echo "rc=$? $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:50] + " grid " + r.get("Grid_Size", r.get("Grid_Size_X", ""))
        if "fattn_mma" not in k and "fattn_dma" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-28s per dispatch %16.0f   (n=%d)" % (c, v / n, n))
PY
