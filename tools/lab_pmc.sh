#!/bin/bash
# tools/lab_pmc.sh <shape> -- SQ counters of the mat-vec lab kernels (tools/mmv_lab.hip built with -DLAB_ONE): where do the wave cycles go?
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
S=${1:-0}
OUT=gpurun_out/lab_pmc
rm -rf "$OUT"; mkdir -p "$OUT"
for bin in mmv_lab_one0 mmv_lab_one3; do
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE"; do
  tag=${bin}_$(echo $grp | tr ' ' '_' | cut -c1-30)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- build/$bin $S > "$OUT/$tag.txt" 2> "$OUT/$tag.err" < /dev/null
  echo "rc=$? $bin $grp"
done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    b = f.split("/")[2][:12]
    for r in csv.DictReader(open(f)):
        k = b + " " + r["Kernel_Name"][:70]
        if "k_mv1" not in k and "k_mmv_pair" not in k and "k_mmv_multi" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-28s per dispatch %14.0f   (n=%d)" % (c, v / n, n))
PY
