#!/bin/bash
# tools/mv2_pmc.sh [shape] -- SQ counters of the decode mat-vec kernels (tools/mmv2_lab.hip: the LDS-DMA engine k_mv2 beside the register-load
# k_mv1, same shape, same weights): where do the wave cycles go?  One rocprofv3 --pmc pass per counter group (8 SQ slots), kernel trace only.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
S=${1:-0}
OUT=gpurun_out/mv2_pmc_s$S
rm -rf "$OUT"; mkdir -p "$OUT"
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" "FETCH_SIZE"; do
  tag=$(echo $grp | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- build/mmv2_lab $S > "$OUT/$tag.txt" 2> "$OUT/$tag.err" < /dev/null
  echo "rc=$? $grp"
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:64]
        if "k_mv1" not in k and "k_mv2" not in k: continue
        a = agg[k][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print(k)
    for c, (n, v) in sorted(d.items()):
        print("   %-28s per dispatch %16.0f   (n=%d)" % (c, v / n, n))
PY
