#!/bin/bash
# tools/q6k_fetch_pmc.sh -- the Q6_K ffn_down "over-fetch" (FETCH_SIZE x 2 = 1.11 x the matrix at K = 12288, 1.00 x for the Q6_K lm-head at K = 4096): which fabric requests
# make it up?  TCC_EA0_RDREQ (all read requests of the L2 towards the fabric) against TCC_EA0_RDREQ_32B (the 32-byte ones) and FETCH_SIZE itself, per dispatch of
# tools/bin/mmv2_lab shapes 7 (Q6_K 4096 x 12288), 4 (Q4_K 4096 x 12288), 8 (Q6_K lm-head), 5 (Q4_K 151936 x 4096); one --pmc pass per group, kernel trace only.
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=gpurun_out/q6k_fetch
rm -rf "$OUT"; mkdir -p "$OUT"
for S in 7 4 8 5; do
  for grp in "FETCH_SIZE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_READ_REQ_sum TCC_REQ_sum" "TCC_HIT_sum TCC_MISS_sum"; do
    tag=s${S}_$(echo $grp | tr ' ' '_' | cut -c1-40)
    timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d "$OUT/$tag" -- tools/bin/mmv2_lab_one $S > "$OUT/$tag.txt" 2> "$OUT/$tag.err" < /dev/null
    echo "rc=$? shape $S: $grp"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    shape = re.search(r"/s(\d+)_", f).group(1)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_mv2" not in k: continue
        a = agg[(shape, k[:60])][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, d in sorted(agg.items()):
    print("shape", k[0], k[1])
    for c, (n, v) in sorted(d.items()):
        print("   %-28s per dispatch %16.0f   (n=%d)" % (c, v / n, n))
PY
