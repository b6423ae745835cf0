#!/bin/bash
# tools/fattn_prof.sh -- kernel durations of tools/fattn_bench.py under rocprofv3 (graph replay off); prints the per-kernel averages
set -u
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355X_GRAPHS=0
OUT=gpurun_out/prof_fattn
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -- python tools/fattn_bench.py > "$OUT/bench.txt" 2> "$OUT/trace.err" < /dev/null
echo "trace rc=$?"; cat "$OUT/bench.txt"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_trace.csv", recursive=True):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[(r["Kernel_Name"][:70], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", ""))].append(d)
    for k, v in sorted(agg.items()):
        v.sort()
        print("%-72s grid=%8s n=%5d  median %8.2f us  min %8.2f us" % (k[0], k[1], len(v), v[len(v) // 2] / 1e3, v[0] / 1e3))
PY
