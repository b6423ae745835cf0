#!/bin/bash
# tools/profile_round.sh <tag> -- rocprofv3 passes whose summaries are kept under profiles/ (run on the GPU box via gpurun).
#   pass 1: --kernel-trace --stats of the bench (decode + pp512), graph replay off (rocprofv3 + hipGraph replay is unreliable here)
#   pass 2: --pmc FETCH_SIZE (own run, no trace domains besides kernel-trace) for HBM traffic per dispatch
set -u
TAG=${1:-r02}
cd "$(dirname "$0")/.."
export TMPDIR=/tmp MI355X_GRAPHS=0
OUT=gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python bench.py --steps 32 --warmup 8 --no-cpu-baseline --no-c3 --no-libllama > "$OUT/bench_under_rocprof.json" 2> "$OUT/trace.err"
echo "trace rc=$?"
MI355X_BENCH_NO_PP=1 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/pmc" -- python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-c3 --no-libllama > "$OUT/bench_under_pmc.json" 2> "$OUT/pmc.err"
echo "pmc rc=$?"
find "$OUT" -name "*.csv" | head -20
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    print("== kernel stats:", f)
    for r in rows[:14]:
        print("  %-90s calls=%6s avg_ns=%10s total_pct=%s" % (r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"]))
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r.get("Counter_Name") != "FETCH_SIZE": continue
        k = r["Kernel_Name"][:90]
        agg[k][0] += 1; agg[k][1] += float(r["Counter_Value"])
    print("== FETCH_SIZE (KiB as reported; x2 for wide streaming reads on gfx950, MI355X_MICROARCH.md) per dispatch:", f)
    for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        print("  %-90s n=%5d avg=%12.1f" % (k, n, v / n))
PY
