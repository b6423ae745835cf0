#!/bin/bash
# tools/pp512_prof.sh [extra args of pp512_prof.py] -- rocprofv3 --kernel-trace --stats of ONLY the prefill leg (bench.py's pp512 graph), summary on stdout
cd "$(dirname "$0")/.."
ROOT=$PWD; TAG=$(echo "pp$*" | tr ' -' '__'); OUT=$ROOT/gpurun_out/$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o pp -- python $ROOT/tools/pp512_prof.py "$@" > $OUT/run.txt 2>&1
cd $ROOT
tail -1 $OUT/run.txt
python - <<PY
import csv, glob
f = glob.glob("$OUT/prof/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total device time %.2f ms over all reps (7 graph submissions: 2 warm-up + 5 timed)" % (tot / 1e6))
for r in rows[:22]:
    print("%8.3f ms %6s calls %9.2f us avg %5.1f%%  %s" % (float(r["TotalDurationNs"]) / 1e6, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["Percentage"]), r["Name"][:110]))
PY
