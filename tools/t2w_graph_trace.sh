#!/bin/bash
# tools/t2w_graph_trace.sh -- rocprofv3 --kernel-trace of the reference's Token2Wav on the plug-in WITH hipGraph replay on (the steady-state windows): per-kernel
# durations and the gaps between consecutive kernels inside a replayed window, which is what a window's device time is made of (run on the GPU box)
set -u
cd "$(dirname "$0")/.."
ROOT=$PWD; OUT=$ROOT/gpurun_out/t2w_graph_trace; rm -rf $OUT; mkdir -p $OUT
python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$ROOT/llama.cpp-omni_amd/lib/libggml-mi355x.so
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t2w -- $ROOT/oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 6 > $OUT/run.txt 2>&1
echo "rc=$?"; tail -2 $OUT/run.txt | cut -c1-300
cd $ROOT
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/prof/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
print("dispatches", len(rows))
# windows: split at gaps > 2 ms
segs, cur = [], [rows[0]]
for a, b in zip(rows, rows[1:]):
    if int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) > 1000000:
        segs.append(cur); cur = []
    cur.append(b)
segs.append(cur)
print("segments (dispatches, span ms):", [(len(s), round((int(s[-1]["End_Timestamp"]) - int(s[0]["Start_Timestamp"])) / 1e6, 2)) for s in segs])
big = [s for s in segs if len(s) > 5000]
S = big[-2] if len(big) >= 2 else big[-1]
span = (int(S[-1]["End_Timestamp"]) - int(S[0]["Start_Timestamp"])) / 1e3
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in S) / 1e3
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(S, S[1:])]
print("window: %d dispatches, span %.1f us, kernel time %.1f us, gaps total %.1f us (avg %.2f us, median %.2f)" % (len(S), span, busy, sum(gaps) / 1e3, sum(gaps) / len(gaps) / 1e3, sorted(gaps)[len(gaps) // 2] / 1e3))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in S:
    k = r["Kernel_Name"][:70] + " g" + r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"]
    agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
with open("$OUT/window_kernels.txt", "w") as o:
    o.write("one replayed window: %d dispatches, span %.1f us, kernel time %.1f us, gaps %.1f us\n" % (len(S), span, busy, sum(gaps) / 1e3))
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        o.write("%9.1f us %6d calls %7.2f us avg  %s\n" % (t, n, t / n, k))
print(open("$OUT/window_kernels.txt").read()[:6000])
PY
