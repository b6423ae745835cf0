#!/bin/bash
# tools/t2w_vocoder_slices.sh -- in-graph device time of every launch of the vocoder graph (2373 nodes) of a Token2Wav window, aggregated by op and shape (see t2w_slices.sh)
set -u
cd "$(dirname "$0")/.."
python tools/make_synth_omni_gguf.py --module t2w -o /tmp/t2w > /dev/null
export GGML_BACKEND_PATH=$PWD/llama.cpp-omni_amd/lib/libggml-mi355x.so
MI355X_GRAPHS=0 MI355X_LAUNCH_LOG=/tmp/ll.txt oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 2 > /dev/null 2>&1
SL=$(python3 - <<'PY'
graphs, cur = [], []
for line in open('/tmp/ll.txt'):
    if line.startswith('=='):
        graphs.append((int(line.split()[6]), cur)); cur = []
    else:
        cur.append(line.split())
n, G = [g for g in graphs if 2000 < g[0] < 3000][1]
idx = [int(e[0]) for e in G] + [n]
open('/tmp/slice_nodes.txt', 'w').write("\n".join(" ".join(e) for e in G))
print(str(n) + ":" + ",".join("%d:%d" % (idx[k], idx[k + 1]) for k in range(len(idx) - 1)))
PY
)
MI355X_GRAPH_SLICE="$SL" oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 2 2>&1 | grep slice | sed -E 's/.*launches, ([0-9.]+) us per pass.*/\1/' > /tmp/slice_times.txt
python3 - <<'PY'
import collections
OPN = {2: "ADD", 6: "SUB", 7: "MUL", 8: "DIV", 9: "SQR", 12: "SIN", 13: "COS", 21: "CONCAT", 23: "NORM", 28: "MUL_MAT", 31: "SCALE", 33: "CPY", 34: "CONT", 45: "SOFT_MAX", 51: "IM2COL", 19: "REPEAT", 80: "UNARY", 53: "CONV_T_1D", 59: "PAD", 60: "PAD_REFLECT", 44: "CLAMP", 11: "LOG", 10: "SQRT", 75: "LEAKY_RELU"}
t = [float(x) for x in open('/tmp/slice_times.txt')]
N = [l.split() for l in open('/tmp/slice_nodes.txt')]
print("launching nodes", len(N), "timed", len(t), "sum %.1f us" % sum(t))
agg = collections.defaultdict(lambda: [0, 0.0])
for e, us in zip(N, t):
    k = (OPN.get(int(e[1]), "op" + e[1]), e[4], "k" + e[2] + "f" + e[3])
    agg[k][0] += 1; agg[k][1] += us
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%9.1f us %5d x %7.2f  %s" % (us, n, us / n, k))
byop = collections.defaultdict(float)
for e, us in zip(N, t): byop[OPN.get(int(e[1]), "op" + e[1])] += us
print(sorted(byop.items(), key=lambda kv: -kv[1]))
PY
