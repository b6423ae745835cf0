#!/bin/bash
# tools/t2w_slices.sh -- in-graph device time of every launch of one DiT block of the reference's Token2Wav window graph on the plug-in: a launch log of the eager
# window gives the launching nodes, MI355X_GRAPH_SLICE captures each [node_k, node_k+1) ten times over and replays it (graph.cpp) -- rocprofv3 cannot trace replays here.
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
n, G = [g for g in graphs if g[0] > 20000][1]
sig = [int(e[1]) for e in G]
pat = [80, 28, 2]
pos = [i for i in range(len(sig) - 3) if sig[i:i + 3] == pat and G[i][4] == '[512,1,2,1]']
a, b = pos[40], pos[41]
idx = [int(e[0]) for e in G[a:b + 1]]
open('/tmp/slice_nodes.txt', 'w').write("\n".join(" ".join(e) for e in G[a:b]))
print(str(n) + ":" + ",".join("%d:%d" % (idx[k], idx[k + 1]) for k in range(len(idx) - 1)) + ",%d:%d" % (idx[0], idx[-1]))
PY
)
echo "slices: $SL" | cut -c1-200
MI355X_GRAPH_SLICE="$SL" oracle/_ref/t2w-min /tmp/t2w /tmp/t2w.f32 gpu --windows 2 2>&1 | grep slice > /tmp/slice_times.txt
paste -d' ' <(sed -E 's/.*launches, ([0-9.]+) us per pass.*/\1 us/' /tmp/slice_times.txt | head -n -1) /tmp/slice_nodes.txt | cut -c1-230
tail -1 /tmp/slice_times.txt
