#!/usr/bin/env python3
"""tools/launch_ngrams.py LOG [graph_index] -- what the launches of an eager graph are made of: MI355X_LAUNCH_LOG=LOG makes run_nodes write one line per node
that launched (node index, op, launches, nodes folded, shape, sources); this prints the per-op launch counts of the largest graph (or the given one) and its most
frequent runs of 2..12 consecutive launching ops -- the repeated chains a fusion would pay for."""
import collections
import sys

OPN = {2: "ADD", 6: "SUB", 7: "MUL", 8: "DIV", 9: "SQR", 10: "SQRT", 11: "LOG", 12: "SIN", 13: "COS", 21: "CONCAT", 23: "NORM", 24: "RMS_NORM", 28: "MUL_MAT", 31: "SCALE", 33: "CPY", 34: "CONT",
       42: "SOFT_MAX", 51: "IM2COL", 16: "REPEAT", 80: "UNARY", 50: "CONV_T_1D", 59: "PAD", 60: "PAD_REFLECT", 40: "GET_ROWS", 15: "SUM_ROWS", 56: "POOL"}
graphs, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith("=="):
        graphs.append(cur); cur = []
        continue
    f = line.split()
    cur.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), f[4], f[5:]))
if cur:
    graphs.append(cur)
gi = int(sys.argv[2]) if len(sys.argv) > 2 else max(range(len(graphs)), key=lambda k: len(graphs[k]))
G = graphs[gi]
print("graphs:", [sum(e[2] for e in g) for g in graphs], "-> graph", gi, "launching nodes", len(G), "launches", sum(e[2] for e in G))
cnt = collections.Counter()
for e in G:
    cnt[OPN.get(e[1], "op%d" % e[1])] += e[2]
print("launches by op:", cnt.most_common())
sig = [OPN.get(e[1], "op%d" % e[1]) + ("" if e[1] != 34 else ("(n)" if e[5] and "n[" in e[5][0] else "(c)")) for e in G]
for n in (12, 10, 8, 6, 5, 4, 3, 2):
    c = collections.Counter(tuple(sig[i:i + n]) for i in range(len(sig) - n + 1))
    print("\n%d-grams:" % n)
    for k, v in c.most_common(8):
        print("  %5d x %s" % (v, " ".join(k)))
c = collections.Counter()
for e in G:
    if e[1] in (34, 33):
        c[(OPN[e[1]], e[4], e[5][0] if e[5] else "")] += e[2]
print("\ncopies by shape / source:")
for k, v in c.most_common(25):
    print("  %5d x %s" % (v, k))
