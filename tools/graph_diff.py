#!/usr/bin/env python3
"""tools/graph_diff.py A.dump B.dump [--graph-a I] [--graph-b J] -- compare two MI355X_DUMP_GRAPH files: the multiset of compute nodes (op, sub-op, result type and
shape, source types and shapes; views / reshapes / permutes / transposes are layout-only and left out) of one graph of each.  Used to check this repo's Python
mirrors of the omni encoders (llama.cpp-omni_amd/encoders.py) against what the reference's own builders (audition.cpp / vision.cpp) submit."""
import argparse
import collections

NAMES = {0: "NONE", 1: "DUP", 2: "ADD", 6: "SUB", 7: "MUL", 8: "DIV", 9: "SQR", 10: "SQRT", 11: "LOG", 12: "SIN", 13: "COS", 15: "SUM_ROWS", 19: "REPEAT", 21: "CONCAT", 23: "NORM",
         24: "RMS_NORM", 28: "MUL_MAT", 31: "SCALE", 33: "CPY", 34: "CONT", 35: "RESHAPE", 36: "VIEW", 37: "PERMUTE", 38: "TRANSPOSE", 39: "GET_ROWS", 41: "SET_ROWS", 45: "SOFT_MAX",
         47: "ROPE", 49: "CLAMP", 50: "CONV_TRANSPOSE_1D", 51: "IM2COL", 58: "POOL_1D", 59: "POOL_2D", 62: "PAD", 63: "PAD_REFLECT_1D", 65: "ARANGE", 66: "TIMESTEP_EMBEDDING",
         68: "LEAKY_RELU", 69: "FLASH_ATTN_EXT", 80: "UNARY", 89: "GLU"}
LAYOUT = {35, 36, 37, 38, 0}


def load(path):
    graphs, cur = [], None
    for line in open(path):
        if line.startswith("graph "):
            cur = []; graphs.append(cur)
        elif cur is not None and line.strip():
            head = line.split(" | ")[0].split()
            op = int(head[0])
            if op in LAYOUT:
                continue
            cur.append((NAMES.get(op, str(op)), head[1]) + tuple(x.strip() for x in [" ".join(head[2:])] + line.strip().split(" | ")[1:]))
    return graphs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("a"); ap.add_argument("b")
    ap.add_argument("--graph-a", type=int, default=-1); ap.add_argument("--graph-b", type=int, default=-1)
    a = ap.parse_args()
    ga, gb = load(a.a)[a.graph_a], load(a.b)[a.graph_b]
    ca, cb = collections.Counter(ga), collections.Counter(gb)
    print(f"A: {len(ga)} compute nodes ({a.a}), B: {len(gb)} compute nodes ({a.b})")
    ops_a, ops_b = collections.Counter(n[0] for n in ga), collections.Counter(n[0] for n in gb)
    print("per op (A / B):", ", ".join(f"{k} {ops_a.get(k, 0)}/{ops_b.get(k, 0)}" for k in sorted(set(ops_a) | set(ops_b))))
    only_a, only_b = ca - cb, cb - ca
    print(f"identical (op, types, shapes) nodes: {sum((ca & cb).values())}; only in A: {sum(only_a.values())}; only in B: {sum(only_b.values())}")
    for name, d in (("only in A", only_a), ("only in B", only_b)):
        for k, v in sorted(d.items(), key=lambda kv: (-kv[1], kv[0]))[:40]:
            print(f"  {name}: {v:4d} x {k[0]}{'(' + k[1] + ')' if k[1] != '-1' else ''} {' <- '.join(k[2:3])} <- {' , '.join(k[3:])}")
