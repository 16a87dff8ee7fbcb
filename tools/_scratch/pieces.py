import sys, numpy as np
sys.path.insert(0, '/root/repo/tools/_scratch')
from etree import *

def pieces(dim, order, struct, parent, cap):
    n = len(order); d = dim[np.array(order)]
    colsz = np.array([d[k]*d[k] + d[k]*sum(d[r] for r in struct[k]) + d[k] for k in range(n)])
    children = [[] for _ in range(n)]
    for k in range(n):
        if parent[k] >= 0: children[parent[k]].append(k)
    piece_of = [-1]*n   # piece id
    open_size = {}      # open piece id (root col) -> size
    plevel = {}         # piece id -> level
    root_of = {}        # piece id -> current root
    members = {}
    for j in range(n):
        ch = [piece_of[c] for c in children[j]]   # pieces of children (all still "open" candidates if their root is the child)
        openp = [p for p in set(ch) if p in open_size]
        openp.sort(key=lambda p: -open_size[p])
        size = colsz[j]; merged = []
        for p in openp:
            if size + open_size[p] <= cap: size += open_size[p]; merged.append(p)
        pid = j
        members[pid] = [j]; lvl = 0
        for p in openp:
            if p in merged:
                members[pid].extend(members.pop(p)); lvl = max(lvl, plevel[p]); del open_size[p]
            else:
                lvl = max(lvl, plevel[p] + 1); del open_size[p]   # closed
        # closed pieces among children that were already closed: (children whose piece isn't open)  -> level +1
        for c in children[j]:
            p = piece_of[c]
            if p not in merged and p not in openp:
                lvl = max(lvl, plevel[p] + 1)
        # but merged pieces internal levels: need max over closed descendants -> tracked in plevel of merged
        for c in members[pid]: piece_of[c] = pid
        open_size[pid] = size; plevel[pid] = lvl
    return piece_of, plevel, members

if __name__ == '__main__':
    g = make_graph(5000, 1000, seed=0)
    adj, dim = build_adj(g)
    order, cs = min_degree(adj)
    struct, parent, level, cf = stats("md", adj, dim, order)
    n = len(order); d = dim[np.array(order)]
    colsz = np.array([d[k]*d[k] + d[k]*sum(d[r] for r in struct[k]) + d[k] for k in range(n)])
    print("max col size doubles", colsz.max(), "top lnz", )
    for cap in (2048, 4096, 8192, 16384, 20000):
        po, pl, mem = pieces(dim, order, struct, parent, cap)
        ids = sorted(set(po))
        nlev = max(pl[p] for p in ids) + 1
        per = np.zeros(nlev, int); szs = np.zeros(nlev); fl = np.zeros(nlev)
        for p in ids:
            per[pl[p]] += 1
        # internal levels per piece
        maxint = 0
        for p in ids:
            ms = mem[p]; lv = {}
            for k in sorted(ms):
                lv.setdefault(k, 0)
                if parent[k] >= 0 and po[parent[k]] == p:
                    lv[parent[k]] = max(lv.get(parent[k], 0), lv[k] + 1)
            maxint = max(maxint, max(lv.values()) + 1)
        print(f"cap {cap}: pieces {len(ids)} piece-levels {nlev} per-level {per.tolist()[:12]}... max internal levels {maxint}")
