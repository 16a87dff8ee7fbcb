import sys, numpy as np, time
sys.path.insert(0, '/root/repo/tools/_scratch')
from etree import *
from collections import deque

def bfs_levels(adj, nodes_set, start):
    lev = {start: 0}; q = deque([start]); order = [start]
    while q:
        v = q.popleft()
        for u in adj[v]:
            if u in nodes_set and u not in lev:
                lev[u] = lev[v] + 1; q.append(u); order.append(u)
    return lev, order

def components(adj, nodes_set):
    seen = set(); comps = []
    for s in nodes_set:
        if s in seen: continue
        lev, order = bfs_levels(adj, nodes_set, s)
        seen.update(order); comps.append(order)
    return comps

def nd_order(adj, nodes, leaf, out, depth=0):
    ns = set(nodes)
    if len(nodes) <= leaf:
        # local min-degree
        idx = {v: i for i, v in enumerate(nodes)}
        ladj = [set(idx[u] for u in adj[v] if u in ns) for v in nodes]
        o, _ = min_degree(ladj)
        out.extend(nodes[i] for i in o); return
    comps = components(adj, ns)
    if len(comps) > 1:
        for c in comps: nd_order(adj, c, leaf, out, depth)
        return
    # pseudo-peripheral start
    s = nodes[0]
    for _ in range(3):
        lev, order = bfs_levels(adj, ns, s); s = order[-1]
    lev, order = bfs_levels(adj, ns, s)
    L = max(lev.values()) + 1
    if L < 3:
        idx = {v: i for i, v in enumerate(nodes)}
        ladj = [set(idx[u] for u in adj[v] if u in ns) for v in nodes]
        o, _ = min_degree(ladj)
        out.extend(nodes[i] for i in o); return
    cnt = np.bincount(list(lev.values()), minlength=L)
    cum = np.cumsum(cnt)
    half = len(nodes) / 2
    # choose level near the middle minimizing size (within 35%-65%)
    cands = [l for l in range(1, L-1) if cum[l-1] >= 0.3*len(nodes) and cum[l-1] <= 0.7*len(nodes)]
    if not cands: cands = [int(np.searchsorted(cum, half))]
    cands = [min(max(c,1),L-2) for c in cands]
    m = min(cands, key=lambda l: (cnt[l], abs(cum[l]-half)))
    A = [v for v in nodes if lev[v] < m]; B = [v for v in nodes if lev[v] > m]; S = [v for v in nodes if lev[v] == m]
    # shrink separator: nodes of S without neighbours in B go to A
    Bs = set(B)
    S2 = []; 
    for v in S:
        if any((u in Bs) for u in adj[v]): S2.append(v)
        else: A.append(v)
    nd_order(adj, A, leaf, out, depth+1); nd_order(adj, B, leaf, out, depth+1)
    out.extend(S2)

if __name__ == '__main__':
    g = make_graph(5000, 1000, seed=0)
    adj, dim = build_adj(g)
    n = len(adj)
    deg = np.array([len(a) for a in adj])
    for hub_thr in (16, 1000):
      for leaf in (16, 32, 64, 128):
        hubs = [v for v in range(n) if deg[v] > hub_thr]
        rest = [v for v in range(n) if deg[v] <= hub_thr]
        out = []
        sys.setrecursionlimit(10000)
        nd_order(adj, rest, leaf, out)
        # hubs last, by degree
        out.extend(sorted(hubs, key=lambda v: deg[v]))
        assert len(out) == n and len(set(out)) == n
        name = f"nd hub>{hub_thr} leaf{leaf}"
        struct, parent, level, cf = stats(name, adj, dim, out)
        for cap in (4096, 8192):
            subtree_report(name, dim, out, struct, parent, level, cf, cap)
