import sys, numpy as np, heapq, time
sys.path.insert(0, '/root/repo')
from semantic_slam_amd.synth import make_graph

def build_adj(g):
    Np, Nl = g.n_poses, g.n_landmarks
    # rows: poses 1..Np-1 (pose 0 fixed) -> 0..Np-2 ; landmarks -> Np-1 + l
    n = Np - 1 + Nl
    adj = [set() for _ in range(n)]
    for i, j in g.odom_ij:
        if i == 0 or j == 0 or i == j: continue
        adj[i-1].add(j-1); adj[j-1].add(i-1)
    for p, l in g.lm_ij:
        if p == 0: continue
        adj[p-1].add(Np-1+l); adj[Np-1+l].add(p-1)
    dim = np.array([6]*(Np-1) + [3]*Nl)
    return adj, dim

def min_degree(adj, nodes=None, constrained_last=None):
    """returns order, cstruct (higher neighbours at elimination). adj: list of sets (copied)."""
    n = len(adj)
    adj = [set(a) for a in adj]
    done = [False]*n
    pq = [(len(adj[v]), v) for v in range(n)]
    heapq.heapify(pq)
    order = []; cs = [None]*n
    while pq:
        d, v = heapq.heappop(pq)
        if done[v] or d != len(adj[v]): continue
        done[v] = True; order.append(v)
        nb = adj[v]; cs[v] = sorted(nb)
        for u in nb:
            au = adj[u]; au |= nb; au.discard(u); au.discard(v)
            heapq.heappush(pq, (len(au), u))
        adj[v] = set()
    return order, cs

def symbolic(adj, order):
    """given order, compute column structures (elimination game via etree-based merging), parent, level."""
    n = len(adj); pos = [0]*n
    for k, v in enumerate(order): pos[v] = k
    # struct[k] = set of positions > k
    struct = [None]*n; parent = [-1]*n
    children = [[] for _ in range(n)]
    for k in range(n):
        v = order[k]
        s = set(pos[u] for u in adj[v] if pos[u] > k)
        for c in children[k]:
            s |= struct[c]
        s.discard(k)
        struct[k] = s
        if s:
            p = min(s); parent[k] = p; children[p].append(k)
    return struct, parent

def stats(name, adj, dim, order):
    n = len(adj)
    struct, parent = symbolic(adj, order)
    d = dim[np.array(order)]
    level = [0]*n
    for k in range(n):
        if parent[k] >= 0: level[parent[k]] = max(level[parent[k]], level[k]+1)
    nlev = max(level)+1
    lnz = 0; flops = 0; nblk = 0; nupd = 0
    colflops = np.zeros(n)
    for k in range(n):
        rows = sorted(struct[k]); dk = d[k]
        rd = [d[r] for r in rows]
        lnz += dk*dk + dk*sum(rd); nblk += 1 + len(rows)
        m = len(rows)
        nupd += m*(m+1)//2
        # flops of outer product updates from col k: sum_{i>=j} 2*di*dj*dk
        s = sum(rd); f = dk * (s*s + sum(x*x for x in rd))  # ~ 2*dk*sum_{i>=j} di dj
        flops += f; colflops[k] = f
    width = np.bincount(level, minlength=nlev)
    print(f"[{name}] n={n} nblk={nblk} lnz={lnz} ({lnz*8/1e6:.2f} MB) updates={nupd} flops={flops/1e6:.1f} MF levels={nlev}")
    return struct, parent, level, colflops

if __name__ == '__main__':
    g = make_graph(5000, 1000, seed=0)
    adj, dim = build_adj(g)
    deg = np.array([len(a) for a in adj])
    print("lm degree: min/med/max", deg[4999:].min(), np.median(deg[4999:]), deg[4999:].max(), " pose deg max", deg[:4999].max())
    t=time.time(); order, cs = min_degree(adj); print("md time", time.time()-t)
    struct, parent, level, cf = stats("min-degree", adj, dim, order)
    np.save('/tmp/md_level.npy', np.array(level))

def subtree_report(name, dim, order, struct, parent, level, cf, cap_doubles):
    n = len(order); d = dim[np.array(order)]
    colsz = np.array([d[k]*d[k] + d[k]*sum(d[r] for r in struct[k]) + d[k] for k in range(n)])
    sub = colsz.astype(np.int64).copy(); subf = cf.copy(); subn = np.ones(n, int)
    for k in range(n):
        if parent[k] >= 0:
            sub[parent[k]] += sub[k]; subf[parent[k]] += subf[k]; subn[parent[k]] += subn[k]
    # maximal subtrees with sub <= cap
    isroot = [(sub[k] <= cap_doubles) and (parent[k] < 0 or sub[parent[k]] > cap_doubles) for k in range(n)]
    roots = [k for k in range(n) if isroot[k]]
    intree = np.zeros(n, bool)
    for k in range(n-1, -1, -1):
        if isroot[k] or (parent[k] >= 0 and intree[parent[k]] ): intree[k] = True
    top = ~intree
    # levels inside top
    tl = {}
    for k in range(n):
        if top[k]:
            tl[k] = 0
    for k in range(n):
        if top[k] and parent[k] >= 0:
            tl[parent[k]] = max(tl[parent[k]], tl[k]+1)
    sizes = subn[roots]
    print(f"[{name}] cap={cap_doubles} subtrees={len(roots)} cols in subtrees={intree.sum()} flops in subtrees={cf[intree].sum()/1e6:.2f} MF;"
          f" top cols={top.sum()} top flops={cf[top].sum()/1e6:.2f} MF top levels={max(tl.values())+1 if tl else 0}"
          f" subtree cols med/max={np.median(sizes)}/{sizes.max()} depth max={max(level[k] for k in roots)+1}")

if __name__ == '__main__':
    for cap in (2048, 4096, 8192, 16384):
        subtree_report("min-degree", dim, order, struct, parent, level, cf, cap)
