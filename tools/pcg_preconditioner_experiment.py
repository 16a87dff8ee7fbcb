"""CPU experiment (round 6): CG iterations of the reduced pose system S = Hpp - Hpl Hll^-1 Hlp of one L graph (5000 poses / 1000 landmarks,
g2o's first lambda) under block-Jacobi, the exact block-tridiagonal factor of the odometry chain, and wider block bands.
usage: python tools/pcg_preconditioner_experiment.py [seed]   (oracle linearisation + scipy; ~2 minutes, no GPU)"""
import os, sys, numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spla
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle.oracle import GraphProblem
from semantic_slam_amd.synth import make_graph
from test_chol_plan_cpu import _internal_order
g = make_graph(5000, 1000, seed=int(sys.argv[1]) if len(sys.argv) > 1 else 0)
gp = GraphProblem.from_synth(g)
U, b = gp.linearize()
U = sp.csr_matrix(U)
H = (U + U.T - sp.diags(U.diagonal())).tocsr()
idx, n = _internal_order(gp)
H = H[idx][:, idx].tocsr(); b = np.asarray(b)[idx]
lam = 1e-5 * H.diagonal().max()          # g2o's initial lambda
H = H + lam * sp.identity(n)
P = 6 * int((np.asarray(gp.vtype) == 0).sum() - 1)
Hpp, Hpl, Hll = H[:P, :P].tocsr(), H[:P, P:].tocsr(), H[P:, P:].tocsc()
Hll_inv = spla.inv(Hll)                   # block diagonal 3x3
S = (Hpp - Hpl @ Hll_inv @ Hpl.T).tocsr()
rhs = b[:P] - Hpl @ (Hll_inv @ b[P:])
print('S nnz', S.nnz, 'P', P)
def pcg(S, rhs, Minv, tol=1e-10, maxit=5000):
    x = np.zeros_like(rhs); r = rhs.copy(); z = Minv(r); p = z.copy(); rz = r @ z; r0 = np.sqrt(rhs @ rhs)
    for it in range(1, maxit + 1):
        Ap = S @ p; a = rz / (p @ Ap); x += a * p; r -= a * Ap
        if np.sqrt(r @ r) <= tol * r0: return it, x
        z = Minv(r); rz2 = r @ z; p = z + (rz2 / rz) * p; rz = rz2
    return maxit, x
# block Jacobi
nb = P // 6
D = sp.block_diag([np.linalg.inv(S[6*i:6*i+6, 6*i:6*i+6].toarray()) for i in range(nb)]).tocsr()
it, x = pcg(S, rhs, lambda r: D @ r); print('block Jacobi iterations', it)
# block tridiagonal part of S (pose i with pose i+1 in internal order = vertex id order = the odometry chain)
rows, cols = S.nonzero()
mask = np.abs(rows // 6 - cols // 6) <= 1
T = sp.csr_matrix((S.data[mask], (rows[mask], cols[mask])), shape=S.shape).tocsc()
lu = spla.splu(T)
it, x2 = pcg(S, rhs, lu.solve); print('block tridiagonal iterations', it)
for bw in (2, 4, 8, 16):
    mask = np.abs(rows // 6 - cols // 6) <= bw
    T = sp.csr_matrix((S.data[mask], (rows[mask], cols[mask])), shape=S.shape).tocsc()
    try:
        lu = spla.splu(T); it, _ = pcg(S, rhs, lu.solve); print('block band', bw, 'iterations', it)
    except Exception as e: print('band', bw, 'failed', e)
