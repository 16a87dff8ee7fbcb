"""Structure of the Cholesky plan of one L graph (host only, no GPU): pieces, columns per piece, update-matrix sizes, launches, LDS needs.
usage: python tools/plan_stats.py"""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semantic_slam_amd import GraphSLAM, _lib
from semantic_slam_amd.synth import make_graph
from tests.chol_plan_exec import Plan
lib = _lib.load_library()
G = GraphSLAM.from_synth(make_graph(5000, 1000, seed=0))
P = Plan(lib, [G])
pc = P.piece
print('pieces', len(pc), 'cols', P.ncol, 'lnz', P.lnz, 'unz', P.unz, 'h', P.h_total)
nc = pc['nc']
print('cols/piece hist', np.bincount(np.minimum(nc, 40))[:41])
col = P.col
dims = col['dim']
# per piece: #landmark cols, #pose cols
lm_only = 0; 
for p in pc[:]:
    d = dims[p['c0']:p['c0']+p['nc']]
print('piece lsize stats', np.percentile(pc['lsize'], [10,50,90,99,100]))
print('piece nuu (U update records)', np.percentile(pc['nuu'], [10,50,90,99,100]), pc['nuu'].sum())
print('piece nuit', np.percentile(pc['nuit'], [10,50,90,99,100]))
print('piece nb blocks', np.percentile(pc['nb'], [10,50,90,99,100]), pc['nb'].sum())
print('nu_i internal updates', pc['nu_i'].sum(), 'nas', pc['nas'].sum(), 'nus', pc['nus'].sum())
ptr = P.plv_ptr
print('launch sizes', np.diff(ptr), 'tail pieces', len(P.tail_pieces))
# col dims in leaf launch
first = P.plv_pieces[ptr[0]:ptr[1]]
n3 = sum(int((dims[pc[p]['c0']:pc[p]['c0']+pc[p]['nc']] == 3).sum()) for p in first)
n6 = sum(int((dims[pc[p]['c0']:pc[p]['c0']+pc[p]['nc']] == 6).sum()) for p in first)
print('first launch: landmark cols', n3, 'pose cols', n6)
# where are landmark columns in the elimination order
order_dims = dims
print('landmark cols total', (dims==3).sum(), 'positions percentiles', np.percentile(np.nonzero(dims==3)[0], [0,10,50,90,100]))
print('nb per col: landmark', np.percentile(col['nb'][dims==3],[10,50,90,100]), 'pose', np.percentile(col['nb'][dims==6],[10,50,90,100]))
print('plv_lds_f', P.plv_lds_f, 'plv_lds_b', P.plv_lds_b, 'tail', P.tail_lds_f, P.tail_lds_b)
