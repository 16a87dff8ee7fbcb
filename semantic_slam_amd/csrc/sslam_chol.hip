// Sparse block Cholesky for the LM normal equations on gfx950 (FP64).
//
// Replaces g2o's BlockSolverX + LinearSolverCSparse pair that the reference selects with
// "lm_var" (reference src/ps_graph_slam/graph_slam.cpp:27,67-73): fill-reducing
// ordering on the BLOCK pattern, symbolic factorisation once per structure, numeric
// factorisation every LM trial, triangular solves.  MI355X design:
//   * symbolic phase on the host (minimum degree with explicit fill on the 6x6/3x3 block graph,
//     elimination tree levels, per-target-block update lists),
//   * numeric phase level-scheduled on the device: one wave per block column, left-looking
//     *gather* form (every L block is written by exactly one wave -> deterministic, no atomics),
//     column staged in LDS, forward substitution fused into the factorisation (b is carried as
//     an extra block row), backward substitution as a second top-down sweep,
//   * all graphs of a batch share the level launches (levels are concatenated across graphs).
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstring>
#include <queue>
#include <unordered_map>
#include <vector>

#include "../../include/sslam.h"
#include "graph_engine.hpp"

namespace sslam {

struct CholView {
  int ncol, nlevels, dim;
  const int* col_xoff;   // [ncol] offset in the unknown vector
  const int* col_dim;    // [ncol] 6 | 3
  const int* col_graph;  // [ncol]
  const int* bp;         // [ncol+1] block range of each column (first = diagonal)
  const int* boff;       // [nblk] offset into Lval
  const int* brow;       // [nblk] column id of the block's row
  const int* bsrc;       // [nblk] offset into H (doubles from Hpp_diag) or -1
  const unsigned char* bfmt;  // [nblk] 0: H[r*dj+c], 1: H[c*di+r]
  const int* up;         // [nblk+1] update list range
  const int* ua;         // offset of L_ik
  const int* ub;         // offset of L_jk
  const int* uk;         // column id k
  const int* ux;         // x offset of column k
  const unsigned char* udk;  // dimension of column k
  const int* lvl_cols;   // columns grouped by level
  double* Lval;
  double* y;             // forward-substituted rhs [dim]
  int* fail;             // [B]
};

struct CholPlan {
  CholView C{};
  std::vector<int> lvl_ptr;
  std::vector<int> lvl_maxlist;  // longest update list among the level's blocks
  std::vector<int> lvl_maxEt;    // largest column (entries + rhs) of the level
  std::vector<void*> allocs;
  int max_col_entries = 0;
  int64_t lnz = 0;
  double* d_multi_y = nullptr;  // scratch for multi-rhs solves
  double* d_multi_x = nullptr;
  int multi_cap = 0;
};

void chol_plan_free(CholPlan* p) {
  if (!p) return;
  for (void* a : p->allocs) (void)hipFree(a);
  if (p->d_multi_y) (void)hipFree(p->d_multi_y);
  if (p->d_multi_x) (void)hipFree(p->d_multi_x);
  delete p;
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
// One 256-thread workgroup per block column j of the current level:
//   S = A(:,j) + lambda I - sum_k L(:,k) L(j,k)^T   (gather over precomputed update lists; long lists
//                                                    are split over up to 4 waves and reduced in LDS
//                                                    in a fixed order -> deterministic)
//   L(j,j) = chol(S(j,j)),  L(i,j) = S(i,j) L(j,j)^-T,   y(j) = L(j,j)^-1 (b(j) - sum_k L(j,k) y(k))
// tail of a column: in-register dense Cholesky of the D x D diagonal block, forward-substituted rhs,
// triangular solves of the off-diagonal rows
template <int D, int NT>
__device__ __forceinline__ void chol_tail(const double* sm, int csize, double* Lw, double* yout, int* fail, int tid) {
  double a[D * D], t[D];
#pragma unroll
  for (int q = 0; q < D * D; ++q) a[q] = sm[q];
#pragma unroll
  for (int q = 0; q < D; ++q) t[q] = sm[csize + q];
  bool ok = true;
#pragma unroll
  for (int c = 0; c < D; ++c) {
    double d = a[c * D + c];
#pragma unroll
    for (int s = 0; s < c; ++s) d -= a[c * D + s] * a[c * D + s];
    if (!(d > 0)) { ok = false; d = 1.0; }
    d = sqrt(d);
    a[c * D + c] = d;
#pragma unroll
    for (int r = c + 1; r < D; ++r) {
      double x = a[r * D + c];
#pragma unroll
      for (int s = 0; s < c; ++s) x -= a[r * D + s] * a[c * D + s];
      a[r * D + c] = x / d;
    }
#pragma unroll
    for (int s = c + 1; s < D; ++s) a[c * D + s] = 0.0;  // strict upper = 0
  }
  if (tid == 0) {
    if (!ok) *fail = 1;
#pragma unroll
    for (int r = 0; r < D; ++r) {
      double v = t[r];
#pragma unroll
      for (int s = 0; s < r; ++s) v -= a[r * D + s] * t[s];
      t[r] = v / a[r * D + r];
    }
#pragma unroll
    for (int q = 0; q < D * D; ++q) Lw[q] = a[q];
#pragma unroll
    for (int q = 0; q < D; ++q) yout[q] = t[q];
  }
  const int nrows_off = (csize - D * D) / D;
  for (int row = tid; row < nrows_off; row += NT) {
    const double* v = sm + D * D + row * D;
    double x[D];
#pragma unroll
    for (int c = 0; c < D; ++c) {
      double w = v[c];
#pragma unroll
      for (int s = 0; s < c; ++s) w -= x[s] * a[c * D + s];
      x[c] = w / a[c * D + c];
    }
    double* o = Lw + D * D + row * D;
#pragma unroll
    for (int c = 0; c < D; ++c) o[c] = x[c];
  }
}

// NT = 256 (4 waves) for wide levels, 1024 (16 waves) for the narrow top levels whose update lists
// are long.  Work item of a wave = (target block, slice of its update list).  For every update the
// wave loads the two source blocks once with contiguous 8-byte lanes (2 x <=288 B), parks them in its
// LDS scratch and the di x dj entry lanes read rows from LDS (broadcast), instead of every entry lane
// gathering 12 scalars from L2.  Slices are reduced through LDS in a fixed order -> deterministic.
constexpr int kPartDoubles = 4096;  // LDS budget for the per-slice partial sums
template <int NT>
__global__ __launch_bounds__(NT) void k_chol_level(BatchView V, CholView C, int lvl_begin) {
  extern __shared__ double sm[];  // [Et] column + rhs entries | [nslice][Et] partial sums | [NW][80] wave scratch
  constexpr int NW = NT / 64;
  const int j = C.lvl_cols[lvl_begin + blockIdx.x];
  const int g = C.col_graph[j];
  if (!V.lm[g].in_trial) return;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63;
  const int dj = C.col_dim[j];
  const int b0 = C.bp[j], b1 = C.bp[j + 1];
  const int nb = b1 - b0;
  const int base = C.boff[b0];
  int csize;
  {
    const int last = b1 - 1;
    csize = C.boff[last] - base + C.col_dim[C.brow[last]] * dj;
  }
  const int Et = csize + dj;  // the last dj "entries" are the forward-substitution rhs
  int maxlen = 0;
  for (int b = b0; b < b1; ++b) maxlen = max(maxlen, C.up[b + 1] - C.up[b]);
  const int nslice = max(1, min(min(NW, (maxlen + 3) >> 2), kPartDoubles / Et));
  const double lambda = V.lm[g].lambda;
  const double* __restrict__ H = V.Hpp_diag;
  const double* __restrict__ L = C.Lval;
  const double* __restrict__ Y = C.y;
  double* part = sm + Et;
  double* scr = sm + Et + nslice * Et + wave * 80;
  const int nitems = nb * nslice;
  for (int item = wave; item < nitems; item += NW) {
    const int b = b0 + item / nslice, sl = item - (item / nslice) * nslice;
    const int di = C.col_dim[C.brow[b]];
    const int nE = di * dj;
    const bool diag = (b == b0);
    const int r = lane / dj, c = lane - r * dj;          // entry lanes: lane < nE
    const int ry = lane - 40;                            // rhs lanes: 40 .. 40 + dj - 1 (diagonal block only)
    double acc = 0;
    const int u1 = C.up[b + 1];
    int u = C.up[b] + sl;
    // software pipeline: registers hold the next update's elements while the current one is consumed from LDS
    double v0 = 0, v1 = 0;
    int dk = 0, nA = 0, nB = 0;
    auto fetch = [&](int uu) {
      dk = C.udk[uu];
      nA = di * dk; nB = dj * dk;
      const double* A = L + C.ua[uu];
      const double* B = L + C.ub[uu];
      const double* Yk = Y + C.ux[uu];
      const int i0 = lane, i1 = lane + 64;
      v0 = i0 < nA ? A[i0] : (i0 < nA + nB ? B[i0 - nA] : ((diag && i0 < nA + nB + dk) ? Yk[i0 - nA - nB] : 0.0));
      v1 = i1 < nA ? A[i1] : (i1 < nA + nB ? B[i1 - nA] : ((diag && i1 < nA + nB + dk) ? Yk[i1 - nA - nB] : 0.0));
    };
    if (u < u1) fetch(u);
    while (u < u1) {
      const int cdk = dk, cnA = nA, cnB = nB;
      scr[lane] = v0;
      if (lane < 16) scr[64 + lane] = v1;
      u += nslice;
      if (u < u1) fetch(u);
      if (lane < nE) {
        const double* pa = scr + r * cdk;
        const double* pb = scr + cnA + c * cdk;
        double s = pa[0] * pb[0] + pa[1] * pb[1] + pa[2] * pb[2];
        if (cdk == 6) s += pa[3] * pb[3] + pa[4] * pb[4] + pa[5] * pb[5];
        acc += s;
      } else if (diag && ry >= 0 && ry < dj) {
        const double* pa = scr + ry * cdk;
        const double* yk = scr + cnA + cnB;
        double s = pa[0] * yk[0] + pa[1] * yk[1] + pa[2] * yk[2];
        if (cdk == 6) s += pa[3] * yk[3] + pa[4] * yk[4] + pa[5] * yk[5];
        acc += s;
      }
    }
    if (lane < nE) part[sl * Et + (C.boff[b] - base) + lane] = acc;
    else if (diag && ry >= 0 && ry < dj) part[sl * Et + csize + ry] = acc;
  }
  __syncthreads();
  for (int e = tid; e < Et; e += NT) {
    double v;
    if (e < csize) {
      int b = b0;
      while (b + 1 < b1 && C.boff[b + 1] - base <= e) ++b;
      const int le = e - (C.boff[b] - base);
      const int r = le / dj, c = le - r * dj;
      const int di = C.col_dim[C.brow[b]];
      const int src = C.bsrc[b];
      v = 0;
      if (src >= 0) v = C.bfmt[b] ? H[src + c * di + r] : H[src + r * dj + c];
      if (b == b0 && r == c) v += lambda;
    } else {
      v = V.bvec[C.col_xoff[j] + (e - csize)];
    }
    for (int s = 0; s < nslice; ++s) v -= part[s * Et + e];
    sm[e] = v;
  }
  __syncthreads();
  // ---- diagonal block: every thread factors its own register copy (D^3/3 flops, no LDS latency chain,
  //      no further barriers); then one thread per off-diagonal row solves x L_jj^T = v and stores to HBM
  double* Lw = C.Lval + base;
  if (dj == 6) chol_tail<6, NT>(sm, csize, Lw, C.y + C.col_xoff[j], C.fail + g, tid);
  else chol_tail<3, NT>(sm, csize, Lw, C.y + C.col_xoff[j], C.fail + g, tid);
}

// forward substitution only (multi right-hand-side form, used for marginals): y_j = L_jj^-1 (b_j - sum_k L_jk y_k)
__global__ __launch_bounds__(64) void k_chol_forward_level(CholView C, int lvl_begin, const double* __restrict__ rhs, double* __restrict__ y) {
  __shared__ double t[8];
  const int j = C.lvl_cols[lvl_begin + blockIdx.x];
  const size_t vo = (size_t)blockIdx.y * C.dim;
  const int lane = threadIdx.x;
  const int dj = C.col_dim[j];
  const int b0 = C.bp[j];
  const double* __restrict__ L = C.Lval;
  if (lane < dj) {
    double a = rhs[vo + C.col_xoff[j] + lane];
    for (int u = C.up[b0]; u < C.up[b0 + 1]; ++u) {
      const int k = C.uk[u];
      const int dk = C.col_dim[k];
      const double* pa = L + C.ua[u] + lane * dk;
      const double* yk = y + vo + C.col_xoff[k];
      for (int q = 0; q < dk; ++q) a -= pa[q] * yk[q];
    }
    t[lane] = a;
  }
  __syncthreads();
  if (lane == 0) {
    const double* D = L + C.boff[b0];
    for (int r = 0; r < dj; ++r) {
      double a = t[r];
      for (int s = 0; s < r; ++s) a -= D[r * dj + s] * t[s];
      t[r] = a / D[r * dj + r];
    }
    for (int r = 0; r < dj; ++r) y[vo + C.col_xoff[j] + r] = t[r];
  }
}

// backward substitution, one wave per column, levels top-down: x_j = L_jj^-T (y_j - sum_i L_ij^T x_i)
__global__ __launch_bounds__(64) void k_chol_backward_level(CholView C, int lvl_begin, const double* __restrict__ y, double* __restrict__ x,
                                                           const LmState* __restrict__ lm) {
  __shared__ double t[8];
  const int j = C.lvl_cols[lvl_begin + blockIdx.x];
  if (lm && !lm[C.col_graph[j]].in_trial) return;
  const size_t vo = (size_t)blockIdx.y * C.dim;
  const int lane = threadIdx.x;
  const int dj = C.col_dim[j];
  const int b0 = C.bp[j], b1 = C.bp[j + 1];
  const double* __restrict__ L = C.Lval;
  const int c = lane & 7, q = lane >> 3;
  double acc = 0;
  if (c < dj) {
    for (int b = b0 + 1 + q; b < b1; b += 8) {
      const int i = C.brow[b];
      const int di = C.col_dim[i];
      const double* Bk = L + C.boff[b];
      const double* xi = x + vo + C.col_xoff[i];
      for (int r = 0; r < di; ++r) acc += Bk[r * dj + c] * xi[r];
    }
  }
  acc += __shfl_xor(acc, 8, 64);
  acc += __shfl_xor(acc, 16, 64);
  acc += __shfl_xor(acc, 32, 64);
  if (q == 0 && c < dj) t[c] = y[vo + C.col_xoff[j] + c] - acc;
  __syncthreads();
  if (lane == 0) {
    const double* D = L + C.boff[b0];
    for (int r = dj - 1; r >= 0; --r) {
      double a = t[r];
      for (int s = r + 1; s < dj; ++s) a -= D[s * dj + r] * t[s];
      t[r] = a / D[r * dj + r];
    }
    for (int r = 0; r < dj; ++r) x[vo + C.col_xoff[j] + r] = t[r];
  }
}

__global__ void k_chol_begin(BatchView V, CholView C) {  // clear failure flags of the graphs being solved
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B && V.lm[g].in_trial) C.fail[g] = 0;
}
__global__ void k_chol_end(BatchView V, CholView C) {  // publish failures through the solver-agnostic flag
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g < V.B && V.lm[g].in_trial) V.pcg_fail[g] = C.fail[g];
}

// ------------------------------------------------------------------------------------------------
// host symbolic phase
// ------------------------------------------------------------------------------------------------
namespace {

struct GraphSym {
  std::vector<int> order;                  // elimination order (local node ids)
  std::vector<std::vector<int>> cstruct;   // per node: higher-ordered neighbours at elimination time
};

// minimum degree with explicit fill (the block graphs here have ~1e4 nodes and fill ~1.7x)
void min_degree(int n, std::vector<std::vector<int>>& adj, GraphSym& out) {
  std::vector<char> done(n, 0);
  using Item = std::pair<int, int>;  // (degree, node); lazy deletion
  std::priority_queue<Item, std::vector<Item>, std::greater<Item>> pq;
  for (int v = 0; v < n; ++v) { std::sort(adj[v].begin(), adj[v].end()); pq.push({(int)adj[v].size(), v}); }
  out.order.clear(); out.order.reserve(n);
  out.cstruct.assign(n, {});
  std::vector<int> merged;
  while (!pq.empty()) {
    const Item it = pq.top(); pq.pop();
    const int v = it.second;
    if (done[v] || it.first != (int)adj[v].size()) continue;
    done[v] = 1;
    out.order.push_back(v);
    std::vector<int>& nb = adj[v];
    out.cstruct[v] = nb;
    for (int u : nb) {
      std::vector<int>& au = adj[u];
      // au <- (au U nb) \ {u, v}   (both sorted)
      merged.clear();
      merged.reserve(au.size() + nb.size());
      std::set_union(au.begin(), au.end(), nb.begin(), nb.end(), std::back_inserter(merged));
      au.clear();
      for (int w : merged) if (w != u && w != v) au.push_back(w);
      pq.push({(int)au.size(), u});
    }
    std::vector<int>().swap(adj[v]);
  }
}

template <typename T>
int up_to_dev(CholPlan& P, hipStream_t s, const std::vector<T>& h, const T** out) {
  void* p = nullptr;
  const size_t n = std::max<size_t>(h.size(), 1);
  SSLAM_HIP_TRY(hipMalloc(&p, n * sizeof(T)));
  P.allocs.push_back(p);
  if (!h.empty()) SSLAM_HIP_TRY(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
  *out = (const T*)p;
  return 0;
}

}  // namespace

int chol_plan_build(Batch& b) {
  if (b.chol) { chol_plan_free(b.chol); b.chol = nullptr; }
  SSLAM_HIP_TRY(hipSetDevice(b.device));
  CholPlan* P = new CholPlan();
  b.chol = P;
  const BatchView& V = b.V;
  const int nPr = V.nPr, nLr = V.nLr, nrow = nPr + nLr;
  // global internal rows: pose row p -> p ; landmark row l -> nPr + l
  auto row_dim = [&](int r) { return r < nPr ? 6 : 3; };
  auto row_xoff = [&](int r) { return r < nPr ? 6 * r : 6 * nPr + 3 * (r - nPr); };
  // H block lookup: (min row, max row) -> (offset, stored as [min][max])
  std::unordered_map<uint64_t, int> hoff;
  hoff.reserve(b.ppoff.size() + b.plblk.size());
  auto key = [](int a, int c) { return ((uint64_t)(uint32_t)a << 32) | (uint32_t)c; };
  for (size_t i = 0; i < b.ppoff.size(); ++i) hoff[key(b.ppoff[i].first, b.ppoff[i].second)] = (int)(b.hpp_off_base + (int64_t)i * 36);
  for (size_t i = 0; i < b.plblk.size(); ++i) hoff[key(b.plblk[i].first, nPr + b.plblk[i].second)] = (int)(b.hpl_base + (int64_t)i * 18);
  // adjacency of the whole batch (graphs are disconnected components; ordered per graph)
  std::vector<std::vector<int>> adj(nrow);
  for (auto& pr : b.ppoff) { adj[pr.first].push_back(pr.second); adj[pr.second].push_back(pr.first); }
  for (auto& pr : b.plblk) { adj[pr.first].push_back(nPr + pr.second); adj[nPr + pr.second].push_back(pr.first); }

  std::vector<int> col_row;            // column -> internal row
  std::vector<int> row_col(nrow, -1);  // internal row -> column
  std::vector<std::vector<int>> cstruct_rows;  // per column: rows (internal ids) of off-diagonal blocks
  std::vector<int> col_graph;
  for (int g = 0; g < V.B; ++g) {
    const GraphSeg& sg = b.seg[g];
    const int n = sg.nprow + sg.nlrow;
    auto loc2row = [&](int v) { return v < sg.nprow ? sg.prow0 + v : nPr + sg.lrow0 + (v - sg.nprow); };
    auto row2loc = [&](int r) { return r < nPr ? r - sg.prow0 : sg.nprow + (r - nPr - sg.lrow0); };
    std::vector<std::vector<int>> ladj(n);
    for (int v = 0; v < n; ++v) {
      const int r = loc2row(v);
      ladj[v].reserve(adj[r].size());
      for (int w : adj[r]) ladj[v].push_back(row2loc(w));
    }
    GraphSym S;
    min_degree(n, ladj, S);
    const int c0 = (int)col_row.size();
    for (int s = 0; s < n; ++s) { const int r = loc2row(S.order[s]); row_col[r] = c0 + s; col_row.push_back(r); col_graph.push_back(g); }
    for (int s = 0; s < n; ++s) {
      std::vector<int> rows;
      for (int w : S.cstruct[S.order[s]]) rows.push_back(loc2row(w));
      cstruct_rows.push_back(std::move(rows));
    }
  }
  const int ncol = (int)col_row.size();
  // column block lists sorted by elimination position of the row
  std::vector<int> bp(ncol + 1, 0), boff, brow, bsrc, col_xoff(ncol), col_dim(ncol);
  std::vector<unsigned char> bfmt;
  std::vector<std::unordered_map<int, int>> colblk(ncol);  // per column: row column-id -> block id
  int64_t lnz = 0;
  int max_entries = 0;
  for (int j = 0; j < ncol; ++j) {
    const int rj = col_row[j];
    const int dj = row_dim(rj);
    col_xoff[j] = row_xoff(rj); col_dim[j] = dj;
    std::vector<int> rows_c;
    for (int r : cstruct_rows[j]) rows_c.push_back(row_col[r]);
    std::sort(rows_c.begin(), rows_c.end());
    bp[j] = (int)boff.size();
    // diagonal
    boff.push_back((int)lnz); brow.push_back(j);
    bsrc.push_back(rj < nPr ? rj * 36 : (int)(b.hll_base + (int64_t)(rj - nPr) * 9)); bfmt.push_back(0);
    colblk[j][j] = bp[j];
    int entries = dj * dj;
    lnz += dj * dj;
    for (int i : rows_c) {
      const int ri = col_row[i];
      const int di = row_dim(ri);
      colblk[j][i] = (int)boff.size();
      boff.push_back((int)lnz); brow.push_back(i);
      const int a = std::min(ri, rj), c = std::max(ri, rj);
      auto it = hoff.find(key(a, c));
      if (it == hoff.end()) { bsrc.push_back(-1); bfmt.push_back(0); }
      else { bsrc.push_back(it->second); bfmt.push_back(ri == a ? 0 : 1); }  // stored [min][max]; we need [i][j]
      lnz += di * dj;
      entries += di * dj;
    }
    max_entries = std::max(max_entries, entries);
    if (lnz >= ((int64_t)1 << 31)) { return set_error(SSLAM_ERR_INVALID, "Cholesky factor too large for int32 offsets"); }
  }
  bp[ncol] = (int)boff.size();
  const int nblk = (int)boff.size();
  // update lists: column k updates every (i, j) pair of its structure with pos(j) <= pos(i)
  std::vector<std::vector<std::array<int, 3>>> ulist(nblk);
  for (int k = 0; k < ncol; ++k) {
    const int k0 = bp[k] + 1, k1 = bp[k + 1];
    for (int p = k0; p < k1; ++p) {      // j = brow[p]
      const int j = brow[p];
      for (int q = p; q < k1; ++q) {     // i = brow[q], pos(i) >= pos(j)
        const int i = brow[q];
        auto it = colblk[j].find(i);
        if (it == colblk[j].end()) return set_error(SSLAM_ERR_NUMERIC, "symbolic factorisation inconsistent (missing fill block)");
        ulist[it->second].push_back({boff[q], boff[p], k});
      }
    }
  }
  std::vector<int> up(nblk + 1, 0), ua, ub, uk, ux;
  std::vector<unsigned char> udk;
  for (int t = 0; t < nblk; ++t) {
    for (auto& u : ulist[t]) {  // already ascending in k
      ua.push_back(u[0]); ub.push_back(u[1]); uk.push_back(u[2]);
      ux.push_back(col_xoff[u[2]]); udk.push_back((unsigned char)col_dim[u[2]]);
    }
    up[t + 1] = (int)ua.size();
  }
  // levels of the block elimination tree
  std::vector<int> level(ncol, 0);
  int nlev = 0;
  for (int j = 0; j < ncol; ++j) {
    if (bp[j + 1] - bp[j] > 1) { const int par = brow[bp[j] + 1]; level[par] = std::max(level[par], level[j] + 1); }
    nlev = std::max(nlev, level[j] + 1);
  }
  P->lvl_maxEt.assign(nlev, 0);
  for (int j = 0; j < ncol; ++j) {
    const int last = bp[j + 1] - 1;
    const int cs = boff[last] - boff[bp[j]] + col_dim[brow[last]] * col_dim[j];
    P->lvl_maxEt[level[j]] = std::max(P->lvl_maxEt[level[j]], cs + col_dim[j]);
  }
  P->lvl_maxlist.assign(nlev, 0);
  for (int j = 0; j < ncol; ++j)
    for (int t = bp[j]; t < bp[j + 1]; ++t) P->lvl_maxlist[level[j]] = std::max(P->lvl_maxlist[level[j]], up[t + 1] - up[t]);
  P->lvl_ptr.assign(nlev + 1, 0);
  for (int j = 0; j < ncol; ++j) P->lvl_ptr[level[j] + 1]++;
  for (int l = 0; l < nlev; ++l) P->lvl_ptr[l + 1] += P->lvl_ptr[l];
  std::vector<int> lvl_cols(ncol), cursor(P->lvl_ptr.begin(), P->lvl_ptr.end() - 1);
  for (int j = 0; j < ncol; ++j) lvl_cols[cursor[level[j]]++] = j;

  CholView& C = P->C;
  C.ncol = ncol; C.nlevels = nlev; C.dim = 6 * nPr + 3 * nLr;
  P->max_col_entries = max_entries; P->lnz = lnz;
  int rc;
  if ((rc = up_to_dev(*P, b.stream, col_xoff, &C.col_xoff))) return rc;
  if ((rc = up_to_dev(*P, b.stream, col_dim, &C.col_dim))) return rc;
  if ((rc = up_to_dev(*P, b.stream, col_graph, &C.col_graph))) return rc;
  if ((rc = up_to_dev(*P, b.stream, bp, &C.bp))) return rc;
  if ((rc = up_to_dev(*P, b.stream, boff, &C.boff))) return rc;
  if ((rc = up_to_dev(*P, b.stream, brow, &C.brow))) return rc;
  if ((rc = up_to_dev(*P, b.stream, bsrc, &C.bsrc))) return rc;
  if ((rc = up_to_dev(*P, b.stream, bfmt, &C.bfmt))) return rc;
  if ((rc = up_to_dev(*P, b.stream, up, &C.up))) return rc;
  if ((rc = up_to_dev(*P, b.stream, ua, &C.ua))) return rc;
  if ((rc = up_to_dev(*P, b.stream, ub, &C.ub))) return rc;
  if ((rc = up_to_dev(*P, b.stream, uk, &C.uk))) return rc;
  if ((rc = up_to_dev(*P, b.stream, ux, &C.ux))) return rc;
  if ((rc = up_to_dev(*P, b.stream, udk, &C.udk))) return rc;
  if ((rc = up_to_dev(*P, b.stream, lvl_cols, &C.lvl_cols))) return rc;
  void* p = nullptr;
  SSLAM_HIP_TRY(hipMalloc(&p, std::max<int64_t>(lnz, 1) * sizeof(double))); P->allocs.push_back(p); C.Lval = (double*)p;
  SSLAM_HIP_TRY(hipMalloc(&p, std::max(C.dim, 1) * sizeof(double))); P->allocs.push_back(p); C.y = (double*)p;
  SSLAM_HIP_TRY(hipMalloc(&p, std::max(V.B, 1) * sizeof(int))); P->allocs.push_back(p); C.fail = (int*)p;
  SSLAM_HIP_TRY(hipMemsetAsync(C.fail, 0, std::max(V.B, 1) * sizeof(int), b.stream));
  SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  return 0;
}

int64_t chol_plan_lnz(const Batch& b) { return b.chol ? b.chol->lnz : 0; }
int chol_plan_levels(const Batch& b) { return b.chol ? b.chol->C.nlevels : 0; }

int chol_factor_and_forward(Batch& b) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  ScopedTimer t(b, "factor");
  hipLaunchKernelGGL(k_chol_begin, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  // LDS per level: Et entries + partial sums (nslice*Et <= kPartDoubles, nslice <= NW) + NW wave scratch areas
  auto lds_for = [&](int l, int nw) {
    const int et = P.lvl_maxEt[l];
    return (size_t)(et + std::max(et, std::min(kPartDoubles, nw * et)) + nw * 80) * sizeof(double);
  };
  size_t lds_max = 0;
  for (int l = 0; l < C.nlevels; ++l) lds_max = std::max(lds_max, lds_for(l, 16));
  if (lds_max > 160 * 1024) return set_error(SSLAM_ERR_UNSUPPORTED, "a factor column needs %zu B of LDS (> 160 KiB)", lds_max);
  if (lds_max > 64 * 1024) {
    SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_chol_level<64>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_chol_level<256>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
    SSLAM_HIP_TRY(hipFuncSetAttribute((const void*)k_chol_level<1024>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
  }
  for (int l = 0; l < C.nlevels; ++l) {
    const int n = P.lvl_ptr[l + 1] - P.lvl_ptr[l];
    if (n <= 0) continue;
    // narrow levels are latency-bound (long update lists -> 16 waves per column); wide levels have enough
    // columns in flight to hide latency and run 4 waves per column
    if (P.lvl_maxlist[l] >= 32 && n < 1024) hipLaunchKernelGGL(k_chol_level<1024>, dim3(n), dim3(1024), lds_for(l, 16), b.stream, b.V, C, P.lvl_ptr[l]);
    else if ((P.lvl_maxlist[l] <= 6 && n >= 2048) || n >= 24576) hipLaunchKernelGGL(k_chol_level<64>, dim3(n), dim3(64), lds_for(l, 1), b.stream, b.V, C, P.lvl_ptr[l]);
    else hipLaunchKernelGGL(k_chol_level<256>, dim3(n), dim3(256), lds_for(l, 4), b.stream, b.V, C, P.lvl_ptr[l]);
  }
  hipLaunchKernelGGL(k_chol_end, dim3((b.V.B + 63) / 64), dim3(64), 0, b.stream, b.V, C);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "cholesky factor launch: %s", hipGetErrorString(e));
  return 0;
}

int chol_backward(Batch& b) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  ScopedTimer t(b, "solve");
  for (int l = C.nlevels - 1; l >= 0; --l) {
    const int n = P.lvl_ptr[l + 1] - P.lvl_ptr[l];
    if (n <= 0) continue;
    hipLaunchKernelGGL(k_chol_backward_level, dim3(n, 1), dim3(64), 0, b.stream, C, P.lvl_ptr[l], C.y, b.V.x, b.V.lm);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(SSLAM_ERR_HIP, "cholesky solve launch: %s", hipGetErrorString(e));
  return 0;
}

// X = (L L^T)^-1 RHS for nrhs right-hand sides (host arrays, internal ordering, [nrhs][dim])
int chol_solve_multi(Batch& b, const double* rhs_host, int nrhs, double* x_host) {
  CholPlan& P = *b.chol;
  const CholView& C = P.C;
  if (nrhs <= 0) return 0;
  const int chunk_max = std::max(1, std::min(nrhs, (int)std::min<int64_t>(4096, ((int64_t)1 << 30) / std::max(1, C.dim) / 8)));
  if (P.multi_cap < chunk_max) {
    if (P.d_multi_y) (void)hipFree(P.d_multi_y);
    if (P.d_multi_x) (void)hipFree(P.d_multi_x);
    SSLAM_HIP_TRY(hipMalloc((void**)&P.d_multi_y, (size_t)chunk_max * C.dim * sizeof(double)));
    SSLAM_HIP_TRY(hipMalloc((void**)&P.d_multi_x, (size_t)chunk_max * C.dim * sizeof(double)));
    P.multi_cap = chunk_max;
  }
  for (int r0 = 0; r0 < nrhs; r0 += chunk_max) {
    const int nr = std::min(chunk_max, nrhs - r0);
    const size_t bytes = (size_t)nr * C.dim * sizeof(double);
    SSLAM_HIP_TRY(hipMemcpyAsync(P.d_multi_x, rhs_host + (size_t)r0 * C.dim, bytes, hipMemcpyHostToDevice, b.stream));
    for (int l = 0; l < C.nlevels; ++l) {
      const int n = P.lvl_ptr[l + 1] - P.lvl_ptr[l];
      if (n > 0) hipLaunchKernelGGL(k_chol_forward_level, dim3(n, nr), dim3(64), 0, b.stream, C, P.lvl_ptr[l], (const double*)P.d_multi_x, P.d_multi_y);
    }
    for (int l = C.nlevels - 1; l >= 0; --l) {
      const int n = P.lvl_ptr[l + 1] - P.lvl_ptr[l];
      if (n > 0) hipLaunchKernelGGL(k_chol_backward_level, dim3(n, nr), dim3(64), 0, b.stream, C, P.lvl_ptr[l], (const double*)P.d_multi_y, P.d_multi_x, (const LmState*)nullptr);
    }
    SSLAM_HIP_TRY(hipMemcpyAsync(x_host + (size_t)r0 * C.dim, P.d_multi_x, bytes, hipMemcpyDeviceToHost, b.stream));
    SSLAM_HIP_TRY(hipStreamSynchronize(b.stream));
  }
  return 0;
}

}  // namespace sslam
