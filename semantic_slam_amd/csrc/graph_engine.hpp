// Batch engine state shared by the backend translation units (LM driver, sparse Cholesky).
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "graph_host.hpp"
#include "graph_kernels.hpp"
#include "sslam_common.hpp"

namespace sslam {

struct CholPlan;
void chol_plan_free(CholPlan*);
void batch_comm_destroy(void* comm);   // ncclCommDestroy

struct KernelTimer {
  double total_ms = 0;
  int64_t launches = 0;
};

// Device memory of a single-graph handle across structure rebuilds.  The orchestrator's graph grows every tick and every tick rebuilds
// its batch of one (about 65 device arrays with the factorisation plan): a hipMalloc / hipFree pair per array costs more than
// optimising a small graph.  Bump allocation out of one block that is kept from rebuild to rebuild and regrown (x 1.5) when a
// rebuild needed more than it holds; what did not fit meanwhile lives in spill blocks until the next reset.
struct DevArena {
  // One device block, two regions: [0, small_cap) holds the small index / table arrays of a rebuild (each <= kMirrorMax), the rest the
  // large arrays (H, L, update matrices).  Only the table region has a pinned host mirror: the ~65 small arrays of a rebuild are written
  // into it and travel to the device as ONE copy (flush) instead of one hipMemcpyAsync + hipMemsetAsync each -- the uploads were most of
  // the 3 ms a tick spent on its structure rebuild.  (Round 3 mirrored the whole block: tens to hundreds of MB of pinned host memory per
  // graph handle for tables of a few hundred KB -- ADVICE r3.)  Large arrays keep their own memset / copy.
  char* base = nullptr;
  size_t cap = 0, used = 0, need = 0;                     // whole block; the large region is [small_cap, cap), `used` counts from 0
  size_t small_cap = 0, small_used = 0, small_need = 0;   // table region
  std::vector<void*> spill;
  char* mirror = nullptr;
  size_t mirror_cap = 0, flushed = 0;
  bool mirror_off = false;   // the pinned mirror could not be allocated during this rebuild: every table array is written directly until reset()
  static constexpr size_t kMirrorMax = 256 << 10;   // arrays above this size live in the large region
  // direct: the caller writes the array itself (memset / copy on its stream, issued before the flush): never from the table region,
  // whose bytes all come from the mirror
  void* take(size_t bytes, bool direct = false) {
    bytes = (bytes + 255) & ~(size_t)255;
    need += bytes;
    if (bytes <= kMirrorMax && !direct) {
      small_need += bytes;
      if (small_used + bytes <= small_cap) { void* p = base + small_used; small_used += bytes; return p; }
    }
    if (used + bytes <= cap) { void* p = base + used; used += bytes; return p; }
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    spill.push_back(p);
    return p;
  }
  bool in_block(const void* p) const { return base && (const char*)p >= base && (const char*)p < base + cap; }
  // host address that shadows device address p (table region only), or nullptr
  char* shadow(void* p, size_t bytes) {
    if (mirror_off || !base || (const char*)p < base || (const char*)p + bytes > base + small_cap) return nullptr;
    if (mirror_cap < small_cap) {
      if (mirror) (void)hipHostFree(mirror);
      mirror = nullptr; mirror_cap = 0;
      void* q = nullptr;
      // (round-4 ADVICE: once an array of this rebuild has been written directly, a mirror allocated later would flush uninitialised bytes over it)
      if (hipHostMalloc(&q, small_cap, hipHostMallocDefault) != hipSuccess) { mirror_off = true; return nullptr; }
      mirror = (char*)q; mirror_cap = small_cap; flushed = 0;
    }
    return mirror + ((char*)p - base);
  }
  // everything written to the mirror since the last flush -> device, one copy (every byte of the table region is shadowed: no holes)
  int flush(hipStream_t st) {
    if (mirror && !mirror_off && small_used > flushed && hipMemcpyAsync(base + flushed, mirror + flushed, small_used - flushed, hipMemcpyHostToDevice, st) != hipSuccess) return -3;
    flushed = small_used;
    return 0;
  }
  void note_direct(void*, size_t) {}   // large-region arrays are written by their own memset / copy; nothing of the mirror overlaps them
  void reset() {   // the caller guarantees that no kernel still uses the memory (the owning batch has been released)
    for (void* p : spill) (void)hipFree(p);
    spill.clear();
    const size_t big_need = need - small_need;
    size_t new_small = small_cap;
    if (small_need > small_cap) new_small = (small_need + small_need / 2 + (64u << 10) + 255) & ~(size_t)255;
    if (new_small + big_need > cap) {
      if (base) (void)hipFree(base);
      base = nullptr; cap = 0; small_cap = 0;
      const size_t want = new_small + big_need + big_need / 2 + (1u << 20);
      void* p = nullptr;
      if (hipMalloc(&p, want) == hipSuccess) { base = (char*)p; cap = want; }
    }
    small_cap = base ? new_small : 0;
    used = small_cap; small_used = 0; need = 0; small_need = 0; flushed = 0; mirror_off = false;
  }
  ~DevArena() {
    for (void* p : spill) (void)hipFree(p);
    if (base) (void)hipFree(base);
    if (mirror) (void)hipHostFree(mirror);
  }
};

void persist_forget_stream(int device, hipStream_t stream);   // sslam_chol.hip: the per-device chain of persistent launches forgets a stream that is going away

struct Batch {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = true;      // false: the stream belongs to the graph handle and outlives this batch (no create / destroy per structure rebuild)
  std::vector<HostGraph*> graphs;
  std::vector<uint64_t> versions;
  BatchView V{};
  std::vector<void*> allocs;
  DevArena* arena = nullptr;   // set for the batch of one behind a graph handle: device arrays come out of the handle's arena
  double plan_build_ms = 0;    // host time of chol_plan_build since the last sslam_graph_optimize read it (sslam_opt_stats::host_plan_us)
  PinnedScratch own_pin;       // staging of the small copies ...
  PinnedScratch* pin = &own_pin;   // ... or the graph handle's (kept across the per-tick rebuilds of its batch of one)
  // host-side metadata
  std::vector<GraphSeg> seg;
  std::vector<std::vector<int>> v2pose, v2lm;     // per graph: vertex id -> pose / landmark index (global), -1
  std::vector<int> pose_row, lm_row;              // global
  std::vector<int> prow_pose, lrow_lm;
  std::vector<std::pair<int, int>> ppoff;         // unique pose-pose blocks (row a < row b)
  std::vector<std::pair<int, int>> plblk;         // unique pose-landmark blocks (pose row, lm row)
  std::vector<std::pair<int, int>> llblk;         // unique landmark-landmark blocks (lm row a < lm row b): point-point edges
  int64_t hll_off_base = 0;
  std::vector<int> pose_vertex, lm_vertex;        // global pose/lm index -> vertex id in its graph
  int64_t hpp_off_base = 0, hpl_base = 0, hll_base = 0;
  double* d_part_e = nullptr;  // [B*maxEdgeChunks]
  double* d_part_m = nullptr;  // [B*maxRowChunks] (max diag)
  bool profiling = false;
  std::map<std::string, KernelTimer> timers;
  struct Pending { std::string name; hipEvent_t a, b; };
  std::vector<Pending> pending;
  std::vector<hipEvent_t> event_pool;
  bool uploaded = false;
  bool has_duplicate_blocks = false;
  bool has_planes = false;
  std::vector<int> dup_eo, dup_el;
  int max_row_slots = 0;
  CholPlan* chol = nullptr;    // piece plan (chol_plan.hpp): multi right-hand-side solves of the marginals; SSLAM_CHOL_LEGACY=1: the LM loop too
  // edge-sharded mode
  bool sharded = false;        // linearize only the edges of this rank's range
  void* comm = nullptr;        // ncclComm_t (RCCL), or null: partial systems are left unsummed (single-device tests)
  int64_t hb_doubles = 0;      // doubles in the contiguous [H || b] buffer
  double* d_hb_part = nullptr; // edge-sharded mode: this rank's partial [H || b] (send buffer of the out-of-place all-reduce)
  int64_t allreduce_calls = 0; // ncclAllReduce calls issued so far (tests: the collective really ran)
  int shard_rank = 0, shard_world = 1;

  ~Batch() { release(); }
  void release() {
    if (chol) { chol_plan_free(chol); chol = nullptr; }
    if (comm) { batch_comm_destroy(comm); comm = nullptr; }
    if (stream) { hipSetDevice(device); hipStreamSynchronize(stream); }
    for (auto& p : pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    pending.clear();
    for (auto e : event_pool) hipEventDestroy(e);
    event_pool.clear();
    for (void* p : allocs) hipFree(p);
    allocs.clear();
    if (stream && own_stream) { hipStreamSynchronize(stream); persist_forget_stream(device, stream); hipStreamDestroy(stream); }
    stream = nullptr;
  }
  hipEvent_t get_event() {
    if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
    hipEvent_t e; hipEventCreate(&e); return e;
  }
  void harvest() {  // call after a stream sync
    for (auto& p : pending) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { timers[p.name].total_ms += ms; timers[p.name].launches += 1; }
      event_pool.push_back(p.a); event_pool.push_back(p.b);
    }
    pending.clear();
  }
};

struct ScopedTimer {
  Batch& b; const char* name; hipEvent_t a{}, e{}; bool on;
  ScopedTimer(Batch& bb, const char* n) : b(bb), name(n), on(bb.profiling) {
    if (on) { a = b.get_event(); e = b.get_event(); hipEventRecord(a, b.stream); }
  }
  ~ScopedTimer() {
    if (on) { hipEventRecord(e, b.stream); b.pending.push_back({name, a, e}); }
  }
};

template <typename T>
inline int dev_upload(Batch& b, const std::vector<T>& h, T** out, size_t min_elems = 1) {
  const size_t n = std::max(h.size(), min_elems);
  void* p = nullptr;
  if (b.arena) {
    p = b.arena->take(n * sizeof(T));
    if (!p) return set_error(-3, "device allocation of %zu bytes failed", n * sizeof(T));
    if (char* m = b.arena->shadow(p, n * sizeof(T))) {   // travels with the arena's next flush
      if (!h.empty()) memcpy(m, h.data(), h.size() * sizeof(T));
      *out = (T*)p;
      return 0;
    }
    b.arena->note_direct(p, n * sizeof(T));
  } else { SSLAM_HIP_TRY(hipMalloc(&p, n * sizeof(T))); b.allocs.push_back(p); }
  if (!h.empty()) SSLAM_HIP_TRY(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, b.stream));
  *out = (T*)p;
  return 0;
}
template <typename T>
inline int dev_alloc(Batch& b, size_t n, T** out, bool zero = true) {
  void* p = nullptr;
  n = std::max<size_t>(n, 1);
  if (b.arena) {
    p = b.arena->take(n * sizeof(T));
    if (!p) return set_error(-3, "device allocation of %zu bytes failed", n * sizeof(T));
    if (char* m = b.arena->shadow(p, n * sizeof(T))) {
      memset(m, 0, n * sizeof(T));
      *out = (T*)p;
      return 0;
    }
    b.arena->note_direct(p, n * sizeof(T));
  } else { SSLAM_HIP_TRY(hipMalloc(&p, n * sizeof(T))); b.allocs.push_back(p); }
  if (zero) SSLAM_HIP_TRY(hipMemsetAsync(p, 0, n * sizeof(T), b.stream));
  *out = (T*)p;
  return 0;
}


// sparse block Cholesky (sslam_chol.hip; symbolic phase in chol_plan.hpp)
struct SymIn;
void chol_sym_input(const Batch& b, SymIn& in);
int chol_plan_build(Batch& b);
int chol_plan_launches(const Batch& b);
int chol_factor_and_forward(Batch& b, bool flat = false);   // (H + lambda I) = L L^T for in_trial graphs, y = L^-1 b
int chol_backward(Batch& b);             // x = L^-T y  -> V.x
int chol_set_active(Batch& b, const std::vector<char>* active);   // LM endgame: size the launches for the graphs still active (nullptr: all)
int64_t chol_plan_lnz(const Batch& b);
int chol_plan_levels(const Batch& b);
int chol_solve_multi(Batch& b, const double* rhs_host, int nrhs, double* x_host);  // uses the last factorisation
bool chol_plan_flow(const Batch& b);        // the plan runs factor + both solves in one dependency-driven launch (small batches)
int chol_solve_flow(Batch& b);              // (H + lambda I) dx = b for in_trial graphs -> V.x, one launch
int chol_lm_step_flow(Batch& b, int max_iters);   // begin step + one-launch solve + update / chi2 / accept-reject / commit: 3 launches per damping trial
bool chol_plan_spec(const Batch& b);        // one small graph: the damping trials of an LM iteration can run side by side
int chol_spec_mode(const Batch& b);               // 0 off, 1 adaptive (lanes join after the first rejected trial of an iteration), 2 always
int chol_lm_step_spec(Batch& b, int max_iters);   // one LM iteration: up to ten speculative trials + the accept / reject replay
int chol_factor_flat_flow(Batch& b);        // flat factor (marginals) through the single launch
int chol_flow_check(Batch& b);              // error flag of that launch (synchronises the stream)
int chol_marginal_diag(Batch& b, const std::vector<int>& xoff, const std::vector<int>& dims, double* out36);  // diagonal blocks of H^-1 along the tree paths, one launch


}  // namespace sslam
