// Shared error plumbing for the C-ABI (thread-local last-error string, HIP status checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <string>

namespace sslam {
std::string& last_error_ref();
inline int set_error(int code, const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error_ref() = buf;
  return code;
}
// Largest dynamic-LDS opt-in a kernel may ask for on `device`: the device's own limit (hipDeviceProp_t::sharedMemPerBlockOptin /
// maxSharedMemoryPerBlock) minus `reserve` bytes for the kernels' static LDS, never more than `want` (the gfx950 figure the plans were
// sized for).  A part with less LDS than an MI355X then fails with "needs N B of LDS" instead of a failed hipFuncSetAttribute.
inline int lds_optin_limit(int device, int want, int reserve) {
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeSharedMemPerBlockOptin, device) != hipSuccess || v <= 0) {
    v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, device) != hipSuccess) v = 0;
  }
  if (v <= 64 * 1024) return want;   // attribute not reported (or only the 64 KiB no-opt-in figure): keep the gfx950 constant, hipFuncSetAttribute fails loudly if it is too much
  return v - reserve < want ? v - reserve : want;
}
// Page-locked staging for the small host <-> device copies of an optimise call (estimates up / down, the looks at the LM state, error
// flags).  From pageable memory every such copy is a staged blit with its own host round trip: ~38 of them per tick of the orchestrator,
// 13 us apart -- 0.6 ms of a 5.6 ms tick (rocprofv3 trace, round 5).  One buffer, used by one copy sequence at a time: every user
// synchronises its stream before the host touches the bytes and before the next user fills them.  Lives in the graph handle (a batch of
// one is rebuilt every tick) or in the batch itself.
struct PinnedScratch {
  char* p = nullptr;
  size_t cap = 0;
  PinnedScratch() = default;
  PinnedScratch(const PinnedScratch&) = delete;
  PinnedScratch& operator=(const PinnedScratch&) = delete;
  char* get(size_t bytes) {
    if (bytes <= cap) return p;
    if (p) (void)hipHostFree(p);
    p = nullptr; cap = 0;
    void* q = nullptr;
    const size_t want = bytes + bytes / 2 + 4096;
    if (hipHostMalloc(&q, want, hipHostMallocDefault) != hipSuccess) return nullptr;
    p = (char*)q; cap = want;
    return p;
  }
  ~PinnedScratch() { if (p) (void)hipHostFree(p); }
};

}  // namespace sslam

#define SSLAM_HIP_TRY(expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return ::sslam::set_error(_e == hipErrorNoDevice || _e == hipErrorInvalidDevice ? -2 : -3,     \
                                "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)
