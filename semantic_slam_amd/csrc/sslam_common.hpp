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
}  // namespace sslam

#define SSLAM_HIP_TRY(expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess)                                                                            \
      return ::sslam::set_error(_e == hipErrorNoDevice || _e == hipErrorInvalidDevice ? -2 : -3,     \
                                "%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
  } while (0)
