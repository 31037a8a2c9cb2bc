// csrc/common.h -- shared helpers of libdsr_hip.so (error convention, HIP checks, device buffers).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include <stdexcept>
#include "../../include/dsr.h"

namespace dsr {

// Mirrors j_error + error_type (btk/common/jexception.h:41-70): thrown inside the library,
// translated to a dsr_status at the C-ABI.
struct Error : std::exception {
  int code; std::string msg;
  Error(int c, const char* fmt, ...) : code(c) {
    char buf[1024]; va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap); msg = buf;
  }
  const char* what() const noexcept override { return msg.c_str(); }
};

void set_last_error(const std::string& s);

#define DSR_HIP(expr) do { hipError_t e__ = (expr); if (e__ != hipSuccess) \
  throw dsr::Error(DSR_E_INITIALIZATION, "HIP error %s at %s:%d (%s)", hipGetErrorString(e__), __FILE__, __LINE__, #expr); } while (0)

// C-ABI guard: run body, map exceptions to status codes.
template <class F> static inline dsr_status guard(F&& f) {
  try { f(); return DSR_OK; }
  catch (const Error& e) { set_last_error(e.msg); return e.code; }
  catch (const std::bad_alloc&) { set_last_error("out of host memory"); return DSR_E_ALLOCATION; }
  catch (const std::exception& e) { set_last_error(e.what()); return DSR_E_ERROR; }
}

void require_device();   // throws DSR_E_INITIALIZATION when no HIP device is usable

// Owning device buffer (grows, never shrinks).  Growing frees the old block: a kernel enqueued on ANY stream may still be reading it, so the
// device is drained first (growth happens on the first call of a shape, never in a steady state).
template <class T> struct DevBuf {
  T* p = nullptr; size_t n = 0;
  DevBuf() = default; DevBuf(const DevBuf&) = delete; DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void) hipFree(p); }
  void reserve(size_t m) {
    if (m <= n) return;
    if (p) { DSR_HIP(hipDeviceSynchronize()); DSR_HIP(hipFree(p)); p = nullptr; n = 0; }
    hipError_t e = hipMalloc((void**) &p, m * sizeof(T));
    if (e != hipSuccess) { p = nullptr; throw Error(DSR_E_ALLOCATION, "hipMalloc of %zu bytes failed: %s", m * sizeof(T), hipGetErrorString(e)); }
    n = m;
  }
  void upload(const T* h, size_t m, hipStream_t s = nullptr) {
    reserve(m);
    if (m) DSR_HIP(hipMemcpyAsync(p, h, m * sizeof(T), hipMemcpyHostToDevice, s));
    if (m) DSR_HIP(hipStreamSynchronize(s));
  }
  void upload(const std::vector<T>& v, hipStream_t s = nullptr) { upload(v.data(), v.size(), s); }
};

// Scratch that a kernel hands to a later kernel of the same stream (work lists, flags, staged weights): one instance per stream, so launches
// enqueued on different streams -- two pipes in flight -- never share it, and within a stream the launches are ordered.  Never `static`.
template <class T> struct PerStream {
  std::mutex mu; std::map<hipStream_t, std::unique_ptr<T>> m;
  T& at(hipStream_t s) { std::lock_guard<std::mutex> g(mu); std::unique_ptr<T>& p = m[s]; if (!p) p.reset(new T()); return *p; }
};

// pinned host memory (the target of asynchronous device-to-host copies)
template <class T> struct PinBuf {
  T* p = nullptr; size_t n = 0;
  PinBuf() = default; PinBuf(const PinBuf&) = delete; PinBuf& operator=(const PinBuf&) = delete;
  ~PinBuf() { if (p) (void) hipHostFree(p); }
  void reserve(size_t m) {
    if (m <= n) return;
    if (p) { DSR_HIP(hipHostFree(p)); p = nullptr; n = 0; }
    hipError_t e = hipHostMalloc((void**) &p, m * sizeof(T), hipHostMallocDefault);
    if (e != hipSuccess) { p = nullptr; throw Error(DSR_E_ALLOCATION, "hipHostMalloc of %zu bytes failed: %s", m * sizeof(T), hipGetErrorString(e)); }
    n = m;
  }
};

static inline int ilog2(unsigned v) { int l = 0; while ((1u << l) < v) l++; return l; }
static inline bool is_pow2(unsigned v) { return v && !(v & (v - 1)); }
static inline int cdiv(long a, long b) { return (int) ((a + b - 1) / b); }

}  // namespace dsr
