// Test infrastructure (see ../../simt_state.h): what the engine calls of rocPRIM's scans, as loops.
#pragma once
#include <hip/hip_runtime.h>
namespace rocprim {
template <class T> struct plus { T operator()(const T& a, const T& b) const { return a + b; } };
template <class T> struct maximum { T operator()(const T& a, const T& b) const { return a < b ? b : a; } };
template <class T, class Op>
hipError_t inclusive_scan(void* tmp, size_t& bytes, const T* in, T* out, size_t n, Op op, hipStream_t) {
  if (!tmp) { bytes = 256; return hipSuccess; }
  T acc{};
  for (size_t i = 0; i < n; ++i) { acc = i ? op(acc, in[i]) : in[i]; out[i] = acc; }
  return hipSuccess;
}
}  // namespace rocprim
