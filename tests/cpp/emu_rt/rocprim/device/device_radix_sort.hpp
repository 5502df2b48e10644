// Test infrastructure (see ../../simt_state.h): what the engine calls of rocPRIM's radix sort, by std::stable_sort.
#pragma once
#include <hip/hip_runtime.h>
namespace rocprim {
template <class K, class V>
hipError_t radix_sort_pairs(void* tmp, size_t& bytes, const K* kin, K* kout, const V* vin, V* vout, size_t n, unsigned b0, unsigned b1, hipStream_t) {
  if (!tmp) { bytes = 256; return hipSuccess; }
  std::vector<size_t> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = i;
  const K mask = b1 - b0 >= sizeof(K) * 8 ? ~(K)0 : (((K)1 << (b1 - b0)) - 1);
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return ((kin[a] >> b0) & mask) < ((kin[b] >> b0) & mask); });
  for (size_t i = 0; i < n; ++i) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
  return hipSuccess;
}
}  // namespace rocprim
