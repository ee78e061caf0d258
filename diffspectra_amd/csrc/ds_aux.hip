// diffspectra_amd auxiliary kernels that are not part of the denoising arithmetic: the weight fingerprint that tells the
// drop-in DMT.forward whether its packed weights are stale.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/diffspectra_hip.h"

namespace {

// 64-bit position-sensitive checksum over a list of fp32 tensors: sum over elements of bits(x) * odd(hash(global index)) + index,
// in wrap-around integer arithmetic (associative, so the atomic accumulation order does not matter: deterministic).  A sign
// flip, a row / head permutation or a copy from an equal-norm tensor all change it (the norm pair did not).
__global__ __launch_bounds__(256) void k_fingerprint(const float* const* __restrict__ ptrs, const int64_t* __restrict__ prefix, int n,
                                                      unsigned long long* __restrict__ out) {
  const int64_t total = prefix[n];
  const int64_t chunk = 256 * 16;
  unsigned long long acc = 0;
  for (int64_t base = (int64_t)blockIdx.x * chunk; base < total; base += (int64_t)gridDim.x * chunk) {
    int lo = 0, hi = n;                                  // tensor of the chunk's first element (block-uniform search)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (prefix[mid] <= base) lo = mid; else hi = mid;
    }
    int t = lo;
    for (int j = 0; j < 16; ++j) {
      const int64_t g = base + j * 256 + threadIdx.x;
      if (g >= total) break;
      while (g >= prefix[t + 1]) ++t;
      const unsigned int v = __float_as_uint(ptrs[t][g - prefix[t]]);
      const unsigned long long m = ((unsigned long long)g * 0x9E3779B97F4A7C15ull) | 1ull;
      acc += (unsigned long long)v * m + (unsigned long long)g;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned int lo32 = __shfl_down((unsigned int)acc, off), hi32 = __shfl_down((unsigned int)(acc >> 32), off);
    acc += ((unsigned long long)hi32 << 32) | lo32;
  }
  if ((threadIdx.x & 63) == 0) atomicAdd(out, acc);
}

}  // namespace

extern "C" int ds_fingerprint(const float* const* ptrs, const int64_t* prefix, int32_t n, uint64_t* out, void* stream) {
  if (!ptrs || !prefix || !out || n <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync(out, 0, sizeof(uint64_t), s) != hipSuccess) return DS_ERR_LAUNCH;
  hipLaunchKernelGGL(k_fingerprint, dim3(1024), dim3(256), 0, s, ptrs, prefix, (int)n, reinterpret_cast<unsigned long long*>(out));
  return hipGetLastError() == hipSuccess ? DS_OK : DS_ERR_LAUNCH;
}
