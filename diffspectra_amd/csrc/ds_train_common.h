// diffspectra_amd - small device helpers shared by the training sources (ds_train.hip, ds_train_gemm.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dst {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// kind: 1 SiLU, 2 GELU(erf), 3 tanh
__device__ __forceinline__ float act_apply(float v, int kind) {
  if (kind == 1) return v * sigmoidf_(v);
  if (kind == 2) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  return tanhf(v);
}
// derivative factor: ref = pre-activation for SiLU / GELU, ref = tanh output for tanh
__device__ __forceinline__ float act_deriv(float v, int kind) {
  if (kind == 1) { const float s = sigmoidf_(v); return s * (1.0f + v * (1.0f - s)); }
  if (kind == 2) return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * expf(-0.5f * v * v) * 0.39894228040143267794f;
  return 1.0f - v * v;
}

// Philox4x32-10 (Salmon et al., SC'11), the generator of the FF-dropout masks: block q of stream `stream_id` under key `seed`
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned int)p1, n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned int)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__device__ __forceinline__ void dropout_block(unsigned long long seed, unsigned int stream_id, int64_t q, unsigned int (&c)[4]) {
  c[0] = (unsigned int)q; c[1] = (unsigned int)(q >> 32); c[2] = stream_id; c[3] = 0x44524f50u;
  unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}
__device__ __forceinline__ unsigned int dropout_threshold(float p) { return (unsigned int)fminf(p * 4294967296.0f, 4294967040.0f); }
// keep-or-drop of the element with flat index i of a tensor (the mask is a pure function of (seed, stream, i))
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, unsigned int stream_id, int64_t i, unsigned int thr) {
  unsigned int c[4];
  dropout_block(seed, stream_id, i >> 2, c);
  const int j = (int)(i & 3);
  const unsigned int w = j == 0 ? c[0] : j == 1 ? c[1] : j == 2 ? c[2] : c[3];
  return w >= thr;
}

}  // namespace dst
