// Micro-benchmark: does the fp32 MFMA SHAPE change how much VALU work can issue beside a saturated matrix pipe?
// NMW MFMA waves (NMW/4 per SIMD) run back-to-back v_mfma_f32_32x32x2_f32 (SHAPE 0, 64 cycles, 2048 MACs) or
// v_mfma_f32_16x16x4_f32 (SHAPE 1, 32 cycles, 1024 MACs) on independent accumulators; four more waves (one per SIMD) run a
// VALU fma stream (MODE 1) or exit (MODE 0).  Reports cycles per 2048 MACs for the matrix waves and cycles per VALU op.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int MODE, int PRIO, int NMW>
__global__ __launch_bounds__(NMW * 64 + 256) void k(float* out, unsigned long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (wave < NMW) {
    float y = 1.0f + lane * 0.002f, r = 0;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (SHAPE == 0) {
      f32x16 a0 = {0}, a1 = {0};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a1, 0, 0, 0);
        }
      }
      f32x16 s = a0 + a1;
      for (int i = 0; i < 16; ++i) r += s[i];
    } else {
      f32x4 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
      for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // 16 x 1024 MACs = the same work as the 8 x 2048 of SHAPE 0
          a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a1, 0, 0, 0);
          a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a2, 0, 0, 0);
          a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
        }
      }
      f32x4 s = a0 + a1 + a2 + a3;
      for (int i = 0; i < 4; ++i) r += s[i];
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  } else {
    if (MODE == 0) return;
    __builtin_amdgcn_s_setprio(PRIO);
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 24; ++j) {
        v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0002f, 0.25f); v2 = fmaf(v2, 0.9999f, 0.125f); v3 = fmaf(v3, 0.9998f, 1.0f);
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = v0 + v1 + v2 + v3;
    if (threadIdx.x == NMW * 64 && blockIdx.x == 0) cyc[1] = t1 - t0;
  }
}

template <int SHAPE, int MODE, int PRIO, int NMW>
void run(const char* name) {
  float* out; unsigned long long* cyc; unsigned long long h[2] = {0, 0};
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
  const int iters = 4096;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<SHAPE, MODE, PRIO, NMW>), dim3(256), dim3(NMW * 64 + 256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-10s %-22s mfma waves/SIMD %d: %.1f cycles per 2048 MACs per wave; co-runner %.1f cycles per VALU op\n",
         SHAPE ? "16x16x4" : "32x32x2", name, NMW / 4, (double)h[0] / (iters * 8.0), (double)h[1] / (iters * 96.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0, 0, 4>("alone");
  run<1, 0, 0, 4>("alone");
  run<0, 1, 0, 4>("+VALU prio 0");
  run<1, 1, 0, 4>("+VALU prio 0");
  run<0, 1, 3, 4>("+VALU prio 3");
  run<1, 1, 3, 4>("+VALU prio 3");
  run<0, 0, 0, 8>("alone");
  run<1, 0, 0, 8>("alone");
  run<0, 1, 0, 8>("+VALU prio 0");
  run<1, 1, 0, 8>("+VALU prio 0");
  run<0, 1, 3, 8>("+VALU prio 3");
  run<1, 1, 3, 8>("+VALU prio 3");
  return 0;
}
