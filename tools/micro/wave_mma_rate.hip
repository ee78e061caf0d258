// Micro-benchmark: MFMA issue efficiency of ds_device.h's wave_mma (LDS A fragments, L2-streamed packed weights) in
// isolation: cycles per v_mfma_f32_32x32x2_f32 per SIMD (64 = the pipe's rate).  Development tool only.
#include "../../diffspectra_amd/csrc/ds_device.h"
#include <cstdio>
#include <vector>

template <int MT, bool TRANS, int NW>
__global__ __launch_bounds__(NW * 64) void k(const float* Wp, float* out, unsigned long long* cyc, int iters) {
  constexpr int T = 32 * MT, LD = 256 + DS_LDP;
  __shared__ __attribute__((aligned(16))) float X[T][LD];
  for (int i = threadIdx.x; i < T * LD; i += NW * 64) (&X[0][0])[i] = (i % 97) * 1e-3f;
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  f32x16 acc[MT];
  acc_zero<MT>(acc);
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    asm volatile("" ::: "memory");
    wave_mma<MT, TRANS>(&X[0][0], LD, Wp, 256, ((wave + it) & 7) * 32, 0, 32, acc);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int m = 0; m < MT; ++m)
    for (int i = 0; i < 16; ++i) r += acc[m][i];
  out[blockIdx.x * NW * 64 + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int MT, bool TRANS, int NW>
void run(const float* Wp, int wg_per_cu) {
  float* out; unsigned long long* cyc; unsigned long long h = 0;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  const int iters = 64;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MT, TRANS, NW>), dim3(256 * wg_per_cu), dim3(NW * 64), 0, 0, Wp, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  const double mfma_per_wave = (double)iters * 128 * MT, waves_per_simd = NW / 4.0 * wg_per_cu;
  printf("wave_mma<MT=%d,%s> %d waves/WG x %d WG/CU (%.0f waves/SIMD): %.1f cycles per MFMA per SIMD\n", MT, TRANS ? "T" : "N", NW,
         wg_per_cu, waves_per_simd, (double)h / mfma_per_wave / waves_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  float* Wp;
  hipMalloc(&Wp, 256 * 256 * 4);
  std::vector<float> h(256 * 256, 0.001f);
  hipMemcpy(Wp, h.data(), 256 * 256 * 4, hipMemcpyHostToDevice);
  run<1, false, 4>(Wp, 1); run<1, false, 4>(Wp, 2); run<1, false, 8>(Wp, 1);
  run<2, false, 4>(Wp, 1); run<2, false, 4>(Wp, 2); run<2, false, 8>(Wp, 1);
  run<2, true, 4>(Wp, 1);  run<2, true, 8>(Wp, 1);
  run<4, false, 4>(Wp, 1); run<4, false, 8>(Wp, 1);
  return 0;
}
