// Micro-benchmark: issue cost of v_mfma_f32_32x32x2_f32 under different accumulator dependency patterns.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_issue.hip -o gpurun_out/mfma_issue ; development tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int PATTERN>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  float x = threadIdx.x * 0.001f, y = 1.0f + threadIdx.x * 0.002f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (PATTERN == 0) {   // same accumulator, 8 in a row
#pragma unroll
      for (int j = 0; j < 8; ++j) a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
    } else if (PATTERN == 1) {   // strict alternation of two
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      }
    } else if (PATTERN == 2) {   // the order the compiler picked in wave_mma<2>: 0 0 1 0 1 0 1 1
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
    } else if (PATTERN == 3) {   // four accumulators round robin
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  f32x16 s = a0 + a1 + a2 + a3;
  float r = 0;
  for (int i = 0; i < 16; ++i) r += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int P>
void run(const char* name, int waves_per_simd) {
  float* out; unsigned long long* cyc; unsigned long long h;
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 8);
  const int iters = 4096;
  // 256 threads = 4 waves = one wave per SIMD; launch waves_per_simd workgroups per CU
  hipLaunchKernelGGL(k<P>, dim3(256 * waves_per_simd), dim3(256), 0, 0, out, cyc, iters);
  hipLaunchKernelGGL(k<P>, dim3(256 * waves_per_simd), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-28s waves/SIMD %d: %.1f cycles per MFMA per wave -> %.1f cycles per MFMA per SIMD\n", name, waves_per_simd,
         (double)h / (iters * 8.0), (double)h / (iters * 8.0) / waves_per_simd);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w = 1; w <= 2; ++w) {
    run<0>("same acc x8", w);
    run<1>("alternate 2 accs", w);
    run<2>("0 0 1 0 1 0 1 1", w);
    run<3>("round robin 4 accs", w);
  }
  return 0;
}
