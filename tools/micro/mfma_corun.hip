// Micro-benchmark: how much a co-resident VALU / LDS-writing wave slows a wave that issues back-to-back fp32 MFMAs.
// 512 threads: waves 0-3 run the MFMA stream (one per SIMD), waves 4-7 run MODE: 0 idle (exit), 1 VALU fma chain,
// 2 VALU + LDS writes, 3 transcendental (exp) chain; PRIO is s_setprio of the co-runner.  FEED: 0 register operands,
// 1 operands re-read from LDS (ds_read_b128 per 4 MFMAs per accumulator, like wave_mma<2,true>).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int PRIO, int FEED, int NMW>
__global__ __launch_bounds__(NMW * 64 + 256) void k(float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float X[64][260];
  __shared__ float sink[4][64 * 4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 260; i += NMW * 64 + 256) (&X[0][0])[i] = i * 1e-4f;
  __syncthreads();
  if (wave < NMW) {
    f32x16 a0 = {0}, a1 = {0};
    float y = 1.0f + lane * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      if (FEED == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a0, 0, 0, 0);
          a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a1, 0, 0, 0);
        }
      } else {
        const int kg = i & 31;
        const float4 b0 = *reinterpret_cast<const float4*>(&X[lane & 31][kg * 8 + 4 * (lane >> 5)]);
        const float4 b1 = *reinterpret_cast<const float4*>(&X[32 + (lane & 31)][kg * 8 + 4 * (lane >> 5)]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.x, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.z, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.z, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.w, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.w, a1, 0, 0, 0);
      }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x16 s = a0 + a1;
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
  } else {
    if (MODE == 0) return;
    __builtin_amdgcn_s_setprio(PRIO);
    float v0 = lane, v1 = lane + 1, v2 = lane + 2, v3 = lane + 3;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // run about as long as the MFMA waves: 8 MFMAs = 512 cycles per iteration = 128 VALU slots
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int j = 0; j < 24; ++j) {
        if (MODE == 4) {   // dependent DPP chain (wave reduction shape)
          v0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0), 0xB1, 0xF, 0xF, true));
          v0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0), 0x4E, 0xF, 0xF, true));
          v0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0), 0x141, 0xF, 0xF, true));
          v0 += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v0), 0x140, 0xF, 0xF, true));
        } else if (MODE == 3) { v0 = __expf(v0 * 0.5f); v1 = __expf(v1 * 0.5f); v2 = __expf(v2 * 0.25f); v3 = __expf(v3 * 0.125f); }
        else { v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0002f, 0.25f); v2 = fmaf(v2, 0.9999f, 0.125f); v3 = fmaf(v3, 0.9998f, 1.0f); }
      }
      if (MODE == 2) *reinterpret_cast<float4*>(&sink[wave - NMW][lane * 4]) = make_float4(v0, v1, v2, v3);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = v0 + v1 + v2 + v3;
    if (threadIdx.x == NMW * 64 && blockIdx.x == 0) cyc[1] = t1 - t0;
  }
}

template <int MODE, int PRIO, int FEED, int NMW = 4>
void run(const char* name) {
  float* out; unsigned long long* cyc; unsigned long long h[2] = {0, 0};
  hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
  const int iters = 4096;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<MODE, PRIO, FEED, NMW>), dim3(256), dim3(NMW * 64 + 256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-44s mfma waves %d feed %d: %.1f cycles per MFMA; co-runner %.1f cycles per VALU op\n", name, NMW, FEED, (double)h[0] / (iters * 8.0),
         (double)h[1] / (iters * (MODE == 4 ? 24.0 * 8 : 96.0)));
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0, 0>("MFMA alone");
  run<0, 0, 1>("MFMA alone");
  run<1, 0, 0>("co-run VALU fma (96/iter) prio 0");
  run<1, 3, 0>("co-run VALU fma (96/iter) prio 3");
  run<1, 0, 1>("co-run VALU fma (96/iter) prio 0");
  run<1, 3, 1>("co-run VALU fma (96/iter) prio 3");
  run<2, 3, 1>("co-run VALU + LDS writes prio 3");
  run<4, 0, 1>("co-run DPP reduction chain prio 0");
  run<4, 3, 1>("co-run DPP reduction chain prio 3");
  run<3, 0, 1>("co-run exp chain prio 0");
  run<3, 3, 1>("co-run exp chain prio 3");
  run<0, 0, 1, 8>("MFMA alone");
  run<1, 0, 1, 8>("co-run VALU fma prio 0");
  run<1, 3, 1, 8>("co-run VALU fma prio 3");
  run<2, 3, 1, 8>("co-run VALU + LDS writes prio 3");
  run<4, 3, 1, 8>("co-run DPP chain prio 3");
  return 0;
}
