// Micro-benchmark: cost of ISSUING vector-memory loads from a wave that shares its SIMD with saturated fp32-MFMA waves.
// NMW MFMA waves (NMW/4 per SIMD; FEED 1 re-reads an operand from LDS per 4 MFMAs like wave_mma) + 4 loader waves, each
// issuing batches of 8 independent 1-KiB loads (global_load_dwordx4, rows of a 64 MB table: L2/MALL-resident after the
// first pass), waiting for the batch, accumulating.  Reports cycles per load instruction (issue + exposed latency / 8).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NMW, int FEED, int PRIO, int MFMA_ON>
__global__ __launch_bounds__(NMW * 64 + 256) void k(const float4* __restrict__ table, int rows, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float X[64][260];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 64 * 260; i += NMW * 64 + 256) (&X[0][0])[i] = i * 1e-4f;
  __syncthreads();
  if (wave < NMW) {
    f32x16 a0 = {0}, a1 = {0};
    float y = 1.0f + lane * 0.002f;
    if (MFMA_ON) {
      for (int i = 0; i < iters * 6; ++i) {
        float4 b0 = make_float4(y, y, y, y);
        if (FEED) b0 = *reinterpret_cast<const float4*>(&X[lane & 31][(i & 31) * 8 + 4 * (lane >> 5)]);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.x, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.y, a1, 0, 0, 0);
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.z, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.w, a1, 0, 0, 0);
      }
    }
    f32x16 s = a0 + a1;
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * 1024 + threadIdx.x] = r;
  } else {
    __builtin_amdgcn_s_setprio(PRIO);
    float4 acc = make_float4(0, 0, 0, 0);
    unsigned int row = (blockIdx.x * 4 + (wave - NMW)) * 977u;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
      float4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        row = (row * 1664525u + 1013904223u);
        v[u] = table[(size_t)((row >> 8) & (unsigned)(rows - 1)) * 64 + lane];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 1024 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    if (threadIdx.x == NMW * 64 && blockIdx.x == 0) cyc[0] = t1 - t0;
  }
}

template <int NMW, int FEED, int PRIO, int MFMA_ON>
void run(const float4* table, int rows, const char* name) {
  float* out; unsigned long long* cyc; unsigned long long h = 0;
  (void)hipMalloc(&out, 1 << 22); (void)hipMalloc(&cyc, 16); (void)hipMemset(cyc, 0, 16);
  const int iters = 512;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NMW, FEED, PRIO, MFMA_ON>), dim3(256), dim3(NMW * 64 + 256), 0, 0, table, rows, out, cyc, iters);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  printf("%-34s mfma waves/SIMD %d lds-feed %d loader prio %d: %.0f cycles per 1-KiB load instruction\n", name, MFMA_ON ? NMW / 4 : 0, FEED, PRIO,
         (double)h / (iters * 8.0));
  (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
  const int rows = 65536;   // 64 MB of 1-KiB rows
  float4* table; (void)hipMalloc(&table, (size_t)rows * 1024); (void)hipMemset(table, 0, (size_t)rows * 1024);
  run<8, 0, 0, 0>(table, rows, "loaders alone");
  run<4, 0, 0, 1>(table, rows, "beside MFMA");
  run<4, 0, 3, 1>(table, rows, "beside MFMA");
  run<8, 0, 0, 1>(table, rows, "beside MFMA");
  run<8, 0, 3, 1>(table, rows, "beside MFMA");
  run<8, 1, 0, 1>(table, rows, "beside MFMA + LDS operand reads");
  run<8, 1, 3, 1>(table, rows, "beside MFMA + LDS operand reads");
  return 0;
}
