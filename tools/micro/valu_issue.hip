// Micro-benchmark: issue cost per wave64 instruction of v_fma_f32, v_pk_fma_f32, v_pk_mul_f32, v_exp_f32, v_rcp_f32 with 1 / 2 / 4
// waves per SIMD (independent instruction streams: 8 accumulators in rotation), and of a tanh pair in the packed and scalar forms.
// Build: hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o gpurun_out/valu_issue ; development tool only.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  float a[16];
  f32x2 p[8];
  for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 0.001f + j;
  for (int j = 0; j < 8; ++j) { p[j].x = a[2 * j]; p[j].y = a[2 * j + 1]; }
  const float x = 1.0001f, y = 0.0001f;
  const f32x2 px = {x, x}, py = {y, y};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
    if (OP == 0) {          // 16 v_fma_f32
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_fmaf(a[j], x, y);
    } else if (OP == 1) {   // 8 v_pk_fma_f32 (same 16 results)
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = __builtin_elementwise_fma(p[j], px, py);
    } else if (OP == 2) {   // 8 v_pk_mul_f32
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = p[j] * px;
    } else if (OP == 3) {   // 16 v_exp_f32
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_amdgcn_exp2f(a[j]);
    } else if (OP == 4) {   // 16 v_rcp_f32
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_amdgcn_rcpf(a[j]);
    } else if (OP == 5) {   // 8 packed tanh pairs (16 values): pk_mul, 2 exp, pk_add, 2 rcp, pk_fma
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const f32x2 t = p[j] * 2.8853900817779268f;
        f32x2 e;
        e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
        const f32x2 d = e + 1.0f;
        f32x2 r;
        r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
        p[j] = r * -2.0f + 1.0f;
      }
    } else if (OP == 6) {   // 16 scalar tanh: mul, exp, add, rcp, fma
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float e = __builtin_amdgcn_exp2f(a[j] * 2.8853900817779268f);
        a[j] = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
      }
    } else if (OP == 7) {   // 16 v_mul + 16 v_fma (the logit inner step, scalar)
#pragma unroll
      for (int j = 0; j < 16; ++j) a[j] = __builtin_fmaf(a[j] * x, y, a[j]);
    } else if (OP == 8) {   // 8 pk_mul + 8 pk_fma (the same, packed)
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = __builtin_elementwise_fma(p[j] * px, py, p[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0;
  for (int j = 0; j < 16; ++j) r += a[j];
  for (int j = 0; j < 8; ++j) r += p[j].x + p[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int OP>
void run(const char* name, int n_instr) {
  float* out; unsigned long long* cyc; unsigned long long h;
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 8);
  const int iters = 4096;
  for (int wps : {1, 2, 4}) {
    hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters);
    hipLaunchKernelGGL(k<OP>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s waves/SIMD %d: %6.1f cycles per loop body per wave -> %5.2f cycles per instruction per SIMD (%d instr)\n", name, wps,
           (double)h / iters, (double)h / iters / n_instr / wps, n_instr);
  }
  hipFree(out); hipFree(cyc);
}

int main() {
  run<0>("16 v_fma_f32", 16);
  run<1>("8 v_pk_fma_f32", 8);
  run<2>("8 v_pk_mul_f32", 8);
  run<3>("16 v_exp_f32", 16);
  run<4>("16 v_rcp_f32", 16);
  run<5>("8 packed tanh pairs (3 pk + 4 trans each)", 56);
  run<6>("16 scalar tanh (3 + 2 trans each)", 80);
  run<7>("16 v_mul + 16 v_fma", 32);
  run<8>("8 v_pk_mul + 8 v_pk_fma", 16);
  return 0;
}
