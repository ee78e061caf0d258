// Micro-benchmark: cost of the equi loader's LayerNorm batch (8 rows of 256 features per wave, register resident)
// when the wave shares its SIMD with 0 or 2 waves streaming fp32 MFMAs fed from LDS.  Development tool only.
#include "../../diffspectra_amd/csrc/ds_device.h"
#include <cstdio>

template <int NMW, int PRIO, int VARIANT, int SWAP>
__global__ __launch_bounds__(768) void k(const float* in, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) float X[2][64][260];
  const int wave0 = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wave = SWAP ? (wave0 + 8) % 12 : wave0;   // SWAP: hardware waves 0-3 run the LN role (logical 8-11)
  for (int i = threadIdx.x; i < 2 * 64 * 260; i += 768) (&X[0][0][0])[i] = i * 1e-4f;
  __syncthreads();
  if (wave < 8) {
    if (wave >= NMW) return;
    f32x16 a0 = {0}, a1 = {0};
    float y = 1.0f + lane * 0.002f;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float4 b0 = *reinterpret_cast<const float4*>(&X[0][lane & 31][4 * (lane >> 5)]);
    float4 b1 = *reinterpret_cast<const float4*>(&X[0][32 + (lane & 31)][4 * (lane >> 5)]);
    for (int i = 0; i < iters * 40; ++i) {
      const int kg = (i + 1) & 31;
      const float4 n0 = *reinterpret_cast<const float4*>(&X[0][lane & 31][kg * 8 + 4 * (lane >> 5)]);
      const float4 n1 = *reinterpret_cast<const float4*>(&X[0][32 + (lane & 31)][kg * 8 + 4 * (lane >> 5)]);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.x, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.x, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.y, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.z, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.z, a1, 0, 0, 0);
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b0.w, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, b1.w, a1, 0, 0, 0);
      b0 = n0; b1 = n1;
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    f32x16 s = a0 + a1;
    float r = 0;
    for (int i = 0; i < 16; ++i) r += s[i];
    out[blockIdx.x * 768 + threadIdx.x] = r;
    if (wave == 0 && lane == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
  } else {
    __builtin_amdgcn_s_setprio(PRIO);
    float4 Aa[4], Ca[4], Ab[4], Cb[4], ve[4], sh[4], sc[4];
    const float4* src = reinterpret_cast<const float4*>(in) + lane;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      Aa[u] = src[(u * 7 + 0) * 64]; Ca[u] = src[(u * 7 + 1) * 64]; Ab[u] = src[(u * 7 + 2) * 64]; Cb[u] = src[(u * 7 + 3) * 64];
      ve[u] = src[(u * 7 + 4) * 64]; sh[u] = src[(u * 7 + 5) * 64]; sc[u] = src[(u * 7 + 6) * 64];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
if (VARIANT == 2) {
        // all 8 rows of the batch advance through every reduction step together (ILP 8)
        float4 x[8];
        float st[8];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          x[2 * u].x = (Aa[u].x + Cb[u].x) + ve[u].x; x[2 * u].y = (Aa[u].y + Cb[u].y) + ve[u].y;
          x[2 * u].z = (Aa[u].z + Cb[u].z) + ve[u].z; x[2 * u].w = (Aa[u].w + Cb[u].w) + ve[u].w;
          x[2 * u + 1].x = (Ab[u].x + Ca[u].x) + ve[u].x; x[2 * u + 1].y = (Ab[u].y + Ca[u].y) + ve[u].y;
          x[2 * u + 1].z = (Ab[u].z + Ca[u].z) + ve[u].z; x[2 * u + 1].w = (Ab[u].w + Ca[u].w) + ve[u].w;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] = (x[r].x + x[r].y) + (x[r].z + x[r].w);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0xB1>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0x4E>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0x141>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0x140>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(st[r]), 0x142, 0xa, 0xf, false));
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(st[r]), 0x143, 0xc, 0xf, false));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(st[r]), 63)) * (1.0f / 256.0f);
          x[r].x -= m; x[r].y -= m; x[r].z -= m; x[r].w -= m;
          st[r] = (x[r].x * x[r].x + x[r].y * x[r].y) + (x[r].z * x[r].z + x[r].w * x[r].w);
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0xB1>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0x4E>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0x141>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += dpp_mov<0x140>(st[r]);
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(st[r]), 0x142, 0xa, 0xf, false));
#pragma unroll
        for (int r = 0; r < 8; ++r) st[r] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(st[r]), 0x143, 0xc, 0xf, false));
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          const float rs = __builtin_amdgcn_rsqf(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(st[r]), 63)) * (1.0f / 256.0f) + 1e-6f);
          const int u = r >> 1;
          x[r].x = x[r].x * rs * (1.0f + sc[u].x) + sh[u].x; x[r].y = x[r].y * rs * (1.0f + sc[u].y) + sh[u].y;
          x[r].z = x[r].z * rs * (1.0f + sc[u].z) + sh[u].z; x[r].w = x[r].w * rs * (1.0f + sc[u].w) + sh[u].w;
          reinterpret_cast<float4*>(&X[1][2 * (u * 4 + wave - 8) + (r & 1)][0])[lane] = x[r];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { Aa[u].x += x[2 * u].x * 1e-6f; Ab[u].y += x[2 * u + 1].y * 1e-6f; }
      } else
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        float4 x1, x2;
        x1.x = (Aa[u].x + Cb[u].x) + ve[u].x; x1.y = (Aa[u].y + Cb[u].y) + ve[u].y;
        x1.z = (Aa[u].z + Cb[u].z) + ve[u].z; x1.w = (Aa[u].w + Cb[u].w) + ve[u].w;
        x2.x = (Ab[u].x + Ca[u].x) + ve[u].x; x2.y = (Ab[u].y + Ca[u].y) + ve[u].y;
        x2.z = (Ab[u].z + Ca[u].z) + ve[u].z; x2.w = (Ab[u].w + Ca[u].w) + ve[u].w;
        if (VARIANT == 0) {
          x1 = ln_mod_reg256(x1, sh[u], sc[u]);
          x2 = ln_mod_reg256(x2, sh[u], sc[u]);
        } else {
          // both rows' reductions interleaved explicitly, stats kept in VGPRs (no readlane round trip through SGPRs)
          float s1 = (x1.x + x1.y) + (x1.z + x1.w), s2 = (x2.x + x2.y) + (x2.z + x2.w);
          s1 = row16_sum(s1); s2 = row16_sum(s2);
          s1 += __shfl_xor(s1, 16); s2 += __shfl_xor(s2, 16);
          s1 += __shfl_xor(s1, 32); s2 += __shfl_xor(s2, 32);
          const float m1 = s1 * (1.0f / 256.0f), m2 = s2 * (1.0f / 256.0f);
          x1.x -= m1; x1.y -= m1; x1.z -= m1; x1.w -= m1; x2.x -= m2; x2.y -= m2; x2.z -= m2; x2.w -= m2;
          float q1 = (x1.x * x1.x + x1.y * x1.y) + (x1.z * x1.z + x1.w * x1.w), q2 = (x2.x * x2.x + x2.y * x2.y) + (x2.z * x2.z + x2.w * x2.w);
          q1 = row16_sum(q1); q2 = row16_sum(q2);
          q1 += __shfl_xor(q1, 16); q2 += __shfl_xor(q2, 16);
          q1 += __shfl_xor(q1, 32); q2 += __shfl_xor(q2, 32);
          const float r1 = __builtin_amdgcn_rsqf(q1 * (1.0f / 256.0f) + 1e-6f), r2 = __builtin_amdgcn_rsqf(q2 * (1.0f / 256.0f) + 1e-6f);
          x1.x = x1.x * r1 * (1.0f + sc[u].x) + sh[u].x; x1.y = x1.y * r1 * (1.0f + sc[u].y) + sh[u].y;
          x1.z = x1.z * r1 * (1.0f + sc[u].z) + sh[u].z; x1.w = x1.w * r1 * (1.0f + sc[u].w) + sh[u].w;
          x2.x = x2.x * r2 * (1.0f + sc[u].x) + sh[u].x; x2.y = x2.y * r2 * (1.0f + sc[u].y) + sh[u].y;
          x2.z = x2.z * r2 * (1.0f + sc[u].z) + sh[u].z; x2.w = x2.w * r2 * (1.0f + sc[u].w) + sh[u].w;
        }
        reinterpret_cast<float4*>(&X[1][2 * (u * 4 + wave - 8)][0])[lane] = x1;
        reinterpret_cast<float4*>(&X[1][2 * (u * 4 + wave - 8) + 1][0])[lane] = x2;
        Aa[u].x += x1.x * 1e-6f;   // loop-carried so iterations are not hoisted
        Ab[u].y += x2.y * 1e-6f;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 768 + threadIdx.x] = Aa[0].x + Ab[1].y + X[1][lane][3];
    if (wave == 8 && lane == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
  }
}

template <int NMW, int PRIO, int VARIANT, int SWAP = 0>
void run(const char* name) {
  float *in, *out; unsigned long long* cyc; unsigned long long h[2] = {0, 0};
  hipMalloc(&in, 1 << 20); hipMemset(in, 0x3c, 1 << 20); hipMalloc(&out, 1 << 22); hipMalloc(&cyc, 16); hipMemset(cyc, 0, 16);
  const int iters = 512;
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((k<NMW, PRIO, VARIANT, SWAP>), dim3(256), dim3(768), 0, 0, in, out, cyc, iters);
  hipDeviceSynchronize();
  hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("%-34s mfma waves %d prio %d variant %d: %7.1f cycles per MFMA per SIMD; LN batch of 8 rows %8.1f cycles\n", name, NMW, PRIO,
         VARIANT, NMW ? (double)h[0] / (iters * 40 * 8.0) / (NMW / 4) : 0.0, (double)h[1] / iters);
  hipFree(in); hipFree(out); hipFree(cyc);
}

int main() {
  run<0, 0, 0>("LN alone");
  run<4, 0, 0>("LN + 1 MFMA wave/SIMD");
  run<4, 3, 0>("LN + 1 MFMA wave/SIMD");
  run<8, 0, 0>("LN + 2 MFMA waves/SIMD");
  run<8, 3, 0>("LN + 2 MFMA waves/SIMD");
  run<0, 0, 1>("LN alone");
  run<8, 0, 1>("LN + 2 MFMA waves/SIMD");
  run<8, 3, 1>("LN + 2 MFMA waves/SIMD");
  run<0, 0, 2>("LN alone, 8 rows interleaved");
  run<8, 0, 2>("LN 8 rows interleaved");
  run<8, 3, 2>("LN 8 rows interleaved");
  return 0;
}
