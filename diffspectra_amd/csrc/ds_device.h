// Device-side building blocks for the gfx950 DMT kernels (wave64, fp32 MFMA 32x32x2).
//
// GEMM building block: a row tile X[T][K] lives in LDS (row stride K+4 floats: ds_read_b128 of 16
// different rows then hits 16 distinct 4-bank slots — conflict-free, MI355X_MICROARCH §LDS), the weight
// matrix streams from L2 in an MFMA-B-operand packed layout Wp[K/8][2][Npad][4] so that each lane's
// 4 k-values per k-group are one 16-byte load that is coalesced over the 32 columns of the tile.
// v_mfma_f32_32x32x2_f32: lane l supplies A[row=l&31][k=l>>5], B[k=l>>5][col=l&31]; the accumulator
// element reg of lane l is C[row = (reg&3) + 8*(reg>>2) + 4*(l>>5)][col = l&31].
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DS_LDP 4  // LDS row padding (floats)

__device__ __forceinline__ float ds_silu(float x) { return x / (1.0f + expf(-x)); }
__device__ __forceinline__ float ds_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int ACT>
__device__ __forceinline__ float ds_act(float x) {
  if (ACT == 1) return ds_silu(x);
  if (ACT == 2) return ds_gelu(x);
  if (ACT == 3) return tanhf(x);
  return x;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Row of accumulator register `reg` (0..15) for a lane in half `hh` (lane>>5) of a 32x32 tile.
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// acc[m] += X[m*32 .. m*32+31][8*kg0 .. 8*kg1) * Wp[.., col0 .. col0+31]
template <int MT>
__device__ __forceinline__ void wave_mma(const float* X, int ldx, const float* __restrict__ Wp, int Npad, int col0,
                                         int kg0, int kg1, f32x16 (&acc)[MT]) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const float4* wp = reinterpret_cast<const float4*>(Wp) + (size_t)hh * Npad + col0 + r;
  const float* xr = X + r * ldx + 4 * hh;
#pragma unroll 2
  for (int kg = kg0; kg < kg1; ++kg) {
    const float4 b = wp[(size_t)kg * 2 * Npad];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const float4 a = *reinterpret_cast<const float4*>(xr + m * 32 * ldx + kg * 8);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc[m], 0, 0, 0);
    }
  }
}

template <int MT>
__device__ __forceinline__ void acc_zero(f32x16 (&acc)[MT]) {
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[m][i] = 0.0f;
}

// Visit every accumulator element of this lane: f(row_in_tile, col, value).
template <int MT, class F>
__device__ __forceinline__ void acc_foreach(const f32x16 (&acc)[MT], int row_base, int col0, F f) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) f(row_base + m * 32 + acc_row(i, hh), col0 + r, acc[m][i]);
}

// Whole-workgroup tile GEMM: Y = epi(X[ROWS][K] * Wp[:, 0..NCH*32)), ROWS = 32*MTOT.
// Waves split (column chunk, row tile) work items round-robin; each item is a 32*MT x 32 output block.
template <int MTOT, int MT, class F>
__device__ __forceinline__ void tile_gemm(const float* X, int ldx, int K, const float* __restrict__ Wp, int Npad, int nch,
                                          F epi) {
  static_assert(MTOT % MT == 0, "row tiling");
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  constexpr int RG = MTOT / MT;
  for (int it = wave; it < nch * RG; it += nw) {
    const int ch = it / RG, rg = it % RG;
    f32x16 acc[MT];
    acc_zero<MT>(acc);
    wave_mma<MT>(X + rg * MT * 32 * ldx, ldx, Wp, Npad, ch * 32, 0, K >> 3, acc);
    acc_foreach<MT>(acc, rg * MT * 32, ch * 32, epi);
  }
}

// Read element (k, n) of an MFMA-packed weight (for the few VALU-sized projections).
__device__ __forceinline__ float wp_at(const float* __restrict__ Wp, int Npad, int k, int n) {
  return Wp[((size_t)((k >> 3) * 2 + ((k >> 2) & 1)) * Npad + n) * 4 + (k & 3)];
}

// LayerNorm (no affine, eps 1e-6, biased variance) + adaLN modulate of one LDS row by one wave.
// W = 256: each lane owns 4 consecutive columns; W = 64: one column per lane.
template <int W>
__device__ __forceinline__ void ln_mod_row(float* row, const float* __restrict__ shift, const float* __restrict__ scale) {
  const int lane = threadIdx.x & 63;
  if (W == 256) {
    float4 v = reinterpret_cast<float4*>(row)[lane];
    const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
    v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
    const float var = wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 256.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    const float4 sh = reinterpret_cast<const float4*>(shift)[lane];
    const float4 sc = reinterpret_cast<const float4*>(scale)[lane];
    v.x = v.x * rstd * (1.0f + sc.x) + sh.x;
    v.y = v.y * rstd * (1.0f + sc.y) + sh.y;
    v.z = v.z * rstd * (1.0f + sc.z) + sh.z;
    v.w = v.w * rstd * (1.0f + sc.w) + sh.w;
    reinterpret_cast<float4*>(row)[lane] = v;
  } else {
    float v = row[lane];
    const float mean = wave_sum(v) * (1.0f / 64.0f);
    v -= mean;
    const float var = wave_sum(v * v) * (1.0f / 64.0f);
    const float rstd = 1.0f / sqrtf(var + 1e-6f);
    row[lane] = v * rstd * (1.0f + scale[lane]) + shift[lane];
  }
}
