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
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DS_LDP 4  // LDS row padding (floats)

// Transcendentals: hardware exp2/rcp based forms (abs. error ~1e-7, far inside the 1e-5 per-kernel gate); the
// libm versions cost 3-5x the VALU issue slots and these sit in MFMA epilogues.
__device__ __forceinline__ float ds_silu(float x) { return __fdividef(x, 1.0f + __expf(-x)); }
// tanh(x) = 1 - 2 / (exp(2x) + 1): five VALU issues (mul, exp2, add, rcp, fma), saturates correctly through inf/0, absolute
// error <= 1.2e-7 (one ulp of 1.0) - the 32 768 tanh per k_edge_geom tile are that kernel's VALU bill next to its MFMAs.
__device__ __forceinline__ float ds_tanh(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// Two at a time: the multiply, add and fma become packed-fp32 issues (v_pk_*), 7 VALU issues per 2 values instead of 10.
__device__ __forceinline__ f32x2 ds_tanh2(f32x2 x) {
  const f32x2 t = x * 2.8853900817779268f;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return r * -2.0f + 1.0f;
}
// tanh of a pair that already carries the factor 2 log2(e) (DS_TANH_PRESCALE, folded into the packed lin_edge0 / lin_edge1 weights by
// engine.py): one packed multiply per pair less next to the transcendental-bound projection of k_attn_fused
__device__ __forceinline__ f32x2 ds_tanh2_prescaled(f32x2 t) {
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return r * -2.0f + 1.0f;
}
__device__ __forceinline__ f32x2 ds_silu2(f32x2 x) {   // x / (1 + exp(-x)), 8 issues per 2 values instead of 10
  const f32x2 t = x * -1.4426950408889634f;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(t.x); e.y = __builtin_amdgcn_exp2f(t.y);
  const f32x2 d = e + 1.0f;
  f32x2 r;
  r.x = __builtin_amdgcn_rcpf(d.x); r.y = __builtin_amdgcn_rcpf(d.y);
  return x * r;
}
__device__ __forceinline__ float ds_gelu(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

template <int ACT>
__device__ __forceinline__ float ds_act(float x) {
  if (ACT == 1) return ds_silu(x);
  if (ACT == 2) return ds_gelu(x);
  if (ACT == 3) return ds_tanh(x);
  return x;
}

// Cross-lane sums on the DPP path (no LDS crossbar round trips): quad xor-1, quad xor-2, row_half_mirror, row_mirror
// leave every lane of a 16-lane row holding the row's sum; the wave sum adds the four rows via two more steps.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  return v;
}
__device__ __forceinline__ float wave_sum(float v) {
  v = row16_sum(v);
  // rows 1,3 += row 0,2 (row_bcast:15, row_mask 0xA); rows 2,3 += lane 31 (row_bcast:31, row_mask 0xC): lane 63 then
  // holds (r3 + r2) + (r1 + r0).  Pure VALU: no trip through the LDS pipe the MFMA waves' operand reads keep busy.
  // Written as the fused v_add_f32_dpp (rows outside row_mask keep their value): through the builtin the compiler emits
  // v_mov 0 + v_mov_dpp + v_add per step, and VALU issue slots next to the MFMA stream are what the LayerNorms cost.
  asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa\n\ts_nop 1\n\t"
               "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc" : "+v"(v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// LayerNorm(no affine, eps 1e-6, biased variance) + modulate on register-resident rows.
// 256-wide row: one float4 per lane of a wave.  64-wide row: one float4 per lane of a 16-lane DPP row.
__device__ __forceinline__ float4 ln_mod_reg256(float4 v, float4 sh, float4 sc) {
  const float mean = wave_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 256.0f);
  v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
  const float var = wave_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 256.0f);
  const float rstd = __builtin_amdgcn_rsqf(var + 1e-6f);
  v.x = v.x * rstd * (1.0f + sc.x) + sh.x; v.y = v.y * rstd * (1.0f + sc.y) + sh.y;
  v.z = v.z * rstd * (1.0f + sc.z) + sh.z; v.w = v.w * rstd * (1.0f + sc.w) + sh.w;
  return v;
}
__device__ __forceinline__ float4 ln_mod_reg64(float4 v, float4 sh, float4 sc) {
  const float mean = row16_sum((v.x + v.y) + (v.z + v.w)) * (1.0f / 64.0f);
  v.x -= mean; v.y -= mean; v.z -= mean; v.w -= mean;
  const float var = row16_sum((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w)) * (1.0f / 64.0f);
  const float rstd = __builtin_amdgcn_rsqf(var + 1e-6f);
  v.x = v.x * rstd * (1.0f + sc.x) + sh.x; v.y = v.y * rstd * (1.0f + sc.y) + sh.y;
  v.z = v.z * rstd * (1.0f + sc.z) + sh.z; v.w = v.w * rstd * (1.0f + sc.w) + sh.w;
  return v;
}

// 256-wide rows, FOUR rows per wave pass: a row lives in one 16-lane DPP row, lane j of it holds the float4s at columns
// 4j + 64u (u = 0..3).  Every lane works on every step and the row sum is four DPP steps with no broadcast/readlane tail:
// ~15 VALU issues per row against ~55 for the one-row-per-wave form.  v[u], sh[u], sc[u]: the lane's four float4s.
__device__ __forceinline__ void ln_mod_quad256(float4 (&v)[4], const float4 (&sh)[4], const float4 (&sc)[4]) {
  f32x2 lo[4], hi[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { lo[u].x = v[u].x; lo[u].y = v[u].y; hi[u].x = v[u].z; hi[u].y = v[u].w; }
  const f32x2 s2 = ((lo[0] + hi[0]) + (lo[1] + hi[1])) + ((lo[2] + hi[2]) + (lo[3] + hi[3]));
  const float mean = row16_sum(s2.x + s2.y) * (1.0f / 256.0f);
#pragma unroll
  for (int u = 0; u < 4; ++u) { lo[u] -= mean; hi[u] -= mean; }
  const f32x2 q2 = ((lo[0] * lo[0] + hi[0] * hi[0]) + (lo[1] * lo[1] + hi[1] * hi[1])) +
                   ((lo[2] * lo[2] + hi[2] * hi[2]) + (lo[3] * lo[3] + hi[3] * hi[3]));
  const float rstd = __builtin_amdgcn_rsqf(row16_sum(q2.x + q2.y) * (1.0f / 256.0f) + 1e-6f);
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    f32x2 slo, shi, clo, chi;
    slo.x = sh[u].x; slo.y = sh[u].y; shi.x = sh[u].z; shi.y = sh[u].w;
    clo.x = sc[u].x; clo.y = sc[u].y; chi.x = sc[u].z; chi.y = sc[u].w;
    lo[u] = (lo[u] * rstd) * (clo + 1.0f) + slo;
    hi[u] = (hi[u] * rstd) * (chi + 1.0f) + shi;
    v[u] = make_float4(lo[u].x, lo[u].y, hi[u].x, hi[u].y);
  }
}

// Row of accumulator register `reg` (0..15) for a lane in half `hh` (lane>>5) of a 32x32 tile.
__device__ __forceinline__ int acc_row(int reg, int hh) { return (reg & 3) + 8 * (reg >> 2) + 4 * hh; }

// First group (4 k-groups) of B fragments of a wave_mma call.  Weights do not depend on anything a workgroup computes, so a
// kernel requests the NEXT GEMM phase's first group BEFORE the barrier / epilogue in front of it and the L2 latency
// (1-2 us under load: it was ~15k cycles of dead time per barrier-separated phase) overlaps with that work.
struct BFrag {
  float4 v[4];
};

// Weight stream through buffer loads: one 128-bit resource per matrix in SGPRs, the lane's position inside a k-group as a
// 32-bit VGPR offset computed once per call, the k-group stride as an SGPR offset - no per-load 64-bit VALU address
// arithmetic (two VALU issues per load otherwise, and VALU issue slots next to the MFMA stream are the scarce resource).
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
struct WStream {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;      // bytes: ((lane >> 5) * Npad + col0 + (lane & 31)) * 16
  int kstride;   // bytes between k-groups: 2 * Npad * 16
};
__device__ __forceinline__ WStream wstream(const float* __restrict__ Wp, int Npad, int col0) {
  const int lane = threadIdx.x & 63;
  WStream w;
  // the weight pointer is wave-uniform by construction; readfirstlane tells the compiler so (otherwise it wraps every
  // buffer load in a waterfall loop over "possibly different" descriptors)
  const unsigned long long pw = reinterpret_cast<unsigned long long>(Wp);
  const unsigned long long pu = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw >> 32))) << 32) |
                                static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw)));
  w.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(pu), 0, 0x7fffffff, 0x00020000);
  w.voff = (((lane >> 5) * Npad + col0 + (lane & 31)) << 4);
  w.kstride = Npad << 5;
  return w;
}
__device__ __forceinline__ float4 wload(const WStream& w, int kg) {
  const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, kg * w.kstride, 0);
  return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ BFrag bfrag_load(const float* __restrict__ Wp, int Npad, int col0, int kg0, int kg1) {
  const WStream w = wstream(Wp, Npad, col0);
  BFrag f;
#pragma unroll
  for (int j = 0; j < 4; ++j) f.v[j] = wload(w, min(kg0 + j, kg1 - 1));
  return f;
}

// acc[m] += X[m*32 .. m*32+31][8*kg0 .. 8*kg1) * Wp[.., col0 .. col0+31]
// Software-pipelined: the B fragments (L2/MALL latency, 300-900 cycles) of the NEXT group of 4 k-groups are
// requested before the current group's MFMAs issue, so one wave per SIMD already covers the load latency.
// TRANS = true swaps the MFMA operands: the accumulator then holds the TRANSPOSED block — lane l owns row
// (l & 31) of the X tile and register reg is output column col0 + acc_row(reg, l >> 5) — which is exactly the B-operand
// layout a following MFMA needs to contract over those columns without touching LDS (k_equi_pairs).
// xkg0: k-group of the weight matrix that column 0 of X corresponds to (X holds a K-slice of the operand).
// first: the call's first B group if the caller already requested it (bfrag_load with the same col0 / kg0 / kg1).
template <int MT, bool TRANS = false>
__device__ __forceinline__ void wave_mma(const float* X, int ldx, const float* __restrict__ Wp, int Npad, int col0,
                                         int kg0, int kg1, f32x16 (&acc)[MT], int xkg0 = 0, const BFrag* first = nullptr) {
  constexpr int G = 4;
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const WStream ws = wstream(Wp, Npad, col0);
  const float* xr = X + r * ldx + 4 * hh - xkg0 * 8;
  float4 bc[G], bn[G];
  if (first) {
#pragma unroll
    for (int j = 0; j < G; ++j) bc[j] = first->v[j];
  } else {
#pragma unroll
    for (int j = 0; j < G; ++j) bc[j] = wload(ws, min(kg0 + j, kg1 - 1));
  }
  // A fragments are double-buffered one k-group ahead: hipcc otherwise issues each ds_read_b128 right in front of the
  // MFMA that consumes it and the LDS latency (~100 cycles per 512 cycles of MFMA) is exposed at 1-2 waves per SIMD.
  float4 a_cur[MT], a_nxt[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m) a_cur[m] = *reinterpret_cast<const float4*>(xr + m * 32 * ldx + kg0 * 8);
  for (int g = kg0; g < kg1; g += G) {
    if (g + G < kg1) {
#pragma unroll
      for (int j = 0; j < G; ++j) bn[j] = wload(ws, min(g + G + j, kg1 - 1));
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      if (g + j < kg1) {
        const int kn = min(g + j + 1, kg1 - 1);
#pragma unroll
        for (int m = 0; m < MT; ++m) a_nxt[m] = *reinterpret_cast<const float4*>(xr + m * 32 * ldx + kn * 8);
        __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of this k-group's MFMAs
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const float4 a = a_cur[m];
          if (TRANS) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[j].x, a.x, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[j].y, a.y, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[j].z, a.z, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bc[j].w, a.w, acc[m], 0, 0, 0);
          } else {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bc[j].x, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bc[j].y, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bc[j].z, acc[m], 0, 0, 0);
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bc[j].w, acc[m], 0, 0, 0);
          }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) a_cur[m] = a_nxt[m];
      }
    }
#pragma unroll
    for (int j = 0; j < G; ++j) bc[j] = bn[j];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Split-fp16 GEMM building block: fp32-level accuracy on the f16 matrix pipe (v_mfma_f32_32x32x16_f16: 32 cycles for
// 16 384 MACs against 64 cycles for 2 048 on the fp32 MFMA).  Every operand value a is carried as two fp16 numbers,
// a = a1 + a2 / 2048 with a1 = fp16(a) (round to nearest) and a2 = fp16((a - a1) * 2048): the difference is exact in fp32 and
// |a - a1 - a2/2048| <= 2^-23 |a|.  A product is evaluated as a1 b1 + (a1 b2 + a2 b1) / 2048 - three MFMAs per 16-deep
// k-block, the first into `hi`, the other two into `lo` (fp16 x fp16 products are exact in the fp32 accumulate); the dropped
// a2 b2 / 2^22 term is below the representation error.  Measured through the whole DMT: parity gates and the 1000-step
// trajectory drift are unchanged against the fp32-MFMA build.
//   X tile in LDS : [row][plane 0: K halves | plane 1: K halves | 8 halves pad]  (row stride 2K + 8 halves = K + 4 dwords: the
//                   16-byte fragment reads of 16 consecutive rows hit 16 distinct 4-bank groups, like the fp32 tiles)
//   weights       : two planes in MFMA operand order, halves [plane][K/16][k-half][N][8] (engine.pack_linear_f16_split)
// Range: a split value covers fp16's exponent range, not fp32's.  Every kernel that converts fp32 to split planes first calls
// ds_fp16_saturate(): MODE.FP16_OVFL (hwreg MODE bit 23) makes v_cvt_f16_f32 clamp an overflowing result to +-65504 instead
// of producing inf (true inf / NaN inputs are preserved), at no instruction cost - measured on gfx950 with
// tools/micro/fp16_ovfl.hip: 7e4 -> (65504, 65504) = 65536, 1e6 -> 65536, inf -> inf, NaN -> NaN.  An activation with
// |x| >= 65536 therefore SATURATES at +-65535.98 (both planes clamp at 65504) where it used to turn into inf - inf = NaN; values below
// 65520 are converted exactly as before.  Weights are range-checked at pack time (engine.pack_linear_f16_split).
__device__ __forceinline__ void ds_fp16_saturate() { __builtin_amdgcn_s_setreg(1 | (23 << 6) | (0 << 11), 1); }

__device__ __forceinline__ void split_store4(_Float16* row, int K, int col, float4 v) {   // columns col .. col+3 of a tile row
  const float xs[4] = {v.x, v.y, v.z, v.w};
  h4 h1, h2;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    h1[t] = (_Float16)xs[t];
    h2[t] = (_Float16)((xs[t] - (float)h1[t]) * 2048.0f);
  }
  *reinterpret_cast<h4*>(row + col) = h1;
  *reinterpret_cast<h4*>(row + K + col) = h2;
}
__device__ __forceinline__ void split_store1(_Float16* row, int K, int col, float v) {
  const _Float16 h1 = (_Float16)v;
  row[col] = h1;
  row[K + col] = (_Float16)((v - (float)h1) * 2048.0f);
}
__device__ __forceinline__ float split_load1(const _Float16* row, int K, int col) {   // a1 + a2/2048 (2^-23 |a| from the original)
  return fmaf((float)row[K + col], 1.0f / 2048.0f, (float)row[col]);
}

struct WStreamH {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;      // bytes: ((lane >> 5) * N + col0 + (lane & 31)) * 16
  int kstride;   // bytes between k-blocks of one plane: 2 * N * 16
  int pstride;   // bytes between the planes: (K/16) * kstride
};
__device__ __forceinline__ WStreamH wstream_h(const float* __restrict__ Wh, int N, int K, int col0) {
  const int lane = threadIdx.x & 63;
  WStreamH w;
  const unsigned long long pw = reinterpret_cast<unsigned long long>(Wh);
  const unsigned long long pu = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw >> 32))) << 32) |
                                static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw)));
  w.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(pu), 0, 0x7fffffff, 0x00020000);
  w.voff = ((lane >> 5) * N + col0 + (lane & 31)) << 4;
  w.kstride = N << 5;
  w.pstride = (K >> 4) * w.kstride;
  return w;
}
__device__ __forceinline__ h8 wload_h(const WStreamH& w, int plane, int kb) {
  return __builtin_bit_cast(h8, __builtin_amdgcn_raw_buffer_load_b128(w.rsrc, w.voff, plane * w.pstride + kb * w.kstride, 0));
}

// hi[m] += X1 W1, lo[m] += X1 W2 + X2 W1 over k-blocks [kb0, kb1) of the weight matrix; the X tile's column 0 is k-block xkb0.
// TRANS swaps the operands (accumulator = transposed block: lane = tile row, registers = output columns), as in wave_mma.
// UNROLL: k-blocks per unrolled loop body (measured: 16 for the 16-block equi GEMM, 4 for the 4-block edge GEMMs; 1-2 lose 5-15 %)
// XPF: request the next k-block's X fragments ahead of this block's MFMAs (costs 8 MT registers; off for MT = 4).
template <int MT, bool TRANS = false, int UNROLL = 4, bool XPF = true>
__device__ __forceinline__ void wave_mma_h(const _Float16* X, int K_tile, const float* __restrict__ Wh, int N, int K, int col0,
                                           int kb0, int kb1, f32x16 (&hi)[MT], f32x16 (&lo)[MT], int xkb0 = 0) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int ldh = 2 * K_tile + 8;
  const WStreamH ws = wstream_h(Wh, N, K, col0);
  const _Float16* xr = X + r * ldh + 8 * hh - xkb0 * 16;
  h8 w1 = wload_h(ws, 0, kb0), w2 = wload_h(ws, 1, kb0);
  h8 xa[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    xa[m][0] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + kb0 * 16);
    xa[m][1] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + K_tile + kb0 * 16);
  }
#pragma unroll UNROLL
  for (int kb = kb0; kb < kb1; ++kb) {
    const int kn = min(kb + 1, kb1 - 1);
    const h8 w1n = wload_h(ws, 0, kn), w2n = wload_h(ws, 1, kn);
    h8 xn[MT][2];
    if (XPF) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        xn[m][0] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + kn * 16);
        xn[m][1] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + K_tile + kn * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);   // next block's operands are requested ahead of this block's MFMAs
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (TRANS) {
        hi[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, xa[m][0], hi[m], 0, 0, 0);
        lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, xa[m][1], lo[m], 0, 0, 0);
        lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, xa[m][0], lo[m], 0, 0, 0);
      } else {
        hi[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[m][0], w1, hi[m], 0, 0, 0);
        lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[m][1], w1, lo[m], 0, 0, 0);
        lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xa[m][0], w2, lo[m], 0, 0, 0);
      }
    }
    w1 = w1n; w2 = w2n;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      if (XPF) { xa[m][0] = xn[m][0]; xa[m][1] = xn[m][1]; }
      else {
        xa[m][0] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + kn * 16);
        xa[m][1] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + K_tile + kn * 16);
      }
    }
  }
}
// The same product with the weight stream requested PF k-blocks ahead through a register ring (8 registers per block), for a
// wave that is alone on its SIMD's matrix pipe: one block ahead covers 6 MT x 32 cycles of a ~1-2 us L2 round trip, and with
// no second MFMA wave to fill the gap every k-block then waits for its weights (k_equi_pairs: ISA showed s_waitcnt vmcnt
// in front of every block).  NKB (k-blocks) is a compile-time constant so that the ring indices are.  `ring` carries the
// first PF blocks in (requested by the caller with wring_h, e.g. under the previous epilogue) - no request is exposed at all.
#ifndef DS_MMA_XD
#define DS_MMA_XD 1   // X fragments ahead of the MFMAs (k-blocks); measured 1 / 2 / 3: no difference, 1 costs the fewest registers
#endif
template <int PF>
struct WRingH {
  h8 w1[PF], w2[PF];
};
template <int PF>
__device__ __forceinline__ void wring_h(WRingH<PF>& ring, const WStreamH& ws, int kb0) {
#pragma unroll
  for (int i = 0; i < PF; ++i) {
    ring.w1[i] = wload_h(ws, 0, kb0 + i); ring.w2[i] = wload_h(ws, 1, kb0 + i);
    __builtin_amdgcn_sched_barrier(0);   // in consumption order: loads return in issue order and the compiler would sort them by plane
  }
}
template <int MT, bool TRANS, int NKB, int PF, int XD = DS_MMA_XD>
__device__ __forceinline__ void wave_mma_h_deep(const _Float16* X, int K_tile, const WStreamH& ws, WRingH<PF>& ring, int kb0,
                                                f32x16 (&hi)[MT], f32x16 (&lo)[MT], int xkb0 = 0) {
  static_assert(PF <= NKB && XD <= NKB, "ring deeper than the product");
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int ldh = 2 * K_tile + 8;
  const _Float16* xr = X + r * ldh + 8 * hh + (kb0 - xkb0) * 16;
  h8 xq[XD][MT][2];   // the X fragments run XD k-blocks ahead as well (LDS is shared with every other wave of the workgroup)
#pragma unroll
  for (int d = 0; d < XD; ++d)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xq[d][m][0] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + d * 16);
      xq[d][m][1] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + K_tile + d * 16);
    }
#pragma unroll
  for (int i = 0; i < NKB; ++i) {
    __builtin_amdgcn_sched_barrier(0);
    const h8 w1 = ring.w1[i % PF], w2 = ring.w2[i % PF];
    h8 xa[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) { xa[m][0] = xq[i % XD][m][0]; xa[m][1] = xq[i % XD][m][1]; }
    // issue order: no two consecutive MFMAs accumulate into the same registers when there are two row tiles
#define DS_MMA_(ACC, WV, XV) ACC = TRANS ? __builtin_amdgcn_mfma_f32_32x32x16_f16(WV, XV, ACC, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(XV, WV, ACC, 0, 0, 0)
#pragma unroll
    for (int m = 0; m < MT; ++m) DS_MMA_(lo[m], w1, xa[m][1]);
#pragma unroll
    for (int m = 0; m < MT; ++m) DS_MMA_(hi[m], w1, xa[m][0]);
#pragma unroll
    for (int m = 0; m < MT; ++m) DS_MMA_(lo[m], w2, xa[m][0]);
#undef DS_MMA_
    if (i + PF < NKB) {   // the slots just consumed take blocks i + PF / i + XD
      ring.w1[i % PF] = wload_h(ws, 0, kb0 + i + PF);
      ring.w2[i % PF] = wload_h(ws, 1, kb0 + i + PF);
    }
    if (i + XD < NKB) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        xq[i % XD][m][0] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + (i + XD) * 16);
        xq[i % XD][m][1] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + K_tile + (i + XD) * 16);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
// Two 32-column chunks (of one weight matrix or of two) against the SAME X fragments: every X fragment read from LDS feeds
// 6 MT MFMAs instead of 3 MT - half the LDS operand traffic of two wave_mma_h_deep calls and twice the matrix work per
// k-block to cover the operand latencies.  Costs a second accumulator pair and a second ring.
template <int MT, bool TRANS, int NKB, int PF>
__device__ __forceinline__ void wave_mma_h_deep_t2(const _Float16* X, int K_tile, const WStreamH& wsA, const WStreamH& wsB,
                                                   WRingH<PF>& ringA, WRingH<PF>& ringB, int kb0, f32x16 (&hiA)[MT], f32x16 (&loA)[MT],
                                                   f32x16 (&hiB)[MT], f32x16 (&loB)[MT], int xkb0 = 0) {
  static_assert(PF <= NKB, "ring deeper than the product");
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int ldh = 2 * K_tile + 8;
  const _Float16* xr = X + r * ldh + 8 * hh + (kb0 - xkb0) * 16;
  h8 xa[MT][2];
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    xa[m][0] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh);
    xa[m][1] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + K_tile);
  }
#define DS_MMA_(ACC, WV, XV) ACC = TRANS ? __builtin_amdgcn_mfma_f32_32x32x16_f16(WV, XV, ACC, 0, 0, 0) : __builtin_amdgcn_mfma_f32_32x32x16_f16(XV, WV, ACC, 0, 0, 0)
#pragma unroll
  for (int i = 0; i < NKB; ++i) {
    const int in = i + 1 < NKB ? i + 1 : i;
    h8 xn[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      xn[m][0] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + in * 16);
      xn[m][1] = *reinterpret_cast<const h8*>(xr + m * 32 * ldh + K_tile + in * 16);
    }
    __builtin_amdgcn_sched_barrier(0);
    const h8 a1 = ringA.w1[i % PF], a2 = ringA.w2[i % PF], b1 = ringB.w1[i % PF], b2 = ringB.w2[i % PF];
#pragma unroll
    for (int m = 0; m < MT; ++m) { DS_MMA_(loA[m], a1, xa[m][1]); DS_MMA_(loB[m], b1, xa[m][1]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) { DS_MMA_(hiA[m], a1, xa[m][0]); DS_MMA_(hiB[m], b1, xa[m][0]); }
#pragma unroll
    for (int m = 0; m < MT; ++m) { DS_MMA_(loA[m], a2, xa[m][0]); DS_MMA_(loB[m], b2, xa[m][0]); }
    if (i + PF < NKB) {
      ringA.w1[i % PF] = wload_h(wsA, 0, kb0 + i + PF); ringA.w2[i % PF] = wload_h(wsA, 1, kb0 + i + PF);
      ringB.w1[i % PF] = wload_h(wsB, 0, kb0 + i + PF); ringB.w2[i % PF] = wload_h(wsB, 1, kb0 + i + PF);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MT; ++m) { xa[m][0] = xn[m][0]; xa[m][1] = xn[m][1]; }
  }
#undef DS_MMA_
}
// One-call form: both streams' rings set up inside.
template <int MT, bool TRANS, int NKB, int PF>
__device__ __forceinline__ void wave_mma_h_ring_t2(const _Float16* X, int K_tile, const WStreamH& wsA, const WStreamH& wsB, int kb0,
                                                   f32x16 (&hiA)[MT], f32x16 (&loA)[MT], f32x16 (&hiB)[MT], f32x16 (&loB)[MT], int xkb0 = 0) {
  WRingH<PF> ringA, ringB;
#pragma unroll
  for (int i = 0; i < PF; ++i) {   // interleaved in consumption order
    ringA.w1[i] = wload_h(wsA, 0, kb0 + i); ringA.w2[i] = wload_h(wsA, 1, kb0 + i);
    ringB.w1[i] = wload_h(wsB, 0, kb0 + i); ringB.w2[i] = wload_h(wsB, 1, kb0 + i);
    __builtin_amdgcn_sched_barrier(0);
  }
  wave_mma_h_deep_t2<MT, TRANS, NKB, PF>(X, K_tile, wsA, wsB, ringA, ringB, kb0, hiA, loA, hiB, loB, xkb0);
}

// Convenience form: stream + ring set up inside the call (one exposed weight round trip per call instead of one per k-block).
template <int MT, bool TRANS, int NKB, int PF>
__device__ __forceinline__ void wave_mma_h_ring(const _Float16* X, int K_tile, const float* __restrict__ Wh, int N, int K, int col0,
                                                int kb0, f32x16 (&hi)[MT], f32x16 (&lo)[MT], int xkb0 = 0) {
  const WStreamH ws = wstream_h(Wh, N, K, col0);
  WRingH<PF> ring;
  wring_h<PF>(ring, ws, kb0);
  wave_mma_h_deep<MT, TRANS, NKB, PF>(X, K_tile, ws, ring, kb0, hi, lo, xkb0);
}
template <int MT>
__device__ __forceinline__ void split_finish(f32x16 (&hi)[MT], const f32x16 (&lo)[MT]) {   // hi += lo / 2048
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) hi[m][i] = fmaf(lo[m][i], 1.0f / 2048.0f, hi[m][i]);
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

// Store a lane's accumulator block to a row-major global matrix: dst points at element (tile row 0, col0) of the
// destination, LD is its compile-time row stride.  One 64-bit base per lane; the 16*MT row offsets are immediates; full
// tiles (rows_valid >= 32*MT) take a guard-free path.  f(col_in_chunk_lane, value) -> value applies bias / activation.
// The stores are buffer stores: the (wave-uniform) destination in an SGPR resource, the lane's position in one VGPR
// offset, the row as an SGPR offset - a global_store needs a 64-bit VALU address add per row beyond its 4 KB immediate.
struct RowStore {
  __amdgpu_buffer_rsrc_t rsrc;
  int voff;
};
template <int LD>
__device__ __forceinline__ RowStore row_store(float* dst) {
  const int lane = threadIdx.x & 63;
  const unsigned long long pw = reinterpret_cast<unsigned long long>(dst);
  const unsigned long long pu = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw >> 32))) << 32) |
                                static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw)));
  RowStore s;
  s.rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(pu), 0, 0x7fffffff, 0x00020000);
  s.voff = ((4 * (lane >> 5)) * LD + (lane & 31)) * 4;
  return s;
}
__device__ __forceinline__ void row_put(const RowStore& s, int row_byte_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), s.rsrc, s.voff, row_byte_off, 0);
}
template <int MT, int LD, class F>
__device__ __forceinline__ void acc_store(const f32x16 (&acc)[MT], float* __restrict__ dst, int rows_valid, F f) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const RowStore st = row_store<LD>(dst);
  if (rows_valid >= 32 * MT) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) row_put(st, (m * 32 + (i & 3) + 8 * (i >> 2)) * LD * 4, f(r, acc[m][i]));
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m * 32 + (i & 3) + 8 * (i >> 2);
        if (row + 4 * hh < rows_valid) row_put(st, row * LD * 4, f(r, acc[m][i]));
      }
  }
}

// acc_store with a two-wide epilogue f2(f32x2) -> f32x2 (packed-fp32 arithmetic for activation epilogues).
template <int MT, int LD, class F2>
__device__ __forceinline__ void acc_store2(const f32x16 (&acc)[MT], float* __restrict__ dst, int rows_valid, F2 f2) {
  const int hh = (threadIdx.x & 63) >> 5;
  const RowStore st = row_store<LD>(dst);
  if (rows_valid >= 32 * MT) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {   // i even: registers i, i+1 are two consecutive rows
        f32x2 v;
        v.x = acc[m][i]; v.y = acc[m][i + 1];
        v = f2(v);
        const int row0 = m * 32 + (i & 3) + 8 * (i >> 2);
        row_put(st, row0 * LD * 4, v.x);
        row_put(st, (row0 + 1) * LD * 4, v.y);
      }
  } else {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int i = 0; i < 16; i += 2) {
        f32x2 v;
        v.x = acc[m][i]; v.y = acc[m][i + 1];
        v = f2(v);
        const int row0 = m * 32 + (i & 3) + 8 * (i >> 2);
        if (row0 + 4 * hh < rows_valid) row_put(st, row0 * LD * 4, v.x);
        if (row0 + 1 + 4 * hh < rows_valid) row_put(st, (row0 + 1) * LD * 4, v.y);
      }
  }
}

// Whole-workgroup tile GEMM: Y = epi(X[ROWS][K] * Wp[:, 0..NCH*32)), ROWS = 32*MTOT.
// Waves split (column chunk, row tile) work items round-robin; each item is a 32*MT x 32 output block.  The first B group
// of the next item is requested before the current item's epilogue runs; tile_first() returns this wave's first item's
// group so a kernel can request it ahead of the barrier in front of the GEMM.
template <int MTOT, int MT>
__device__ __forceinline__ BFrag tile_first(const float* __restrict__ Wp, int Npad, int K, int nch) {
  constexpr int RG = MTOT / MT;
  const int wave = threadIdx.x >> 6;
  const int it = wave < nch * RG ? wave : 0;
  return bfrag_load(Wp, Npad, (it / RG) * 32, 0, K >> 3);
}

// g(chunk, row_group, acc) receives the whole accumulator block of an item.
template <int MTOT, int MT, class G>
__device__ __forceinline__ void tile_gemm_blk(const float* X, int ldx, int K, const float* __restrict__ Wp, int Npad, int nch,
                                              G g, const BFrag* first = nullptr) {
  static_assert(MTOT % MT == 0, "row tiling");
  const int wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  constexpr int RG = MTOT / MT;
  const int total = nch * RG;
  if (wave >= total) return;
  BFrag cur = first ? *first : bfrag_load(Wp, Npad, (wave / RG) * 32, 0, K >> 3);
  for (int it = wave; it < total; it += nw) {
    const int ch = it / RG, rg = it % RG;
    // compiler barrier: without it LICM hoists the loop-invariant A-fragment LDS reads of ALL k-groups out of this
    // loop (256 VGPRs, spills, occupancy 1); re-reading LDS per chunk is nearly free next to the MFMAs.
    asm volatile("" ::: "memory");
    f32x16 acc[MT];
    acc_zero<MT>(acc);
    wave_mma<MT>(X + rg * MT * 32 * ldx, ldx, Wp, Npad, ch * 32, 0, K >> 3, acc, 0, &cur);
    const int nx = it + nw < total ? it + nw : it;
    cur = bfrag_load(Wp, Npad, (nx / RG) * 32, 0, K >> 3);   // in flight during the epilogue below
    g(ch, rg, acc);
  }
}

// epi(row_in_tile, col, value) per accumulator element.
template <int MTOT, int MT, class F>
__device__ __forceinline__ void tile_gemm(const float* X, int ldx, int K, const float* __restrict__ Wp, int Npad, int nch,
                                          F epi, const BFrag* first = nullptr) {
  constexpr int RG = MTOT / MT;
  tile_gemm_blk<MTOT, MT>(X, ldx, K, Wp, Npad, nch, [&](int ch, int rg, const f32x16 (&acc)[MT]) {
    acc_foreach<MT>(acc, rg * MT * 32, ch * 32, epi);
  }, first);
}

// Read element (k, n) of an MFMA-packed weight (for the few VALU-sized projections).
__device__ __forceinline__ float wp_at(const float* __restrict__ Wp, int Npad, int k, int n) {
  return Wp[((size_t)((k >> 3) * 2 + ((k >> 2) & 1)) * Npad + n) * 4 + (k & 3)];
}
