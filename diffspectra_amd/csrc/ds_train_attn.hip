// diffspectra_amd - SpecFormer's residual-score attention for the bf16 training mode (BASELINE config 5) WITHOUT the [B,H,L,L] tensors.
//
// specformer.py:385-425 adds the previous layer's pre-softmax scores to its own (res_attention): scores_l = scale q_l k_l^T + scores_{l-1}.
// Unrolled, scores_l = scale * [q_0 | .. | q_l] [k_0 | .. | k_l]^T - a product over the concatenated head slices (d_k = 8 per layer, so at
// most 24 of the 32 k-values of two v_mfma_f32_32x32x16_bf16).  So layer l's attention is a flash-style attention whose "head dimension"
// is 8 (l + 1) for the scores and 8 for the values, recomputed on the matrix pipe from the q | k | v projections of layers 0 .. l: no score
// tensor is written (round 3: 2 GB per layer and direction at 256 molecules, 16 ms of a 64 ms step), the backward re-creates the
// probabilities from (row maximum, row sum), and the score gradient that the reference chains through the layers becomes direct
// contributions of layer l's softmax gradient to dq_j, dk_j of every j <= l.
//
// One workgroup per (molecule, head); a wave owns 32-row tiles.  Forward and the query-side backward compute S^T = K Q^T (lane = query,
// registers = keys): row maxima / sums are in-lane reductions plus one exchange between the lane halves, and P^T is at once the B operand
// of the following product over the keys (O^T = V^T P^T, dQ^T = K^T dS^T; accumulator-as-operand, cdna_hip_programming.md section 3).  The
// key-side backward owns key tiles and computes S = Q K^T (lane = key), so dV^T = dO^T P and dK^T = Q^T dS again sum over the register
// index.  The A operands of those second products need the k-permutation of the accumulator layout (element j of lane half h = row
// 16 s + 8 (j >> 2) + 4 h + (j & 3)); they are laid out in LDS once per workgroup.
// The fp32 mode keeps round 3's kernels (materialised scores, fp32 arithmetic: what golden G13 pins).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/diffspectra_hip.h"
#include "../../include/diffspectra_train.h"

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef unsigned short u16x8_t __attribute__((ext_vector_type(8)));

namespace {

#define DST_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DS_OK : DS_ERR_LAUNCH)

constexpr int DK = 8, DM = 128, ROWLD = 3 * DM;   // head slice, model width, row stride of a q | k | v buffer
constexpr int KLD = 40;                           // bf16 per LDS row of the concatenated slices: 32 + 8 of padding (80 bytes)
constexpr float LOG2E = 1.4426950408889634f, LN2 = 0.6931471805599453f;

struct QkvPtrs { const float* p[3]; };
struct GradPtrs { float* p[3]; };

__device__ __forceinline__ bf16x8_t zero8() {
  bf16x8_t z;
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = (__bf16)0.0f;
  return z;
}
__device__ __forceinline__ bf16x8_t load8(const float* p, float mul) {           // eight consecutive floats -> bf16 x 8
  const f32x4_t a = *reinterpret_cast<const f32x4_t*>(p), b = *reinterpret_cast<const f32x4_t*>(p + 4);
  bf16x8_t r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = (__bf16)(a[j] * mul); r[4 + j] = (__bf16)(b[j] * mul); }
  return r;
}
// Workgroups are dealt round-robin over the 8 XCDs (each with its own L2): with (molecule, head) = blockIdx the 16 heads of a molecule -
// which read 32-byte slices of the SAME 128-byte lines of q | k | v - land on eight different L2s.  Remapped, an XCD runs consecutive
// (molecule, head) pairs, so a line is fetched once per XCD that needs it instead of once per head (profiles/r05_sfa_prologue.txt).
__device__ __forceinline__ int xcd_local(int block, int nblocks) {
#ifdef SFA_NO_XCD_MAP
  return block;
#else
  return (nblocks & 7) == 0 ? (block & 7) * (nblocks >> 3) + (block >> 3) : block;
#endif
}
// x as a sum of two bf16 (hi + lo, residual ~2^-17 |x|): the per-row offsets of the backward (row maximum + log2 row sum, D = dO . O) ride
// in SPARE columns of the products that need them - S - m = [q | -m_hi | -m_lo] [k | 1 | 1]^T - instead of one subtraction per score.
__device__ __forceinline__ bf16x8_t split2(float x) {
  bf16x8_t r = zero8();
  const __bf16 hi = (__bf16)x;
  r[0] = hi;
  r[1] = (__bf16)(x - (float)hi);
  return r;
}
__device__ __forceinline__ bf16x8_t ones2() {
  bf16x8_t r = zero8();
  r[0] = (__bf16)1.0f;
  r[1] = (__bf16)1.0f;
  return r;
}
struct Raw8 { f32x4_t a, b; };                                                    // eight consecutive floats as they come from memory
__device__ __forceinline__ Raw8 raw8(const float* p) { return Raw8{*reinterpret_cast<const f32x4_t*>(p), *reinterpret_cast<const f32x4_t*>(p + 4)}; }
__device__ __forceinline__ Raw8 raw8_zero() { const f32x4_t z = {0.0f, 0.0f, 0.0f, 0.0f}; return Raw8{z, z}; }
__device__ __forceinline__ bf16x8_t cvt8(const Raw8& v, float mul) {
  bf16x8_t r;
#pragma unroll
  for (int j = 0; j < 4; ++j) { r[j] = (__bf16)(v.a[j] * mul); r[4 + j] = (__bf16)(v.b[j] * mul); }
  return r;
}
// The table fills of a workgroup REQUEST everything first and write LDS afterwards (FILL_U items per thread and round): written as
// load - convert - store per item, each round was a dependent memory round trip (the stores may alias the loads for all the compiler
// knows) - 8 - 12 of them, 12 us of a workgroup's 30 - 60 us (profiles/r05_sfa_prologue.txt).
constexpr int FILL_U = 4;
__device__ __forceinline__ bf16x8_t acc8(const f32x16_t& x, int s) {              // registers 8 s .. 8 s + 7 of an accumulator as a B fragment
  bf16x8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (__bf16)x[8 * s + j];
  return r;
}
__device__ __forceinline__ int acc_row(int i, int hh) { return (i & 3) + 8 * (i >> 2) + 4 * hh; }   // row of register i in a 32 x 32 accumulator
// position of (row r of a 32-row tile, column c) in a fragment table [tile][s][hh][C][8]: the k-permuted A operand of a product over r
__device__ __forceinline__ int perm_index(int tile, int r, int c, int C) {
  const int s = r >> 4, r16 = r & 15, hh = (r16 >> 2) & 1, j = (r16 >> 3) * 4 + (r16 & 3);
  return ((((tile * 2 + s) * 2 + hh) * C + c) << 3) + j;
}
__device__ __forceinline__ unsigned short bf16_bits(float v) {
  const __bf16 b = (__bf16)v;
  return __builtin_bit_cast(unsigned short, b);
}

// ------------------------------------------------------------------------------------------------------------------ forward
// out [B*L, 128] (head h at columns 8 h ..), stats [B*H*L, 2] = (row maximum of the log2-domain scores, row sum of exp2)
// (eight waves per workgroup for the forward and the query-side backward: the LDS tables of a (molecule, head) are per workgroup, so twice the
// waves share them - 16 waves per CU instead of 8 - and the 11 query tiles split 2 / 1 over the waves instead of 3 / 2)
#ifndef SFA_NW_F
#define SFA_NW_F 8
#endif
#ifndef SFA_NW_Q
#define SFA_NW_Q 8
#endif
#ifndef SFA_NW_K
#define SFA_NW_K 4
#endif
#ifndef SFA_K_MINB
#define SFA_K_MINB 1
#endif
template <int SFA_NW>
__global__ __launch_bounds__(SFA_NW * 64) void k_sfa_fwd(QkvPtrs qkv, int nl, float* __restrict__ stats, float* __restrict__ out, int L, int H, float scale) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int SFA_NT = SFA_NW * 64;
  const int NT = (L + 31) >> 5, LP = NT * 32;
  unsigned short* Kc = lds;                       // [LP][KLD]: concatenated key slices, row = key
  unsigned short* Vf = lds + LP * KLD;            // [NT][2][2][8][8]: V^T in the k-permuted A-operand order
  const int bh = xcd_local(blockIdx.x, gridDim.x), b = bh / H, h = bh % H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int64_t row0 = (int64_t)b * L;
  const float* vsrc = qkv.p[nl - 1];
  for (int base = 0; base < LP * 4; base += SFA_NT * FILL_U) {      // (key, 8-wide chunk): layers 0 .. nl-1, the rest zero
    Raw8 kv[FILL_U], vv = raw8_zero();
    const int vk = (base >> 2) + tid;                              // the value row of key vk rides with the first FILL_U rounds' keys
    const bool vlive = tid < SFA_NT * FILL_U / 4 && vk < LP;
    if (vlive && vk < L) vv = raw8(vsrc + (row0 + vk) * ROWLD + 2 * DM + h * DK);
#pragma unroll
    for (int u = 0; u < FILL_U; ++u) {
      const int i = base + u * SFA_NT + tid, k = i >> 2, c = i & 3;
      kv[u] = raw8_zero();
      if (k < L && c < nl) kv[u] = raw8(qkv.p[c] + (row0 + k) * ROWLD + DM + h * DK);
    }
#pragma unroll
    for (int u = 0; u < FILL_U; ++u) {
      const int i = base + u * SFA_NT + tid, k = i >> 2, c = i & 3;
      if (i < LP * 4) *reinterpret_cast<bf16x8_t*>(Kc + k * KLD + 8 * c) = cvt8(kv[u], 1.0f);
    }
    if (vlive) {
      const u16x8_t vb_ = __builtin_bit_cast(u16x8_t, cvt8(vv, 1.0f));
#pragma unroll
      for (int e = 0; e < DK; ++e) Vf[perm_index(vk >> 5, vk & 31, e, DK)] = vb_[e];
    }
  }
  __syncthreads();
#ifdef SFA_PROLOGUE_ONLY
  if (L > 0) return;
#endif
  const float qmul = scale * LOG2E;
  for (int qt = wave; qt < NT; qt += SFA_NW) {
    const int q = qt * 32 + r, qc = q < L ? q : L - 1;
    bf16x8_t qb[2];
    qb[0] = hh < nl ? load8(qkv.p[hh] + (row0 + qc) * ROWLD + h * DK, qmul) : zero8();
    qb[1] = 2 + hh < nl ? load8(qkv.p[2 + hh] + (row0 + qc) * ROWLD + h * DK, qmul) : zero8();
    float m = -INFINITY, l = 0.0f;
    f32x16_t o;
#pragma unroll
    for (int i = 0; i < 16; ++i) o[i] = 0.0f;
    for (int kt = 0; kt < NT; ++kt) {
      const unsigned short* krow = Kc + (kt * 32 + r) * KLD + 8 * hh;
      f32x16_t s;
#pragma unroll
      for (int i = 0; i < 16; ++i) s[i] = 0.0f;
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(krow), qb[0], s, 0, 0, 0);
      if (nl > 2) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(krow + 16), qb[1], s, 0, 0, 0);
      if (kt == NT - 1) {                                                        // padding keys exist in the last tile only
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kt * 32 + acc_row(i, hh) >= L) s[i] = -INFINITY;
      }
      float mx = -INFINITY;
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[i]);
      mx = fmaxf(mx, __shfl_xor(mx, 32));
      const float mn = fmaxf(m, mx);
      const float alpha = __builtin_amdgcn_exp2f(m - mn);
      float rs = 0.0f;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = __builtin_amdgcn_exp2f(s[i] - mn); rs += s[i]; }
      rs += __shfl_xor(rs, 32);
      l = l * alpha + rs;
      m = mn;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] *= alpha;
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8_t va = r < DK ? *reinterpret_cast<const bf16x8_t*>(Vf + ((((kt * 2 + st) * 2 + hh) * DK + r) << 3)) : zero8();
        o = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, acc8(s, st), o, 0, 0, 0);
      }
    }
    if (q < L) {
      const float inv = 1.0f / l;
      const f32x4_t w = {o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv};      // rows 4 hh .. 4 hh + 3 of O^T = head dims
      *reinterpret_cast<f32x4_t*>(out + (row0 + q) * DM + h * DK + 4 * hh) = w;
      if (hh == 0) { stats[((int64_t)bh * L + q) * 2] = m; stats[((int64_t)bh * L + q) * 2 + 1] = l; }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ backward, query side
// dq_j[q] += scale * sum_k dS[q,k] k_j[k] for every layer j < nl, dS = P (dP - D), dP = dO V^T, D = dO . O
template <int SFA_NW>
__global__ __launch_bounds__(SFA_NW * 64) void k_sfa_bwd_q(QkvPtrs qkv, int nl, const float* __restrict__ stats, const float* __restrict__ out,
                                                   const float* __restrict__ dout, GradPtrs dqkv, int L, int H, float scale, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int SFA_NT = SFA_NW * 64;
  const int NT = (L + 31) >> 5, LP = NT * 32;
  unsigned short* Kc = lds;                       // [LP][KLD]
  unsigned short* Vk = Kc + LP * KLD;             // [LP][8]: value rows (A operand of dP^T = V dO^T)
  unsigned short* KTf = Vk + LP * DK;             // [NT][2][2][32][8]: K^T, k-permuted (A operand of dQ^T = K^T dS^T)
  const int bh = xcd_local(blockIdx.x, gridDim.x), b = bh / H, h = bh % H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int64_t row0 = (int64_t)b * L;
  const float* vsrc = qkv.p[nl - 1];
  for (int base = 0; base < LP * 4; base += SFA_NT * FILL_U) {
    Raw8 kv[FILL_U], vv = raw8_zero();
    const int vk = (base >> 2) + tid;
    const bool vlive = vk < LP;
    if (vlive && vk < L) vv = raw8(vsrc + (row0 + vk) * ROWLD + 2 * DM + h * DK);
#pragma unroll
    for (int u = 0; u < FILL_U; ++u) {
      const int i = base + u * SFA_NT + tid, k = i >> 2, c = i & 3;
      kv[u] = raw8_zero();
      if (k < L && c < nl) kv[u] = raw8(qkv.p[c] + (row0 + k) * ROWLD + DM + h * DK);
    }
#pragma unroll
    for (int u = 0; u < FILL_U; ++u) {
      const int i = base + u * SFA_NT + tid, k = i >> 2, c = i & 3;
      if (i < LP * 4) {
        const bf16x8_t v = cvt8(kv[u], 1.0f);
        *reinterpret_cast<bf16x8_t*>(Kc + k * KLD + 8 * c) = c == nl ? ones2() : v;      // chunk nl: the 1 | 1 columns that meet -m_hi | -m_lo
        const u16x8_t vb_ = __builtin_bit_cast(u16x8_t, v);     // (a bit_cast of the single element v[e] returned element 0 for every e: hipcc 7.0)
#pragma unroll
        for (int e = 0; e < 8; ++e) KTf[perm_index(k >> 5, k & 31, 8 * c + e, 32)] = vb_[e];
      }
    }
    if (vlive) *reinterpret_cast<bf16x8_t*>(Vk + vk * DK) = cvt8(vv, 1.0f);
  }
  __syncthreads();
#ifdef SFA_PROLOGUE_ONLY
  if (L > 0) return;
#endif
  const float qmul = scale * LOG2E;
  for (int qt = wave; qt < NT; qt += SFA_NW) {
    const int q = qt * 32 + r, qc = q < L ? q : L - 1;
    bf16x8_t qb[2];
    qb[0] = hh < nl ? load8(qkv.p[hh] + (row0 + qc) * ROWLD + h * DK, qmul) : zero8();
    qb[1] = 2 + hh < nl ? load8(qkv.p[2 + hh] + (row0 + qc) * ROWLD + h * DK, qmul) : zero8();
    const float* dop = dout + (row0 + qc) * DM + h * DK;
    const float* op = out + (row0 + qc) * DM + h * DK;
    float D = 0.0f;
#pragma unroll
    for (int e = 0; e < DK; ++e) D += dop[e] * op[e];
    // P = exp2(s - m) / l = exp2(s - (m + log2 l)); dS = P (dP - D).  Both offsets are per query = per lane: they enter the products as two
    // extra columns each (split2), so the loop below is exp2 and one product per score.
    const float m = stats[((int64_t)bh * L + qc) * 2] + __log2f(stats[((int64_t)bh * L + qc) * 2 + 1]);
    const bf16x8_t dob = hh == 0 ? load8(dop, 1.0f) : split2(-D);                 // B operand of dP^T: dO[q][d = 8 hh + j] | -D
    const bf16x8_t vone = ones2();
    if (hh == nl) qb[0] = split2(-m);
    if (2 + hh == nl) qb[1] = split2(-m);
    f32x16_t dq;
#pragma unroll
    for (int i = 0; i < 16; ++i) dq[i] = 0.0f;
    for (int kt = 0; kt < NT; ++kt) {
      const unsigned short* krow = Kc + (kt * 32 + r) * KLD + 8 * hh;
      f32x16_t s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.0f; dp[i] = 0.0f; }
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(krow), qb[0], s, 0, 0, 0);
      if (nl >= 2) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(krow + 16), qb[1], s, 0, 0, 0);
      const bf16x8_t va = hh == 0 ? *reinterpret_cast<const bf16x8_t*>(Vk + (kt * 32 + r) * DK) : vone;
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, dob, dp, 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s[i] = __builtin_amdgcn_exp2f(s[i]) * dp[i];                            // dS^T = P^T (dP^T - D)
      }
      if (kt == NT - 1) {                                                        // padding keys exist in the last tile only
#pragma unroll
        for (int i = 0; i < 16; ++i)
          if (kt * 32 + acc_row(i, hh) >= L) s[i] = 0.0f;
      }
#pragma unroll
      for (int st = 0; st < 2; ++st)
        dq = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(KTf + ((((kt * 2 + st) * 2 + hh) * 32 + r) << 3)), acc8(s, st), dq, 0, 0, 0);
    }
    if (q < L)
      for (int j = 0; j < nl; ++j) {                                             // rows 8 j + 4 hh .. + 3 of dQ^T = layer j, head dims 4 hh ..
        float* g = dqkv.p[j] + (row0 + q) * ROWLD + h * DK + 4 * hh;
        f32x4_t w = {0.0f, 0.0f, 0.0f, 0.0f};
        if (accumulate) w = *reinterpret_cast<f32x4_t*>(g);
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] += scale * (j == 0 ? dq[e] : j == 1 ? dq[4 + e] : dq[8 + e]);
        *reinterpret_cast<f32x4_t*>(g) = w;
      }
  }
}

// ------------------------------------------------------------------------------------------------------------------ backward, key side
// dv[k] = sum_q P[q,k] dO[q]; dk_j[k] += scale * sum_q dS[q,k] q_j[q]
template <int SFA_NW>
__global__ __launch_bounds__(SFA_NW * 64, SFA_K_MINB) void k_sfa_bwd_kv(QkvPtrs qkv, int nl, const float* __restrict__ stats, const float* __restrict__ out,
                                                    const float* __restrict__ dout, GradPtrs dqkv, int L, int H, float scale, int accumulate) {
  extern __shared__ __attribute__((aligned(16))) unsigned short lds[];
  constexpr int SFA_NT = SFA_NW * 64;
  const int NT = (L + 31) >> 5, LP = NT * 32;
  unsigned short* Qs = lds;                       // [LP][KLD]: concatenated query slices * scale * log2 e, row = query (A operand of S)
  unsigned short* QTf = Qs + LP * KLD;            // [NT][2][2][32][8]: Q^T (scaled), k-permuted (A operand of dK^T = Q^T dS)
  unsigned short* dOTf = QTf + NT * 4 * 32 * 8;   // [NT][2][2][8][8]: dO^T, k-permuted (A operand of dV^T = dO^T P)
  unsigned int* DH = reinterpret_cast<unsigned int*>(dOTf + NT * 4 * DK * 8);   // [LP]: -D = -(dO . O) as two bf16 (split2), one word per query
  const int bh = xcd_local(blockIdx.x, gridDim.x), b = bh / H, h = bh % H;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int64_t row0 = (int64_t)b * L;
  const float qmul = scale * LOG2E;
  for (int base = 0; base < LP * 4; base += SFA_NT * FILL_U) {
    Raw8 qv[FILL_U], dv_ = raw8_zero(), ov = raw8_zero();
    const int qq = (base >> 2) + tid;                              // the per-query items (dO row, D, row statistics) ride with the rounds' queries
    const bool qlive = qq < LP;
    float m = 0.0f, l = 0.0f;
    if (qlive && qq < L) {
      dv_ = raw8(dout + (row0 + qq) * DM + h * DK);
      ov = raw8(out + (row0 + qq) * DM + h * DK);
      m = stats[((int64_t)bh * L + qq) * 2];
      l = stats[((int64_t)bh * L + qq) * 2 + 1];
    }
#pragma unroll
    for (int u = 0; u < FILL_U; ++u) {
      const int i = base + u * SFA_NT + tid, q = i >> 2, c = i & 3;
      qv[u] = raw8_zero();
      if (q < L && c < nl) qv[u] = raw8(qkv.p[c] + (row0 + q) * ROWLD + h * DK);
    }
#pragma unroll
    for (int u = 0; u < FILL_U; ++u) {
      const int i = base + u * SFA_NT + tid, q = i >> 2, c = i & 3;
      if (i < LP * 4) {
        const bf16x8_t v = cvt8(qv[u], qmul);
        if (c != nl) *reinterpret_cast<bf16x8_t*>(Qs + q * KLD + 8 * c) = v;    // (chunk nl: the query's own thread writes -m there, below)
        const u16x8_t vb_ = __builtin_bit_cast(u16x8_t, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) QTf[perm_index(q >> 5, q & 31, 8 * c + e, 32)] = vb_[e];
      }
    }
    if (qlive) {
      float D = 0.0f;
#pragma unroll
      for (int e = 0; e < 4; ++e) D += dv_.a[e] * ov.a[e];
#pragma unroll
      for (int e = 0; e < 4; ++e) D += dv_.b[e] * ov.b[e];
      const u16x8_t vb_ = __builtin_bit_cast(u16x8_t, cvt8(dv_, 1.0f));
#pragma unroll
      for (int e = 0; e < 8; ++e) dOTf[perm_index(qq >> 5, qq & 31, e, DK)] = vb_[e];
      // P = exp2(s - (m + log2 l)); dS = P (dP - D): both offsets are per query = per ROW of the products S = Q K^T and dP = dO V^T, so they
      // ride as two extra columns each (split2) against 1 | 1 on the key side.  Padded queries: s = -1e30 -> P = 0.
      *reinterpret_cast<bf16x8_t*>(Qs + qq * KLD + 8 * nl) = split2(qq < L ? -(m + __log2f(l)) : -1e30f);
      const bf16x8_t dsp = split2(-D);
      DH[qq] = (unsigned int)bf16_bits((float)dsp[0]) | ((unsigned int)bf16_bits((float)dsp[1]) << 16);
    }
  }
  __syncthreads();
#ifdef SFA_PROLOGUE_ONLY
  if (L > 0) return;
#endif
  const float* vsrc = qkv.p[nl - 1];
  for (int kt = wave; kt < NT; kt += SFA_NW) {
    const int k = kt * 32 + r, kc = k < L ? k : L - 1;
    bf16x8_t kb[2];                                                              // B operand of S: K[key][8 hh + j] of k-block 0 / 1
    kb[0] = hh < nl ? load8(qkv.p[hh] + (row0 + kc) * ROWLD + DM + h * DK, 1.0f) : zero8();
    kb[1] = 2 + hh < nl ? load8(qkv.p[2 + hh] + (row0 + kc) * ROWLD + DM + h * DK, 1.0f) : zero8();
    if (hh == nl) kb[0] = ones2();
    if (2 + hh == nl) kb[1] = ones2();
    const bf16x8_t vb = hh == 0 ? load8(vsrc + (row0 + kc) * ROWLD + 2 * DM + h * DK, 1.0f) : ones2();   // B operand of dP: V[key][d] | 1 | 1
    f32x16_t dv, dkc;
#pragma unroll
    for (int i = 0; i < 16; ++i) { dv[i] = 0.0f; dkc[i] = 0.0f; }
    // A operand of dP: dO[query r][d] (the rows stay in L2: 32 bytes per query and head), fetched ONE QUERY TILE AHEAD from a clamped
    // address and masked when it is used - loaded where it was consumed, every tile paid an L2 round trip in front of its MFMAs
    const float* dobase = dout + row0 * DM + h * DK;
    f32x4_t dn0 = *reinterpret_cast<const f32x4_t*>(dobase + (int64_t)min(r, L - 1) * DM);
    f32x4_t dn1 = *reinterpret_cast<const f32x4_t*>(dobase + (int64_t)min(r, L - 1) * DM + 4);
    for (int qt = 0; qt < NT; ++qt) {
      const unsigned short* qrow = Qs + (qt * 32 + r) * KLD + 8 * hh;
      f32x16_t s, dp;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s[i] = 0.0f; dp[i] = 0.0f; }
      const int qa = qt * 32 + r;
      bf16x8_t doa;
      {
        const bool live = hh == 0 && qa < L;
#pragma unroll
        for (int j = 0; j < 4; ++j) { doa[j] = (__bf16)(live ? dn0[j] : 0.0f); doa[4 + j] = (__bf16)(live ? dn1[j] : 0.0f); }
        if (hh == 1) {                                                           // columns 8, 9 of the dO row: -D_hi | -D_lo
          const unsigned int dh = DH[qa];
          doa[0] = __builtin_bit_cast(__bf16, (unsigned short)(dh & 0xffffu));
          doa[1] = __builtin_bit_cast(__bf16, (unsigned short)(dh >> 16));
        }
        const int qn = min(qa + 32, L - 1);
        dn0 = *reinterpret_cast<const f32x4_t*>(dobase + (int64_t)qn * DM);
        dn1 = *reinterpret_cast<const f32x4_t*>(dobase + (int64_t)qn * DM + 4);
      }
      s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(qrow), kb[0], s, 0, 0, 0);
      if (nl >= 2) s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(qrow + 16), kb[1], s, 0, 0, 0);
      dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doa, vb, dp, 0, 0, 0);
      f32x16_t p;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        p[i] = __builtin_amdgcn_exp2f(s[i]);
        s[i] = p[i] * dp[i];                                                     // dS = P (dP - D)
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const bf16x8_t da = r < DK ? *reinterpret_cast<const bf16x8_t*>(dOTf + ((((qt * 2 + st) * 2 + hh) * DK + r) << 3)) : zero8();
        dv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da, acc8(p, st), dv, 0, 0, 0);
        dkc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8_t*>(QTf + ((((qt * 2 + st) * 2 + hh) * 32 + r) << 3)), acc8(s, st), dkc, 0, 0, 0);
      }
    }
    if (k < L) {
      const f32x4_t w = {dv[0], dv[1], dv[2], dv[3]};
      *reinterpret_cast<f32x4_t*>(dqkv.p[nl - 1] + (row0 + k) * ROWLD + 2 * DM + h * DK + 4 * hh) = w;
      for (int j = 0; j < nl; ++j) {
        float* g = dqkv.p[j] + (row0 + k) * ROWLD + DM + h * DK + 4 * hh;
        f32x4_t u = {0.0f, 0.0f, 0.0f, 0.0f};
        if (accumulate) u = *reinterpret_cast<f32x4_t*>(g);
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] += LN2 * (j == 0 ? dkc[e] : j == 1 ? dkc[4 + e] : dkc[8 + e]);   // Q^T carried scale * log2 e
        *reinterpret_cast<f32x4_t*>(g) = u;
      }
    }
  }
}

}  // namespace

extern "C" {

int dst_spec_attn_flash_fwd(const float* qkv0, const float* qkv1, const float* qkv2, int32_t n_layers, float* stats, float* out, int32_t B,
                            int32_t L, int32_t H, int32_t dk, float scale, void* stream) {
  if (n_layers < 1 || n_layers > 3 || !qkv0 || (n_layers > 1 && !qkv1) || (n_layers > 2 && !qkv2) || !stats || !out || B <= 0 || L <= 0 || L > 512 ||
      H * dk != DM || dk != DK)
    return DS_ERR_ARG;
  const int NT = (L + 31) / 32, LP = NT * 32;
  const size_t lds = (size_t)(LP * KLD + NT * 4 * DK * 8) * 2;
  QkvPtrs q{{qkv0, qkv1, qkv2}};
  hipLaunchKernelGGL(k_sfa_fwd<SFA_NW_F>, dim3(B * H), dim3(SFA_NW_F * 64), lds, (hipStream_t)stream, q, (int)n_layers, stats, out, (int)L, (int)H, scale);
  return DST_CHECK_LAUNCH();
}

int dst_spec_attn_flash_bwd(const float* qkv0, const float* qkv1, const float* qkv2, int32_t n_layers, const float* stats, const float* out,
                            const float* dout, float* dqkv0, float* dqkv1, float* dqkv2, int32_t B, int32_t L, int32_t H, int32_t dk, float scale,
                            int32_t part, int32_t accumulate, void* stream) {
  if (n_layers < 1 || n_layers > 3 || !qkv0 || !dqkv0 || (n_layers > 1 && (!qkv1 || !dqkv1)) || (n_layers > 2 && (!qkv2 || !dqkv2)) || !stats || !out ||
      !dout || B <= 0 || L <= 0 || L > 512 || H * dk != DM || dk != DK)
    return DS_ERR_ARG;
  const int NT = (L + 31) / 32, LP = NT * 32;
  QkvPtrs q{{qkv0, qkv1, qkv2}};
  GradPtrs g{{dqkv0, dqkv1, dqkv2}};
  hipStream_t s = (hipStream_t)stream;
  const size_t lds_q = (size_t)(LP * KLD + LP * DK + NT * 4 * 32 * 8) * 2;
  const size_t lds_kv = (size_t)(LP * KLD + NT * 4 * 32 * 8 + NT * 4 * DK * 8) * 2 + (size_t)LP * 4;
  if (lds_kv > 64 * 1024) return DS_ERR_ARG;
  if (part < 0 || part > 2) return DS_ERR_ARG;
  if (part != 2) hipLaunchKernelGGL(k_sfa_bwd_q<SFA_NW_Q>, dim3(B * H), dim3(SFA_NW_Q * 64), lds_q, s, q, (int)n_layers, stats, out, dout, g, (int)L, (int)H, scale, (int)(accumulate != 0));
  if (part != 1) hipLaunchKernelGGL(k_sfa_bwd_kv<SFA_NW_K>, dim3(B * H), dim3(SFA_NW_K * 64), lds_kv, s, q, (int)n_layers, stats, out, dout, g, (int)L, (int)H, scale, (int)(accumulate != 0));
  return DST_CHECK_LAUNCH();
}

}  // extern "C"
