// diffspectra_amd — gfx950 kernels of the TRAINING path (include/diffspectra_train.h): forward and hand-written backward of every
// operation of the DMT graph over the packed-ragged layout.  fp32 storage and arithmetic, GEMMs on v_mfma_f32_32x32x2_f32.
// Stage A of row N1: correctness against the reference's autograd first; one workgroup per molecule for everything that
// reduces over a molecule's rows (adaLN gradients, attention, coordinate update), fixed summation orders everywhere (no float
// atomics), so a step is reproducible bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/diffspectra_hip.h"
#include "../../include/diffspectra_train.h"
#include "ds_train_common.h"

typedef float f32x16_t __attribute__((ext_vector_type(16)));

namespace {

#ifndef DST_SPEC_SLICES
#define DST_SPEC_SLICES 2   // workgroups per (batch, head) in the SpecFormer attention kernels: each stages K and V of the head once
#endif
#define DST_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DS_OK : DS_ERR_LAUNCH)

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int pair_index(int n, int lo, int hi) { return lo * (2 * n - lo - 1) / 2 + (hi - lo - 1); }

// ------------------------------------------------------------------------------------------------------------------ colsum / sumsq
__global__ __launch_bounds__(256) void k_colsum_partial(const float* __restrict__ X, int64_t ld, int R, int C, float* __restrict__ partial,
                                                         int rows_per_chunk) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s = 0.0f;
  if (col < C)
    for (int r = r0 + rl; r < r1; r += 4) s += X[(int64_t)r * ld + col];
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < C) partial[(int64_t)blockIdx.y * C + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
__global__ void k_colsum_final(const float* __restrict__ partial, int chunks, int C, float* __restrict__ out, int accumulate) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= C) return;
  float s = 0.0f;
  for (int k = 0; k < chunks; ++k) s += partial[(int64_t)k * C + col];
  out[col] = accumulate ? out[col] + s : s;
}
__global__ __launch_bounds__(256) void k_sumsq_partial(const float* __restrict__ x, int64_t n, float* __restrict__ partial) {
  __shared__ float red[4];
  float s = 0.0f;
  if ((reinterpret_cast<uintptr_t>(x) & 15) == 0) {        // 16-byte loads, four independent partial sums per thread (the scalar grid-stride loop
    const int64_t n4 = n >> 2;                             // was a chain of ~140 dependent 4-byte loads: 180 us for 37 MB)
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
      const float4 v = reinterpret_cast<const float4*>(x)[i];
      s0 += v.x * v.x; s1 += v.y * v.y; s2 += v.z * v.z; s3 += v.w * v.w;
    }
    s = (s0 + s1) + (s2 + s3);
    if (blockIdx.x == 0 && threadIdx.x < (int)(n & 3)) { const float t = x[(n4 << 2) + threadIdx.x]; s += t * t; }
  } else {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) s += x[i] * x[i];
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------------------------ elementwise
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
__global__ void k_act_fwd(const float* __restrict__ x, float* __restrict__ y, int64_t n, int kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = x[i];
  float r;
  if (kind == 1) r = v * sigmoidf_(v);
  else if (kind == 2) r = 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));
  else r = tanhf(v);
  y[i] = r;
}
__global__ void k_act_bwd(const float* __restrict__ dy, const float* __restrict__ ref, float* __restrict__ dx, int64_t n, int kind) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float v = ref[i];
  float d;
  if (kind == 1) { const float s = sigmoidf_(v); d = s * (1.0f + v * (1.0f - s)); }
  else if (kind == 2) d = 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * expf(-0.5f * v * v) * 0.39894228040143267794f;
  else d = 1.0f - v * v;
  dx[i] = dy[i] * d;
}
__global__ void k_axpy(float a, const float* __restrict__ x, float* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += a * x[i];
}
__global__ void k_axpy4(float a, const float4* __restrict__ x, float4* __restrict__ y, int64_t n4) {   // 16-byte form (aligned, n % 4 == 0)
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float4 u = x[i];
  float4 v = y[i];
  v.x += a * u.x; v.y += a * u.y; v.z += a * u.z; v.w += a * u.w;
  y[i] = v;
}

// adjacency bits of the self-conditioning prediction (dmt.py:338-340,361): bit 0 = cond edge channel 0 >= edge_quan_th, bit 1 = cond d^2 <= cut-off
__global__ void k_adj_bits(const float* __restrict__ cond_e, int64_t ld, const float* __restrict__ d2c, float th, float cutoff, int n, int32_t* __restrict__ adj) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) adj[p] = (cond_e[(int64_t)p * ld] >= th ? 1 : 0) | (d2c[p] <= cutoff ? 2 : 0);
}

// many small strided 2-D copies in one launch (the per-step concatenated weight buffers and the scatter of their gradients)
__global__ __launch_bounds__(256) void k_copy_pieces(const dst_piece* __restrict__ table) {
  const dst_piece pc = table[blockIdx.x];
  const unsigned int rows = (unsigned int)pc.rows, cols = (unsigned int)pc.cols;
  const bool vec = ((cols | (unsigned int)pc.dst_ld | (unsigned int)pc.src_ld) & 3u) == 0 &&
                   (((uintptr_t)pc.dst | (uintptr_t)pc.src) & 15u) == 0;
  if (vec) {                                                      // 16-byte pieces; 32-bit index arithmetic (a piece is < 2^31 elements, checked on the host)
    const unsigned int c4 = cols >> 2, total = rows * c4;
    for (unsigned int i = blockIdx.y * 256u + threadIdx.x; i < total; i += gridDim.y * 256u) {
      const unsigned int r = i / c4, c = (i - r * c4) << 2;
      *reinterpret_cast<float4*>(pc.dst + (int64_t)r * pc.dst_ld + c) = *reinterpret_cast<const float4*>(pc.src + (int64_t)r * pc.src_ld + c);
    }
  } else {
    const unsigned int total = rows * cols;
    for (unsigned int i = blockIdx.y * 256u + threadIdx.x; i < total; i += gridDim.y * 256u) {
      const unsigned int r = i / cols, c = i - r * cols;
      pc.dst[(int64_t)r * pc.dst_ld + c] = pc.src[(int64_t)r * pc.src_ld + c];
    }
  }
}

// the same pieces rounded to bf16 (round to nearest even, as the GEMM kernels round their operands): dst is a bf16 buffer, dst_ld in
// bf16 elements.  The fused row chains (ds_train_chain.hip) stream their weights from L2 once per 32-row tile: as bf16 that stream is half
// as wide and needs no conversion in the kernel.
__global__ __launch_bounds__(256) void k_pack_bf16_pieces(const dst_piece* __restrict__ table) {
  const dst_piece pc = table[blockIdx.x];
  const unsigned int rows = (unsigned int)pc.rows, cols = (unsigned int)pc.cols, total = rows * cols;
  unsigned short* dst = reinterpret_cast<unsigned short*>(pc.dst);
  const bool transposed = pc.dst_ld < 0;                         // dst[c * (-dst_ld) + r]: the weight as the input-gradient products read it
  const int64_t dld = transposed ? -pc.dst_ld : pc.dst_ld;
  for (unsigned int i = blockIdx.y * 256u + threadIdx.x; i < total; i += gridDim.y * 256u) {
    const unsigned int r = i / cols, c = i - r * cols;
    const __bf16 h = (__bf16)pc.src[(int64_t)r * pc.src_ld + c];
    dst[transposed ? (int64_t)c * dld + r : (int64_t)r * dld + c] = __builtin_bit_cast(unsigned short, h);
  }
}

// ------------------------------------------------------------------------------------------------------------------ dropout
// nn.Dropout(p) in training mode (dmt.py:114-120): y = x * keep / (1 - p) with keep ~ Bernoulli(1 - p) from a counter-based Philox4x32-10
// stream keyed on (seed, stream id): the mask of element i is a pure function of (seed, stream, i), so the backward pass re-creates it
// instead of storing it.  (The reference's masks come from torch's generator; only the distribution can agree.)
__device__ __forceinline__ void philox_round(unsigned int (&c)[4], unsigned int k0, unsigned int k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned int n0 = (unsigned int)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned int)p1, n2 = (unsigned int)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned int)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ void k_dropout(const float* __restrict__ x, float* __restrict__ y, int64_t n, float p, float scale, unsigned long long seed,
                          unsigned int stream_id) {
  const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;        // one Philox block = 4 elements
  if (q * 4 >= n) return;
  unsigned int c[4] = {(unsigned int)q, (unsigned int)(q >> 32), stream_id, 0x44524f50u};
  unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    philox_round(c, k0, k1);
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  const unsigned int thr = (unsigned int)fminf(p * 4294967296.0f, 4294967040.0f);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int64_t i = q * 4 + j;
    if (i < n) y[i] = (c[j] >= thr) ? x[i] * scale : 0.0f;
  }
}

// ------------------------------------------------------------------------------------------------------------------ LN + modulate
// The molecule-level row kernels (LayerNorm + modulate, gated residual): one 1024-thread workgroup per molecule - its rows are
// contiguous and the adaLN gradients are sums over exactly those rows, reduced here in a fixed order.  A lane owns four consecutive
// columns (16-byte accesses): a 256-wide row is one wave, four 64-wide rows share a wave (16 lanes each, 16-lane reductions).
// Round 3 ran these with 256 threads and 4-byte accesses: one workgroup per CU at four waves left the memory pipe idle (60 - 190 us
// per launch on the directed rows; the rows of 256 molecules are 250 MB).
constexpr int MOLW = 16;                                   // waves per molecule workgroup
typedef float f4_t __attribute__((ext_vector_type(4)));
template <int G>
__device__ __forceinline__ float group_sum(float v) {      // sum over aligned groups of G lanes
#pragma unroll
  for (int o = G / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ f4_t ld4(const float* p) { return *reinterpret_cast<const f4_t*>(p); }
__device__ __forceinline__ void st4(float* p, f4_t v) { *reinterpret_cast<f4_t*>(p) = v; }

template <int C>
__global__ __launch_bounds__(1024) void k_lnmod_fwd(const float* __restrict__ x, const int32_t* __restrict__ seg_off, int seg_mul,
                                                     const float* __restrict__ ada, int64_t ada_ld, int shift_off, int scale_off,
                                                     float* __restrict__ y, float* __restrict__ stats) {
  constexpr int LPR = C / 4, RPW = 64 / LPR;               // lanes per row, rows per wave
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPR, cl = (lane % LPR) * 4;
  const int r0 = seg_off[m] * seg_mul, r1 = seg_off[m + 1] * seg_mul;
  const f4_t sh = ld4(ada + (int64_t)m * ada_ld + shift_off + cl), sc = ld4(ada + (int64_t)m * ada_ld + scale_off + cl);
  // gridDim.y workgroups share a molecule's rows (interleaved passes): one workgroup per molecule lasts as long as the largest molecule
  // (812 directed rows against a mean of ~310), four per molecule let the dispatcher even the CUs out
  for (int r = r0 + (blockIdx.y * MOLW + wave) * RPW + sub; r < r1; r += gridDim.y * MOLW * RPW) {
    const f4_t v = ld4(x + (int64_t)r * C + cl);
    const float mean = group_sum<LPR>((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / C);
    const f4_t d = v - mean;
    const float rstd = 1.0f / sqrtf(group_sum<LPR>((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / C) + 1e-6f);
    st4(y + (int64_t)r * C + cl, (d * rstd) * (1.0f + sc) + sh);
    if (lane % LPR == 0) { stats[(int64_t)r * 2] = mean; stats[(int64_t)r * 2 + 1] = rstd; }
  }
}

template <int C>
__global__ __launch_bounds__(1024) void k_lnmod_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats,
                                                     const int32_t* __restrict__ seg_off, int seg_mul, const float* __restrict__ ada,
                                                     float* __restrict__ d_ada, int64_t ada_ld, int shift_off, int scale_off,
                                                     float* __restrict__ dx, int accumulate, float* __restrict__ part) {
  constexpr int LPR = C / 4, RPW = 64 / LPR, SLOTS = MOLW * RPW;
  __shared__ float red[2][SLOTS][C];
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPR, cl = (lane % LPR) * 4;
  const int r0 = seg_off[m] * seg_mul, r1 = seg_off[m + 1] * seg_mul;
  const f4_t sc1 = ld4(ada + (int64_t)m * ada_ld + scale_off + cl) + 1.0f;
  f4_t dsh = {0.0f, 0.0f, 0.0f, 0.0f}, dsc = {0.0f, 0.0f, 0.0f, 0.0f};
  // the next row's operands are requested (from a clamped row index) before this row is worked on: one row per wave and pass is a chain of
  // dependent round trips otherwise
  const int rstep = gridDim.y * MOLW * RPW;
  int r = r0 + (blockIdx.y * MOLW + wave) * RPW + sub;
  int rc = min(r, r1 - 1);
  float mean_n = 0.0f, rstd_n = 0.0f;
  f4_t g_n = {0.0f, 0.0f, 0.0f, 0.0f}, x_n = g_n;
  if (r1 > r0) { mean_n = stats[(int64_t)rc * 2]; rstd_n = stats[(int64_t)rc * 2 + 1]; g_n = ld4(dy + (int64_t)rc * C + cl); x_n = ld4(x + (int64_t)rc * C + cl); }
  for (; r < r1; r += rstep) {
    const float mean = mean_n, rstd = rstd_n;
    const f4_t g = g_n;
    const f4_t xh = (x_n - mean) * rstd;
    rc = min(r + rstep, r1 - 1);
    mean_n = stats[(int64_t)rc * 2]; rstd_n = stats[(int64_t)rc * 2 + 1];
    g_n = ld4(dy + (int64_t)rc * C + cl); x_n = ld4(x + (int64_t)rc * C + cl);
    dsh += g;
    dsc += g * xh;
    const f4_t gg = g * sc1, gx = gg * xh;
    const float m1 = group_sum<LPR>((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.0f / C);
    const float m2 = group_sum<LPR>((gx[0] + gx[1]) + (gx[2] + gx[3])) * (1.0f / C);
    f4_t d = rstd * (gg - m1 - xh * m2);
    float* o = dx + (int64_t)r * C + cl;
    if (accumulate) d += ld4(o);
    st4(o, d);
  }
  st4(&red[0][wave * RPW + sub][cl], dsh);
  st4(&red[1][wave * RPW + sub][cl], dsc);
  __syncthreads();
  for (int c = threadIdx.x; c < 2 * C; c += MOLW * 64) {
    const int w = c / C, cc = c % C;
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) t += red[w][k][cc];
    // split rows: this workgroup's share goes to part[split][molecule][2 C]; k_lnmod_bwd_finish adds the shares in split order
    if (gridDim.y > 1) part[((int64_t)blockIdx.y * gridDim.x + m) * (2 * C) + c] = t;
    else d_ada[(int64_t)m * ada_ld + (w ? scale_off : shift_off) + cc] = t;
  }
}
template <int C>
__global__ __launch_bounds__(2 * C) void k_lnmod_bwd_finish(const float* __restrict__ part, int splits, int B, float* __restrict__ d_ada, int64_t ada_ld,
                                                           int shift_off, int scale_off) {
  const int m = blockIdx.x, c = threadIdx.x, w = c / C, cc = c % C;
  float t = 0.0f;
  for (int k = 0; k < splits; ++k) t += part[((int64_t)k * B + m) * (2 * C) + c];
  d_ada[(int64_t)m * ada_ld + (w ? scale_off : shift_off) + cc] = t;
}

// ------------------------------------------------------------------------------------------------------------------ gated residual
template <int C>
__global__ __launch_bounds__(1024) void k_gate_add_fwd(const float* __restrict__ r_, const float* __restrict__ z, const int32_t* __restrict__ seg_off,
                                                        int seg_mul, const float* __restrict__ ada, int64_t ada_ld, int gate_off,
                                                        float* __restrict__ out) {
  constexpr int LPR = C / 4, RPW = 64 / LPR;
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPR, cl = (lane % LPR) * 4;
  const int r0 = seg_off[m] * seg_mul, r1 = seg_off[m + 1] * seg_mul;
  const f4_t g = ld4(ada + (int64_t)m * ada_ld + gate_off + cl);
  for (int r = r0 + wave * RPW + sub; r < r1; r += MOLW * RPW) {
    const int64_t i = (int64_t)r * C + cl;
    st4(out + i, ld4(r_ + i) + g * ld4(z + i));
  }
}
// dz = gate * dout, optionally times the dropout mask of the tensor z was (drop_p > 0: z = drop(.) in the forward, dmt.py:116,120 - the
// mask of element (r, c) is Philox block (r * C + c) / 4 of stream (seed, stream_id), the four columns of a lane are one block)
template <int C>
__global__ __launch_bounds__(1024) void k_gate_add_bwd(const float* __restrict__ dout, const float* __restrict__ z, const int32_t* __restrict__ seg_off,
                                                        int seg_mul, const float* __restrict__ ada, float* __restrict__ d_ada, int64_t ada_ld,
                                                        int gate_off, float* __restrict__ dr, int accumulate_r, float* __restrict__ dz, float drop_p,
                                                        unsigned long long drop_seed, unsigned int drop_stream) {
  constexpr int LPR = C / 4, RPW = 64 / LPR, SLOTS = MOLW * RPW;
  __shared__ float red[SLOTS][C];
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int sub = lane / LPR, cl = (lane % LPR) * 4;
  const int r0 = seg_off[m] * seg_mul, r1 = seg_off[m + 1] * seg_mul;
  const f4_t g = ld4(ada + (int64_t)m * ada_ld + gate_off + cl);
  const unsigned int thr = dst::dropout_threshold(drop_p);
  const float keep_scale = 1.0f / (1.0f - drop_p);
  f4_t dg = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int r = r0 + wave * RPW + sub; r < r1; r += MOLW * RPW) {
    const int64_t i = (int64_t)r * C + cl;
    const f4_t d = ld4(dout + i);
    dg += d * ld4(z + i);
    f4_t o = g * d;
    if (drop_p > 0.0f) {
      unsigned int w[4];
      dst::dropout_block(drop_seed, drop_stream, i >> 2, w);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = w[e] >= thr ? o[e] * keep_scale : 0.0f;
    }
    st4(dz + i, o);
    if (dr) st4(dr + i, accumulate_r ? ld4(dr + i) + d : d);
  }
  st4(&red[wave * RPW + sub][cl], dg);
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += MOLW * 64) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < SLOTS; ++k) t += red[k][c];
    d_ada[(int64_t)m * ada_ld + gate_off + c] = t;
  }
}

// ------------------------------------------------------------------------------------------------------------------ geometry
// local pair tables of one molecule in LDS
__device__ __forceinline__ void fill_pair_tables(int n, unsigned char* pa, unsigned char* pb) {
  for (int a = threadIdx.x; a < n; a += blockDim.x)
    for (int b = a + 1; b < n; ++b) {
      const int idx = pair_index(n, a, b);
      pa[idx] = (unsigned char)a;
      pb[idx] = (unsigned char)b;
    }
}

#define DST_GAUSS_A 2.50662732f /* fp32((2 * 3.14159) ** 0.5): the Python-float constant of layers.py:293-294 as torch applies it */

// (1024 threads per molecule: with 256 a 29-atom molecule took 100 dependent iterations per wave, and one workgroup per CU at four waves
// left the memory pipe idle)
constexpr int GEOM_NT = 1024, GEOM_NW = GEOM_NT / 64;
__global__ __launch_bounds__(GEOM_NT) void k_geom_fwd(dst_layout L, const float* __restrict__ pos, const float* __restrict__ ada, int64_t ada_ld,
                                                   int dist_off, const float* __restrict__ means, const float* __restrict__ stds,
                                                   float* __restrict__ X, int64_t ldx, float* __restrict__ xs, float* __restrict__ d2s) {
  __shared__ unsigned char pa[406], pb[406];
  __shared__ float sp[29][3];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  fill_pair_tables(n, pa, pb);
  for (int i = threadIdx.x; i < n * 3; i += GEOM_NT) sp[i / 3][i % 3] = pos[(int64_t)(n0 + i / 3) * 3 + i % 3];
  __syncthreads();
  const float a = DST_GAUSS_A;
  const float dsc = ada[(int64_t)m * ada_ld + dist_off], dsh = ada[(int64_t)m * ada_ld + dist_off + 1];
  for (int it = threadIdx.x; it < np * 64; it += GEOM_NT) {
    const int p = it >> 6, k = it & 63;
    const int ia = pa[p], ib = pb[p];
    const float dx = sp[ia][0] - sp[ib][0], dy = sp[ia][1] - sp[ib][1], dz = sp[ia][2] - sp[ib][2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    const float x = d2 * (dsc + 1.0f) + dsh;
    float f;
    if (k == 0) {
      f = x;
      xs[p0 + p] = x;
      d2s[p0 + p] = d2;
    } else {
      const float sd = fabsf(stds[k - 1]) + 1e-5f;
      const float u = (x - means[k - 1]) / sd;
      f = expf(-0.5f * (u * u)) / (a * sd);
    }
    X[(int64_t)(p0 + p) * ldx + k] = f;
  }
}

__global__ __launch_bounds__(GEOM_NT) void k_geom_bwd(dst_layout L, const float* __restrict__ pos, const float* __restrict__ ada, float* __restrict__ d_ada,
                                                   int64_t ada_ld, int dist_off, const float* __restrict__ means, const float* __restrict__ stds,
                                                   const float* __restrict__ xs, const float* __restrict__ d2s, const float* __restrict__ g1, int64_t ld1,
                                                   const float* __restrict__ g2, int64_t ld2, float* __restrict__ dms, float* __restrict__ dd2,
                                                   float* __restrict__ dpos) {
  __shared__ float red[GEOM_NW][130];
  __shared__ float sp[29][3];
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  for (int i = threadIdx.x; i < n * 3; i += GEOM_NT) sp[i / 3][i % 3] = pos[(int64_t)(n0 + i / 3) * 3 + i % 3];
  const float a = DST_GAUSS_A;
  const float dsc = ada[(int64_t)m * ada_ld + dist_off];
  float mu = 0.0f, sraw = 1.0f, sd = 1.0f;
  if (lane > 0) { mu = means[lane - 1]; sraw = stds[lane - 1]; sd = fabsf(sraw) + 1e-5f; }
  float dmu = 0.0f, dsd = 0.0f, a_dsc = 0.0f, a_dsh = 0.0f;
  for (int p = wave; p < np; p += GEOM_NW) {                       // one wave per pair, lane = feature
    const float x = xs[p0 + p];
    float g = g1[(int64_t)(p0 + p) * ld1 + lane];
    if (g2) g += g2[(int64_t)(p0 + p) * ld2 + lane];
    float dxl;
    if (lane == 0) {
      dxl = g;
    } else {
      const float u = (x - mu) / sd;
      const float G = expf(-0.5f * (u * u)) / (a * sd);
      const float t = g * G;
      dxl = -t * u / sd;
      dmu += t * u / sd;
      dsd += t * (u * u - 1.0f) / sd;
    }
    const float dx = wave_sum(dxl);
    if (lane == 0) {
      const float d2 = d2s[p0 + p];
      a_dsc += dx * d2;
      a_dsh += dx;
      dd2[p0 + p] = dx * (1.0f + dsc);
    }
  }
  red[wave][lane] = dmu;
  red[wave][64 + lane] = dsd * (sraw < 0.0f ? -1.0f : 1.0f);
  if (lane == 0) { red[wave][128] = a_dsc; red[wave][129] = a_dsh; }
  __syncthreads();
  if (threadIdx.x < 130) {                                 // the waves' shares in wave order
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < GEOM_NW; ++w) t += red[w][threadIdx.x];
    if (threadIdx.x < 128) dms[(int64_t)m * 128 + threadIdx.x] = t;
    else d_ada[(int64_t)m * ada_ld + dist_off + threadIdx.x - 128] = t;
  }
  if (dpos) {                                              // d d2 / d pos, fixed partner order
    for (int it = threadIdx.x; it < n * 3; it += GEOM_NT) {
      const int i = it / 3, c = it % 3;
      float s = 0.0f;
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        const int p = p0 + (i < j ? pair_index(n, i, j) : pair_index(n, j, i));
        s += 2.0f * dd2[p] * (sp[i][c] - sp[j][c]);
      }
      dpos[(int64_t)(n0 + i) * 3 + c] += s;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------ attention
// logits / alpha scratch in LDS: [directed edge (2 * 406)][16 heads]
__global__ __launch_bounds__(1024) void k_attn_fwd(dst_layout L, const float* __restrict__ qkv, const float* __restrict__ te0, const float* __restrict__ te1,
                                                   int64_t ldt, const int32_t* __restrict__ adj, float* __restrict__ out, float* __restrict__ alpha) {
  __shared__ float lg[812 * 16];
  __shared__ unsigned char pa[406], pb[406];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  fill_pair_tables(n, pa, pb);
  __syncthreads();
  // logits: directed edge d = 2p + dir; dir 0: source a -> target b, dir 1: source b -> target a
  for (int it = threadIdx.x; it < np * 2 * 16; it += blockDim.x) {
    const int d = it >> 4, hd = it & 15, p = d >> 1, dir = d & 1;
    const int src = dir ? pb[p] : pa[p], tgt = dir ? pa[p] : pb[p];
    float v;
    if (hd < 2) {
      v = ((adj[p0 + p] >> hd) & 1) ? 1.0f : -1e10f;
    } else {
      const int c0 = (hd - 2) * 18;                       // 72-byte head slices: 8-byte aligned, nine float2 each
      typedef float f2_t __attribute__((ext_vector_type(2)));
      const f2_t* q = reinterpret_cast<const f2_t*>(qkv + (int64_t)(n0 + tgt) * 768 + c0);
      const f2_t* k = reinterpret_cast<const f2_t*>(qkv + (int64_t)(n0 + src) * 768 + 256 + c0);
      const f2_t* e = reinterpret_cast<const f2_t*>(te0 + (int64_t)(p0 + p) * ldt + c0);
      float s = 0.0f;
#pragma unroll
      for (int c = 0; c < 9; ++c) {                       // same summation order as the scalar loop (c ascending)
        const f2_t qq = q[c], kk = k[c], ee = e[c];
        s += qq[0] * kk[0] * ee[0];
        s += qq[1] * kk[1] * ee[1];
      }
      v = s / 4.0f;
    }
    lg[it] = v;
  }
  __syncthreads();
  // softmax over the sources of every (target, head)
  for (int it = threadIdx.x; it < n * 16; it += blockDim.x) {
    const int t = it >> 4, hd = it & 15;
    float mx = -INFINITY;
    for (int s = 0; s < n; ++s) {
      if (s == t) continue;
      const int d = s < t ? 2 * pair_index(n, s, t) : 2 * pair_index(n, t, s) + 1;   // s<t: a=s -> b=t (dir 0); s>t: b=s -> a=t (dir 1)
      mx = fmaxf(mx, lg[d * 16 + hd]);
    }
    float den = 0.0f;
    for (int s = 0; s < n; ++s) {
      if (s == t) continue;
      const int d = s < t ? 2 * pair_index(n, s, t) : 2 * pair_index(n, t, s) + 1;
      const float ex = expf(lg[d * 16 + hd] - mx);
      lg[d * 16 + hd] = ex;
      den += ex;
    }
    den += 1e-16f;
    for (int s = 0; s < n; ++s) {
      if (s == t) continue;
      const int d = s < t ? 2 * pair_index(n, s, t) : 2 * pair_index(n, t, s) + 1;
      const float al = lg[d * 16 + hd] / den;
      lg[d * 16 + hd] = al;
      alpha[(int64_t)(2 * p0 + d) * 16 + hd] = al;
    }
  }
  __syncthreads();
  // aggregation onto the target, ascending source order
  for (int it = threadIdx.x; it < n * 64; it += blockDim.x) {      // four columns (one head) per thread
    const int t = it >> 6, col = (it & 63) * 4, hd = col >> 4;
    f4_t s = {0.0f, 0.0f, 0.0f, 0.0f};
    // sources below and above the target as two branch-free ranges (same ascending order): a `continue` in the loop kept the compiler
    // from putting more than one iteration's loads in flight
#pragma unroll 4
    for (int sN = 0; sN < t; ++sN) {
      const int p = pair_index(n, sN, t);
      s += ld4(qkv + (int64_t)(n0 + sN) * 768 + 512 + col) * ld4(te1 + (int64_t)(p0 + p) * ldt + col) * lg[(2 * p) * 16 + hd];
    }
#pragma unroll 4
    for (int sN = t + 1; sN < n; ++sN) {
      const int p = pair_index(n, t, sN);
      s += ld4(qkv + (int64_t)(n0 + sN) * 768 + 512 + col) * ld4(te1 + (int64_t)(p0 + p) * ldt + col) * lg[(2 * p + 1) * 16 + hd];
    }
    st4(out + (int64_t)(n0 + t) * 256 + col, s);
  }
  if (n == 1)
    for (int col = threadIdx.x; col < 256; col += blockDim.x) out[(int64_t)n0 * 256 + col] = 0.0f;
}

// MODE 0: the whole backward in one workgroup per molecule.  MODE 1 + MODE 2: the same in two launches - d logit (phases 1, 2) written to
// `dlg` [2 Pp, 16], then the node- and pair-side gradients (phases 3, 4: ~80 % of the work) with gridDim.y workgroups sharing a molecule's
// items: one workgroup per molecule lasts as long as the largest molecule (29 atoms: 2.6x the mean work), the shares let the dispatcher even
// the CUs out.  alpha is staged in LDS next to d logit (dynamic LDS, 104 kB): the softmax backward read it from global memory inside a
// dependent loop.  Same arithmetic and summation order in every mode.
template <int MODE>
__global__ __launch_bounds__(1024) void k_attn_bwd(dst_layout L, const float* __restrict__ qkv, const float* __restrict__ te0, const float* __restrict__ te1,
                                                   int64_t ldt, const float* __restrict__ alpha, const float* __restrict__ dout, float* __restrict__ dqkv,
                                                   float* __restrict__ dte0, float* __restrict__ dte1, int te_tanh, float* __restrict__ dlg) {
  extern __shared__ __attribute__((aligned(16))) float attn_smem[];
  float* dl = attn_smem;                  // [812 * 16] d alpha, then d logit
  // alpha of this molecule: an LDS copy where the softmax backward walks it (MODE 0, 1); the gradient phases use each value once, so MODE 2
  // reads it in place and keeps its LDS at 52 kB (three workgroups per CU instead of one)
  float* als = attn_smem + 812 * 16;
  const float* __restrict__ al = MODE == 2 ? alpha + (int64_t)2 * L.pair_off[blockIdx.x] * 16 : als;
  __shared__ unsigned char pa[406], pb[406];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  const int tid0 = threadIdx.x + (MODE == 2 ? 1024 * blockIdx.y : 0), tstep = 1024 * (MODE == 2 ? gridDim.y : 1);
  fill_pair_tables(n, pa, pb);
  if (MODE != 2)
    for (int it = threadIdx.x; it < np * 32; it += blockDim.x) als[it] = alpha[(int64_t)2 * p0 * 16 + it];
  if (MODE == 2)
    for (int it = threadIdx.x; it < np * 32; it += blockDim.x) dl[it] = dlg[(int64_t)2 * p0 * 16 + it];
  __syncthreads();
  if (MODE != 2) {
  // d alpha[d, hd] = sum_c dout[tgt, hd, c] v[src, hd, c] te1[p, hd, c]
  for (int it = threadIdx.x; it < np * 32; it += blockDim.x) {
    const int d = it >> 4, hd = it & 15, p = d >> 1, dir = d & 1;
    const int src = dir ? pb[p] : pa[p], tgt = dir ? pa[p] : pb[p];
    const float* go = dout + (int64_t)(n0 + tgt) * 256 + hd * 16;
    const float* v = qkv + (int64_t)(n0 + src) * 768 + 512 + hd * 16;
    const float* e = te1 + (int64_t)(p0 + p) * ldt + hd * 16;
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; c += 4) {
      const f4_t a = ld4(go + c) * ld4(v + c) * ld4(e + c);
      s += a[0]; s += a[1]; s += a[2]; s += a[3];
    }
    dl[it] = s;
  }
  __syncthreads();
  // softmax backward per (target, head): dlogit = alpha (dalpha - sum alpha dalpha); sources below / above the target as two ranges
  for (int it = threadIdx.x; it < n * 16; it += blockDim.x) {
    const int t = it >> 4, hd = it & 15;
    float dot = 0.0f;
    for (int s = 0; s < t; ++s) { const int d = 2 * pair_index(n, s, t); dot += al[d * 16 + hd] * dl[d * 16 + hd]; }
    for (int s = t + 1; s < n; ++s) { const int d = 2 * pair_index(n, t, s) + 1; dot += al[d * 16 + hd] * dl[d * 16 + hd]; }
    for (int s = 0; s < t; ++s) { const int d = 2 * pair_index(n, s, t); dl[d * 16 + hd] = al[d * 16 + hd] * (dl[d * 16 + hd] - dot); }
    for (int s = t + 1; s < n; ++s) { const int d = 2 * pair_index(n, t, s) + 1; dl[d * 16 + hd] = al[d * 16 + hd] * (dl[d * 16 + hd] - dot); }
  }
  __syncthreads();
  if (MODE == 1) {
    for (int it = threadIdx.x; it < np * 32; it += blockDim.x) dlg[(int64_t)2 * p0 * 16 + it] = dl[it];
    return;
  }
  }   // MODE != 2
  // node-side gradients: thread per (node, four columns of the 768-wide q|k|v row) - 16-byte accesses, a quarter of the index arithmetic
  for (int it = tid0; it < n * 192; it += tstep) {
    const int i = it / 192, col = (it % 192) * 4;
    f4_t s = {0.0f, 0.0f, 0.0f, 0.0f};
    if (col < 252) {                                   // dq[i]: i is the target (252 = 4 * 63: a quad never straddles the padding)
      int hd[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hd[e] = (col + e) / 18 + 2;
      // (the partners below and above i as two branch-free ranges, ascending as before: several iterations' loads in flight)
#pragma unroll 4
      for (int j = 0; j < i; ++j) {
        const int p = pair_index(n, j, i), d = 2 * p;  // source j -> target i
        const f4_t w = {dl[d * 16 + hd[0]], dl[d * 16 + hd[1]], dl[d * 16 + hd[2]], dl[d * 16 + hd[3]]};
        s += w * ld4(qkv + (int64_t)(n0 + j) * 768 + 256 + col) * ld4(te0 + (int64_t)(p0 + p) * ldt + col);
      }
#pragma unroll 4
      for (int j = i + 1; j < n; ++j) {
        const int p = pair_index(n, i, j), d = 2 * p + 1;
        const f4_t w = {dl[d * 16 + hd[0]], dl[d * 16 + hd[1]], dl[d * 16 + hd[2]], dl[d * 16 + hd[3]]};
        s += w * ld4(qkv + (int64_t)(n0 + j) * 768 + 256 + col) * ld4(te0 + (int64_t)(p0 + p) * ldt + col);
      }
      s *= 0.25f;
    } else if (col >= 256 && col < 508) {              // dk[i]: i is the source
      const int c = col - 256;
      int hd[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) hd[e] = (c + e) / 18 + 2;
#pragma unroll 4
      for (int t = 0; t < i; ++t) {
        const int p = pair_index(n, t, i), d = 2 * p + 1;   // source i -> target t
        const f4_t w = {dl[d * 16 + hd[0]], dl[d * 16 + hd[1]], dl[d * 16 + hd[2]], dl[d * 16 + hd[3]]};
        s += w * ld4(qkv + (int64_t)(n0 + t) * 768 + c) * ld4(te0 + (int64_t)(p0 + p) * ldt + c);
      }
#pragma unroll 4
      for (int t = i + 1; t < n; ++t) {
        const int p = pair_index(n, i, t), d = 2 * p;
        const f4_t w = {dl[d * 16 + hd[0]], dl[d * 16 + hd[1]], dl[d * 16 + hd[2]], dl[d * 16 + hd[3]]};
        s += w * ld4(qkv + (int64_t)(n0 + t) * 768 + c) * ld4(te0 + (int64_t)(p0 + p) * ldt + c);
      }
      s *= 0.25f;
    } else if (col >= 512) {                           // dv[i]: i is the source
      const int c = col - 512, hd = c >> 4;
#pragma unroll 4
      for (int t = 0; t < i; ++t) {
        const int p = pair_index(n, t, i);
        s += ld4(dout + (int64_t)(n0 + t) * 256 + c) * ld4(te1 + (int64_t)(p0 + p) * ldt + c) * al[(2 * p + 1) * 16 + hd];
      }
#pragma unroll 4
      for (int t = i + 1; t < n; ++t) {
        const int p = pair_index(n, i, t);
        s += ld4(dout + (int64_t)(n0 + t) * 256 + c) * ld4(te1 + (int64_t)(p0 + p) * ldt + c) * al[(2 * p) * 16 + hd];
      }
    }
    st4(dqkv + (int64_t)(n0 + i) * 768 + col, s);
  }
  // pair-side gradients (both directions of a pair), four columns per thread
  for (int it = tid0; it < np * 64; it += tstep) {
    const int p = it >> 6, col = (it & 63) * 4;
    const int a = pa[p], b = pb[p];
    {
      const int hd = col >> 4;
      const f4_t va = ld4(qkv + (int64_t)(n0 + a) * 768 + 512 + col), vb = ld4(qkv + (int64_t)(n0 + b) * 768 + 512 + col);
      f4_t g1 = ld4(dout + (int64_t)(n0 + b) * 256 + col) * va * al[(2 * p) * 16 + hd] + ld4(dout + (int64_t)(n0 + a) * 256 + col) * vb * al[(2 * p + 1) * 16 + hd];
      if (te_tanh) {                                   // te1 = tanh(lin_edge1 e): hand back the gradient in front of the tanh (1 - te^2)
        const f4_t t1 = ld4(te1 + (int64_t)(p0 + p) * ldt + col);
        g1 = g1 * (1.0f - t1 * t1);
      }
      st4(dte1 + (int64_t)(p0 + p) * ldt + col, g1);
    }
    f4_t g0 = {0.0f, 0.0f, 0.0f, 0.0f};
    if (col < 252) {
      const f4_t qa = ld4(qkv + (int64_t)(n0 + a) * 768 + col), qb = ld4(qkv + (int64_t)(n0 + b) * 768 + col);
      const f4_t ka = ld4(qkv + (int64_t)(n0 + a) * 768 + 256 + col), kb = ld4(qkv + (int64_t)(n0 + b) * 768 + 256 + col);
      f4_t w0, w1;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int hd = (col + e) / 18 + 2;
        w0[e] = dl[(2 * p) * 16 + hd];
        w1[e] = dl[(2 * p + 1) * 16 + hd];
      }
      g0 = 0.25f * (w0 * qb * ka + w1 * qa * kb);
      if (te_tanh) {
        const f4_t t0 = ld4(te0 + (int64_t)(p0 + p) * ldt + col);
        g0 = g0 * (1.0f - t0 * t0);
      }
    }
    st4(dte0 + (int64_t)(p0 + p) * ldt + col, g0);
  }
}

// ------------------------------------------------------------------------------------------------------------------ gathers
__global__ __launch_bounds__(256) void k_pair_sum_fwd(dst_layout L, const float* __restrict__ u, int C, const float* __restrict__ bias, float* __restrict__ s) {
  __shared__ unsigned char pa[406], pb[406];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  fill_pair_tables(n, pa, pb);
  __syncthreads();
  for (int it = threadIdx.x + 256 * blockIdx.y; it < np * C; it += 256 * gridDim.y) {
    const int p = it / C, c = it % C;
    s[(int64_t)(p0 + p) * C + c] = u[(int64_t)(n0 + pa[p]) * C + c] + u[(int64_t)(n0 + pb[p]) * C + c] + (bias ? bias[c] : 0.0f);
  }
}
__global__ __launch_bounds__(256) void k_pair_sum_bwd(dst_layout L, const float* __restrict__ ds, int C, float* __restrict__ du, int accumulate) {
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m];
  for (int it = threadIdx.x + 256 * blockIdx.y; it < n * C; it += 256 * gridDim.y) {
    const int i = it / C, c = it % C;
    float s = 0.0f;
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      const int p = i < j ? pair_index(n, i, j) : pair_index(n, j, i);
      s += ds[(int64_t)(p0 + p) * C + c];
    }
    float* o = du + (int64_t)(n0 + i) * C + c;
    *o = accumulate ? *o + s : s;
  }
}

// four columns per thread (16-byte accesses; round 3's scalar form moved its 125 / 210 MB per launch at 1.1 - 1.5 TB/s)
__global__ __launch_bounds__(256) void k_zbuild_fwd(dst_layout L, const float* __restrict__ ac, const float* __restrict__ ed, float* __restrict__ z) {
  __shared__ unsigned char pa[406], pb[406];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  fill_pair_tables(n, pa, pb);
  __syncthreads();
  for (int it = threadIdx.x + 256 * blockIdx.y; it < np * 2 * 64; it += 256 * gridDim.y) {
    const int d = it >> 6, c = (it & 63) * 4, p = d >> 1, dir = d & 1;
    const int row = dir ? pb[p] : pa[p], col = dir ? pa[p] : pb[p];
    st4(z + (int64_t)(2 * p0 + d) * 256 + c, ld4(ac + (int64_t)(n0 + row) * 512 + c) + ld4(ac + (int64_t)(n0 + col) * 512 + 256 + c) + ld4(ed + (int64_t)(p0 + p) * 256 + c));
  }
}
__global__ __launch_bounds__(256) void k_zbuild_bwd(dst_layout L, const float* __restrict__ dz, float* __restrict__ dac, float* __restrict__ ded) {
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  for (int it = threadIdx.x + 256 * blockIdx.y; it < np * 64; it += 256 * gridDim.y) {
    const int p = it >> 6, c = (it & 63) * 4;
    st4(ded + (int64_t)(p0 + p) * 256 + c, ld4(dz + (int64_t)(2 * (p0 + p)) * 256 + c) + ld4(dz + (int64_t)(2 * (p0 + p) + 1) * 256 + c));
  }
  for (int it = threadIdx.x + 256 * blockIdx.y; it < n * 128; it += 256 * gridDim.y) {
    const int i = it >> 7, c = (it & 127) * 4, as_col = c >> 8, cc = c & 255;
    f4_t s = {0.0f, 0.0f, 0.0f, 0.0f};
    // directed edge with row = i (as_col 0) or col = i (as_col 1): dir 0 has row = lo, dir 1 has row = hi.  Partners below and above i as
    // two branch-free ranges in ascending order (several loads in flight)
#pragma unroll 4
    for (int j = 0; j < i; ++j) s += ld4(dz + (int64_t)(2 * (p0 + pair_index(n, j, i)) + (as_col == 0 ? 1 : 0)) * 256 + cc);
#pragma unroll 4
    for (int j = i + 1; j < n; ++j) s += ld4(dz + (int64_t)(2 * (p0 + pair_index(n, i, j)) + (as_col == 0 ? 0 : 1)) * 256 + cc);
    st4(dac + (int64_t)(n0 + i) * 512 + c, s);
  }
}

// ------------------------------------------------------------------------------------------------------------------ coordinates
__global__ __launch_bounds__(256) void k_coord_fwd(dst_layout L, const float* __restrict__ pos, const float* __restrict__ c2, const int32_t* __restrict__ adj,
                                                    const float* __restrict__ coord_scale, float* __restrict__ pos_out) {
  __shared__ float sp[29][3], sn[29][3];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m];
  for (int i = threadIdx.x; i < n * 3; i += 256) sp[i / 3][i % 3] = pos[(int64_t)(n0 + i / 3) * 3 + i % 3];
  __syncthreads();
  const float scale = coord_scale[0];
  for (int it = threadIdx.x; it < n * 3; it += 256) {
    const int r = it / 3, comp = it % 3;
    float agg = 0.0f;
    for (int c = 0; c < n; ++c) {                         // ascending col = the order of the reference's scatter
      if (c == r) continue;
      const int p = r < c ? pair_index(n, r, c) : pair_index(n, c, r);
      const int d = 2 * (p0 + p) + (r < c ? 0 : 1);       // directed edge (row r, col c)
      const float dx = sp[r][0] - sp[c][0], dy = sp[r][1] - sp[c][1], dz = sp[r][2] - sp[c][2];
      const float nrm = fmaxf(sqrtf(dx * dx + dy * dy + dz * dz), 1e-8f);
      const int bits = adj[p0 + p];
      const float inv = (tanhf(c2[(int64_t)d * 3]) + ((bits & 1) ? tanhf(c2[(int64_t)d * 3 + 1]) : 0.0f) +
                         ((bits & 2) ? tanhf(c2[(int64_t)d * 3 + 2]) : 0.0f)) / 3.0f;
      const float df = comp == 0 ? dx : (comp == 1 ? dy : dz);
      agg += (df / nrm * scale) * inv;
    }
    sn[r][comp] = sp[r][comp] + agg;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < n * 3; it += 256) {
    const int comp = it % 3;
    float mean = 0.0f;
    for (int i = 0; i < n; ++i) mean += sn[i][comp];
    mean /= (float)n;
    pos_out[(int64_t)(n0 + it / 3) * 3 + comp] = sn[it / 3][comp] - mean;
  }
}

__global__ __launch_bounds__(256) void k_coord_bwd(dst_layout L, const float* __restrict__ pos, const float* __restrict__ c2, const int32_t* __restrict__ adj,
                                                    const float* __restrict__ coord_scale, const float* __restrict__ dpos_out, float* __restrict__ dpos_in,
                                                    float* __restrict__ dc2, float* __restrict__ dscale_part) {
  __shared__ float sp[29][3], g[29][3];
  __shared__ float dcd[812][3];                          // d coord_diff per directed edge
  __shared__ float red[4];
  __shared__ unsigned char pa[406], pb[406];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  fill_pair_tables(n, pa, pb);
  for (int i = threadIdx.x; i < n * 3; i += 256) sp[i / 3][i % 3] = pos[(int64_t)(n0 + i / 3) * 3 + i % 3];
  __syncthreads();
  // centre-of-mass projection of the incoming gradient
  for (int it = threadIdx.x; it < n * 3; it += 256) {
    const int comp = it % 3;
    float mean = 0.0f;
    for (int i = 0; i < n; ++i) mean += dpos_out[(int64_t)(n0 + i) * 3 + comp];
    mean /= (float)n;
    g[it / 3][comp] = dpos_out[(int64_t)(n0 + it / 3) * 3 + comp] - mean;
  }
  __syncthreads();
  const float scale = coord_scale[0];
  float dscale = 0.0f;
  for (int d = threadIdx.x; d < np * 2; d += 256) {       // thread per directed edge (row r, col c)
    const int p = d >> 1, dir = d & 1;
    const int r = dir ? pb[p] : pa[p], c = dir ? pa[p] : pb[p];
    const float dx = sp[r][0] - sp[c][0], dy = sp[r][1] - sp[c][1], dz = sp[r][2] - sp[c][2];
    const float raw = sqrtf(dx * dx + dy * dy + dz * dz);
    const float nrm = fmaxf(raw, 1e-8f);
    const float ux = dx / nrm, uy = dy / nrm, uz = dz / nrm;
    const int bits = adj[p0 + p];
    const int64_t e = (int64_t)(2 * p0 + d) * 3;
    const float t0 = tanhf(c2[e]), t1 = tanhf(c2[e + 1]), t2 = tanhf(c2[e + 2]);
    const float inv = (t0 + ((bits & 1) ? t1 : 0.0f) + ((bits & 2) ? t2 : 0.0f)) / 3.0f;
    const float gx = g[r][0], gy = g[r][1], gz = g[r][2];   // d trans = d pos_new[row]
    const float gu = gx * ux + gy * uy + gz * uz;
    dscale += gu * inv;
    const float dinv = gu * scale;
    dc2[e] = dinv / 3.0f * (1.0f - t0 * t0);
    dc2[e + 1] = (bits & 1) ? dinv / 3.0f * (1.0f - t1 * t1) : 0.0f;
    dc2[e + 2] = (bits & 2) ? dinv / 3.0f * (1.0f - t2 * t2) : 0.0f;
    // d unit = g * scale * inv ; d diff = (d unit - u (u . d unit)) / nrm   (zero when the norm is clamped)
    const float k = scale * inv;
    float ex = 0.0f, ey = 0.0f, ez = 0.0f;
    if (raw > 1e-8f) {
      ex = k * (gx - ux * gu) / nrm;
      ey = k * (gy - uy * gu) / nrm;
      ez = k * (gz - uz * gu) / nrm;
    } else {
      ex = k * gx / nrm; ey = k * gy / nrm; ez = k * gz / nrm;   // clamp(min) passes the gradient of diff / const
    }
    dcd[d][0] = ex; dcd[d][1] = ey; dcd[d][2] = ez;
  }
  dscale = wave_sum(dscale);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dscale;
  __syncthreads();
  if (threadIdx.x == 0) dscale_part[m] = (red[0] + red[1]) + (red[2] + red[3]);
  for (int it = threadIdx.x; it < n * 3; it += 256) {
    const int i = it / 3, comp = it % 3;
    float s = g[i][comp];                                  // identity path pos -> pos_new
    for (int j = 0; j < n; ++j) {
      if (j == i) continue;
      const int p = i < j ? pair_index(n, i, j) : pair_index(n, j, i);
      const int d_row = 2 * p + (i < j ? 0 : 1);           // edge (row i, col j): + d diff
      const int d_col = 2 * p + (i < j ? 1 : 0);           // edge (row j, col i): - d diff
      s += dcd[d_row][comp] - dcd[d_col][comp];
    }
    dpos_in[(int64_t)(n0 + i) * 3 + comp] = s;
  }
}

// ------------------------------------------------------------------------------------------------------------------ time features
__global__ void k_time_feat_fwd(const float* __restrict__ nl, const float* __restrict__ w, int B, float* __restrict__ f) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const float x = nl[b];
  f[(int64_t)b * 17] = x;
  for (int i = 0; i < 8; ++i) {
    const float fr = ((x * w[i]) * 2.0f) * 3.14159265358979323846f;
    f[(int64_t)b * 17 + 1 + i] = sinf(fr);
    f[(int64_t)b * 17 + 9 + i] = cosf(fr);
  }
}
__global__ __launch_bounds__(64) void k_time_feat_bwd(const float* __restrict__ nl, const float* __restrict__ w, const float* __restrict__ df, int B,
                                                       float* __restrict__ dw) {
  const int i = blockIdx.x;                                // one wave per frequency
  float s = 0.0f;
  for (int b = threadIdx.x; b < B; b += 64) {
    const float x = nl[b];
    const float fr = ((x * w[i]) * 2.0f) * 3.14159265358979323846f;
    s += (df[(int64_t)b * 17 + 1 + i] * cosf(fr) - df[(int64_t)b * 17 + 9 + i] * sinf(fr)) * (x * 2.0f * 3.14159265358979323846f);
  }
  s = wave_sum(s);
  if (threadIdx.x == 0) dw[i] = s;
}

// ------------------------------------------------------------------------------------------------------------------ loss
__global__ __launch_bounds__(256) void k_loss(dst_layout L, const float* __restrict__ pos, const float* __restrict__ feat, const float* __restrict__ edge,
                                               const float* __restrict__ tpos, const float* __restrict__ tfeat, const float* __restrict__ tedge,
                                               const float* __restrict__ wm, float w_pos, float w_type, float w_edge, float* __restrict__ loss_m,
                                               float* __restrict__ dpos, float* __restrict__ dfeat, float* __restrict__ dedge) {
  __shared__ float red[3][4];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  const float w = wm[m];
  float lp = 0.0f, lt = 0.0f, le = 0.0f;
  for (int it = threadIdx.x; it < n * 3; it += 256) {
    const float d = pos[(int64_t)n0 * 3 + it] - tpos[(int64_t)n0 * 3 + it];
    lp += d * d;
    dpos[(int64_t)n0 * 3 + it] = w * w_pos * (2.0f / 3.0f) * d;
  }
  for (int it = threadIdx.x; it < n * 6; it += 256) {
    const float d = feat[(int64_t)n0 * 6 + it] - tfeat[(int64_t)n0 * 6 + it];
    lt += d * d;
    dfeat[(int64_t)n0 * 6 + it] = w * w_type * (2.0f / 6.0f) * d;
  }
  for (int it = threadIdx.x; it < np * 2; it += 256) {
    const float d = edge[(int64_t)p0 * 2 + it] - tedge[(int64_t)p0 * 2 + it];
    le += d * d;                                            // every pair sits in two cells of the dense edge tensor
    dedge[(int64_t)p0 * 2 + it] = w * w_edge * 2.0f * d;    // 2 cells * (1/2 channel mean) * 2 d
  }
  lp = wave_sum(lp); lt = wave_sum(lt); le = wave_sum(le);
  if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = lp; red[1][threadIdx.x >> 6] = lt; red[2][threadIdx.x >> 6] = le; }
  __syncthreads();
  if (threadIdx.x == 0) {
    const float a = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    const float b = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    const float c = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    loss_m[m] = w * (w_pos * a / 3.0f + w_type * b / 6.0f + w_edge * c);     // c: 2 cells * mean over 2 channels = 1
  }
}

// ------------------------------------------------------------------------------------------------------------------ noising / Kabsch
__global__ __launch_bounds__(256) void k_noising(dst_layout L, const float* __restrict__ alpha, const float* __restrict__ sigma, const float* __restrict__ x,
                                                  const float* __restrict__ raw, float* __restrict__ z, const float* __restrict__ ex,
                                                  const float* __restrict__ eraw, float* __restrict__ ez) {
  __shared__ float mean[3];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  const float a = alpha[m], s = sigma[m];
  if (threadIdx.x < 3) {
    float t = 0.0f;
    for (int i = 0; i < n; ++i) t += raw[(int64_t)(n0 + i) * 9 + threadIdx.x];
    mean[threadIdx.x] = t / (float)n;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < n * 9; it += 256) {
    const int c = it % 9;
    float e = raw[(int64_t)n0 * 9 + it];
    if (c < 3) e -= mean[c];
    z[(int64_t)n0 * 9 + it] = a * x[(int64_t)n0 * 9 + it] + s * e;
  }
  for (int it = threadIdx.x; it < np * 2; it += 256) ez[(int64_t)p0 * 2 + it] = a * ex[(int64_t)p0 * 2 + it] + s * eraw[(int64_t)p0 * 2 + it];
}

// process_edge_batch + get_data_scaler (losses.py:498-529, utils.py:33-68; centered data): CoM-free positions / pos_norm, one-hot types
// * 2 - 1 over type_norm, charges over charge_norm; pair features * 2 - 1 over edge_norm.
__global__ __launch_bounds__(256) void k_prepare_batch(dst_layout L, const float* __restrict__ pos, const float* __restrict__ one_hot, const float* __restrict__ fc,
                                                        const float* __restrict__ edge, float pos_norm, float type_norm, float fc_norm, float edge_norm,
                                                        float* __restrict__ x, float* __restrict__ ex) {
  __shared__ float mean[3];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  if (threadIdx.x < 3) {
    float t = 0.0f;
    for (int i = 0; i < n; ++i) t += pos[(int64_t)(n0 + i) * 3 + threadIdx.x];
    mean[threadIdx.x] = t / (float)n;
  }
  __syncthreads();
  for (int it = threadIdx.x; it < n * 9; it += 256) {
    const int i = it / 9, c = it % 9;
    float v;
    if (c < 3) v = (pos[(int64_t)(n0 + i) * 3 + c] - mean[c]) / pos_norm;
    else if (c < 8) v = (one_hot[(int64_t)(n0 + i) * 5 + (c - 3)] * 2.0f - 1.0f) / type_norm;
    else v = fc[n0 + i] / fc_norm;
    x[(int64_t)n0 * 9 + it] = v;
  }
  for (int it = threadIdx.x; it < np * 2; it += 256) ex[(int64_t)p0 * 2 + it] = (edge[(int64_t)p0 * 2 + it] * 2.0f - 1.0f) / edge_norm;
}

// 3x3 SVD by one-sided Jacobi on columns (fp64): A V = U S.
__device__ void svd3(const double A[3][3], double U[3][3], double S[3], double V[3][3]) {
  double W[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { W[i][j] = A[i][j]; V[i][j] = (i == j) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0.0;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int i = 0; i < 3; ++i) { al += W[i][p] * W[i][p]; be += W[i][q] * W[i][q]; ga += W[i][p] * W[i][q]; }
        off = fmax(off, fabs(ga) / (sqrt(al * be) + 1e-300));
        if (fabs(ga) < 1e-300) continue;
        const double zeta = (be - al) / (2.0 * ga);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
        const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
        for (int i = 0; i < 3; ++i) {
          const double wp = W[i][p], wq = W[i][q];
          W[i][p] = c * wp - s * wq; W[i][q] = s * wp + c * wq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  for (int j = 0; j < 3; ++j) {
    S[j] = sqrt(W[0][j] * W[0][j] + W[1][j] * W[1][j] + W[2][j] * W[2][j]);
    for (int i = 0; i < 3; ++i) U[i][j] = S[j] > 1e-300 ? W[i][j] / S[j] : 0.0;
  }
  // sort singular values descending (the sign correction of Kabsch acts on the smallest one)
  for (int a = 0; a < 2; ++a)
    for (int b = a + 1; b < 3; ++b)
      if (S[b] > S[a]) {
        const double ts = S[a]; S[a] = S[b]; S[b] = ts;
        for (int i = 0; i < 3; ++i) {
          const double tu = U[i][a]; U[i][a] = U[i][b]; U[i][b] = tu;
          const double tv = V[i][a]; V[i][a] = V[i][b]; V[i][b] = tv;
        }
      }
  // complete a rank-deficient U to an orthonormal basis (columns with zero singular value)
  if (S[2] <= 1e-300 * 0 + 1e-14 * (S[0] + 1e-300)) {
    if (S[1] > 1e-14 * (S[0] + 1e-300)) {
      U[0][2] = U[1][0] * U[2][1] - U[2][0] * U[1][1];
      U[1][2] = U[2][0] * U[0][1] - U[0][0] * U[2][1];
      U[2][2] = U[0][0] * U[1][1] - U[1][0] * U[0][1];
    }
  }
}

__global__ __launch_bounds__(64) void k_kabsch(dst_layout L, const float* __restrict__ pred, int64_t ldp, const float* __restrict__ tar, int64_t ldt,
                                                float* __restrict__ rot, float* __restrict__ aligned) {
  __shared__ float R[9];
  const int m = blockIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0;
  if (threadIdx.x == 0) {
    double A[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, U[3][3], S[3], V[3][3];
    for (int k = 0; k < n; ++k)
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] += (double)pred[(int64_t)(n0 + k) * ldp + i] * (double)tar[(int64_t)(n0 + k) * ldt + j];
    svd3(A, U, S, V);
    const double det = A[0][0] * (A[1][1] * A[2][2] - A[1][2] * A[2][1]) - A[0][1] * (A[1][0] * A[2][2] - A[1][2] * A[2][0]) +
                       A[0][2] * (A[1][0] * A[2][1] - A[1][1] * A[2][0]);
    const double sg = det > 0 ? 1.0 : (det < 0 ? -1.0 : 0.0);
    for (int i = 0; i < 3; ++i)
      for (int l = 0; l < 3; ++l) R[i * 3 + l] = (float)(U[i][0] * V[l][0] + U[i][1] * V[l][1] + sg * U[i][2] * V[l][2]);
  }
  __syncthreads();
  if (threadIdx.x < 9) rot[(int64_t)m * 9 + threadIdx.x] = R[threadIdx.x];
  // aligned[j, k] = sum_i rot[k, i] tar[j, i]   (einsum "...ki, ...ji -> ...jk", losses.py:420)
  for (int it = threadIdx.x; it < n * 3; it += 64) {
    const int j = it / 3, k = it % 3;
    const float* t = tar + (int64_t)(n0 + j) * ldt;
    aligned[(int64_t)(n0 + j) * 3 + k] = R[k * 3] * t[0] + R[k * 3 + 1] * t[1] + R[k * 3 + 2] * t[2];
  }
}

// ------------------------------------------------------------------------------------------------------------------ BatchNorm (training)
// stage 1: per-chunk column sums of x and x^2 in fp64-free two-pass form: first the mean, then the centred second moment.
__global__ __launch_bounds__(256) void k_bn_sum(const float* __restrict__ x, int R, int C, const float* __restrict__ mean, float* __restrict__ partial,
                                                 int rows_per_chunk) {
  __shared__ float red[4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s = 0.0f;
  if (col < C) {
    const float mu = mean ? mean[col] : 0.0f;
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) {
      const float v = x[(int64_t)r * C + col];
      s += mean ? (v - mu) * (v - mu) : v;
    }
  }
  red[rl][threadIdx.x & 63] = s;
  __syncthreads();
  if (rl == 0 && col < C) partial[(int64_t)blockIdx.y * C + col] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}
// one wave per channel: lane l adds chunks l, l + 64, ..., then a fixed butterfly (the one-thread-per-channel loop was ~170 dependent
// L2 round trips = 40 us for a 128-channel result)
__device__ __forceinline__ float bn_chunk_sum(const float* __restrict__ partial, int chunks, int C, int c) {
  float s = 0.0f;
  for (int k = threadIdx.x; k < chunks; k += 64) s += partial[(int64_t)k * C + c];
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  return s;
}
__global__ __launch_bounds__(64) void k_bn_finish_mean(const float* __restrict__ partial, int chunks, int C, int R, float* __restrict__ stats) {
  const int c = blockIdx.x;
  const float s = bn_chunk_sum(partial, chunks, C, c);
  if (threadIdx.x == 0) stats[c] = s / (float)R;
}
__global__ __launch_bounds__(64) void k_bn_finish_var(const float* __restrict__ partial, int chunks, int C, int R, float eps, float* __restrict__ stats,
                                                      float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x;
  const float s = bn_chunk_sum(partial, chunks, C, c);
  if (threadIdx.x != 0) return;
  const float var = s / (float)R;
  stats[C + c] = 1.0f / sqrtf(var + eps);
  stats[2 * C + c] = s / (float)(R - 1);                          // the unbiased variance the running statistic takes (dst_bn_running_again)
  if (running_mean) {
    running_mean[c] = 0.9f * running_mean[c] + 0.1f * stats[c];
    running_var[c] = 0.9f * running_var[c] + 0.1f * stats[2 * C + c];
  }
}
__global__ void k_bn_running_again(const float* __restrict__ stats, int C, float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  running_mean[c] = 0.9f * running_mean[c] + 0.1f * stats[c];
  running_var[c] = 0.9f * running_var[c] + 0.1f * stats[2 * C + c];
}
__global__ void k_bn_apply(const float* __restrict__ x, int64_t total, int C, const float* __restrict__ stats, const float* __restrict__ gamma,
                           const float* __restrict__ beta, float* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  y[i] = (x[i] - stats[c]) * stats[C + c] * gamma[c] + beta[c];
}
// backward stage 1: per-chunk sums of dy and dy * xhat
__global__ __launch_bounds__(256) void k_bn_bwd_sum(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats, int R, int C,
                                                     float* __restrict__ partial, int rows_per_chunk, int chunks) {
  __shared__ float red[2][4][64];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per_chunk, r1 = min(R, r0 + rows_per_chunk);
  float s1 = 0.0f, s2 = 0.0f;
  if (col < C) {
    const float mu = stats[col], rs = stats[C + col];
#pragma unroll 8
    for (int r = r0 + rl; r < r1; r += 4) {
      const float g = dy[(int64_t)r * C + col];
      s1 += g;
      s2 += g * ((x[(int64_t)r * C + col] - mu) * rs);
    }
  }
  red[0][rl][threadIdx.x & 63] = s1;
  red[1][rl][threadIdx.x & 63] = s2;
  __syncthreads();
  if (rl == 0 && col < C) {
    const int t = threadIdx.x;
    partial[(int64_t)blockIdx.y * C + col] = (red[0][0][t] + red[0][1][t]) + (red[0][2][t] + red[0][3][t]);
    partial[(int64_t)(chunks + blockIdx.y) * C + col] = (red[1][0][t] + red[1][1][t]) + (red[1][2][t] + red[1][3][t]);
  }
}
__global__ __launch_bounds__(64) void k_bn_bwd_finish(const float* __restrict__ partial, int chunks, int C, float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x;
  const float s1 = bn_chunk_sum(partial, chunks, C, c);
  const float s2 = bn_chunk_sum(partial + (int64_t)chunks * C, chunks, C, c);
  if (threadIdx.x == 0) { dbeta[c] = s1; dgamma[c] = s2; }
}
__global__ void k_bn_bwd_apply(const float* __restrict__ dy, const float* __restrict__ x, int64_t total, int C, int R, const float* __restrict__ stats,
                               const float* __restrict__ gamma, const float* __restrict__ dgamma, const float* __restrict__ dbeta, float* __restrict__ dx) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  const float xh = (x[i] - stats[c]) * stats[C + c];
  dx[i] = gamma[c] * stats[C + c] * (dy[i] - dbeta[c] / (float)R - xh * dgamma[c] / (float)R);
}

// ------------------------------------------------------------------------------------------------------------------ SpecFormer attention
// One workgroup per (batch, head, slice of the query rows), one wave per query row, lanes over the keys (L <= 512: up to 8 per lane,
// the row lives in registers).  K and V of the head sit in LDS as [dk][L] (consecutive lanes = consecutive keys: conflict-free; the
// first version's [L][dk] was a 16-way bank conflict).  dk is a template constant - with a run-time dk the per-row register arrays
// were indexed dynamically - and the NEXT row's global loads are issued before this row's stores: vector-memory operations retire in
// issue order, so a row that loads after the previous row's stores waits for those stores first (14 us per row in that form).
template <int DK>
__global__ __launch_bounds__(256) void k_spec_attn_fwd(const float* __restrict__ qkv, const float* __restrict__ prev, float* __restrict__ scores,
                                                        float* __restrict__ stats, float* __restrict__ out, int B, int Lq, int H, float scale, int Lp) {
  extern __shared__ float sm[];
  float* Ks = sm;
  float* Vs = sm + (size_t)Lq * DK;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;     // consecutive workgroups (blockIdx.x) work on the same [L, L] matrix
  const int D = H * DK;
  for (int i = threadIdx.x; i < Lq * DK; i += 256) {
    const int l = i / DK, c = i % DK;
    Ks[c * Lq + l] = qkv[((int64_t)b * Lq + l) * 3 * D + D + h * DK + c];
    Vs[c * Lq + l] = qkv[((int64_t)b * Lq + l) * 3 * D + 2 * D + h * DK + c];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int MAXJ = 8;
  const int step = gridDim.x * 4;
  int qi = blockIdx.x * 4 + wave;
  float qn[DK], pn[MAXJ];
  auto issue = [&](int row) {                              // the global loads of one row: q (wave-uniform) and the previous layer's scores
    const float* qp = qkv + ((int64_t)b * Lq + row) * 3 * D + h * DK;
#pragma unroll
    for (int c = 0; c < DK; ++c) qn[c] = qp[c];
    const int64_t rb = (((int64_t)b * H + h) * Lq + row) * Lp;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) pn[j] = (prev && lane + 64 * j < Lq) ? prev[rb + lane + 64 * j] : 0.0f;
  };
  if (qi < Lq) issue(qi);
  for (; qi < Lq; qi += step) {
    float q[DK], pv[MAXJ];
#pragma unroll
    for (int c = 0; c < DK; ++c) q[c] = qn[c];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) pv[j] = pn[j];
    if (qi + step < Lq) issue(qi + step);
    const int64_t rowbase = (((int64_t)b * H + h) * Lq + qi) * Lp;
    float sv[MAXJ];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int k = lane + 64 * j;
      sv[j] = -INFINITY;
      if (k < Lq) {
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < DK; ++c) s += q[c] * Ks[c * Lq + k];
        s = s * scale + pv[j];
        sv[j] = s;
        mx = fmaxf(mx, s);
      }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float den = 0.0f, ev[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      ev[j] = (lane + 64 * j < Lq) ? expf(sv[j] - mx) : 0.0f;
      den += ev[j];
    }
    den = wave_sum(den);
    float acc[DK];
#pragma unroll
    for (int c = 0; c < DK; ++c) acc[c] = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int k = lane + 64 * j;
      if (k < Lq) {
        const float a = ev[j] / den;                     // the probabilities are NOT stored: the backward re-creates them from scores + (max, sum)
#pragma unroll
        for (int c = 0; c < DK; ++c) acc[c] += a * Vs[c * Lq + k];
      }
    }
#pragma unroll
    for (int c = 0; c < DK; ++c) acc[c] = wave_sum(acc[c]);
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (lane + 64 * j < Lq) scores[rowbase + lane + 64 * j] = sv[j];
    if (lane == 0) {
      stats[(((int64_t)b * H + h) * Lq + qi) * 2] = mx;
      stats[(((int64_t)b * H + h) * Lq + qi) * 2 + 1] = den;
#pragma unroll
      for (int c = 0; c < DK; ++c) out[((int64_t)b * Lq + qi) * D + h * DK + c] = acc[c];
    }
  }
}

// backward, pass 1 (per query row): dS = p * (dP - sum p dP) (+ dscores_in) with p re-created from scores + stats, and dq.  Same row
// pipeline as the forward.
template <int DK>
__global__ __launch_bounds__(256) void k_spec_attn_bwd_q(const float* __restrict__ qkv, const float* __restrict__ scores, const float* __restrict__ stats,
                                                          const float* __restrict__ dout, const float* __restrict__ dscores_in, float* __restrict__ dqkv,
                                                          float* __restrict__ dscores, int B, int Lq, int H, float scale, int Lp) {
  extern __shared__ float sm[];
  float* Ks = sm;
  float* Vs = sm + (size_t)Lq * DK;
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int D = H * DK;
  for (int i = threadIdx.x; i < Lq * DK; i += 256) {
    const int l = i / DK, c = i % DK;
    Ks[c * Lq + l] = qkv[((int64_t)b * Lq + l) * 3 * D + D + h * DK + c];
    Vs[c * Lq + l] = qkv[((int64_t)b * Lq + l) * 3 * D + 2 * D + h * DK + c];
  }
  __syncthreads();
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  constexpr int MAXJ = 8;
  const int step = gridDim.x * 4;
  int qi = blockIdx.x * 4 + wave;
  float gn[DK], sn[MAXJ], dn[MAXJ], mxn = 0.0f, denn = 1.0f;
  auto issue = [&](int row) {
    const float* gp = dout + ((int64_t)b * Lq + row) * D + h * DK;
#pragma unroll
    for (int c = 0; c < DK; ++c) gn[c] = gp[c];
    const int64_t rb = (((int64_t)b * H + h) * Lq + row) * Lp;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int k = lane + 64 * j;
      sn[j] = k < Lq ? scores[rb + k] : -INFINITY;
      dn[j] = (dscores_in && k < Lq) ? dscores_in[rb + k] : 0.0f;
    }
    mxn = stats[(((int64_t)b * H + h) * Lq + row) * 2];
    denn = stats[(((int64_t)b * H + h) * Lq + row) * 2 + 1];
  };
  if (qi < Lq) issue(qi);
  for (; qi < Lq; qi += step) {
    float go[DK], av[MAXJ], din[MAXJ];
#pragma unroll
    for (int c = 0; c < DK; ++c) go[c] = gn[c];
    const float mx = mxn, den = denn;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) { av[j] = sn[j]; din[j] = dn[j]; }
    if (qi + step < Lq) issue(qi + step);
    const int64_t rowbase = (((int64_t)b * H + h) * Lq + qi) * Lp;
    float dot = 0.0f, dav[MAXJ];
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int k = lane + 64 * j;
      dav[j] = 0.0f;
      av[j] = k < Lq ? expf(av[j] - mx) / den : 0.0f;   // the forward's probabilities, re-created
      if (k < Lq) {
        float da = 0.0f;
#pragma unroll
        for (int c = 0; c < DK; ++c) da += go[c] * Vs[c * Lq + k];
        dav[j] = da;
        dot += av[j] * da;
      }
    }
    dot = wave_sum(dot);
    float dq[DK];
#pragma unroll
    for (int c = 0; c < DK; ++c) dq[c] = 0.0f;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j) {
      const int k = lane + 64 * j;
      if (k < Lq) {
        const float ds = av[j] * (dav[j] - dot) + din[j];
        av[j] = ds;
#pragma unroll
        for (int c = 0; c < DK; ++c) dq[c] += ds * Ks[c * Lq + k];
      }
    }
#pragma unroll
    for (int c = 0; c < DK; ++c) dq[c] = wave_sum(dq[c]) * scale;
#pragma unroll
    for (int j = 0; j < MAXJ; ++j)
      if (lane + 64 * j < Lq) dscores[rowbase + lane + 64 * j] = av[j];
    if (lane == 0) {
#pragma unroll
      for (int c = 0; c < DK; ++c) dqkv[((int64_t)b * Lq + qi) * 3 * D + h * DK + c] = dq[c];
    }
  }
}

// backward, pass 2: dk[k] = scale * sum_q dS[q,k] q[q], dv[k] = sum_q p[q,k] dout[q].  One thread per key, the query loop outside:
// every step reads one row segment of dS / scores with consecutive lanes on consecutive keys (coalesced) and needs no reduction -
// the first version walked the columns of the [L, L] matrices with a stride of L floats and took 11.5 ms per layer at 256 molecules.
template <int DK>
__global__ __launch_bounds__(128) void k_spec_attn_bwd_kv(const float* __restrict__ qkv, const float* __restrict__ scores, const float* __restrict__ stats,
                                                           const float* __restrict__ dout, const float* __restrict__ dscores, float* __restrict__ dqkv, int B,
                                                           int Lq, int H, float scale, int Lp) {
  extern __shared__ float sm[];
  float* Qs = sm;
  float* Gs = sm + (size_t)Lq * DK;
  float* St = sm + (size_t)2 * Lq * DK;                    // (max, 1 / sum) of every query row
  const int bh = blockIdx.y, b = bh / H, h = bh % H;
  const int D = H * DK;
  for (int i = threadIdx.x; i < Lq * DK; i += blockDim.x) {
    const int l = i / DK, c = i % DK;
    Qs[i] = qkv[((int64_t)b * Lq + l) * 3 * D + h * DK + c];
    Gs[i] = dout[((int64_t)b * Lq + l) * D + h * DK + c];
  }
  for (int i = threadIdx.x; i < Lq; i += blockDim.x) {
    St[2 * i] = stats[((int64_t)b * H + h) * Lq * 2 + 2 * i];
    St[2 * i + 1] = 1.0f / stats[((int64_t)b * H + h) * Lq * 2 + 2 * i + 1];
  }
  __syncthreads();
  const int ki = blockIdx.x * blockDim.x + threadIdx.x;
  if (ki >= Lq) return;
  const int64_t base = ((int64_t)b * H + h) * Lq * Lp + ki;
  float dkk[DK], dvv[DK];
#pragma unroll
  for (int c = 0; c < DK; ++c) { dkk[c] = 0.0f; dvv[c] = 0.0f; }
#pragma unroll 4
  for (int q = 0; q < Lq; ++q) {
    const float ds = dscores[base + (int64_t)q * Lp];
    const float a = expf(scores[base + (int64_t)q * Lp] - St[2 * q]) * St[2 * q + 1];
#pragma unroll
    for (int c = 0; c < DK; ++c) { dkk[c] += ds * Qs[q * DK + c]; dvv[c] += a * Gs[q * DK + c]; }
  }
#pragma unroll
  for (int c = 0; c < DK; ++c) {
    dqkv[((int64_t)b * Lq + ki) * 3 * D + D + h * DK + c] = dkk[c] * scale;
    dqkv[((int64_t)b * Lq + ki) * 3 * D + 2 * D + h * DK + c] = dvv[c];
  }
}

// ------------------------------------------------------------------------------------------------------------------ LayerNorm (affine)
__global__ __launch_bounds__(64) void k_ln_affine_fwd(const float* __restrict__ x, int C, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float eps, float* __restrict__ y, float* __restrict__ stats) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float s = 0.0f;
  for (int c = lane; c < C; c += 64) s += x[(int64_t)r * C + c];
  const float mean = wave_sum(s) / (float)C;
  float q = 0.0f;
  for (int c = lane; c < C; c += 64) { const float d = x[(int64_t)r * C + c] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)C + eps);
  for (int c = lane; c < C; c += 64) y[(int64_t)r * C + c] = (x[(int64_t)r * C + c] - mean) * rstd * gamma[c] + beta[c];
  if (lane == 0) { stats[(int64_t)r * 2] = mean; stats[(int64_t)r * 2 + 1] = rstd; }
}
__global__ __launch_bounds__(64) void k_ln_affine_bwd(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats, int C,
                                                       const float* __restrict__ gamma, float* __restrict__ dx) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float mean = stats[(int64_t)r * 2], rstd = stats[(int64_t)r * 2 + 1];
  float s1 = 0.0f, s2 = 0.0f;
  for (int c = lane; c < C; c += 64) {
    const float g = dy[(int64_t)r * C + c] * gamma[c], xh = (x[(int64_t)r * C + c] - mean) * rstd;
    s1 += g;
    s2 += g * xh;
  }
  const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
  for (int c = lane; c < C; c += 64) {
    const float g = dy[(int64_t)r * C + c] * gamma[c], xh = (x[(int64_t)r * C + c] - mean) * rstd;
    dx[(int64_t)r * C + c] = rstd * (g - m1 - xh * m2);
  }
}
__global__ void k_ln_affine_bwd_params(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ stats, int R, int C,
                                       float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.0f, s2 = 0.0f;
  for (int r = 0; r < R; ++r) {
    const float g = dy[(int64_t)r * C + c];
    s1 += g;
    s2 += g * ((x[(int64_t)r * C + c] - stats[(int64_t)r * 2]) * stats[(int64_t)r * 2 + 1]);
  }
  dbeta[c] = s1;
  dgamma[c] = s2;
}

// ------------------------------------------------------------------------------------------------------------------ optimizer
__global__ void k_adamw_ema(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, float* __restrict__ vmax,
                            float* __restrict__ ema, int64_t n, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                            float clip, const float* __restrict__ clip_dev, float ema_omd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float grad = g[i] * (clip_dev ? clip * clip_dev[0] : clip);
  float w = p[i];
  w *= (1.0f - lr * wd);                                    // decoupled weight decay (torch.optim.AdamW)
  const float mi = beta1 * m[i] + (1.0f - beta1) * grad;
  const float vi = beta2 * v[i] + (1.0f - beta2) * grad * grad;
  const float vm = fmaxf(vmax[i], vi);                      // amsgrad
  m[i] = mi; v[i] = vi; vmax[i] = vm;
  const float denom = sqrtf(vm) / sqrtf(bc2) + eps;
  w -= (lr / bc1) * (mi / denom);
  p[i] = w;
  if (ema) { const float s = ema[i]; ema[i] = s - ema_omd * (s - w); }
}

// gradient_clipping (losses.py:28-50) on the device: the norm history (Queue, losses.py:53-72: newest first, at most 50), the allowed
// norm min(1.5 mean + 2 std, max_grad), the queue update and the clip coefficient min(1, allowed / (norm + 1e-6)) - in double, as numpy
// evaluates them.  state: [0..49] history, [50] count, [51] coefficient, [52] norm, [53] allowed norm.  One thread: ~50 values.
__global__ void k_clip_update(const float* __restrict__ norm_sq, float inv_world, float max_grad, float* __restrict__ st) {
  if (blockIdx.x || threadIdx.x) return;
  const double norm = sqrt((double)norm_sq[0]) * (double)inv_world;
  double allowed = (double)max_grad;
  if (max_grad > 1.0f) {
    const int cnt = (int)st[50];
    double mean = 0.0, var = 0.0;
    for (int i = 0; i < cnt; ++i) mean += (double)st[i];
    mean /= (double)(cnt > 0 ? cnt : 1);
    for (int i = 0; i < cnt; ++i) { const double d = (double)st[i] - mean; var += d * d; }
    var /= (double)(cnt > 0 ? cnt : 1);
    allowed = fmin(1.5 * mean + 2.0 * sqrt(var), (double)max_grad);
    const int keep = cnt < 49 ? cnt : 49;                      // insert at the front, drop the oldest beyond 50
    for (int i = keep; i > 0; --i) st[i] = st[i - 1];
    st[0] = (float)(norm > allowed ? allowed : norm);
    st[50] = (float)(keep + 1);
  }
  st[51] = (float)fmin(1.0, allowed / (norm + 1e-6));
  st[52] = (float)norm;
  st[53] = (float)allowed;
}

// four parameters per thread, 16-byte accesses (same arithmetic per element)
__global__ void k_adamw_ema4(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m, float4* __restrict__ v, float4* __restrict__ vmax,
                             float4* __restrict__ ema, int64_t n4, float lr, float beta1, float beta2, float eps, float wd, float bc1, float bc2,
                             float clip, const float* __restrict__ clip_dev, float ema_omd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n4) return;
  const float cl = clip_dev ? clip * clip_dev[0] : clip;
  const float4 G = g[i];
  float4 W = p[i], M = m[i], V = v[i], X = vmax[i], S = ema ? ema[i] : W;
  float* w = &W.x; float* mm = &M.x; float* vv = &V.x; float* xx = &X.x; float* ss = &S.x;
  const float* gg = &G.x;
  const float sq2 = sqrtf(bc2);
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float grad = gg[e] * cl;
    float wv = w[e] * (1.0f - lr * wd);
    const float mi = beta1 * mm[e] + (1.0f - beta1) * grad;
    const float vi = beta2 * vv[e] + (1.0f - beta2) * grad * grad;
    const float vm = fmaxf(xx[e], vi);
    mm[e] = mi; vv[e] = vi; xx[e] = vm;
    const float denom = sqrtf(vm) / sq2 + eps;
    wv -= (lr / bc1) * (mi / denom);
    w[e] = wv;
    ss[e] = ss[e] - ema_omd * (ss[e] - wv);
  }
  p[i] = W; m[i] = M; v[i] = V; vmax[i] = X;
  if (ema) ema[i] = S;
}

inline dim3 grid1d(int64_t n, int block = 256) { return dim3((unsigned)((n + block - 1) / block)); }

}  // namespace

// ====================================================================================================================== C-ABI
extern "C" {

int dst_adj_bits(const float* cond_e, int64_t ld, const float* d2c, float edge_th, float cutoff, int32_t Pp, int32_t* adj, void* stream) {
  if (Pp < 0 || (Pp > 0 && (!cond_e || !d2c || !adj))) return DS_ERR_ARG;
  if (Pp == 0) return DS_OK;
  hipLaunchKernelGGL(k_adj_bits, grid1d(Pp), dim3(256), 0, (hipStream_t)stream, cond_e, ld, d2c, edge_th, cutoff, (int)Pp, adj);
  return DST_CHECK_LAUNCH();
}

int dst_copy_pieces(const dst_piece* table, int32_t n, void* stream) {
  if (n < 0 || (n > 0 && !table)) return DS_ERR_ARG;
  if (n == 0) return DS_OK;
  hipLaunchKernelGGL(k_copy_pieces, dim3(n, 128), dim3(256), 0, (hipStream_t)stream, table);
  return DST_CHECK_LAUNCH();
}

int dst_pack_bf16_pieces(const dst_piece* table, int32_t n, void* stream) {
  if (n < 0 || (n > 0 && !table)) return DS_ERR_ARG;
  if (n == 0) return DS_OK;
  hipLaunchKernelGGL(k_pack_bf16_pieces, dim3(n, 16), dim3(256), 0, (hipStream_t)stream, table);
  return DST_CHECK_LAUNCH();
}

int dst_colsum(const float* X, int64_t ld, int32_t R, int32_t C, float* out, int32_t accumulate, float* scratch, int64_t scratch_cap,
               void* stream) {
  if (!X || !out || !scratch || R < 0 || C <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int chunks = (R + 511) / 512;
  if (chunks < 1) chunks = 1;
  if (chunks > 1024) chunks = 1024;
  if ((int64_t)chunks * C > scratch_cap) chunks = (int)(scratch_cap / C);
  if (chunks < 1) return DS_ERR_ARG;
  const int rpc = (R + chunks - 1) / chunks > 0 ? (R + chunks - 1) / chunks : 1;
  hipLaunchKernelGGL(k_colsum_partial, dim3((C + 63) / 64, chunks), dim3(256), 0, s, X, ld, (int)R, (int)C, scratch, rpc);
  hipLaunchKernelGGL(k_colsum_final, grid1d(C), dim3(256), 0, s, (const float*)scratch, chunks, (int)C, out, (int)accumulate);
  return DST_CHECK_LAUNCH();
}

int dst_sumsq(const float* x, int64_t n, float* out, int32_t accumulate, float* scratch, int64_t scratch_cap, void* stream) {
  if (!x || !out || !scratch || scratch_cap < 256) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_sumsq_partial, dim3(256), dim3(256), 0, s, x, n, scratch);
  hipLaunchKernelGGL(k_colsum_partial, dim3(1, 1), dim3(256), 0, s, (const float*)scratch, (int64_t)1, 256, 1, scratch + 256, 256);
  hipLaunchKernelGGL(k_colsum_final, dim3(1), dim3(64), 0, s, (const float*)(scratch + 256), 1, 1, out, (int)accumulate);
  return DST_CHECK_LAUNCH();
}

int dst_act_fwd(const float* x, float* y, int64_t n, int32_t kind, void* stream) {
  if (!x || !y || kind < 1 || kind > 3) return DS_ERR_ARG;
  if (n == 0) return DS_OK;
  hipLaunchKernelGGL(k_act_fwd, grid1d(n), dim3(256), 0, (hipStream_t)stream, x, y, n, (int)kind);
  return DST_CHECK_LAUNCH();
}
int dst_act_bwd(const float* dy, const float* ref, float* dx, int64_t n, int32_t kind, void* stream) {
  if (!dy || !ref || !dx || kind < 1 || kind > 3) return DS_ERR_ARG;
  if (n == 0) return DS_OK;
  hipLaunchKernelGGL(k_act_bwd, grid1d(n), dim3(256), 0, (hipStream_t)stream, dy, ref, dx, n, (int)kind);
  return DST_CHECK_LAUNCH();
}
int dst_axpy(float a, const float* x, float* y, int64_t n, void* stream) {
  if (!x || !y) return DS_ERR_ARG;
  if (n == 0) return DS_OK;
  if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0)
    hipLaunchKernelGGL(k_axpy4, grid1d(n / 4), dim3(256), 0, (hipStream_t)stream, a, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y), n / 4);
  else
    hipLaunchKernelGGL(k_axpy, grid1d(n), dim3(256), 0, (hipStream_t)stream, a, x, y, n);
  return DST_CHECK_LAUNCH();
}

int dst_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint32_t stream_id, void* stream) {
  if (!x || !y || !(p >= 0.0f && p < 1.0f)) return DS_ERR_ARG;
  if (n == 0) return DS_OK;
  hipLaunchKernelGGL(k_dropout, grid1d((n + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, n, p, 1.0f / (1.0f - p), (unsigned long long)seed,
                     (unsigned int)stream_id);
  return DST_CHECK_LAUNCH();
}

int dst_lnmod_fwd(const float* x, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B, const float* ada, int64_t ada_ld,
                  int32_t shift_off, int32_t scale_off, float* y, float* stats, void* stream) {
  if (!x || !seg_off || !ada || !y || !stats || (C != 64 && C != 256) || B <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int splits = (C == 256 && seg_mul >= 2) ? 8 : 1;                 // the directed rows: hundreds of 256-wide rows per molecule
  if (C == 64) hipLaunchKernelGGL(k_lnmod_fwd<64>, dim3(B), dim3(1024), 0, s, x, seg_off, (int)seg_mul, ada, ada_ld, (int)shift_off, (int)scale_off, y, stats);
  else hipLaunchKernelGGL(k_lnmod_fwd<256>, dim3(B, splits), dim3(1024), 0, s, x, seg_off, (int)seg_mul, ada, ada_ld, (int)shift_off, (int)scale_off, y, stats);
  return DST_CHECK_LAUNCH();
}
int dst_lnmod_bwd(const float* dy, const float* x, const float* stats, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B,
                  const float* ada, float* d_ada, int64_t ada_ld, int32_t shift_off, int32_t scale_off, float* dx, int32_t accumulate,
                  float* scratch, int64_t scratch_cap, void* stream) {
  if (!dy || !x || !stats || !seg_off || !ada || !d_ada || !dx || (C != 64 && C != 256) || B <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int splits = (C == 256 && seg_mul >= 2) ? 8 : 1;                       // as dst_lnmod_fwd; the per-molecule sums then take a second, fixed-order pass
  if (splits > 1 && (!scratch || scratch_cap < (int64_t)splits * B * 2 * C)) splits = 1;
  if (C == 64)
    hipLaunchKernelGGL(k_lnmod_bwd<64>, dim3(B), dim3(1024), 0, s, dy, x, stats, seg_off, (int)seg_mul, ada, d_ada, ada_ld, (int)shift_off, (int)scale_off, dx, (int)accumulate,
                       (float*)nullptr);
  else {
    hipLaunchKernelGGL(k_lnmod_bwd<256>, dim3(B, splits), dim3(1024), 0, s, dy, x, stats, seg_off, (int)seg_mul, ada, d_ada, ada_ld, (int)shift_off, (int)scale_off, dx,
                       (int)accumulate, scratch);
    if (splits > 1)
      hipLaunchKernelGGL(k_lnmod_bwd_finish<256>, dim3(B), dim3(512), 0, s, (const float*)scratch, splits, (int)B, d_ada, ada_ld, (int)shift_off, (int)scale_off);
  }
  return DST_CHECK_LAUNCH();
}

int dst_gate_add_fwd(const float* r, const float* z, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B, const float* ada,
                     int64_t ada_ld, int32_t gate_off, float* out, void* stream) {
  if (!r || !z || !seg_off || !ada || !out || (C != 64 && C != 256) || B <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (C == 64) hipLaunchKernelGGL(k_gate_add_fwd<64>, dim3(B), dim3(1024), 0, s, r, z, seg_off, (int)seg_mul, ada, ada_ld, (int)gate_off, out);
  else hipLaunchKernelGGL(k_gate_add_fwd<256>, dim3(B), dim3(1024), 0, s, r, z, seg_off, (int)seg_mul, ada, ada_ld, (int)gate_off, out);
  return DST_CHECK_LAUNCH();
}
int dst_gate_add_bwd(const float* dout, const float* z, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B, const float* ada,
                     float* d_ada, int64_t ada_ld, int32_t gate_off, float* dr, int32_t accumulate_r, float* dz, float drop_p, uint64_t drop_seed,
                     uint32_t drop_stream, void* stream) {
  if (!dout || !z || !seg_off || !ada || !d_ada || !dz || (C != 64 && C != 256) || B <= 0 || !(drop_p >= 0.0f && drop_p < 1.0f)) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (C == 64)
    hipLaunchKernelGGL(k_gate_add_bwd<64>, dim3(B), dim3(1024), 0, s, dout, z, seg_off, (int)seg_mul, ada, d_ada, ada_ld, (int)gate_off, dr, (int)accumulate_r, dz, drop_p, (unsigned long long)drop_seed, (unsigned int)drop_stream);
  else
    hipLaunchKernelGGL(k_gate_add_bwd<256>, dim3(B), dim3(1024), 0, s, dout, z, seg_off, (int)seg_mul, ada, d_ada, ada_ld, (int)gate_off, dr, (int)accumulate_r, dz, drop_p, (unsigned long long)drop_seed, (unsigned int)drop_stream);
  return DST_CHECK_LAUNCH();
}

#define DST_L_OK(L) ((L) && (L)->B > 0 && (L)->node_off && (L)->pair_off)

int dst_geom_fwd(const dst_layout* L, const float* pos, const float* ada, int64_t ada_ld, int32_t dist_off, const float* means,
                 const float* stds, float* X, int64_t ldx, float* xs, float* d2s, void* stream) {
  if (!DST_L_OK(L) || !pos || !ada || !means || !stds || !X || !xs || !d2s) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_geom_fwd, dim3(L->B), dim3(GEOM_NT), 0, (hipStream_t)stream, *L, pos, ada, ada_ld, (int)dist_off, means, stds, X, ldx, xs, d2s);
  return DST_CHECK_LAUNCH();
}
int dst_geom_bwd(const dst_layout* L, const float* pos, const float* ada, float* d_ada, int64_t ada_ld, int32_t dist_off,
                 const float* means, const float* stds, const float* xs, const float* d2s, const float* g1, int64_t ld1, const float* g2,
                 int64_t ld2, float* dms, float* dd2_scratch, float* dpos, void* stream) {
  if (!DST_L_OK(L) || !pos || !ada || !d_ada || !means || !stds || !xs || !d2s || !g1 || !dms || !dd2_scratch) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_geom_bwd, dim3(L->B), dim3(GEOM_NT), 0, (hipStream_t)stream, *L, pos, ada, d_ada, ada_ld, (int)dist_off, means, stds, xs, d2s, g1,
                     ld1, g2, ld2, dms, dd2_scratch, dpos);
  return DST_CHECK_LAUNCH();
}

int dst_attn_fwd(const dst_layout* L, const float* qkv, const float* te0, const float* te1, int64_t ld_te, const int32_t* adj, float* out,
                 float* alpha, void* stream) {
  if (!DST_L_OK(L) || !qkv || !te0 || !te1 || ld_te < 256 || !adj || !out || !alpha) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_attn_fwd, dim3(L->B), dim3(1024), 0, (hipStream_t)stream, *L, qkv, te0, te1, ld_te, adj, out, alpha);
  return DST_CHECK_LAUNCH();
}
int dst_attn_bwd(const dst_layout* L, const float* qkv, const float* te0, const float* te1, int64_t ld_te, const float* alpha, const float* dout,
                 float* dqkv, float* dte0, float* dte1, int32_t te_is_tanh, float* scratch, int64_t scratch_cap, void* stream) {
  if (!DST_L_OK(L) || !qkv || !te0 || !te1 || ld_te < 256 || !alpha || !dout || !dqkv || !dte0 || !dte1) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = (size_t)2 * 812 * 16 * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attn_bwd<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    attr_done = true;
  }
  if (scratch && scratch_cap >= (int64_t)2 * L->Pp * 16 && L->Pp > 0) {
    hipLaunchKernelGGL(k_attn_bwd<1>, dim3(L->B), dim3(1024), lds, s, *L, qkv, te0, te1, ld_te, alpha, dout, dqkv, dte0, dte1, (int)te_is_tanh, scratch);
    hipLaunchKernelGGL(k_attn_bwd<2>, dim3(L->B, 4), dim3(1024), lds / 2, s, *L, qkv, te0, te1, ld_te, alpha, dout, dqkv, dte0, dte1, (int)te_is_tanh, scratch);
  } else {
    hipLaunchKernelGGL(k_attn_bwd<0>, dim3(L->B), dim3(1024), lds, s, *L, qkv, te0, te1, ld_te, alpha, dout, dqkv, dte0, dte1, (int)te_is_tanh, (float*)nullptr);
  }
  return DST_CHECK_LAUNCH();
}

int dst_pair_sum_fwd(const dst_layout* L, const float* u, int32_t C, const float* bias, float* s, void* stream) {
  if (!DST_L_OK(L) || !u || !s || C <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_pair_sum_fwd, dim3(L->B, 4), dim3(256), 0, (hipStream_t)stream, *L, u, (int)C, bias, s);
  return DST_CHECK_LAUNCH();
}
int dst_pair_sum_bwd(const dst_layout* L, const float* ds, int32_t C, float* du, int32_t accumulate, void* stream) {
  if (!DST_L_OK(L) || !ds || !du || C <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_pair_sum_bwd, dim3(L->B, 4), dim3(256), 0, (hipStream_t)stream, *L, ds, (int)C, du, (int)accumulate);
  return DST_CHECK_LAUNCH();
}
int dst_zbuild_fwd(const dst_layout* L, const float* ac, const float* ed, float* z, void* stream) {
  if (!DST_L_OK(L) || !ac || !ed || !z) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_zbuild_fwd, dim3(L->B, 8), dim3(256), 0, (hipStream_t)stream, *L, ac, ed, z);
  return DST_CHECK_LAUNCH();
}
int dst_zbuild_bwd(const dst_layout* L, const float* dz, float* dac, float* ded, void* stream) {
  if (!DST_L_OK(L) || !dz || !dac || !ded) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_zbuild_bwd, dim3(L->B, 8), dim3(256), 0, (hipStream_t)stream, *L, dz, dac, ded);
  return DST_CHECK_LAUNCH();
}

int dst_coord_fwd(const dst_layout* L, const float* pos, const float* c2, const int32_t* adj, const float* coord_scale, float* pos_out,
                  void* stream) {
  if (!DST_L_OK(L) || !pos || !c2 || !adj || !coord_scale || !pos_out) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_coord_fwd, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, pos, c2, adj, coord_scale, pos_out);
  return DST_CHECK_LAUNCH();
}
int dst_coord_bwd(const dst_layout* L, const float* pos, const float* c2, const int32_t* adj, const float* coord_scale, const float* dpos_out,
                  float* dpos_in, float* dc2, float* dscale_part, void* stream) {
  if (!DST_L_OK(L) || !pos || !c2 || !adj || !coord_scale || !dpos_out || !dpos_in || !dc2 || !dscale_part) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_coord_bwd, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, pos, c2, adj, coord_scale, dpos_out, dpos_in, dc2, dscale_part);
  return DST_CHECK_LAUNCH();
}

int dst_time_feat_fwd(const float* noise_level, const float* w, int32_t B, float* f, void* stream) {
  if (!noise_level || !w || !f || B <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_time_feat_fwd, grid1d(B, 64), dim3(64), 0, (hipStream_t)stream, noise_level, w, (int)B, f);
  return DST_CHECK_LAUNCH();
}
int dst_time_feat_bwd(const float* noise_level, const float* w, const float* df, int32_t B, float* dw, void* stream) {
  if (!noise_level || !w || !df || !dw || B <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_time_feat_bwd, dim3(8), dim3(64), 0, (hipStream_t)stream, noise_level, w, df, (int)B, dw);
  return DST_CHECK_LAUNCH();
}

int dst_loss(const dst_layout* L, const float* pos, const float* feat, const float* edge, const float* tpos, const float* tfeat,
             const float* tedge, const float* wm, float w_pos, float w_type, float w_edge, float* loss_m, float* dpos, float* dfeat,
             float* dedge, void* stream) {
  if (!DST_L_OK(L) || !pos || !feat || !edge || !tpos || !tfeat || !tedge || !wm || !loss_m || !dpos || !dfeat || !dedge) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_loss, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, pos, feat, edge, tpos, tfeat, tedge, wm, w_pos, w_type, w_edge, loss_m,
                     dpos, dfeat, dedge);
  return DST_CHECK_LAUNCH();
}

int dst_noising(const dst_layout* L, const float* alpha, const float* sigma, const float* x, const float* raw, float* z, const float* ex,
                const float* eraw, float* ez, void* stream) {
  if (!DST_L_OK(L) || !alpha || !sigma || !x || !raw || !z || !ex || !eraw || !ez) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_noising, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, alpha, sigma, x, raw, z, ex, eraw, ez);
  return DST_CHECK_LAUNCH();
}

int dst_prepare_batch(const dst_layout* L, const float* pos, const float* one_hot, const float* fc, const float* edge, float pos_norm,
                      float type_norm, float fc_norm, float edge_norm, float* x, float* ex, void* stream) {
  if (!DST_L_OK(L) || !pos || !one_hot || !fc || !edge || !x || !ex) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_prepare_batch, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, pos, one_hot, fc, edge, pos_norm, type_norm, fc_norm, edge_norm, x, ex);
  return DST_CHECK_LAUNCH();
}

int dst_kabsch(const dst_layout* L, const float* pred, int64_t ld_pred, const float* tar, int64_t ld_tar, float* rot, float* aligned,
               void* stream) {
  if (!DST_L_OK(L) || !pred || !tar || !rot || !aligned) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_kabsch, dim3(L->B), dim3(64), 0, (hipStream_t)stream, *L, pred, ld_pred, tar, ld_tar, rot, aligned);
  return DST_CHECK_LAUNCH();
}

int dst_bn_fwd(const float* x, int32_t R, int32_t C, const float* gamma, const float* beta, float eps, float* y, float* stats,
               float* running_mean, float* running_var, float* scratch, int64_t scratch_cap, void* stream) {
  if (!x || !gamma || !beta || !y || !stats || !scratch || R < 2 || C <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int chunks = (R + 127) / 128;                              // 32 rows per thread: the loop is a chain of dependent loads, so short chains and many workgroups
  if (chunks > 4096) chunks = 4096;
  if ((int64_t)chunks * C > scratch_cap) chunks = (int)(scratch_cap / C);
  if (chunks < 1) return DS_ERR_ARG;
  const int rpc = (R + chunks - 1) / chunks;
  hipLaunchKernelGGL(k_bn_sum, dim3((C + 63) / 64, chunks), dim3(256), 0, s, x, (int)R, (int)C, (const float*)nullptr, scratch, rpc);
  hipLaunchKernelGGL(k_bn_finish_mean, dim3(C), dim3(64), 0, s, (const float*)scratch, chunks, (int)C, (int)R, stats);
  hipLaunchKernelGGL(k_bn_sum, dim3((C + 63) / 64, chunks), dim3(256), 0, s, x, (int)R, (int)C, (const float*)stats, scratch, rpc);
  hipLaunchKernelGGL(k_bn_finish_var, dim3(C), dim3(64), 0, s, (const float*)scratch, chunks, (int)C, (int)R, eps, stats, running_mean, running_var);
  hipLaunchKernelGGL(k_bn_apply, grid1d((int64_t)R * C), dim3(256), 0, s, x, (int64_t)R * C, (int)C, (const float*)stats, gamma, beta, y);
  return DST_CHECK_LAUNCH();
}
int dst_bn_running_again(const float* stats, int32_t C, float* running_mean, float* running_var, void* stream) {
  if (!stats || !running_mean || !running_var || C <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_bn_running_again, grid1d(C), dim3(256), 0, (hipStream_t)stream, stats, (int)C, running_mean, running_var);
  return DST_CHECK_LAUNCH();
}
int dst_bn_bwd(const float* dy, const float* x, const float* stats, int32_t R, int32_t C, const float* gamma, float* dx, float* dgamma,
               float* dbeta, float* scratch, int64_t scratch_cap, void* stream) {
  if (!dy || !x || !stats || !gamma || !dx || !dgamma || !dbeta || !scratch || R < 2 || C <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  int chunks = (R + 127) / 128;                              // 32 rows per thread: the loop is a chain of dependent loads, so short chains and many workgroups
  if (chunks > 4096) chunks = 4096;
  if ((int64_t)2 * chunks * C > scratch_cap) chunks = (int)(scratch_cap / (2 * C));
  if (chunks < 1) return DS_ERR_ARG;
  const int rpc = (R + chunks - 1) / chunks;
  hipLaunchKernelGGL(k_bn_bwd_sum, dim3((C + 63) / 64, chunks), dim3(256), 0, s, dy, x, stats, (int)R, (int)C, scratch, rpc, chunks);
  hipLaunchKernelGGL(k_bn_bwd_finish, dim3(C), dim3(64), 0, s, (const float*)scratch, chunks, (int)C, dgamma, dbeta);
  hipLaunchKernelGGL(k_bn_bwd_apply, grid1d((int64_t)R * C), dim3(256), 0, s, dy, x, (int64_t)R * C, (int)C, (int)R, stats, gamma, (const float*)dgamma,
                     (const float*)dbeta, dx);
  return DST_CHECK_LAUNCH();
}

int dst_spec_attn_fwd(const float* qkv, const float* prev, float* scores, float* stats, float* out, int32_t B, int32_t L, int32_t H,
                      int32_t dk, float scale, void* stream) {
  if (!qkv || !scores || !stats || !out || B <= 0 || L <= 0 || L > 512 || H <= 0 || dk != 8) return DS_ERR_ARG;   // d_k = 8 (specformer.py:19, d_model 128 / 16 heads)
  const size_t lds = (size_t)2 * L * dk * sizeof(float);
  if (lds > 64 * 1024) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_spec_attn_fwd<8>, dim3(DST_SPEC_SLICES, B * H), dim3(256), lds, (hipStream_t)stream, qkv, prev, scores, stats, out, (int)B, (int)L, (int)H, scale,
                     (int)((L + 31) / 32 * 32));
  return DST_CHECK_LAUNCH();
}
int dst_spec_attn_bwd(const float* qkv, const float* scores, const float* stats, const float* dout, const float* dscores_in, float* dqkv,
                      float* dscores, int32_t B, int32_t L, int32_t H, int32_t dk, float scale, void* stream) {
  if (!qkv || !scores || !stats || !dout || !dqkv || !dscores || B <= 0 || L <= 0 || L > 512 || H <= 0 || dk != 8) return DS_ERR_ARG;
  const size_t lds = (size_t)2 * L * dk * sizeof(float);
  if (lds > 64 * 1024) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int Lp = (L + 31) / 32 * 32;
  hipLaunchKernelGGL(k_spec_attn_bwd_q<8>, dim3(DST_SPEC_SLICES, B * H), dim3(256), lds, s, qkv, scores, stats, dout, dscores_in, dqkv, dscores, (int)B, (int)L, (int)H, scale, Lp);
  hipLaunchKernelGGL(k_spec_attn_bwd_kv<8>, dim3((L + 127) / 128, B * H), dim3(128), lds + (size_t)2 * L * sizeof(float), s, qkv, scores, stats, dout, (const float*)dscores, dqkv, (int)B, (int)L,
                     (int)H, scale, Lp);
  return DST_CHECK_LAUNCH();
}

int dst_ln_affine_fwd(const float* x, int32_t R, int32_t C, const float* gamma, const float* beta, float eps, float* y, float* stats,
                      void* stream) {
  if (!x || !gamma || !beta || !y || !stats || R <= 0 || C <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_ln_affine_fwd, dim3(R), dim3(64), 0, (hipStream_t)stream, x, (int)C, gamma, beta, eps, y, stats);
  return DST_CHECK_LAUNCH();
}
int dst_ln_affine_bwd(const float* dy, const float* x, const float* stats, int32_t R, int32_t C, const float* gamma, float* dx,
                      float* dgamma, float* dbeta, void* stream) {
  if (!dy || !x || !stats || !gamma || !dx || !dgamma || !dbeta || R <= 0 || C <= 0) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k_ln_affine_bwd, dim3(R), dim3(64), 0, s, dy, x, stats, (int)C, gamma, dx);
  hipLaunchKernelGGL(k_ln_affine_bwd_params, grid1d(C), dim3(256), 0, s, dy, x, stats, (int)R, (int)C, dgamma, dbeta);
  return DST_CHECK_LAUNCH();
}

int dst_adamw_ema(float* p, const float* g, float* m, float* v, float* vmax, float* ema, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float bc1, float bc2, float clip_coef, const float* clip_coef_dev, float ema_one_minus_decay,
                  void* stream) {
  if (!p || !g || !m || !v || !vmax || n < 0) return DS_ERR_ARG;
  if (n == 0) return DS_OK;
  auto al16 = [](const void* q) { return (reinterpret_cast<uintptr_t>(q) & 15) == 0; };
  if ((n & 3) == 0 && al16(p) && al16(g) && al16(m) && al16(v) && al16(vmax) && (!ema || al16(ema)))
    hipLaunchKernelGGL(k_adamw_ema4, grid1d(n >> 2), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<float4*>(p), reinterpret_cast<const float4*>(g),
                       reinterpret_cast<float4*>(m), reinterpret_cast<float4*>(v), reinterpret_cast<float4*>(vmax), reinterpret_cast<float4*>(ema), n >> 2, lr,
                       beta1, beta2, eps, weight_decay, bc1, bc2, clip_coef, clip_coef_dev, ema_one_minus_decay);
  else
    hipLaunchKernelGGL(k_adamw_ema, grid1d(n), dim3(256), 0, (hipStream_t)stream, p, g, m, v, vmax, ema, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                       clip_coef, clip_coef_dev, ema_one_minus_decay);
  return DST_CHECK_LAUNCH();
}

int dst_clip_update(const float* norm_sq, float inv_world, float max_grad, float* state, void* stream) {
  if (!norm_sq || !state || !(max_grad >= 0.0f) || !(inv_world > 0.0f)) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_clip_update, dim3(1), dim3(64), 0, (hipStream_t)stream, norm_sq, inv_world, max_grad, state);
  return DST_CHECK_LAUNCH();
}

}  // extern "C"
