// diffspectra_amd - the GEMM of the TRAINING path (include/diffspectra_train.h: dst_gemm), gfx950.
//
// Every nn.Linear forward, input gradient and weight gradient of the DMT / SpecFormer training graph is one call of dst_gemm with
// element strides (a transpose is a view).  Two kernels:
//
//  * k_tr_gemm_bf16 (config 5, args.bf16): 128 x BN tiles, BK = 32.  Operands are fp32 in HBM (master weights, fp32 tape); a tile is
//    fetched with 16-byte loads one k-step ahead (registers), rounded to bf16 ONCE as it is written to LDS (v_cvt_pk_bf16_f32) in the
//    k-contiguous order of the MFMA operands - an operand whose memory order is row-contiguous (dY^T and X of a weight gradient, W of
//    an input gradient) is transposed in registers, 4 x 4 at a time, on its way in - so a fragment of v_mfma_f32_32x32x16_bf16 is one
//    ds_read_b128 and the MFMA loop holds no conversion (round 3's kernel kept fp32 in LDS and converted eight values per lane per
//    MFMA).  LDS rows are 80 bytes (32 bf16 + 16 bytes of padding): the sixteen lanes of a ds_read_b128 group then hit sixteen
//    different 16-byte slots of the 256-byte bank row.
//  * k_tr_gemm_big: round 3's fp32-in-LDS kernel - the fp32 mode (v_mfma_f32_32x32x2_f32: the arithmetic golden G13 pins) and the
//    fallback for operands that are not 16-byte aligned (K = 17 time features, 3-wide coordinate heads).
//
// Both end in the same fused epilogue (bias, activation, activation derivative, Philox dropout mask, second output).  A long-K product
// (weight gradients: K = rows) is split over K: every slice writes its fp32 slab and k_tr_gemm_reduce adds the slabs in slice order
// 0 .. S-1 and runs the epilogue - a fixed summation order.  (Tried and measured on the MI355X: the reduction inside the GEMM launch by
// the slice that arrives last at a per-tile counter - one workgroup then reads 128 slabs of its tile serially, 1.2 ms for a 256 x 256
// weight gradient over 80 800 rows against ~0.1 ms with the reduction spread over the chip.)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/diffspectra_hip.h"
#include "../../include/diffspectra_train.h"
#include "ds_train_common.h"

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

namespace {

#define DST_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DS_OK : DS_ERR_LAUNCH)

// ------------------------------------------------------------------------------------------------------------------ epilogue
// One output element.  v = acc + bias; v *= f'(ref) (dact); act: C <- v (pre-activation) and C2 <- drop(f(v)) when C2 is given, else
// C <- drop(f(v)); no act: C (+)= drop(v).  Column N is the fused row sum (the virtual all-ones column of B).
// The activation / dropout form is long (transcendentals, ten Philox rounds): inlined at every one of a lane's 32 - 64 accumulator
// registers it made the kernels instruction-cache bound (a 128 x 128 tile ran 3x slower than a 128 x 64 one), and as a real call it
// forced the argument struct into scratch memory (every kernel 2x slower).  So finish_tiles runs it in a ROLLED loop over a tile staged
// through wave-private LDS: one copy of the code per kernel.
__device__ __forceinline__ void epi_fused(const dst_gemm_args& g, int row, int col, float v) {
  if (g.dact == 4) v += g.ref[(int64_t)row * g.ldref + col];                   // dact 4: + ref (a residual operand read where the sum is formed)
  else if (g.dact) v *= dst::act_deriv(g.ref[(int64_t)row * g.ldref + col], g.dact);
  float keep = 1.0f;
  if (g.drop_p > 0.0f)
    keep = dst::dropout_keep(g.drop_seed, g.drop_stream, (int64_t)row * g.drop_ld + col, dst::dropout_threshold(g.drop_p)) ? 1.0f / (1.0f - g.drop_p) : 0.0f;
  float* c = g.C + (int64_t)row * g.ldc + col;
  if (g.act) {
    if (g.C2) {
      *c = v;
      g.C2[(int64_t)row * g.ldc2 + col] = dst::act_apply(v, g.act) * keep;
      return;
    }
    v = dst::act_apply(v, g.act);
  }
  v *= keep;
  if (g.accumulate) v += *c;
  *c = v;
}
// Four consecutive columns (col % 4 == 0, all inside N) at once: 16-byte accesses and ONE Philox block for the four masks (the scalar form
// runs the ten rounds once per element).  dst_gemm sets g._pad when every pointer and stride of the epilogue allows it.
__device__ __forceinline__ void epi_fused4(const dst_gemm_args& g, int row, int col, f32x4_t v) {
  if (g.dact) {
    const f32x4_t r = *reinterpret_cast<const f32x4_t*>(g.ref + (int64_t)row * g.ldref + col);
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = g.dact == 4 ? v[e] + r[e] : v[e] * dst::act_deriv(r[e], g.dact);
  }
  f32x4_t keep = {1.0f, 1.0f, 1.0f, 1.0f};
  if (g.drop_p > 0.0f) {
    unsigned int c[4];
    dst::dropout_block(g.drop_seed, g.drop_stream, ((int64_t)row * g.drop_ld + col) >> 2, c);
    const unsigned int thr = dst::dropout_threshold(g.drop_p);
    const float inv = 1.0f / (1.0f - g.drop_p);
#pragma unroll
    for (int e = 0; e < 4; ++e) keep[e] = c[e] >= thr ? inv : 0.0f;
  }
  f32x4_t* cp = reinterpret_cast<f32x4_t*>(g.C + (int64_t)row * g.ldc + col);
  if (g.act) {
    if (g.C2) {
      *cp = v;
      f32x4_t w;
#pragma unroll
      for (int e = 0; e < 4; ++e) w[e] = dst::act_apply(v[e], g.act) * keep[e];
      *reinterpret_cast<f32x4_t*>(g.C2 + (int64_t)row * g.ldc2 + col) = w;
      return;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = dst::act_apply(v[e], g.act);
  }
  v = v * keep;
  if (g.accumulate) v = v + *cp;
  *cp = v;
}
__device__ __forceinline__ void epi_plain(const dst_gemm_args& g, int row, int col, float acc) {
  if (col == g.N) {
    g.rowsum[row] = g.accumulate ? g.rowsum[row] + acc : acc;
    return;
  }
  const float v = acc + (g.bias ? g.bias[col] : 0.0f);
  float* c = g.C + (int64_t)row * g.ldc + col;
  *c = g.accumulate ? v + *c : v;
}

constexpr int STAGE_LD = 36;                                   // floats per row of a wave's 32 x 32 staging tile (16-byte rows for the vector epilogue)
constexpr int STAGE_BYTES = 4 * 32 * STAGE_LD * 4;             // four waves

// The wave's TM x TN accumulator tiles (32 x 32 each; rows rbase + 32 i, columns cbase + 32 j) -> memory: slice z's fp32 slab when
// the product is split over K (k_tr_gemm_reduce adds the slabs in slice order and runs the epilogue), else the output - directly
// from the registers for the plain epilogue, through `stage` (wave-private LDS, the operand tiles are dead by now) for the fused one.
template <int TM, int TN>
__device__ __forceinline__ void finish_tiles(const dst_gemm_args& g, f32x16_t (&acc)[TM][TN], int rbase, int cbase, int splits, int z, float* stage) {
  const int lane = threadIdx.x & 63;
  const int Nx = g.N + (g.rowsum ? 1 : 0);
  const bool fused = g.act || g.dact || g.drop_p > 0.0f;
  if (splits > 1 || (!fused && !(g._pad & 2))) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int col = cbase + j * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + i * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
          if (row < g.M && col < Nx) {
            if (splits > 1) g.partial[((int64_t)z * g.M + row) * Nx + col] = acc[i][j][r];
            else epi_plain(g, row, col, acc[i][j][r]);
          }
        }
      }
    return;
  }
  float* st = stage + (threadIdx.x >> 6) * (32 * STAGE_LD);
#pragma unroll 1
  for (int t = 0; t < TM * TN; ++t) {
    f32x16_t sel = acc[0][0];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        if (i * TN + j == t) sel = acc[i][j];
#pragma unroll
    for (int r = 0; r < 16; ++r) st[((r >> 2) * 8 + (lane >> 5) * 4 + (r & 3)) * STAGE_LD + (lane & 31)] = sel[r];
    const int ti = t / TN, tj = t % TN;
    if (g._pad & 1) {                                                         // vector form: a lane takes four columns of rows lr, lr + 8, lr + 16, lr + 24
      const int c4 = 4 * (lane & 7), col4 = cbase + tj * 32 + c4;
      f32x4_t b4 = {0.0f, 0.0f, 0.0f, 0.0f};
      if (g.bias && col4 < g.N) b4 = *reinterpret_cast<const f32x4_t*>(g.bias + col4);
#pragma unroll 2
      for (int e = 0; e < 4; ++e) {
        const int lr = 8 * e + (lane >> 3), row = rbase + ti * 32 + lr;
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(st + lr * STAGE_LD + c4);
        if (row < g.M && col4 < g.N) epi_fused4(g, row, col4, a + b4);
      }
      continue;
    }
    const int col = cbase + tj * 32 + (lane & 31);
#pragma unroll 4
    for (int e = 0; e < 16; ++e) {                                            // four elements in flight: the loop is latency-bound (LDS read, ref load, stores)
      const int lr = 2 * e + (lane >> 5);
      const int row = rbase + ti * 32 + lr;
      const float a = st[lr * STAGE_LD + (lane & 31)];
      if (row < g.M && col < Nx) {
        if (col == g.N) epi_plain(g, row, col, a);
        else epi_fused(g, row, col, a + (g.bias ? g.bias[col] : 0.0f));
      }
    }
  }
}

// Workgroup -> (k-slice, m-tile, n-tile).  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so linear ids
// L, L + 8, L + 16 .. share an L2: those walk the `inner` tiles that read the same operand rows - the n-tiles of one m-tile (A rows;
// without it a 256-wide output fetched A four times from HBM, once per XCD), or every tile of one k-slice of a split product.
// Speed only: any placement computes the same result.  false = padding workgroup (outer index beyond the range).
__device__ __forceinline__ bool tile_of_block(int tm, int tn, int splits, int& z, int& mt, int& nt) {
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int inner = splits > 1 ? tm * tn : tn, outer_n = splits > 1 ? splits : tm;
  const int outer = (slot / inner) * 8 + xcd, in = slot % inner;
  if (outer >= outer_n) return false;
  if (splits > 1) { z = outer; mt = in / tn; nt = in % tn; }
  else { z = 0; mt = outer; nt = in; }
  return true;
}

// sum of the k-slices' slabs in slice order, then the epilogue: one element per thread (a 256 x 256 weight gradient is 256 workgroups:
// every CU takes part; with four elements per thread the 64 workgroups of the same reduction took 3x as long), eight slabs in flight
__global__ __launch_bounds__(256) void k_tr_gemm_reduce(dst_gemm_args g, int splits) {
  const int Nx = g.N + (g.rowsum ? 1 : 0);
  const int64_t total = (int64_t)g.M * Nx;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const bool fused = g.act || g.dact || g.drop_p > 0.0f;
  const float* p = g.partial + i;
  float v = 0.0f;
  int z = 0;
  for (; z + 8 <= splits; z += 8) {
    float t[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) t[e] = p[(int64_t)(z + e) * total];
#pragma unroll
    for (int e = 0; e < 8; ++e) v += t[e];
  }
  for (; z < splits; ++z) v += p[(int64_t)z * total];
  const int row = (int)(i / Nx), col = (int)(i % Nx);
  if (!fused || col == g.N) epi_plain(g, row, col, v);
  else epi_fused(g, row, col, v + (g.bias ? g.bias[col] : 0.0f));
}

// ------------------------------------------------------------------------------------------------------------------ bf16 kernel
constexpr int BK = 32;       // k-step
constexpr int LDK = 40;      // bf16 elements per LDS row: 64 bytes of operand + 16 of padding

__device__ __forceinline__ unsigned int pack2(float a, float b) {
  const bf16x2_t v = {(__bf16)a, (__bf16)b};
  return __builtin_bit_cast(unsigned int, v);
}

// Fetch one ROWS x 32 operand tile into registers.  The operand is addressed as X[row * rs + k * ks] with either ks == 1 ("k-fast": a
// row of the tile is 128 contiguous bytes; vec4 u of 256-thread pass i covers row u >> 3, k 4 (u & 7) ..) or rs == 1 ("row-fast": the
// memory order runs along the rows; a thread takes a 4 (rows) x 4 (k) micro-tile, four 16-byte loads, transposed when committed).
// Out-of-range elements are zero.  ones_row (B only): that row of the tile is the virtual all-ones vector of the fused row sum.
// BRANCH-FREE: every load is a full 16-byte load from a clamped, always-valid address and the edges are applied by selects.  (With a
// branch per edge case the compiler closed every load with s_waitcnt vmcnt(0) at the join: the loads of a k-step ran one memory
// round trip after the other - 0.9 us per step for two MFMAs - and the two steps of prefetch hid nothing.)  Validity of the clamped
// loads: the leading stride of a vector operand is a multiple of 4 and >= its extent (dst_gemm checks), so the aligned group that
// holds the last valid element lies inside the stride of its row.
// Both forms give a thread NV = ROWS * 8 / NTHR vectors (1, 2 or 4), all of them loaded on every path - an array with elements that one
// path leaves undefined ends up in scratch, and zero-filling it first puts a full s_waitcnt in front of the loads (write-after-write on
// registers with loads in flight).  Row-fast: lane (kg, rq) of k-selection ksel holds k = 4 kg + NV ksel .. + NV - 1 of rows 4 rq .. 4 rq + 3.
template <int ROWS, int NTHR = 256>
__device__ __forceinline__ void fetch_tile(const float* __restrict__ X, int64_t rs, int64_t ks, bool rfast, int row0, int R, int K, int k0,
                                           f32x4_t (&v)[ROWS * 8 / NTHR]) {
  constexpr int NV = ROWS * 8 / NTHR;
  const int tid = threadIdx.x;
  if (!rfast) {
    const int klast = (K - 1) & ~3;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int u = tid + i * NTHR;
      const int gr = row0 + (u >> 3), gk = k0 + 4 * (u & 7);
      v[i] = *reinterpret_cast<const f32x4_t*>(X + (int64_t)min(gr, R - 1) * rs + min(gk, klast));
    }
  } else {
    const int kg = tid & 7, rq = (tid >> 3) % (ROWS / 4), ksel = tid / (ROWS * 2);
    const int gr0 = min(row0 + 4 * rq, (R - 1) & ~3);
#pragma unroll
    for (int kk = 0; kk < NV; ++kk) {
      const int gk = k0 + 4 * kg + NV * ksel + kk;
      v[kk] = *reinterpret_cast<const f32x4_t*>(X + (int64_t)min(gk, K - 1) * ks + gr0);
    }
  }
}

// Round the fetched tile to bf16 into LDS ([row][k], LDK elements per row); the edges (rows >= R, k >= kend, the virtual ones row) are
// applied HERE, a k-step or two after the loads were issued - a select next to the load would make the load's latency part of the step.
template <int ROWS, int NTHR = 256>
__device__ __forceinline__ void commit_tile(unsigned short* __restrict__ Xs, bool rfast, const f32x4_t (&v)[ROWS * 8 / NTHR], int row0, int R, int k0,
                                            int kend, int ones_row) {
  constexpr int NV = ROWS * 8 / NTHR;
  const int tid = threadIdx.x;
  if (!rfast) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int u = tid + i * NTHR;
      const int gr = row0 + (u >> 3), gk = k0 + 4 * (u & 7);
      const float edge = gr == ones_row ? 1.0f : 0.0f;
      const bool rok = gr < R;
      float t[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) t[e] = gk + e < kend ? (rok ? v[i][e] : edge) : 0.0f;
      const uint2 w = {pack2(t[0], t[1]), pack2(t[2], t[3])};
      *reinterpret_cast<uint2*>(Xs + (u >> 3) * LDK + 4 * (u & 7)) = w;
    }
  } else {
    const int kg = tid & 7, rq = (tid >> 3) % (ROWS / 4), ksel = tid / (ROWS * 2);
    const int gr0 = row0 + 4 * rq, gk0 = k0 + 4 * kg + NV * ksel;
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const bool rok = gr0 + rr < R;
      const float edge = gr0 + rr == ones_row ? 1.0f : 0.0f;
      float t[NV];
#pragma unroll
      for (int kk = 0; kk < NV; ++kk) t[kk] = gk0 + kk < kend ? (rok ? v[kk][rr] : edge) : 0.0f;
      unsigned short* dst = Xs + (4 * rq + rr) * LDK + 4 * kg + NV * ksel;
      if constexpr (NV == 4) {
        const uint2 w = {pack2(t[0], t[1]), pack2(t[2], t[3])};
        *reinterpret_cast<uint2*>(dst) = w;
      } else if constexpr (NV == 2) {
        *reinterpret_cast<unsigned int*>(dst) = pack2(t[0], t[1]);
      } else {
        const __bf16 h = (__bf16)t[0];
        *dst = __builtin_bit_cast(unsigned short, h);
      }
    }
  }
}

// NTHR = 256: 2 x 2 waves (BM, BN in {64, 128}).  NTHR = 512: 4 x 2 waves on a 256-row tile - the weight-gradient form: a whole
// 256 x 256 (x 128, x 64) output per workgroup, so that a k-slice of dY^T and X is read ONCE (eight 128 x 64 tiles re-read the slice's
// rows four and two times), one workgroup per CU, the k-range split over the CUs.
template <int BM, int BN, int NTHR = 256>
__global__ __launch_bounds__(NTHR) void k_tr_gemm_bf16(dst_gemm_args g, int splits, int kchunk, int a_rfast, int b_rfast, int tm, int tn) {
  constexpr int WM = NTHR / 128, WN = 2;                    // waves along M and N
  constexpr int TM = BM / (32 * WM), TN = BN / (32 * WN);   // 32 x 32 tiles per wave
  constexpr int STG = STAGE_BYTES * (NTHR / 256);
  constexpr int LDS_BYTES = (BM + BN) * LDK * 2 > STG ? (BM + BN) * LDK * 2 : STG;
  __shared__ __attribute__((aligned(16))) unsigned short lds[LDS_BYTES / 2];
  unsigned short* As = lds;
  unsigned short* Bs = lds + BM * LDK;
  int z, mt, nt;
  if (!tile_of_block(tm, tn, splits, z, mt, nt)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave / WN, wn = wave % WN;
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = z * kchunk;
  const int kend = min(g.K, kbeg + kchunk);
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  // A[m, k] = A[m * a_rs + k * a_cs]; B[k, n] = B[k * b_rs + n * b_cs]: as a (row = n, k) operand its row stride is b_cs, its k stride b_rs
  const int ones_row = g.rowsum ? g.N : -1;
  // two k-steps of operands in flight (two register sets, the loop unrolled by two so that each set is addressed statically): a
  // workgroup that is alone on its CU - split products have ~1 per CU - otherwise exposes a full memory round trip per step
  f32x4_t ra0[BM * 8 / NTHR], rb0[BN * 8 / NTHR], ra1[BM * 8 / NTHR], rb1[BN * 8 / NTHR];
  const bool arf = a_rfast != 0, brf = b_rfast != 0;
  // (no guards: fetch_tile clamps every address into the operand, commit_tile zeroes what lies beyond kend - a k-range with an odd
  // number of steps runs one all-zero step, and the compiler sees ONE straight-line loop body whose waits it can count)
  fetch_tile<BM, NTHR>(g.A, g.a_rs, g.a_cs, arf, m0, g.M, g.K, kbeg, ra0);
  fetch_tile<BN, NTHR>(g.B, g.b_cs, g.b_rs, brf, n0, g.N, g.K, kbeg, rb0);
  fetch_tile<BM, NTHR>(g.A, g.a_rs, g.a_cs, arf, m0, g.M, g.K, kbeg + BK, ra1);
  fetch_tile<BN, NTHR>(g.B, g.b_cs, g.b_rs, brf, n0, g.N, g.K, kbeg + BK, rb1);
  const int arow = (wm * (BM / WM) + (lane & 31)) * LDK + 8 * (lane >> 5);
  const int brow = (wn * (BN / WN) + (lane & 31)) * LDK + 8 * (lane >> 5);
  auto step = [&](f32x4_t (&xa)[BM * 8 / NTHR], f32x4_t (&xb)[BN * 8 / NTHR], int k0) __attribute__((always_inline)) {
    commit_tile<BM, NTHR>(As, arf, xa, m0, g.M, k0, kend, -1);
    commit_tile<BN, NTHR>(Bs, brf, xb, n0, g.N, k0, kend, ones_row);
    __syncthreads();
    fetch_tile<BM, NTHR>(g.A, g.a_rs, g.a_cs, arf, m0, g.M, g.K, k0 + 2 * BK, xa);
    fetch_tile<BN, NTHR>(g.B, g.b_cs, g.b_rs, brf, n0, g.N, g.K, k0 + 2 * BK, xb);
    __builtin_amdgcn_sched_barrier(0);          // the loads stay ahead of the step's MFMAs
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8_t a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const bf16x8_t*>(As + arow + i * 32 * LDK + ks * 16);
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const bf16x8_t*>(Bs + brow + j * 32 * LDK + ks * 16);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  };
  for (int k0 = kbeg; k0 < kend; k0 += 2 * BK) {
    step(ra0, rb0, k0);
    step(ra1, rb1, k0 + BK);
  }
  finish_tiles<TM, TN>(g, acc, m0 + wm * (BM / WM), n0 + wn * (BN / WN), splits, z, reinterpret_cast<float*>(lds));
}

// ------------------------------------------------------------------------------------------------------------------ fp32 / unaligned kernel
// BM x BN per 256-thread workgroup (2 x 2 waves, each (BM/2) x (BN/2) as 32 x 32 MFMA tiles), BK = 16, the next k-slab's global
// loads in flight (registers) while the current one is multiplied out of LDS; scalar loads by element strides, guards on all three
// dimensions.  BF16: operands rounded to bf16 as they leave LDS (the fallback of k_tr_gemm_bf16 for unaligned operands).
template <int BM, int BN, bool BF16>
__global__ __launch_bounds__(256) void k_tr_gemm_big(dst_gemm_args g, int splits, int kchunk, int tm, int tn) {
  constexpr int TM = BM / 64, TN = BN / 64;              // 32 x 32 tiles per wave in each direction (BM, BN in {64, 128})
  constexpr int LA = BM * 16 / 256, LB = BN * 16 / 256;  // elements per thread per slab
  constexpr int OPER = 16 * (BM + 4 + BN + 4);                     // floats of the two operand slabs; the staging tiles of the fused epilogue reuse them
  __shared__ __attribute__((aligned(16))) float lds[OPER * 4 > STAGE_BYTES ? OPER : STAGE_BYTES / 4];
  float (*As)[BM + 4] = reinterpret_cast<float (*)[BM + 4]>(lds);
  float (*Bs)[BN + 4] = reinterpret_cast<float (*)[BN + 4]>(lds + 16 * (BM + 4));
  int z, mt, nt;
  if (!tile_of_block(tm, tn, splits, z, mt, nt)) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = mt * BM, n0 = nt * BN;
  const int kbeg = z * kchunk;
  const int kend = min(g.K, kbeg + kchunk);
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const bool a_kfast = (g.a_cs == 1), b_nfast = (g.b_cs == 1);
  float ra[LA], rb[LB];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      int mm, kk;
      if (a_kfast) { kk = e & 15; mm = e >> 4; } else { mm = e % BM; kk = e / BM; }
      const int gm = m0 + mm, gk = k0 + kk;
      ra[i] = (gm < g.M && gk < kend) ? g.A[(int64_t)gm * g.a_rs + (int64_t)gk * g.a_cs] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      int nn, kk;
      if (b_nfast) { nn = e % BN; kk = e / BN; } else { kk = e & 15; nn = e >> 4; }
      const int gn = n0 + nn, gk = k0 + kk;
      // column N is the virtual all-ones column of the fused row sum (bias gradient of a weight-gradient product)
      rb[i] = gk < kend ? (gn < g.N ? g.B[(int64_t)gk * g.b_rs + (int64_t)gn * g.b_cs] : ((g.rowsum && gn == g.N) ? 1.0f : 0.0f)) : 0.0f;
    }
  };
  auto commit = [&]() {
#pragma unroll
    for (int i = 0; i < LA; ++i) {
      const int e = tid + i * 256;
      int mm, kk;
      if (a_kfast) { kk = e & 15; mm = e >> 4; } else { mm = e % BM; kk = e / BM; }
      As[kk][mm] = ra[i];
    }
#pragma unroll
    for (int i = 0; i < LB; ++i) {
      const int e = tid + i * 256;
      int nn, kk;
      if (b_nfast) { nn = e % BN; kk = e / BN; } else { kk = e & 15; nn = e >> 4; }
      Bs[kk][nn] = rb[i];
    }
  };
  if (kbeg < kend) fetch(kbeg);
  for (int k0 = kbeg; k0 < kend; k0 += 16) {
    commit();
    __syncthreads();
    if (k0 + 16 < kend) fetch(k0 + 16);
    if constexpr (BF16) {
      bf16x8_t a8[TM], b8[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int t = 0; t < 8; ++t) a8[i][t] = (__bf16)As[8 * (lane >> 5) + t][wm * (BM / 2) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int t = 0; t < 8; ++t) b8[j][t] = (__bf16)Bs[8 * (lane >> 5) + t][wn * (BN / 2) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[i], b8[j], acc[i][j], 0, 0, 0);
    } else
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      float a[TM], b[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) a[i] = As[kk + (lane >> 5)][wm * (BM / 2) + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[j] = Bs[kk + (lane >> 5)][wn * (BN / 2) + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  finish_tiles<TM, TN>(g, acc, m0 + wm * (BM / 2), n0 + wn * (BN / 2), splits, z, lds);
}

// ------------------------------------------------------------------------------------------------------------------ weight-stationary kernel
// C[M, N] = A[M, K] B[K, N] for the tall products of the step (M = node / pair / directed / token rows, 4 600 .. 89 000; K <= 512,
// N <= 512): every Linear forward and input gradient.  In the tiled kernel above such a product re-stages the same small weight in
// every workgroup and k-step, and a 128 x 64 tile exchanges operands through LDS eight times for 0.4 us of matrix work - measured inside
// a training step it moved its bytes at 1.7 - 2.2 TB/s.  Here the WEIGHT is staged once per workgroup (bf16, row n = output column, k
// contiguous, 16 bytes of padding per row: conflict-free ds_read_b128 B fragments) and stays; each of the eight waves then streams
// 32-row tiles of A straight from global memory into A fragments (a lane reads the 32 bytes of its row per k-step; chunks of eight
// k-steps, the next chunk's 16 kB per wave in flight under the current chunk's MFMAs), multiplies them against the resident weight
// (NCT column tiles of 32) and writes whole 128-byte output segments through the common epilogue.  No barrier after the weight staging.
// Wide outputs are split into chunks of 32 NCT columns over neighbouring workgroups of one XCD (tile_of_block's placement idea).
constexpr int WS_KS = 8;   // k-steps (of 16) per chunk of A
template <int NCT>
__global__ __launch_bounds__(512) void k_tr_gemm_ws(dst_gemm_args g, int b_rfast, int Kp, int nchunks, int ngroups) {
  extern __shared__ __attribute__((aligned(16))) unsigned short wlds[];
  constexpr int NC = 32 * NCT;
  const int WLD = Kp + 8;                                   // bf16 per weight row
  unsigned short* Wt = wlds;
  float* stage = reinterpret_cast<float*>(wlds + NC * WLD);
  const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
  const int grp = (slot / nchunks) * 8 + xcd, chunk = slot % nchunks;
  if (grp >= ngroups) return;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int n0 = chunk * NC;
  const int kq = Kp >> 2;
  if (!b_rfast) {                                           // B[k, n] = B[n * b_cs + k]: rows of W are k-contiguous
    for (int u = tid; u < NC * kq; u += 512) {
      const int n = u / kq, k = 4 * (u % kq);
      f32x4_t t = {0.0f, 0.0f, 0.0f, 0.0f};
      if (n0 + n < g.N && k < g.K) t = *reinterpret_cast<const f32x4_t*>(g.B + (int64_t)(n0 + n) * g.b_cs + k);   // K % 4 == 0
      const uint2 w = {pack2(t[0], t[1]), pack2(t[2], t[3])};
      *reinterpret_cast<uint2*>(Wt + n * WLD + k) = w;
    }
  } else {                                                  // B[k, n] = B[k * b_rs + n]: 4 x 4 micro-tiles, transposed in registers
    constexpr int NQ = NC / 4;
    for (int u = tid; u < NQ * kq; u += 512) {
      const int nq = u % NQ, k = 4 * (u / NQ), n = n0 + 4 * nq;
      f32x4_t t[4];
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        t[kk] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        if (k + kk < g.K) {
          const float* p = g.B + (int64_t)(k + kk) * g.b_rs + n;
          if (n + 3 < g.N) t[kk] = *reinterpret_cast<const f32x4_t*>(p);
          else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
              if (n + e < g.N) t[kk][e] = p[e];
          }
        }
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const uint2 w = {pack2(t[0][rr], t[1][rr]), pack2(t[2][rr], t[3][rr])};
        *reinterpret_cast<uint2*>(Wt + (4 * nq + rr) * WLD + k) = w;
      }
    }
  }
  __syncthreads();
  const int nsteps = Kp >> 4, nch = (nsteps + WS_KS - 1) / WS_KS;
  const int ntiles = (g.M + 31) >> 5;
  f32x4_t raw[WS_KS][2];
  // branch-free (see fetch_tile): K is a multiple of 128 here (dst_gemm), so every k-step of every chunk is a full, valid load and the
  // sixteen loads of a chunk share one base address with immediate offsets
  auto issue = [&](int tile, int kc) __attribute__((always_inline)) {   // this lane's 32 bytes per k-step of chunk kc of its row
    const int row = min(tile * 32 + r, g.M - 1);
    const float* ap = g.A + (int64_t)row * g.a_rs + 8 * hh + 16 * WS_KS * kc;
#pragma unroll
    for (int s = 0; s < WS_KS; ++s) {
      raw[s][0] = *reinterpret_cast<const f32x4_t*>(ap + 16 * s);
      raw[s][1] = *reinterpret_cast<const f32x4_t*>(ap + 16 * s + 4);
    }
  };
  const int t0 = grp * 8 + wave, tstride = ngroups * 8;
  if (t0 < ntiles) issue(t0, 0);
  const unsigned short* wrow = Wt + r * WLD + 8 * hh;
  for (int tile = t0; tile < ntiles; tile += tstride) {
    f32x16_t acc[1][NCT];
#pragma unroll
    for (int j = 0; j < NCT; ++j)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][j][i] = 0.0f;
    for (int kc = 0; kc < nch; ++kc) {
      bf16x8_t a[WS_KS];
#pragma unroll
      for (int s = 0; s < WS_KS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) { a[s][e] = (__bf16)raw[s][0][e]; a[s][4 + e] = (__bf16)raw[s][1][e]; }
      {   // ONE load site (two sites and a no-load path made the register allocator copy the whole chunk and wait on it): after the
          // last chunk of the last tile the first chunk of that tile is simply loaded again
        const bool wrap = kc + 1 == nch;
        issue(wrap ? (tile + tstride < ntiles ? tile + tstride : tile) : tile, wrap ? 0 : kc + 1);
        __builtin_amdgcn_sched_barrier(0);      // the loads stay AHEAD of the chunk's MFMAs (the scheduler sank them to the end of the block)
      }
#pragma unroll
      for (int s = 0; s < WS_KS; ++s)
#pragma unroll
        for (int j = 0; j < NCT; ++j)
          acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[s], *reinterpret_cast<const bf16x8_t*>(wrow + j * 32 * WLD + 16 * (kc * WS_KS + s)), acc[0][j], 0, 0, 0);
    }
    finish_tiles<1, NCT>(g, acc, tile * 32, n0, 1, 0, stage);
  }
}

// ------------------------------------------------------------------------------------------------------------------ K <= 8
// C[M, N] = A[M, K] B[K, N] with a handful of k (coord_mlp.2: 256 -> 3, its input gradient is an 81 000 x 256 output from K = 3; the
// one-hot / distance embeddings): an outer-product-sized job that a 32-k-step MFMA tile spends on padding (108 us for that product).
// One thread per output element, consecutive threads on consecutive columns; fp32 FMAs in k order; the common epilogue.
__global__ __launch_bounds__(256) void k_tr_gemm_skinny(dst_gemm_args g) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)g.M * g.N) return;
  const int row = (int)(i / g.N), col = (int)(i - (int64_t)row * g.N);
  const float* a = g.A + (int64_t)row * g.a_rs;
  const float* b = g.B + (int64_t)col * g.b_cs;
  float v = 0.0f;
  for (int k = 0; k < g.K; ++k) v += a[(int64_t)k * g.a_cs] * b[(int64_t)k * g.b_rs];
  if (g.act || g.dact || g.drop_p > 0.0f) epi_fused(g, row, col, v + (g.bias ? g.bias[col] : 0.0f));
  else epi_plain(g, row, col, v);
}

// the same with four consecutive columns per thread (B rows contiguous along n, every epilogue operand 16-byte friendly: g._pad)
__global__ __launch_bounds__(256) void k_tr_gemm_skinny4(dst_gemm_args g) {
  const int n4 = g.N >> 2;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)g.M * n4) return;
  const int row = (int)(i / n4), col = (int)(i - (int64_t)row * n4) << 2;
  const float* a = g.A + (int64_t)row * g.a_rs;
  const float* b = g.B + col;
  f32x4_t v = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int k = 0; k < g.K; ++k) v = v + a[(int64_t)k * g.a_cs] * *reinterpret_cast<const f32x4_t*>(b + (int64_t)k * g.b_rs);
  if (g.bias) v = v + *reinterpret_cast<const f32x4_t*>(g.bias + col);
  epi_fused4(g, row, col, v);
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}

}  // namespace

extern "C" {

int dst_struct_sizes(int64_t* out) {
  if (!out) return DS_ERR_ARG;
  out[0] = sizeof(dst_gemm_args);
  out[1] = sizeof(dst_layout);
  out[2] = sizeof(dst_piece);
  return DS_OK;
}

int dst_gemm(const dst_gemm_args* a, void* stream) {
  if (!a || !a->C || a->M < 0 || a->N < 0 || a->K < 0 || (a->K > 0 && (!a->A || !a->B))) return DS_ERR_ARG;
  if ((a->dact && !a->ref) || a->act < 0 || a->act > 3 || a->dact < 0 || a->dact > 4 || !(a->drop_p >= 0.0f && a->drop_p < 1.0f)) return DS_ERR_ARG;
  if (a->act && a->accumulate) return DS_ERR_ARG;                       // an activated output is written, never accumulated
  if (a->M == 0 || (a->N == 0 && !a->rowsum)) return DS_OK;
  hipStream_t s = (hipStream_t)stream;
  dst_gemm_args g = *a;
  {   // vector epilogue (epi_fused4): every pointer of the fused epilogue 16-byte aligned, every stride and N a multiple of 4, no fused row sum
    static const bool vec_epi_on = env_int("DST_GEMM_VEC_EPI", 1) != 0;
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool fused = g.act || g.dact || g.drop_p > 0.0f;
    static const bool vec_plain_on = env_int("DST_GEMM_VEC_PLAIN", 1) != 0;
    const bool ok = !g.rowsum && (g.N & 3) == 0 && (g.ldc & 3) == 0 && al16(g.C) && (!g.bias || al16(g.bias)) &&
                    (!g.ref || (al16(g.ref) && (g.ldref & 3) == 0)) && (!g.C2 || (al16(g.C2) && (g.ldc2 & 3) == 0)) &&
                    (!(g.drop_p > 0.0f) || (g.drop_ld & 3) == 0) && vec_epi_on;
    // bit 0: the vector form of the staged epilogue; bit 1: a plain epilogue (bias / accumulate only) takes the staged vector form too
    // (four 16-byte stores per tile and lane instead of sixteen 4-byte ones)
    g._pad = ok ? (1 | ((!fused && vec_plain_on) ? 2 : 0)) : 0;
  }
  const int Nx = g.N + (g.rowsum ? 1 : 0);            // the fused row sum is one more (virtual, all-ones) column of B
  const bool bf = g.bf16 != 0;
  // the vector kernel needs every operand either k-contiguous or row-contiguous, 16-byte aligned with a leading stride of whole vec4s
  // ... and a leading stride that covers the fast extent rounded up to a vec4 (fetch_tile's clamped 16-byte loads stay inside the row that
  // way; a broadcast operand - leading stride 0 - or overlapping rows take the element-wise kernel)
  auto vec_ok = [](const float* p, int64_t fast, int64_t slow, int64_t extent) {
    return fast == 1 && (reinterpret_cast<uintptr_t>(p) & 15) == 0 && (slow & 3) == 0 && slow >= ((extent + 3) & ~(int64_t)3);
  };
  const bool a_k = vec_ok(g.A, g.a_cs, g.a_rs, g.K), a_r = !a_k && vec_ok(g.A, g.a_rs, g.a_cs, g.M);
  const bool b_k = vec_ok(g.B, g.b_rs, g.b_cs, g.K), b_r = !b_k && vec_ok(g.B, g.b_cs, g.b_rs, g.N);
  static const int force_old = env_int("DST_GEMM_OLD", 0), bn_pref = env_int("DST_GEMM_BN", 0), bm_pref = env_int("DST_GEMM_BM", 0),
                   split_target = env_int("DST_GEMM_SPLIT_WGS", 1024), wg_target = env_int("DST_GEMM_WGS", 768);
  const bool vec = bf && !force_old && (a_k || a_r) && (b_k || b_r) && g.K >= 8 && g.N > 0;
  static const int ws_off = env_int("DST_GEMM_WS", 1) == 0, ws_min_m = env_int("DST_GEMM_WS_MIN_M", 65536);
  // measured inside the training step (same box, DST_GEMM_WS=0 / 1): the resident-weight form wins where the weight is large and the
  // rows are many (81 014 x 256 x 256 input gradient: 66 us against 113) and loses on the short-K products (K = 64 / 128: its 256
  // workgroups re-stage the weight for too little work per row), so it takes K >= 256 on the directed rows only
  if (vec && !ws_off && a_k && g.K >= 256 && g.K <= 512 && (g.K & 127) == 0 && g.M >= ws_min_m && !g.rowsum && g.N >= 16) {
    const int Kp = (g.K + 15) / 16 * 16;
    int nct = g.N > 64 ? 4 : 2;
    if ((size_t)32 * nct * (Kp + 8) * 2 + STAGE_BYTES * 2 > 160 * 1024) nct = 2;      // K = 512: 64 columns of the weight at a time
    const int NC = 32 * nct;
    const int nchunks = (g.N + NC - 1) / NC;
    const size_t lds = (size_t)NC * (Kp + 8) * 2 + STAGE_BYTES * 2;          // eight waves' staging tiles
    int ngroups = (g.M + 255) / 256;
    const int max_groups = (256 + nchunks - 1) / nchunks;                     // one workgroup per CU (the resident weight takes most of its LDS)
    if (ngroups > max_groups) ngroups = max_groups;
    const dim3 grid(8 * ((ngroups + 7) / 8) * nchunks), blk(512);
    static bool attr_done[2] = {false, false};
    if (nct == 4) {
      if (!attr_done[0]) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tr_gemm_ws<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done[0] = true; }
      hipLaunchKernelGGL((k_tr_gemm_ws<4>), grid, blk, lds, s, g, (int)b_r, Kp, nchunks, ngroups);
    } else {
      if (!attr_done[1]) { hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tr_gemm_ws<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_done[1] = true; }
      hipLaunchKernelGGL((k_tr_gemm_ws<2>), grid, blk, lds, s, g, (int)b_r, Kp, nchunks, ngroups);
    }
    return DST_CHECK_LAUNCH();
  }
  static const int skinny_off = env_int("DST_GEMM_SKINNY", 1) == 0;
  if (bf && !skinny_off && g.K <= 8 && g.M >= 1024 && !g.rowsum && g.N > 0) {       // (bf16 mode only: the fp32 mode keeps one summation order everywhere)
    if ((g._pad & 1) && g.b_cs == 1 && (g.b_rs & 3) == 0 && (reinterpret_cast<uintptr_t>(g.B) & 15) == 0)
      hipLaunchKernelGGL(k_tr_gemm_skinny4, dim3((unsigned)(((int64_t)g.M * (g.N >> 2) + 255) / 256)), dim3(256), 0, s, g);
    else
      hipLaunchKernelGGL(k_tr_gemm_skinny, dim3((unsigned)(((int64_t)g.M * g.N + 255) / 256)), dim3(256), 0, s, g);
    return DST_CHECK_LAUNCH();
  }
  static const int wide_off = env_int("DST_GEMM_WIDE", 1) == 0;
  // (K = 4 633 node-row gradients: 52 us in this form against 40 us as 1 024 workgroups of 128 x 64 - too few k-steps per CU)
  const bool wide = vec && !wide_off && g.K >= 16384 && g.M >= 160 && g.M <= 1024 && Nx <= 1024 && g.partial;   // weight gradients: K = rows
  int BM, BN;
  if (wide) {
    BM = 256;
    BN = Nx > 128 ? 256 : Nx > 64 ? 128 : 64;
  } else if (vec) {
    // tiles sized so that the launch has at least ~3 workgroups per CU where the problem allows it: the k-loop of a workgroup exposes
    // one memory round trip per step, and what hides it is the neighbours on the CU (24 - 32 kB in flight per workgroup and step)
    // 128 x 64 / 64 x 64 everywhere except the split products: seeded same-box A/B over the training step - 128-wide tiles (199 registers, two
    // waves per SIMD) cost 0.8 ms of 39.6 wherever they were chosen for the long forward products
    BN = bn_pref ? bn_pref : ((Nx > 64 && g.K >= 4096) ? 128 : 64);
    BM = 128;
    auto count = [&](int bm, int bn) { return (int64_t)((g.M + bm - 1) / bm) * ((Nx + bn - 1) / bn); };
    if (!bn_pref && BN == 128 && count(BM, BN) < wg_target && g.K < 4096) BN = 64;   // (a split product keeps the wide tile: fewer re-reads of its k-slices)
    if (count(BM, BN) < wg_target && g.K < 1024) BM = 64;
    if (bm_pref) BM = bm_pref;
  } else {
    // 128 x 64 tiles (64 x 64 for short M): 102 VGPRs = four waves per SIMD (round 3: 128 x 128 was 5-12 % slower over a training step)
    BM = g.M >= 96 ? 128 : 64;
    BN = 64;
  }
  const int tm = (g.M + BM - 1) / BM, tn = (Nx + BN - 1) / BN;
  const int64_t tiles = (int64_t)tm * tn;
  const int kq = vec ? BK : 16;
  int splits = 1;
  if (g.K >= 1024 && tiles < 512 && g.partial) {
    splits = (int)((wide ? 256 : split_target) / tiles);       // the 512-thread form: one workgroup per CU
    const int max_by_k = (g.K + 255) / 256;
    if (splits > max_by_k) splits = max_by_k;
    const int64_t cap = g.partial_cap / ((int64_t)g.M * Nx);
    if (splits > cap) splits = (int)cap;
    if (splits < 1) splits = 1;
  }
  int kchunk = ((g.K + splits - 1) / splits + kq - 1) / kq * kq;
  if (kchunk < kq) kchunk = kq;
  splits = g.K > 0 ? (g.K + kchunk - 1) / kchunk : 1;
  const int inner = splits > 1 ? (int)tiles : tn, outer = splits > 1 ? splits : tm;
  const dim3 grid(8 * ((outer + 7) / 8) * inner), blk(wide ? 512 : 256);
  if (wide) {
    if (BN == 256) hipLaunchKernelGGL((k_tr_gemm_bf16<256, 256, 512>), grid, blk, 0, s, g, splits, kchunk, (int)a_r, (int)b_r, tm, tn);
    else if (BN == 128) hipLaunchKernelGGL((k_tr_gemm_bf16<256, 128, 512>), grid, blk, 0, s, g, splits, kchunk, (int)a_r, (int)b_r, tm, tn);
    else hipLaunchKernelGGL((k_tr_gemm_bf16<256, 64, 512>), grid, blk, 0, s, g, splits, kchunk, (int)a_r, (int)b_r, tm, tn);
  } else if (vec) {
#define DST_LAUNCH_BF16(M_, N_) hipLaunchKernelGGL((k_tr_gemm_bf16<M_, N_>), grid, blk, 0, s, g, splits, kchunk, (int)a_r, (int)b_r, tm, tn)
    if (BM == 128 && BN == 128) DST_LAUNCH_BF16(128, 128);
    else if (BM == 128) DST_LAUNCH_BF16(128, 64);
    else if (BN == 128) DST_LAUNCH_BF16(64, 128);
    else DST_LAUNCH_BF16(64, 64);
#undef DST_LAUNCH_BF16
  } else if (BM == 128) {
    if (bf) hipLaunchKernelGGL((k_tr_gemm_big<128, 64, true>), grid, blk, 0, s, g, splits, kchunk, tm, tn);
    else hipLaunchKernelGGL((k_tr_gemm_big<128, 64, false>), grid, blk, 0, s, g, splits, kchunk, tm, tn);
  } else {
    if (bf) hipLaunchKernelGGL((k_tr_gemm_big<64, 64, true>), grid, blk, 0, s, g, splits, kchunk, tm, tn);
    else hipLaunchKernelGGL((k_tr_gemm_big<64, 64, false>), grid, blk, 0, s, g, splits, kchunk, tm, tn);
  }
  if (splits > 1) hipLaunchKernelGGL(k_tr_gemm_reduce, dim3((unsigned)(((int64_t)g.M * Nx + 255) / 256)), dim3(256), 0, s, g, splits);
  return DST_CHECK_LAUNCH();
}

}  // extern "C"
