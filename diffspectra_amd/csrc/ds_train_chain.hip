// diffspectra_amd - fused row chains of the bf16 training forward (BASELINE config 5).
//
// dst_pair_chain_fwd: the pair rows of a block BEHIND the attention (dmt.py:156-157,165-169,388 and the edge part of
// equi_update.input_lin, dmt.py:39) as ONE kernel instead of ten launches:
//   he   = node2edge(h_a) + node2edge(h_b) + bias          (the per-node product u comes from the node stream)
//   xe1  = e + edge_gate_msa * he
//   ye1  = LayerNorm(xe1) * (1 + edge_scale_mlp) + edge_shift_mlp        (eps 1e-6, no affine)
//   f3   = ff_linear3(ye1)                s3 = dropout(SiLU(f3))
//   f4   = dropout(ff_linear4(s3))        e_out = ye1 + edge_gate_mlp * f4
//   ed   = input_lin[:, 512:640] [e_out | CondGaussian features] + bias       ro = edge_i(e_out)
// Every intermediate the hand-written backward reads (train_engine.DmtTrainGraph.backward) is written exactly as the unfused kernels
// write it - he, xe1, (mean, rstd), ye1, f3, s3, f4, e_out, X2 = [e_out | features], ed, ro - or skipped when its pointer is NULL (the
// no-gradient self-conditioning forward keeps only e_out, ed and ro).  Products: operands rounded to bf16 (round to nearest even, as
// k_tr_gemm_bf16 rounds them while staging), v_mfma_f32_32x32x16_bf16, fp32 accumulation onto the bias; everything else fp32.  The
// dropout masks are dst_dropout's (Philox block (row * N + col) / 4 of stream 4 * block + site): bit-identical to the unfused path and
// to golden G17's injected masks.
//
// FLAT tiles, FOUR WAVES PER TILE: a 256-thread workgroup owns 32 consecutive rows of the packed pair (or directed) rows, whatever molecules
// they belong to - the per-molecule adaLN rows are looked up per row (pair_mol), the atoms of a pair come from the layout's pair_a /
// pair_b tables.  The row passes of the LayerNorm stages and the 32-column chunks of every product are dealt over the four waves, the
// tiles they share live in the workgroup's LDS (35 - 50 kB: three workgroups = twelve waves per CU), one barrier between phases.
// History (profiles/r05_train_fused_ab.txt): one workgroup per molecule lost 2 - 3x to the imbalance between a 29-atom and a 9-atom
// molecule; one WAVE per tile (wave-private LDS, no barrier) left a wave alone on its SIMD with 35 kB of LDS and ~8 k dependent
// instructions per tile - 40 us per tile whatever was done to its memory accesses.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/diffspectra_hip.h"
#include "../../include/diffspectra_train.h"
#include "ds_train_common.h"

typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

namespace {

#define DST_CHECK_LAUNCH() (hipGetLastError() == hipSuccess ? DS_OK : DS_ERR_LAUNCH)

constexpr int CH_NW = 4, CH_NT = CH_NW * 64;      // waves per workgroup
constexpr int LD_Y = 72, LD_S = 136, LD_F = 68, LD_ST = 36;   // LDS row strides: bf16 tiles in halves (16-byte rows), fp32 tiles in floats

struct ChainLds {
  float yf[32][LD_F];            // ye1, fp32 (the residual of the FF)
  float stage[CH_NW][32][LD_ST]; // per wave: one 32 x 32 accumulator tile on its way from the MFMA layout to rows
  __bf16 yb[32][LD_Y];           // ye1, bf16: A operand of ff_linear3
  __bf16 sb[32][LD_S];           // s3, bf16: A operand of ff_linear4
  __bf16 eb[32][LD_S];           // [e_out | features], bf16: A operand of input_lin's edge part and of the read-out slice
};

__device__ __forceinline__ f4_t ld4(const float* p) { return *reinterpret_cast<const f4_t*>(p); }
__device__ __forceinline__ void st4(float* p, f4_t v) { *reinterpret_cast<f4_t*>(p) = v; }
__device__ __forceinline__ float sum16(float v) {          // sum over aligned groups of 16 lanes
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// The same sum with DPP row rotations (a DPP row IS 16 lanes): no LDS-crossbar traffic.  The fused BACKWARD kernels use this form: with
// ds_bpermute shuffles in flight next to their LDS reads, two executions inside a step differed in single rows (profiles/HISTORY.md, round 5).
__device__ __forceinline__ float sum16_dpp(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));   // row_ror:8
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));   // row_ror:4
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));   // row_ror:2
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));   // row_ror:1
  return v;
}
// sum over the 64 lanes without the LDS crossbar: DPP row sums, then the four row totals by readlane
__device__ __forceinline__ float sum64_dpp(float v) {
  v = sum16_dpp(v);
  const int b = __builtin_bit_cast(int, v);
  const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
  const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
  return ((r0 + r1) + r2) + r3;
}
__device__ __forceinline__ bf16x4_t to_bf4(f4_t v) {
  bf16x4_t r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = (__bf16)v[j];
  return r;
}
// SiLU and tanh in their hardware exp2 / rcp forms (absolute error ~1e-7: five orders below the bf16 step of the products around them; the
// libm forms of the unfused epilogues cost 35 - 50 instructions per value, and a wave is alone on its SIMD here)
__device__ __forceinline__ float fast_silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f)); }
__device__ __forceinline__ float fast_tanh(float x) { return fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 2.8853900817779268f) + 1.0f), 1.0f); }
// Wave-private LDS: order this wave's LDS writes before its LDS reads.  LDS operations of a wave complete in order, so waiting for the LDS
// counter is enough; a workgroup-scope fence would also wait for every global store in flight (the tape stores: a memory round trip per call -
// the first version of these kernels spent most of its time there).
__device__ __forceinline__ void wave_lds_sync() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_wave_barrier();
}

// acc += A[32 x 16 KB] W^T for output columns col0 .. col0 + 31: A = bf16 rows in LDS (row stride lda halves, first column a0), W = torch
// Linear weight [out, in] as bf16 bits with row stride ldw (first input column w0; dst_pack_bf16_pieces rounded it from the fp32 master
// weights, nearest even - what the GEMM kernels do while staging); output columns >= n_out are zero.
// A tile's weights come from L2 - 256 kB of fp32 for the 256 x 256 coord_mlp.0 per 32-row tile was the bound of these kernels: with the
// loads halved (an experiment that read half of every fragment) the directed chain went from 112 to 64 us.  As bf16 a fragment is ONE
// 16-byte load per k-block and goes into the MFMA as it is.
// The weight fragments are FETCHED (wfetch) and APPLIED (mma_apply) separately: the chunk loops request the next chunk's fragments before
// the current chunk's epilogue - otherwise every chunk pays one L2 round trip in front of its MFMAs.
template <int KB>
struct WFrag {
  bf16x8_t w[KB];
  bool ok;
};
template <int KB>
__device__ __forceinline__ void wfetch(WFrag<KB>& f, const uint16_t* __restrict__ W, int64_t ldw, int w0, int col0, int n_out) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int col = col0 + r;
  const uint16_t* wrow = W + (int64_t)min(col, n_out - 1) * ldw + w0 + 8 * hh;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) f.w[kb] = *reinterpret_cast<const bf16x8_t*>(wrow + 16 * kb);
  f.ok = col < n_out;
}
// MASKED: the product has fewer than 32 output columns in this chunk (the 16-column read-out slice, the 3-column coord_mlp.2): columns
// beyond n_out multiply by zero.
template <int KB, bool MASKED = false>
__device__ __forceinline__ void mma_apply(const __bf16* A, int lda, int a0, const WFrag<KB>& f, f32x16_t& acc) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const __bf16* arow = A + r * lda + a0 + 8 * hh;
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    bf16x8_t b = f.w[kb];
    if (MASKED && !f.ok) {
#pragma unroll
      for (int j = 0; j < 8; ++j) b[j] = (__bf16)0.0f;
    }
    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(arow + 16 * kb);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}
template <int KB>
__device__ __forceinline__ void mma_rows(const __bf16* A, int lda, int a0, const uint16_t* __restrict__ W, int64_t ldw, int w0, int col0, int n_out,
                                         f32x16_t& acc) {
  WFrag<KB> f;
  wfetch<KB>(f, W, ldw, w0, col0, n_out);
  mma_apply<KB>(A, lda, a0, f, acc);
}
__device__ __forceinline__ void acc_to_stage(const f32x16_t& acc, float (*stage)[LD_ST]) {   // accumulator: lane = column, register i = row
  const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) stage[(i & 3) + 8 * (i >> 2) + 4 * hh][c] = acc[i];
}

// Vector-memory operations of a wave retire in issue order: a load issued behind stores waits for them.  So inside a tile every load is
// requested before the stores of its phase: the index tables and the rows of stage 1 first, the per-row gate rows of the FF epilogue with
// them, each GEMM chunk's weight fragments and bias before the previous epilogue.
__global__ __launch_bounds__(CH_NT, 3) void k_pair_chain_fwd(dst_layout L, dst_pair_chain_args a, const int32_t* __restrict__ pair_a,
                                                          const int32_t* __restrict__ pair_b, const int32_t* __restrict__ pair_mol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  ChainLds& w = *reinterpret_cast<ChainLds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float (*stage)[LD_ST] = w.stage[wave];
  const int Pp = L.Pp;
  const int t0 = blockIdx.x * 32;
  const int valid = min(32, Pp - t0);
  const int64_t g0 = t0;                                     // global pair row of the tile's row 0
  const unsigned int thr = dst::dropout_threshold(a.drop_p);
  const float keep_scale = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const int sub = lane >> 4, cl = (lane & 15) * 4;          // row layout of stage 1: a row = 16 lanes x float4
  const int er = lane >> 3, ec = (lane & 7) * 4;            // row layout of the GEMM epilogues: a 32-column chunk row = 8 lanes x float4
  const f4_t bias = ld4(a.n2e_bias + cl);
  const float b3c = a.b3[wave * 32 + (lane & 31)], b4c = a.b4[(wave & 1) * 32 + (lane & 31)];
  // edge_gate_mlp of the rows this lane finishes in the ff_linear4 epilogue (waves 0, 1: row it * 8 + er, columns wave * 32 + ec ..)
  f4_t g2c[4];
#pragma unroll
  for (int it = 0; it < 4; ++it)
    g2c[it] = ld4(a.ada + (int64_t)pair_mol[min(t0 + it * 8 + er, Pp - 1)] * a.ada_ld + a.gate2_off + (wave & 1) * 32 + ec);
  WFrag<4> f3w;
  wfetch<4>(f3w, a.W3, 64, 0, wave * 32, 128);
  // ---- stage 1: gather, gated residual, LayerNorm + modulate; wave w takes passes 2 w, 2 w + 1 (rows 8 w .. 8 w + 7)
  {
    int ia[2], ib[2], im[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int gp = min(t0 + (2 * wave + q) * 4 + sub, Pp - 1);
      ia[q] = pair_a[gp]; ib[q] = pair_b[gp]; im[q] = pair_mol[gp];
    }
    f4_t ua[2], ub[2], ev[2], fv[2], g1v[2], shv[2], scv[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t gp = min(t0 + (2 * wave + q) * 4 + sub, Pp - 1);
      const float* adm = a.ada + (int64_t)im[q] * a.ada_ld;
      ua[q] = ld4(a.u + (int64_t)ia[q] * 64 + cl);
      ub[q] = ld4(a.u + (int64_t)ib[q] * 64 + cl);
      ev[q] = ld4(a.e_in + gp * 64 + cl);
      fv[q] = ld4(a.feat + gp * a.ld_feat + cl);
      g1v[q] = ld4(adm + a.gate1_off + cl); shv[q] = ld4(adm + a.shift_off + cl); scv[q] = ld4(adm + a.scale_off + cl);
    }
    __builtin_amdgcn_sched_barrier(0);                       // (the output pointers may alias the inputs for all the compiler knows: keep every load above the first store)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = (2 * wave + q) * 4 + sub;
      const int64_t gp = min(t0 + row, Pp - 1);
      const f4_t he = (ua[q] + ub[q]) + bias;
      const f4_t x = ev[q] + g1v[q] * he;
      f4_t ft = fv[q];
      const float mean = sum16((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 64.0f);
      const f4_t d = x - mean;
      const float rstd = 1.0f / sqrtf(sum16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 64.0f) + 1e-6f);
      f4_t y = (d * rstd) * (1.0f + scv[q]) + shv[q];
      if (row < valid) {
        if (a.he) st4(a.he + gp * 64 + cl, he);
        if (a.xe1) st4(a.xe1 + gp * 64 + cl, x);
        if (a.st && (lane & 15) == 0) { a.st[gp * 2] = mean; a.st[gp * 2 + 1] = rstd; }
        if (a.ye1) st4(a.ye1 + gp * 64 + cl, y);
        if (a.X2) st4(a.X2 + gp * 128 + 64 + cl, ft);
      } else {
        y = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
        ft = y;
      }
      st4(&w.yf[row][cl], y);
      *reinterpret_cast<bf16x4_t*>(&w.yb[row][cl]) = to_bf4(y);
      *reinterpret_cast<bf16x4_t*>(&w.eb[row][64 + cl]) = to_bf4(ft);
    }
  }
  __syncthreads();
  // ---- ff_linear3 (64 -> 128), SiLU, dropout: chunk `wave`
  WFrag<8> edw;                                              // ff_linear4's fragments (waves 0, 1), then input_lin's
  {
    const int ch = wave;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = b3c;
    mma_apply<4>(&w.yb[0][0], LD_Y, 0, f3w, acc);
    // the next product's fragments fly under this epilogue: ff_linear4's on waves 0, 1, input_lin's first chunk on the waves that sit
    // ff_linear4 out (one fragment set in flight per wave: three workgroups per CU leave a wave 168 registers)
    if (wave < 2) wfetch<8>(edw, a.W4, 128, 0, wave * 32, 64);
    else wfetch<8>(edw, a.Wed, a.ld_wed, 0, wave * 32, 256);
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er, col = ch * 32 + ec;
      const int64_t gr = g0 + row;
      const f4_t v = ld4(&stage[row][ec]);
      f4_t sv;
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[e] = fast_silu(v[e]);
      if (a.drop_p > 0.0f) {
        unsigned int c[4];
        dst::dropout_block(a.seed, a.stream3, (gr * 128 + col) >> 2, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = c[e] >= thr ? sv[e] * keep_scale : 0.0f;
      }
      if (row < valid) {
        if (a.f3) st4(a.f3 + gr * 128 + col, v);
        if (a.s3) st4(a.s3 + gr * 128 + col, sv);
      } else {
        sv = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      }
      *reinterpret_cast<bf16x4_t*>(&w.sb[row][col]) = to_bf4(sv);
    }
  }
  __syncthreads();
  // ---- ff_linear4 (128 -> 64), dropout, gated residual: chunks 0, 1 on waves 0, 1
  if (wave < 2) {
    const int ch = wave;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = b4c;
    mma_apply<8>(&w.sb[0][0], LD_S, 0, edw, acc);
    wfetch<8>(edw, a.Wed, a.ld_wed, 0, wave * 32, 256);
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er, col = ch * 32 + ec;
      const int64_t gr = g0 + row;
      f4_t v = ld4(&stage[row][ec]);
      if (a.drop_p > 0.0f) {
        unsigned int c[4];
        dst::dropout_block(a.seed, a.stream4, (gr * 64 + col) >> 2, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c[e] >= thr ? v[e] * keep_scale : 0.0f;
      }
      f4_t eo = ld4(&w.yf[row][col]) + g2c[it] * v;
      if (row < valid) {
        if (a.f4) st4(a.f4 + gr * 64 + col, v);
        st4(a.e_out + gr * 64 + col, eo);
        if (a.X2) st4(a.X2 + gr * 128 + col, eo);
      } else {
        eo = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      }
      *reinterpret_cast<bf16x4_t*>(&w.eb[row][col]) = to_bf4(eo);
    }
  }
  __syncthreads();
  // ---- input_lin's edge part ([e_out | features] 128 -> 256): chunks wave, wave + 4; the read-out slice (e_out 64 -> 16) on wave 3
  float bcur = a.bed[wave * 32 + (lane & 31)];
  WFrag<4> row_w;
  if (wave == 3) wfetch<4>(row_w, a.Wro, 64, 0, 0, 16);
  const float broc = (lane & 31) < 16 ? a.bro[lane & 31] : 0.0f;
#pragma unroll 1
  for (int k = 0; k < 3; ++k) {
    const bool ro = k == 2;
    if (ro && wave != 3) break;
    const int ch = wave + 4 * k;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = ro ? broc : bcur;
    if (ro) mma_apply<4, true>(&w.eb[0][0], LD_S, 0, row_w, acc);
    else mma_apply<8>(&w.eb[0][0], LD_S, 0, edw, acc);
    if (k == 0) { wfetch<8>(edw, a.Wed, a.ld_wed, 0, (wave + 4) * 32, 256); bcur = a.bed[(wave + 4) * 32 + (lane & 31)]; }   // before this chunk's stores
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      const int64_t gr = g0 + row;
      const f4_t v = ld4(&stage[row][ec]);
      if (row < valid) {
        if (!ro) st4(a.ed + gr * 256 + ch * 32 + ec, v);
        else if (ec < 16) st4(a.ro + gr * 16 + ec, v);
      }
    }
    wave_lds_sync();
  }
}

// dst_pair_front_fwd: the pair rows of a block IN FRONT of the attention (dmt.py:136-139,145-149; layers.py:291-295,328-334,165-166,183):
//   d2 = |pos_a - pos_b|^2;  x' = d2 (1 + ada[dist]) + ada[dist + 1];  feat = [x', gaussian_k(x')];  X1 = [feat | e]
//   e1 = edge_emb(X1);  en = LN(e1) (1 + ada[scale]) + ada[shift];  te = tanh(en [lin_edge0 | lin_edge1]^T)
// replacing dst_geom_fwd, a copy, two dst_gemm calls and dst_lnmod_fwd.  Same arithmetic per element as those kernels for the features
// (expf, the truncated-pi constant, divisions where they divide), bf16-rounded MFMA operands with fp32 accumulation.
#define DST_GAUSS_A 2.50662732f /* fp32((2 * 3.14159) ** 0.5), as in ds_train.hip */
struct FrontLds {
  float ef[32][LD_F];            // e1 (both 32-column chunks), fp32: the LayerNorm reads whole rows
  float stage[CH_NW][32][LD_ST];
  __bf16 xb[32][LD_S];           // X1 = [feat | e], bf16
  __bf16 nb[32][LD_Y];           // en, bf16
};

__global__ __launch_bounds__(CH_NT, 3) void k_pair_front_fwd(dst_layout L, dst_pair_front_args a, const int32_t* __restrict__ pair_a,
                                                          const int32_t* __restrict__ pair_b, const int32_t* __restrict__ pair_mol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  FrontLds& w = *reinterpret_cast<FrontLds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float (*stage)[LD_ST] = w.stage[wave];
  const int Pp = L.Pp;
  const int t0 = blockIdx.x * 32;
  const int valid = min(32, Pp - t0);
  const int64_t g0 = t0;
  const int sub = lane >> 4, cl = (lane & 15) * 4, er = lane >> 3, ec = (lane & 7) * 4;
  // the lane's four Gaussians (features cl .. cl + 3; feature 0 is x' itself)
  float mu[4], sd[4], nrm[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int k = cl + j;
    mu[j] = k ? a.means[k - 1] : 0.0f;
    sd[j] = k ? fabsf(a.stds[k - 1]) + 1e-5f : 1.0f;
    nrm[j] = DST_GAUSS_A * sd[j];
  }
  const float beec = a.bee[(wave & 1) * 32 + (lane & 31)];
  WFrag<8> eew;
  if (wave < 2) wfetch<8>(eew, a.Wee, 128, 0, wave * 32, 64);
  WFrag<4> tew;
  wfetch<4>(tew, a.Wte, 64, 0, wave * 32, 512);
  // ---- features + X1: wave w takes passes 2 w, 2 w + 1 (every load before the first store)
  int ia[2], ib[2], im[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int gp = min(t0 + (2 * wave + q) * 4 + sub, Pp - 1);
    ia[q] = pair_a[gp]; ib[q] = pair_b[gp]; im[q] = pair_mol[gp];
  }
  f4_t evs[2], shs[2], scs[2];
  float d2v[2], dscv[2], dshv[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t gp = min(t0 + (2 * wave + q) * 4 + sub, Pp - 1);
    const float* pa_ = a.pos + (int64_t)ia[q] * 3;
    const float* pb_ = a.pos + (int64_t)ib[q] * 3;
    const float dx = pa_[0] - pb_[0], dy = pa_[1] - pb_[1], dz = pa_[2] - pb_[2];
    d2v[q] = dx * dx + dy * dy + dz * dz;
    const float* adm = a.ada + (int64_t)im[q] * a.ada_ld;
    dscv[q] = adm[a.dist_off]; dshv[q] = adm[a.dist_off + 1];
    shs[q] = ld4(adm + a.shift_off + cl); scs[q] = ld4(adm + a.scale_off + cl);
    evs[q] = ld4(a.e_in + gp * 64 + cl);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (2 * wave + q) * 4 + sub;
    const int64_t gp = min(t0 + row, Pp - 1);
    const float d2 = d2v[q];
    const float x = d2 * (dscv[q] + 1.0f) + dshv[q];
    f4_t ft;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float u = (x - mu[j]) / sd[j];
      ft[j] = expf(-0.5f * (u * u)) / nrm[j];
    }
    if (cl == 0) ft[0] = x;
    f4_t ev = evs[q];
    if (row < valid) {
      st4(a.X1 + gp * 128 + cl, ft);
      st4(a.X1 + gp * 128 + 64 + cl, ev);
      if (cl == 0) {
        if (a.xs) a.xs[gp] = x;
        if (a.d2) a.d2[gp] = d2;
      }
    } else {
      ft = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      ev = ft;
    }
    *reinterpret_cast<bf16x4_t*>(&w.xb[row][cl]) = to_bf4(ft);
    *reinterpret_cast<bf16x4_t*>(&w.xb[row][64 + cl]) = to_bf4(ev);
  }
  __syncthreads();
  // ---- edge_emb (128 -> 64): chunks 0, 1 on waves 0, 1
  if (wave < 2) {
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = beec;
    mma_apply<8>(&w.xb[0][0], LD_S, 0, eew, acc);
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) w.ef[(i & 3) + 8 * (i >> 2) + 4 * hh][wave * 32 + c] = acc[i];
  }
  __syncthreads();
  // ---- LayerNorm + modulate: passes 2 w, 2 w + 1
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (2 * wave + q) * 4 + sub;
    const int64_t gp = g0 + row;
    const f4_t x = ld4(&w.ef[row][cl]);
    const float mean = sum16((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 64.0f);
    const f4_t d = x - mean;
    const float rstd = 1.0f / sqrtf(sum16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 64.0f) + 1e-6f);
    f4_t y = (d * rstd) * (1.0f + scs[q]) + shs[q];
    if (row < valid) {
      if (a.e1) st4(a.e1 + gp * 64 + cl, x);
      if (a.st && (lane & 15) == 0) { a.st[gp * 2] = mean; a.st[gp * 2 + 1] = rstd; }
      if (a.en) st4(a.en + gp * 64 + cl, y);
    } else {
      y = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
    }
    *reinterpret_cast<bf16x4_t*>(&w.nb[row][cl]) = to_bf4(y);
  }
  __syncthreads();
  // ---- tanh(en [lin_edge0 | lin_edge1]^T) (64 -> 512): chunks wave, wave + 4, wave + 8, wave + 12
#pragma unroll 1
  for (int k = 0; k < 4; ++k) {
    const int ch = wave + 4 * k;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    mma_apply<4>(&w.nb[0][0], LD_Y, 0, tew, acc);
    if (k < 3) wfetch<4>(tew, a.Wte, 64, 0, (ch + 4) * 32, 512);
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      f4_t v = ld4(&stage[row][ec]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = fast_tanh(v[e]);
      if (row < valid) st4(a.te + (g0 + row) * 512 + ch * 32 + ec, v);
    }
    wave_lds_sync();
  }
}

// dst_dir_chain_fwd: the DIRECTED rows of a block (dmt.py:37-48: both directions of every pair through equi_update's input LayerNorm and
// coord_mlp) as one kernel instead of dst_zbuild_fwd, dst_lnmod_fwd and two dst_gemm calls:
//   zz[2p + dir] = ac[row][0:256] + ac[col][256:512] + ed[p]   (dir 0: row = a, col = b; dir 1: swapped)
//   zn = LN(zz) (1 + ada[scale]) + ada[shift];  c0 = zn W0^T + b0;  sc0 = SiLU(c0);  c2 = sc0 W2^T   (W2 [3,256], no bias)
// Tape: zz, (mean, rstd), zn, c0, sc0 (each may be NULL); c2 [2 Pp, 3] always.  sc0 is not staged whole: each 32-column chunk's bf16
// tile goes straight into the 256 -> 3 product of its wave (two k-blocks per chunk); the four waves' partial products are added at the end
// in wave order.
constexpr int LD_Z = 264;         // bf16 row of 256 + 8 (16-byte rows, 528 bytes: the 16-byte fragments of 16 rows hit distinct banks)
constexpr int LD_C = 40;          // bf16 row of one 32-column chunk + 8
struct DirLds {
  float stage[CH_NW][32][LD_ST];
  __bf16 zb[32][LD_Z];           // zn, bf16
  __bf16 cb[CH_NW][32][LD_C];    // per wave: the current chunk of sc0, bf16
};

__global__ __launch_bounds__(CH_NT, 3) void k_dir_chain_fwd(dst_layout L, dst_dir_chain_args a, const int32_t* __restrict__ pair_a,
                                                         const int32_t* __restrict__ pair_b, const int32_t* __restrict__ pair_mol) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  DirLds& w = *reinterpret_cast<DirLds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float (*stage)[LD_ST] = w.stage[wave];
  __bf16 (*cb)[LD_C] = w.cb[wave];
  const int nd = 2 * L.Pp;
  const int t0 = blockIdx.x * 32;
  const int valid = min(32, nd - t0);
  const int64_t g0 = t0;                                     // global directed row of the tile's row 0
  const int sub = lane >> 4, j16 = lane & 15, er = lane >> 3, ec = (lane & 7) * 4;
  float bcur = a.b0[wave * 32 + (lane & 31)];
  // ---- z, LayerNorm + modulate: a row = 16 lanes, lane j holds the float4s at columns 4 j + 64 u; wave w takes passes 2 w, 2 w + 1,
  //      one at a time (a pass holds 20 float4 per lane in flight; the second pass's loads queue behind the first one's stores - the
  //      CU's other waves cover that)
  int ra[2], cb_[2], im[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int dl = min(t0 + (2 * wave + q) * 4 + sub, nd - 1), pl = dl >> 1, dir = dl & 1;
    const int xa = pair_a[pl], xb = pair_b[pl];
    ra[q] = dir ? xb : xa; cb_[q] = dir ? xa : xb; im[q] = pair_mol[pl];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = (2 * wave + q) * 4 + sub;
    const int dl = min(t0 + row, nd - 1), pl = dl >> 1;
    const int64_t gd = dl;
    const float* pr = a.ac + (int64_t)ra[q] * 512 + 4 * j16;
    const float* pc = a.ac + (int64_t)cb_[q] * 512 + 256 + 4 * j16;
    const float* pe = a.ed + (int64_t)pl * 256 + 4 * j16;
    const float* adm = a.ada + (int64_t)im[q] * a.ada_ld + 4 * j16;
    f4_t xr[4], xc[4], xe[4], shq[4], scq[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      xr[u] = ld4(pr + 64 * u); xc[u] = ld4(pc + 64 * u); xe[u] = ld4(pe + 64 * u);
      shq[u] = ld4(adm + a.shift_off + 64 * u); scq[u] = ld4(adm + a.scale_off + 64 * u);
    }
    __builtin_amdgcn_sched_barrier(0);                       // the pass's loads above its first store
    f4_t x[4];
    float s1 = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      x[u] = (xr[u] + xc[u]) + xe[u];
      s1 += (x[u][0] + x[u][1]) + (x[u][2] + x[u][3]);
    }
    const float mean = sum16(s1) * (1.0f / 256.0f);
    float s2 = 0.0f;
    f4_t dv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      dv[u] = x[u] - mean;
      s2 += (dv[u][0] * dv[u][0] + dv[u][1] * dv[u][1]) + (dv[u][2] * dv[u][2] + dv[u][3] * dv[u][3]);
    }
    const float rstd = 1.0f / sqrtf(sum16(s2) * (1.0f / 256.0f) + 1e-6f);
    const bool live = row < valid;
    if (live && a.st && j16 == 0) { a.st[gd * 2] = mean; a.st[gd * 2 + 1] = rstd; }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      f4_t y = (dv[u] * rstd) * (1.0f + scq[u]) + shq[u];
      if (live) {
        if (a.zz) st4(a.zz + gd * 256 + 64 * u + 4 * j16, x[u]);
        if (a.zn) st4(a.zn + gd * 256 + 64 * u + 4 * j16, y);
      } else {
        y = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      }
      *reinterpret_cast<bf16x4_t*>(&w.zb[row][64 * u + 4 * j16]) = to_bf4(y);
    }
  }
  WFrag<8> c0w, c0v;                                          // (requested behind stage 1: its rows need the registers; the other waves of the CU cover the trip)
  wfetch<8>(c0w, a.W0, 256, 0, wave * 32, 256);
  WFrag<2> w2;
  wfetch<2>(w2, a.W2, 256, wave * 32, 0, 3);
  __syncthreads();
  // ---- coord_mlp.0 (256 -> 256) + SiLU: chunks wave, wave + 4; coord_mlp.2 (256 -> 3) chunk by chunk into this wave's partial product
  f32x16_t acc2;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc2[i] = 0.0f;
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    const int ch = wave + 4 * k;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bcur;
    wfetch<8>(c0v, a.W0, 256, 128, ch * 32, 256);            // the second k-half flies under the first half's MFMAs
    mma_apply<8>(&w.zb[0][0], LD_Z, 0, c0w, acc);
    mma_apply<8>(&w.zb[0][0], LD_Z, 128, c0v, acc);
    WFrag<2> w2n = w2;
    if (k == 0) {                                            // the second chunk's first half and bias: requested before this chunk's stores
      wfetch<8>(c0w, a.W0, 256, 0, (ch + 4) * 32, 256);
      wfetch<2>(w2n, a.W2, 256, (ch + 4) * 32, 0, 3);
      bcur = a.b0[(ch + 4) * 32 + (lane & 31)];
    }
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er, col = ch * 32 + ec;
      const int64_t gr = g0 + row;
      const f4_t v = ld4(&stage[row][ec]);
      f4_t sv;
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[e] = fast_silu(v[e]);
      if (row < valid) {
        if (a.c0) st4(a.c0 + gr * 256 + col, v);
        if (a.sc0) st4(a.sc0 + gr * 256 + col, sv);
      } else {
        sv = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      }
      *reinterpret_cast<bf16x4_t*>(&cb[row][ec]) = to_bf4(sv);
    }
    wave_lds_sync();
    mma_apply<2, true>(&cb[0][0], LD_C, 0, w2, acc2);
    w2 = w2n;
    wave_lds_sync();
  }
  // the four waves' partial 256 -> 3 products, added in wave order by wave 0
  {
    const int c = lane & 31, hh = lane >> 5;
    if (c < 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) stage[(i & 3) + 8 * (i >> 2) + 4 * hh][c] = acc2[i];
    }
  }
  __syncthreads();
  if (wave == 0 && lane < 32 && lane < valid) {
    float* o = a.c2 + (g0 + lane) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = ((w.stage[0][lane][c] + w.stage[1][lane][c]) + w.stage[2][lane][c]) + w.stage[3][lane][c];
  }
}

// dst_node_chain_fwd: the NODE rows of a block behind the attention (dmt.py:113-116,158-163,387 and the node parts of
// equi_update.input_lin, dmt.py:39) as one kernel instead of seven launches on the node stream:
//   x1 = h + node_gate_msa * attn;  y1 = LN(x1) (1 + node_scale_mlp) + node_shift_mlp
//   f1 = ff_linear1(y1);  s1 = dropout(SiLU(f1));  f2 = dropout(ff_linear2(s1));  h_out = y1 + node_gate_mlp * f2
//   ac = h_out [W_row | W_col]^T (the two node parts of input_lin, no bias);  rn = node_i(h_out)
// 4 600 node rows are 145 tiles: the launches this replaces were each shorter than the gap between two dependent launches, and the
// directed rows of the block (main stream) wait for `ac` at the end of that chain - leaving the FF out of the forward (an experiment)
// shortened the step by 0.55 ms.  One 512-thread workgroup per 32-row tile, eight waves: a wave owns four rows of the LayerNorm
// stage and two (ff_linear1, input_lin) or one (ff_linear2) 32-column chunk of every product; weights as bf16 (dst_pack_bf16_pieces).
constexpr int NC_NW = 8, NC_NT = NC_NW * 64;
constexpr int LD_YF = 260, LD_S2 = 520;   // fp32 row of 256 + 4; bf16 row of 512 + 8
struct NodeLds {
  float yf[32][LD_YF];             // y1, fp32: the residual of the FF
  float stage[NC_NW][32][LD_ST];
  __bf16 yb[32][LD_Z];             // y1, then h_out, bf16: A operand of ff_linear1, then of input_lin / the read-out slice
  __bf16 sb[32][LD_S2];            // s1, bf16: A operand of ff_linear2
};
__device__ __forceinline__ float sum64(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__global__ __launch_bounds__(NC_NT) void k_node_chain_fwd(dst_layout L, dst_node_chain_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  NodeLds& w = *reinterpret_cast<NodeLds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float (*stage)[LD_ST] = w.stage[wave];
  const int Nn = L.Nn;
  const int t0 = blockIdx.x * 32;
  const int valid = min(32, Nn - t0);
  const int64_t g0 = t0;
  const unsigned int thr = dst::dropout_threshold(a.drop_p);
  const float keep_scale = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const int er = lane >> 3, ec = (lane & 7) * 4, cl = lane * 4;
  // ff_linear1's first fragments fly under stage 1
  WFrag<8> fa, fb;
  wfetch<8>(fa, a.W1, 256, 0, wave * 32, 512);
  float bcur = a.b1[wave * 32 + (lane & 31)];
  // ---- stage 1: gated residual, LayerNorm + modulate; wave w takes rows 4 w .. 4 w + 3 (a row = 64 lanes x float4), loads first
  {
    f4_t hv[4], av[4], g1v[4], shv[4], scv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t gr = min(t0 + 4 * wave + q, Nn - 1);
      const float* adm = a.ada + (int64_t)a.node_mol[gr] * a.ada_ld;
      hv[q] = ld4(a.h_in + gr * 256 + cl); av[q] = ld4(a.attn + gr * 256 + cl);
      g1v[q] = ld4(adm + a.gate1_off + cl); shv[q] = ld4(adm + a.shift_off + cl); scv[q] = ld4(adm + a.scale_off + cl);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * wave + q;
      const int64_t gr = min(t0 + row, Nn - 1);
      const f4_t x = hv[q] + g1v[q] * av[q];
      const float mean = sum64((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 256.0f);
      const f4_t d = x - mean;
      const float rstd = 1.0f / sqrtf(sum64((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 256.0f) + 1e-6f);
      f4_t y = (d * rstd) * (1.0f + scv[q]) + shv[q];
      if (row < valid) {
        if (a.x1) st4(a.x1 + gr * 256 + cl, x);
        if (a.st && lane == 0) { a.st[gr * 2] = mean; a.st[gr * 2 + 1] = rstd; }
        if (a.y1) st4(a.y1 + gr * 256 + cl, y);
      } else {
        y = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      }
      st4(&w.yf[row][cl], y);
      *reinterpret_cast<bf16x4_t*>(&w.yb[row][cl]) = to_bf4(y);
    }
  }
  __syncthreads();
  // ---- ff_linear1 (256 -> 512), SiLU, dropout: chunks wave, wave + 8
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    const int ch = wave + 8 * k;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bcur;
    wfetch<8>(fb, a.W1, 256, 128, ch * 32, 512);               // the second k-half flies under the first half's MFMAs
    mma_apply<8>(&w.yb[0][0], LD_Z, 0, fa, acc);
    mma_apply<8>(&w.yb[0][0], LD_Z, 128, fb, acc);
    if (k == 0) { wfetch<8>(fa, a.W1, 256, 0, (ch + 8) * 32, 512); bcur = a.b1[(ch + 8) * 32 + (lane & 31)]; }
    else { wfetch<8>(fa, a.W2, 512, 0, wave * 32, 256); bcur = a.b2[wave * 32 + (lane & 31)]; }      // ff_linear2's first quarter
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er, col = ch * 32 + ec;
      const int64_t gr = g0 + row;
      const f4_t v = ld4(&stage[row][ec]);
      f4_t sv;
#pragma unroll
      for (int e = 0; e < 4; ++e) sv[e] = fast_silu(v[e]);
      if (a.drop_p > 0.0f) {
        unsigned int c[4];
        dst::dropout_block(a.seed, a.stream1, (gr * 512 + col) >> 2, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) sv[e] = c[e] >= thr ? sv[e] * keep_scale : 0.0f;
      }
      if (row < valid) {
        if (a.f1) st4(a.f1 + gr * 512 + col, v);
        if (a.s1) st4(a.s1 + gr * 512 + col, sv);
      } else {
        sv = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      }
      *reinterpret_cast<bf16x4_t*>(&w.sb[row][col]) = to_bf4(sv);
    }
    wave_lds_sync();
  }
  // the gate rows of the ff_linear2 epilogue (row it * 8 + er, columns wave * 32 + ec ..): requested before the barrier
  f4_t g2c[4];
#pragma unroll
  for (int it = 0; it < 4; ++it)
    g2c[it] = ld4(a.ada + (int64_t)a.node_mol[min(t0 + it * 8 + er, Nn - 1)] * a.ada_ld + a.gate2_off + wave * 32 + ec);
  __syncthreads();
  // ---- ff_linear2 (512 -> 256), dropout, gated residual: chunk `wave`, four k-quarters
  {
    const int ch = wave;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bcur;
    wfetch<8>(fb, a.W2, 512, 128, ch * 32, 256);
    mma_apply<8>(&w.sb[0][0], LD_S2, 0, fa, acc);
    wfetch<8>(fa, a.W2, 512, 256, ch * 32, 256);
    mma_apply<8>(&w.sb[0][0], LD_S2, 128, fb, acc);
    wfetch<8>(fb, a.W2, 512, 384, ch * 32, 256);
    mma_apply<8>(&w.sb[0][0], LD_S2, 256, fa, acc);
    wfetch<8>(fa, a.Wac, 256, 0, wave * 32, 512);              // input_lin's first fragments
    mma_apply<8>(&w.sb[0][0], LD_S2, 384, fb, acc);
    acc_to_stage(acc, stage);
    wave_lds_sync();
    f4_t ho[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er, col = ch * 32 + ec;
      const int64_t gr = g0 + row;
      f4_t v = ld4(&stage[row][ec]);
      if (a.drop_p > 0.0f) {
        unsigned int c[4];
        dst::dropout_block(a.seed, a.stream2, (gr * 256 + col) >> 2, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c[e] >= thr ? v[e] * keep_scale : 0.0f;
      }
      ho[it] = ld4(&w.yf[row][col]) + g2c[it] * v;
      if (row < valid) {
        if (a.f2) st4(a.f2 + gr * 256 + col, v);
        st4(a.h_out + gr * 256 + col, ho[it]);
      } else {
        ho[it] = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      }
    }
    // h_out takes y1's bf16 tile (its last readers, ff_linear1's MFMAs, finished before the barrier above; a wave writes its own 32 columns)
#pragma unroll
    for (int it = 0; it < 4; ++it) *reinterpret_cast<bf16x4_t*>(&w.yb[it * 8 + er][ch * 32 + ec]) = to_bf4(ho[it]);
  }
  __syncthreads();
  // ---- input_lin's node parts (256 -> 512, no bias): chunks wave, wave + 8; the read-out slice (256 -> 64 + bias) on waves 0, 1
#pragma unroll 1
  for (int k = 0; k < 3; ++k) {
    const bool ro = k == 2;
    if (ro && wave >= 2) break;
    const int ch = ro ? wave : wave + 8 * k;
    f32x16_t acc;
    const float bro = ro ? a.bn[wave * 32 + (lane & 31)] : 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = bro;
    const uint16_t* Wk = ro ? a.Wn : a.Wac;
    const int nk = ro ? 64 : 512;
    wfetch<8>(fb, Wk, 256, 128, ch * 32, nk);
    mma_apply<8>(&w.yb[0][0], LD_Z, 0, fa, acc);
    mma_apply<8>(&w.yb[0][0], LD_Z, 128, fb, acc);
    if (k == 0) wfetch<8>(fa, a.Wac, 256, 0, (wave + 8) * 32, 512);
    else if (k == 1 && wave < 2) wfetch<8>(fa, a.Wn, 256, 0, wave * 32, 64);
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      const int64_t gr = g0 + row;
      const f4_t v = ld4(&stage[row][ec]);
      if (row < valid) {
        if (!ro) st4(a.ac + gr * 512 + ch * 32 + ec, v);
        else st4(a.rn + gr * 64 + ch * 32 + ec, v);
      }
    }
    wave_lds_sync();
  }
}

// dst_dir_chain_bwd: the backward of the directed rows of a block (coord_mlp and equi_update's LayerNorm; dmt.py:37-48) as one kernel and a
// small finishing kernel instead of dst_gemm (K = 3, SiLU'), dst_gemm (256 -> 256 input gradient) and dst_lnmod_bwd (three launches):
//   dc0 = (dc2 W2) * SiLU'(c0);  dzn = dc0 W0;  dz = LayerNorm'(zz; dzn (1 + scale));  d shift += sum_rows dzn, d scale += sum_rows dzn x^
// TILES ARE MOLECULE-ALIGNED here (a host table: first directed row, row count <= 32, molecule): the adaLN gradients are sums over the rows of
// ONE molecule, so a tile adds its rows up in a fixed order into part[tile][512] and k_dir_bwd_finish adds a molecule's tiles in tile order.
// W2 [3,256] fp32 (K = 3: plain FMAs, as the K <= 8 GEMM kernel); W0T = W0 transposed as bf16 ([in][out]: the B fragment of dc0 W0 is eight
// consecutive `out` of one `in`).  dc0 and dz go to global memory (the weight-gradient product of coord_mlp.0 and dst_zbuild_bwd read them).
struct DirBwdLds {
  float zf[32][LD_YF];             // dzn, fp32: the LayerNorm backward reads whole rows; then the column sums [wave][group][512]
  float d2[32][4];                 // dc2 of the tile's rows
  __bf16 db[32][LD_Z];             // dc0, bf16: A operand of dc0 W0
};
__device__ __forceinline__ float fast_silu_deriv(float x) {
  const float sg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.4426950408889634f));
  return sg * (1.0f + x * (1.0f - sg));
}

__global__ __launch_bounds__(CH_NT, 2) void k_dir_chain_bwd(dst_layout L, dst_dir_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  DirBwdLds& w = *reinterpret_cast<DirBwdLds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int tile = blockIdx.x;
  const int64_t g0 = a.tile_row0[tile];
  const int valid = a.tile_rows[tile], mol = a.tile_mol[tile];
  const int64_t glast = g0 + valid - 1;
  const int sub = lane >> 4, j16 = lane & 15, er = lane >> 3, ec = (lane & 7) * 4;
  WFrag<8> fa, fb;
  wfetch<8>(fa, a.W0T, 256, 0, wave * 32, 256);
  if (threadIdx.x < 96) {
    const int row = threadIdx.x / 3, o = threadIdx.x % 3;
    w.d2[row][o] = row < valid ? a.dc2[(g0 + row) * 3 + o] : 0.0f;
  }
  // ---- dc0 = (dc2 W2) SiLU'(c0): chunks wave, wave + 4; a chunk row = 8 lanes x float4; every load of the phase before its first store
  f4_t cv[2][4], w2v[2][3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int col = (wave + 4 * k) * 32 + ec;
#pragma unroll
    for (int it = 0; it < 4; ++it) cv[k][it] = ld4(a.c0 + min(g0 + it * 8 + er, glast) * 256 + col);
#pragma unroll
    for (int o = 0; o < 3; ++o) w2v[k][o] = ld4(a.W2 + o * 256 + col);
  }
  __syncthreads();                                             // dc2 tile
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int col = (wave + 4 * k) * 32 + ec;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      const float d0 = w.d2[row][0], d1 = w.d2[row][1], d2_ = w.d2[row][2];
      f4_t v;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float t = d0 * w2v[k][0][e];
        t = fmaf(d1, w2v[k][1][e], t);
        t = fmaf(d2_, w2v[k][2][e], t);
        v[e] = t * fast_silu_deriv(cv[k][it][e]);
      }
      if (row < valid) st4(a.dc0 + (g0 + row) * 256 + col, v);
      else v = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      *reinterpret_cast<bf16x4_t*>(&w.db[row][col]) = to_bf4(v);
    }
  }
  __syncthreads();
  // ---- dzn = dc0 W0 (256 -> 256): chunks wave, wave + 4 of the INPUT columns, into the fp32 tile
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    const int ch = wave + 4 * k;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    wfetch<8>(fb, a.W0T, 256, 128, ch * 32, 256);
    mma_apply<8>(&w.db[0][0], LD_Z, 0, fa, acc);
    mma_apply<8>(&w.db[0][0], LD_Z, 128, fb, acc);
    if (k == 0) wfetch<8>(fa, a.W0T, 256, 0, (ch + 4) * 32, 256);
    const int c = lane & 31, hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < 16; ++i) w.zf[(i & 3) + 8 * (i >> 2) + 4 * hh][ch * 32 + c] = acc[i];
  }
  // the LayerNorm backward's operands (wave w: rows 8 w .. 8 w + 7 as two passes of four; a row = 16 lanes, lane j the float4s at columns
  // 4 j + 64 u): requested before the barrier
  f4_t zv[2][4], sc1[4];
  float mean[2], rstd[2];
  {
    const float* adm = a.ada + (int64_t)mol * a.ada_ld + a.scale_off + 4 * j16;
#pragma unroll
    for (int u = 0; u < 4; ++u) sc1[u] = ld4(adm + 64 * u) + 1.0f;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int64_t gr = min(g0 + 8 * wave + 4 * q + sub, glast);
      mean[q] = a.st[gr * 2]; rstd[q] = a.st[gr * 2 + 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) zv[q][u] = ld4(a.zz + gr * 256 + 64 * u + 4 * j16);
    }
  }
  __syncthreads();
  f4_t psh[4], psc[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) { psh[u] = f4_t{0.0f, 0.0f, 0.0f, 0.0f}; psc[u] = psh[u]; }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 8 * wave + 4 * q + sub;
    f4_t g[4], xh[4];
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f4_t dzn = ld4(&w.zf[row][64 * u + 4 * j16]);     // (rows beyond the tile: dc0 = 0 -> dzn = 0)
      xh[u] = (zv[q][u] - mean[q]) * rstd[q];
      psh[u] += dzn;
      psc[u] += dzn * xh[u];
      g[u] = dzn * sc1[u];
      const f4_t gx = g[u] * xh[u];
      s1 += (g[u][0] + g[u][1]) + (g[u][2] + g[u][3]);
      s2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
    }
    const float m1 = sum16_dpp(s1) * (1.0f / 256.0f), m2 = sum16_dpp(s2) * (1.0f / 256.0f);
    if (row < valid) {
#pragma unroll
      for (int u = 0; u < 4; ++u) st4(a.dz + (g0 + row) * 256 + 64 * u + 4 * j16, rstd[q] * (g[u] - m1 - xh[u] * m2));
    }
  }
  // column sums of the wave's eight rows: the four 16-lane groups and the four waves through LDS, added in (wave, group) order (the fp32 dzn
  // tile is dead: every wave has read its rows).  Not with xor-16 / xor-32 shuffles: see k_pair_chain_bwd.
  __syncthreads();
  {
    float* red = &w.zf[0][0] + (wave * 4 + sub) * 512;        // [wave][group][shift 256 | scale 256]
#pragma unroll
    for (int u = 0; u < 4; ++u) { st4(red + 64 * u + 4 * j16, psh[u]); st4(red + 256 + 64 * u + 4 * j16, psc[u]); }
  }
  __syncthreads();
#pragma unroll
  for (int t = threadIdx.x; t < 512; t += CH_NT) {
    const float* red = &w.zf[0][0];
    float s_ = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s_ += red[k * 512 + t];
    a.part[(int64_t)tile * 512 + t] = s_;
  }
}

__global__ __launch_bounds__(512) void k_dir_bwd_finish(dst_dir_bwd_args a) {
  const int m = blockIdx.x, t = threadIdx.x;
  float s = 0.0f;
  for (int k = a.mol_tile_off[m]; k < a.mol_tile_off[m + 1]; ++k) s += a.part[(int64_t)k * 512 + t];
  a.d_ada[(int64_t)m * a.ada_ld + (t < 256 ? a.shift_off + t : a.scale_off + t - 256)] = s;
}

// dst_pair_chain_bwd: the backward of the pair rows of a block BEHIND the attention (the forward of dst_pair_chain_fwd) as one kernel and a
// finishing kernel instead of five dst_gemm input gradients, 2 x dst_gate_add_bwd and dst_lnmod_bwd (eleven launches):
//   de_tot = de + dro Wro + ded Wed[:, e];  dfeat = ded Wed[:, dist]
//   df4 = gate2 de_tot (dropout mask 4);  df3 = (df4 W4) SiLU'(f3) (dropout mask 3);  dye1 = de_tot + df3 W3
//   dxe1 = LayerNorm'(xe1; dye1 (1 + scale));  de_in = dxe1;  dhe = gate1 dxe1
//   d_ada: gate2 += sum de_tot f4, shift += sum dye1, scale += sum dye1 x^, gate1 += sum dxe1 he      (sums over the rows of a molecule)
// Molecule-aligned tiles of 32 pair rows, as dst_dir_chain_bwd.  Weights TRANSPOSED as bf16 ([in][out]): WedT [128][256], WroT [64][16],
// W4T [128][64], W3T [64][128].  df4, df3 (the weight-gradient products read them), dfeat, de_in, dhe go to global memory.
// The pair- and directed-row backward kernels ask for CHAIN_BWD_LDS bytes of LDS - more than they use - so that NO other workgroup that uses LDS
// shares their CU.  Next to a weight-gradient product (other stream) on the same CU, single rows of their LayerNorm stage came out different
// from run to run on identical inputs (1 - 3 % of the training steps; inputs verified unchanged, no shuffles, explicit waits tried); alone on the
// CU, 240 of 240 repeated steps were bit-identical (with 142 kB - only 18 kB kernels beside it - 120 of 120; with 128 kB not: the neighbour that matters is
// k_tr_gemm_bf16<128, 128, 256>, 20 kB of LDS).  The mechanism is not understood (profiles/HISTORY.md, round 5); the cost is ~0.3 ms per step.
constexpr size_t CHAIN_BWD_LDS = 150 * 1024;
struct PairBwdLds {
  float yf[32][LD_F];              // de_tot, the base of dye1
  float stage[CH_NW][32][LD_ST];
  float red[CH_NW * 4][256];       // the column sums of every (wave, 16-lane group): gate2 | shift | scale | gate1
  __bf16 eb[32][LD_Z];             // ded, bf16
  __bf16 rb[32][24];               // dro (16 columns), bf16
  __bf16 fb[32][LD_Y];             // df4, bf16
  __bf16 gb[32][LD_S];             // df3, bf16
};

__global__ __launch_bounds__(CH_NT, 2) void k_pair_chain_bwd(dst_layout L, dst_pair_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  PairBwdLds& w = *reinterpret_cast<PairBwdLds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float (*stage)[LD_ST] = w.stage[wave];
  const int tile = blockIdx.x;
  const int64_t g0 = a.tile_row0[tile];
  const int valid = a.tile_rows[tile], mol = a.tile_mol[tile];
  const int64_t glast = g0 + valid - 1;
  const unsigned int thr = dst::dropout_threshold(a.drop_p);
  const float keep_scale = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const int sub = lane >> 4, j16 = lane & 15, cl = j16 * 4, er = lane >> 3, ec = (lane & 7) * 4;
  const float* adm = a.ada + (int64_t)mol * a.ada_ld;
  WFrag<8> fa, fb;
  wfetch<8>(fa, a.WedT, 256, 0, wave * 32, 128);
  // ---- ded (32 x 256) and dro (32 x 16) as bf16 tiles: wave w rows 8 w .. 8 w + 7, a ded row = 64 lanes x float4
  {
    f4_t dv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) dv[q] = ld4(a.ded + min(g0 + 8 * wave + q, glast) * 256 + lane * 4);
    f4_t rv = {0.0f, 0.0f, 0.0f, 0.0f};
    const int rrow = threadIdx.x >> 2, rc = (threadIdx.x & 3) * 4;              // 128 threads: (row, four of the 16 columns)
    if (threadIdx.x < 128 && rrow < valid) rv = ld4(a.dro + (g0 + rrow) * a.ld_dro + rc);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int row = 8 * wave + q;
      if (row >= valid) dv[q] = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      *reinterpret_cast<bf16x4_t*>(&w.eb[row][lane * 4]) = to_bf4(dv[q]);
    }
    if (threadIdx.x < 128) *reinterpret_cast<bf16x4_t*>(&w.rb[rrow][rc]) = to_bf4(rv);
  }
  __syncthreads();
  // ---- ded Wed (256 -> 128 = e | dist), + dro Wro on the e half: chunk `wave`
  {
    const int ch = wave;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    wfetch<8>(fb, a.WedT, 256, 128, ch * 32, 128);
    WFrag<1> fr;
    if (wave < 2) wfetch<1>(fr, a.WroT, 16, 0, ch * 32, 64);
    mma_apply<8>(&w.eb[0][0], LD_Z, 0, fa, acc);
    mma_apply<8>(&w.eb[0][0], LD_Z, 128, fb, acc);
    if (wave < 2) mma_apply<1>(&w.rb[0][0], 24, 0, fr, acc);
    f4_t dein[4];
    if (wave < 2) {
#pragma unroll
      for (int it = 0; it < 4; ++it) dein[it] = ld4(a.de + min(g0 + it * 8 + er, glast) * 64 + ch * 32 + ec);
    }
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      const f4_t v = ld4(&stage[row][ec]);
      if (wave < 2) st4(&w.yf[row][ch * 32 + ec], row < valid ? v + dein[it] : f4_t{0.0f, 0.0f, 0.0f, 0.0f});
      else if (row < valid) st4(a.dfeat + (g0 + row) * 64 + (ch - 2) * 32 + ec, v);
    }
  }
  // operands of the gated-residual stage (wave w: rows 8 w .. 8 w + 7 as two passes of four; a row = 16 lanes x float4): before the barrier
  const f4_t g2 = ld4(adm + a.gate2_off + cl);
  f4_t f4v[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) f4v[q] = ld4(a.f4 + min(g0 + 8 * wave + 4 * q + sub, glast) * 64 + cl);
  WFrag<4> f4w;
  wfetch<4>(f4w, a.W4T, 64, 0, wave * 32, 128);
  __syncthreads();
  f4_t pg2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 8 * wave + 4 * q + sub;
    const int64_t gr = g0 + row;
    const f4_t d = ld4(&w.yf[row][cl]);                        // (zero beyond the tile)
    pg2 += d * f4v[q];
    f4_t o = g2 * d;
    if (a.drop_p > 0.0f) {
      unsigned int c[4];
      dst::dropout_block(a.seed, a.stream4, (gr * 64 + cl) >> 2, c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = c[e] >= thr ? o[e] * keep_scale : 0.0f;
    }
    if (row < valid) st4(a.df4 + gr * 64 + cl, o);
    else o = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
    *reinterpret_cast<bf16x4_t*>(&w.fb[row][cl]) = to_bf4(o);
  }
  __syncthreads();
  // ---- df3 = (df4 W4) SiLU'(f3), dropout mask 3 (64 -> 128): chunk `wave`
  {
    const int ch = wave;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    mma_apply<4>(&w.fb[0][0], LD_Y, 0, f4w, acc);
    f4_t f3v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) f3v[it] = ld4(a.f3 + min(g0 + it * 8 + er, glast) * 128 + ch * 32 + ec);
    wfetch<4>(f4w, a.W3T, 128, 64 * (wave >> 1), (wave & 1) * 32, 64);          // ff_linear3's input gradient: chunk wave & 1, k-half wave >> 1
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er, col = ch * 32 + ec;
      const int64_t gr = g0 + row;
      f4_t v = ld4(&stage[row][ec]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= fast_silu_deriv(f3v[it][e]);
      if (a.drop_p > 0.0f) {
        unsigned int c[4];
        dst::dropout_block(a.seed, a.stream3, (gr * 128 + col) >> 2, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c[e] >= thr ? v[e] * keep_scale : 0.0f;
      }
      if (row < valid) st4(a.df3 + gr * 128 + col, v);
      else v = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      *reinterpret_cast<bf16x4_t*>(&w.gb[row][col]) = to_bf4(v);
    }
  }
  __syncthreads();
  // ---- df3 W3 (128 -> 64): chunk wave & 1, k-half wave >> 1 -> this wave's staging tile; the two halves are added in the next stage
  {
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    mma_apply<4>(&w.gb[0][0], LD_S, 64 * (wave >> 1), f4w, acc);
    acc_to_stage(acc, stage);
  }
  // operands of the LayerNorm / gate stage: before the barrier
  const f4_t sc1 = ld4(adm + a.scale_off + cl) + 1.0f, g1 = ld4(adm + a.gate1_off + cl);
  f4_t xv[2], hv[2];
  float mean[2], rstd[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int64_t gr = min(g0 + 8 * wave + 4 * q + sub, glast);
    xv[q] = ld4(a.xe1 + gr * 64 + cl); hv[q] = ld4(a.he + gr * 64 + cl);
    mean[q] = a.st[gr * 2]; rstd[q] = a.st[gr * 2 + 1];
  }
  __syncthreads();
  f4_t psh = {0.0f, 0.0f, 0.0f, 0.0f}, psc = psh, pg1 = psh;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int row = 8 * wave + 4 * q + sub;
    const int64_t gr = g0 + row;
    const int chn = j16 >> 3, ecn = (j16 & 7) * 4;
    const f4_t dy = (ld4(&w.yf[row][cl]) + ld4(&w.stage[chn][row][ecn])) + ld4(&w.stage[2 + chn][row][ecn]);   // (zero beyond the tile)
    const f4_t xh = (xv[q] - mean[q]) * rstd[q];
    psh += dy;
    psc += dy * xh;
    const f4_t gg = dy * sc1, gx = gg * xh;
    const float m1 = sum16_dpp((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.0f / 64.0f);
    const float m2 = sum16_dpp((gx[0] + gx[1]) + (gx[2] + gx[3])) * (1.0f / 64.0f);
    const f4_t dx = rstd[q] * (gg - m1 - xh * m2);
    if (row < valid) {
      pg1 += dx * hv[q];
      st4(a.de_in + gr * 64 + cl, dx);
      st4(a.dhe + gr * 64 + cl, g1 * dx);
    }
  }
  // column sums: the four 16-lane groups and the four waves through LDS, added in (wave, group) order.  (Not with xor-16 / xor-32 shuffles: 32
  // ds_bpermute in flight per wave gave results that differed from run to run while other kernels shared the CU - profiles/HISTORY.md, round 5.)
  {
    float* red = &w.red[0][0] + (wave * 4 + sub) * 256;        // [wave][group][gate2 | shift | scale | gate1]
    st4(red + cl, pg2); st4(red + 64 + cl, psh); st4(red + 128 + cl, psc); st4(red + 192 + cl, pg1);
  }
  __syncthreads();
  {
    const float* red = &w.red[0][0];
    float s_ = 0.0f;
#pragma unroll
    for (int k = 0; k < 16; ++k) s_ += red[k * 256 + threadIdx.x];
    a.part[(int64_t)tile * 256 + threadIdx.x] = s_;
  }
}

__global__ __launch_bounds__(256) void k_pair_bwd_finish(dst_pair_bwd_args a) {
  const int m = blockIdx.x, t = threadIdx.x, v = t >> 6, c = t & 63;
  float s = 0.0f;
  for (int k = a.mol_tile_off[m]; k < a.mol_tile_off[m + 1]; ++k) s += a.part[(int64_t)k * 256 + t];
  const int off = v == 0 ? a.gate2_off : v == 1 ? a.shift_off : v == 2 ? a.scale_off : a.gate1_off;
  a.d_ada[(int64_t)m * a.ada_ld + off + c] = s;
}

// dst_node_chain_bwd: the backward of dst_node_chain_fwd (the node rows of a block behind the attention) as one kernel + a finishing kernel
// instead of five dst_gemm input gradients, 2 x dst_gate_add_bwd and dst_lnmod_bwd on the node stream - the attention backward (main stream)
// waits for the end of that chain:
//   dh_tot = dh + drn Wn + dac Wac;  df2 = gate2 dh_tot (dropout mask 2);  df1 = (df2 W2) SiLU'(f1) (dropout mask 1);  dy1 = dh_tot + df1 W1
//   dx1 = LayerNorm'(x1; dy1 (1 + scale));  dh_in = dx1;  dattn = gate1 dx1
//   d_ada: gate2 = sum dh_tot f2, shift = sum dy1, scale = sum dy1 x^, gate1 = sum dx1 attn       (sums over the rows of a molecule)
// Molecule-aligned tiles of <= 32 node rows, eight waves (a wave owns four rows of the row stages - a row = 64 lanes x float4 - and one or two
// 32-column chunks of every product).  Weights TRANSPOSED as bf16 ([in][out]): WacT [256][512], WnT [256][64], W2T [512][256], W1T [256][512].
struct NodeBwdLds {
  float yf[32][LD_YF];             // dh_tot, then dy1
  float stage[NC_NW][32][LD_ST];
  __bf16 ab[32][LD_S2];            // dac, then df1, bf16 (then the waves' column sums, fp32 [8][1024])
  __bf16 fb[32][LD_Z];             // df2, bf16
  __bf16 rb[32][LD_Y];             // drn (64 columns), bf16
};
static_assert(sizeof(__bf16) * 32 * LD_S2 >= sizeof(float) * NC_NW * 1024, "the column sums alias the df1 tile");

__global__ __launch_bounds__(NC_NT) void k_node_chain_bwd(dst_layout L, dst_node_bwd_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  NodeBwdLds& w = *reinterpret_cast<NodeBwdLds*>(lds_raw);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  float (*stage)[LD_ST] = w.stage[wave];
  const int tile = blockIdx.x;
  const int64_t g0 = a.tile_row0[tile];
  const int valid = a.tile_rows[tile], mol = a.tile_mol[tile];
  const int64_t glast = g0 + valid - 1;
  const unsigned int thr = dst::dropout_threshold(a.drop_p);
  const float keep_scale = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const int er = lane >> 3, ec = (lane & 7) * 4, cl = lane * 4;
  const float* adm = a.ada + (int64_t)mol * a.ada_ld;
  WFrag<8> fa, fb;
  wfetch<8>(fa, a.WacT, 512, 0, wave * 32, 256);
  // ---- dac (32 x 512) and drn (32 x 64) as bf16 tiles: wave w rows 4 w .. 4 w + 3
  {
    f4_t dv[4][2], rv[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int64_t gr = min(g0 + 4 * wave + q, glast);
      dv[q][0] = ld4(a.dac + gr * 512 + cl); dv[q][1] = ld4(a.dac + gr * 512 + 256 + cl);
      rv[q] = lane < 16 ? ld4(a.drn + gr * a.ld_drn + cl) : f4_t{0.0f, 0.0f, 0.0f, 0.0f};
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int row = 4 * wave + q;
      if (row >= valid) { dv[q][0] = f4_t{0.0f, 0.0f, 0.0f, 0.0f}; dv[q][1] = dv[q][0]; rv[q] = dv[q][0]; }
      *reinterpret_cast<bf16x4_t*>(&w.ab[row][cl]) = to_bf4(dv[q][0]);
      *reinterpret_cast<bf16x4_t*>(&w.ab[row][256 + cl]) = to_bf4(dv[q][1]);
      if (lane < 16) *reinterpret_cast<bf16x4_t*>(&w.rb[row][cl]) = to_bf4(rv[q]);
    }
  }
  __syncthreads();
  // ---- dh_tot = dh + dac Wac (512 -> 256) + drn Wn (64 -> 256): chunk `wave`
  {
    const int ch = wave;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    WFrag<4> fr;
    wfetch<4>(fr, a.WnT, 64, 0, ch * 32, 256);
    wfetch<8>(fb, a.WacT, 512, 128, ch * 32, 256);
    mma_apply<8>(&w.ab[0][0], LD_S2, 0, fa, acc);
    wfetch<8>(fa, a.WacT, 512, 256, ch * 32, 256);
    mma_apply<8>(&w.ab[0][0], LD_S2, 128, fb, acc);
    wfetch<8>(fb, a.WacT, 512, 384, ch * 32, 256);
    mma_apply<8>(&w.ab[0][0], LD_S2, 256, fa, acc);
    f4_t dhv[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) dhv[it] = ld4(a.dh + min(g0 + it * 8 + er, glast) * 256 + ch * 32 + ec);
    mma_apply<8>(&w.ab[0][0], LD_S2, 384, fb, acc);
    mma_apply<4>(&w.rb[0][0], LD_Y, 0, fr, acc);
    wfetch<8>(fa, a.W2T, 256, 0, wave * 32, 512);              // ff_linear2's input gradient, first chunk
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      const f4_t v = ld4(&stage[row][ec]);
      st4(&w.yf[row][ch * 32 + ec], row < valid ? v + dhv[it] : f4_t{0.0f, 0.0f, 0.0f, 0.0f});
    }
  }
  const f4_t g2 = ld4(adm + a.gate2_off + cl);
  f4_t f2v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) f2v[q] = ld4(a.f2 + min(g0 + 4 * wave + q, glast) * 256 + cl);
  __syncthreads();
  // ---- gated residual of the FF: df2 = gate2 dh_tot (dropout mask 2); wave w rows 4 w .. 4 w + 3
  f4_t pg2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * wave + q;
    const int64_t gr = g0 + row;
    const f4_t d = ld4(&w.yf[row][cl]);
    pg2 += d * f2v[q];
    f4_t o = g2 * d;
    if (a.drop_p > 0.0f) {
      unsigned int c[4];
      dst::dropout_block(a.seed, a.stream2, (gr * 256 + cl) >> 2, c);
#pragma unroll
      for (int e = 0; e < 4; ++e) o[e] = c[e] >= thr ? o[e] * keep_scale : 0.0f;
    }
    if (row < valid) st4(a.df2 + gr * 256 + cl, o);
    else o = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
    *reinterpret_cast<bf16x4_t*>(&w.fb[row][cl]) = to_bf4(o);
  }
  __syncthreads();
  // ---- df1 = (df2 W2) SiLU'(f1), dropout mask 1 (256 -> 512): chunks wave, wave + 8 (the df1 tile takes dac's bytes)
#pragma unroll 1
  for (int k = 0; k < 2; ++k) {
    const int ch = wave + 8 * k;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    wfetch<8>(fb, a.W2T, 256, 128, ch * 32, 512);
    mma_apply<8>(&w.fb[0][0], LD_Z, 0, fa, acc);
    mma_apply<8>(&w.fb[0][0], LD_Z, 128, fb, acc);
    f4_t f1v[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) f1v[it] = ld4(a.f1 + min(g0 + it * 8 + er, glast) * 512 + ch * 32 + ec);
    if (k == 0) wfetch<8>(fa, a.W2T, 256, 0, (ch + 8) * 32, 512);
    else wfetch<8>(fa, a.W1T, 512, 0, wave * 32, 256);         // ff_linear1's input gradient, first k-quarter
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er, col = ch * 32 + ec;
      const int64_t gr = g0 + row;
      f4_t v = ld4(&stage[row][ec]);
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] *= fast_silu_deriv(f1v[it][e]);
      if (a.drop_p > 0.0f) {
        unsigned int c[4];
        dst::dropout_block(a.seed, a.stream1, (gr * 512 + col) >> 2, c);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = c[e] >= thr ? v[e] * keep_scale : 0.0f;
      }
      if (row < valid) st4(a.df1 + gr * 512 + col, v);
      else v = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
      *reinterpret_cast<bf16x4_t*>(&w.ab[row][col]) = to_bf4(v);
    }
    wave_lds_sync();
  }
  __syncthreads();
  // ---- dy1 = dh_tot + df1 W1 (512 -> 256): chunk `wave`, added into the fp32 tile (a wave owns its 32 columns)
  {
    const int ch = wave;
    f32x16_t acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    wfetch<8>(fb, a.W1T, 512, 128, ch * 32, 256);
    mma_apply<8>(&w.ab[0][0], LD_S2, 0, fa, acc);
    wfetch<8>(fa, a.W1T, 512, 256, ch * 32, 256);
    mma_apply<8>(&w.ab[0][0], LD_S2, 128, fb, acc);
    wfetch<8>(fb, a.W1T, 512, 384, ch * 32, 256);
    mma_apply<8>(&w.ab[0][0], LD_S2, 256, fa, acc);
    mma_apply<8>(&w.ab[0][0], LD_S2, 384, fb, acc);
    acc_to_stage(acc, stage);
    wave_lds_sync();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int row = it * 8 + er;
      float* y = &w.yf[row][ch * 32 + ec];
      st4(y, ld4(y) + ld4(&stage[row][ec]));
    }
  }
  // operands of the LayerNorm / gate stage: before the barrier
  const f4_t sc1 = ld4(adm + a.scale_off + cl) + 1.0f, g1 = ld4(adm + a.gate1_off + cl);
  f4_t xv[4], av[4];
  float mean[4], rstd[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int64_t gr = min(g0 + 4 * wave + q, glast);
    xv[q] = ld4(a.x1 + gr * 256 + cl); av[q] = ld4(a.attn + gr * 256 + cl);
    mean[q] = a.st[gr * 2]; rstd[q] = a.st[gr * 2 + 1];
  }
  __syncthreads();
  f4_t psh = {0.0f, 0.0f, 0.0f, 0.0f}, psc = psh, pg1 = psh;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int row = 4 * wave + q;
    const int64_t gr = g0 + row;
    const f4_t dy = ld4(&w.yf[row][cl]);                       // (zero beyond the tile)
    const f4_t xh = (xv[q] - mean[q]) * rstd[q];
    psh += dy;
    psc += dy * xh;
    const f4_t gg = dy * sc1, gx = gg * xh;
    const float m1 = sum64_dpp((gg[0] + gg[1]) + (gg[2] + gg[3])) * (1.0f / 256.0f);
    const float m2 = sum64_dpp((gx[0] + gx[1]) + (gx[2] + gx[3])) * (1.0f / 256.0f);
    const f4_t dx = rstd[q] * (gg - m1 - xh * m2);
    if (row < valid) {
      pg1 += dx * av[q];
      st4(a.dh_in + gr * 256 + cl, dx);
      st4(a.dattn + gr * 256 + cl, g1 * dx);
    }
  }
  // column sums: a lane owns its four columns over the wave's four rows; the waves are added in wave order (the sums take the df1 tile's bytes:
  // its last readers, the MFMAs above, finished before the barrier)
  float* red = reinterpret_cast<float*>(&w.ab[0][0]);
  st4(red + wave * 1024 + cl, pg2); st4(red + wave * 1024 + 256 + cl, psh);
  st4(red + wave * 1024 + 512 + cl, psc); st4(red + wave * 1024 + 768 + cl, pg1);
  __syncthreads();
#pragma unroll
  for (int t = threadIdx.x; t < 1024; t += NC_NT) {
    float s_ = 0.0f;
#pragma unroll
    for (int k = 0; k < NC_NW; ++k) s_ += red[k * 1024 + t];
    a.part[(int64_t)tile * 1024 + t] = s_;
  }
}

__global__ __launch_bounds__(1024) void k_node_bwd_finish(dst_node_bwd_args a) {
  const int m = blockIdx.x, t = threadIdx.x, v = t >> 8, c = t & 255;
  float s = 0.0f;
  for (int k = a.mol_tile_off[m]; k < a.mol_tile_off[m + 1]; ++k) s += a.part[(int64_t)k * 1024 + t];
  const int off = v == 0 ? a.gate2_off : v == 1 ? a.shift_off : v == 2 ? a.scale_off : a.gate1_off;
  a.d_ada[(int64_t)m * a.ada_ld + off + c] = s;
}

}  // namespace

extern "C" {

int dst_pair_chain_fwd(const dst_layout* L, const dst_pair_chain_args* a, void* stream) {
  if (!L || !a || !a->pair_a || !a->pair_b || !a->pair_mol || !a->u || !a->n2e_bias || !a->e_in || !a->feat || !a->ada || !a->W3 || !a->b3 || !a->W4 || !a->b4 || !a->Wed || !a->bed || !a->Wro ||
      !a->bro || !a->e_out || !a->ed || !a->ro)
    return DS_ERR_ARG;
  if (L->B <= 0 || (a->ld_feat & 3) || (a->ld_wed & 7) || (a->ada_ld & 3) || ((a->gate1_off | a->shift_off | a->scale_off | a->gate2_off) & 3) ||
      !(a->drop_p >= 0.0f && a->drop_p < 1.0f))
    return DS_ERR_ARG;
  const void* ptrs[] = {a->u, a->n2e_bias, a->e_in, a->feat, a->ada, a->W3, a->W4, a->Wed, a->Wro, a->he, a->xe1, a->ye1, a->f3, a->s3, a->f4, a->e_out, a->X2, a->ed, a->ro};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;               // 16-byte accesses throughout
  if (L->Pp <= 0) return DS_OK;
  static bool attr_done = false;
  const size_t lds = sizeof(ChainLds);
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair_chain_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(k_pair_chain_fwd, dim3((L->Pp + 31) / 32), dim3(CH_NT), lds, (hipStream_t)stream, *L, *a, a->pair_a, a->pair_b, a->pair_mol);
  return DST_CHECK_LAUNCH();
}

int dst_pair_front_fwd(const dst_layout* L, const dst_pair_front_args* a, void* stream) {
  if (!L || !a || !a->pair_a || !a->pair_b || !a->pair_mol || !a->pos || !a->ada || !a->means || !a->stds || !a->e_in || !a->Wee || !a->bee || !a->Wte || !a->X1 || !a->te) return DS_ERR_ARG;
  if (L->B <= 0 || (a->ada_ld & 3) || ((a->shift_off | a->scale_off) & 3)) return DS_ERR_ARG;
  const void* ptrs[] = {a->ada, a->e_in, a->Wee, a->Wte, a->X1, a->e1, a->en, a->te};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;
  if (L->Pp <= 0) return DS_OK;
  static bool attr_done = false;
  const size_t lds = sizeof(FrontLds);
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair_front_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(k_pair_front_fwd, dim3((L->Pp + 31) / 32), dim3(CH_NT), lds, (hipStream_t)stream, *L, *a, a->pair_a, a->pair_b, a->pair_mol);
  return DST_CHECK_LAUNCH();
}

int dst_dir_chain_fwd(const dst_layout* L, const dst_dir_chain_args* a, void* stream) {
  if (!L || !a || !a->pair_a || !a->pair_b || !a->pair_mol || !a->ac || !a->ed || !a->ada || !a->W0 || !a->b0 || !a->W2 || !a->c2) return DS_ERR_ARG;
  if (L->B <= 0 || (a->ada_ld & 3) || ((a->shift_off | a->scale_off) & 3)) return DS_ERR_ARG;
  const void* ptrs[] = {a->ac, a->ed, a->ada, a->W0, a->W2, a->zz, a->zn, a->c0, a->sc0};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;
  if (L->Pp <= 0) return DS_OK;
  static bool attr_done = false;
  const size_t lds = sizeof(DirLds);
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dir_chain_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(k_dir_chain_fwd, dim3((2 * L->Pp + 31) / 32), dim3(CH_NT), lds, (hipStream_t)stream, *L, *a, a->pair_a, a->pair_b, a->pair_mol);
  return DST_CHECK_LAUNCH();
}

int dst_node_chain_fwd(const dst_layout* L, const dst_node_chain_args* a, void* stream) {
  if (!L || !a || !a->node_mol || !a->h_in || !a->attn || !a->ada || !a->W1 || !a->b1 || !a->W2 || !a->b2 || !a->Wac || !a->Wn || !a->bn || !a->h_out || !a->ac ||
      !a->rn)
    return DS_ERR_ARG;
  if (L->B <= 0 || (a->ada_ld & 3) || ((a->gate1_off | a->shift_off | a->scale_off | a->gate2_off) & 3) || !(a->drop_p >= 0.0f && a->drop_p < 1.0f))
    return DS_ERR_ARG;
  const void* ptrs[] = {a->h_in, a->attn, a->ada, a->W1, a->W2, a->Wac, a->Wn, a->x1, a->y1, a->f1, a->s1, a->f2, a->h_out, a->ac, a->rn};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;
  if (L->Nn <= 0) return DS_OK;
  static bool attr_done = false;
  const size_t lds = sizeof(NodeLds);
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_node_chain_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(k_node_chain_fwd, dim3((L->Nn + 31) / 32), dim3(NC_NT), lds, (hipStream_t)stream, *L, *a);
  return DST_CHECK_LAUNCH();
}

int dst_dir_chain_bwd(const dst_layout* L, const dst_dir_bwd_args* a, void* stream) {
  if (!L || !a || !a->tile_row0 || !a->tile_rows || !a->tile_mol || !a->mol_tile_off || !a->dc2 || !a->c0 || !a->zz || !a->st || !a->ada || !a->d_ada || !a->W2 ||
      !a->W0T || !a->dc0 || !a->dz || !a->part)
    return DS_ERR_ARG;
  if (L->B <= 0 || a->n_tiles < 0 || (a->ada_ld & 3) || ((a->shift_off | a->scale_off) & 3)) return DS_ERR_ARG;
  const void* ptrs[] = {a->c0, a->zz, a->ada, a->W2, a->W0T, a->dc0, a->dz};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (a->n_tiles > 0) {
    static bool attr_done = false;
    static_assert(sizeof(DirBwdLds) <= CHAIN_BWD_LDS, "");
    const size_t lds = CHAIN_BWD_LDS;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dir_chain_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
      attr_done = true;
    }
    hipLaunchKernelGGL(k_dir_chain_bwd, dim3(a->n_tiles), dim3(CH_NT), lds, s, *L, *a);
  }
  hipLaunchKernelGGL(k_dir_bwd_finish, dim3(L->B), dim3(512), 0, s, *a);     // (molecules without pairs: zero sums)
  return DST_CHECK_LAUNCH();
}

int dst_pair_chain_bwd(const dst_layout* L, const dst_pair_bwd_args* a, void* stream) {
  if (!L || !a || !a->tile_row0 || !a->tile_rows || !a->tile_mol || !a->mol_tile_off || !a->de || !a->dro || !a->ded || !a->f4 || !a->f3 || !a->xe1 || !a->st ||
      !a->he || !a->ada || !a->d_ada || !a->WedT || !a->WroT || !a->W4T || !a->W3T || !a->dfeat || !a->df4 || !a->df3 || !a->de_in || !a->dhe || !a->part)
    return DS_ERR_ARG;
  if (L->B <= 0 || a->n_tiles < 0 || (a->ada_ld & 3) || (a->ld_dro & 3) || ((a->gate1_off | a->shift_off | a->scale_off | a->gate2_off) & 3) ||
      !(a->drop_p >= 0.0f && a->drop_p < 1.0f))
    return DS_ERR_ARG;
  const void* ptrs[] = {a->de, a->dro, a->ded, a->f4, a->f3, a->xe1, a->he, a->ada, a->WedT, a->WroT, a->W4T, a->W3T, a->dfeat, a->df4, a->df3, a->de_in, a->dhe};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (a->n_tiles > 0) {
    static bool attr_done = false;
    static_assert(sizeof(PairBwdLds) <= CHAIN_BWD_LDS, "");
    const size_t lds = CHAIN_BWD_LDS;
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair_chain_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
      attr_done = true;
    }
    hipLaunchKernelGGL(k_pair_chain_bwd, dim3(a->n_tiles), dim3(CH_NT), lds, s, *L, *a);
  }
  hipLaunchKernelGGL(k_pair_bwd_finish, dim3(L->B), dim3(256), 0, s, *a);
  return DST_CHECK_LAUNCH();
}

int dst_node_chain_bwd(const dst_layout* L, const dst_node_bwd_args* a, void* stream) {
  if (!L || !a || !a->tile_row0 || !a->tile_rows || !a->tile_mol || !a->mol_tile_off || !a->dh || !a->drn || !a->dac || !a->f2 || !a->f1 || !a->x1 || !a->st ||
      !a->attn || !a->ada || !a->d_ada || !a->WacT || !a->WnT || !a->W2T || !a->W1T || !a->df2 || !a->df1 || !a->dh_in || !a->dattn || !a->part)
    return DS_ERR_ARG;
  if (L->B <= 0 || a->n_tiles < 0 || (a->ada_ld & 3) || (a->ld_drn & 3) || ((a->gate1_off | a->shift_off | a->scale_off | a->gate2_off) & 3) ||
      !(a->drop_p >= 0.0f && a->drop_p < 1.0f))
    return DS_ERR_ARG;
  const void* ptrs[] = {a->dh, a->drn, a->dac, a->f2, a->f1, a->x1, a->attn, a->ada, a->WacT, a->WnT, a->W2T, a->W1T, a->df2, a->df1, a->dh_in, a->dattn};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (a->n_tiles > 0) {
    static bool attr_done = false;
    const size_t lds = sizeof(NodeBwdLds);
    if (!attr_done) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_node_chain_bwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
      attr_done = true;
    }
    hipLaunchKernelGGL(k_node_chain_bwd, dim3(a->n_tiles), dim3(NC_NT), lds, s, *L, *a);
  }
  hipLaunchKernelGGL(k_node_bwd_finish, dim3(L->B), dim3(1024), 0, s, *a);
  return DST_CHECK_LAUNCH();
}

}  // extern "C"
