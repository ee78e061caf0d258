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
// One workgroup = (molecule, split): the adaLN rows are per molecule and a molecule's pair rows are contiguous.  A wave owns 32-row tiles
// from its first load to its last store (wave-private LDS, no workgroup barrier after the pair tables).
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

struct WaveLds {
  float yf[32][LD_F];            // ye1, fp32 (the residual of the FF)
  float stage[32][LD_ST];        // one 32 x 32 accumulator tile on its way from the MFMA layout to rows
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
__device__ __forceinline__ bf16x4_t to_bf4(f4_t v) {
  bf16x4_t r;
#pragma unroll
  for (int j = 0; j < 4; ++j) r[j] = (__bf16)v[j];
  return r;
}
__device__ __forceinline__ int pair_idx(int n, int a, int b) { return a * (2 * n - a - 1) / 2 + (b - a - 1); }
__device__ __forceinline__ void wave_lds_sync() {          // wave-private LDS: order this wave's writes before its reads
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
}

// acc += A[32 x 16 KB] W^T for output columns col0 .. col0 + 31: A = bf16 rows in LDS (row stride lda halves, first column a0), W = torch
// Linear weight [out, in] in fp32 with row stride ldw (first input column w0), rounded to bf16 here; output columns >= n_out are zero.
template <int KB>
__device__ __forceinline__ void mma_rows(const __bf16* A, int lda, int a0, const float* __restrict__ W, int64_t ldw, int w0, int col0, int n_out,
                                         f32x16_t& acc) {
  const int lane = threadIdx.x & 63, r = lane & 31, hh = lane >> 5;
  const int col = col0 + r;
  const float* wrow = W + (int64_t)min(col, n_out - 1) * ldw + w0 + 8 * hh;
  const __bf16* arow = A + r * lda + a0 + 8 * hh;
  f4_t wa[KB], wb[KB];
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) { wa[kb] = ld4(wrow + 16 * kb); wb[kb] = ld4(wrow + 16 * kb + 4); }
#pragma unroll
  for (int kb = 0; kb < KB; ++kb) {
    bf16x8_t b;
#pragma unroll
    for (int j = 0; j < 4; ++j) { b[j] = (__bf16)(col < n_out ? wa[kb][j] : 0.0f); b[4 + j] = (__bf16)(col < n_out ? wb[kb][j] : 0.0f); }
    const bf16x8_t a = *reinterpret_cast<const bf16x8_t*>(arow + 16 * kb);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
}
__device__ __forceinline__ void acc_to_stage(const f32x16_t& acc, float (*stage)[LD_ST]) {   // accumulator: lane = column, register i = row
  const int lane = threadIdx.x & 63, c = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 16; ++i) stage[(i & 3) + 8 * (i >> 2) + 4 * hh][c] = acc[i];
}

__global__ __launch_bounds__(CH_NT) void k_pair_chain_fwd(dst_layout L, dst_pair_chain_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ unsigned char pa[406], pb[406];
  WaveLds& w = reinterpret_cast<WaveLds*>(lds_raw)[threadIdx.x >> 6];
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  for (int i = threadIdx.x; i < n; i += CH_NT)
    for (int j = i + 1; j < n; ++j) { const int idx = pair_idx(n, i, j); pa[idx] = (unsigned char)i; pb[idx] = (unsigned char)j; }
  __syncthreads();
  const float* adm = a.ada + (int64_t)m * a.ada_ld;
  const unsigned int thr = dst::dropout_threshold(a.drop_p);
  const float keep_scale = a.drop_p > 0.0f ? 1.0f / (1.0f - a.drop_p) : 1.0f;
  const int sub = lane >> 4, cl = (lane & 15) * 4;          // row layout of stage 1: a row = 16 lanes x float4
  const int er = lane >> 3, ec = (lane & 7) * 4;            // row layout of the GEMM epilogues: a 32-column chunk row = 8 lanes x float4
  const int ntiles = (np + 31) >> 5;
  for (int tile = wave + CH_NW * blockIdx.y; tile < ntiles; tile += CH_NW * gridDim.y) {
    const int t0 = tile * 32, valid = min(32, np - t0);
    const int64_t g0 = (int64_t)p0 + t0;                    // global pair row of the tile's row 0
    // ---- stage 1: gather, gated residual, LayerNorm + modulate
    {
      const f4_t bias = ld4(a.n2e_bias + cl), g1 = ld4(adm + a.gate1_off + cl), sh = ld4(adm + a.shift_off + cl), sc = ld4(adm + a.scale_off + cl);
#pragma unroll 2
      for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 4 + sub, pl = min(t0 + row, np - 1);
        const int64_t gp = (int64_t)p0 + pl;
        const f4_t he = (ld4(a.u + (int64_t)(n0 + pa[pl]) * 64 + cl) + ld4(a.u + (int64_t)(n0 + pb[pl]) * 64 + cl)) + bias;
        const f4_t x = ld4(a.e_in + gp * 64 + cl) + g1 * he;
        f4_t ft = ld4(a.feat + gp * a.ld_feat + cl);
        const float mean = sum16((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 64.0f);
        const f4_t d = x - mean;
        const float rstd = 1.0f / sqrtf(sum16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 64.0f) + 1e-6f);
        f4_t y = (d * rstd) * (1.0f + sc) + sh;
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
    wave_lds_sync();
    // ---- ff_linear3 (64 -> 128), SiLU, dropout
#pragma unroll 1
    for (int ch = 0; ch < 4; ++ch) {
      f32x16_t acc;
      const float b = a.b3[ch * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = b;
      mma_rows<4>(&w.yb[0][0], LD_Y, 0, a.W3, 64, 0, ch * 32, 128, acc);
      acc_to_stage(acc, w.stage);
      wave_lds_sync();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + er, col = ch * 32 + ec;
        const int64_t gr = g0 + row;
        const f4_t v = ld4(&w.stage[row][ec]);
        f4_t s;
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] = dst::act_apply(v[e], 1);
        if (a.drop_p > 0.0f) {
          unsigned int c[4];
          dst::dropout_block(a.seed, a.stream3, (gr * 128 + col) >> 2, c);
#pragma unroll
          for (int e = 0; e < 4; ++e) s[e] = c[e] >= thr ? s[e] * keep_scale : 0.0f;
        }
        if (row < valid) {
          if (a.f3) st4(a.f3 + gr * 128 + col, v);
          if (a.s3) st4(a.s3 + gr * 128 + col, s);
        } else {
          s = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
        }
        *reinterpret_cast<bf16x4_t*>(&w.sb[row][col]) = to_bf4(s);
      }
      wave_lds_sync();
    }
    // ---- ff_linear4 (128 -> 64), dropout, gated residual
#pragma unroll 1
    for (int ch = 0; ch < 2; ++ch) {
      f32x16_t acc;
      const float b = a.b4[ch * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = b;
      mma_rows<8>(&w.sb[0][0], LD_S, 0, a.W4, 128, 0, ch * 32, 64, acc);
      acc_to_stage(acc, w.stage);
      wave_lds_sync();
      const f4_t g2 = ld4(adm + a.gate2_off + ch * 32 + ec);
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + er, col = ch * 32 + ec;
        const int64_t gr = g0 + row;
        f4_t v = ld4(&w.stage[row][ec]);
        if (a.drop_p > 0.0f) {
          unsigned int c[4];
          dst::dropout_block(a.seed, a.stream4, (gr * 64 + col) >> 2, c);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = c[e] >= thr ? v[e] * keep_scale : 0.0f;
        }
        f4_t eo = ld4(&w.yf[row][col]) + g2 * v;
        if (row < valid) {
          if (a.f4) st4(a.f4 + gr * 64 + col, v);
          st4(a.e_out + gr * 64 + col, eo);
          if (a.X2) st4(a.X2 + gr * 128 + col, eo);
        } else {
          eo = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
        }
        *reinterpret_cast<bf16x4_t*>(&w.eb[row][col]) = to_bf4(eo);
      }
      wave_lds_sync();
    }
    // ---- input_lin's edge part ([e_out | features] 128 -> 256) and the read-out slice (e_out 64 -> 16)
#pragma unroll 1
    for (int ch = 0; ch < 9; ++ch) {
      const bool ro = ch == 8;
      f32x16_t acc;
      const int oc = lane & 31;
      const float b = ro ? (oc < 16 ? a.bro[oc] : 0.0f) : a.bed[ch * 32 + oc];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = b;
      if (ro) mma_rows<4>(&w.eb[0][0], LD_S, 0, a.Wro, 64, 0, 0, 16, acc);
      else mma_rows<8>(&w.eb[0][0], LD_S, 0, a.Wed, a.ld_wed, 0, ch * 32, 256, acc);
      acc_to_stage(acc, w.stage);
      wave_lds_sync();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + er;
        const int64_t gr = g0 + row;
        const f4_t v = ld4(&w.stage[row][ec]);
        if (row < valid) {
          if (!ro) st4(a.ed + gr * 256 + ch * 32 + ec, v);
          else if (ec < 16) st4(a.ro + gr * 16 + ec, v);
        }
      }
      wave_lds_sync();
    }
  }
}


// dst_pair_front_fwd: the pair rows of a block IN FRONT of the attention (dmt.py:136-139,145-149; layers.py:291-295,328-334,165-166,183):
//   d2 = |pos_a - pos_b|^2;  x' = d2 (1 + ada[dist]) + ada[dist + 1];  feat = [x', gaussian_k(x')];  X1 = [feat | e]
//   e1 = edge_emb(X1);  en = LN(e1) (1 + ada[scale]) + ada[shift];  te = tanh(en [lin_edge0 | lin_edge1]^T)
// replacing dst_geom_fwd, a copy, two dst_gemm calls and dst_lnmod_fwd.  Same arithmetic per element as those kernels (expf, the
// truncated-pi constant, divisions where they divide), bf16-rounded MFMA operands with fp32 accumulation.
#define DST_GAUSS_A 2.50662732f /* fp32((2 * 3.14159) ** 0.5), as in ds_train.hip */
struct FrontLds {
  float ef[32][LD_F];            // e1 (both 32-column chunks), fp32: the LayerNorm reads whole rows
  float stage[32][LD_ST];
  __bf16 xb[32][LD_S];           // X1 = [feat | e], bf16
  __bf16 nb[32][LD_Y];           // en, bf16
};

__global__ __launch_bounds__(CH_NT) void k_pair_front_fwd(dst_layout L, dst_pair_front_args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  __shared__ unsigned char pa[406], pb[406];
  __shared__ float sp[29][4];
  FrontLds& w = reinterpret_cast<FrontLds*>(lds_raw)[threadIdx.x >> 6];
  const int m = blockIdx.x, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0, p0 = L.pair_off[m], np = n * (n - 1) / 2;
  for (int i = threadIdx.x; i < n; i += CH_NT)
    for (int j = i + 1; j < n; ++j) { const int idx = pair_idx(n, i, j); pa[idx] = (unsigned char)i; pb[idx] = (unsigned char)j; }
  for (int i = threadIdx.x; i < n * 3; i += CH_NT) sp[i / 3][i % 3] = a.pos[(int64_t)(n0 + i / 3) * 3 + i % 3];
  __syncthreads();
  const float* adm = a.ada + (int64_t)m * a.ada_ld;
  const float dsc = adm[a.dist_off], dsh = adm[a.dist_off + 1];
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
  const int ntiles = (np + 31) >> 5;
  for (int tile = wave + CH_NW * blockIdx.y; tile < ntiles; tile += CH_NW * gridDim.y) {
    const int t0 = tile * 32, valid = min(32, np - t0);
    const int64_t g0 = (int64_t)p0 + t0;
    // ---- features + X1
#pragma unroll 2
    for (int ps = 0; ps < 8; ++ps) {
      const int row = ps * 4 + sub, pl = min(t0 + row, np - 1);
      const int64_t gp = (int64_t)p0 + pl;
      const int ia = pa[pl], ib = pb[pl];
      const float dx = sp[ia][0] - sp[ib][0], dy = sp[ia][1] - sp[ib][1], dz = sp[ia][2] - sp[ib][2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      const float x = d2 * (dsc + 1.0f) + dsh;
      f4_t ft;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float u = (x - mu[j]) / sd[j];
        ft[j] = expf(-0.5f * (u * u)) / nrm[j];
      }
      if (cl == 0) ft[0] = x;
      f4_t ev = ld4(a.e_in + gp * 64 + cl);
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
    wave_lds_sync();
    // ---- edge_emb (128 -> 64)
#pragma unroll 1
    for (int ch = 0; ch < 2; ++ch) {
      f32x16_t acc;
      const float b = a.bee[ch * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = b;
      mma_rows<8>(&w.xb[0][0], LD_S, 0, a.Wee, 128, 0, ch * 32, 64, acc);
      const int c = lane & 31, hh = lane >> 5;
#pragma unroll
      for (int i = 0; i < 16; ++i) w.ef[(i & 3) + 8 * (i >> 2) + 4 * hh][ch * 32 + c] = acc[i];
    }
    wave_lds_sync();
    // ---- LayerNorm + modulate
    {
      const f4_t sh = ld4(adm + a.shift_off + cl), sc = ld4(adm + a.scale_off + cl);
#pragma unroll 2
      for (int ps = 0; ps < 8; ++ps) {
        const int row = ps * 4 + sub;
        const int64_t gp = g0 + row;
        const f4_t x = ld4(&w.ef[row][cl]);
        const float mean = sum16((x[0] + x[1]) + (x[2] + x[3])) * (1.0f / 64.0f);
        const f4_t d = x - mean;
        const float rstd = 1.0f / sqrtf(sum16((d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3])) * (1.0f / 64.0f) + 1e-6f);
        f4_t y = (d * rstd) * (1.0f + sc) + sh;
        if (row < valid) {
          if (a.e1) st4(a.e1 + gp * 64 + cl, x);
          if (a.st && (lane & 15) == 0) { a.st[gp * 2] = mean; a.st[gp * 2 + 1] = rstd; }
          if (a.en) st4(a.en + gp * 64 + cl, y);
        } else {
          y = f4_t{0.0f, 0.0f, 0.0f, 0.0f};
        }
        *reinterpret_cast<bf16x4_t*>(&w.nb[row][cl]) = to_bf4(y);
      }
    }
    wave_lds_sync();
    // ---- tanh(en [lin_edge0 | lin_edge1]^T) (64 -> 512)
#pragma unroll 1
    for (int ch = 0; ch < 16; ++ch) {
      f32x16_t acc;
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
      mma_rows<4>(&w.nb[0][0], LD_Y, 0, a.Wte, 64, 0, ch * 32, 512, acc);
      acc_to_stage(acc, w.stage);
      wave_lds_sync();
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + er;
        f4_t v = ld4(&w.stage[row][ec]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = dst::act_apply(v[e], 3);
        if (row < valid) st4(a.te + (g0 + row) * 512 + ch * 32 + ec, v);
      }
      wave_lds_sync();
    }
  }
}

}  // namespace

extern "C" {

int dst_pair_chain_fwd(const dst_layout* L, const dst_pair_chain_args* a, void* stream) {
  if (!L || !a || !a->u || !a->n2e_bias || !a->e_in || !a->feat || !a->ada || !a->W3 || !a->b3 || !a->W4 || !a->b4 || !a->Wed || !a->bed || !a->Wro ||
      !a->bro || !a->e_out || !a->ed || !a->ro)
    return DS_ERR_ARG;
  if (L->B <= 0 || (a->ld_feat & 3) || (a->ld_wed & 3) || (a->ada_ld & 3) || ((a->gate1_off | a->shift_off | a->scale_off | a->gate2_off) & 3) ||
      !(a->drop_p >= 0.0f && a->drop_p < 1.0f))
    return DS_ERR_ARG;
  const void* ptrs[] = {a->u, a->n2e_bias, a->e_in, a->feat, a->ada, a->W3, a->W4, a->Wed, a->Wro, a->he, a->xe1, a->ye1, a->f3, a->s3, a->f4, a->e_out, a->X2, a->ed, a->ro};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;               // 16-byte accesses throughout
  if (L->Pp <= 0) return DS_OK;
  static bool attr_done = false;
  const size_t lds = sizeof(WaveLds) * CH_NW;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair_chain_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(k_pair_chain_fwd, dim3(L->B, 2), dim3(CH_NT), lds, (hipStream_t)stream, *L, *a);
  return DST_CHECK_LAUNCH();
}

int dst_pair_front_fwd(const dst_layout* L, const dst_pair_front_args* a, void* stream) {
  if (!L || !a || !a->pos || !a->ada || !a->means || !a->stds || !a->e_in || !a->Wee || !a->bee || !a->Wte || !a->X1 || !a->te) return DS_ERR_ARG;
  if (L->B <= 0 || (a->ada_ld & 3) || ((a->shift_off | a->scale_off) & 3)) return DS_ERR_ARG;
  const void* ptrs[] = {a->ada, a->e_in, a->Wee, a->Wte, a->X1, a->e1, a->en, a->te};
  for (const void* p : ptrs)
    if (reinterpret_cast<uintptr_t>(p) & 15) return DS_ERR_ARG;
  if (L->Pp <= 0) return DS_OK;
  static bool attr_done = false;
  const size_t lds = sizeof(FrontLds) * CH_NW;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pair_front_fwd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return DS_ERR_LAUNCH;
    attr_done = true;
  }
  hipLaunchKernelGGL(k_pair_front_fwd, dim3(L->B, 2), dim3(CH_NT), lds, (hipStream_t)stream, *L, *a);
  return DST_CHECK_LAUNCH();
}

}  // extern "C"
