// gfx950 kernels + C-ABI for the DMT denoiser evaluation, the ancestral update and post-processing.
// Reference arithmetic being reproduced: models/dmt.py:306-412 and the files cited in include/diffspectra_hip.h.
// Layout and fusion plan: DESIGN.md §3-§4.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <utility>
#include <vector>

#include "../../include/diffspectra_hip.h"
#include "ds_device.h"

#define ADAC DS_ADA_COLS
// k-blocks of weights in flight per wave in the 16-block GEMMs (ds_device.h, wave_mma_h_deep); measured 4 / 8 / 12: 8
#ifndef DS_NODE_PF
#define DS_NODE_PF 8
#endif
#ifndef DS_NODE_PF2
#define DS_NODE_PF2 4   // per ring when two chunks share the X fragments
#endif
#ifndef DS_QKV_PF
#define DS_QKV_PF 4
#endif
#ifndef DS_TWO_STREAM_MAX_PAIRS
#define DS_TWO_STREAM_MAX_PAIRS 400000   // ds_forward: side stream for batches below this many pair rows (~2 500 molecules)
#endif
#ifndef DS_ATTN_FMA
#define DS_ATTN_FMA 1
#endif
#ifndef DS_EQUI_T2
#define DS_EQUI_T2 1   // k_equi_pairs: a MFMA wave runs its two feature chunks against shared X fragments
#endif
#ifndef DS_EQUI_NCW
#define DS_EQUI_NCW 4   // MFMA waves of k_equi_pairs (each owns 8 / NCW feature chunks)
#define DS_EQUI_NLW 4   // loader waves (each owns 32 / NLW pairs of a tile); 4 + 4 leaves both roles 256 registers
#endif

// Diagnostic build only (-DDS_STAMPS): per-phase shader-clock sums of wave 0 of every workgroup, accumulated into
// registers and flushed once at kernel exit to the tail of ws.flags (64-bit counters at int32 index 16 + 2*phase).  Never compiled into the shipped library.
#ifdef DS_STAMPS
#define DS_STAMP_INIT()                                                                                 \
  unsigned long long _sa[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};                        \
  unsigned long long _t0 = __builtin_amdgcn_s_memtime()
#define DS_STAMP(i)                                                                                     \
  do {                                                                                                  \
    const unsigned long long _t1 = __builtin_amdgcn_s_memtime();                                        \
    _sa[i] += _t1 - _t0;                                                                                \
    _t0 = _t1;                                                                                          \
  } while (0)
#define DS_STAMP_W(i)                                                                                   \
  do {                                                                                                  \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                         \
    DS_STAMP(i);                                                                                        \
  } while (0)
#define DS_STAMP_FLUSH(t)                                                                               \
  do {                                                                                                  \
    if (threadIdx.x == (t))                                                                             \
      for (int _i = 0; _i < 16; ++_i)                                                                   \
        if (_sa[_i]) atomicAdd(reinterpret_cast<unsigned long long*>(c.ws.flags + 16) + _i, _sa[_i]);   \
  } while (0)
#else
#define DS_STAMP_INIT()
#define DS_STAMP(i)
#define DS_STAMP_W(i)
#define DS_STAMP_FLUSH(t)
#endif

namespace {

struct Ctx {  // by-value kernel argument: everything a stage needs
  ds_layout L;
  ds_workspace ws;
  const float* wbase;
  const int64_t* woff;  // device copy of ds_weights::off
  float edge_th, cutoff;
};

__device__ __forceinline__ const float* BW(const Ctx& c, int blk, int slot) {
  return c.wbase + c.woff[blk * DS_W_BLOCK_SLOTS + slot];
}
__device__ __forceinline__ const float* GW(const Ctx& c, int slot) {
  return c.wbase + c.woff[DS_NBLOCKS * DS_W_BLOCK_SLOTS + slot];
}

// CondGaussianLayer feature k of the modulated squared distance x (layers.py:291-295,331-334).
__device__ __forceinline__ float rbf_feature(float x, int k, const float* __restrict__ mean, const float* __restrict__ stdv,
                                             const float* __restrict__ astd) {
  if (k == 0) return x;
  const float z = (x - mean[k - 1]) / stdv[k - 1];
  return expf(-0.5f * (z * z)) / astd[k - 1];
}

// ------------------------------------------------------------------------------------------------
// Prologue: sinusoid features of the noise level (layers.py:283-288), [B,24] (17 used, rest 0).
__global__ void k_time_feat(Ctx c, const float* __restrict__ noise_level) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= c.L.B) return;
  const float* w = GW(c, DS_GW_SIN_W);
  const float x = noise_level[b];
  float* f = c.ws.tfeat + (size_t)b * 24;
  f[0] = x;
  for (int i = 0; i < 8; ++i) {
    const float fr = ((x * w[i]) * 2.0f) * 3.14159265358979323846f;
    f[1 + i] = sinf(fr);
    f[9 + i] = cosf(fr);
  }
  for (int i = 17; i < 24; ++i) f[i] = 0.0f;
}

// ------------------------------------------------------------------------------------------------
// Init (dmt.py:323-345,363-377): per-pair adjacency bits + "any non-zero conditioning distance" flag.
__global__ void k_pair_flags(Ctx c, const float* __restrict__ cond_x, const float* __restrict__ cond_edge_x) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const bool valid = p < c.L.Pp;
  int bits = 3;
  bool nonzero = false;
  if (valid && cond_x != nullptr) {
    const int da = c.L.node_dense[c.L.pair_a[p]], db = c.L.node_dense[c.L.pair_b[p]];
    const float* pa = cond_x + (size_t)da * 9;
    const float* pb = cond_x + (size_t)db * 9;
    const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
    const float d2 = dx * dx + dy * dy + dz * dz;
    const int m = c.L.pair_mol[p];
    const int lb = db - m * c.L.N;
    const float ce = cond_edge_x[((size_t)da * c.L.N + lb) * 2 + 0];
    bits = (ce >= c.edge_th ? 1 : 0) | (d2 <= c.cutoff ? 2 : 0);
    nonzero = d2 != 0.0f;
  }
  if (valid) c.ws.adj[p] = bits;
  // "distances.sum() == 0" (dmt.py:364): the flag is raised by a plain store of 1 per workgroup that saw a non-zero distance.
  // Read-modify-write atomics on the one address - per pair, then per wave - serialised at the L2 and WERE this kernel's time
  // (0.13 ms with one atomicOr per wave, 8.6 us without); every writer stores the same value, so no atomic is needed, and
  // 1024-thread workgroups keep the writers few (same-address stores still cost ~35 ns each at the L2).
  if (cond_x != nullptr) {
    const int any = __syncthreads_or(nonzero ? 1 : 0);
    if (any && threadIdx.x == 0) __hip_atomic_store(&c.ws.flags[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// Node init: packed positions, h0 = node_emb([h, cond_h]) (12 -> 256), atom_hids[:, 0:256] = h0.
__global__ __launch_bounds__(256) void k_node_init(Ctx c, const float* __restrict__ xh, const float* __restrict__ cond_x) {
  constexpr int R = 8;                       // rows per workgroup: one workgroup per row ran at the dispatcher's rate (73 k workgroups)
  const int row0 = blockIdx.x * R;
  const int col = threadIdx.x;
  __shared__ __attribute__((aligned(16))) float in[R][16];
  if (col < R * 16) {
    const int rr = col >> 4, k = col & 15, row = row0 + rr;
    float v = 0.0f;
    if (row < c.L.Nn) {
      const int d = c.L.node_dense[row];
      if (k < 6) v = xh[(size_t)d * 9 + 3 + k];
      else if (k < 12) v = cond_x ? cond_x[(size_t)d * 9 + 3 + (k - 6)] : 0.0f;
      else if (k < 15) v = xh[(size_t)d * 9 + (k - 12)];      // positions ride along in slots 12..14
    }
    in[rr][k] = v;
  }
  __syncthreads();
  const float* W = GW(c, DS_GW_NODE_EMB_W);
  float w[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) w[k] = wp_at(W, 256, k, col);
  const float bias = GW(c, DS_GW_NODE_EMB_B)[col];
#pragma unroll
  for (int rr = 0; rr < R; ++rr) {
    const int row = row0 + rr;
    if (row >= c.L.Nn) break;
    float acc = 0.0f;
#pragma unroll
    for (int k = 0; k < 12; ++k) acc += in[rr][k] * w[k];
    acc += bias;
    c.ws.h[(size_t)row * 256 + col] = acc;
    c.ws.atom_hids[(size_t)row * 768 + col] = acc;
    if (col < 4) c.ws.pos[(size_t)row * 4 + col] = col < 3 ? in[rr][12 + col] : 0.0f;
  }
}

// Pair init: edge_attr0 = edge_emb([edge_x(2), cond_edge_x(2), dist(64)]) (68 -> 64); edge_hids[:, 0:64].
__global__ __launch_bounds__(256) void k_pair_init(Ctx c, const float* __restrict__ edge_x, const float* __restrict__ cond_x,
                                                   const float* __restrict__ cond_edge_x) {
  ds_fp16_saturate();
  constexpr int T = 64;
  __shared__ __attribute__((aligned(16))) float X[T][72 + DS_LDP];
  __shared__ __attribute__((aligned(16))) float xs[T];
  const int tid = threadIdx.x, row0 = blockIdx.x * T;
  const bool use_rbf = (cond_x != nullptr) && (c.ws.flags[0] != 0);   // dmt.py:364-368
  if (tid < T) {
    const int p = row0 + tid;
    float x = 0.0f;
    float e0 = 0, e1 = 0, c0 = 0, c1 = 0;
    if (p < c.L.Pp) {
      const int da = c.L.node_dense[c.L.pair_a[p]], db = c.L.node_dense[c.L.pair_b[p]];
      const int m = c.L.pair_mol[p];
      const size_t eidx = ((size_t)da * c.L.N + (db - m * c.L.N)) * 2;
      e0 = edge_x[eidx]; e1 = edge_x[eidx + 1];
      if (cond_x != nullptr) {
        c0 = cond_edge_x[eidx]; c1 = cond_edge_x[eidx + 1];
        const float* pa = cond_x + (size_t)da * 9;
        const float* pb = cond_x + (size_t)db * 9;
        const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
        const float d2 = dx * dx + dy * dy + dz * dz;
        const float* ad = c.ws.ada + (size_t)m * ADAC + DS_ADA_TOP;
        x = use_rbf ? d2 * (ad[0] + 1.0f) + ad[1] : d2;
      }
    }
    X[tid][0] = e0; X[tid][1] = e1; X[tid][2] = c0; X[tid][3] = c1;
    X[tid][68] = 0.0f; X[tid][69] = 0.0f; X[tid][70] = 0.0f; X[tid][71] = 0.0f;   // K padded to the 8-wide k-group
    xs[tid] = x;
  }
  __syncthreads();
  const float* mean = GW(c, DS_GW_RBF_MEAN);
  const float* stdv = GW(c, DS_GW_RBF_STD);
  const float* astd = GW(c, DS_GW_RBF_ASTD);
  for (int idx = tid; idx < T * 64; idx += 256) {
    const int row = idx >> 6, k = idx & 63;
    float v = 0.0f;
    if (row0 + row < c.L.Pp) v = use_rbf ? rbf_feature(xs[row], k, mean, stdv, astd) : xs[row];  // zeros repeat (dmt.py:365)
    X[row][4 + k] = v;
  }
  __syncthreads();
  const float* bias = GW(c, DS_GW_EDGE_EMB_B);
  const int Pp = c.L.Pp;
  float* e = c.ws.e;
  float* hid = c.ws.edge_hids;
  tile_gemm<2, 1>(&X[0][0], 72 + DS_LDP, 72, GW(c, DS_GW_EDGE_EMB_W), 64, 2, [&](int row, int col, float v) {
    const int p = row0 + row;
    if (p < Pp) {
      const float y = v + bias[col];
      e[(size_t)p * 64 + col] = y;
      hid[(size_t)p * 192 + col] = y;
    }
  });
}

// ------------------------------------------------------------------------------------------------
// Block stage A (pairs): d^2 -> CondGaussian RBF -> edge_emb(128->64) -> LN -> modulate -> the 256-byte split-fp16 row `ye`
// (k_attn_fused evaluates tanh(lin_edge0/1 ye) on chip) and the modulated distance x' (k_edge_update recomputes the RBF from it).
// dmt.py:136-139,145-149; layers.py:328-334.
__global__ __launch_bounds__(256) void k_edge_geom(Ctx c, int blk) {
  ds_fp16_saturate();
  constexpr int T = 64, LDH = 2 * 128 + 8;
  // [x', rbf63 | e64] in the split-fp16 layout (ds_device.h): edge_emb runs on the f16 matrix pipe (the fp32 form was 256
  // 64-cycle MFMAs per workgroup, ~40 % of this kernel's time; now 96 32-cycle ones)
  __shared__ __attribute__((aligned(16))) _Float16 Xh[T][LDH];
  __shared__ __attribute__((aligned(16))) float Y[T][64 + DS_LDP];
  __shared__ __attribute__((aligned(16))) float xs[T];
  __shared__ int rmol[T];
  const int tid = threadIdx.x, row0 = blockIdx.x * T;
  // Thread (k = feature, rb = row phase) of the RBF stage further down: its RBF centre/width and its 16 previous-block
  // edge features depend on nothing computed here, so they are requested first and ride along with the index chain.
  const int kf = tid & 63, rbp = tid >> 6;
  const float mk = kf ? BW(c, blk, DS_BW_RBF_MEAN)[kf - 1] : 0.0f;
  const float sk = kf ? BW(c, blk, DS_BW_RBF_STD)[kf - 1] : 1.0f;
  const float ak = kf ? BW(c, blk, DS_BW_RBF_ASTD)[kf - 1] : 1.0f;
  float ev[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) ev[j] = c.ws.e[(size_t)min(row0 + rbp + 4 * j, c.L.Pp - 1) * 64 + kf];
  if (tid < T) {
    const int p = row0 + tid;
    float x = 0.0f;
    int m = 0;
    if (p < c.L.Pp) {
      m = c.L.pair_mol[p];
      const float* pa = c.ws.pos + (size_t)c.L.pair_a[p] * 4;
      const float* pb = c.ws.pos + (size_t)c.L.pair_b[p] * 4;
      const float dx = pa[0] - pb[0], dy = pa[1] - pb[1], dz = pa[2] - pb[2];
      const float d2 = dx * dx + dy * dy + dz * dz;
      const float* ad = c.ws.ada + (size_t)m * ADAC + blk * DS_ADA_BLOCK_STRIDE + DS_ADA_DIST;
      x = d2 * (ad[0] + 1.0f) + ad[1];   // scale, shift (layers.py:330-331)
    }
    xs[tid] = x;
    rmol[tid] = m;
    if (p < c.L.Pp) c.ws.dist[p] = x;   // k_edge_update recomputes the 64 CondGaussian features from it: 4 bytes per pair instead of 256 each way
  }
  __syncthreads();
  // adaLN shift/scale of the LayerNorm further down, requested now (they only need rmol): wave w normalises rows
  // 16w .. 16w+15 as four passes of four rows, one row per 16-lane DPP row, float4 per lane
  float4 lsh[4], lsc[4];
  {
    const float* ade = c.ws.ada + blk * DS_ADA_BLOCK_STRIDE + DS_ADA_EDGE;
    const int lane = tid & 63, wv = tid >> 6;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float* am = ade + (size_t)rmol[16 * wv + 4 * j + (lane >> 4)] * ADAC;
      lsh[j] = reinterpret_cast<const float4*>(am)[lane & 15];        // edge_shift_msa
      lsc[j] = reinterpret_cast<const float4*>(am + 64)[lane & 15];   // edge_scale_msa
    }
  }
  {   // thread owns feature k of rows rb, rb+4, ...
    const int k = kf, rb = rbp;
    const float rsk = __builtin_amdgcn_rcpf(sk), rak = __builtin_amdgcn_rcpf(ak);   // once per thread, not once per element
    const int Pp = c.L.Pp;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int row = rb + 4 * j, p = row0 + row;
      const float x = xs[row];
      const float z = (x - mk) * rsk;
      float v = k ? __expf(-0.5f * (z * z)) * rak : x;             // layers.py:291-295,334 (k = 0 is the raw x')
      if (p >= Pp) { v = 0.0f; ev[j] = 0.0f; }
      split_store1(&Xh[row][0], 128, k, v);
      split_store1(&Xh[row][0], 128, 64 + k, ev[j]);
    }
  }
  {   // edge_emb (128 -> 64): wave w owns row tile w >> 1, column chunk w & 1; the whole weight chunk (8 k-blocks) is requested
      // ahead of the barrier
    const int wv = tid >> 6, mt = wv >> 1, cc = wv & 1, lane = tid & 63;
    const WStreamH wse = wstream_h(BW(c, blk, DS_BW_EDGE_EMB_H), 64, 128, cc * 32);
    WRingH<8> ring;
    wring_h<8>(ring, wse, 0);
    const float bcol = BW(c, blk, DS_BW_EDGE_EMB_B)[cc * 32 + (lane & 31)];
    __syncthreads();
    f32x16 acc[1], lo[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = bcol;
    acc_zero<1>(lo);
    wave_mma_h_deep<1, false, 8, 8>(&Xh[mt * 32][0], 128, wse, ring, 0, acc, lo);
    split_finish<1>(acc, lo);
#pragma unroll
    for (int i = 0; i < 16; ++i) Y[mt * 32 + acc_row(i, lane >> 5)][cc * 32 + (lane & 31)] = acc[0][i];
  }
  __syncthreads();
  // norm1_edge + modulate (dmt.py:149) -> ws.ye in the split-fp16 layout (two 64-half planes per pair row).  The two 64 -> 256
  // projections lin_edge0 / lin_edge1 and their tanh are NOT evaluated here any more: k_attn_fused recomputes them per molecule
  // on the f16 matrix pipe from these 256-byte rows, so the 2 x 1 kB per pair of round 1's te0 / te1 never travel through HBM.
  {
    const int lane = tid & 63, wv = tid >> 6;
    _Float16* ye = reinterpret_cast<_Float16*>(c.ws.ye);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = 16 * wv + 4 * j + (lane >> 4);
      const float4 v = ln_mod_reg64(reinterpret_cast<const float4*>(&Y[row][0])[lane & 15], lsh[j], lsc[j]);
      if (row0 + row < c.L.Pp) split_store4(ye + (size_t)(row0 + row) * 128, 64, 4 * (lane & 15), v);
    }
  }
}

// Block stage B (nodes): LN -> modulate -> q|k|v projection (256 -> 768).  dmt.py:148; layers.py:147-149.
// 64-row tiles, two row tiles per weight fragment (with one, the q|k|v weight stream alone asks the L2 for ~75 GB/s per
// CU at full MFMA rate, the measured per-CU L2 ceiling); blockIdx.y picks one 384-column half of the projection so
// that the launch keeps >= 4 workgroups per CU at bench sizes - the LayerNorm of the tile is simply done by both halves.
template <int NW>
__global__ __launch_bounds__(NW * 64) void k_node_qkv(Ctx c, int blk) {
  ds_fp16_saturate();
  constexpr int NT = NW * 64;
  constexpr int T = 64;
  constexpr int LDH = 2 * 256 + 8;
  __shared__ __attribute__((aligned(16))) _Float16 Xh[T][LDH];    // LayerNorm output in the split-fp16 layout (ds_device.h)
  __shared__ int rmol[T];
  const int tid = threadIdx.x, row0 = blockIdx.x * T, col0 = blockIdx.y * 384;
  if (tid < T) rmol[tid] = (row0 + tid < c.L.Nn) ? c.L.node_mol[row0 + tid] : 0;
  __syncthreads();
  {   // 64 rows: wave w normalises rows 16w .. 16w+15, four rows per pass (one per 16-lane DPP row), two passes in flight
    const float* adn = c.ws.ada + blk * DS_ADA_BLOCK_STRIDE + DS_ADA_NODE;
    const int lane = tid & 63, wv = tid >> 6, g = lane >> 4, j = lane & 15;
    static_assert(NT == 256, "four waves x 16 rows");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float4 v[2][4], sh[2][4], sc[2][4];
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int row = 16 * wv + 8 * half + 4 * ps + g;
        const float4* hp = reinterpret_cast<const float4*>(c.ws.h + (size_t)min(row0 + row, c.L.Nn - 1) * 256) + j;
        const float4* ap = reinterpret_cast<const float4*>(adn + (size_t)rmol[row] * ADAC) + j;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          v[ps][u] = hp[16 * u];
          sh[ps][u] = ap[16 * u];        // node_shift_msa
          sc[ps][u] = ap[64 + 16 * u];   // node_scale_msa
        }
      }
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int row = 16 * wv + 8 * half + 4 * ps + g;
        ln_mod_quad256(v[ps], sh[ps], sc[ps]);   // dmt.py:148
#pragma unroll
        for (int u = 0; u < 4; ++u)
          split_store4(&Xh[row][0], 256, 64 * u + 4 * j, row0 + row < c.L.Nn ? v[ps][u] : make_float4(0, 0, 0, 0));
      }
    }
  }
  __syncthreads();
  const float* bias = BW(c, blk, DS_BW_QKV_B) + col0;
  float* qkv = c.ws.qkv + (size_t)row0 * 768 + col0;
  const int valid = c.L.Nn - row0, wave = tid >> 6;
  for (int ch = wave; ch < 12; ch += NW) {   // 32-column chunks of this half of q|k|v; both row tiles per weight fragment
    asm volatile("" ::: "memory");
    const float b = bias[ch * 32 + (tid & 31)];
    f32x16 acc[2], lo[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[m][i] = b;
    acc_zero<2>(lo);
    wave_mma_h_ring<2, false, 16, DS_QKV_PF>(&Xh[0][0], 256, BW(c, blk, DS_BW_QKV_H), 768, 256, col0 + ch * 32, 0, acc, lo);
    split_finish<2>(acc, lo);
    acc_store<2, 768>(acc, qkv + ch * 32, valid, [](int, float v) { return v; });
  }
}

// Block stage C (one 1024-thread workgroup per molecule): the whole edge-modulated attention of TransMixLayer
// (layers.py:131-186 + PyG propagate / softmax) with tanh(lin_edge0 e) and tanh(lin_edge1 e) recomputed on chip.
// The sixteen waves have two roles (each SIMD holds two of either): waves 8-15 PROJECT - per chunk of 32 pair rows the ye rows
// go to LDS and tanh(ye . lin_edge) comes off the f16 matrix pipe into one of two LDS tiles (transcendental / VALU-bound) -,
// waves 0-7 CONSUME the tile of the chunk before (LDS-bound), so a chunk costs the longer of the two, one barrier per chunk:
//   phase 1: logits of both directions of every pair: lg[p][0][h] source a -> target b, lg[p][1][h] source b -> target a;
//            14 learned heads (q_t . k_s . te0 over 18 channels, / sqrt(16)) + the 2 adjacency heads (0 -> -1e10;
//            layers.py:165-174)
//   phase 2a (all waves): segment softmax over the sources of every target (PyG softmax: max-shift, exp, / (sum + 1e-16)), one
//            half wave per target, written back over the logits
//   phase 2b: out[t] += (v_s * te1) * alpha (layers.py:178-186).  The pair rows are taken in DIFFERENCE-CLASS order here: row
//            R = (d - 1) n + i is the pair {i, (i + d) mod n}, d = 1 .. n/2 (the last class of an even n has n/2 rows).  A class
//            touches every atom exactly twice, so any run of consecutive rows holds the same number of incoming edges (+-2) for
//            every target - in the (a, b)-sorted order of the layout the first rows hold ALL sources of atoms 0 .. 3 and few of the
//            others'.  One quarter wave per target (lane = 16 channels) walks the target's visit list; sum order per target:
//            ascending R (fixed; the reference's scatter-add is unordered).
// q|k of the molecule's atoms are staged in LDS for phase 1, V takes their place for phase 2.
__global__ __launch_bounds__(1024) void k_attn_fused(Ctx c, int blk) {
  ds_fp16_saturate();
  DS_STAMP_INIT();   // diagnostic build: wave 0 (consumer) P0-P7, wave 8 (producer) P8-P15
  constexpr int NT = 1024, QS = 512 + 32, LDT = 256 + 4, LDY = 2 * 64 + 8, CR = 32, MAXP = DS_MAX_ATOMS * (DS_MAX_ATOMS - 1) / 2;
  __shared__ __attribute__((aligned(16))) float QK[DS_MAX_ATOMS * QS];       // 63,104 B; phase 2: V [n][256]
  __shared__ __attribute__((aligned(16))) float Tt[2][CR][LDT];              // 66,560 B  tanh(te0 / te1) of two chunks
  __shared__ __attribute__((aligned(16))) _Float16 Yc[2][CR][LDY];           // 17,408 B  ye rows of two chunks (split-fp16 layout)
  __shared__ __attribute__((aligned(16))) float AL[2][CR][32];               //  8,192 B  alpha of two chunks' pairs, both directions
  // phase 1: every pair of the molecule in layout order: (a << 8) | b (local atom indices) | adjacency bits << 16;
  // phase 2: the pair row of class-ordered row R
  __shared__ int PT[(MAXP + 15) & ~15];
  // phase 2b's visit lists: VT[t][q], q = 0 .. n - 2, is target t's q-th incoming edge in ascending class-ordered row R:
  // R | source << 9 | direction << 14; 511 (a row no chunk reaches) behind the end
  __shared__ int VT[32][32];
  static_assert(DS_MAX_ATOMS <= 32 && MAXP <= 511 - CR, "visit table packing");
  static_assert(DS_MAX_ATOMS * 128 <= 4 * NT && DS_MAX_ATOMS * 64 <= 2 * NT && MAXP <= NT, "single-pass staging");
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hh = lane >> 5;
  int n0, n, p0, P;
  if (c.L.mol_by_size) {                  // largest molecules first, one record per workgroup
    const int4 r = reinterpret_cast<const int4*>(c.L.mol_by_size)[blockIdx.x];
    n0 = r.x; n = r.y; p0 = r.z; P = r.w;
  } else {
    const int m = blockIdx.x;
    n0 = c.L.node_off[m]; n = c.L.node_off[m + 1] - n0;
    p0 = c.L.pair_off[m]; P = c.L.pair_off[m + 1] - p0;
  }
  // the weight streams' descriptors are set up here, next to the offsets: each needs a table entry from memory, and behind the
  // prologue's requests that small load would return only after them (loads return in order)
  const WStreamH ws_e0 = wstream_h(BW(c, blk, DS_BW_E0_H), 256, 64, (wave & 7) * 32);
  const WStreamH ws_e1 = wstream_h(BW(c, blk, DS_BW_E1_H), 256, 64, (wave & 7) * 32);
  if (n <= 0) return;
  if (P <= 0) {   // single atom: no message reaches it
    for (int i = tid; i < 256; i += NT) c.ws.attn[(size_t)n0 * 256 + i] = 0.0f;
    return;
  }
  const int nch = (P + CR - 1) / CR;
  const bool producer = wave >= 8;                 // wave-uniform
  const int ptid = tid & 511, prow = ptid >> 4, ppiece = ptid & 15;   // a producer thread's 16-byte piece of a chunk's ye rows
  // (one 128-bit register tuple per piece: as a struct of four scalars a set in flight is split up by the register allocator, and
  // the moves that put it together again wait for the load)
  const u32x4* ye4 = reinterpret_cast<const u32x4*>(c.ws.ye) + (size_t)p0 * 16;   // 16 x 16 bytes per pair row
  auto row_pair = [&](int R) {   // pair row (relative to p0) of class-ordered row R < P
    const int d = R / n + 1, i = R - (d - 1) * n;
    int j = i + d;
    if (j >= n) j -= n;
    const int a = min(i, j), b = max(i, j);
    return a * (2 * n - a - 1) / 2 + (b - a - 1);
  };
  // A chunk's ye rows are requested FOUR chunks ahead into one of two register sets and land in LDS two chunks ahead: with one
  // workgroup per CU nothing else hides their latency.
  auto fetch_y = [&](int ck, bool class_order) {
    const int R = ck * CR + prow;
    // no branch around the load and no select behind it (a load inside a conditional block is not requested ahead of the code in
    // front of the block; a select waits for it on the spot): rows behind the end repeat the last row - nothing reads their tile rows
    const int Rc = min(R, P - 1);
    return ye4[(size_t)(class_order ? PT[Rc] : Rc) * 16 + ppiece];   // phase 2: PT holds row_pair()
  };
  auto commit_y = [&](int buf, const u32x4& v) { *reinterpret_cast<u32x4*>(&Yc[buf][prow][ppiece * 8]) = v; };
  // producer wave w owns output columns 32 (w & 7) .. +31; its lin_edge0 / lin_edge1 fragments (64 -> 256, split-fp16 planes) live in
  // registers for a whole phase: 8 x 16 bytes
  h8 wf[2][4];
  auto load_weights = [&](const WStreamH& ws_) {
#pragma unroll
    for (int pl = 0; pl < 2; ++pl)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) wf[pl][kb] = wload_h(ws_, pl, kb);
  };
  auto project = [&](int buf) {   // Tt[buf] = tanh(Yc[buf] . W): 32 rows x this wave's 32 columns
    f32x16 acc[1], lo[1];
    acc_zero<1>(acc);
    acc_zero<1>(lo);
    const _Float16* xr = &Yc[buf][lane & 31][8 * hh];
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const h8 x1 = *reinterpret_cast<const h8*>(xr + kb * 16);
      const h8 x2 = *reinterpret_cast<const h8*>(xr + 64 + kb * 16);
      // transposed product (weights as the A operand): lane = tile row, registers = 4 consecutive features per quad, so the
      // tanh'ed tile goes to LDS as four 16-byte stores per lane instead of sixteen 4-byte ones
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][kb], x1, acc[0], 0, 0, 0);
      lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][kb], x2, lo[0], 0, 0, 0);
      lo[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[1][kb], x1, lo[0], 0, 0, 0);
    }
    split_finish<1>(acc, lo);
    float* trow = &Tt[buf][lane & 31][(wave & 7) * 32 + 4 * hh];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x2 v0, v1;
      v0.x = acc[0][4 * q]; v0.y = acc[0][4 * q + 1]; v1.x = acc[0][4 * q + 2]; v1.y = acc[0][4 * q + 3];
      v0 = ds_tanh2_prescaled(v0); v1 = ds_tanh2_prescaled(v1);   // layers.py:165-166,183; the packed weights carry 2 log2(e)
      *reinterpret_cast<float4*>(trow + 8 * q) = make_float4(v0.x, v0.y, v1.x, v1.y);
    }
  };
  // The producers' side of a phase: chunk 0 is projected while the consumers wait, then one chunk ahead of them.  Entered with chunk 0's
  // rows in Yc[0] (visible), chunk 1's in ya, chunk 2's in yb.  Barriers: 1 + nch, as in the consumers' loops.
  auto produce = [&](u32x4 ya, u32x4 yb, bool class_order, auto&& after_first) {
    project(0);
    after_first();
    if (1 < nch) commit_y(1, ya);
    if (3 < nch) ya = fetch_y(3, class_order);
    __syncthreads();
    if (class_order) { DS_STAMP(13); } else { DS_STAMP(9); }
    // interval k: chunk k + 2's rows (register set B) go to Yc[k & 1] - last read by chunk k's projection, an interval ago -, chunk
    // k + 4 is requested into the same registers, chunk k + 1 is projected.  The two register sets swap ROLES from one interval
    // to the next (the loop is unrolled by two): a register move of a set would wait for the load just issued into it.
    auto interval = [&](int k, u32x4& B) {
      if (k + 2 < nch) commit_y(k & 1, B);
      if (k + 4 < nch) B = fetch_y(k + 4, class_order);
      if (k + 1 < nch) project((k + 1) & 1);
      if (class_order) { DS_STAMP(14); } else { DS_STAMP(10); }
      __syncthreads();
      if (class_order) { DS_STAMP(15); } else { DS_STAMP(11); }
    };
    for (int k = 0; k < nch; k += 2) {
      interval(k, yb);
      if (k + 1 >= nch) break;
      interval(k + 1, ya);
    }
  };

  // ---- phase 0: q (256) | k (256) of every atom -> LDS (n * 128 <= 4 * NT 16-byte pieces).  Every request of the prologue goes out
  // here, the weights (L2) in front - loads return in order -, q|k last: the producers' first projection needs only the first
  // chunk and the weights and runs while q|k are on their way.  The stores behind these loads are UNCONDITIONAL (pieces behind the
  // end go to a junk area: the idle second tile): a load whose only use sits in a conditional block is sunk into that block by
  // the compiler, which turns "request everything, wait once" into one round of memory latency per piece.
  {
    u32x4 y0 = {0, 0, 0, 0}, y1 = y0, y2 = y0;
    if (producer) {
      load_weights(ws_e0);
      y0 = fetch_y(0, false);
      y1 = fetch_y(1, false);
      y2 = fetch_y(2, false);
    } else if (tid < P) {
      PT[tid] = ((c.L.pair_a[p0 + tid] - n0) << 8) | (c.L.pair_b[p0 + tid] - n0) | (c.ws.adj[p0 + tid] << 16);
    }
    f32x4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = min(tid + u * NT, n * 128 - 1);
      v[u] = reinterpret_cast<const f32x4*>(c.ws.qkv + (size_t)(n0 + (idx >> 7)) * 768)[idx & 127];
    }
    __builtin_amdgcn_sched_barrier(0);
    auto store_qk = [&]() {
      f32x4* junk = reinterpret_cast<f32x4*>(&Tt[1][0][0]) + tid;     // 16 kB of the second tile, first written in interval 0
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = tid + u * NT;
        f32x4* dst = idx < n * 128 ? reinterpret_cast<f32x4*>(QK) + (idx >> 7) * (QS / 4) + (idx & 127) : junk;
        *dst = v[u];
      }
    };
    if (producer) commit_y(0, y0);
    __syncthreads();                       // PT, Yc[0]
    if (producer) { DS_STAMP(8); } else { DS_STAMP(0); }
    // ---- phase 1: logits
    if (producer) {
      produce(y1, y2, false, store_qk);   // q|k land in LDS behind the first projection, which does not read them
      load_weights(ws_e1);                 // lin_edge1 fragments fly during the softmax
    } else {
      store_qk();
      __syncthreads();                     // Tt[0], QK
      DS_STAMP(1);
      const int row = tid >> 4, hs = tid & 15;
      for (int k = 0; k < nch; ++k) {
        const int pl = k * CR + row;
        if (pl < P) {
          const int e_ = PT[pl];
          float* out = c.ws.lg + (size_t)(p0 + pl) * 32;
          if (hs == 14) {
            const float h0 = (e_ & 0x10000) ? 1.0f : -1e10f, h1 = (e_ & 0x20000) ? 1.0f : -1e10f;   // layers.py:171-174
            out[0] = h0; out[1] = h1; out[16] = h0; out[17] = h1;
          } else if (hs < 14) {
            const int a = (e_ >> 8) & 255, b = e_ & 255;
            // volatile: kept as 45 ds_read_b64 (2 LDS cycles each).  Merged in pairs into ds_read2_b64 - what the compiler does with
            // adjacent 8-byte loads - they take 8 cycles per pair.
            typedef const volatile f32x2 __attribute__((address_space(3))) * lds_v2;
            lds_v2 t0 = (lds_v2)(&Tt[k & 1][row][hs * 18]);
            lds_v2 qa = (lds_v2)(QK + a * QS + hs * 18);
            lds_v2 qb = (lds_v2)(QK + b * QS + hs * 18);
            lds_v2 ka = (lds_v2)(QK + a * QS + 256 + hs * 18);
            lds_v2 kb = (lds_v2)(QK + b * QS + 256 + hs * 18);
            float s_ab = 0.0f, s_ba = 0.0f;
#pragma unroll
            for (int j0 = 0; j0 < 9; j0 += 3) {   // three batches of 15 reads
              f32x2 e[3], xa[3], xb[3], ya[3], yb[3];
#pragma unroll
              for (int j = 0; j < 3; ++j) { e[j] = t0[j0 + j]; xa[j] = qa[j0 + j]; xb[j] = qb[j0 + j]; ya[j] = ka[j0 + j]; yb[j] = kb[j0 + j]; }
#pragma unroll
              for (int j = 0; j < 3; ++j) {
#if DS_ATTN_FMA   // one multiply + one fused multiply-add per term (4-cycle issues) instead of the packed multiply + add the compiler forms (8 + 4)
                s_ab = __builtin_fmaf(xb[j].x * ya[j].x, e[j].x, s_ab); s_ab = __builtin_fmaf(xb[j].y * ya[j].y, e[j].y, s_ab);
                s_ba = __builtin_fmaf(xa[j].x * yb[j].x, e[j].x, s_ba); s_ba = __builtin_fmaf(xa[j].y * yb[j].y, e[j].y, s_ba);
#else
                s_ab += (xb[j].x * ya[j].x) * e[j].x; s_ab += (xb[j].y * ya[j].y) * e[j].y;
                s_ba += (xa[j].x * yb[j].x) * e[j].x; s_ba += (xa[j].y * yb[j].y) * e[j].y;
#endif
              }
            }
            out[2 + hs] = s_ab / 4.0f;        // / sqrt(out_channels = 16) (layers.py:167)
            out[16 + 2 + hs] = s_ba / 4.0f;
          }
        }
        DS_STAMP(2);
        __syncthreads();
        DS_STAMP(3);
      }
    }
  }
  __threadfence_block();
  __syncthreads();                         // all logits written (and visible to this workgroup); QK is dead from here
  // ---- phase 2a: V -> LDS (over QK), the visit table, softmax per target written back over the logits
  float* V = QK;
  int* ET = reinterpret_cast<int*>(&Tt[0][0][0]);
  {   // class d reaches target t from source t + d (row i = t) and from source t - d (row i = t - d), both mod n; the last class of an
      // even n holds each atom once
    const int t = tid >> 5, q = tid & 31, d = (q >> 1) + 1;
    int e = 511;
    if (t < n && q < n - 1) {
      int i, s_;
      if (2 * d == n) {
        i = t < d ? t : t - d;
        s_ = t < d ? t + d : t - d;
      } else {
        int sa = t + d, ib = t - d;
        if (sa >= n) sa -= n;
        if (ib < 0) ib += n;
        const bool first_is_a = t < ib;            // rows of the class in ascending order
        const bool take_a = (q & 1) ? !first_is_a : first_is_a;
        i = take_a ? t : ib;
        s_ = take_a ? sa : ib;
      }
      e = ((d - 1) * n + i) | (s_ << 9) | ((s_ < t ? 0 : 1) << 14);   // direction 0 is source a -> target b with a < b
    }
    VT[t][q] = e;
    // ET[t][s]: where the logit / alpha of edge s -> t lives, as a float offset from the molecule's first logit (-1: no such edge).
    // The softmax below reads 2 x 15 of them per lane; computed in place, the index arithmetic (~25 VALU issues each, sixteen
    // lanes deriving the same number) was most of that phase.  The table borrows Tt, idle until phase 2b's first projection.
    const int s = q;
    int eo = -1;
    if (t < n && s < n && s != t) {
      const int lo_ = s < t ? s : t, hi_ = s < t ? t : s;
      eo = (lo_ * (2 * n - lo_ - 1) / 2 + (hi_ - lo_ - 1)) * 32 + (s < t ? 0 : 16);
    }
    ET[t * 32 + s] = eo;
    if (tid < P) PT[tid] = row_pair(tid);  // phase 1's pair table is dead
  }
  f32x4 vst[2];   // V rows: requested here, stored to LDS behind the softmax (unconditional stores, as in phase 0; junk area: Yc)
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i0 = min(tid + u * NT, n * 64 - 1);
    vst[u] = reinterpret_cast<const f32x4*>(c.ws.qkv + (size_t)(n0 + (i0 >> 6)) * 768 + 512)[i0 & 63];
  }
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();                         // ET, PT, VT
  u32x4 y0 = {0, 0, 0, 0}, y1 = y0, y2 = y0;
  if (producer) {                          // the first class-ordered chunks do not wait for the softmax
    y0 = fetch_y(0, true);
    y1 = fetch_y(1, true);
    y2 = fetch_y(2, true);
  }
  auto store_v = [&]() {
    f32x4* junk = reinterpret_cast<f32x4*>(&Yc[0][0][0]) + tid;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      f32x4* dst = tid + u * NT < n * 64 ? reinterpret_cast<f32x4*>(V) + tid + u * NT : junk;
      *dst = vst[u];
    }
  };
  {   // one half wave per target (all targets at once: a molecule's softmax is one round of L2 latency), lane = (source parity, head)
    const int h = lane & 15, sq = (lane >> 4) & 1, t = 2 * wave + hh;
    const int* et = ET + t * 32 + sq;
    float* lgm = c.ws.lg + (size_t)p0 * 32 + h;
    if (t < n) {
      float x[15];
      int eos[15];
      float mx = -INFINITY;
      // all fifteen logits are requested before any is used: unconditional loads from clamped offsets, the "no such edge" select
      // afterwards (a load inside a conditional block is waited for at the end of its block: fifteen L2 round trips in a row)
#pragma unroll
      for (int j = 0; j < 15; ++j) eos[j] = et[2 * j];
#pragma unroll
      for (int j = 0; j < 15; ++j) x[j] = lgm[max(eos[j], 0)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 15; ++j) x[j] = eos[j] >= 0 ? x[j] : -INFINITY;
#pragma unroll
      for (int j = 0; j < 15; ++j) mx = fmaxf(mx, x[j]);
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
      float sum = 0.0f;
#pragma unroll
      for (int j = 0; j < 15; ++j) {
        x[j] = __expf(x[j] - mx);           // hardware exp2 form (1e-7 relative, as the other transcendentals here); a missing edge: exp(-inf) = 0
        sum += x[j];
      }
      sum += __shfl_xor(sum, 16, 64);
      const float den = sum + 1e-16f;
#pragma unroll
      for (int j = 0; j < 15; ++j)
        if (eos[j] >= 0) lgm[eos[j]] = x[j] / den;
    }
  }
  store_v();                               // V is first read behind phase 2b's opening barriers: its latency runs under the softmax
  __threadfence_block();
  __syncthreads();                         // alphas written, V and VT visible
  if (producer) { DS_STAMP(12); } else { DS_STAMP(4); }
  // ---- phase 2b: aggregation
  if (producer) {
    commit_y(0, y0);
    __syncthreads();                       // Yc[0] (and the consumers' AL[0])
    produce(y1, y2, true, [] {});
    DS_STAMP_FLUSH(512);
  } else {
    // a chunk's alpha rows (32 x 128 bytes: consumer threads 0 .. 255) travel like the producers' ye rows: requested three chunks
    // ahead into one of two register sets, in LDS one chunk ahead
    const int arow = (tid & 255) >> 3, apiece = tid & 7;   // threads 256 .. 511 request the same rows again (no branch around the load)
    auto fetch_a = [&](int ck) {
      const int R = ck * CR + arow;
      return reinterpret_cast<const u32x4*>(c.ws.lg + (size_t)(p0 + PT[min(R, P - 1)]) * 32)[apiece];   // as fetch_y
    };
    auto commit_a = [&](int buf, const u32x4& v) {
      if (tid < 256) reinterpret_cast<u32x4*>(&AL[buf][arow][0])[apiece] = v;
    };
    const u32x4 a0 = fetch_a(0);
    u32x4 ab = fetch_a(1), aa = fetch_a(2);
    commit_a(0, a0);
    __syncthreads();                       // AL[0] (and the producers' Yc[0])
    __syncthreads();                       // Tt[0]
    DS_STAMP(5);
    const int t_me = tid >> 4, q16 = tid & 15;   // one quarter wave per target; lane = channels 4 q16 + 64 u .. + 3, u = 0 .. 3
    const int* vt_row = &VT[t_me][0];
    int vt_q = 0, vt_e = vt_row[0];
    float4 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc[u] = make_float4(0, 0, 0, 0);
    // interval k: chunk k + 1's alpha rows (register set A) go to AL[(k + 1) & 1] - last read in interval k - 1 -, chunk k + 3 is
    // requested into the same registers (the two sets swap roles from one interval to the next, as the producers'), then this
    // quarter wave's visits of chunk k: the entries of its list with R0 <= R < R0 + 32 (the list is sorted and the chunks ascend,
    // so a pointer walks it once per molecule)
    auto interval = [&](int k, u32x4& A) {
      if (k + 1 < nch) commit_a((k + 1) & 1, A);
      if (k + 3 < nch) A = fetch_a(k + 3);
      const int R0 = k * CR;
      while ((vt_e & 511) - R0 < CR) {
        const int r = (vt_e & 511) - R0, s_ = (vt_e >> 9) & 31;
        // a quarter wave reads 256 contiguous bytes per instruction (ds_read_b128 is conflict-free only for lane-contiguous pieces)
        const float4* gp = reinterpret_cast<const float4*>(&Tt[k & 1][r][4 * q16]);
        const float4* vp = reinterpret_cast<const float4*>(V + s_ * 256 + 4 * q16);
        const float* ap = &AL[k & 1][r][((vt_e >> 14) & 1) * 16 + (q16 >> 2)];
        float4 g[4], v[4];
        float al[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { g[u] = gp[16 * u]; v[u] = vp[16 * u]; al[u] = ap[4 * u]; }
        vt_e = vt_row[++vt_q];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          acc[u].x += (v[u].x * g[u].x) * al[u]; acc[u].y += (v[u].y * g[u].y) * al[u];
          acc[u].z += (v[u].z * g[u].z) * al[u]; acc[u].w += (v[u].w * g[u].w) * al[u];
        }
      }
      DS_STAMP(6);
      __syncthreads();
      DS_STAMP(7);
    };
    for (int k = 0; k < nch; k += 2) {     // entered with chunk 1 in ab, chunk 2 in aa
      interval(k, ab);
      if (k + 1 >= nch) break;
      interval(k + 1, aa);
    }
    if (t_me < n) {
      float4* o = reinterpret_cast<float4*>(c.ws.attn + (size_t)(n0 + t_me) * 256 + 4 * q16);
#pragma unroll
      for (int u = 0; u < 4; ++u) o[16 * u] = acc[u];
    }
    DS_STAMP_FLUSH(0);
  }
}

// node2edge_lin applied per node (256 -> 64, dmt.py:156-157) as a kernel of its own: 64 attention rows per workgroup, one 32-row
// tile x one 32-column chunk per wave.  Same split-fp16 product, fragment order and k order as the N2E = true form of k_node_update
// (a row's result does not depend on its tile), so ws.u holds the same bits either way.
__global__ __launch_bounds__(256) void k_node_n2e(Ctx c, int blk) {
  ds_fp16_saturate();
  constexpr int T = 64, LDH = 2 * 256 + 8;
  __shared__ __attribute__((aligned(16))) _Float16 A1[T][LDH];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, row0 = blockIdx.x * T;
  const int Nn = c.L.Nn;
  {
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {   // 64 rows x 64 float4
      const int idx = tid + u * 256, row = idx >> 6, k4 = idx & 63;
      v[u] = reinterpret_cast<const float4*>(c.ws.attn + (size_t)min(row0 + row, Nn - 1) * 256)[k4];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const int idx = tid + u * 256, row = idx >> 6, k4 = idx & 63;
      split_store4(&A1[row][0], 256, 4 * k4, row0 + row < Nn ? v[u] : make_float4(0, 0, 0, 0));
    }
  }
  __syncthreads();
  const int mt = wave >> 1, cc = wave & 1;
  if (row0 + mt * 32 >= Nn) return;
  f32x16 acc[1], lo[1];
  acc_zero<1>(acc);
  acc_zero<1>(lo);
  wave_mma_h_ring<1, false, 16, DS_NODE_PF>(&A1[mt * 32][0], 256, BW(c, blk, DS_BW_N2E_H), 64, 256, cc * 32, 0, acc, lo);
  split_finish<1>(acc, lo);
  acc_store<1, 64>(acc, c.ws.u + (size_t)(row0 + mt * 32) * 64 + cc * 32, Nn - row0 - mt * 32, [](int, float v) { return v; });
  (void)lane;
}

// Block stage D (nodes, 32 rows): node2edge partial, gated residual, LN, modulate, FF(256->512->256) in two
// hidden halves (the 256-wide half aliases the attention tile, FF2 accumulators persist in registers), gated residual
// in place, per-block readout slice (256->64) and the node parts of equi_update.input_lin (256->512).
// dmt.py:156-163,387,39-45.  66.5 kB LDS -> two workgroups per CU.
// N2E = false: node2edge_lin has been applied by k_node_n2e (the two-stream forward, ds_forward: the pair rows' k_edge_update then
// runs beside this kernel instead of behind it).
template <bool N2E>
__global__ __launch_bounds__(256, 2) void k_node_update(Ctx c, int blk) {
  ds_fp16_saturate();
  constexpr int T = 32, LDH = 2 * 256 + 8;
  // both tiles in the split-fp16 layout of ds_device.h (every use is an MFMA operand; the one fp32 re-read - the residual of
  // the FF - reconstructs x1 + x2/2048, 2^-23 relative from the original); same bytes as the fp32 tiles they replace
  __shared__ __attribute__((aligned(16))) _Float16 B1[T][LDH];   // attention tile, then FF hidden halves
  __shared__ __attribute__((aligned(16))) _Float16 H2[T][LDH];   // normalised residual stream, then h_out in place
  __shared__ int rmol[T];
  const int tid = threadIdx.x, wave = tid >> 6, row0 = blockIdx.x * T;
  const int Nn = c.L.Nn;
  const float* ada = c.ws.ada;
  if (tid < T) rmol[tid] = (row0 + tid < Nn) ? c.L.node_mol[row0 + tid] : 0;
  __syncthreads();
  {   // 32 rows: wave w takes rows 8w .. 8w+7 as two passes of four rows (one row per 16-lane DPP row, lane j holds the
      // float4s at columns 4j + 64u); residual, LayerNorm and modulate in registers
    const int lane = tid & 63, g = lane >> 4, j = lane & 15;
    float4 va[2][4], vh[2][4], vg[2][4], sh[2][4], sc[2][4];
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = 8 * wave + 4 * ps + g;
      const size_t gr = (size_t)min(row0 + row, Nn - 1);   // clamp instead of branching: loads stay batched
      const float4* ad = reinterpret_cast<const float4*>(ada + (size_t)rmol[row] * ADAC + blk * DS_ADA_BLOCK_STRIDE + DS_ADA_NODE) + j;
      const float4* ap = reinterpret_cast<const float4*>(c.ws.attn + gr * 256) + j;
      const float4* hp = reinterpret_cast<const float4*>(c.ws.h + gr * 256) + j;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        va[ps][u] = ap[16 * u];
        vh[ps][u] = hp[16 * u];
        vg[ps][u] = ad[128 + 16 * u];   // node_gate_msa   (+512 floats)
        sh[ps][u] = ad[192 + 16 * u];   // node_shift_mlp  (+768)
        sc[ps][u] = ad[256 + 16 * u];   // node_scale_mlp  (+1024)
      }
    }
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int row = 8 * wave + 4 * ps + g;
      float4 r[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (row0 + row >= Nn) va[ps][u] = vh[ps][u] = make_float4(0, 0, 0, 0);
        if (N2E) split_store4(&B1[row][0], 256, 64 * u + 4 * j, va[ps][u]);
        // h_in + gate_msa * attn (dmt.py:159)
        r[u].x = vh[ps][u].x + vg[ps][u].x * va[ps][u].x; r[u].y = vh[ps][u].y + vg[ps][u].y * va[ps][u].y;
        r[u].z = vh[ps][u].z + vg[ps][u].z * va[ps][u].z; r[u].w = vh[ps][u].w + vg[ps][u].w * va[ps][u].w;
      }
      ln_mod_quad256(r, sh[ps], sc[ps]);   // norm2_node + modulate (dmt.py:160)
#pragma unroll
      for (int u = 0; u < 4; ++u) split_store4(&H2[row][0], 256, 64 * u + 4 * j, r[u]);
    }
  }
  __syncthreads();
  if (N2E) {
    if (wave < 2) {   // node2edge_lin applied per node (256 -> 64): two 32-column chunks
      f32x16 acc[1], lo[1];
      acc_zero<1>(acc);
      acc_zero<1>(lo);
      wave_mma_h_ring<1, false, 16, DS_NODE_PF>(&B1[0][0], 256, BW(c, blk, DS_BW_N2E_H), 64, 256, wave * 32, 0, acc, lo);
      split_finish<1>(acc, lo);
      acc_store<1, 64>(acc, c.ws.u + (size_t)row0 * 64 + wave * 32, Nn - row0, [](int, float v) { return v; });
    }
    __syncthreads();   // H2 normalised; every wave is done reading B1 (node2edge)
  }
  {
    const float* b1 = BW(c, blk, DS_BW_FF1_B);
    const float* W1 = BW(c, blk, DS_BW_FF1_H);
    const float* W2 = BW(c, blk, DS_BW_FF2_H);
    f32x16 acc2[2][1], lo2[2][1];
    acc_zero<1>(acc2[0]); acc_zero<1>(acc2[1]);
    acc_zero<1>(lo2[0]); acc_zero<1>(lo2[1]);
    // node_gate_mlp (dmt.py:162) is per molecule: the gate columns of the tile's first / last molecule and the FF2 bias are
    // requested here, a whole FF ahead of their use (fetched after it they cost the epilogue one exposed round trip)
    const int mA = rmol[0], mB = rmol[min(T, Nn - row0) - 1];   // first / last VALID row (padded rows carry molecule 0)
    const float* gsec = ada + blk * DS_ADA_BLOCK_STRIDE + DS_ADA_NODE + 1280;
    float gAv[2], gBv[2], bbv[2];
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int col = (wave + 4 * cc) * 32 + (tid & 31);
      gAv[cc] = gsec[(size_t)mA * ADAC + col]; gBv[cc] = gsec[(size_t)mB * ADAC + col]; bbv[cc] = BW(c, blk, DS_BW_FF2_B)[col];
    }
    // Both 32-column chunks of the wave (ch = wave, wave + 4) run against the same X fragments (wave_mma_h_ring_t2): half the
    // LDS operand reads and 6 MFMAs per k-block to cover the operand latencies instead of 3.
    for (int half = 0; half < 2; ++half) {
      {
        asm volatile("" ::: "memory");
        const int hhf = (tid & 63) >> 5;
        f32x16 acc[2][1], lo[2][1];
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const float bf = b1[half * 256 + (wave + 4 * cc) * 32 + (tid & 31)];
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[cc][0][i] = bf;
          acc_zero<1>(lo[cc]);
        }
        wave_mma_h_ring_t2<1, false, 16, DS_NODE_PF2>(&H2[0][0], 256, wstream_h(W1, 512, 256, (half * 8 + wave) * 32),
                                                     wstream_h(W1, 512, 256, (half * 8 + wave + 4) * 32), 0, acc[0], lo[0], acc[1], lo[1]);
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          split_finish<1>(acc[cc], lo[cc]);
          const int colf = (wave + 4 * cc) * 32 + (tid & 31);
          // SiLU(FF1 + bias) -> hidden tile, two rows at a time (packed-fp32 epilogue)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            f32x2 v;
            v.x = acc[cc][0][i]; v.y = acc[cc][0][i + 1];
            v = ds_silu2(v);
            const int r0 = acc_row(i, hhf);
            split_store1(&B1[r0][0], 256, colf, v.x);
            split_store1(&B1[r0 + 1][0], 256, colf, v.y);
          }
        }
      }
      __syncthreads();
      asm volatile("" ::: "memory");
      wave_mma_h_ring_t2<1, false, 16, DS_NODE_PF2>(&B1[0][0], 256, wstream_h(W2, 256, 512, wave * 32), wstream_h(W2, 256, 512, (wave + 4) * 32),
                                                   half * 16, acc2[0], lo2[0], acc2[1], lo2[1], half * 16);
      __syncthreads();
    }
    split_finish<1>(acc2[0], lo2[0]);
    split_finish<1>(acc2[1], lo2[1]);
    // which of the tile's molecules a row belongs to is decided once per row (the predicates live in SGPR lane masks),
    // not once per element; rows are sorted by molecule, so a third molecule only occurs for tiny ones (uniform slow path)
    const int hhl = (tid & 63) >> 5;
    const bool two = mB - mA <= 1;
    bool isB[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) isB[i] = rmol[acc_row(i, hhl)] != mA;
    const int rows_here = Nn - row0;
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int col = (wave + 4 * cc) * 32 + (tid & 31);
      const float gA = gAv[cc], gB = gBv[cc], bb = bbv[cc];
      const RowStore st = row_store<256>(c.ws.h + (size_t)row0 * 256 + (wave + 4 * cc) * 32);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = acc_row(i, hhl);
        float g = isB[i] ? gB : gA;
        if (!two) { const int m = rmol[row]; if (m != mA && m != mB) g = gsec[(size_t)m * ADAC + col]; }
        const float out = split_load1(&H2[row][0], 256, col) + g * (acc2[cc][0][i] + bb);   // in place
        split_store1(&H2[row][0], 256, col, out);
        if (row < rows_here) row_put(st, ((i & 3) + 8 * (i >> 2)) * 256 * 4, out);
      }
    }
  }
  __syncthreads();
  {   // per-block readout slice (256 -> 64, chunks 0-1) and the node parts of equi_update.input_lin (256 -> 512, chunks 2-17):
      // 18 chunks over 4 waves, two at a time against shared X fragments (chunks it, it + 4), the last two singly
    const float* br = BW(c, blk, DS_BW_NODE_RO_B);
    auto ws_of = [&](int it) {
      return it < 2 ? wstream_h(BW(c, blk, DS_BW_NODE_RO_H), 64, 256, it * 32) : wstream_h(BW(c, blk, DS_BW_AC_H), 512, 256, (it - 2) * 32);
    };
    auto bias_of = [&](int it) { return it < 2 ? br[it * 32 + (tid & 31)] : 0.0f; };
    auto store_of = [&](int it, const f32x16 (&acc)[1]) {
      if (it < 2) acc_store<1, 768>(acc, c.ws.atom_hids + (size_t)row0 * 768 + 256 + 64 * blk + it * 32, Nn - row0, [](int, float v) { return v; });
      else acc_store<1, 512>(acc, c.ws.ac + (size_t)row0 * 512 + (it - 2) * 32, Nn - row0, [](int, float v) { return v; });
    };
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      asm volatile("" ::: "memory");
      const int itA = wave + 8 * pr, itB = itA + 4;
      f32x16 acc[2][1], lo[2][1];
      const float bA = bias_of(itA), bB = bias_of(itB);
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc[0][0][i] = bA; acc[1][0][i] = bB; }
      acc_zero<1>(lo[0]); acc_zero<1>(lo[1]);
      wave_mma_h_ring_t2<1, false, 16, DS_NODE_PF2>(&H2[0][0], 256, ws_of(itA), ws_of(itB), 0, acc[0], lo[0], acc[1], lo[1]);
      split_finish<1>(acc[0], lo[0]);
      store_of(itA, acc[0]);
      split_finish<1>(acc[1], lo[1]);
      store_of(itB, acc[1]);
    }
    if (wave < 2) {   // chunks 16, 17
      asm volatile("" ::: "memory");
      const int it = 16 + wave;
      f32x16 acc[1], lo[1];
      acc_zero<1>(acc);
      acc_zero<1>(lo);
      wave_mma_h_ring<1, false, 16, DS_NODE_PF>(&H2[0][0], 256, BW(c, blk, DS_BW_AC_H), 512, 256, (it - 2) * 32, 0, acc, lo);
      split_finish<1>(acc, lo);
      store_of(it, acc);
    }
  }
}

// Block stage E (pairs): h_edge = node2edge(h_a + h_b), gated residual, LN, modulate, FF(64->128->64), gated residual,
// readout slice (64->16), edge+dist part of equi_update.input_lin (128->256).  dmt.py:156-157,165-169,388.
// ROW-PARALLEL: every wave owns 32 pair rows from staging to the last store, with wave-private LDS tiles and NO workgroup
// barrier (the five barrier-separated GEMM phases of the tile-parallel form cost ~15k cycles of dead time each).  The FF is
// chained in registers: FF3 is computed transposed (lane = row, registers = hidden features) and its SiLU'd accumulators
// are the B operand of the FF4 MFMAs; the gated residual is applied in that transposed layout with 16-byte LDS accesses.
__global__ __launch_bounds__(256, 2) void k_edge_update(Ctx c, int blk) {
  ds_fp16_saturate();
  DS_STAMP_INIT();   // diagnostic build (tools/stamp_attn.py k_edge_update): wave 0's cycles per stage
  constexpr int R = 32, LDW = 64 + DS_LDP;
  __shared__ __attribute__((aligned(16))) float E2s[4][R][LDW];   // residual stream, then e_out in place
  __shared__ __attribute__((aligned(16))) float Ds[4][R][LDW];    // CondGaussian features of this block
  __shared__ int idx_s[4][3][R];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hh = lane >> 5, r31 = lane & 31;
  constexpr int LDW2 = 2 * 64 + 8;                       // split-fp16 row: 272 bytes, as an fp32 row of 64 + 4
  float(*E2)[LDW] = E2s[wave];                             // fp32 view of the tile (e_out, from the gated residual on)
  _Float16* E2h = reinterpret_cast<_Float16*>(&E2s[wave][0][0]);   // split-fp16 view of the same bytes
  _Float16* Dh = reinterpret_cast<_Float16*>(&Ds[wave][0][0]);
  int* rmol = idx_s[wave][0];
  int* rpa = idx_s[wave][1];
  int* rpb = idx_s[wave][2];
  const int Pp = c.L.Pp;
  const int row0 = (blockIdx.x * 4 + wave) * R;
  if (row0 >= Pp) return;                    // whole wave leaves; nothing below synchronises across waves
  const int valid = min(R, Pp - row0);
  const float* ada = c.ws.ada + blk * DS_ADA_BLOCK_STRIDE + DS_ADA_EDGE;
  if (lane < R) {
    const int p = min(row0 + lane, Pp - 1);
    rmol[lane] = c.L.pair_mol[p]; rpa[lane] = c.L.pair_a[p]; rpb[lane] = c.L.pair_b[p];
  }
  DS_STAMP(0);
  {
    const float4* bn = reinterpret_cast<const float4*>(BW(c, blk, DS_BW_N2E_B));
    // CondGaussian features of this block (layers.py:291-295,334), recomputed from the modulated squared distance k_edge_geom
    // left in ws.dist - the same expression, constants and hardware ops as there, so the same bits - instead of 256 bytes per
    // pair through HBM each way: a lane owns features 4 k4 .. 4 k4 + 3 (feature 0 is the raw x')
    float rm[4], rs[4], ra[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int k = 4 * (lane & 15) + t;
      rm[t] = k ? BW(c, blk, DS_BW_RBF_MEAN)[k - 1] : 0.0f;
      rs[t] = __builtin_amdgcn_rcpf(k ? BW(c, blk, DS_BW_RBF_STD)[k - 1] : 1.0f);
      ra[t] = __builtin_amdgcn_rcpf(k ? BW(c, blk, DS_BW_RBF_ASTD)[k - 1] : 1.0f);
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {   // 32 rows x 16 float4 (one 16-lane DPP row per tile row); 24 gathers in flight
      float4 ua[4], ub[4], ve[4], vd[4], vg[4], sh[4], sc[4];
      float xd[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = lane + (it * 4 + u) * 64, row = idx >> 4, k4 = idx & 15;
        const size_t pc = (size_t)min(row0 + row, Pp - 1);   // clamp instead of branching: loads stay batched
        const float* ad = ada + (size_t)rmol[row] * ADAC;
        ua[u] = reinterpret_cast<const float4*>(c.ws.u + (size_t)rpa[row] * 64)[k4];
        ub[u] = reinterpret_cast<const float4*>(c.ws.u + (size_t)rpb[row] * 64)[k4];
        ve[u] = reinterpret_cast<const float4*>(c.ws.e + pc * 64)[k4];
        xd[u] = c.ws.dist[pc];
        vg[u] = reinterpret_cast<const float4*>(ad + 128)[k4];   // edge_gate_msa
        sh[u] = reinterpret_cast<const float4*>(ad + 192)[k4];   // edge_shift_mlp
        sc[u] = reinterpret_cast<const float4*>(ad + 256)[k4];   // edge_scale_mlp
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int idx = lane + (it * 4 + u) * 64, row = idx >> 4, k4 = idx & 15;
        const float4 b = bn[k4];
        float4 r;   // e_in + edge_gate_msa * node2edge_lin(h_a + h_b) (dmt.py:156-157,165)
        r.x = ve[u].x + vg[u].x * ((ua[u].x + ub[u].x) + b.x); r.y = ve[u].y + vg[u].y * ((ua[u].y + ub[u].y) + b.y);
        r.z = ve[u].z + vg[u].z * ((ua[u].z + ub[u].z) + b.z); r.w = ve[u].w + vg[u].w * ((ua[u].w + ub[u].w) + b.w);
        r = ln_mod_reg64(r, sh[u], sc[u]);   // norm2_edge + modulate (dmt.py:166)
        {
          float f[4];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const float z = (xd[u] - rm[t]) * rs[t];
            f[t] = __expf(-0.5f * (z * z)) * ra[t];
          }
          if (k4 == 0) f[0] = xd[u];
          vd[u] = make_float4(f[0], f[1], f[2], f[3]);
        }
        if (row >= valid) { r = make_float4(0, 0, 0, 0); vd[u] = r; }
        split_store4(E2h + row * LDW2, 64, 4 * k4, r);      // the FF input: an MFMA operand and (reconstructed) the residual
        split_store4(Dh + row * LDW2, 64, 4 * k4, vd[u]);   // CondGaussian features: only ever an MFMA operand -> split-fp16 layout
      }
    }
  }
  DS_STAMP(1);
  // ---- FF (dmt.py:118-120) on the f16 matrix pipe, chained in registers.  ff_linear3 is computed transposed, 32 hidden features
  //      at a time (lane = row, register i = hidden feature hc*32 + acc_row(i, hh)); SiLU; the accumulator registers 8s .. 8s+7,
  //      split in two fp16 planes, ARE the B fragment of the ff_linear4 MFMAs (weights pre-permuted to that k order: DS_BW_FF4_C)
  f32x16 a4[2][1], a4lo[2][1];
  acc_zero<1>(a4[0]); acc_zero<1>(a4[1]);
  acc_zero<1>(a4lo[0]); acc_zero<1>(a4lo[1]);
  {
    const float* b3 = BW(c, blk, DS_BW_FF3_B);
    const float* W3 = BW(c, blk, DS_BW_FF3_H);
    const uint4* W4 = reinterpret_cast<const uint4*>(BW(c, blk, DS_BW_FF4_C)) + lane;   // [plane][hc][s][ft][lane]
#pragma unroll
    for (int hc = 0; hc < 4; ++hc) {
      f32x16 a3[1], a3lo[1];
#pragma unroll
      for (int q = 0; q < 4; ++q) {   // accumulate onto the bias: registers 4q .. 4q+3 are features hc*32 + 8q + 4hh ..
        const float4 bb = *reinterpret_cast<const float4*>(b3 + hc * 32 + 8 * q + 4 * hh);
        a3[0][4 * q] = bb.x; a3[0][4 * q + 1] = bb.y; a3[0][4 * q + 2] = bb.z; a3[0][4 * q + 3] = bb.w;
      }
      acc_zero<1>(a3lo);
      wave_mma_h<1, true, 4>(E2h, 64, W3, 128, 64, hc * 32, 0, 4, a3, a3lo);
      split_finish<1>(a3, a3lo);
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        h8 y1, y2;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {   // packed-fp32 SiLU, two features at a time, then the fp16 split
          f32x2 v;
          v.x = a3[0][8 * s_ + j]; v.y = a3[0][8 * s_ + j + 1];
          v = ds_silu2(v);
          y1[j] = (_Float16)v.x; y1[j + 1] = (_Float16)v.y;
          y2[j] = (_Float16)((v.x - (float)y1[j]) * 2048.0f); y2[j + 1] = (_Float16)((v.y - (float)y1[j + 1]) * 2048.0f);
        }
#pragma unroll
        for (int ft = 0; ft < 2; ++ft) {
          const h8 w1 = __builtin_bit_cast(h8, W4[(((0 * 4 + hc) * 2 + s_) * 2 + ft) * 64]);
          const h8 w2 = __builtin_bit_cast(h8, W4[(((1 * 4 + hc) * 2 + s_) * 2 + ft) * 64]);
          a4[ft][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, y1, a4[ft][0], 0, 0, 0);
          a4lo[ft][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, y2, a4lo[ft][0], 0, 0, 0);
          a4lo[ft][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, y1, a4lo[ft][0], 0, 0, 0);
        }
      }
    }
    split_finish<1>(a4[0], a4lo[0]);
    split_finish<1>(a4[1], a4lo[1]);
  }
  DS_STAMP(2);
  // ---- gated residual in the transposed layout (lane = row, 4 consecutive features per register quad): the FF input is read back
  //      from its split planes (x1 + x2/2048), e_out goes into the same tile bytes as fp32 once every lane has read
  {
    const float* b4 = BW(c, blk, DS_BW_FF4_B);
    const float* grow = ada + (size_t)rmol[r31] * ADAC + 320;          // edge_gate_mlp of this row's molecule (dmt.py:168)
    float4 xo[2][4];
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n0 = ft * 32 + 8 * q + 4 * hh;
        const float4 g = *reinterpret_cast<const float4*>(grow + n0);
        const float4 bb = *reinterpret_cast<const float4*>(b4 + n0);
        const h4 p0 = *reinterpret_cast<const h4*>(E2h + r31 * LDW2 + n0);
        const h4 p1 = *reinterpret_cast<const h4*>(E2h + r31 * LDW2 + 64 + n0);
        float4 x;
        x.x = fmaf((float)p1[0], 1.0f / 2048.0f, (float)p0[0]); x.y = fmaf((float)p1[1], 1.0f / 2048.0f, (float)p0[1]);
        x.z = fmaf((float)p1[2], 1.0f / 2048.0f, (float)p0[2]); x.w = fmaf((float)p1[3], 1.0f / 2048.0f, (float)p0[3]);
        x.x += g.x * (a4[ft][0][4 * q + 0] + bb.x); x.y += g.y * (a4[ft][0][4 * q + 1] + bb.y);
        x.z += g.z * (a4[ft][0][4 * q + 2] + bb.z); x.w += g.w * (a4[ft][0][4 * q + 3] + bb.w);
        xo[ft][q] = x;
      }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int ft = 0; ft < 2; ++ft)
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<float4*>(&E2[r31][ft * 32 + 8 * q + 4 * hh]) = xo[ft][q];
  }
  // ---- e_out to global, coalesced (4 rows x 256 B per wave-instruction)
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const int idx = lane + u * 64, row = idx >> 4, k4 = idx & 15;
    const float4 v = reinterpret_cast<const float4*>(&E2[row][0])[k4];
    if (row < valid) reinterpret_cast<float4*>(c.ws.e + (size_t)(row0 + row) * 64)[k4] = v;
  }
  DS_STAMP(3);
  // ---- the 64 -> 16 readout slice (fp32 MFMA on the fp32 tile), then [e_out | dist] (128) -> 256 (input_lin edge part +
  //      bias) on the f16 matrix pipe: e_out is re-written in place in the split-fp16 layout first
  {
    f32x16 acc[1];
    acc_zero<1>(acc);
    wave_mma<1>(&E2[0][0], LDW, BW(c, blk, DS_BW_EDGE_RO_W), 32, 0, 0, 8, acc);
    if (r31 < 16) {
      const float b = BW(c, blk, DS_BW_EDGE_RO_B)[r31];
      acc_store<1, 192>(acc, c.ws.edge_hids + (size_t)row0 * 192 + 64 + 16 * blk, valid, [b](int, float v) { return v + b; });
    }
  }
  DS_STAMP(4);
  {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = reinterpret_cast<const float4*>(&E2[(lane + u * 64) >> 4][0])[lane & 15];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");   // every read of the fp32 tile (MFMA fragments above included) before any write
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < 8; ++u) split_store4(E2h + ((lane + u * 64) >> 4) * LDW2, 64, 4 * (lane & 15), v[u]);
  }
  DS_STAMP(5);
  {
    const float* bd = BW(c, blk, DS_BW_ED_B);
    const float* Wd = BW(c, blk, DS_BW_ED_H);
    float* ed = c.ws.ed + (size_t)row0 * 256;
    for (int ch = 0; ch < 8; ++ch) {
      asm volatile("" ::: "memory");
      f32x16 acc[1], lo[1];
      const float b = bd[ch * 32 + r31];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = b;   // accumulate onto the bias: no epilogue arithmetic left
      acc_zero<1>(lo);
      wave_mma_h<1>(E2h, 64, Wd, 256, 128, ch * 32, 0, 4, acc, lo);
      wave_mma_h<1>(Dh, 64, Wd, 256, 128, ch * 32, 4, 8, acc, lo, 4);
      split_finish<1>(acc, lo);
      acc_store<1, 256>(acc, ed + ch * 32, valid, [](int, float v) { return v; });
    }
  }
  DS_STAMP(6);
  DS_STAMP_FLUSH(0);
}

// Block stage F (flat tiles of 32 pairs = 64 directed edges): MultiCondEquiUpdate.  dmt.py:37-60; layers.py:344-347.
// Tile row 2q is pair q's edge a -> b, row 2q+1 its edge b -> a, so one gather of ed_p, ac[a], ac[b] and one adaLN row
// serve both directions (ed is read from HBM once per step instead of twice).
// x = A_r + C_c + ed_{rc} -> LN -> modulate -> [256->256, SiLU] -> [256->3] -> tanh -> head mix -> CoorsNorm; the per-edge
// translation vectors go to tr[2p + dir] and k_pos_update sums them per atom in the reference's edge order.
// The 256->256 GEMM runs on the f16 matrix pipe with split operands (ds_device.h), TRANSPOSED (lane = edge row, registers =
// output features), so that the 256->3 layer is 48 packed-fp32 FMAs per row block on the lane's own registers: the hidden
// activations never touch LDS.
//
// Warp-specialised and persistent: one (NCW + NLW) x 64-thread workgroup per CU owns a contiguous range of tiles.  The NLW
// loader waves (raised priority, one per SIMD) gather + LayerNorm + modulate tile i+1 into the other half of a
// double-buffered split-fp16 LDS tile and precompute its unit coordinate differences while the NCW consumer waves (one per
// SIMD with 4 + 4, priority 0) run the MFMA chain on tile i, each owning 8 / NCW 32-feature chunks of the 256 hidden
// features (both of a wave's chunks against the same X fragments); consumer wave 0 finishes tile i-1's tail (tanh, head mix,
// CoorsNorm -> tr).  One barrier per tile.  Why the roles are split (tools/micro/): a wave that alternates
// gather and MFMA phases loses its memory issue slots to a co-resident MFMA stream, and dependent VALU work next to that
// stream stretches ~2.4x.  What each role needed (profiles/r02_equi_roles.md, in-kernel stamps): the loaders spent 65 % of
// their time waiting for rows they had just requested - their fetches now run a tile ahead (see the loader branch); the
// consumers waited for weights requested one k-block ahead - their weight stream now runs 4 k-blocks ahead per chunk through
// register rings that are refilled for the next tile under the epilogue and the barrier.  4 + 4 waves: both roles need more
// than the 168 registers a 12-wave workgroup leaves.
template <int NCW, int NLW>
__global__ __launch_bounds__((NCW + NLW) * 64) void k_equi_pairs(Ctx c, int blk) {
  ds_fp16_saturate();
  constexpr int T = 64, TP = 32, LDH = 2 * 256 + 8, NCH = 8;
  constexpr int CPW = NCH / NCW;     // 32-feature chunks of the hidden layer per consumer wave
  constexpr int PPW = TP / NLW;      // pairs per loader wave and tile
  constexpr int NPASS = PPW / 2;     // LayerNorm passes (two pairs = four rows each)
  static_assert(NCH % NCW == 0 && TP % NLW == 0 && PPW % 2 == 0, "role split");
  // X tile in the split-fp16 layout of ds_device.h (x = x1 + x2 / 2048): [buffer][row][plane 0 | plane 1 | pad].  A row's
  // 1040-byte slot first receives the fp32 `ed` row by LDS-DMA, then the split LayerNorm output.
  __shared__ __attribute__((aligned(16))) _Float16 Xh[2][T][LDH];      // 133,120 B
  __shared__ __attribute__((aligned(16))) float part[2][NCH][T][4];   //  16,384 B
  __shared__ __attribute__((aligned(16))) float dirs[3][T][4];        //   3,072 B  (unit diff * coord_scale, adjacency bits)
  __shared__ __attribute__((aligned(16))) float cb0[NCH][2][16];      //   1,024 B  coord_mlp.0 bias in accumulator order [chunk][lane half][reg]
  __shared__ __attribute__((aligned(16))) float cw2[NCH][2][4][16];   //   4,096 B  coord_mlp.2 A-fragments [chunk][lane half][output 0..2][reg]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, hh = lane >> 5;
  const int Pp = c.L.Pp;
  // Each workgroup owns a CONTIGUOUS range of tiles: pair rows are molecule-major, so consecutive tiles gather the same few
  // `ac` rows and adaLN rows, which then come from this CU's L1 / this XCD's L2 - with tiles dealt round-robin every XCD
  // fetched every molecule's rows from HBM for itself (counter traffic 1.40 GB per launch against 0.83 GB compulsory).
  const int ntiles_all = (Pp + TP - 1) / TP;
  const int tiles_per_wg = (ntiles_all + (int)gridDim.x - 1) / (int)gridDim.x;
  const int ntiles = min(ntiles_all, ((int)blockIdx.x + 1) * tiles_per_wg);   // one past this workgroup's last tile
  const float* adq = c.ws.ada + blk * DS_ADA_BLOCK_STRIDE + DS_ADA_EQUI;
  const bool consumer = wave < NCW;
  const float cscale = BW(c, blk, DS_BW_COORD_SCALE)[0];
  DS_STAMP_INIT();
  {   // per-block constants of the consumers, laid out so that a lane reads its 16 values as four 16-byte LDS loads
    const float* b0 = BW(c, blk, DS_BW_CM0_B);
    const float* w2 = BW(c, blk, DS_BW_CM2_W);
    for (int idx = tid; idx < NCH * 2 * 16; idx += (NCW + NLW) * 64) {
      const int ch = idx >> 5, h = (idx >> 4) & 1, i = idx & 15, f = ch * 32 + acc_row(i, h);
      cb0[ch][h][i] = b0[f];
#pragma unroll
      for (int o = 0; o < 3; ++o) cw2[ch][h][o][i] = wp_at(w2, 32, f, o);
      cw2[ch][h][3][i] = 0.0f;
    }
  }
  if (consumer) __builtin_amdgcn_s_setprio(0); else __builtin_amdgcn_s_setprio(3);

  const int lw = __builtin_amdgcn_readfirstlane(wave - NCW);   // loader wave lw owns pairs lw*PPW .. lw*PPW + PPW-1 of a tile

  // consumer wave 0: tail of a finished tile, one lane per directed edge (row 2q+dir of the tile); the loaders are the critical
  // path of a tile interval (phase stamps: 97 % busy against the consumers' 88 %), so the tail rides with a consumer
  auto tail = [&](int tile, int pb, int gen) {
    const int p0 = tile * TP;
    const int npairs = min(TP, Pp - p0);
    if ((lane >> 1) < npairs) {
      float4 sacc = reinterpret_cast<const float4*>(&part[pb][0][lane][0])[0];
#pragma unroll
      for (int w = 1; w < NCH; ++w) {   // the eight feature chunks' partial sums, in chunk order
        const float4 pw = reinterpret_cast<const float4*>(&part[pb][w][lane][0])[0];
        sacc.x += pw.x; sacc.y += pw.y; sacc.z += pw.z;
      }
      const float inv[3] = {ds_tanh(sacc.x), ds_tanh(sacc.y), ds_tanh(sacc.z)};
      const float4 d = reinterpret_cast<const float4*>(&dirs[gen][lane][0])[0];
      const int bits = __float_as_int(d.w);
      const float w = ((inv[0] + ((bits & 1) ? inv[1] : 0.0f)) + ((bits & 2) ? inv[2] : 0.0f)) * (1.0f / 3.0f);   // dmt.py:51-53
      float4 t;
      t.x = d.x * w; t.y = d.y * w; t.z = d.z * w; t.w = 0.0f;
      reinterpret_cast<float4*>(c.ws.tr)[(size_t)p0 * 2 + lane] = t;   // coord_diff * inv (dmt.py:56)
    }
  };

  // Two role loops with the same barrier count (one before the first tile, one per tile), kept apart so that neither
  // role's registers stay live through the other's code.
  const int first = blockIdx.x * tiles_per_wg, stride = 1;
  if (first >= ntiles) return;
  if (consumer) {
    // The weights do not depend on anything the workgroup computes: their stream runs PF k-blocks ahead through a register
    // ring that is refilled for the next chunk as soon as a chunk's MFMAs are issued (ds_device.h, wave_mma_h_deep).
    constexpr int PF = DS_EQUI_T2 ? 4 : 8;
    WStreamH wsh[CPW];
#pragma unroll
    for (int cc = 0; cc < CPW; ++cc) wsh[cc] = wstream_h(BW(c, blk, DS_BW_CM0_H), 256, 256, (wave + NCW * cc) * 32);
    WRingH<PF> ring;
    wring_h<PF>(ring, wsh[0], 0);
#if DS_EQUI_T2
    WRingH<PF> ringB;
    wring_h<PF>(ringB, wsh[CPW - 1], 0);
#endif
    __syncthreads();
    DS_STAMP(0);
    int it = 0;
    for (int tile = first; tile < ntiles; tile += stride, ++it) {
      const int buf = it & 1;
      // coord_mlp.0 (256 -> 256) on the f16 matrix pipe with split operands (ds_device.h), transposed: lane = edge row,
      // registers = a chunk's 32 output features; the MFMA chain accumulates onto the coord_mlp.0 bias.
      auto bias_init = [&](int ch, f32x16 (&acc1)[2]) {
        const float4* bp = reinterpret_cast<const float4*>(&cb0[ch][hh][0]);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int q4 = 0; q4 < 4; ++q4) {
            const float4 bv = bp[q4];
            acc1[m][4 * q4] = bv.x; acc1[m][4 * q4 + 1] = bv.y; acc1[m][4 * q4 + 2] = bv.z; acc1[m][4 * q4 + 3] = bv.w;
          }
      };
      // SiLU (dmt.py:32-33), then coord_mlp.2 (256 -> 3, dmt.py:34) on the VALU, fp32: a lane holds 16 of its row's 32 hidden
      // features of this chunk, so the three outputs are 3 x 16 fused multiply-adds per row block (packed two at a time) - 48
      // v_pk_fma against the 32 64-cycle 32x32x2 fp32 MFMAs (29 of 32 output rows padding) that used to take 40 % of this wave's
      // matrix-pipe time.
      auto epilogue = [&](int ch, const f32x16 (&acc1)[2]) {
        f32x2 so[2][3];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int o = 0; o < 3; ++o) so[m][o] = f32x2{0.0f, 0.0f};
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          float4 wv[3];
#pragma unroll
          for (int o = 0; o < 3; ++o) wv[o] = reinterpret_cast<const float4*>(&cw2[ch][hh][o][0])[q4];
#pragma unroll
          for (int m = 0; m < 2; ++m) {
            f32x2 y0, y1;   // packed-fp32, two hidden features at a time
            y0.x = acc1[m][4 * q4]; y0.y = acc1[m][4 * q4 + 1]; y1.x = acc1[m][4 * q4 + 2]; y1.y = acc1[m][4 * q4 + 3];
            y0 = ds_silu2(y0); y1 = ds_silu2(y1);
#pragma unroll
            for (int o = 0; o < 3; ++o) {
              so[m][o] = __builtin_elementwise_fma(f32x2{wv[o].x, wv[o].y}, y0, so[m][o]);
              so[m][o] = __builtin_elementwise_fma(f32x2{wv[o].z, wv[o].w}, y1, so[m][o]);
            }
          }
        }
        // the other 16 features of the row sit in lane ^ 32: one v_permlane32_swap per output hands row block 0's upper
        // halves to lanes 0-31 and row block 1's lower halves to lanes 32-63, so lane l ends up owning tile row l
        float4 o4;
        float* op = &o4.x;
#pragma unroll
        for (int o = 0; o < 3; ++o) {
          const float s0 = so[0][o].x + so[0][o].y, s1 = so[1][o].x + so[1][o].y;
          const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(s0), __float_as_uint(s1), false, false);
          op[o] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
        }
        o4.w = 0.0f;
        reinterpret_cast<float4*>(&part[buf][ch][lane][0])[0] = o4;
      };
#if DS_EQUI_T2
      if constexpr (CPW == 2) {   // both chunks of the wave against the same X fragments: half the LDS operand reads
        f32x16 accA[2], loA[2], accB[2], loB[2];
        bias_init(wave, accA);
        bias_init(wave + NCW, accB);
        acc_zero<2>(loA);
        acc_zero<2>(loB);
        wave_mma_h_deep_t2<2, true, 16, PF>(&Xh[buf][0][0], 256, wsh[0], wsh[1], ring, ringB, 0, accA, loA, accB, loB);
        wring_h<PF>(ring, wsh[0], 0);    // the next tile's first blocks fly under the epilogues and the barrier
        wring_h<PF>(ringB, wsh[1], 0);
        DS_STAMP(3);
        split_finish<2>(accA, loA);
        epilogue(wave, accA);
        DS_STAMP(10);
        split_finish<2>(accB, loB);
        epilogue(wave + NCW, accB);
        DS_STAMP(11);
      } else
#endif
      {
#pragma unroll
        for (int cc = 0; cc < CPW; ++cc) {
          const int ch = wave + NCW * cc;
          f32x16 acc1[2], acclo[2];
          bias_init(ch, acc1);
          acc_zero<2>(acclo);
          wave_mma_h_deep<2, true, 16, PF>(&Xh[buf][0][0], 256, wsh[cc], ring, 0, acc1, acclo);
          wring_h<PF>(ring, wsh[(cc + 1) % CPW], 0);   // the next chunk's (next tile's) first blocks fly under the epilogue and the barrier
          split_finish<2>(acc1, acclo);
          epilogue(ch, acc1);
        }
      }
      if (wave == 0 && it > 0) tail(tile - stride, buf ^ 1, (it + 2) % 3);   // previous tile: its partial sums were complete one barrier ago
      DS_STAMP(1);
      __syncthreads();
      DS_STAMP(2);
    }
    if (wave == 0) tail(ntiles - 1, (it + 1) & 1, (it + 2) % 3);   // the last tile (it - 1)
    DS_STAMP_FLUSH(0);
  } else {
    // Loader wave: gather + LayerNorm + modulate PPW pairs (2 PPW rows) per tile, four rows per pass - both directions of two
    // pairs, one row per 16-lane DPP row, lane j of it holding the float4s at columns 4j + 64u, so the LayerNorm sums are four
    // DPP steps with every lane busy.  Memory latency is what a loader wave spends its time on (phase stamps,
    // profiles/r02_equi_roles.md: 65 % of it sat in the wait at the head of each pass when every pass fetched its own rows),
    // so the fetches run AHEAD of the arithmetic:
    //  * the h_col / h_row parts of the b atoms (`ac`, 4 gathers per pass) of tile k+1 are requested into pass bt's registers
    //    as soon as pass bt of tile k has consumed them - a whole tile interval before they are needed;
    //  * the pair-table entries are fetched two tiles ahead (lane = (pair slot, direction));
    //  * the pair rows `ed` (the one HBM stream, 0.66 GB per launch) go by LDS-DMA straight into the X row slots they are
    //    consumed from, all PPW rows at the head of the interval - the buffer is only free then - and that one latency per
    //    tile is what remains exposed; the previous tile's tail and the unit vectors are computed under it;
    //  * the a atom's part is kept while a does not change (pairs are (a, b)-row-major and a wave owns consecutive pairs).
    const int g = lane >> 4, j = lane & 15, up = g >> 1, gdir = g & 1, slot = (lane >> 1) & (PPW - 1), dir = lane & 1;
    const float4* ac4 = reinterpret_cast<const float4*>(c.ws.ac);
    const float4* ed4 = reinterpret_cast<const float4*>(c.ws.ed);
    int pv_c = 0, av_c = 0, bv_c = 0, mv_c = 0, adj_c = 0;   // table entries of the tile being produced
    int pv_x = 0, av_x = 0, bv_x = 0, mv_x = 0, adj_x = 0;   // ... and of the one after it
#define DS_EQUI_FETCH_IDX(TILE, PV, AV, BV, MV, ADJ)                                                   \
    do {                                                                                               \
      const int p0_ = (TILE) * TP, qv_ = lw * PPW + slot;                                              \
      PV = qv_ < min(TP, Pp - p0_) ? p0_ + qv_ : 0; /* pairs past the end gather pair 0, zeroed below */ \
      AV = c.L.pair_a[PV]; BV = c.L.pair_b[PV]; MV = c.L.pair_mol[PV]; ADJ = c.ws.adj[PV];             \
    } while (0)
    // b: h_row part for b -> a lanes, h_col part for a -> b lanes (dmt.py:39,45)
#define DS_EQUI_ISSUE_B(BT, BVREG)                                                                     \
    do {                                                                                               \
      const int r0_ = __builtin_amdgcn_readlane(BVREG, 4 * (BT)), r1_ = __builtin_amdgcn_readlane(BVREG, 4 * (BT) + 2); \
      const float4* pb_ = ac4 + (size_t)(up ? r1_ : r0_) * 128 + (gdir ? 0 : 64) + j;                  \
      _Pragma("unroll") for (int u = 0; u < 4; ++u) Bv[BT][u] = pb_[16 * u];                           \
    } while (0)
    float4 Bv[NPASS][4], Ka[4], sh[4], sc[4];
    int cur_m = -1, cur_a = -1;
    DS_EQUI_FETCH_IDX(first, pv_c, av_c, bv_c, mv_c, adj_c);
    float4 pr = make_float4(0, 0, 0, 0), pc = pr;
    if (lane < 2 * PPW) {
      pr = reinterpret_cast<const float4*>(c.ws.pos)[dir ? bv_c : av_c];   // row atom (edge_index[0])
      pc = reinterpret_cast<const float4*>(c.ws.pos)[dir ? av_c : bv_c];
    }
    if (first + stride < ntiles) DS_EQUI_FETCH_IDX(first + stride, pv_x, av_x, bv_x, mv_x, adj_x);
#pragma unroll
    for (int bt = 0; bt < NPASS; ++bt) DS_EQUI_ISSUE_B(bt, bv_c);
    DS_STAMP(4);
    int k = 0;
    for (int tile = first;; tile += stride, ++k) {
      const bool have = tile < ntiles;
      if (have) {
        const int buf = k & 1, gen = k % 3;
        const int p0 = tile * TP, npairs = min(TP, Pp - p0);
        const bool have_next = tile + stride < ntiles;
        int na[PPW], pm[PPW], pp[PPW];
#pragma unroll
        for (int u = 0; u < PPW; ++u) {
          pp[u] = __builtin_amdgcn_readlane(pv_c, 2 * u); na[u] = __builtin_amdgcn_readlane(av_c, 2 * u);
          pm[u] = __builtin_amdgcn_readlane(mv_c, 2 * u);
        }
        // a: h_row part for a -> b lanes, h_col part for b -> a lanes; M: adaLN rows of the molecule (they change once per ~160 pairs)
#define DS_EQUI_ISSUE_A(BT)                                                                            \
        do {                                                                                           \
          const int ra_ = up ? na[2 * (BT) + 1] : na[2 * (BT)];                                        \
          if (ra_ != cur_a) {                                                                          \
            const float4* pa_ = ac4 + (size_t)ra_ * 128 + (gdir ? 64 : 0) + j;                         \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) Ka[u] = pa_[16 * u];                         \
            cur_a = ra_;                                                                               \
          }                                                                                            \
        } while (0)
#define DS_EQUI_ISSUE_M(BT)                                                                            \
        do {                                                                                           \
          const int m_ = up ? pm[2 * (BT) + 1] : pm[2 * (BT)];                                         \
          if (m_ != cur_m) {                                                                           \
            const float4* ps_ = reinterpret_cast<const float4*>(adq + (size_t)m_ * ADAC) + j;          \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) { sh[u] = ps_[16 * u]; sc[u] = ps_[64 + 16 * u]; } /* shift, scale (dmt.py:44) */ \
            cur_m = m_;                                                                                \
          }                                                                                            \
        } while (0)
        DS_EQUI_ISSUE_A(0);
        DS_EQUI_ISSUE_M(0);
#pragma unroll
        for (int i = 0; i < PPW; ++i)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ed4 + (size_t)pp[i] * 64 + lane),
                                           (__attribute__((address_space(3))) void*)&Xh[buf][2 * (lw * PPW + i)][0], 16, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        DS_STAMP(5);
        if (lane < 2 * PPW) {   // unit vector of pos[row] - pos[col], scaled (layers.py:345-346), + adjacency bits, for the tail
          const float dx = pr.x - pc.x, dy = pr.y - pc.y, dz = pr.z - pc.z;
          const float nrm = fmaxf(__builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz), 1e-8f);
          const float sn = cscale * __builtin_amdgcn_rcpf(nrm);   // hardware sqrt / rcp (1 ulp): loader VALU issues are the scarce resource
          float4 d;
          d.x = dx * sn; d.y = dy * sn; d.z = dz * sn; d.w = __int_as_float(adj_c);
          reinterpret_cast<float4*>(&dirs[gen][2 * (lw * PPW + slot) + dir][0])[0] = d;
          if (have_next) {   // the next tile's positions, a tile interval ahead like its other rows
            pr = reinterpret_cast<const float4*>(c.ws.pos)[dir ? bv_x : av_x];
            pc = reinterpret_cast<const float4*>(c.ws.pos)[dir ? av_x : bv_x];
          }
        }
        DS_STAMP(6);
        // the DMA'd rows must have landed before they are read back (and nothing may be hoisted above this)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        DS_STAMP(8);
#pragma unroll
        for (int bt = 0; bt < NPASS; ++bt) {
          const int q = lw * PPW + 2 * bt + up;
          float4 x[4];   // a -> b: input_lin([h_a, h_b, e, d]);  b -> a: input_lin([h_b, h_a, e, d])  (dmt.py:39,45)
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const float4 e = reinterpret_cast<const float4*>(&Xh[buf][2 * q][0])[16 * u + j];
            x[u].x = (Ka[u].x + Bv[bt][u].x) + e.x; x[u].y = (Ka[u].y + Bv[bt][u].y) + e.y;
            x[u].z = (Ka[u].z + Bv[bt][u].z) + e.z; x[u].w = (Ka[u].w + Bv[bt][u].w) + e.w;
          }
          __builtin_amdgcn_sched_barrier(0);
          if (have_next) DS_EQUI_ISSUE_B(bt, bv_x);     // the next tile's pass bt, a tile interval ahead
          if (bt + 1 < NPASS) DS_EQUI_ISSUE_A(bt + 1);  // next pass's a part, if a changes
          __builtin_amdgcn_sched_barrier(0);
          DS_STAMP(12);
          ln_mod_quad256(x, sh, sc);
          _Float16* xr = &Xh[buf][2 * q + gdir][0];
#pragma unroll
          for (int u = 0; u < 4; ++u) split_store4(xr, 256, 64 * u + 4 * j, x[u]);
          if (bt + 1 < NPASS) DS_EQUI_ISSUE_M(bt + 1);  // sh / sc are free now: next pass's adaLN rows, if the molecule changes
          DS_STAMP(14);
        }
        if (npairs < TP) {   // last tile only: rows of the pairs past the end (they were computed from pair 0) become zero rows
          for (int i = 0; i < PPW; ++i) {
            const int qz = lw * PPW + i;
            if (qz >= npairs) {   // two 1040-byte row slots = 130 float4
              float4* z = reinterpret_cast<float4*>(&Xh[buf][2 * qz][0]);
              z[lane] = make_float4(0, 0, 0, 0);
              z[64 + lane] = make_float4(0, 0, 0, 0);
              if (lane < 2) z[128 + lane] = make_float4(0, 0, 0, 0);
            }
          }
        }
      }
      DS_STAMP(7);
      __syncthreads();
      DS_STAMP(9);
      if (!have) break;
      pv_c = pv_x; av_c = av_x; bv_c = bv_x; mv_c = mv_x; adj_c = adj_x;
      if (tile + 2 * stride < ntiles) DS_EQUI_FETCH_IDX(tile + 2 * stride, pv_x, av_x, bv_x, mv_x, adj_x);
    }
#undef DS_EQUI_FETCH_IDX
#undef DS_EQUI_ISSUE_B
#undef DS_EQUI_ISSUE_A
#undef DS_EQUI_ISSUE_M
    DS_STAMP_FLUSH(NCW * 64);
  }
}

// One wave per molecule: pos_r += sum_c trans(r -> c) in ascending c (the reference's scatter-add order, dmt.py:57-58),
// then the per-layer CoM removal (dmt.py:385-386; models/utils.py:38-45).
__global__ __launch_bounds__(64) void k_pos_update(Ctx c, int last) {
  const int m = blockIdx.x, r = threadIdx.x;
  const int n0 = c.L.node_off[m], n = c.L.node_off[m + 1] - n0;
  const int p0 = c.L.pair_off[m];
  if (n <= 0) return;
  float x = 0.0f, y = 0.0f, z = 0.0f;
  if (r < n) {
    const float4* tr = reinterpret_cast<const float4*>(c.ws.tr);
    float sx = 0.0f, sy = 0.0f, sz = 0.0f;
    for (int c0 = 0; c0 < n; c0 += 8) {   // eight partners' vectors in flight, added in ascending partner order
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int cc = c0 + u;
        t[u] = make_float4(0, 0, 0, 0);
        if (cc < n && cc != r) {
          const int lo = r < cc ? r : cc, hi = r < cc ? cc : r;
          const int pl = lo * (2 * n - lo - 1) / 2 + (hi - lo - 1);
          t[u] = tr[(size_t)(p0 + pl) * 2 + (r < cc ? 0 : 1)];
        }
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (c0 + u < n && c0 + u != r) { sx += t[u].x; sy += t[u].y; sz += t[u].z; }
    }
    const float* pp = c.ws.pos + (size_t)(n0 + r) * 4;
    x = pp[0] + sx; y = pp[1] + sy; z = pp[2] + sz;
  }
  const float fn = (float)n;
  const float mx = wave_sum(x) / fn, my = wave_sum(y) / fn, mz = wave_sum(z) / fn;
  if (r < n) {
    float* pp = c.ws.pos + (size_t)(n0 + r) * 4;
    const float ox = x - mx, oy = y - my, oz = z - mz;
    pp[0] = ox; pp[1] = oy; pp[2] = oz;
    if (last && (isnan(ox) || isnan(oy) || isnan(oz))) atomicOr(&c.ws.flags[1], 1);
  }
}

// ------------------------------------------------------------------------------------------------
// Readout: node_pred_mlp (768->256->128->6) -> out_xh[..., 3:9] (dmt.py:391-393).
__global__ __launch_bounds__(256) void k_node_readout(Ctx c, float* __restrict__ out_xh) {
  ds_fp16_saturate();
  constexpr int T = 32;
  // the 768- and 256-wide tiles in the split-fp16 layout (ds_device.h): their GEMMs run on the f16 pipe; the last, 128 -> 6, stays fp32
  __shared__ __attribute__((aligned(16))) _Float16 X[T][2 * 768 + 8];
  __shared__ __attribute__((aligned(16))) _Float16 Y1[T][2 * 256 + 8];
  __shared__ __attribute__((aligned(16))) float Y2[T][128 + DS_LDP];
  __shared__ int dn[T];   // dense output row of each tile row (-1 past the end)
  const int tid = threadIdx.x, wave = tid >> 6, row0 = blockIdx.x * T;
  const int Nn = c.L.Nn;
  if (tid < T) dn[tid] = row0 + tid < Nn ? c.L.node_dense[row0 + tid] : -1;
  for (int i0 = tid; i0 < T * 192; i0 += 256 * 8) {   // eight loads in flight per thread (a plain loop pays one trip per load)
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = i0 + u * 256, row = idx / 192, k4 = idx - row * 192;
      v[u] = reinterpret_cast<const float4*>(c.ws.atom_hids + (size_t)min(row0 + row, Nn - 1) * 768)[k4];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = i0 + u * 256, row = idx / 192, k4 = idx - row * 192;
      split_store4(&X[row][0], 768, 4 * k4, row0 + row < Nn ? v[u] : make_float4(0, 0, 0, 0));
    }
  }
  __syncthreads();
  {
    const float* b = GW(c, DS_GW_NP0_B);
    for (int ch = wave; ch < 8; ch += 4) {
      asm volatile("" ::: "memory");
      const int col = ch * 32 + (tid & 31), hhf = (tid & 63) >> 5;
      const float bc = b[col];
      f32x16 acc[1], lo[1];
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[0][i] = bc;
      acc_zero<1>(lo);
      wave_mma_h_ring<1, false, 48, 8>(&X[0][0], 768, GW(c, DS_GW_NP0_H), 256, 768, ch * 32, 0, acc, lo);
      split_finish<1>(acc, lo);
#pragma unroll
      for (int i = 0; i < 16; ++i) split_store1(&Y1[acc_row(i, hhf)][0], 256, col, ds_silu(acc[0][i]));
    }
  }
  __syncthreads();
  {
    const float* b = GW(c, DS_GW_NP2_B);
    const int col = wave * 32 + (tid & 31), hhf = (tid & 63) >> 5;
    const float bc = b[col];
    f32x16 acc[1], lo[1];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[0][i] = bc;
    acc_zero<1>(lo);
    wave_mma_h_ring<1, false, 16, 8>(&Y1[0][0], 256, GW(c, DS_GW_NP2_H), 128, 256, wave * 32, 0, acc, lo);
    split_finish<1>(acc, lo);
#pragma unroll
    for (int i = 0; i < 16; ++i) Y2[acc_row(i, hhf)][col] = ds_silu(acc[0][i]);
  }
  __syncthreads();
  {
    const float* b = GW(c, DS_GW_NP4_B);
    const float bc = b[min((int)(threadIdx.x & 31), 5)];
    tile_gemm<1, 1>(&Y2[0][0], 128 + DS_LDP, 128, GW(c, DS_GW_NP4_W), 32, 1, [&](int row, int col, float v) {
      if (col < 6 && dn[row] >= 0) out_xh[(size_t)dn[row] * 9 + 3 + col] = v + bc;
    });
  }
}

// Readout: edge_exist_mlp / edge_type_mlp (192->64->32->1 each) -> dense symmetric out_edge (dmt.py:394-399).
// ROW-PARALLEL and register-resident: every wave owns 32 pair rows and there is no LDS tile and no barrier (the tile form had
// seven barrier-separated phases per 64 rows and ran at 0.42 ms per step against ~0.1 ms of HBM time).  A lane (row r, k-half h)
// loads its row's 8 consecutive inputs of every 16-deep k-block straight from `edge_hids` and keeps them as split-fp16 B
// fragments (96 registers for the 32 x 192 tile); 192 -> 64 runs transposed (lane = row, registers = 32 output features per
// chunk), its SiLU'd accumulators - split in registers - are the B operand of the 64 -> 32 MFMAs (weights in accumulator-chain
// order, DS_GW_EX2_C / ET2_C), and 32 -> 1 is 16 fused multiply-adds per lane plus one v_permlane32_swap.
__global__ __launch_bounds__(256) void k_edge_readout(Ctx c, float* __restrict__ out_edge) {
  ds_fp16_saturate();
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hh = lane >> 5;
  const int Pp = c.L.Pp;
  const int row0 = (blockIdx.x * 4 + wave) * 32;
  if (row0 >= Pp) return;                    // whole wave leaves; nothing below synchronises across waves
  const int p = min(row0 + r, Pp - 1);
  const bool valid = row0 + r < Pp;
  // dense output offsets of (a, b) and (b, a): an index chain of two trips, started first
  const int pa = c.L.pair_a[p], pb = c.L.pair_b[p], pm = c.L.pair_mol[p];
  h8 x1[12], x2[12];
  {
    const float4* src = reinterpret_cast<const float4*>(c.ws.edge_hids + (size_t)p * 192 + 8 * hh);
#pragma unroll
    for (int g4 = 0; g4 < 3; ++g4) {         // four k-blocks (eight 16-byte loads) in flight at a time
      float4 v[4][2];
#pragma unroll
      for (int u = 0; u < 4; ++u) { v[u][0] = src[(g4 * 4 + u) * 4]; v[u][1] = src[(g4 * 4 + u) * 4 + 1]; }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float xs[8] = {v[u][0].x, v[u][0].y, v[u][0].z, v[u][0].w, v[u][1].x, v[u][1].y, v[u][1].z, v[u][1].w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          const _Float16 a1 = (_Float16)xs[t];
          x1[g4 * 4 + u][t] = a1;
          x2[g4 * 4 + u][t] = (_Float16)((xs[t] - (float)a1) * 2048.0f);
        }
      }
    }
  }
  const int da = c.L.node_dense[pa], db = c.L.node_dense[pb];
  const int mN = pm * c.L.N;
  const size_t oab = ((size_t)da * c.L.N + (db - mN)) * 2, oba = ((size_t)db * c.L.N + (da - mN)) * 2;
  for (int ch = 0; ch < 2; ++ch) {   // channel 0: edge_exist_mlp, channel 1: edge_type_mlp (dmt.py:394)
    const int g0 = ch == 0 ? DS_GW_EX0_W : DS_GW_ET0_W;
    const float* b0 = GW(c, g0 + 1);
    const float* b2 = GW(c, g0 + 3);
    const uint4* W2 = reinterpret_cast<const uint4*>(GW(c, ch == 0 ? DS_GW_EX2_C : DS_GW_ET2_C)) + lane;   // [plane][hc][s][lane]
    f32x16 a2, a2lo;
#pragma unroll
    for (int q = 0; q < 4; ++q) {      // accumulate onto the layer's bias: registers 4q .. 4q+3 are features 8q + 4h ..
      const float4 bb = *reinterpret_cast<const float4*>(b2 + 8 * q + 4 * hh);
      a2[4 * q] = bb.x; a2[4 * q + 1] = bb.y; a2[4 * q + 2] = bb.z; a2[4 * q + 3] = bb.w;
    }
#pragma unroll
    for (int i = 0; i < 16; ++i) a2lo[i] = 0.0f;
#pragma unroll
    for (int hc = 0; hc < 2; ++hc) {
      asm volatile("" ::: "memory");
      const WStreamH ws0 = wstream_h(GW(c, ch == 0 ? DS_GW_EX0_H : DS_GW_ET0_H), 64, 192, hc * 32);
      WRingH<6> ring;
      wring_h<6>(ring, ws0, 0);
      f32x16 a1, a1lo;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 bb = *reinterpret_cast<const float4*>(b0 + hc * 32 + 8 * q + 4 * hh);
        a1[4 * q] = bb.x; a1[4 * q + 1] = bb.y; a1[4 * q + 2] = bb.z; a1[4 * q + 3] = bb.w;
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) a1lo[i] = 0.0f;
#pragma unroll
      for (int kb = 0; kb < 12; ++kb) {
        const h8 w1 = ring.w1[kb % 6], w2 = ring.w2[kb % 6];
        a1lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x2[kb], a1lo, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, x1[kb], a1, 0, 0, 0);
        a1lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, x1[kb], a1lo, 0, 0, 0);
        if (kb + 6 < 12) { ring.w1[kb % 6] = wload_h(ws0, 0, kb + 6); ring.w2[kb % 6] = wload_h(ws0, 1, kb + 6); }
        __builtin_amdgcn_sched_barrier(0);
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) a1[i] = fmaf(a1lo[i], 1.0f / 2048.0f, a1[i]);
#pragma unroll
      for (int s_ = 0; s_ < 2; ++s_) {
        h8 y1, y2;
#pragma unroll
        for (int j = 0; j < 8; j += 2) {   // packed-fp32 SiLU, two features at a time, then the fp16 split
          f32x2 v;
          v.x = a1[8 * s_ + j]; v.y = a1[8 * s_ + j + 1];
          v = ds_silu2(v);
          y1[j] = (_Float16)v.x; y1[j + 1] = (_Float16)v.y;
          y2[j] = (_Float16)((v.x - (float)y1[j]) * 2048.0f); y2[j + 1] = (_Float16)((v.y - (float)y1[j + 1]) * 2048.0f);
        }
        const h8 w1 = __builtin_bit_cast(h8, W2[((0 * 2 + hc) * 2 + s_) * 64]);
        const h8 w2 = __builtin_bit_cast(h8, W2[((1 * 2 + hc) * 2 + s_) * 64]);
        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, y1, a2, 0, 0, 0);
        a2lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, y2, a2lo, 0, 0, 0);
        a2lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(w2, y1, a2lo, 0, 0, 0);
      }
    }
    // 32 -> 1 on the VALU: this lane holds features (i & 3) + 8 (i >> 2) + 4 h of its row; the other half sits in lane ^ 32
    const float* W4 = GW(c, g0 + 4);
    f32x2 s2 = {0.0f, 0.0f};
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
      f32x2 v;
      v.x = fmaf(a2lo[i], 1.0f / 2048.0f, a2[i]); v.y = fmaf(a2lo[i + 1], 1.0f / 2048.0f, a2[i + 1]);
      v = ds_silu2(v);
      const f32x2 w = {wp_at(W4, 32, acc_row(i, hh), 0), wp_at(W4, 32, acc_row(i + 1, hh), 0)};
      s2 = __builtin_elementwise_fma(w, v, s2);
    }
    const float part = s2.x + s2.y;
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(part), __float_as_uint(part), false, false);
    const float val = (__uint_as_float(sw[0]) + __uint_as_float(sw[1])) + GW(c, g0 + 5)[0];   // 0.5*(x + x) == x: the symmetrisation of dmt.py:399 is exact here
    if (valid && hh == 0) {
      out_edge[oab + ch] = val;
      out_edge[oba + ch] = val;
    }
  }
}

// Final positions: mask, NaN guard (whole batch -> zeros), CoM removal (dmt.py:402-412).
__global__ void k_final_pos(Ctx c, float* __restrict__ out_xh) {
  const int m = blockIdx.x * blockDim.x + threadIdx.x;
  if (m >= c.L.B) return;
  const int n0 = c.L.node_off[m], n = c.L.node_off[m + 1] - n0;
  const bool nan = c.ws.flags[1] != 0;
  float sx = 0.0f, sy = 0.0f, sz = 0.0f;
  for (int a = 0; a < n; ++a) {
    const float* pp = c.ws.pos + (size_t)(n0 + a) * 4;
    if (!nan) { sx += pp[0]; sy += pp[1]; sz += pp[2]; }
  }
  const float fn = (float)n;
  for (int a = 0; a < n; ++a) {
    const float* pp = c.ws.pos + (size_t)(n0 + a) * 4;
    float* o = out_xh + (size_t)c.L.node_dense[n0 + a] * 9;
    const float px = nan ? 0.0f : pp[0], py = nan ? 0.0f : pp[1], pz = nan ? 0.0f : pp[2];
    o[0] = px - sx / fn; o[1] = py - sy / fn; o[2] = pz - sz / fn;
  }
}

// temb = tm3_out (+ ctx); store SiLU(temb) — the input of every *time_mlp Linear (dmt.py:354; nn.SiLU first in each).
__global__ void k_temb_finish(Ctx c, const float* __restrict__ tm3, int tm3_rows, const float* __restrict__ ctx) {
  ds_fp16_saturate();
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)c.L.B * 1024) return;
  const size_t b = i >> 10, col = i & 1023;
  float v = tm3[(tm3_rows == 1 ? 0 : b) * 1024 + col];
  if (ctx) v += ctx[i];
  // the adaLN GEMM's A operand, written once in the split-fp16 layout (two planes per row) instead of re-split per tile
  split_store1(reinterpret_cast<_Float16*>(c.ws.temb_silu) + b * 2048, 1024, (int)col, ds_silu(v));
}

// The per-step adaLN table GEMM ada[M, N] = temb_silu[M, 1024] * W + bias on the f16 matrix pipe with split operands
// (ds_device.h).  A arrives pre-split from k_temb_finish (halves [M][2][K]); 64 x 128 output tile per workgroup, each wave 64
// rows x 32 columns (a weight fragment feeds 6 MFMAs; 128-row tiles spilled their staging registers), A chunks of 64 k
// double-buffered in LDS with the next chunk fetched into registers while the current one is multiplied.
__global__ __launch_bounds__(256) void k_gemm_ada(const _Float16* __restrict__ A, const float* __restrict__ Wh, const float* __restrict__ bias,
                                                  float* __restrict__ C, int ldc, int M, int K, int N) {
  ds_fp16_saturate();
  constexpr int T = 64, KC = 64, LDH = 2 * KC + 8, MT = T / 32;
  __shared__ __attribute__((aligned(16))) _Float16 X[2][T][LDH];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int row0 = blockIdx.x * T;
  const int col0 = (blockIdx.y * 4 + wave) * 32;
  const bool active = col0 < N;
  const int nchunks = K / KC;
  // staging registers as four named values (an array here ended up in scratch memory)
  float4 st0, st1, st2, st3;
  const int srow = tid >> 4, spiece = tid & 15;               // thread's piece of rows srow, srow + 16, srow + 32, srow + 48
  const size_t scol = (size_t)(spiece >> 3) * K + (spiece & 7) * 8;
  auto fetch = [&](int kc) {
    const _Float16* base = A + scol + kc * KC;
    st0 = *reinterpret_cast<const float4*>(base + (size_t)min(row0 + srow, M - 1) * 2 * K);
    st1 = *reinterpret_cast<const float4*>(base + (size_t)min(row0 + srow + 16, M - 1) * 2 * K);
    st2 = *reinterpret_cast<const float4*>(base + (size_t)min(row0 + srow + 32, M - 1) * 2 * K);
    st3 = *reinterpret_cast<const float4*>(base + (size_t)min(row0 + srow + 48, M - 1) * 2 * K);
  };
  auto stash = [&](int buf) {   // planes are adjacent in a tile row: piece 0..7 plane 0, 8..15 plane 1
    *reinterpret_cast<float4*>(&X[buf][srow][spiece * 8]) = st0;
    *reinterpret_cast<float4*>(&X[buf][srow + 16][spiece * 8]) = st1;
    *reinterpret_cast<float4*>(&X[buf][srow + 32][spiece * 8]) = st2;
    *reinterpret_cast<float4*>(&X[buf][srow + 48][spiece * 8]) = st3;
  };
  f32x16 acc[MT], lo[MT];
  acc_zero<MT>(acc);
  acc_zero<MT>(lo);
  // weights: a chunk's four k-blocks sit in a register ring that is re-requested for the NEXT chunk as soon as this chunk's
  // MFMAs are issued - their L2 round trip flies under the A staging and the barrier (ds_device.h, wave_mma_h_deep)
  const WStreamH wsw = wstream_h(Wh, N, K, active ? col0 : 0);
  WRingH<4> ring;
  wring_h<4>(ring, wsw, 0);
  fetch(0);
  stash(0);
  __syncthreads();
  for (int kc = 0; kc < nchunks; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < nchunks) fetch(kc + 1);
    if (active) wave_mma_h_deep<MT, false, 4, 4>(&X[cur][0][0], KC, wsw, ring, kc * 4, acc, lo, kc * 4);
    if (kc + 1 < nchunks) { wring_h<4>(ring, wsw, kc * 4 + 4); stash(cur ^ 1); }
    __syncthreads();
  }
  if (!active) return;
  split_finish<MT>(acc, lo);
  const int lane = tid & 63, r = lane & 31, hh = lane >> 5, col = col0 + r;
  const float bcol = bias ? bias[col] : 0.0f;
  const unsigned long long pw = reinterpret_cast<unsigned long long>(C + (size_t)row0 * ldc + col0);
  const unsigned long long pu = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw >> 32))) << 32) |
                                static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw)));
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(pu), 0, 0x7fffffff, 0x00020000);
  const int voff = (4 * hh * ldc + r) * 4, rowb = ldc * 4;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = m * 32 + (i & 3) + 8 * (i >> 2);
      if (row0 + row + 4 * hh < M) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[m][i] + bcol), rc, voff, row * rowb, 0);
    }
}

// ------------------------------------------------------------------------------------------------
// Generic GEMM (see header).  64x128 output tile per workgroup, A staged through LDS in K-chunks of 64.
// Row-group addressing lets A be an unfold view (SpecFormer patches), C a slice of a [B, L, D] token buffer and
// R a per-position table broadcast over molecules, without any host-side copy.
struct GemmArgs {
  const float* A; int64_t lda; int a_grp_rows; int64_t a_grp_stride;
  const float* Wp; const float* bias;
  float* C; int64_t ldc; int c_grp_rows; int64_t c_grp_stride;
  int M, K, N, Npad;
  const float* R; int64_t ldr; int r_grp_rows;
  const float* cs; const float* csh;
  int a_silu;
};

__device__ __forceinline__ size_t grp_off(int row, int64_t ld, int grp_rows, int64_t grp_stride) {
  if (grp_rows <= 0) return (size_t)row * ld;
  const int g = row / grp_rows;
  return (size_t)g * grp_stride + (size_t)(row - g * grp_rows) * ld;
}

template <int ACT>
__global__ __launch_bounds__(256) void k_gemm(GemmArgs g) {
  constexpr int T = 64, KC = 64;
  __shared__ __attribute__((aligned(16))) float X[T][KC + DS_LDP];
  __shared__ size_t arow[T];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int row0 = blockIdx.x * T;
  const int col0 = (blockIdx.y * 4 + wave) * 32;
  const bool active = col0 < g.Npad;
  if (tid < T) arow[tid] = (row0 + tid < g.M) ? grp_off(row0 + tid, g.lda, g.a_grp_rows, g.a_grp_stride) : 0;
  f32x16 acc[2];
  acc_zero<2>(acc);
  const int Kpad = (g.K + 7) & ~7;
  for (int k0 = 0; k0 < Kpad; k0 += KC) {
    __syncthreads();
    for (int idx = tid; idx < T * KC; idx += 256) {
      const int row = idx >> 6, k = idx & 63;
      float v = 0.0f;
      if (row0 + row < g.M && k0 + k < g.K) {
        v = g.A[arow[row] + k0 + k];
        if (g.a_silu) v = ds_silu(v);
      }
      X[row][k] = v;
    }
    __syncthreads();
    if (active) {
      const int kgs = min(KC, Kpad - k0) >> 3;
      wave_mma<2>(&X[0][0], KC + DS_LDP, g.Wp + (size_t)(k0 >> 3) * 2 * g.Npad * 4, g.Npad, col0, 0, kgs, acc);
    }
  }
  if (!active) return;
  acc_foreach<2>(acc, 0, col0, [&](int row, int col, float v) {
    const int gr = row0 + row;
    if (gr < g.M && col < g.N) {
      if (g.bias) v += g.bias[col];
      v = ds_act<ACT>(v);
      if (g.R) v += g.R[(size_t)(g.r_grp_rows > 0 ? gr % g.r_grp_rows : gr) * g.ldr + col];
      if (g.cs) v = v * g.cs[col] + g.csh[col];
      g.C[grp_off(gr, g.ldc, g.c_grp_rows, g.c_grp_stride) + col] = v;
    }
  });
}

// ------------------------------------------------------------------------------------------------
// Ancestral update, one workgroup per molecule (sampling.py:604-624; models/utils.py:38-45,67-106).
__global__ __launch_bounds__(256) void k_sampler_step(ds_layout L, float c_x, float c_pred, float sigma, float temp,
                                                      float* __restrict__ x, float* __restrict__ edge_x,
                                                      const float* __restrict__ pred, const float* __restrict__ edge_pred,
                                                      const float* __restrict__ raw_pos, const float* __restrict__ raw_feat,
                                                      const float* __restrict__ raw_edge, float* __restrict__ x_mean,
                                                      float* __restrict__ edge_mean) {
  __shared__ __attribute__((aligned(16))) float mean[3];
  __shared__ int dn[32];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0;
  if (n <= 0) return;
  if (tid < n) dn[tid] = L.node_dense[n0 + tid];
  __syncthreads();
  if (tid < 3) {
    float s = 0.0f;
    for (int a = 0; a < n; ++a) s += raw_pos[(size_t)dn[a] * 3 + tid];
    mean[tid] = s / (float)n;
  }
  __syncthreads();
  for (int idx = tid; idx < n * 9; idx += 256) {
    const int a = idx / 9, ch = idx - a * 9;
    const size_t d = (size_t)dn[a];
    const float nz = ch < 3 ? raw_pos[d * 3 + ch] - mean[ch] : raw_feat[d * 6 + (ch - 3)];
    const float xm = c_x * x[d * 9 + ch] + c_pred * pred[d * 9 + ch];
    x_mean[d * 9 + ch] = xm;
    x[d * 9 + ch] = xm + (sigma * nz) * temp;
  }
  const int N = L.N;
  for (int idx = tid; idx < n * n * 2; idx += 256) {
    const int ch = idx & 1, ij = idx >> 1;
    const int a = ij / n, b = ij - a * n;
    if (a == b) continue;
    const int la = dn[a] - m * N, lb = dn[b] - m * N;
    const int hi = la > lb ? la : lb, lo = la > lb ? lb : la;
    const float nz = raw_edge[(((size_t)m * 2 + ch) * N + hi) * N + lo];   // tril(-1) + transpose
    const size_t o = ((size_t)dn[a] * N + lb) * 2 + ch;
    const float em = c_x * edge_x[o] + c_pred * edge_pred[o];
    edge_mean[o] = em;
    edge_x[o] = em + (sigma * nz) * temp;
  }
}

// ---- in-kernel noise: Philox4x32-10 (Salmon et al., SC'11; the generator torch/curand use) + Box-Muller ----
struct Philox4 { unsigned int x, y, z, w; };
__device__ __forceinline__ Philox4 philox4x32_10(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3,
                                                 unsigned int k0, unsigned int k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned int hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned int hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const unsigned int n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return Philox4{c0, c1, c2, c3};
}
// u = (x + 0.5) / 2^32 in (0, 1]; (z0, z1) = sqrt(-2 ln u0) * (cos, sin)(2 pi u1)
__device__ __forceinline__ float2 box_muller(unsigned int a, unsigned int b) {
  const float u0 = __fmaf_rn((float)a, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  const float u1 = __fmaf_rn((float)b, 2.3283064365386963e-10f, 1.1641532182693481e-10f);
  const float r = sqrtf(-2.0f * logf(u0));
  const float th = 6.283185307179586f * u1;
  return make_float2(r * cosf(th), r * sinf(th));
}
__device__ __forceinline__ float4 philox_normal4(unsigned int elem, unsigned int draw, unsigned long long mol, unsigned int kind,
                                                 unsigned long long seed) {
  // counter words: (element, draw, mol_id low 32 bits, kind | mol_id high bits << 1)
  const Philox4 p = philox4x32_10(elem, draw, (unsigned int)mol, kind | ((unsigned int)(mol >> 32) << 1),
                                  (unsigned int)seed, (unsigned int)(seed >> 32));
  const float2 a = box_muller(p.x, p.y), b = box_muller(p.z, p.w);
  return make_float4(a.x, a.y, b.x, b.y);
}

// One workgroup per molecule.  MODE 0: initial noise (x, edge_x := noise; masked entries were zeroed by the caller's
// memset).  MODE 1: ancestral update with in-kernel noise (the Philox twin of k_sampler_step).
// MODE 2: as MODE 1 with (c_x, c_pred, sigma) and the draw index read from device memory (graph replay).
template <int MODE>
__global__ __launch_bounds__(256) void k_noise_step(ds_layout L, float c_x, float c_pred, float sigma, float temp,
                                                    unsigned long long seed, unsigned int draw, const int64_t* __restrict__ mol_id,
                                                    float* __restrict__ x, float* __restrict__ edge_x,
                                                    const float* __restrict__ pred, const float* __restrict__ edge_pred,
                                                    float* __restrict__ x_mean, float* __restrict__ edge_mean,
                                                    const float* __restrict__ table, const int32_t* __restrict__ step) {
  __shared__ __attribute__((aligned(16))) float nz[32][12];
  __shared__ float mean[3];
  __shared__ int dn[32];
  const int m = blockIdx.x, tid = threadIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0;
  if (n <= 0) return;
  if (MODE == 2) {
    const int i = *step;
    c_x = table[4 * i]; c_pred = table[4 * i + 1]; sigma = table[4 * i + 2];
    draw = (unsigned int)i + 1u;
  }
  const unsigned long long mol = (unsigned long long)mol_id[m];
  if (tid < n) dn[tid] = L.node_dense[n0 + tid];
  if (tid < n * 3) {
    const int a = tid / 3, j = tid - a * 3;
    const float4 v = philox_normal4((unsigned int)tid, draw, mol, 0u, seed);
    reinterpret_cast<float4*>(&nz[a][4 * j])[0] = v;
  }
  __syncthreads();
  if (tid < 3) {   // CoM projection of the position noise (models/utils.py:38-45,88-93), atoms in ascending order
    float s = 0.0f;
    for (int a = 0; a < n; ++a) s += nz[a][tid];
    mean[tid] = s / (float)n;
  }
  __syncthreads();
  for (int idx = tid; idx < n * 9; idx += 256) {
    const int a = idx / 9, ch = idx - a * 9;
    const size_t d = (size_t)dn[a];
    const float v = ch < 3 ? nz[a][ch] - mean[ch] : nz[a][ch];
    if (MODE == 0) {
      x[d * 9 + ch] = v;
    } else {   // MODE 1, 2
      const float xm = c_x * x[d * 9 + ch] + c_pred * pred[d * 9 + ch];
      x_mean[d * 9 + ch] = xm;
      x[d * 9 + ch] = xm + (sigma * v) * temp;
    }
  }
  const int N = L.N, P = n * (n - 1) / 2;
  for (int p = tid; p < P; p += 256) {   // unordered pair lo < hi: p = hi(hi-1)/2 + lo, independent of n and of padding
    int hi = (int)((1.0f + sqrtf(1.0f + 8.0f * (float)p)) * 0.5f);
    while (hi * (hi - 1) / 2 > p) --hi;
    while ((hi + 1) * hi / 2 <= p) ++hi;
    const int lo = p - hi * (hi - 1) / 2;
    const float4 v = philox_normal4((unsigned int)p, draw, mol, 1u, seed);
    const int la = dn[lo] - m * N, lb = dn[hi] - m * N;
    const size_t o1 = ((size_t)dn[lo] * N + lb) * 2, o2 = ((size_t)dn[hi] * N + la) * 2;
#pragma unroll
    for (int ch = 0; ch < 2; ++ch) {
      const float nzv = ch ? v.y : v.x;
      if (MODE == 0) {
        edge_x[o1 + ch] = nzv; edge_x[o2 + ch] = nzv;
      } else {
        const float em1 = c_x * edge_x[o1 + ch] + c_pred * edge_pred[o1 + ch];
        const float em2 = c_x * edge_x[o2 + ch] + c_pred * edge_pred[o2 + ch];
        edge_mean[o1 + ch] = em1; edge_mean[o2 + ch] = em2;
        edge_x[o1 + ch] = em1 + (sigma * nzv) * temp; edge_x[o2 + ch] = em2 + (sigma * nzv) * temp;
      }
    }
  }
}

// Opens a graph-replayable denoise iteration: ++*step, noise_level[b] = table[*step][3] (one workgroup).
__global__ __launch_bounds__(256) void k_step_begin(const float* __restrict__ table, int n_steps, int32_t* __restrict__ step, int B,
                                                    float* __restrict__ noise_level) {
  __shared__ int cur;
  if (threadIdx.x == 0) {
    const int i = min(*step + 1, n_steps - 1);
    *step = i;
    cur = i;
  }
  __syncthreads();
  const float nl = table[4 * cur + 3];
  for (int b = threadIdx.x; b < B; b += 256) noise_level[b] = nl;
}

// post_process (sampling.py:53-97) with the inverse scaler of utils.py:88-103 (norms 1,4,4,1; centered).
__global__ void k_post_process(ds_layout L, const float* __restrict__ xh, const float* __restrict__ edge_x,
                               float* __restrict__ pos_out, int32_t* __restrict__ atom_type, int32_t* __restrict__ fc,
                               float* __restrict__ edge_type) {
  const int m = blockIdx.x, tid = threadIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0;
  const int N = L.N;
  for (int a = tid; a < n; a += blockDim.x) {
    const size_t d = (size_t)L.node_dense[n0 + a];
    const float* r = xh + d * 9;
    pos_out[d * 3 + 0] = r[0] * 1.0f; pos_out[d * 3 + 1] = r[1] * 1.0f; pos_out[d * 3 + 2] = r[2] * 1.0f;
    int best = 0;
    float bv = (r[3] * 4.0f + 1.0f) / 2.0f;
    for (int t = 1; t < 5; ++t) {
      const float v = (r[3 + t] * 4.0f + 1.0f) / 2.0f;
      if (v > bv) { bv = v; best = t; }
    }
    atom_type[d] = best;
    fc[d] = (int32_t)rintf(r[8] * 4.0f);
  }
  for (int idx = tid; idx < n * n; idx += blockDim.x) {
    const int a = idx / n, b = idx - a * n;
    if (a == b) continue;
    const int da = L.node_dense[n0 + a], lb = L.node_dense[n0 + b] - m * N;
    const size_t o = (size_t)da * N + lb;
    const float ex = (edge_x[o * 2 + 0] * 1.0f + 1.0f) / 2.0f;
    const float t = ((edge_x[o * 2 + 1] * 1.0f + 1.0f) / 2.0f) * 3.0f;
    float et = 0.0f;
    if (t >= 2.5f) et = 3.0f; else if (t >= 1.5f) et = 2.0f; else if (t >= 0.5f) et = 1.0f;
    edge_type[o] = (ex >= 0.5f ? 1.0f : 0.0f) * et;
  }
}

// Stability check, one workgroup per molecule (evaluation/stability.py:40-73; tables of evaluation/bond_analyze.py:5-45 for
// H, C, N, O, F; 0 = no such bond).  Thread a walks the other atoms of its molecule.
__constant__ int c_bond1[5][5] = {{74, 109, 101, 96, 92}, {109, 154, 147, 143, 135}, {101, 147, 145, 140, 136},
                                  {96, 143, 140, 148, 142}, {92, 135, 136, 142, 142}};
__constant__ int c_bond2[5][5] = {{0, 0, 0, 0, 0}, {0, 134, 129, 120, 0}, {0, 129, 125, 121, 0}, {0, 120, 121, 121, 0}, {0, 0, 0, 0, 0}};
__constant__ int c_bond3[5][5] = {{0, 0, 0, 0, 0}, {0, 120, 116, 113, 0}, {0, 116, 110, 0, 0}, {0, 113, 0, 0, 0}, {0, 0, 0, 0, 0}};
__constant__ int c_valence[5] = {1, 4, 3, 2, 1};
__global__ __launch_bounds__(64) void k_check_stability(ds_layout L, const float* __restrict__ pos, const int32_t* __restrict__ atom_type,
                                                        int32_t* __restrict__ bond_order, int32_t* __restrict__ nr_stable,
                                                        int32_t* __restrict__ mol_stable) {
  __shared__ float px[32], py[32], pz[32];
  __shared__ int ty[32], dn[32];
  const int m = blockIdx.x, a = threadIdx.x;
  const int n0 = L.node_off[m], n = L.node_off[m + 1] - n0;
  if (n <= 0) { if (a == 0) { nr_stable[m] = 0; mol_stable[m] = 1; } return; }
  if (a < n) {
    const int d = L.node_dense[n0 + a];
    dn[a] = d;
    px[a] = pos[(size_t)d * 3]; py[a] = pos[(size_t)d * 3 + 1]; pz[a] = pos[(size_t)d * 3 + 2];
    ty[a] = min(max(atom_type[d], 0), 4);
  }
  __syncthreads();
  int ok = 0;
  if (a < n) {
    int bonds = 0;
    const int ta = ty[a], N = L.N;
    for (int b = 0; b < n; ++b) {
      if (b == a) continue;
      const float dx = px[a] - px[b], dy = py[a] - py[b], dz = pz[a] - pz[b];
      const float d = __fmul_rn(__fsqrt_rn(__fadd_rn(__fadd_rn(__fmul_rn(dx, dx), __fmul_rn(dy, dy)), __fmul_rn(dz, dz))), 100.0f);
      const int tb = ty[b];
      int order = 0;
      if (d < (float)(c_bond1[ta][tb] + 10)) {
        order = 1;
        if (c_bond2[ta][tb] != 0 && d < (float)(c_bond2[ta][tb] + 5)) {
          order = 2;
          if (c_bond3[ta][tb] != 0 && d < (float)(c_bond3[ta][tb] + 3)) order = 3;
        }
      }
      bonds += order;
      if (bond_order) bond_order[(size_t)dn[a] * N + (dn[b] - m * N)] = order;
    }
    ok = bonds == c_valence[ta] ? 1 : 0;
  }
  const int cnt = __popcll(__ballot(ok != 0));
  if (a == 0) { nr_stable[m] = cnt; mol_stable[m] = cnt == n ? 1 : 0; }
}

// SpecFormer residual-score attention (specformer.py:401-424): one workgroup per (molecule, head) and up to 1024 queries - one
// query per thread, K / V of the head staged in LDS once for all of them.
// qkv [B, L, 3*heads*dk] (q | k | v); scores [B, heads, L, L] holds prev on entry (if has_prev) and the new
// pre-softmax scores on exit; out [B, L, heads*dk].
__global__ __launch_bounds__(1024) void k_spec_attention(const float* __restrict__ qkv, float* __restrict__ scores,
                                                       float* __restrict__ out, int B, int L, int heads, float scale,
                                                       int has_prev) {
  constexpr int DK = 8;
  extern __shared__ __attribute__((aligned(16))) float kv[];   // K [L][8] then V [L][8]
  const int b = blockIdx.z, h = blockIdx.y, q0 = blockIdx.x * blockDim.x, tid = threadIdx.x;
  const int D = heads * DK;
  float* Ks = kv;
  float* Vs = kv + (size_t)L * DK;
  for (int idx = tid; idx < L * DK; idx += blockDim.x) {
    const int j = idx / DK, d = idx - j * DK;
    const float* base = qkv + ((size_t)b * L + j) * 3 * D + h * DK + d;
    Ks[idx] = base[D];
    Vs[idx] = base[2 * D];
  }
  __syncthreads();
  const int i = q0 + tid;
  if (i >= L) return;
  float q[DK];
  for (int d = 0; d < DK; ++d) q[d] = qkv[((size_t)b * L + i) * 3 * D + h * DK + d];
  // residual scores are kept TRANSPOSED, [b][h][key j][query i]: the lanes of a wave are consecutive queries, so every
  // access below is one contiguous 256-byte segment (query-major rows made each lane touch its own cache line)
  float* scol = scores + ((size_t)b * heads + h) * L * L + i;
  // One pass with a running maximum (the scores are written for the next layer and never read back here: the [B, heads, L, L]
  // tensor is this kernel's whole HBM bill - 4.1 GB per launch with the two-pass form).  Blocks of 8 keys: one rescale per block.
  float mx = -INFINITY, den = 0.0f, o[DK];
  for (int d = 0; d < DK; ++d) o[d] = 0.0f;
  for (int j0 = 0; j0 < L; j0 += 8) {
    float sv[8];
    float bm = -INFINITY;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int j = min(j0 + u, L - 1);
      float s = 0.0f;
#pragma unroll
      for (int d = 0; d < DK; ++d) s += q[d] * Ks[j * DK + d];
      s *= scale;
      if (has_prev) s += scol[(size_t)j * L];
      sv[u] = s;
      if (j0 + u < L) { scol[(size_t)j * L] = s; bm = fmaxf(bm, s); }
    }
    const float nm = fmaxf(mx, bm), r = expf(mx - nm);   // mx = -inf on the first block: r = 0
    den *= r;
#pragma unroll
    for (int d = 0; d < DK; ++d) o[d] *= r;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (j0 + u < L) {
        const float p = expf(sv[u] - nm);
        den += p;
#pragma unroll
        for (int d = 0; d < DK; ++d) o[d] += p * Vs[(j0 + u) * DK + d];
      }
    }
    mx = nm;
  }
  for (int d = 0; d < DK; ++d) out[((size_t)b * L + i) * D + h * DK + d] = o[d] / den;
}

__global__ void k_layernorm_affine(const float* __restrict__ x, const float* __restrict__ g, const float* __restrict__ bt,
                                   float* __restrict__ y, int rows, int cols, float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;   // one wave per row
  if (row >= rows) return;
  const float* xr = x + (size_t)row * cols;
  float s = 0.0f;
  for (int k = lane; k < cols; k += 64) s += xr[k];
  const float mean = wave_sum(s) / (float)cols;
  float v = 0.0f;
  for (int k = lane; k < cols; k += 64) { const float d = xr[k] - mean; v += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(v) / (float)cols + eps);
  for (int k = lane; k < cols; k += 64) y[(size_t)row * cols + k] = (xr[k] - mean) * rstd * g[k] + bt[k];
}

// ------------------------------------------------------------------------------------------------ host side
bool make_ctx(Ctx& c, const ds_weights* w, const ds_layout* L, const ds_workspace* ws, hipStream_t s) {
  if (!w || !L || !ws || !w->base) return false;
  if (L->max_n > DS_MAX_ATOMS || L->B <= 0 || !w->off_dev) return false;
  (void)s;
  c.L = *L; c.ws = *ws; c.wbase = w->base;
  c.woff = w->off_dev;
  c.edge_th = w->edge_th; c.cutoff = w->spatial_cut_off;
  return c.woff != nullptr;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? DS_OK : DS_ERR_LAUNCH; }

// Compute units of the current device (256 on a full MI355X; fewer in a partitioned mode): the persistent
// k_equi_pairs launches exactly one workgroup per CU.
inline int device_cus() {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      cus = n;
    else
      cus = 256;
  }
  return cus;
}

// Large plain GEMMs (the per-step adaLN table [B,1024] x [1024,19744] is 6 % of a denoising step): 128x128 output tile,
// MT = 4 (each B fragment feeds 16 MFMAs), A double-buffered in LDS with the next K-chunk fetched into registers while
// the current one is multiplied (one barrier per chunk), the next chunk's first B group requested ahead of that barrier.
// Requires contiguous rows (no row groups), K % 64 == 0 and 16-byte aligned rows.
template <int ACT>
__global__ __launch_bounds__(256, 2) void k_gemm_big(GemmArgs g) {
  constexpr int T = 128, KC = 64, LD = KC + DS_LDP;
  __shared__ __attribute__((aligned(16))) float X[2][T][LD];
  const int tid = threadIdx.x, wave = tid >> 6;
  const int row0 = blockIdx.x * T;
  const int col0 = (blockIdx.y * 4 + wave) * 32;
  const bool active = col0 < g.Npad;
  const int nchunks = g.K / KC;
  float4 st[8];
  auto fetch = [&](int kc) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * 256, row = idx >> 4, k4 = idx & 15;
      const size_t gr = (size_t)min(row0 + row, g.M - 1);
      st[u] = reinterpret_cast<const float4*>(g.A + gr * g.lda + (size_t)kc * KC)[k4];
    }
  };
  auto stash = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = tid + u * 256, row = idx >> 4, k4 = idx & 15;
      float4 v = st[u];
      if (g.a_silu) { v.x = ds_silu(v.x); v.y = ds_silu(v.y); v.z = ds_silu(v.z); v.w = ds_silu(v.w); }
      if (row0 + row >= g.M) v = make_float4(0, 0, 0, 0);
      reinterpret_cast<float4*>(&X[buf][row][0])[k4] = v;
    }
  };
  f32x16 acc[4];
  acc_zero<4>(acc);
  fetch(0);
  stash(0);
  BFrag bf = bfrag_load(g.Wp, g.Npad, active ? col0 : 0, 0, 8);
  __syncthreads();
  for (int kc = 0; kc < nchunks; ++kc) {
    const int cur = kc & 1;
    if (kc + 1 < nchunks) fetch(kc + 1);
    const float* wp = g.Wp + (size_t)(kc * (KC / 8)) * 2 * g.Npad * 4;
    if (active) wave_mma<4>(&X[cur][0][0], LD, wp, g.Npad, col0, 0, KC / 8, acc, 0, &bf);
    if (kc + 1 < nchunks) {
      stash(cur ^ 1);
      bf = bfrag_load(wp + (size_t)(KC / 8) * 2 * g.Npad * 4, g.Npad, active ? col0 : 0, 0, 8);
    }
    __syncthreads();
  }
  if (!active) return;
  {   // epilogue: the lane's column constants are fetched once, rows go out as buffer stores with SGPR row offsets
    const int lane = tid & 63, r = lane & 31, hh = lane >> 5, col = col0 + r;
    const bool colok = col < g.N;
    const float bcol = (g.bias && colok) ? g.bias[col] : 0.0f;
    const float csc = (g.cs && colok) ? g.cs[col] : 1.0f, csh = (g.cs && colok) ? g.csh[col] : 0.0f;
    const unsigned long long pw = reinterpret_cast<unsigned long long>(g.C + (size_t)row0 * g.ldc + col0);
    const unsigned long long pu = (static_cast<unsigned long long>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw >> 32))) << 32) |
                                  static_cast<unsigned int>(__builtin_amdgcn_readfirstlane(static_cast<int>(pw)));
    const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<float*>(pu), 0, 0x7fffffff, 0x00020000);
    const int voff = (4 * hh * g.ldc + r) * 4, rowb = g.ldc * 4;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int row = m * 32 + (i & 3) + 8 * (i >> 2), gr = row0 + row + 4 * hh;
        if (gr < g.M && colok) {
          float v = ds_act<ACT>(acc[m][i] + bcol);
          if (g.R) v += g.R[(size_t)(g.r_grp_rows > 0 ? gr % g.r_grp_rows : gr) * g.ldr + col];
          if (g.cs) v = v * csc + csh;
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rc, voff, row * rowb, 0);
        }
      }
  }
}

int gemm_dispatch(const GemmArgs& g, int act, hipStream_t s) {
  if (!g.A || !g.Wp || !g.C || g.M <= 0 || g.K <= 0 || g.N <= 0 || (g.cs && !g.csh)) return DS_ERR_ARG;
  const bool big = g.M >= 512 && g.a_grp_rows <= 0 && g.c_grp_rows <= 0 && g.K % 64 == 0 && g.lda % 4 == 0 &&
                   (reinterpret_cast<uintptr_t>(g.A) & 15) == 0;
  if (big) {
    dim3 gridb((g.M + 127) / 128, (g.Npad + 127) / 128);
    switch (act) {
      case 0: hipLaunchKernelGGL(k_gemm_big<0>, gridb, dim3(256), 0, s, g); break;
      case 1: hipLaunchKernelGGL(k_gemm_big<1>, gridb, dim3(256), 0, s, g); break;
      case 2: hipLaunchKernelGGL(k_gemm_big<2>, gridb, dim3(256), 0, s, g); break;
      case 3: hipLaunchKernelGGL(k_gemm_big<3>, gridb, dim3(256), 0, s, g); break;
      default: return DS_ERR_ARG;
    }
    return launch_status();
  }
  dim3 grid((g.M + 63) / 64, (g.Npad + 127) / 128);
  switch (act) {
    case 0: hipLaunchKernelGGL(k_gemm<0>, grid, dim3(256), 0, s, g); break;
    case 1: hipLaunchKernelGGL(k_gemm<1>, grid, dim3(256), 0, s, g); break;
    case 2: hipLaunchKernelGGL(k_gemm<2>, grid, dim3(256), 0, s, g); break;
    case 3: hipLaunchKernelGGL(k_gemm<3>, grid, dim3(256), 0, s, g); break;
    default: return DS_ERR_ARG;
  }
  return launch_status();
}

int gemm_simple(const float* A, int64_t lda, const float* Wp, const float* bias, float* C, int64_t ldc, int M, int K, int N,
                int act, int a_silu, hipStream_t s) {
  GemmArgs g{};
  g.A = A; g.lda = lda; g.Wp = Wp; g.bias = bias; g.C = C; g.ldc = ldc; g.M = M; g.K = K; g.N = N;
  g.Npad = (N + 31) & ~31; g.a_silu = a_silu;
  return gemm_dispatch(g, act, s);
}

// ---- optional HIP-event timing of one block-stage kernel (bench.py's live roofline measurement) ----
struct ProfState {
  int kernel = -1;          // 0 edge_geom, 1 node_qkv, 2 attn_logits, 3 node_update, 4 edge_update, 5 equi_pairs, 6 attn_agg
  int every = 1;
  long long seen = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pool;
  size_t used = 0;
};
ProfState g_prof;

struct ProfScope {   // records start/stop events around one launch when sampling says so
  hipStream_t s;
  bool on = false;
  ProfScope(int kernel, hipStream_t st) : s(st) {
    if (g_prof.kernel != kernel) return;
    if ((g_prof.seen++ % g_prof.every) != 0 || g_prof.used >= g_prof.pool.size()) return;
    on = true;
    (void)hipEventRecord(g_prof.pool[g_prof.used].first, s);
  }
  ~ProfScope() {
    if (on) (void)hipEventRecord(g_prof.pool[g_prof.used++].second, s);
  }
};

}  // namespace

extern "C" {

void ds_struct_sizes(int64_t* out) {
  out[0] = sizeof(ds_weights);
  out[1] = sizeof(ds_layout);
  out[2] = sizeof(ds_workspace);
  out[3] = sizeof(ds_gemm_args);
}

int ds_gemm(const ds_gemm_args* a, void* stream) {
  if (!a) return DS_ERR_ARG;
  GemmArgs g{};
  g.A = a->A; g.lda = a->lda; g.a_grp_rows = a->a_grp_rows; g.a_grp_stride = a->a_grp_stride;
  g.Wp = a->Wp; g.bias = a->bias;
  g.C = a->C; g.ldc = a->ldc; g.c_grp_rows = a->c_grp_rows; g.c_grp_stride = a->c_grp_stride;
  g.M = a->M; g.K = a->K; g.N = a->N; g.Npad = (a->N + 31) & ~31;
  g.R = a->R; g.ldr = a->ldr; g.r_grp_rows = a->r_grp_rows;
  g.cs = a->col_scale; g.csh = a->col_shift; g.a_silu = a->a_silu;
  return gemm_dispatch(g, a->act, (hipStream_t)stream);
}

int ds_gemm_split(const void* A_split, const float* W_split, const float* bias, float* C, int64_t ldc, int32_t M, int32_t K,
                  int32_t N, void* stream) {
  if (!A_split || !W_split || !C || M <= 0 || K <= 0 || N <= 0 || K % 64 != 0 || N % 32 != 0 || ldc < N) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_gemm_ada, dim3((M + 63) / 64, (N + 127) / 128), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const _Float16*>(A_split), W_split, bias, C, (int)ldc, M, K, N);
  return launch_status();
}

int ds_stage_time(const ds_weights* w, const ds_layout* L, ds_workspace* ws, const float* noise_level, const float* ctx_emb,
                  void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Ctx c;
  if (!make_ctx(c, w, L, ws, s) || !noise_level) return DS_ERR_ARG;
  const int B = L->B;
  const int64_t* off = w->off + DS_NBLOCKS * DS_W_BLOCK_SLOTS;
  hipLaunchKernelGGL(k_time_feat, dim3((B + 63) / 64), dim3(64), 0, s, c, noise_level);
  // time_mlp: Linear(17,1024) -> GELU -> Linear(1024,1024)  (dmt.py:252-257); tmid reuses ws->tmid, output in ws->ada scratch
  int st = gemm_simple(ws->tfeat, 24, w->base + off[DS_GW_TM1_W], w->base + off[DS_GW_TM1_B], ws->tmid, 1024, B, 24, 1024, 2, 0, s);
  if (st) return st;
  float* tm3 = ws->ada;   // [B,1024] scratch inside the (larger) ada buffer, consumed before ada is produced
  st = gemm_simple(ws->tmid, 1024, w->base + off[DS_GW_TM3_W], w->base + off[DS_GW_TM3_B], tm3, 1024, B, 1024, 1024, 0, 0, s);
  if (st) return st;
  const size_t tot = (size_t)B * 1024;
  hipLaunchKernelGGL(k_temb_finish, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, s, c, (const float*)tm3, B, ctx_emb);
  // every *time_mlp Linear of the model in one GEMM: [B,1024] x [1024, DS_ADA_COLS]
  static_assert(ADAC % 32 == 0, "adaLN table width");
  hipLaunchKernelGGL(k_gemm_ada, dim3((B + 63) / 64, (ADAC + 127) / 128), dim3(256), 0, s, reinterpret_cast<const _Float16*>(ws->temb_silu),
                     w->base + off[DS_GW_ADA_W], w->base + off[DS_GW_ADA_B], ws->ada, (int)ADAC, B, 1024, (int)ADAC);
  return launch_status();
}

int ds_stage_init(const ds_weights* w, const ds_layout* L, ds_workspace* ws, const float* xh, const float* edge_x,
                  const float* cond_x, const float* cond_edge_x, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Ctx c;
  if (!make_ctx(c, w, L, ws, s) || !xh || !edge_x || ((cond_x == nullptr) != (cond_edge_x == nullptr))) return DS_ERR_ARG;
  if (hipMemsetAsync(ws->flags, 0, 8 * sizeof(int32_t), s) != hipSuccess) return DS_ERR_LAUNCH;
  if (L->Pp > 0) hipLaunchKernelGGL(k_pair_flags, dim3((L->Pp + 1023) / 1024), dim3(1024), 0, s, c, cond_x, cond_edge_x);
  hipLaunchKernelGGL(k_node_init, dim3((L->Nn + 7) / 8), dim3(256), 0, s, c, xh, cond_x);
  if (L->Pp > 0) hipLaunchKernelGGL(k_pair_init, dim3((L->Pp + 63) / 64), dim3(256), 0, s, c, edge_x, cond_x, cond_edge_x);
  return launch_status();
}

int ds_stage_block(const ds_weights* w, const ds_layout* L, ds_workspace* ws, int blk, int last, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Ctx c;
  if (!make_ctx(c, w, L, ws, s) || blk < 0 || blk >= DS_NBLOCKS) return DS_ERR_ARG;
  const int pt = (L->Pp + 63) / 64, nt = (L->Nn + 31) / 32;
  if (pt > 0) { ProfScope ps(0, s); hipLaunchKernelGGL(k_edge_geom, dim3(pt), dim3(256), 0, s, c, blk); }
  { ProfScope ps(1, s); hipLaunchKernelGGL(k_node_qkv<4>, dim3((L->Nn + 63) / 64, 2), dim3(256), 0, s, c, blk); }
  { ProfScope ps(2, s); hipLaunchKernelGGL(k_attn_fused, dim3(L->B), dim3(1024), 0, s, c, blk); }
  { ProfScope ps(3, s); hipLaunchKernelGGL(k_node_update<true>, dim3(nt), dim3(256), 0, s, c, blk); }
  if (pt > 0) { ProfScope ps(4, s); hipLaunchKernelGGL(k_edge_update, dim3((L->Pp + 127) / 128), dim3(256), 0, s, c, blk); }
  if (pt > 0) { ProfScope ps(5, s); { const int nt_ = (L->Pp + 31) / 32, cu_ = device_cus(); hipLaunchKernelGGL((k_equi_pairs<DS_EQUI_NCW, DS_EQUI_NLW>), dim3(nt_ < cu_ ? nt_ : cu_), dim3((DS_EQUI_NCW + DS_EQUI_NLW) * 64), 0, s, c, blk); } }
  hipLaunchKernelGGL(k_pos_update, dim3(L->B), dim3(64), 0, s, c, last);
  return launch_status();
}

int ds_stage_readout(const ds_weights* w, const ds_layout* L, ds_workspace* ws, float* out_xh, float* out_edge, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  Ctx c;
  if (!make_ctx(c, w, L, ws, s) || !out_xh || !out_edge) return DS_ERR_ARG;
  const size_t nx = (size_t)L->B * L->N * 9, ne = (size_t)L->B * L->N * L->N * 2;
  if (hipMemsetAsync(out_xh, 0, nx * sizeof(float), s) != hipSuccess) return DS_ERR_LAUNCH;
  if (hipMemsetAsync(out_edge, 0, ne * sizeof(float), s) != hipSuccess) return DS_ERR_LAUNCH;
  hipLaunchKernelGGL(k_node_readout, dim3((L->Nn + 31) / 32), dim3(256), 0, s, c, out_xh);
  if (L->Pp > 0) hipLaunchKernelGGL(k_edge_readout, dim3((L->Pp + 127) / 128), dim3(256), 0, s, c, out_edge);
  hipLaunchKernelGGL(k_final_pos, dim3((L->B + 63) / 64), dim3(64), 0, s, c, out_xh);
  return launch_status();
}

// The blocks of one forward over TWO streams (ds_forward).  A block's node rows after the attention - gated residual, FF, read-out
// slice, the node parts of input_lin (k_node_update<false>) - and the next block's q|k|v projection depend on nothing the pair
// rows compute in k_edge_update, and the two sides load different parts of a CU (k_edge_update: HBM streams and waits, matrix pipe
// 24 % busy; k_node_update / k_node_qkv: L2 weight streams + MFMA), so they run BESIDE k_edge_update on a side stream instead of in
// front of it / behind it:
//   main : edge_geom(b) -> [q|k|v(b)] attn(b) -> n2e(b) -> edge_update(b) -> [side(b)] equi_pairs(b) -> pos_update(b)
//   side :                                        [n2e(b)] node_update(b) -> q|k|v(b + 1)
// k_equi_pairs waits for the side stream (it needs `ac`, and as a persistent one-workgroup-per-CU kernel it must not find CUs taken).
// One workgroup of either side fits beside one of the other in a CU's LDS (66.5 + 71 kB).  Results do not depend on the mode: the
// same kernels' arithmetic, ws.u from k_node_n2e bit-identical to the fused form (tests: stage API = one stream, forward = two).
struct SideStream {
  hipStream_t s = nullptr;
  hipEvent_t fork[DS_NBLOCKS + 1] = {}, join[DS_NBLOCKS + 1] = {};
  bool ok = false, tried = false;
};
static SideStream g_side[16];
static int g_two_stream = -1;   // -1: from the environment (DIFFSPECTRA_TWO_STREAM), else by batch size

static SideStream* side_stream() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  SideStream& ss = g_side[dev];
  if (!ss.tried) {
    ss.tried = true;
    bool ok = hipStreamCreateWithFlags(&ss.s, hipStreamNonBlocking) == hipSuccess;
    for (int i = 0; ok && i <= DS_NBLOCKS; ++i)
      ok = hipEventCreateWithFlags(&ss.fork[i], hipEventDisableTiming) == hipSuccess &&
           hipEventCreateWithFlags(&ss.join[i], hipEventDisableTiming) == hipSuccess;
    ss.ok = ok;
  }
  return ss.ok ? &ss : nullptr;
}

static int forward_blocks_two_streams(const ds_weights* w, const ds_layout* L, ds_workspace* ws, hipStream_t s, SideStream* ss) {
  Ctx c;
  if (!make_ctx(c, w, L, ws, s)) return DS_ERR_ARG;
  const int pt = (L->Pp + 63) / 64, nt = (L->Nn + 31) / 32;
  hipStream_t s2 = ss->s;
#define DS_HIP_OK(x) do { if ((x) != hipSuccess) return DS_ERR_LAUNCH; } while (0)
  // q|k|v of block 0 on the side stream, beside edge_geom(0)
  DS_HIP_OK(hipEventRecord(ss->fork[DS_NBLOCKS], s));
  DS_HIP_OK(hipStreamWaitEvent(s2, ss->fork[DS_NBLOCKS], 0));
  { ProfScope ps(1, s2); hipLaunchKernelGGL(k_node_qkv<4>, dim3((L->Nn + 63) / 64, 2), dim3(256), 0, s2, c, 0); }
  DS_HIP_OK(hipEventRecord(ss->join[DS_NBLOCKS], s2));
  for (int blk = 0; blk < DS_NBLOCKS; ++blk) {
    if (pt > 0) { ProfScope ps(0, s); hipLaunchKernelGGL(k_edge_geom, dim3(pt), dim3(256), 0, s, c, blk); }
    DS_HIP_OK(hipStreamWaitEvent(s, blk == 0 ? ss->join[DS_NBLOCKS] : ss->join[blk - 1], 0));     // q|k|v(blk)
    { ProfScope ps(2, s); hipLaunchKernelGGL(k_attn_fused, dim3(L->B), dim3(1024), 0, s, c, blk); }
    hipLaunchKernelGGL(k_node_n2e, dim3((L->Nn + 63) / 64), dim3(256), 0, s, c, blk);
    DS_HIP_OK(hipEventRecord(ss->fork[blk], s));
    DS_HIP_OK(hipStreamWaitEvent(s2, ss->fork[blk], 0));
    { ProfScope ps(3, s2); hipLaunchKernelGGL(k_node_update<false>, dim3(nt), dim3(256), 0, s2, c, blk); }
    if (blk + 1 < DS_NBLOCKS) { ProfScope ps(1, s2); hipLaunchKernelGGL(k_node_qkv<4>, dim3((L->Nn + 63) / 64, 2), dim3(256), 0, s2, c, blk + 1); }
    DS_HIP_OK(hipEventRecord(ss->join[blk], s2));
    if (pt > 0) { ProfScope ps(4, s); hipLaunchKernelGGL(k_edge_update, dim3((L->Pp + 127) / 128), dim3(256), 0, s, c, blk); }
    DS_HIP_OK(hipStreamWaitEvent(s, ss->join[blk], 0));                                           // ac, h, atom_hids, q|k|v(blk + 1)
    if (pt > 0) { ProfScope ps(5, s); { const int nt_ = (L->Pp + 31) / 32, cu_ = device_cus(); hipLaunchKernelGGL((k_equi_pairs<DS_EQUI_NCW, DS_EQUI_NLW>), dim3(nt_ < cu_ ? nt_ : cu_), dim3((DS_EQUI_NCW + DS_EQUI_NLW) * 64), 0, s, c, blk); } }
    hipLaunchKernelGGL(k_pos_update, dim3(L->B), dim3(64), 0, s, c, blk == DS_NBLOCKS - 1 ? 1 : 0);
  }
#undef DS_HIP_OK
  return launch_status();
}

int ds_set_two_stream(int on) {   // -1: follow DIFFSPECTRA_TWO_STREAM (default on); 0 / 1: force.  Returns the previous setting.
  const int prev = g_two_stream;
  g_two_stream = on;
  return prev;
}

int ds_forward(const ds_weights* w, const ds_layout* L, ds_workspace* ws, const float* xh, const float* edge_x,
               const float* cond_x, const float* cond_edge_x, const float* noise_level, const float* ctx_emb, float* out_xh,
               float* out_edge, void* stream) {
  int st = ds_stage_time(w, L, ws, noise_level, ctx_emb, stream);
  if (st) return st;
  st = ds_stage_init(w, L, ws, xh, edge_x, cond_x, cond_edge_x, stream);
  if (st) return st;
  // measured (profiles/r05_two_stream_ab.txt): +0.8 % at 1 250 resident molecules (the side kernels fill the tails of the main
  // stream's launches), -1 % at 5 000 (every kernel fills the chip by itself and is bound by the latency of its own waves at an
  // LDS-limited occupancy; workgroups of a second kernel take slots, they do not add any) - so it is on for small batches only
  static const int env_two = [] { const char* e = getenv("DIFFSPECTRA_TWO_STREAM"); return e ? atoi(e) : -1; }();
  const int pick = g_two_stream >= 0 ? g_two_stream : env_two;
  const int two = pick >= 0 ? pick : (L->Pp < DS_TWO_STREAM_MAX_PAIRS ? 1 : 0);
  SideStream* ss = two ? side_stream() : nullptr;
  if (ss) {
    st = forward_blocks_two_streams(w, L, ws, (hipStream_t)stream, ss);
    if (st) return st;
  } else {
    for (int b = 0; b < DS_NBLOCKS; ++b) {
      st = ds_stage_block(w, L, ws, b, b == DS_NBLOCKS - 1, stream);
      if (st) return st;
    }
  }
  return ds_stage_readout(w, L, ws, out_xh, out_edge, stream);
}

int ds_sampler_step(const ds_layout* L, float c_x, float c_pred, float sigma, float temperature, float* x, float* edge_x,
                     const float* pred, const float* edge_pred, const float* raw_pos, const float* raw_feat,
                     const float* raw_edge, float* x_mean, float* edge_mean, void* stream) {
  if (!L || !x || !edge_x || !pred || !edge_pred || !raw_pos || !raw_feat || !raw_edge || !x_mean || !edge_mean) return DS_ERR_ARG;
  if (L->B <= 0 || L->max_n > DS_MAX_ATOMS || L->max_n > L->N) return DS_ERR_ARG;   // k_sampler_step stages <= 32 node indices in LDS
  hipLaunchKernelGGL(k_sampler_step, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, c_x, c_pred, sigma, temperature, x,
                     edge_x, pred, edge_pred, raw_pos, raw_feat, raw_edge, x_mean, edge_mean);
  return launch_status();
}

int ds_initial_noise(const ds_layout* L, uint64_t seed, const int64_t* mol_id, float* x, float* edge_x, void* stream) {
  if (!L || !mol_id || !x || !edge_x) return DS_ERR_ARG;
  if (L->B <= 0 || L->max_n > DS_MAX_ATOMS || L->max_n > L->N) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t nb = (size_t)L->B * L->N;
  if (hipMemsetAsync(x, 0, nb * 9 * sizeof(float), s) != hipSuccess) return DS_ERR_LAUNCH;
  if (hipMemsetAsync(edge_x, 0, nb * L->N * 2 * sizeof(float), s) != hipSuccess) return DS_ERR_LAUNCH;
  hipLaunchKernelGGL(k_noise_step<0>, dim3(L->B), dim3(256), 0, s, *L, 0.0f, 0.0f, 0.0f, 0.0f, (unsigned long long)seed, 0u, mol_id,
                     x, edge_x, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr,
                     (const float*)nullptr, (const int32_t*)nullptr);
  return launch_status();
}

int ds_sampler_step_philox(const ds_layout* L, float c_x, float c_pred, float sigma, float temperature, uint64_t seed, int32_t step,
                           const int64_t* mol_id, float* x, float* edge_x, const float* pred, const float* edge_pred,
                           float* x_mean, float* edge_mean, void* stream) {
  if (!L || !mol_id || !x || !edge_x || !pred || !edge_pred || !x_mean || !edge_mean || step < 0) return DS_ERR_ARG;
  if (L->B <= 0 || L->max_n > DS_MAX_ATOMS || L->max_n > L->N) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_noise_step<1>, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, c_x, c_pred, sigma, temperature,
                     (unsigned long long)seed, (unsigned int)step + 1u, mol_id, x, edge_x, pred, edge_pred, x_mean, edge_mean,
                     (const float*)nullptr, (const int32_t*)nullptr);
  return launch_status();
}

int ds_step_begin(const float* table, int32_t n_steps, int32_t* step, int32_t B, float* noise_level, void* stream) {
  if (!table || !step || !noise_level || B <= 0 || n_steps <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(256), 0, (hipStream_t)stream, table, n_steps, step, B, noise_level);
  return launch_status();
}

int ds_sampler_step_philox_dev(const ds_layout* L, const float* table, const int32_t* step, float temperature, uint64_t seed,
                               const int64_t* mol_id, float* x, float* edge_x, const float* pred, const float* edge_pred,
                               float* x_mean, float* edge_mean, void* stream) {
  if (!L || !table || !step || !mol_id || !x || !edge_x || !pred || !edge_pred || !x_mean || !edge_mean) return DS_ERR_ARG;
  if (L->B <= 0 || L->max_n > DS_MAX_ATOMS || L->max_n > L->N) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_noise_step<2>, dim3(L->B), dim3(256), 0, (hipStream_t)stream, *L, 0.0f, 0.0f, 0.0f, temperature,
                     (unsigned long long)seed, 0u, mol_id, x, edge_x, pred, edge_pred, x_mean, edge_mean, table, step);
  return launch_status();
}

int ds_post_process(const ds_layout* L, const float* xh, const float* edge_x, float* pos_out, int32_t* atom_type, int32_t* fc,
                    float* edge_type, void* stream) {
  if (!L || !xh || !edge_x || !pos_out || !atom_type || !fc || !edge_type) return DS_ERR_ARG;
  if (L->B <= 0 || L->max_n > DS_MAX_ATOMS || L->max_n > L->N) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  const size_t nb = (size_t)L->B * L->N;
  if (hipMemsetAsync(pos_out, 0, nb * 3 * sizeof(float), s) != hipSuccess) return DS_ERR_LAUNCH;
  if (hipMemsetAsync(atom_type, 0, nb * sizeof(int32_t), s) != hipSuccess) return DS_ERR_LAUNCH;
  if (hipMemsetAsync(fc, 0, nb * sizeof(int32_t), s) != hipSuccess) return DS_ERR_LAUNCH;
  if (hipMemsetAsync(edge_type, 0, nb * L->N * sizeof(float), s) != hipSuccess) return DS_ERR_LAUNCH;
  hipLaunchKernelGGL(k_post_process, dim3(L->B), dim3(128), 0, s, *L, xh, edge_x, pos_out, atom_type, fc, edge_type);
  return launch_status();
}

int ds_check_stability(const ds_layout* L, const float* pos, const int32_t* atom_type, int32_t* bond_order, int32_t* nr_stable,
                       int32_t* mol_stable, void* stream) {
  if (!L || !pos || !atom_type || !nr_stable || !mol_stable) return DS_ERR_ARG;
  if (L->B <= 0 || L->max_n > DS_MAX_ATOMS || L->max_n > L->N) return DS_ERR_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (bond_order && hipMemsetAsync(bond_order, 0, (size_t)L->B * L->N * L->N * sizeof(int32_t), s) != hipSuccess) return DS_ERR_LAUNCH;
  hipLaunchKernelGGL(k_check_stability, dim3(L->B), dim3(64), 0, s, *L, pos, atom_type, bond_order, nr_stable, mol_stable);
  return launch_status();
}

int ds_spec_attention(const float* qkv, float* scores, float* out, int B, int L, int heads, int dk, float scale, int has_prev,
                      void* stream) {
  if (!qkv || !scores || !out || dk != 8 || B <= 0 || L <= 0) return DS_ERR_ARG;
  const int threads = min(1024, (L + 63) / 64 * 64);
  dim3 grid((L + threads - 1) / threads, heads, B);
  hipLaunchKernelGGL(k_spec_attention, grid, dim3(threads), (size_t)L * 8 * 2 * sizeof(float), (hipStream_t)stream, qkv, scores, out, B,
                     L, heads, scale, has_prev);
  return launch_status();
}

int ds_profile_config(int kernel, int every, int max_samples) {
  for (auto& e : g_prof.pool) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  g_prof.pool.clear();
  g_prof.used = 0; g_prof.seen = 0;
  g_prof.kernel = kernel; g_prof.every = every > 0 ? every : 1;
  if (kernel < 0) return DS_OK;
  if (kernel > 6 || max_samples <= 0) return DS_ERR_ARG;
  g_prof.pool.resize((size_t)max_samples);
  for (auto& e : g_prof.pool)
    if (hipEventCreate(&e.first) != hipSuccess || hipEventCreate(&e.second) != hipSuccess) return DS_ERR_LAUNCH;
  return DS_OK;
}

int ds_profile_read(double* total_ms, int64_t* samples) {
  if (!total_ms || !samples) return DS_ERR_ARG;
  double tot = 0.0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    float ms = 0.0f;
    if (hipEventSynchronize(g_prof.pool[i].second) != hipSuccess) return DS_ERR_LAUNCH;
    if (hipEventElapsedTime(&ms, g_prof.pool[i].first, g_prof.pool[i].second) != hipSuccess) return DS_ERR_LAUNCH;
    tot += ms;
  }
  *total_ms = tot; *samples = (int64_t)g_prof.used;
  g_prof.used = 0; g_prof.seen = 0;
  return DS_OK;
}

int ds_layernorm_affine(const float* x, const float* gamma, const float* beta, float* y, int rows, int cols, float eps,
                        void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || cols <= 0) return DS_ERR_ARG;
  hipLaunchKernelGGL(k_layernorm_affine, dim3(rows), dim3(64), 0, (hipStream_t)stream, x, gamma, beta, y, rows, cols, eps);
  return launch_status();
}

}  // extern "C"
