/*
 * diffspectra_train.h — C-ABI of the MI355X (gfx950) TRAINING path of the DMT (BASELINE config 5).
 *
 * The reference trains through torch.autograd over models/dmt.py; this library provides the forward AND the hand-written
 * backward of every operation of that graph as explicit kernels over the packed-ragged layout of diffspectra_hip.h
 * (node rows / unordered-pair rows / directed rows d = 2p + dir with dir 0 = (row a, col b), dir 1 = (row b, col a)).
 * The host side (diffspectra_amd/train_engine.py) strings them into the forward tape and its reverse; there is no autograd
 * graph and no PyTorch arithmetic in between.  Plain device pointers + sizes + hipStream_t, int status.
 *
 * Reference operations covered (file:line in /root/reference):
 *   dst_gemm                 every nn.Linear forward, input gradient and weight gradient
 *   dst_lnmod_*              LayerNorm(eps 1e-6, no affine) + modulate             models/dmt.py:13-14,86-97,148-149,160,166
 *   dst_gate_add_*           gated residuals                                         models/dmt.py:159-169
 *   dst_geom_*               coord2dist + CondGaussianLayer                          models/utils.py:129-133, models/layers.py:291-295,328-334
 *   dst_attn_*               TransMixLayer.message + PyG softmax / propagate         models/layers.py:131-186
 *   dst_pair_sum_*, dst_zbuild_*   h[row] + h[col] gathers                           models/dmt.py:39,156
 *   dst_coord_*              CoorsNorm, head mixing, scatter, remove_mean_with_mask  models/dmt.py:40-58,385-386, models/layers.py:344-347
 *   dst_time_feat_*          LearnedSinusodialposEmb                                 models/layers.py:283-288
 *   dst_loss                 the three MSE terms and their gradients                 losses.py:359-394
 *   dst_noising, dst_kabsch  forward diffusion + Kabsch alignment                    losses.py:312-327,414-452; models/utils.py:67-106
 *   dst_bn_*, dst_spec_attn_* SpecFormer in training mode                            models/specformer.py:247,260,385-425
 *   dst_adamw_ema            AdamW(amsgrad) + EMA update, fused                      losses.py:20,92; models/ema.py:24-42
 *   dst_clip_update          adaptive gradient clipping and its norm history         losses.py:28-72
 */
#ifndef DIFFSPECTRA_TRAIN_H
#define DIFFSPECTRA_TRAIN_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Generic GEMM, fp32 storage: C[M,N] (+)= A[M,K] * B[K,N] (+ bias[N]) on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32) or - args.bf16,
 * BASELINE config 5 - with the operands rounded to bf16 as they are staged into LDS, products on v_mfma_f32_32x32x16_bf16, fp32 accumulation.
 * Element strides make every transpose a view: A[m,k] = A[m*a_rs + k*a_cs], B[k,n] = B[k*b_rs + n*b_cs], C row stride ldc.
 *   Linear forward   Y = X W^T + b : A = X (a_rs = ldx, a_cs = 1), B = W^T (b_rs = 1, b_cs = ldw)
 *   input gradient   dX = dY W     : A = dY,                        B = W   (b_rs = ldw, b_cs = 1)
 *   weight gradient  dW = dY^T X   : A = dY^T (a_rs = 1, a_cs = ldy), B = X (b_rs = ldx, b_cs = 1), K = rows
 * accumulate != 0 adds to C.  Long-K products are split over K: every slice writes an fp32 slab into `partial` (caller scratch of
 * partial_cap floats) and a second kernel adds the slabs in slice order: results do not depend on scheduling.
 * Fused epilogue, per element (all optional):  v = acc + bias;  dact 1..3: v *= f'(ref[m,n]), dact 4: v += ref[m,n] (a residual);  act: C = v and C2 = drop(f(v)) when C2 is
 * given, else C = drop(f(v));  no act: C (+)= drop(v).  f: 1 SiLU, 2 GELU(erf), 3 tanh; f' takes ref = pre-activation (SiLU, GELU) or
 * ref = tanh output.  drop(x) = x * keep / (1 - p) with the Philox mask of dst_dropout at element index m * drop_ld + n of stream
 * (drop_seed, drop_stream) - the dropout of dmt.py:114-120 applied where the value is produced, and re-created in the backward.
 * Operand access in the bf16 mode: an operand that is contiguous along k or along its rows, 16-byte aligned, with a leading stride that is
 * a multiple of 4 takes the vector kernels, which read whole 16-byte groups from addresses clamped into the matrix and mask afterwards -
 * every row must therefore be READABLE up to the next multiple of 4 (<= its leading stride), which any [rows, ld] allocation is.  Other
 * operands and K < 8 take element-wise kernels (K <= 8 with M >= 1024: plain fp32 FMAs, no bf16 rounding). */
typedef struct dst_gemm_args {
  const float* A; int64_t a_rs, a_cs;
  const float* B; int64_t b_rs, b_cs;
  float* C; int64_t ldc;
  const float* bias;
  int32_t M, N, K, accumulate;
  float* partial; int64_t partial_cap;
  int32_t bf16; int32_t _pad;   /* != 0: bf16 products, fp32 accumulate (config 5) */
  float* rowsum;                /* optional [M]: rowsum[m] (+)= sum_k A[m,k] in the same pass (a virtual all-ones column of B): with
                                   A = dY^T this is the bias gradient of the weight-gradient product, no separate column-sum launches */
  int32_t act, dact;            /* fused activation / activation derivative (0 = none) */
  const float* ref; int64_t ldref;
  float* C2; int64_t ldc2;
  float drop_p; uint32_t drop_stream; uint64_t drop_seed; int64_t drop_ld;
} dst_gemm_args;
int dst_gemm(const dst_gemm_args* a, void* stream);

/* out[0] = sizeof(dst_gemm_args), out[1] = sizeof(dst_layout), out[2] = sizeof(dst_piece): the binding checks its own struct layouts
 * against the library's. */
int dst_struct_sizes(int64_t* out);

/* n strided 2-D copies in ONE launch: dst[r*dst_ld + c] = src[r*src_ld + c] for r < rows, c < cols of every piece; `table` is a DEVICE
 * array.  (The per-step concatenation of Linears that share an input - q | k | v, the adaLN table, ... - and the scatter of the
 * concatenated gradients: ~150 pieces each way, one launch instead of one copy per piece.) */
typedef struct dst_piece {
  const float* src; float* dst;
  int32_t rows, cols;
  int64_t src_ld, dst_ld;
} dst_piece;
int dst_copy_pieces(const dst_piece* table, int32_t n, void* stream);
/* The same pieces ROUNDED TO bf16 (nearest even): `dst` of every piece addresses a bf16 buffer (uint16_t bits), dst_ld in bf16 elements;
 * dst_ld < 0 stores the piece TRANSPOSED (dst[c * -dst_ld + r] = src[r * src_ld + c]).  The weights of the fused row chains below are
 * passed in this form. */
int dst_pack_bf16_pieces(const dst_piece* table, int32_t n, void* stream);

/* out[c] (+)= sum_r X[r*ld + c], two fixed-order stages through `scratch` (bias gradients, per-molecule partial sums). */
int dst_colsum(const float* X, int64_t ld, int32_t R, int32_t C, float* out, int32_t accumulate, float* scratch,
               int64_t scratch_cap, void* stream);

/* Elementwise.  kind: 1 SiLU, 2 GELU(erf), 3 tanh.  Backward: dx = dy * f'(.), with ref = x for SiLU / GELU and ref = y for
 * tanh; dx may alias dy.  dst_axpy: y += a * x.  dst_scale_rows: y[r, :] = x[r, :] * s[r]. */
int dst_act_fwd(const float* x, float* y, int64_t n, int32_t kind, void* stream);
int dst_act_bwd(const float* dy, const float* ref, float* dx, int64_t n, int32_t kind, void* stream);
int dst_axpy(float a, const float* x, float* y, int64_t n, void* stream);

/* nn.Dropout(p) in training mode (dmt.py:114-120): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from the Philox4x32-10 stream
 * (seed, stream_id) indexed by the element.  The same call on the gradient is the backward (the mask is re-created, not stored).
 * y may alias x. */
int dst_dropout(const float* x, float* y, int64_t n, float p, uint64_t seed, uint32_t stream_id, void* stream);

/* Segments: rows of molecule m are seg_off[m]*seg_mul .. seg_off[m+1]*seg_mul (node_off / pair_off with seg_mul 1, directed rows
 * with pair_off and seg_mul 2).  ada [B, ada_ld] is the per-molecule adaLN table and d_ada its gradient; *_off are column offsets. */

/* y = LN(x) * (1 + scale[m]) + shift[m], C in {64, 256}; stats [rows,2] = (mean, rstd).  Backward writes dx (accumulate != 0
 * adds) and d_ada[m, shift_off + c] = sum_rows dy, d_ada[m, scale_off + c] = sum_rows dy * xhat.  scratch (optional, >= 8 * B * 2 C
 * floats): with it the directed rows (C = 256, seg_mul >= 2) are shared by eight workgroups per molecule and the sums take a second pass
 * that adds the eight shares in a fixed order. */
int dst_lnmod_fwd(const float* x, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B, const float* ada, int64_t ada_ld,
                  int32_t shift_off, int32_t scale_off, float* y, float* stats, void* stream);
int dst_lnmod_bwd(const float* dy, const float* x, const float* stats, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B,
                  const float* ada, float* d_ada, int64_t ada_ld, int32_t shift_off, int32_t scale_off, float* dx, int32_t accumulate,
                  float* scratch, int64_t scratch_cap, void* stream);

/* out = r + gate[m] * z.  Backward: dr (accumulate_r != 0 adds) = dout, dz = gate * dout, d_ada[m, gate_off + c] = sum_rows dout * z.
 * drop_p > 0: z was the output of a dropout (dmt.py:116,120) - dz is multiplied by that dropout's mask (dst_dropout's generator at the
 * element index of the contiguous [rows, C] tensor, stream (drop_seed, drop_stream)), i.e. dz is the gradient in front of the dropout. */
int dst_gate_add_fwd(const float* r, const float* z, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B, const float* ada,
                     int64_t ada_ld, int32_t gate_off, float* out, void* stream);
int dst_gate_add_bwd(const float* dout, const float* z, int32_t C, const int32_t* seg_off, int32_t seg_mul, int32_t B, const float* ada,
                     float* d_ada, int64_t ada_ld, int32_t gate_off, float* dr, int32_t accumulate_r, float* dz, float drop_p, uint64_t drop_seed,
                     uint32_t drop_stream, void* stream);

/* Layout tables shared by the molecule-level kernels (device pointers; the tables of ds_layout). */
typedef struct dst_layout {
  int32_t B, Nn, Pp, _pad;
  const int32_t* node_off;   /* [B+1] */
  const int32_t* pair_off;   /* [B+1] */
} dst_layout;

/* coord2dist + CondGaussianLayer: d2 = |pos_a - pos_b|^2, x' = d2 * (1 + ada[m, dist_off]) + ada[m, dist_off + 1],
 * feat = [x', gaussian_k(x')] (63 kernels, std = |stds| + 1e-5, pi = 3.14159) written to X[p*ldx + k] for k < 64; xs[p] = x',
 * d2s[p] = d2.  pos [Nn,3].  Backward: the gradient of the 64 features arrives from up to two places (g1, g2; g2 may be NULL);
 * writes d_ada[m, dist_off .. +1], per-molecule partial sums of d means / d stds to dms [B,128] (d stds with respect to the raw
 * stds.weight: sign included), and - when dpos != NULL - accumulates into dpos [Nn,3]. */
int dst_geom_fwd(const dst_layout* L, const float* pos, const float* ada, int64_t ada_ld, int32_t dist_off, const float* means,
                 const float* stds, float* X, int64_t ldx, float* xs, float* d2s, void* stream);
int dst_geom_bwd(const dst_layout* L, const float* pos, const float* ada, float* d_ada, int64_t ada_ld, int32_t dist_off,
                 const float* means, const float* stds, const float* xs, const float* d2s, const float* g1, int64_t ld1, const float* g2,
                 int64_t ld2, float* dms, float* dd2_scratch, float* dpos, void* stream);

/* Adjacency bits of the self-conditioning prediction (dmt.py:338-340,361) per pair row: bit 0 = cond_e[p*ld] >= edge_th (cond_adj_2d),
 * bit 1 = d2c[p] <= cutoff (cond_adj_spatial; d2c = squared distance of the conditioning positions). */
int dst_adj_bits(const float* cond_e, int64_t ld, const float* d2c, float edge_th, float cutoff, int32_t Pp, int32_t* adj, void* stream);

/* TransMixLayer attention per molecule.  qkv [Nn,768]: q at columns 0..251, k at 256..507, v at 512..767; te0 [Pp,256]
 * (= tanh(lin_edge0 e), 252 used), te1 [Pp,256], both (and their gradients) with row stride ld_te (512 when they are the two halves of
 * one [Pp,512] buffer: lin_edge0 | lin_edge1 evaluated as one product), adj [Pp] bits (1: cond_adj_2d, 2: cond_adj_spatial); out [Nn,256];
 * alpha [2*Pp,16] (row 2p: source a -> target b, row 2p+1: source b -> target a).  16 heads: 0,1 adjacency heads (0 -> -1e10),
 * 2..15 learned (18 channels, scale 1/sqrt(16)); softmax over the sources of a target with + 1e-16 in the denominator.
 * Backward: dqkv [Nn,768], dte0 [Pp,256], dte1 [Pp,256] are written (not accumulated); te_is_tanh != 0: te0 / te1 are tanh outputs and
 * dte0 / dte1 come back as the gradients in FRONT of the tanh (times 1 - te^2), which is what the lin_edge products need.  scratch
 * (optional, >= 32 Pp floats): with it the backward runs as two launches, the second with four workgroups per molecule. */
int dst_attn_fwd(const dst_layout* L, const float* qkv, const float* te0, const float* te1, int64_t ld_te, const int32_t* adj, float* out,
                 float* alpha, void* stream);
int dst_attn_bwd(const dst_layout* L, const float* qkv, const float* te0, const float* te1, int64_t ld_te, const float* alpha, const float* dout,
                 float* dqkv, float* dte0, float* dte1, int32_t te_is_tanh, float* scratch, int64_t scratch_cap, void* stream);

/* s[p] = u[a] + u[b] (+ bias[c]) over C columns; backward du[i] (accumulate != 0 adds) = sum over the pairs of atom i of ds[p]. */
int dst_pair_sum_fwd(const dst_layout* L, const float* u, int32_t C, const float* bias, float* s, void* stream);
int dst_pair_sum_bwd(const dst_layout* L, const float* ds, int32_t C, float* du, int32_t accumulate, void* stream);

/* z[2p + dir] = ac[row, 0:256] + ac[col, 256:512] + ed[p] (dmt.py:39 with input_lin split into its row / col / edge parts).
 * Backward: dac [Nn,512] and ded [Pp,256] are written. */
int dst_zbuild_fwd(const dst_layout* L, const float* ac, const float* ed, float* z, void* stream);
int dst_zbuild_bwd(const dst_layout* L, const float* dz, float* dac, float* ded, void* stream);

/* Equivariant coordinate update + centre-of-mass removal (dmt.py:40-58,385-386): c2 [2*Pp,3] pre-tanh head outputs,
 * inv = (tanh c2_0 + adj2d tanh c2_1 + adjsp tanh c2_2) / 3, pos_new[row] = pos[row] + sum_col CoorsNorm(pos[row] - pos[col]) inv,
 * pos_out = pos_new - mean.  Backward from dpos_out: dpos_in [Nn,3] (written), dc2 [2*Pp,3], dscale_part [B] (per-molecule partial
 * of d coord_norm.scale). */
int dst_coord_fwd(const dst_layout* L, const float* pos, const float* c2, const int32_t* adj, const float* coord_scale,
                  float* pos_out, void* stream);
int dst_coord_bwd(const dst_layout* L, const float* pos, const float* c2, const int32_t* adj, const float* coord_scale,
                  const float* dpos_out, float* dpos_in, float* dc2, float* dscale_part, void* stream);

/* LearnedSinusodialposEmb: f [B,17] = [x, sin(2 pi x w_k), cos(2 pi x w_k)], k < 8.  Backward: dw [8] (written). */
int dst_time_feat_fwd(const float* noise_level, const float* w, int32_t B, float* f, void* stream);
int dst_time_feat_bwd(const float* noise_level, const float* w, const float* df, int32_t B, float* dw, void* stream);

/* Loss of losses.py:359-394 (pred_data, reduce_mean False) on packed predictions and its gradient: per molecule
 * l = wm[m] * (w_pos * sum_atoms mean_3 (pos - tpos)^2 + w_type * sum_atoms mean_6 (feat - tfeat)^2
 *              + w_edge * 2 sum_pairs mean_2 (edge - tedge)^2), wm[m] = sqrt(alpha_m / sigma_m) / B supplied by the caller;
 * loss_m [B]; dpos [Nn,3], dfeat [Nn,6], dedge [Pp,2] written. */
int dst_loss(const dst_layout* L, const float* pos, const float* feat, const float* edge, const float* tpos, const float* tfeat,
             const float* tedge, const float* wm, float w_pos, float w_type, float w_edge, float* loss_m, float* dpos, float* dfeat,
             float* dedge, void* stream);

/* process_edge_batch + get_data_scaler on packed rows (losses.py:498-529, utils.py:33-68, centered data): x [Nn,9] = [(pos - CoM) /
 * pos_norm, (one_hot * 2 - 1) / type_norm, fc / fc_norm]; ex [Pp,2] = (edge * 2 - 1) / edge_norm.  pos [Nn,3], one_hot [Nn,5], fc [Nn],
 * edge [Pp,2] (exist, bond order / 3). */
int dst_prepare_batch(const dst_layout* L, const float* pos, const float* one_hot, const float* fc, const float* edge, float pos_norm,
                      float type_norm, float fc_norm, float edge_norm, float* x, float* ex, void* stream);

/* Forward diffusion on packed rows (losses.py:312-320, models/utils.py:67-106): z = alpha[m] x + sigma[m] noise with the position
 * noise centre-of-mass projected per molecule.  x [Nn,9] clean, raw [Nn,9] standard normals (columns 0..2 positions), out z [Nn,9];
 * ex [Pp,2] clean pair features, eraw [Pp,2] normals, out ez [Pp,2]. */
int dst_noising(const dst_layout* L, const float* alpha, const float* sigma, const float* x, const float* raw, float* z, const float* ex,
                const float* eraw, float* ez, void* stream);

/* Kabsch alignment (losses.py:414-452): rot[m] = U diag(1, 1, sign det A) V^T of A = sum_atoms pred_i tar_i^T (3x3 SVD by one-sided
 * Jacobi in fp64), aligned[i] = rot tar_i.  pred / tar / aligned [Nn, ld] (first three columns); rot [B,9]. */
int dst_kabsch(const dst_layout* L, const float* pred, int64_t ld_pred, const float* tar, int64_t ld_tar, float* rot, float* aligned,
               void* stream);

/* BatchNorm1d in training mode over the columns of X [R, C] (C <= 256): y = (x - mean_c) / sqrt(var_c + eps) * gamma_c + beta_c with
 * biased batch variance; stats [3, C] = (mean, rstd, unbiased variance) saved; running statistics updated in place with momentum 0.1
 * and the unbiased variance.  Backward: dx written, dgamma / dbeta [C] written (reads the first two rows of stats).
 * dst_bn_running_again applies the running-statistics update of one more forward over the SAME batch from the saved stats: the
 * reference's self-conditioning step (losses.py:344-357) runs the conditioning encoder twice on one input - the second pass
 * changes nothing but these buffers. */
int dst_bn_fwd(const float* x, int32_t R, int32_t C, const float* gamma, const float* beta, float eps, float* y, float* stats,
               float* running_mean, float* running_var, float* scratch, int64_t scratch_cap, void* stream);
int dst_bn_running_again(const float* stats, int32_t C, float* running_mean, float* running_var, void* stream);
int dst_bn_bwd(const float* dy, const float* x, const float* stats, int32_t R, int32_t C, const float* gamma, float* dx, float* dgamma,
               float* dbeta, float* scratch, int64_t scratch_cap, void* stream);

/* SpecFormer attention in training form (specformer.py:385-425): scores [B,H,L,Lp] = q k^T * scale (+ prev) are kept (Lp = L rounded
 * up to a multiple of 32: rows start on 128-byte boundaries; pad columns are never touched) - the next layer adds them to its own and
 * the backward re-creates the probabilities exp(scores - max) / sum from them and stats [B,H,L,2] = (row max, row sum), so the
 * [B,H,L,L] probability tensor is never written.  out [B,L,H*dk]; qkv [B,L,3*H*dk] (q | k | v).  Backward: dscores_in (gradient arriving
 * at THIS layer's scores from the next layer's `prev` use; may be NULL) is added to the softmax gradient; writes dqkv and dscores (total
 * gradient of this layer's scores = what flows on to the previous layer's scores). */
int dst_spec_attn_fwd(const float* qkv, const float* prev, float* scores, float* stats, float* out, int32_t B, int32_t L, int32_t H,
                      int32_t dk, float scale, void* stream);
int dst_spec_attn_bwd(const float* qkv, const float* scores, const float* stats, const float* dout, const float* dscores_in, float* dqkv,
                      float* dscores, int32_t B, int32_t L, int32_t H, int32_t dk, float scale, void* stream);

/* The same attention WITHOUT the [B,H,L,L] tensors (bf16 training mode, BASELINE config 5): the residual scores of layer l are
 * scale * sum_{j<=l} q_j k_j^T, so the kernels recompute them on the bf16 matrix pipe from the q | k slices of the layers 0 .. l
 * (qkv0 .. qkv2, n_layers = l + 1 of them; each [B*L, 3*H*dk]) flash-style: forward writes out [B*L, H*dk] and stats [B*H*L, 2] =
 * (row maximum of the log2-domain scores, row sum); backward re-creates the probabilities from them and adds layer l's softmax
 * gradient to the q and k columns of dqkv0 .. dqkv(l) (the gradient the reference chains through `prev`) and writes the v columns of
 * dqkv(l).  The caller runs the layers last to first: accumulate = 0 for the FIRST call (the last layer: it ASSIGNS the q and k columns of
 * every buffer, which therefore need no zeroing and are not read), 1 for the others.  part: 0 = the whole backward; 1 = the query side
 * (q columns) only, 2 = the key side (k and v columns) only - the two touch disjoint columns and may run on two streams. */
int dst_spec_attn_flash_fwd(const float* qkv0, const float* qkv1, const float* qkv2, int32_t n_layers, float* stats, float* out, int32_t B,
                            int32_t L, int32_t H, int32_t dk, float scale, void* stream);
int dst_spec_attn_flash_bwd(const float* qkv0, const float* qkv1, const float* qkv2, int32_t n_layers, const float* stats, const float* out,
                            const float* dout, float* dqkv0, float* dqkv1, float* dqkv2, int32_t B, int32_t L, int32_t H, int32_t dk, float scale,
                            int32_t part, int32_t accumulate, void* stream);

/* LayerNorm with affine over the last dimension (specformer.py:67,119), training form.  Backward: dx written, dgamma / dbeta
 * accumulated through per-row-block partials (fixed order). */
int dst_ln_affine_fwd(const float* x, int32_t R, int32_t C, const float* gamma, const float* beta, float eps, float* y, float* stats,
                      void* stream);
int dst_ln_affine_bwd(const float* dy, const float* x, const float* stats, int32_t R, int32_t C, const float* gamma, float* dx,
                      float* dgamma, float* dbeta, void* stream);

/* Fused optimizer step over one flat fp32 parameter buffer (losses.py:20 AdamW(amsgrad=True, weight_decay), torch semantics) followed
 * by the EMA update of models/ema.py:24-42: p, g, m, v, vmax, ema all [n].  The gradient is first multiplied by clip_coef and, when
 * clip_coef_dev is not NULL, by clip_coef_dev[0] (the device-resident coefficient of dst_clip_update: the step then needs no
 * device -> host synchronisation); bias corrections are passed in (1 - beta^t). */
int dst_adamw_ema(float* p, const float* g, float* m, float* v, float* vmax, float* ema, int64_t n, float lr, float beta1, float beta2,
                  float eps, float weight_decay, float bc1, float bc2, float clip_coef, const float* clip_coef_dev, float ema_one_minus_decay,
                  void* stream);

/* gradient_clipping (losses.py:28-50) with its norm history (Queue, losses.py:53-72) kept on the device.  norm_sq[0] = sum of squares of
 * the gradient (dst_sumsq, all-reduced over ranks), the norm is sqrt(norm_sq) * inv_world.  state [54] floats: [0..49] the history,
 * newest first; [50] its length (the reference starts it as {3000}); on return [51] = min(1, allowed / (norm + 1e-6)), [52] = norm,
 * [53] = allowed = min(1.5 mean + 2 std of the history, max_grad) (max_grad <= 1: allowed = max_grad, no history), and the history has
 * taken min(norm, allowed).  Evaluated in double, as numpy does. */
int dst_clip_update(const float* norm_sq, float inv_world, float max_grad, float* state, void* stream);

/* sum of squares of x [n] into out[0] (accumulate != 0 adds): the global gradient norm of clip_grad_norm_. */
int dst_sumsq(const float* x, int64_t n, float* out, int32_t accumulate, float* scratch, int64_t scratch_cap, void* stream);

/* The pair rows of one block behind the attention as ONE kernel (bf16 products, fp32 accumulation and fp32 everything else) - reference
 * dmt.py:156-157,165-169,388 and the edge part of equi_update.input_lin (dmt.py:39); replaces dst_pair_sum_fwd, 2 x dst_gate_add_fwd,
 * dst_lnmod_fwd and four dst_gemm calls of the unfused forward:
 *   he = u[a] + u[b] + n2e_bias;  xe1 = e_in + ada[gate1] * he;  ye1 = LN(xe1) * (1 + ada[scale]) + ada[shift];
 *   f3 = ye1 W3^T + b3;  s3 = dropout(SiLU(f3));  f4 = dropout(s3 W4^T + b4);  e_out = ye1 + ada[gate2] * f4;
 *   ed = [e_out | feat] Wed^T + bed;  ro = e_out Wro^T + bro.
 * u [Nn,64]; e_in [Pp,64]; feat [Pp,64] with row stride ld_feat; ada [B, ada_ld] with the four column offsets; W3 [128,64], W4 [64,128],
 * Wed [256 rows, row stride ld_wed, 128 used columns: e | dist], Wro [16,64] (torch Linear layout) as bf16 BITS (dst_pack_bf16_pieces;
 * ld_wed in bf16 elements, a multiple of 8): the kernel streams the weights from L2 once per 32-row tile, and that stream bounds it
 * (profiles/r05_train_fused_ab.txt).  Biases fp32.  Dropout: dst_dropout's masks
 * (element (row, col) of the [Pp,128] / [Pp,64] tensor under stream3 / stream4).  Outputs: e_out [Pp,64], ed [Pp,256], ro [Pp,16]
 * always; the tape tensors he, xe1, st [Pp,2] = (mean, rstd), ye1, f3 [Pp,128], s3 [Pp,128], f4, X2 [Pp,128] = [e_out | feat] may each
 * be NULL (not written).  Every pointer 16-byte aligned. */
typedef struct dst_pair_chain_args {
  const int32_t* pair_a; const int32_t* pair_b; const int32_t* pair_mol;   /* [Pp]: node rows of a pair's atoms, its molecule */
  const float* u; const float* n2e_bias; const float* e_in; const float* feat; int64_t ld_feat;
  const float* ada; int64_t ada_ld; int32_t gate1_off, shift_off, scale_off, gate2_off;
  const uint16_t* W3; const float* b3; const uint16_t* W4; const float* b4; const uint16_t* Wed; int64_t ld_wed; const float* bed;
  const uint16_t* Wro; const float* bro;
  float drop_p; uint32_t stream3, stream4, _pad; uint64_t seed;
  float* he; float* xe1; float* st; float* ye1; float* f3; float* s3; float* f4; float* e_out; float* X2; float* ed; float* ro;
} dst_pair_chain_args;
int dst_pair_chain_fwd(const dst_layout* L, const dst_pair_chain_args* a, void* stream);

/* The pair rows of one block in FRONT of the attention as one kernel (bf16 products; dmt.py:136-139,145-149, layers.py:291-295,328-334,
 * 165-166,183); replaces dst_geom_fwd, the [feat | e] copy, dst_lnmod_fwd and two dst_gemm calls:
 *   X1 = [CondGaussian features of the modulated squared distance (64) | e_in (64)];  e1 = X1 Wee^T + bee;
 *   en = LN(e1) * (1 + ada[scale]) + ada[shift];  te = tanh(en Wte^T)   (Wte [512,64] = lin_edge0 | lin_edge1, rows 252..255 zero).
 * pos [Nn,3]; means, stds [63]; ada column dist_off holds the distance scale, dist_off + 1 its shift; Wee [64,128] and Wte as bf16 bits.
 * Outputs: X1 [Pp,128] and te [Pp,512]
 * always; xs [Pp] (x'), d2 [Pp], e1 [Pp,64], st [Pp,2] = (mean, rstd), en [Pp,64] may be NULL. */
typedef struct dst_pair_front_args {
  const int32_t* pair_a; const int32_t* pair_b; const int32_t* pair_mol;
  const float* pos; const float* ada; int64_t ada_ld; int32_t dist_off, shift_off, scale_off, _pad;
  const float* means; const float* stds; const float* e_in; const uint16_t* Wee; const float* bee; const uint16_t* Wte;
  float* X1; float* xs; float* d2; float* e1; float* st; float* en; float* te;
} dst_pair_front_args;
int dst_pair_front_fwd(const dst_layout* L, const dst_pair_front_args* a, void* stream);

/* The directed rows of one block (both directions of every pair; dmt.py:37-48) as one kernel (bf16 products); replaces dst_zbuild_fwd,
 * dst_lnmod_fwd (256 wide, two rows per pair) and two dst_gemm calls:
 *   zz[2p + dir] = ac[row, 0:256] + ac[col, 256:512] + ed[p];  zn = LN(zz) (1 + ada[scale]) + ada[shift];
 *   c0 = zn W0^T + b0;  sc0 = SiLU(c0);  c2 = sc0 W2^T.
 * ac [Nn,512], ed [Pp,256], W0 [256,256], W2 [3,256] (torch Linear layout, bf16 bits).  Outputs: c2 [2 Pp, 3] always; zz, st [2 Pp, 2], zn, c0, sc0
 * [2 Pp, 256] may each be NULL. */
typedef struct dst_dir_chain_args {
  const int32_t* pair_a; const int32_t* pair_b; const int32_t* pair_mol;
  const float* ac; const float* ed; const float* ada; int64_t ada_ld; int32_t shift_off, scale_off;
  const uint16_t* W0; const float* b0; const uint16_t* W2;
  float* zz; float* st; float* zn; float* c0; float* sc0; float* c2;
} dst_dir_chain_args;
int dst_dir_chain_fwd(const dst_layout* L, const dst_dir_chain_args* a, void* stream);

/* The node rows of one block behind the attention as one kernel (bf16 products; dmt.py:113-116,158-163,387 and the node parts of
 * equi_update.input_lin, dmt.py:39); replaces 2 x dst_gate_add_fwd, dst_lnmod_fwd and four dst_gemm calls:
 *   x1 = h_in + ada[gate1] * attn;  y1 = LN(x1) (1 + ada[scale]) + ada[shift];  f1 = y1 W1^T + b1;  s1 = dropout(SiLU(f1));
 *   f2 = dropout(s1 W2^T + b2);  h_out = y1 + ada[gate2] * f2;  ac = h_out Wac^T;  rn = h_out Wn^T + bn.
 * h_in, attn [Nn,256]; node_mol [Nn] = molecule of a node row; W1 [512,256], W2 [256,512], Wac [512,256] = the h_row | h_col parts of
 * input_lin, Wn [64,256] as bf16 bits (dst_pack_bf16_pieces); biases fp32.  Dropout: dst_dropout's masks (element (row, col) of the
 * [Nn,512] / [Nn,256] tensor under stream1 / stream2).  Outputs: h_out [Nn,256], ac [Nn,512], rn [Nn,64] always; the tape tensors x1,
 * st [Nn,2] = (mean, rstd), y1, f1 [Nn,512], s1 [Nn,512], f2 may each be NULL.  Every pointer 16-byte aligned. */
typedef struct dst_node_chain_args {
  const int32_t* node_mol; const float* h_in; const float* attn;
  const float* ada; int64_t ada_ld; int32_t gate1_off, shift_off, scale_off, gate2_off;
  const uint16_t* W1; const float* b1; const uint16_t* W2; const float* b2; const uint16_t* Wac; const uint16_t* Wn; const float* bn;
  float drop_p; uint32_t stream1, stream2, _pad; uint64_t seed;
  float* x1; float* st; float* y1; float* f1; float* s1; float* f2; float* h_out; float* ac; float* rn;
} dst_node_chain_args;
int dst_node_chain_fwd(const dst_layout* L, const dst_node_chain_args* a, void* stream);

/* Backward of the directed rows of one block (dmt.py:37-48) as one kernel + a finishing kernel (bf16 product); replaces the K = 3 dst_gemm
 * with SiLU', the 256 -> 256 input-gradient dst_gemm and dst_lnmod_bwd:
 *   dc0 = (dc2 W2) * SiLU'(c0);  dzn = dc0 W0;  dz = LN'(zz, st; dzn (1 + ada[scale]));  d_ada[shift] = sum_rows dzn;  d_ada[scale] = sum_rows dzn x^.
 * Tiles are molecule-aligned: tile_row0 / tile_rows / tile_mol [n_tiles] = first directed row, row count (<= 32) and molecule of a tile (a
 * molecule's tiles consecutive, ascending), mol_tile_off [B + 1] = first tile of a molecule.  dc2 [2 Pp,3]; c0, zz [2 Pp,256], st [2 Pp,2] the
 * forward's tape; W2 [3,256] fp32; W0T = coord_mlp.0's weight TRANSPOSED as bf16 bits ([in][out]; dst_pack_bf16_pieces with dst_ld < 0).
 * Outputs: dc0, dz [2 Pp,256]; d_ada columns shift_off .. + 255 and scale_off .. + 255 of every molecule ASSIGNED; part = scratch of
 * n_tiles * 512 floats. */
typedef struct dst_dir_bwd_args {
  const int32_t* tile_row0; const int32_t* tile_rows; const int32_t* tile_mol; const int32_t* mol_tile_off; int64_t n_tiles;
  const float* dc2; const float* c0; const float* zz; const float* st;
  const float* ada; float* d_ada; int64_t ada_ld; int32_t shift_off, scale_off;
  const float* W2; const uint16_t* W0T;
  float* dc0; float* dz; float* part;
} dst_dir_bwd_args;
int dst_dir_chain_bwd(const dst_layout* L, const dst_dir_bwd_args* a, void* stream);

/* Backward of dst_pair_chain_fwd (the pair rows of a block behind the attention) as one kernel + a finishing kernel (bf16 products);
 * replaces five input-gradient dst_gemm calls, 2 x dst_gate_add_bwd and dst_lnmod_bwd:
 *   de_tot = de + dro Wro + ded Wed[:, 0:64];  dfeat = ded Wed[:, 64:128];  df4 = ada[gate2] de_tot (x dropout mask 4);
 *   df3 = (df4 W4) SiLU'(f3) (x dropout mask 3);  dye1 = de_tot + df3 W3;  dxe1 = LN'(xe1, st; dye1 (1 + ada[scale]));
 *   de_in = dxe1;  dhe = ada[gate1] dxe1;  d_ada[gate2] = sum de_tot f4, [shift] = sum dye1, [scale] = sum dye1 x^, [gate1] = sum dxe1 he.
 * Molecule-aligned PAIR tiles (tile_row0 / tile_rows / tile_mol / mol_tile_off as in dst_dir_bwd_args).  de [Pp,64] = the gradient of e_out
 * that arrives from the next block; dro [Pp,16] with row stride ld_dro = the gradient of the read-out slice; ded [Pp,256]; f4, f3, xe1, st, he:
 * the forward's tape.  Weights TRANSPOSED as bf16 bits ([in][out]): WedT [128][256] (the e | dist columns of input_lin), WroT [64][16],
 * W4T [128][64], W3T [64][128].  Outputs: dfeat [Pp,64], df4 [Pp,64], df3 [Pp,128], de_in [Pp,64], dhe [Pp,64]; the four 64-column slices of
 * d_ada ASSIGNED; part = scratch of n_tiles * 256 floats. */
typedef struct dst_pair_bwd_args {
  const int32_t* tile_row0; const int32_t* tile_rows; const int32_t* tile_mol; const int32_t* mol_tile_off; int64_t n_tiles;
  const float* de; const float* dro; int64_t ld_dro; const float* ded;
  const float* f4; const float* f3; const float* xe1; const float* st; const float* he;
  const float* ada; float* d_ada; int64_t ada_ld; int32_t gate1_off, shift_off, scale_off, gate2_off;
  const uint16_t* WedT; const uint16_t* WroT; const uint16_t* W4T; const uint16_t* W3T;
  float drop_p; uint32_t stream3, stream4, _pad; uint64_t seed;
  float* dfeat; float* df4; float* df3; float* de_in; float* dhe; float* part;
} dst_pair_bwd_args;
int dst_pair_chain_bwd(const dst_layout* L, const dst_pair_bwd_args* a, void* stream);

/* Backward of dst_node_chain_fwd (the node rows of a block behind the attention) as one kernel + a finishing kernel (bf16 products);
 * replaces five input-gradient dst_gemm calls, 2 x dst_gate_add_bwd and dst_lnmod_bwd:
 *   dh_tot = dh + drn Wn + dac Wac;  df2 = ada[gate2] dh_tot (x dropout mask 2);  df1 = (df2 W2) SiLU'(f1) (x dropout mask 1);
 *   dy1 = dh_tot + df1 W1;  dx1 = LN'(x1, st; dy1 (1 + ada[scale]));  dh_in = dx1;  dattn = ada[gate1] dx1;
 *   d_ada[gate2] = sum dh_tot f2, [shift] = sum dy1, [scale] = sum dy1 x^, [gate1] = sum dx1 attn   (256 columns each, ASSIGNED).
 * Molecule-aligned NODE tiles (tile tables as in dst_dir_bwd_args).  dh [Nn,256] = the gradient of h_out from the next block; drn [Nn,64] with
 * row stride ld_drn = the gradient of the read-out slice; dac [Nn,512]; f2, f1, x1, st, attn: the forward's tape.  Weights TRANSPOSED as bf16
 * bits ([in][out]): WacT [256][512], WnT [256][64], W2T [512][256], W1T [256][512].  Outputs: df2 [Nn,256], df1 [Nn,512], dh_in, dattn
 * [Nn,256]; part = scratch of n_tiles * 1024 floats. */
typedef struct dst_node_bwd_args {
  const int32_t* tile_row0; const int32_t* tile_rows; const int32_t* tile_mol; const int32_t* mol_tile_off; int64_t n_tiles;
  const float* dh; const float* drn; int64_t ld_drn; const float* dac;
  const float* f2; const float* f1; const float* x1; const float* st; const float* attn;
  const float* ada; float* d_ada; int64_t ada_ld; int32_t gate1_off, shift_off, scale_off, gate2_off;
  const uint16_t* WacT; const uint16_t* WnT; const uint16_t* W2T; const uint16_t* W1T;
  float drop_p; uint32_t stream1, stream2, _pad; uint64_t seed;
  float* df2; float* df1; float* dh_in; float* dattn; float* part;
} dst_node_bwd_args;
int dst_node_chain_bwd(const dst_layout* L, const dst_node_bwd_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSPECTRA_TRAIN_H */
