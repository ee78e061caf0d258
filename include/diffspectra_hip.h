/*
 * diffspectra_hip.h — C-ABI of the MI355X (gfx950) DMT + SpecFormer denoising library.
 *
 * The reference (AzureLeon1/DiffSpectra) is pure Python: its "FFI" for this path is the Python call
 * convention of SURVEY §8b.  This library is what a binding for that path attaches to: plain device
 * pointers + sizes + a hipStream_t, no torch types, int status returns (0 = ok, <0 = error; the
 * Python mirror turns them into RuntimeError).  All buffers are caller-owned device memory
 * (PyTorch-ROCm allocations in the shipped host code).  No hidden global state: weights are a
 * caller-owned packed buffer described by ds_weights.
 *
 * Reference interfaces replaced (file:line in /root/reference):
 *   ds_forward            DMT.forward                         models/dmt.py:306-412
 *                         EquivariantMixBlock.forward         models/dmt.py:122-174
 *                         MultiCondEquiUpdate.forward         models/dmt.py:37-60
 *                         TransMixLayer.forward/message       models/layers.py:131-186
 *                         CondGaussianLayer / gaussian        models/layers.py:291-295,328-334
 *                         LearnedSinusodialposEmb + time_mlp  models/layers.py:283-288, dmt.py:249-257,353-357
 *   ds_sampler_step       AncestralSampler.sampling loop body sampling.py:604-624 + models/utils.py:67-106
 *   ds_initial_noise /    sample_combined_position_feature_noise, sample_symmetric_edge_feature_noise
 *   ds_sampler_step_philox                                    models/utils.py:67-106 (+ sampling.py:442-447,604-624)
 *   ds_post_process       post_process + inverse scaler       sampling.py:53-97, utils.py:88-103
 *   ds_check_stability    check_stability (distance half)     evaluation/stability.py:40-73, evaluation/bond_analyze.py:108-133
 *   ds_gemm / ds_spec_*   SpecFormer.forward                  models/specformer.py:77-120,167-200,279-309,345-425,457-470
 *
 * Data layout ("packed-ragged", symmetric pair storage — DESIGN.md §3):
 *   node rows   : valid atoms of all molecules, molecule-major           (Nn rows)
 *   pair rows   : unordered pairs a<b of each molecule, upper-triangular row-major,
 *                 p = pair_off[m] + a*(2n-a-1)/2 + (b-a-1)                 (Pp rows)
 *   dense I/O   : xh [B,N,9], edge_x [B,N,N,2] exactly as the reference passes them.
 */
#ifndef DIFFSPECTRA_HIP_H
#define DIFFSPECTRA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DS_OK 0
#define DS_ERR_ARG (-1)
#define DS_ERR_LAUNCH (-2)

#define DS_NBLOCKS 8
#define DS_HID 256
#define DS_EHID 64
#define DS_TDIM 1024
#define DS_ADA_BLOCK_STRIDE 2464 /* 1536 node | 384 edge | 512 equi | 2 dist | pad to 32 */
#define DS_ADA_NODE 0
#define DS_ADA_EDGE 1536
#define DS_ADA_EQUI 1920
#define DS_ADA_DIST 2432
#define DS_ADA_TOP (DS_NBLOCKS * DS_ADA_BLOCK_STRIDE) /* top-level dist_layer scale/shift */
#define DS_ADA_COLS (DS_ADA_TOP + 32)

/* Slots of the packed weight buffer: per-block slots first (DS_W_BLOCK_SLOTS per block), then globals.
 * "W" slots are MFMA-B-operand packed [K/8][2][Npad][4] (k = 8*kg + 4*half + s), zero padded;
 * "B" slots are plain bias vectors padded to Npad. */
enum ds_block_slot {
  DS_BW_EDGE_EMB_W = 0, DS_BW_EDGE_EMB_B,   /* 128 -> 64   rows: [x', rbf63, e64]            dmt.py:139 */
  DS_BW_E0_W, DS_BW_E1_W,                   /* 64 -> 252(256), 64 -> 256, no bias            layers.py:165,183 */
  DS_BW_QKV_W, DS_BW_QKV_B,                 /* 256 -> q252(256)|k252(256)|v256               layers.py:147-149 */
  DS_BW_N2E_W, DS_BW_N2E_B,                 /* 256 -> 64                                      dmt.py:157 */
  DS_BW_FF1_W, DS_BW_FF1_B, DS_BW_FF2_W, DS_BW_FF2_B,   /* 256->512->256                      dmt.py:114-116 */
  DS_BW_FF3_W, DS_BW_FF3_B, DS_BW_FF4_W, DS_BW_FF4_B,   /* 64->128->64                        dmt.py:118-120 */
  DS_BW_NODE_RO_W, DS_BW_NODE_RO_B,         /* 256 -> 64   node_i                            dmt.py:387 */
  DS_BW_EDGE_RO_W, DS_BW_EDGE_RO_B,         /* 64 -> 16(32) edge_i                           dmt.py:388 */
  DS_BW_AC_W,                               /* 256 -> 512  input_lin[:, 0:256] | [:, 256:512] dmt.py:39,45 */
  DS_BW_ED_W, DS_BW_ED_B,                   /* 128 -> 256  input_lin[:, 512:640] rows [e64, dist64] + bias */
  DS_BW_CM0_W, DS_BW_CM0_B,                 /* 256 -> 256  coord_mlp.0                        dmt.py:32 */
  DS_BW_CM2_W,                              /* 256 -> 3(32) coord_mlp.2 (no bias)             dmt.py:34 */
  DS_BW_RBF_MEAN, DS_BW_RBF_STD, DS_BW_RBF_ASTD, /* 63(64): mean, |std|+1e-5, a*std           layers.py:332-334 */
  DS_BW_COORD_SCALE,                        /* 1(32)       CoorsNorm.scale                    layers.py:347 */
  DS_BW_CM0_H,                              /* coord_mlp.0 as two fp16 planes (w = w1 + w2/2048) in f16-MFMA A-operand order:
                                               halves [plane 2][k/16 16][k-half 2][feature 256][8]; see k_equi_pairs */
  DS_BW_E0_H, DS_BW_E1_H,                   /* lin_edge0 / lin_edge1 (64 -> 256) TIMES 2 log2(e) in the same split-fp16 layout: k_attn_fused evaluates
                                               tanh(x) as 1 - 2 / (1 + exp2(2 log2(e) x)) and the factor rides in the weights (engine.TANH_PRESCALE) */
  DS_BW_ED_H,                               /* input_lin edge|dist part (128 -> 256) split-fp16 (k_edge_update) */
  DS_BW_QKV_H,                              /* q|k|v projection (256 -> 768) split-fp16 (k_node_qkv) */
  DS_BW_FF3_H,                              /* ff_linear3 (64 -> 128) split-fp16 (k_edge_update, transposed) */
  DS_BW_FF4_C,                              /* ff_linear4 (128 -> 64) split-fp16 in accumulator-chain order: halves [plane 2][hc 4][s 2][ft 2][lane 64][8],
                                               element j of lane (r, h) = W[ft*32 + r][hc*32 + 16 s + 8 (j>>2) + 4 h + (j&3)] */
  DS_BW_N2E_H, DS_BW_FF1_H, DS_BW_FF2_H, DS_BW_NODE_RO_H, DS_BW_AC_H,   /* the five GEMMs of k_node_update, split-fp16 */
  DS_BW_EDGE_EMB_H,                         /* edge_emb (128 -> 64, rows [x', rbf63, e64]) split-fp16 (k_edge_geom) */
  DS_W_BLOCK_SLOTS
};
enum ds_global_slot {
  DS_GW_SIN_W = 0,                          /* 8(32)       time_mlp.0.weights                 layers.py:285 */
  DS_GW_TM1_W, DS_GW_TM1_B,                 /* 17(24) -> 1024                                 dmt.py:254 */
  DS_GW_TM3_W, DS_GW_TM3_B,                 /* 1024 -> 1024                                   dmt.py:256 */
  DS_GW_ADA_W, DS_GW_ADA_B,                 /* 1024 -> DS_ADA_COLS, all *time_mlp Linears (dmt.py:23-26,102-109; layers.py:321-324);
                                               the weight in the split-fp16 layout of DS_BW_CM0_H: halves [2][64][2][DS_ADA_COLS][8] */
  DS_GW_NODE_EMB_W, DS_GW_NODE_EMB_B,       /* 12(16) -> 256                                  dmt.py:376 */
  DS_GW_EDGE_EMB_W, DS_GW_EDGE_EMB_B,       /* 68(72) -> 64  rows [edge_x2, cond_edge_x2, dist64]  dmt.py:373,377 */
  DS_GW_RBF_MEAN, DS_GW_RBF_STD, DS_GW_RBF_ASTD,
  DS_GW_NP0_W, DS_GW_NP0_B, DS_GW_NP2_W, DS_GW_NP2_B, DS_GW_NP4_W, DS_GW_NP4_B, /* 768->256->128->6(32) dmt.py:227-233 */
  DS_GW_EX0_W, DS_GW_EX0_B, DS_GW_EX2_W, DS_GW_EX2_B, DS_GW_EX4_W, DS_GW_EX4_B, /* edge_exist_mlp 192->64->32->1(32) */
  DS_GW_ET0_W, DS_GW_ET0_B, DS_GW_ET2_W, DS_GW_ET2_B, DS_GW_ET4_W, DS_GW_ET4_B, /* edge_type_mlp  192->64->32->1(32) */
  DS_GW_NP0_H, DS_GW_NP2_H,                 /* node_pred_mlp.0 (768 -> 256) and .2 (256 -> 128) in the split-fp16 layout (k_node_readout) */
  DS_GW_EX0_H, DS_GW_ET0_H,                 /* edge_exist_mlp.0 / edge_type_mlp.0 (192 -> 64) split-fp16 (k_edge_readout) */
  DS_GW_EX2_C, DS_GW_ET2_C,                 /* edge_exist_mlp.2 / edge_type_mlp.2 (64 -> 32) split-fp16 in the accumulator-chain order of DS_BW_FF4_C:
                                               halves [plane 2][hc 2][s 2][ft 1][lane 64][8] (k_edge_readout) */
  DS_W_GLOBAL_SLOTS
};
#define DS_W_NUM_SLOTS (DS_NBLOCKS * DS_W_BLOCK_SLOTS + DS_W_GLOBAL_SLOTS)

#define DS_MAX_ATOMS 29               /* QM9 (data.max_node, configs/diffspectra_qm9s.py:28) */

typedef struct ds_weights {
  const float* base;                 /* device: packed weights */
  const int64_t* off_dev;            /* device copy of off[] (read by the kernels) */
  int64_t off[DS_W_NUM_SLOTS];       /* float offsets of each slot into base (host copy) */
  float edge_th;                     /* model.edge_quan_th   (dmt.py:192) */
  float spatial_cut_off;             /* model.spatial_cut_off, compared with SQUARED distance (models/utils.py:118-126) */
} ds_weights;

typedef struct ds_layout {
  int32_t B, N, Nn, Pp;              /* molecules, padded atoms per molecule, packed node rows, packed pair rows */
  int32_t max_n;                     /* largest molecule (<= DS_MAX_ATOMS) */
  int32_t _pad;
  const int32_t* node_off;           /* [B+1] packed-node prefix */
  const int32_t* pair_off;           /* [B+1] packed-pair prefix */
  const int32_t* node_dense;         /* [Nn]  dense row b*N+i of packed node */
  const int32_t* node_mol;           /* [Nn]  molecule of packed node */
  const int32_t* pair_a;             /* [Pp]  packed node row of the smaller local index */
  const int32_t* pair_b;             /* [Pp]  packed node row of the larger local index */
  const int32_t* pair_mol;           /* [Pp] */
  const int32_t* mol_by_size;        /* [B][4] {node_off, n, pair_off, n(n-1)/2} of the molecules by descending size: workgroup i of a per-molecule
                                        kernel takes record i - one 16-byte load instead of a chain of dependent ones, and the tail of a launch is
                                        made of the small molecules (NULL: index order through node_off / pair_off) */
} ds_layout;

typedef struct ds_workspace {        /* all device fp32 unless noted; sizes in floats */
  float* pos;        /* [Nn,4]  xyz + pad */
  float* h;          /* [Nn,256] */
  float* e;          /* [Pp,64] */
  float* atom_hids;  /* [Nn,768] */
  float* edge_hids;  /* [Pp,192] */
  float* tfeat;      /* [B,24]   sinusoid features (17 used) */
  float* tmid;       /* [B,1024] */
  float* temb_silu;  /* [B,1024 floats]: SiLU(time_mlp(noise_level) + ctx) as two fp16 planes per row, halves [B][2][1024] (a = a1 + a2/2048) */
  float* ada;        /* [B,DS_ADA_COLS] */
  float* qkv;        /* [Nn,768] */
  float* ye;         /* [Pp,64 floats]: LayerNorm'd + modulated edge features of the current block (dmt.py:149) as two fp16 planes per
                        row, halves [Pp][2][64] (a = a1 + a2/2048): the MFMA operand from which k_attn_fused recomputes
                        tanh(lin_edge0 e) / tanh(lin_edge1 e) per molecule instead of streaming them through HBM */
  float* dist;       /* [Pp]     modulated squared distance x' of the current block (layers.py:330-331); its 64 CondGaussian
                        features are recomputed where they are consumed (k_edge_geom, k_edge_update) */
  float* attn;       /* [Nn,256] */
  float* u;          /* [Nn,64]  node2edge_lin weight applied per node (no bias) */
  float* ac;         /* [Nn,512] input_lin row part | col part */
  float* ed;         /* [Pp,256] input_lin edge+dist part + bias */
  float* lg;         /* [Pp,2,16] attention logits: [p][0] source a -> target b, [p][1] source b -> target a */
  float* tr;         /* [Pp,2,4] per-edge translation vectors of the current block: [p][0] a -> b, [p][1] b -> a */
  int32_t* adj;      /* [Pp]     bit0: cond_adj_2d, bit1: cond_adj_spatial */
  int32_t* flags;    /* [64]     0: any nonzero cond distance, 1: NaN in positions; [16..] diagnostic-build counters */
} ds_workspace;

/* sizeof() of ds_weights, ds_layout, ds_workspace, ds_gemm_args for the binding's layout self-check: out[0..3]. */
void ds_struct_sizes(int64_t* out);

/* Generic fp32-MFMA GEMM: C[M, N] = epilogue(A[M,K] * W + bias).  Wp packed as above with Kpad=ceil8(K),
 * Npad=ceil32(N).  act: 0 none, 1 SiLU, 2 GELU(erf), 3 tanh.  Optional: residual R added after act; per-column
 * affine (col_scale/col_shift, eval-mode BatchNorm) applied last; a_silu!=0 applies SiLU to A on load.
 * Row groups (grp_rows > 0): row r lives at (r / grp_rows) * grp_stride + (r % grp_rows) * ld — used for the
 * unfold view of spectra (specformer.py:105), token-buffer slices (:194) and positional tables (:183-188),
 * for which R row = r % r_grp_rows. */
typedef struct ds_gemm_args {
  const float* A; int64_t lda; int32_t a_grp_rows; int32_t _p0; int64_t a_grp_stride;
  const float* Wp; const float* bias;
  float* C; int64_t ldc; int32_t c_grp_rows; int32_t _p1; int64_t c_grp_stride;
  int32_t M, K, N, act;
  const float* R; int64_t ldr; int32_t r_grp_rows; int32_t a_silu;
  const float* col_scale; const float* col_shift;
} ds_gemm_args;
int ds_gemm(const ds_gemm_args* args, void* stream);

/* The split-fp16 GEMM that the block kernels are built from, as a stand-alone entry point (the per-step adaLN table GEMM uses
 * it; tests measure its accuracy against fp64 through it): C[M, N] = A * W + bias with fp32-level accuracy on the f16 matrix pipe.
 * Every operand value a travels as two fp16 numbers, a = a1 + a2/2048 (a1 = fp16(a) round-to-nearest, a2 = fp16((a - a1)*2048)),
 * the product is a1 b1 + (a1 b2 + a2 b1)/2048: three v_mfma_f32_32x32x16_f16 per 16-deep k-block, fp32 accumulate.
 *   A_split : device, halves [M][2][K] (plane 0 | plane 1 per row), K % 64 == 0
 *   W_split : device, the layout of engine.pack_linear_f16_split: halves [2][K/16][2][N][8], N % 32 == 0, as float* */
int ds_gemm_split(const void* A_split, const float* W_split, const float* bias, float* C, int64_t ldc, int32_t M, int32_t K,
                  int32_t N, void* stream);

/* One DMT evaluation (dmt.py:306-412).  xh [B,N,9], edge_x [B,N,N,2] dense; cond_x/cond_edge_x may be NULL
 * (first step, dmt.py:332-335); noise_level [B]; ctx_emb [B,1024] = cond_lin(SpecFormer(context)) (dmt.py:348-350),
 * NULL means zero context embedding.  out_xh [B,N,9], out_edge [B,N,N,2] are fully written (masked entries 0). */
int ds_forward(const ds_weights* w, const ds_layout* L, ds_workspace* ws,
               const float* xh, const float* edge_x, const float* cond_x, const float* cond_edge_x,
               const float* noise_level, const float* ctx_emb,
               float* out_xh, float* out_edge, void* stream);

/* ds_forward runs a block's node rows behind the attention (k_node_update) and the next block's q|k|v projection on a library-owned
 * side stream beside the pair rows' k_edge_update (fork / join by events; safe under stream capture).  on = 0: one stream (the order of
 * ds_stage_block), 1: two streams, -1: follow the environment variable DIFFSPECTRA_TWO_STREAM if set, else two streams for small
 * batches only (below ~2 500 molecules: there the side kernels fill launch tails; at the bench size they do not pay).  Returns the previous
 * setting.  Results are bit-identical in both modes.  (Build extension: the reference has no such switch.) */
int ds_set_two_stream(int on);

/* Stage-level entry points (used by the parity tests to localise a mismatch; same kernels ds_forward launches). */
int ds_stage_time(const ds_weights* w, const ds_layout* L, ds_workspace* ws, const float* noise_level,
                  const float* ctx_emb, void* stream);
int ds_stage_init(const ds_weights* w, const ds_layout* L, ds_workspace* ws, const float* xh, const float* edge_x,
                  const float* cond_x, const float* cond_edge_x, void* stream);
int ds_stage_block(const ds_weights* w, const ds_layout* L, ds_workspace* ws, int block, int last, void* stream);
int ds_stage_readout(const ds_weights* w, const ds_layout* L, ds_workspace* ws, float* out_xh, float* out_edge,
                     void* stream);

/* Ancestral update (sampling.py:604-624): x <- c_x*x + c_pred*pred + (sigma*noise)*temperature with the reference's noise
 * transforms fused (mask, CoM projection of position noise, tril(-1)+transpose edge noise: models/utils.py:67-106).
 * raw_pos [B,N,3], raw_feat [B,N,6], raw_edge [B,2,N,N] are the three randn draws.  x_mean/edge_mean receive the
 * noise-free posterior means (what sampling() returns after the last step). */
int ds_sampler_step(const ds_layout* L, float c_x, float c_pred, float sigma, float temperature,
                    float* x, float* edge_x, const float* pred, const float* edge_pred,
                    const float* raw_pos, const float* raw_feat, const float* raw_edge,
                    float* x_mean, float* edge_mean, void* stream);

/* The same two operations with the noise generated IN the kernel: counter-based Philox4x32-10 + Box-Muller, one stream per
 * molecule keyed on (seed, mol_id[m]) and indexed by (draw, atom / unordered atom pair) - never by the molecule's position
 * in the batch, the batch's padded width or the rank that owns it.  A sampling run therefore produces the same molecules
 * however it is cut into micro-batches and ranks, and the three randn launches + raw-noise tensors of the
 * reference's draw order disappear from the step.  Same distributions as models/utils.py:67-106: masked N(0,1), position
 * noise CoM-projected per molecule, edge noise one draw per unordered pair and channel written to both (i,j) and (j,i).
 *   draw 0           : ds_initial_noise  (z_T, edge_z_T of sampling.py:442-447)
 *   draw 1 + step    : ds_sampler_step_philox for denoise step `step` (sampling.py:611-612,623-624)
 * Philox counter = (element, draw, mol_id, kind) with kind 0: atom a, word block j -> element 3a + j (12 normals per atom,
 * 9 used: xyz, 5 type channels, charge); kind 1: pair lo < hi -> element hi(hi-1)/2 + lo (4 normals, 2 used).  Key = seed.
 * mol_id: device int64 [B].  x [B,N,9] / edge_x [B,N,N,2] are fully written by ds_initial_noise (masked entries 0). */
int ds_initial_noise(const ds_layout* L, uint64_t seed, const int64_t* mol_id, float* x, float* edge_x, void* stream);
int ds_sampler_step_philox(const ds_layout* L, float c_x, float c_pred, float sigma, float temperature,
                           uint64_t seed, int32_t step, const int64_t* mol_id,
                           float* x, float* edge_x, const float* pred, const float* edge_pred,
                           float* x_mean, float* edge_mean, void* stream);

/* Graph-replayable form of one denoise iteration: everything that changes from one iteration to the next
 * is read from device memory, so the launch sequence [ds_step_begin, ds_forward, ds_sampler_step_philox_dev] has constant
 * arguments and can be captured once (hipGraph) and replayed - small batches are otherwise bound by host launch work.
 *   table [S,4] device: (c_x, c_pred, sigma, noise_level) per step (sampling.py:572-584,604-606);
 *   step  device int32: ds_step_begin increments it (so it must hold i-1 before iteration i) and fills noise_level[0..B)
 *         with table[*step][3]; ds_sampler_step_philox_dev reads the coefficients and the Philox draw index from *step. */
int ds_step_begin(const float* table, int32_t n_steps, int32_t* step, int32_t B, float* noise_level, void* stream);
int ds_sampler_step_philox_dev(const ds_layout* L, const float* table, const int32_t* step, float temperature,
                               uint64_t seed, const int64_t* mol_id, float* x, float* edge_x, const float* pred,
                               const float* edge_pred, float* x_mean, float* edge_mean, void* stream);

/* post_process (sampling.py:53-97, compress_edge=True, centered=True, normalize_factors 1,4,4,1):
 * pos_out [B,N,3] f32, atom_type [B,N] i32 (argmax), fc [B,N] i32 (round(4*x)), edge_type [B,N,N] f32 in {0,1,2,3}. */
int ds_post_process(const ds_layout* L, const float* xh, const float* edge_x,
                    float* pos_out, int32_t* atom_type, int32_t* fc, float* edge_type, void* stream);

/* 3-D stability check of generated molecules (evaluation/stability.py:40-73 with evaluation/bond_analyze.py:5-45,85,90,
 * 108-133; QM9 atom set H, C, N, O, F): bond order of every atom pair from its distance in picometres (single if
 * 100 d < L1 + 10, then double if a double-bond length exists and 100 d < L2 + 5, then triple if 100 d < L3 + 3), valence
 * of every atom against allowed_bonds.  pos [B,N,3] Angstrom, atom_type [B,N] in 0..4 (the argmax ds_post_process wrote).
 * bond_order [B,N,N] i32 (may be NULL), nr_stable [B] = atoms with the right valence, mol_stable [B] = 1 if all are. */
int ds_check_stability(const ds_layout* L, const float* pos, const int32_t* atom_type, int32_t* bond_order,
                       int32_t* nr_stable, int32_t* mol_stable, void* stream);

/* SpecFormer pieces that are not plain GEMMs (specformer.py:385-425 residual-score attention; :119 LayerNorm).
 * qkv [B,L,3*heads*dk]; out [B,L,heads*dk]; scores: B*heads*L*L floats of caller-owned scratch that carries the
 * pre-softmax scores from layer to layer (has_prev = 0 on the first layer); its layout ([b][h][key][query]) is private. */
int ds_spec_attention(const float* qkv, float* scores, float* out, int B, int L, int heads, int dk, float scale,
                      int has_prev, void* stream);
int ds_layernorm_affine(const float* x, const float* gamma, const float* beta, float* y, int rows, int cols,
                        float eps, void* stream);

/* Measurement hook (bench.py roofline leg): time every `every`-th launch of one block-stage kernel with HIP events
 * recorded on the launch stream.  kernel: 0 edge_geom, 1 node_qkv, 2 attn_fused, 3 node_update, 4 edge_update,
 * 5 equi_pairs (6 is unused since the attention kernels were fused); kernel < 0 disables.  ds_profile_read synchronises the recorded events, returns the summed
 * duration and the sample count, and resets the counters.  Process-global instrumentation state; off by default. */
int ds_profile_config(int kernel, int every, int max_samples);
int ds_profile_read(double* total_ms, int64_t* samples);

/* Position-sensitive 64-bit checksum of a list of fp32 device tensors (wrap-around sum of bits(x) * odd(hash(index)) + index over
 * the concatenation): what the drop-in DMT.forward uses to notice that its packed weights are stale after an in-place edit that
 * leaves Tensor._version untouched (models/ema.py:55,77 write through .data).  ptrs [n] device array of device pointers,
 * prefix [n+1] device int64 element offsets of each tensor in the concatenation, out device uint64 (zeroed here). */
int ds_fingerprint(const float* const* ptrs, const int64_t* prefix, int32_t n, uint64_t* out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DIFFSPECTRA_HIP_H */
