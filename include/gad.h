/* gad.h - C ABI of the MI355X (gfx950) hot-path library `libgad_hip.so`.
 *
 * The reference (q8888620002/Group-Attribution-for-Diffusion-Models) has no FFI of
 * its own: its hot path is reached through the Python API of the third-party
 * diffusers==0.24.0 wheel, which dispatches to ATen/cuDNN/cuBLAS kernels.  Each entry
 * point below names the reference call site whose device work it replaces
 * (paths relative to the reference root).  The Python host layer
 * (group-attribution-for-diffusion-models_amd/gad) binds these with ctypes; the binding a
 * reference maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to caller-owned memory (PyTorch allocates);
 *     the library never allocates, frees or retains device memory;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), never
 *     synchronises the device and is re-entrant / hipGraph-capturable;
 *   - returns 0 on success, non-zero on error; gad_last_error() gives the
 *     thread-local message;
 *   - activations are fp32 NHWC ([B][H][W][C], "pixel-major"), conv weights are
 *     [Cout][KH][KW][Cin] (= torch channels_last storage of the diffusers
 *     [Cout][Cin][KH][KW] parameter), Linear weights [out][in].
 */
#ifndef GAD_H
#define GAD_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int gad_version(void);
const char* gad_last_error(void);

/* ------------------------------------------------------------------------------
 * Contraction engine (f32-input MFMA v_mfma_f32_32x32x2_f32, exact fp32).
 * C[z][M][N] = epilogue( alpha * sum_k A[z](m,k) * B[z](k,n) )
 * Replaces cuDNN conv fwd/dgrad/wgrad and cuBLAS GEMMs under
 *   diffusers ResnetBlock2D / Downsample2D / Upsample2D / Attention
 *   (unconditional_generation/main.py:707,713; src/diffusion_utils.py:336-341;
 *    src/diffusers/models/attention_processor.py:1301-1329).
 * ---------------------------------------------------------------------------- */
enum gad_a_mode {
  GAD_A_KC = 0,     /* dense A[m][k], k contiguous, row stride lda                       */
  GAD_A_MC = 1,     /* dense A[k][m], m contiguous, row stride lda (i.e. A^T stored)    */
  GAD_A_CONV = 2,   /* im2col gather of NHWC x: m=(img,oh,ow), k=(r,s,c)   (conv fwd)   */
  GAD_A_CONVT = 3   /* transposed gather of NHWC dy: m=(img,ih,iw), k=(r,s,co) (dgrad)  */
};
enum gad_b_mode {
  GAD_B_KC = 0,     /* B[n][k], k contiguous, row stride ldb (torch Linear / conv weight)*/
  GAD_B_MC = 1,     /* B[k][n], n contiguous, row stride ldb                             */
  GAD_B_WDGRAD = 2, /* conv weight W[co][r][s][ci] read as B[k=(r,s,co)][n=ci]  (dgrad)  */
  GAD_B_CONV = 3    /* im2col gather of NHWC x as B[k=(img,oh,ow)][n=(r,s,c)]  (wgrad)   */
};

typedef struct gad_conv_geom {
  int32_t H, W, C;          /* gathered tensor: spatial size and channels (before upsample)   */
  int32_t ldx;              /* pixel stride of the gathered tensor in floats (>= C)          */
  int32_t Ho, Wo;           /* spatial size of the GEMM-row (A) / GEMM-k (B_CONV) pixel grid */
  int32_t KH, KW;
  int32_t stride;
  int32_t pad_t, pad_l;
  int32_t upsample;         /* 1: gathered tensor is nearest-2x upsampled on the fly         */
} gad_conv_geom;

typedef struct gad_gemm_args {
  const float* A;
  const float* B;
  float* C;
  int32_t a_mode, b_mode;
  int32_t M, N, K;
  int32_t lda, ldb, ldc;
  /* batch z = z0 * batch_inner + z1 ; pointer offset = z0*stride?0 + z1*stride?1 (floats) */
  int32_t batch, batch_inner;
  int64_t strideA0, strideA1, strideB0, strideB1, strideC0, strideC1;
  gad_conv_geom g;          /* used by the CONV / CONVT / WDGRAD / B_CONV modes              */
  /* epilogue: C = alpha*acc + bias[n] + rowadd[m / rows_per_group][n] + residual[m][n]      */
  float alpha;
  const float* bias;        /* [N] or NULL                                                   */
  const float* rowadd;      /* [M / rows_per_group][ld_rowadd] or NULL (time-embedding add)  */
  int32_t rows_per_group, ld_rowadd;
  const float* residual;    /* [M][ldr] or NULL (same batch strides as C)                    */
  int32_t ldr;
  /* split-K workspace (caller owned). ws_bytes >= gad_gemm_workspace_bytes(args)            */
  void* ws;
  int64_t ws_bytes;
  int32_t tile_hint;        /* 0 = auto, 1 = force 128x128, 2 = force 64x64, 3 = 128x64 (dense fp32 forms); 7 / 8 = a forward 3x3 convolution
                             * with B_wino / B_wino4 takes the F(2x2) / F(4x4) Winograd route whatever the planner models (tests, A/B tools); 3x3 weight gradient:
                             * 1 / 4 / 5 / 6 = 128 / 96 / 64 / 32 output channels per tile, 1000 + m1 = rows [0, m1) on 128-channel
                             * tiles and the rest planned, as a second launch (A/B tools) */
  int32_t splitk_hint;      /* 0 = auto, >0 = force                                          */
  /* 0: fp32 operands on v_mfma_f32_32x32x2_f32 (exact products; the reference's default precision).
   * 1: A and B may be rounded to bf16 (RNE) in flight and multiplied on v_mfma_f32_32x32x16_bf16 with fp32
   *    accumulation - the analogue of the reference's `--mixed_precision` autocast
   *    (text_to_image/train_text_to_image_lora.py:659-668) - whenever both operands can be read as 16-B aligned
   *    float4 (else the launch runs in fp32).  gad_gemm_uses_bf16() tells which. */
  int32_t operand_precision;
  /* A_CONV only: the gathered tensor is the channel concatenation of A ([..][ldx >= a_split], channels
   * [0, a_split)) and A2 ([..][ldx2], channels [a_split, g.C)) - UpBlock2D's torch.cat without the copy.
   * A2 == NULL: single source.  a_split must be a multiple of 32. */
  const float* A2;
  int32_t a_split, ldx2;
  /* optional bf16 copy of B (same [N][ldb] layout, k contiguous, RNE-rounded; 16-B aligned, ldb % 8 == 0): with
   * operand_precision = 1 the LDS-patch convolution streams it by LDS-DMA instead of converting the fp32 weights in every
   * workgroup.  Ignored by every other kernel. */
  const void* B_bf16;
  /* K-concatenated dense operands - the fused LoRA-compatible linear (diffusers LoRACompatibleLinear as injected at
   * text_to_image/train_text_to_image_lora.py:786-820): for k >= k_split the products read A_k2[m][k - k_split] and
   * B_k2 (laid out like B: [n][k - k_split] for B_KC, [k - k_split][n] for B_MC), so
   *   y = x W^T + mid (s B_up)^T            is ONE launch with A = x, A_k2 = mid, B = W, B_k2 = B_up   (forward)
   *   dx = dy W + dmid A_down               is ONE launch with A = dy, A_k2 = dmid, B = W, B_k2 = A_down (B_MC)
   * instead of a base GEMM plus a residual-accumulating side GEMM.  A_KC x B_KC|B_MC only; k_split % 32 == 0;
   * (K - k_split) % 4 == 0 (ragged LoRA ranks that are not multiples of 4 take the two-launch route). */
  const float* A_k2;
  const float* B_k2;
  int32_t lda_k2, ldb_k2, k_split;
  /* kernel-family switches for A/B tests and invariance checks (0 in production): GAD_GEMM_* bits below.  They
   * travel with the call - the library reads no environment variable and keeps no process-global switch. */
  int32_t flags;
  /* Winograd F(2x2, 3x3) route of the fp32 3x3 / stride 1 / pad 1 forward convolution (A_CONV x B_KC, even output maps,
   * Cin % 32 == 0, N % 4 == 0, N >= 64, single source, float4-addressable epilogue operands): B_wino = the transformed weights
   * U[16][Cout][Cin] = G w G^T made by gad_wino_weights from the [Cout][3][3][Cin] storage B points at; wino_ws = scratch
   * of gad_gemm_wino_bytes(args) bytes for the transformed input V[16][tiles][Cin].  B_wino == NULL (or a launch the
   * planner models no faster than the direct kernels: gad_gemm_wino_bytes returns 0): the direct kernels run and both are
   * ignored.  fp32 throughout; equal to the direct kernels up to fp32 reassociation / transform rounding, deterministic.
   * Replaces what cuDNN's own algorithm choice does for these convolutions in the reference (torch.nn.Conv2d of
   * diffusers' ResnetBlock2D.conv1 / conv2, Upsample2D.conv). */
  const float* B_wino;
  void* wino_ws;
  int64_t wino_ws_bytes;
  /* the F(4x4, 3x3) form of the same route (output maps that are multiples of 4): B_wino4 = U[36][Cout][Cin] made by
   * gad_wino4_weights.  With both forms given the planner takes whichever models fastest (or neither); wino_ws then holds the
   * transformed input AND the 36 product panels (gad_gemm_wino_bytes covers both), ws may be needed for a split of the batched
   * products (gad_gemm_workspace_bytes as usual).  4x fewer multiplies than the direct form; about one decimal digit less
   * accurate than the direct fp32 kernels (still ~4e-6 of the output scale).  Large launches run the one-launch form
   * (csrc/wino4_fused.hip: all 36 products and the output transform in one kernel, wino_ws holds V only); the planner decides,
   * tile_hint 9 / 10 force the one-launch / three-launch forms (tests, A/B tools). */
  const float* B_wino4;
} gad_gemm_args;
enum gad_gemm_flags {
  GAD_GEMM_NO_WINO = 16,      /* never take the Winograd route even when B_wino is given                              */
  GAD_GEMM_WINO_WGRAD = 32,   /* a 3x3 / stride 1 / pad 1 weight gradient (A_MC x B_CONV, output maps multiples of 4) may run in Winograd
                               * F(4x4, 3x3) form: dW = G^T [sum_tiles (A dy A^T) (.) (B^T x B)] G; wino_ws = gad_gemm_wino_bytes(args) bytes
                               * of scratch (transformed dy and x, the 36 product panels); the planner decides (tile_hint 8 forces).  With
                               * GAD_GEMM_WINO_SKIP_INPUT and B_wino4 = the V image [36][B Ho Wo / 16][Cin] the F(4x4) FORWARD launch of the same
                               * convolution left at the start of its wino_ws, the input is not transformed again (bit-identical result) */
  GAD_GEMM_NO_PATCH = 1,      /* never take the LDS-patch convolution kernels (generic im2col-gather engine instead) */
  GAD_GEMM_TAP_MAJOR_K = 2,   /* conv gathers walk K as (tap, channel chunk) instead of (channel chunk, tap)        */
  GAD_GEMM_SCALAR_EPILOGUE = 4, /* dword stores straight from the accumulators instead of the LDS-transposed float4 epilogue */
  GAD_GEMM_GENERAL_LOADERS = 8, /* dense operands with K % 32 == 0: the masking loaders instead of the lean (row-clamping) ones */
  /* Stages of a Winograd forward launch, so that a caller can put an event between them (bench.py's per-stage roofline) or
   * supply the transformed input itself: ONLY_INPUT runs the input transform x -> V into wino_ws and returns; SKIP_INPUT
   * takes wino_ws as already holding V (written by an ONLY_INPUT call with the same arguments) and runs the rest.  The two
   * calls back to back are the same kernels in the same order as one plain call: bit-identical. */
  GAD_GEMM_WINO_ONLY_INPUT = 64,
  GAD_GEMM_WINO_SKIP_INPUT = 128
};

int64_t gad_gemm_workspace_bytes(const gad_gemm_args* a);
/* which kernel instance gad_gemm would launch: row extent of the block tile (128 or 64 for the generic engine - 128 also
 * stands for its 128 x 64 form -, the channel tile 96 / 128 / 160 for the fp32 patch forward, 256 pixels for the vector-ALU
 * conv_out kernel), split-K factor, vector width (4 or 1) */
int gad_gemm_plan(const gad_gemm_args* a, int32_t* tile, int32_t* splitk, int32_t* vec);
int gad_gemm(const gad_gemm_args* a, void* stream);
int gad_gemm_uses_bf16(const gad_gemm_args* a);   /* 1 if gad_gemm(a) would multiply bf16-rounded operands */
/* which kernel family gad_gemm(a) launches: 0 gemm_kernel (fp32, im2col-gather / dense loaders), 1 gemm_bf16_kernel,
 * 2 conv3x3_patch_f32_kernel / wgrad3x3_patch_f32_kernel, 3 conv3x3_patch_bf16_kernel (3x3 / stride 1 / pad 1 convs whose
 * 128-pixel tiles are whole image rows: input patch resident in LDS), 4 conv3x3_fewout_kernel (<= 4 output channels:
 * vector ALUs, weights through the scalar cache) */
int gad_gemm_kernel_id(const gad_gemm_args* a);   /* ... 5 wino_input_kernel + wino_gemm_kernel (Winograd F(2x2, 3x3)), 6 wino4_input_kernel + wino4_gemm_kernel / batched gemm_kernel + wino4_output_kernel (F(4x4, 3x3)), 7 the F(4x4, 3x3) weight gradient
 * (wino4_dy_kernel + wino4_input_kernel + batched gemm_kernel + wino4_dw_kernel) */
/* bytes of wino_ws the Winograd route of gad_gemm(a) needs; 0 when gad_gemm(a) runs a direct kernel (set B_wino first) */
int64_t gad_gemm_wino_bytes(const gad_gemm_args* a);
/* Winograd weight transform U = G w G^T for the 3x3 weights listed in `table`: n_tiles rows of six int64
 * {src offset, dst offset, Cout, Cin, co0, ci0} (offsets in floats from src / dst; one 32 x 32 (co, ci) tile per row);
 * src storage [Cout][3][3][Cin] (diffusers Conv2d weight in channels_last: ResnetBlock2D.conv1/conv2, Up/Downsample2D.conv,
 * SURVEY Appendix A), dst storage [16][Cout][Cin].  One launch transforms every 3x3 weight of a flat parameter buffer. */
int gad_wino_weights(const float* src, float* dst, const int64_t* table, int64_t n_tiles, void* stream);
/* the same for F(4x4, 3x3): dst storage [36][Cout][Cin] */
int gad_wino4_weights(const float* src, float* dst, const int64_t* table, int64_t n_tiles, void* stream);

/* ------------------------------------------------------------------------------
 * GroupNorm (+ optional SiLU), NHWC.  Replaces ATen native_group_norm + SiLU in
 * ResnetBlock2D.norm1/norm2, Attention.group_norm, UNet2DModel.conv_norm_out
 * (diffusers; semantics SURVEY Appendix A.2/A.6; attention_processor.py:1297-1298).
 *   y = act( (x - mean_bg) * rstd_bg * gamma_c + beta_c ),  act = SiLU if silu != 0
 * mean/rstd [B][G] are written for the backward pass.
 * ws: gad_groupnorm_workspace_bytes() bytes of scratch.
 * ---------------------------------------------------------------------------- */
typedef struct gad_groupnorm_args {
  const float* x;           /* [B][HW][C]                                                    */
  float* y;                 /* fwd: output. bwd: dx                                          */
  const float* gamma;       /* [C]                                                           */
  const float* beta;        /* [C]                                                           */
  float* mean;              /* [B][G]  (fwd: out, bwd: in)                                   */
  float* rstd;              /* [B][G]                                                        */
  const float* dy;          /* bwd only: [B][HW][C]                                          */
  float* dgamma;            /* bwd only: [C] (overwritten); NULL together with dbeta: no affine gradients */
  float* dbeta;             /* bwd only: [C] (overwritten)                                   */
  int32_t B, HW, C, G;
  float eps;
  int32_t silu;
  void* ws;
  int64_t ws_bytes;
  /* forward only: channel-concatenated input without the concat (UpBlock2D's torch.cat([h, skip], 1),
   * SURVEY A.1): channels [0, C1) come from x ([B][HW][C1]) and [C1, C) from x2 ([B][HW][C-C1]).
   * x2 == NULL: single source.  C1 % 4 == 0 and (C - C1) % 4 == 0; both the one-pass and the two-pass plan read two sources. */
  const float* x2;
  int32_t C1;
  int32_t flags;            /* GAD_GN_TWO_PASS = 1: force the two-pass plan (A/B tests); 0 in production          */
  /* backward only: dx = (gradient through the norm) + dx_add ([B][HW][C], may alias nothing else): the gradient of a
   * second consumer of x - the residual branch of ResnetBlock2D / the attention block, whose input feeds both the norm
   * and the skip - summed in the store instead of by a separate elementwise launch.  NULL: none. */
  const float* dx_add;
} gad_groupnorm_args;
enum gad_groupnorm_flags { GAD_GN_TWO_PASS = 1 };

int64_t gad_groupnorm_workspace_bytes(const gad_groupnorm_args* a);
/* 1 if the forward of these shapes (incl. the x2/C1 split, if any) runs as the one-pass register-slab kernel */
int gad_groupnorm_one_pass(const gad_groupnorm_args* a);
int gad_groupnorm_silu_fwd(const gad_groupnorm_args* a, void* stream);
int gad_groupnorm_silu_bwd(const gad_groupnorm_args* a, void* stream);
/* Forward whose consumer is a 3x3 / stride-1 convolution on the Winograd F(4x4, 3x3) route of gad_gemm (ResnetBlock2D's
 * norm -> silu -> conv, reference diffusers ResnetBlock2D.forward): writes V[36][B * (H/4) * (W/4)][C] = B^T y_patch B - what
 * the route's own input transform would make of y - instead of y (a->y is ignored and may be NULL), plus mean / rstd.  The
 * convolution then runs with GAD_GEMM_WINO_SKIP_INPUT and wino_ws = V.  W = the map's width (HW = H * W, both multiples of 4),
 * C % 32 == 0, channels per group a multiple of 4; gad_groupnorm_wino4_ok tells whether a plan exists (the normalised
 * (image, channel slab) must fit 128 KB of LDS: up to 32 x 32 maps).  Same normalisation and transform arithmetic as the two
 * separate launches; the moments are reduced in the order of this kernel's channel slabs, so V equals theirs to fp32 rounding
 * (bit for bit where both cut the same slabs). */
int gad_groupnorm_wino4_ok(const gad_groupnorm_args* a, int32_t W);
int gad_groupnorm_silu_wino4(const gad_groupnorm_args* a, float* V, int32_t W, void* stream);

/* ------------------------------------------------------------------------------
 * Fused attention core: o = softmax(scale * q k^T) v per (batch, head), the score matrix never leaves the CU
 * (online softmax over streamed K/V tiles).  Replaces F.scaled_dot_product_attention of AttnProcessor2_0
 * (src/diffusers/models/attention_processor.py:1314-1325): UNet2DModel attention blocks (head dim 256 on CIFAR, 32 on
 * CelebA-HQ) and the self / cross attentions of the SD U-Net (head dims 40 / 80 / 160, Tk = Tq or 77;
 * text_to_image/train_text_to_image_lora.py:1268-1270).
 * Operands are [B][T][heads*d]-shaped views: row r of batch b, head h starts at base + b*stride + r*ld + h*d, so q, k, v
 * may be column blocks of one fused [B*T][3C] projection output (ld = 3C).  Any head dim 1 <= d <= 256
 * (gad_attention_supported).  Launches with d in {16, 24, 32, 40, 48, 64, 80, 96, 128, 160, 192, 224, 256}, every ld and
 * stride a multiple of 4 floats and every pointer 16-byte aligned stream their tiles by LDS-DMA; any other head dim or
 * alignment - the head-grouped-pruned CelebA-HQ model keeps its heads and shrinks the head dim 32 -> 23
 * (unconditional_generation/prune.py:337-342): rows of 322 floats - runs the next larger instance in its dword-staged
 * form (always exact fp32: gad_attention_uses_bf16 tells).
 * fwd writes o and, if lse != NULL, lse[b][h][q] = log2(sum_k exp2(scale*log2(e)*(q.k)))  (the statistics the backward
 * pass recomputes the probabilities from).  bwd needs q, k, v, o, lse, d_o and a caller-owned scratch `delta` of
 * B*heads*Tq floats; it writes dq, dk, dv (no atomics: every element is written once, bit-reproducibly).  Exact-fp32
 * launches up to d = 96 run ONE kernel per key block (S and dP computed once: the five products of the minimal scheme); key
 * blocks of one (b, h) leave dQ partials in `ws` and a fixed-order reduce sums them.  Wider heads and bf16-operand
 * launches run a dQ kernel + a dK/dV kernel that each recompute S and dP.
 * ---------------------------------------------------------------------------- */
typedef struct gad_attention_args {
  const float* q; const float* k; const float* v;
  float* o;                 /* fwd: out; bwd: in                                              */
  float* lse;               /* [B][heads][Tq]; fwd: out (may be NULL); bwd: in                */
  const float* d_o;         /* bwd: gradient w.r.t. o                                         */
  float* delta;             /* bwd scratch [B][heads][Tq]                                     */
  float* dq; float* dk; float* dv;
  int32_t B, heads, Tq, Tk, d;
  int32_t ldq, ldk, ldv, ldo, ld_do, ld_dq, ld_dk, ld_dv;           /* row strides (floats)  */
  int64_t stride_q, stride_k, stride_v, stride_o, stride_do, stride_dq, stride_dk, stride_dv;   /* batch strides */
  float scale;              /* 1/sqrt(d)                                                      */
  int32_t operand_precision;/* 0: exact fp32 products (v_mfma_f32_16x16x4_f32); 1: operands rounded to bf16 (RNE), fp32
                             * accumulation and softmax statistics (v_mfma_f32_16x16x32_bf16) - the autocast analogue */
  void* ws;                 /* bwd: caller-owned workspace of gad_attention_bwd_workspace_bytes(args) bytes (16-byte aligned)
                             * for the single-pass kernel's dQ partial slabs; NULL / too small: the dQ + dK/dV kernel pair runs */
  int64_t ws_bytes;
  int32_t flags;            /* GAD_ATTN_* */
} gad_attention_args;
#define GAD_ATTN_TWO_KERNEL_BWD 1   /* bwd: keep the recomputing dQ + dK/dV kernel pair (A/B tools, tests) */
#define GAD_ATTN_NARROW_FWD 2       /* fwd: keep the 4-wave kernel for d >= 160 instead of the 8-wave split-head-dim kernel (A/B) */
int64_t gad_attention_bwd_workspace_bytes(const gad_attention_args* a);
int gad_attention_supported(int32_t d);      /* 1 if 1 <= d <= 256 */
int gad_attention_uses_bf16(const gad_attention_args* a, int32_t backward);   /* 1 if this launch multiplies bf16 operands */
int gad_attention_fwd(const gad_attention_args* a, void* stream);
int gad_attention_bwd(const gad_attention_args* a, void* stream);

/* ------------------------------------------------------------------------------
 * Row softmax (attention probabilities), in place allowed.
 *   fwd: p = softmax(scale * s) per row of length n
 *   bwd: ds = scale * p * (dp - sum(dp * p))
 * Replaces the softmax inside F.scaled_dot_product_attention
 * (src/diffusers/models/attention_processor.py:1321-1323).
 * ---------------------------------------------------------------------------- */
int gad_softmax_fwd(const float* s, float* p, int64_t rows, int32_t n, float scale, void* stream);
int gad_softmax_bwd(const float* p, const float* dp, float* ds, int64_t rows, int32_t n, float scale,
                    void* stream);

/* ------------------------------------------------------------------------------
 * Transformer-block pieces of UNet2DConditionModel (Stable Diffusion; BasicTransformerBlock in diffusers,
 * reached from text_to_image/train_text_to_image_lora.py:1268-1270): LayerNorm over the last dim and GEGLU
 * (out = h[:, :F] * gelu(h[:, F:]), exact erf GELU).  dgamma_dbeta is [2C]: dgamma then dbeta (NULL: not wanted).
 * ---------------------------------------------------------------------------- */
int64_t gad_layernorm_workspace_bytes(int64_t rows, int32_t C);
int gad_layernorm_fwd(const float* x, float* y, const float* gamma, const float* beta, float* mean, float* rstd,
                      int64_t rows, int32_t C, float eps, void* stream);
/* dx_add (may be NULL): [rows][C] added to dx in the store - the gradient of the residual branch that shares x with the
 * norm (BasicTransformerBlock: x = attn(norm(x)) + x), instead of a separate elementwise launch */
int gad_layernorm_bwd(const float* x, const float* dy, float* dx, const float* dx_add, const float* gamma, const float* mean,
                      const float* rstd, float* dgamma_dbeta, int64_t rows, int32_t C, void* ws, int64_t ws_bytes,
                      void* stream);
int gad_geglu_fwd(const float* h, float* out, int64_t M, int32_t F, void* stream);
int gad_geglu_bwd(const float* h, const float* dout, float* dh, int64_t M, int32_t F, void* stream);

/* ------------------------------------------------------------------------------
 * Elementwise / small kernels (HBM-bound)
 * ---------------------------------------------------------------------------- */
/* diffusers get_timestep_embedding: out[b][dim] fp32; t int64 [B] */
int gad_timestep_embedding(const int64_t* t, float* out, int32_t B, int32_t dim, int32_t flip_sin_to_cos,
                           float freq_shift, float max_period, void* stream);
/* y = x*sigmoid(x) ; dx = dy * s(1 + x(1-s)) */
int gad_silu_fwd(const float* x, float* y, int64_t n, void* stream);
int gad_silu_bwd(const float* x, const float* dy, float* dx, int64_t n, void* stream);
/* Rotated copies of the 3x3 conv weights that live in one flat parameter buffer, in ONE launch:
 *   dst[off + ((ci*3 + 2-r)*3 + 2-s)*Cout + co] = src[off + ((co*3 + r)*3 + s)*Cin + ci]
 * i.e. W'[ci][2-r][2-s][co] = W[co][r][s][ci]: with W' the data gradient of a 3x3 / stride-1 / pad-1 convolution IS the
 * forward convolution of dy (same kernels, same channel tiles).  table (device, int64[n_tiles][5]) = {off, Cout, Cin,
 * co0, ci0}, one row per 32 x 32 (co, ci) tile of every listed weight; src != dst.  Refreshed once per optimizer step. */
int gad_rotate_conv3x3(const float* src, float* dst, const int64_t* table, int32_t n_tiles, void* stream);
/* out[p][0:C1] = a[p][0:C1], out[p][C1:C1+C2] = b[p][0:C2]  (skip-connection concat, NHWC) */
int gad_concat_channels(const float* a, const float* b, float* out, int64_t pixels, int32_t C1, int32_t C2,
                        void* stream);
/* inverse of concat for the gradient: da = d[:, :C1], db = d[:, C1:] */
int gad_split_channels(const float* d, float* da, float* db, int64_t pixels, int32_t C1, int32_t C2,
                       void* stream);
/* NCHW <-> NHWC layout change */
int gad_nchw_to_nhwc(const float* x, float* y, int32_t B, int32_t C, int32_t HW, void* stream);
int gad_nhwc_to_nchw(const float* x, float* y, int32_t B, int32_t C, int32_t HW, void* stream);
/* dx[b][h][w][c] = sum of the 2x2 block of dy[b][2h..2h+1][2w..2w+1][c] (nearest-upsample backward) */
int gad_upsample2x_bwd(const float* dy, float* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* out[s][n] = sum_m dy[s][m][n] for S segments of M rows (S=1: bias gradient; S=B, M=H*W: gradient of the
 * per-image time-embedding add) */
int gad_colsum(const float* dy, float* out, int32_t S, int64_t M, int32_t N, void* ws, int64_t ws_bytes,
               void* stream);

/* DDPMScheduler.add_noise (main.py:698): xt = sqrt(ac[t_b]) x0 + sqrt(1-ac[t_b]) eps ; per_sample = C*H*W */
int gad_add_noise(const float* x0, const float* eps, const int64_t* t, const float* alphas_cumprod,
                  float* xt, int32_t B, int64_t per_sample, void* stream);
/* DDIMScheduler.step, eta = 0 (src/diffusion_utils.py:336-341 via DDPMPipeline):
 *   x0 = (x - sqrt(1-a_t) e)/sqrt(a_t); clamp(+-clip) if clip>0; x' = sqrt(a_p) x0 + sqrt(1-a_p) e      */
int gad_ddim_step(const float* x, const float* eps, float* x_prev, int64_t n, float alpha_t,
                  float alpha_prev, float clip, void* stream);
/* classifier-free-guidance DDIM step (StableDiffusionPipeline loop driven at
 * text_to_image/compute_model_behaviors.py:311-326): eps = eps_u + guidance*(eps_c - eps_u), then the DDIM update.
 * eps_uc holds the U-Net output of the doubled batch: [uncond ; cond], each n floats. */
int gad_cfg_ddim_step(const float* x, const float* eps_uc, float* x_prev, int64_t n, float guidance, float alpha_t,
                      float alpha_prev, float clip, void* stream);
/* final pipeline post-processing: y = clamp(x/2 + 0.5, 0, 1) */
int gad_to_image01(const float* x, float* y, int64_t n, void* stream);
/* MSE loss and its gradient in one pass: loss[0] = mean((a-b)^2) ; d = 2 (a-b) * gscale / n */
int gad_mse_fwd_bwd(const float* a, const float* b, float* loss, float* d, int64_t n, float gscale,
                    void* ws, int64_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------
 * Fused optimizer over the flat parameter buffer
 * (clip_grad_norm_ main.py:718 + Adam/AdamW :557-561,719 + EMAModel.step :725).
 *   gad_sumsq: out[0] = sum(g^2)   (deterministic two-stage reduction)
 *   gad_clip_adam_ema: coef = min(1, max_norm/(sqrt(sumsq[0])+1e-6)); g*=coef;
 *     Adam(W) update with bias correction for `step`; ema -= (1-ema_decay)(ema-p) if ema != NULL
 * ---------------------------------------------------------------------------- */
int gad_sumsq(const float* g, float* out, int64_t n, void* ws, int64_t ws_bytes, void* stream);
typedef struct gad_adam_args {
  float* p; const float* g; float* m; float* v; float* ema;
  int64_t n;
  const float* sumsq;       /* device scalar from gad_sumsq, or NULL for no clipping          */
  float max_norm;
  float lr, beta1, beta2, eps, weight_decay;   /* weight_decay is decoupled (AdamW) if adamw  */
  int32_t adamw;
  int32_t step;             /* 1-based                                                        */
  float ema_decay;
} gad_adam_args;
int gad_clip_adam_ema(const gad_adam_args* a, void* stream);
/* standalone EMAModel.step (diffusers training_utils; main.py:725): ema -= (1-decay) * (ema - p) */
int gad_ema_update(float* ema, const float* p, int64_t n, float decay, void* stream);


/* ==============================================================================
 * Half-precision activation path ("--mixed_precision fp16|bf16" of the reference's Stable-Diffusion jobs:
 * text_to_image/experiments/setup_train_commands.py:127,142,165, setup_unlearn_commands.py:166; frozen weights cast to the
 * 16-bit type and autocast around the U-Net, text_to_image/train_text_to_image_lora.py:752-760,1268-1270).  bf16 is the
 * MI355X-native 16-bit type (fp32's exponent range: no loss scaling).  On this path ACTIVATIONS AND THEIR GRADIENTS LIVE IN
 * HBM AS bf16 (NHWC / [rows][C]); frozen weights are bf16 copies made once; LoRA A / B, their gradients and the optimizer
 * state stay fp32 (train_text_to_image_lora.py:777: "LoRA weights in fp32") with a bf16 shadow refreshed per step;
 * accumulation, normalisation statistics, softmax statistics and every epilogue are fp32.
 * All `void*` activation pointers below are bf16 (uint16 storage) unless a field says fp32.
 * ============================================================================== */
/* hgemm: C[m][n] = epilogue( alpha * sum_k A(m, k) * B[n][k] )   - v_mfma_f32_32x32x16_bf16, operands streamed by LDS-DMA.
 * B is always [n][k] with k contiguous (torch Linear / [Cout][KH][KW][Cin] conv storage; data gradients use transposed /
 * rotated bf16 copies of the frozen weights, made once).  A is
 *   conv == 0: dense A[m][k], row stride lda;
 *   conv == 1: the im2col gather of an NHWC tensor: m = (img, oy, ox), k = (r, s, c): pixel (oy*stride - pad_t + r,
 *              ox*stride - pad_l + s) of image img (>> 1 each if upsample: nearest-2x fused), zero outside;
 *   conv == 2: the gather of a stride-2 convolution's data gradient: m = (img, y, x) on the INPUT grid, k = (r, s, co) with
 *              the ROTATED weight: pixel ((y - pad_t + r) / 2, (x - pad_l + s) / 2) of dy where both are even, else zero.
 * The contraction axis may continue in a second operand pair: dense: k >= k_split reads A2[m][k - k_split] (row stride lda2)
 * and B2[n][k - k_split] (ldb2) - the fused LoRA linear (y = x W^T + mid up^T; dx = dy W + dmid down) -; conv: channels
 * c >= k_split of every tap come from A2 (pixel stride lda2) - UpBlock2D's torch.cat without the copy - while B stays one
 * [N][taps * C] matrix.  Every segment length and k_split must be a multiple of 8.
 * Epilogue (fp32): + bias[n] + rowadd[m / rows_per_group][n] (time-embedding add) + residual[m][n] (bf16), stored as bf16
 * (out_f32 == 0) or fp32 (out_f32 == 1; accumulate != 0 adds to what C holds: parameter-gradient slots).
 * ws: gad_hgemm_workspace_bytes(args) bytes for split-K partials (fp32 slabs reduced in a fixed order by a second launch).
 * Replaces cuDNN / cuBLAS half-precision kernels under the autocast U-Net (train_text_to_image_lora.py:1268-1270). */
typedef struct gad_hgemm_args {
  const void* A; const void* A2; const void* B; const void* B2;
  void* C;
  int32_t M, N, K;          /* K = total contraction length (conv: KH*KW*Cin)                                   */
  int32_t lda, lda2, ldb, ldb2, ldc;
  int32_t k_split;          /* dense: first segment's length (== K: single segment); conv: channels of source 1 */
  int32_t conv;             /* 0 / 1 / 2 as above                                                               */
  int32_t H, W, Cin;        /* gathered tensor (stored size), total channels                                    */
  int32_t Ho, Wo, KH, KW, stride, pad_t, pad_l, upsample;   /* Ho x Wo: the GEMM rows' pixel grid             */
  float alpha;
  const float* bias;        /* fp32 [N] or NULL                                                                 */
  const float* rowadd;      /* fp32 [M / rows_per_group][ld_rowadd] or NULL                                     */
  int32_t rows_per_group, ld_rowadd;
  const void* residual;     /* bf16 [M][ldr] or NULL                                                            */
  int32_t ldr;
  int32_t out_f32, accumulate;
  void* ws; int64_t ws_bytes;
  int32_t tile_hint;        /* 0 auto; 1 / 9 / 8: 128 x 128 (32x32x16 / 16x16x32 MFMAs / 32-deep steps); 2 / 7: 128 x 320; 5 / 6: 256 x 320, eight waves */
  int32_t splitk_hint;      /* 0 auto; > 0 force                                                                */
} gad_hgemm_args;
int64_t gad_hgemm_workspace_bytes(const gad_hgemm_args* a);
int gad_hgemm_plan(const gad_hgemm_args* a, int32_t* tile, int32_t* splitk);   /* tile: as tile_hint */
int gad_hgemm(const gad_hgemm_args* a, void* stream);
/* The token-axis contraction of the LoRA parameter gradients (dB = dy^T mid, dA = dmid^T x, G = dy^T x:
 * train_text_to_image_lora.py:1305 loss.backward() through LoRALinearLayer): C[m][n] = alpha * sum_k A[k][m] B[k][n], BOTH operands
 * stored with the contraction index k (the token) as their row - A: K rows of lda >= M, B: K rows of ldb >= N - so the activations
 * are read in place (transposing LDS reads), no transposed copies.  fp32 output only (out_f32 = 1; accumulate as in gad_hgemm);
 * lda, ldb multiples of 8 and >= M, N rounded up to 8 (rows are read in 16-byte chunks; columns >= M / N feed no output); of gad_hgemm_args only A, B, C, M, N, K, lda, ldb, ldc, alpha, accumulate, ws, splitk_hint are read. */
int64_t gad_hgemm_tn_workspace_bytes(const gad_hgemm_args* a);
int gad_hgemm_tn(const gad_hgemm_args* a, void* stream);
/* dst[c][r] = bf16(src[r][c]) for `batch` matrices (strides in elements); src fp32 (src_f32 != 0) or bf16; the operand
 * transposes of the LoRA parameter gradients (dUp = dy^T mid, dDown = dmid^T x) and of the per-step bf16 shadows of the
 * LoRA matrices' transposes */
int gad_h_transpose(const void* src, void* dst, int32_t rows, int32_t cols, int32_t ld_src, int32_t ld_dst, int32_t src_f32,
                    void* stream);
/* bf16 shadows of the 2-D matrices resident in a flat fp32 parameter buffer (the LoRA matrices of training.flatten_params), ONE launch
 * per optimizer step: dst[off + r*cols + c] = dst_t[off + c*rows + r] = bf16(src[off + r*cols + c]).  table (device, int64[n_tiles][5]) =
 * {off, rows, cols, r0, c0}: one 64 x 64 tile of one matrix per row.  The copies feed the forward products, the transposes the data
 * gradients (text_to_image/train_text_to_image_lora.py:1305 loss.backward() through LoRALinearLayer). */
int gad_h_shadow_pairs(const float* src, void* dst, void* dst_t, const int64_t* table, int32_t n_tiles, void* stream);
/* dst = bf16(src) (to_f32 == 0: src fp32) or dst = fp32(src) (to_f32 != 0: src bf16) */
int gad_h_cast(const void* src, void* dst, int64_t n, int32_t to_f32, void* stream);
/* GroupNorm (+ SiLU) with bf16 x / y / dy / dx, fp32 gamma / beta / mean / rstd: the fields of gad_groupnorm_args read as
 * bf16 where they are activations.  Forward reads two sources (x2 / C1) in place; backward computes dx only (frozen norm:
 * dgamma / dbeta must be NULL) and adds dx_add (bf16) in its store.  ws: gad_h_groupnorm_workspace_bytes(). */
int64_t gad_h_groupnorm_workspace_bytes(const gad_groupnorm_args* a);
int gad_h_groupnorm_silu_fwd(const gad_groupnorm_args* a, void* stream);
int gad_h_groupnorm_silu_bwd(const gad_groupnorm_args* a, void* stream);
/* LayerNorm over the last dim, bf16 x / y / dy / dx / dx_add, fp32 gamma / beta / mean / rstd; backward: dx only */
int gad_h_layernorm_fwd(const void* x, void* y, const float* gamma, const float* beta, float* mean, float* rstd, int64_t rows,
                        int32_t C, float eps, void* stream);
int gad_h_layernorm_bwd(const void* x, const void* dy, void* dx, const void* dx_add, const float* gamma, const float* mean,
                        const float* rstd, int64_t rows, int32_t C, void* stream);
/* GEGLU on bf16: out = h[:, :F] * gelu(h[:, F:]) and its backward */
int gad_h_geglu_fwd(const void* h, void* out, int64_t M, int32_t F, void* stream);
int gad_h_geglu_bwd(const void* h, const void* dout, void* dh, int64_t M, int32_t F, void* stream);
/* dx[b][h][w][c] = sum of the 2 x 2 block of dy (bf16 in / out, fp32 sum) */
int gad_h_upsample2x_bwd(const void* dy, void* dx, int32_t B, int32_t H, int32_t W, int32_t C, void* stream);
/* out = a + b (bf16, fp32 sum): skip-connection gradient sums the autograd engine would otherwise run as at::add */
int gad_h_add(const void* a, const void* b, void* out, int64_t n, void* stream);
/* Fused attention with bf16 q / k / v / o / d_o / dq / dk / dv (fp32 lse / delta): the fields of gad_attention_args read as
 * bf16, strides in elements; head dims that are multiples of 8 up to 160.  Same kernels as operand_precision = 1 of
 * gad_attention_fwd / _bwd with 16-bit tile loads and stores. */
int gad_h_attention_fwd(const gad_attention_args* a, void* stream);
int gad_h_attention_bwd(const gad_attention_args* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif
