/* prismer_hip.h -- C ABI of libprismer_hip.so: the MI355X (gfx950) operator set behind the Prismer
 * forward/backward hot path.
 *
 * The reference (NVlabs/prismer) has no plugin / FFI layer: its hot path is eager PyTorch
 * (model/modules/{vit,resampler,roberta,utils}.py).  The seam this library plugs into is therefore the ATen
 * operator level underneath those modules; each entry point names the reference call site(s) whose
 * arithmetic it replaces (paths relative to the reference root).  See INTEGRATION.md for the ctypes binding
 * and the nn.Module shells that keep the reference's parameter names.
 *
 * Conventions
 *   - plain C: raw device pointers, ints, floats and a hipStream_t; no C++ / torch types cross the ABI.
 *   - every function only ENQUEUES work on `stream`; it never allocates, frees or synchronises, keeps no
 *     pointer past the call and is safe under hipGraph stream capture.
 *   - activations / matrix weights are bf16 (row-major, contiguous unless a leading dimension is given);
 *     vectors (biases, LayerNorm / BatchNorm affine), statistics, gradients of parameters, optimizer state
 *     and losses are fp32.
 *   - return value: PH_OK (0) or a negative PH_ERR_* code; ph_last_error() returns a thread-local message.
 *     No exceptions, no abort().
 */
#ifndef PRISMER_HIP_H_
#define PRISMER_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP_PLATFORM_AMD__
typedef struct ihipStream_t* hipStream_t;
#endif

/* ABI revision: bumped whenever an argument struct or a signature changes (101: ph_conv_gather window fields, ph_gemm_args row map +
 * defer_reduce; 102: round 4; 103: round 5 -- ph_ce_fwd takes a row_loss scratch, no memset nodes anywhere; 104: round 6 -- ph_dense_minmax_partial,
 * ph_resize_remap_nchw_to_nhwc, ph_softmax_gather_bf16, new values of ph_gemm_tuning / ph_attention_tuning; 105: ph_store_words).  A host built against another revision must refuse to run: ph_version() != PH_VERSION. */
#define PH_VERSION 105

enum { PH_OK = 0, PH_ERR_BAD_ARG = -1, PH_ERR_UNSUPPORTED = -2, PH_ERR_LAUNCH = -3 };
enum { PH_ACT_NONE = 0, PH_ACT_QUICKGELU = 1, PH_ACT_RELU2 = 2, PH_ACT_GELU = 3, PH_ACT_RELU = 4,
       PH_ACT_SAVED_GRAD = 5 /* backward only: act_in already holds act'(x) (see pre_grad) */ };

int ph_version(void);
const char* ph_last_error(void);

/* ------------------------------------------------------------------------------------------------
 * GEMM + fused epilogue.   C[M,N] = epi( alpha * sum_k opA[m,k] * opB[n,k] )
 * Replaces: nn.Linear / F.linear everywhere (vit.py:41-47, resampler.py:18-24, utils.py:50-56,
 * roberta.py:86-92,134,163,177,415-421), the packed in-projection of nn.MultiheadAttention (vit.py:53,
 * resampler.py:31), convolutions after im2col (vit.py:86-120), and their autograd dgrad / wgrad.
 *   trans_x == 0: operand stored [rows][K] (K contiguous);  trans_x == 1: stored [K][rows].
 *   epilogue order: +bias -> (pre_out store) -> act | *act'(act_in) -> dropout -> +residual -> (+C) -> store
 * ---------------------------------------------------------------------------------------------- */
/* Implicit-GEMM convolution (vit.py:88-120): one operand is the im2col VIEW col[m][k] of an NHWC bf16 activation
 * x[B][H][W][C] (C % 8 == 0, 3x3 pad 1 or 1x1, any stride), m = (b, oy, ox), k = (ky*ks + kx)*C + c -- gathered inside the kernel's
 * operand loader, never written to memory.  Forward (trans_a = trans_b = 0): A = x, M = B*Ho*Wo, K = ks*ks*C rounded up to 8
 * (B = the [Cout][K] weight shadow).  Weight gradient (trans_a = trans_b = 1): B = x, reduction K = B*Ho*Wo, N = ks*ks*C rounded
 * up to 8 (A = dY [K][Cout]).  lda / ldb of the gathered operand are ignored. */
typedef struct {
  int B, H, W, C, ks, stride;
  /* round 3 -- generalised window (data gradients of the convolutions, gathered from dY): kh x kw taps, tap (ty, tx) reads pixel
   * (oy * stride + off_y + ty, ox * stride + off_x + tx), k = (ty * kw + tx) * C + c, output grid Ho x Wo.  kh == 0 (all six zero):
   * the square ks x ks window with pad ks / 2 described above.  Forward-shaped (A) gathers only. */
  int kh, kw, off_y, off_x, Ho, Wo;
} ph_conv_gather;
#define PH_COLSTAT_SLABS 8

typedef struct {
  const void* A; const void* B; void* C;
  int M, N, K;
  int lda, ldb, ldc;              /* in elements */
  int trans_a, trans_b;
  const float* bias;              /* [N] or NULL */
  int act;                        /* PH_ACT_* */
  void* pre_out;                  /* optional bf16 [M,N] (ld = ldc): value before the activation */
  const void* act_in; int ld_act; /* optional bf16: C = acc * act'(act_in)   (backward through `act`) */
  const void* residual; int ldr;  /* optional [M,N] added last: bf16, or fp32 when residual_f32 */
  float drop_p; const uint64_t* drop_seed; uint32_t drop_stream;   /* inverted dropout on the activation */
  int out_f32;                    /* C is fp32 instead of bf16 */
  int accumulate;                 /* C += result */
  float alpha;
  int split_k;                    /* 0 = auto */
  int residual_f32;
  void* workspace; int64_t workspace_bytes;   /* optional fp32 scratch for split-K partials (deterministic reduce + full
                                     epilogue); without it only plain fp32-accumulate GEMMs are split (atomics) */
  int pre_grad;                   /* pre_out receives act'(x) instead of x: the backward GEMM (act = PH_ACT_SAVED_GRAD) then
                                     multiplies by the saved derivative -- no transcendental in its epilogue, and the
                                     derivative is taken from the fp32 pre-activation instead of its bf16 rounding */
  const ph_conv_gather* conv;     /* optional: the A (forward) / B (weight gradient) operand is an im2col view, see above */
  double* col_stats;              /* optional fp64 [PH_COLSTAT_SLABS][2][N]: += per-column sum and sum of squares of the bf16-rounded outputs over the
                                     M rows (train-mode BatchNorm statistics of a conv output, vit.py:92-118, taken in the
                                     epilogue instead of a second pass over the output); plain epilogues, no split-K.  The
                                     accumulators are replicated PH_COLSTAT_SLABS times (a block adds to slab id % SLABS) to
                                     spread the atomics; consumers sum the slabs */
  /* round 3 -- output row map (rowmap_wo > 0): result row m is stored at C row  m * rowmap_mul - (m % rowmap_wo) * rowmap_sub +
   * rowmap_add  (the other row-indexed operands keep row m).  The data gradient of a stride-2 convolution is computed per parity
   * class of the input pixel: row m = (b, a, c) of class (py, px) is pixel (2a + py, 2c + px) = row 4m - 2(m % Wo) + py * W + px. */
  int rowmap_wo, rowmap_mul, rowmap_sub, rowmap_add;
  /* round 3 -- defer_reduce != 0: when the call is split over K, the pass that folds the partial sums (and applies the epilogue) is
   * not launched; it is queued and issued by ph_gemm_flush_deferred() together with those of the following deferred calls, <=
   * PH_GEMM_GROUP_MAX per launch (the stems' 24 conv weight gradients of a step: 24 fold launches -> 4).  The partials of
   * consecutive deferred calls are placed one after the other in `workspace` (pass the SAME workspace to all of them and to nothing
   * else until the flush; a call that does not fit flushes first).  C is only valid after the flush.  One queue per process:
   * deferred calls and their flush must come from one thread and one stream. */
  int defer_reduce;
} ph_gemm_args;
int ph_gemm_bf16(const ph_gemm_args* args, hipStream_t stream);
/* launches the queued fold passes of the deferred split-K GEMMs (no-op when nothing is queued) */
int ph_gemm_flush_deferred(hipStream_t stream);

/* Grouped GEMM: n <= PH_GEMM_GROUP_MAX independent problems with the SAME trans_a / trans_b in one launch (split_k is ignored;
 * round 3: a [K,M] x [K,N] group of plain fp32-output problems that all carry the same `workspace` and covers less than half of the
 * chip's block slots is ALSO split over K -- partial sums in the workspace, one grouped fold pass right behind the launch;
 * without a workspace there is no split).  Used for the weight gradients, which the reference's autograd emits as one
 * small GEMM per nn.Linear (dW = dY^T X, e.g. model/modules/roberta.py:79-183 has nine per decoder layer): their outputs
 * cover 18..144 tiles each, far fewer than the chip holds, so the host side defers them and issues each layer's set at once. */
#define PH_GEMM_GROUP_MAX 16
int ph_gemm_grouped_bf16(const ph_gemm_args* args, int n, hipStream_t stream);
/* tuning hook for benchmarks and tests: variant of the big-tile (256x128, LDS-DMA) kernel -- 0 = off, 1 = plain main loop, 5 / 6 = ping-pong
 * main loop with the LEAN tail (no surplus DMA, no drain; round 4), 7 = 6 with the LDS-DMA requests spread over the M phase (round 6, default) -- and the
 * tile count from which it is used (1 = every eligible launch, bypassing the dispatch cost model); a negative value leaves the setting unchanged.
 * The library reads no environment variable: the defaults (7, 128) are compiled in and this call is the only switch. */
int ph_gemm_tuning(int big_mode, int big_min_tiles);
/* same, as a BACKGROUND launch: at most `max_blocks` blocks (0 = one per tile), each walking several tiles.  Deferred weight
 * gradients issued beside the latency-bound backward chain of the decoder (roberta.py:212-231 in reverse) then fill the
 * idle CUs without taking every block slot from the chain's small kernels. */
int ph_gemm_grouped_capped_bf16(const ph_gemm_args* args, int n, int max_blocks, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * LayerNorm (fp32 math, eps inside the sqrt).  Replaces model/modules/utils.py:14-19 (F.layer_norm in
 * fp32 + casts) at every call site: vit.py:50-59,130-131,169-171; resampler.py:26-36; utils.py:57-64;
 * roberta.py:139,182,425.
 * Row mapping: logical row r of a mapped tensor lives at physical row
 *     (r / seg_in) * seg_out + seg_off + r % seg_in          (seg_in == 0 -> identity)
 * which lets LN write straight into the [latents ; x] concatenation of the resampler (resampler.py:34).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { int seg_in, seg_out, seg_off; } ph_rowmap;
typedef struct {
  const void* x;                  /* bf16 [M,D] */
  const float* gamma; const float* beta;
  void* y; ph_rowmap y_map;       /* bf16 */
  void* y2; ph_rowmap y2_map;     /* optional second copy of the output */
  float* mean; float* rstd;       /* fp32 [M], optional */
  int M, D; float eps;
  int x_f32;                      /* x is fp32 instead of bf16 (decoder residual stream) */
  void* y_f32;                    /* optional fp32 copy of the output [M,D] (identity row map) */
} ph_layernorm_fwd_args;
int ph_layernorm_fwd(const ph_layernorm_fwd_args* args, hipStream_t stream);
/* 0: always the one-row-per-wave forward kernel (A/B, tests); 1: default (bf16 rows of a multiple of 256 elements and M >= 1024 take the
 * half-wave-per-row kernel with 16-B vectors); < 0: query.  Returns the previous setting. */
int ph_layernorm_tuning(int fwd16);

typedef struct {
  const void* dy; ph_rowmap dy_map;      /* bf16 */
  const void* dy2; ph_rowmap dy2_map;    /* optional second upstream gradient, summed with dy */
  const void* x; const float* mean; const float* rstd; const float* gamma;
  const void* dskip;              /* optional bf16 [M,D]: gradient arriving through the residual branch */
  void* dx;                       /* bf16 [M,D] = LN'(dy) + dskip */
  void* dx_drop;                  /* optional bf16 [M,D] = dx * dropout-mask / (1-p) (mask of the fwd GEMM epilogue) */
  float drop_p; const uint64_t* drop_seed; uint32_t drop_stream;
  float* dgamma; float* dbeta;    /* fp32 [D], ACCUMULATED; NULL when the affine is frozen */
  int M, D;
  int x_f32;                      /* x is fp32 */
  float* partial_ws; int64_t partial_ws_bytes;   /* optional scratch (>= blocks*2*D*4 B): per-block partials + reduce instead of atomics */
  int defer_reduce;               /* != 0 (needs partial_ws): leave the per-block partials in partial_ws and do NOT fold them into
                                     dgamma/dbeta; the caller folds many LayerNorms at once with ph_ln_param_reduce_grouped */
} ph_layernorm_bwd_args;
int ph_layernorm_bwd(const ph_layernorm_bwd_args* args, hipStream_t stream);
/* number of partial rows ph_layernorm_bwd writes for M input rows: partial_ws holds [blocks][2][D] floats */
int ph_layernorm_bwd_blocks(int M);
/* dgamma[c] += sum_b ws[b][0][c], dbeta[c] += sum_b ws[b][1][c] for n <= PH_GEMM_GROUP_MAX deferred LayerNorm backwards */
typedef struct { const float* ws; int blocks, D; float* dgamma; float* dbeta; } ph_ln_reduce_item;
int ph_ln_param_reduce_grouped(const ph_ln_reduce_item* items, int n, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Fused multi-head attention (flash style: online softmax, scores never materialised).
 * Replaces: nn.MultiheadAttention core (vit.py:53; resampler.py:31) and RobertaSelfAttention
 * scores/mask/clamp/softmax/dropout/PV (roberta.py:101-126).
 * Token (b, t), head h, channel c of q lives at  q + b*q_bs + t*q_ts + h*dh + c   (elements).
 * key_mask[b*Sk + j] == 0 or (causal && j > i)  =>  score = finfo.min (finite, as roberta.py:113-115).
 * lse[(b*H+h)*Sq + i] = log-sum-exp of the scaled, masked scores (saved for backward).
 * One (batch, head) slice of q / k / v / dO must span < 2 GiB (Sq*q_ts, Sk*k_ts, Sk*v_ts < 2^30 elements): tiles are fetched
 * through buffer descriptors with 32-bit offsets; larger slices are rejected with PH_ERR.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; const void* k; const void* v; void* o;
  int64_t q_bs, q_ts, k_bs, k_ts, v_bs, v_ts, o_bs, o_ts;
  int B, H, Sq, Sk, dh;
  float scale;
  const uint8_t* key_mask; int causal;
  float drop_p; const uint64_t* drop_seed; uint32_t drop_stream;
  float* lse;
} ph_attn_fwd_args;
int ph_attention_fwd(const ph_attn_fwd_args* args, hipStream_t stream);

typedef struct {
  ph_attn_fwd_args f;             /* same tensors as forward (o = forward output, lse filled) */
  const void* d_o; int64_t do_bs, do_ts;
  void* dq; void* dk; void* dv;   /* bf16, same strides as q / k / v */
  int64_t dq_bs, dq_ts, dk_bs, dk_ts, dv_bs, dv_ts;
  float* delta;                   /* workspace fp32 [B*H*Sq] */
} ph_attn_bwd_args;
int ph_attention_bwd(const ph_attn_bwd_args* args, hipStream_t stream);
/* Kernel family selection (revision 103).  Launches with head dim 64, Sq <= 32 and Sk <= 320 -- the decoder's self- and cross-attention
 * (roberta.py:95-126 at T = 30 text tokens, 260 image tokens) -- run on the small-query kernels: one block per (batch, head), the keys split
 * over the waves, forward merged like split-K decoding, dQ + dK + dV in ONE launch (args.delta is not touched).  Plain launches (no causal cut,
 * key mask or dropout) with head dim 64 and at most 272 queries and keys -- the ViT blocks at 224^2 (vit.py:52-53) -- run on the head-resident
 * kernels (round 6): one block per (batch, head) with K / V (Q / dO) of the whole head staged once, dQ in one pass with the P / dP rows in
 * registers.  small_query_kernels = 0 forces the streaming kernels for every launch (A/B, tests), 1 restores the default, 2 = default without
 * the head-resident kernels, < 0 only queries.  Returns the previous setting. */
int ph_attention_tuning(int small_query_kernels);

/* ------------------------------------------------------------------------------------------------
 * Encoder front end (vit.py:86-160).
 * ---------------------------------------------------------------------------------------------- */
/* fp32 NCHW image -> bf16 patch matrix [B*g*g, Kp], column order (py, px, c) (the order ph_conv_weight_to_shadow
 * gives the Conv2d(3, D, p, stride=p) weight of vit.py:86,138), zero-padded to Kp. */
int ph_patchify(const float* img, void* col, int B, int C, int R, int p, int Kp, hipStream_t stream);
/* nn.UpsamplingBilinear2d (align_corners=True, vit.py:89,106) fused with NCHW fp32 -> NHWC bf16. */
int ph_resize_bilinear_nchw_to_nhwc(const float* x, void* y, int B, int C, int Hin, int Win, int Hout, int Wout,
                                    hipStream_t stream);
/* Dense-expert remap on the device (round 6).  Replaces the depth / normal / edge branch of post_label_process (dataset/utils.py:120-121:
 * `2 * (x - x.min()) / (x.max() - x.min() + eps) - 1`, eps = 1e-6, min / max over the whole [C, H, W] map of ONE sample) followed by
 * nn.UpsamplingBilinear2d (vit.py:88-90): the loader hands over the RAW expert maps [B, C, Hin, Win] fp32,
 *   ph_dense_minmax_partial        part[b][p] = (min, max) of the p-th of `nparts` contiguous shares of sample b (fp32 [B][nparts][2]),
 *   ph_resize_remap_nchw_to_nhwc   folds the pairs per sample and applies the remap to every bilinear tap in the reference's expression order
 *                                  (same taps / weights as ph_resize_bilinear_nchw_to_nhwc), NHWC bf16 out. */
int ph_dense_minmax_partial(const float* x, float* part, int B, int64_t n_per_sample, int nparts, hipStream_t stream);
int ph_resize_remap_nchw_to_nhwc(const float* x, const float* minmax_part, int nparts, void* y, int B, int C, int Hin, int Win, int Hout, int Wout,
                                 hipStream_t stream);
/* Label-expert in-painting fused with the stem's bilinear resize.  Replaces post_label_process (dataset/utils.py:117-160: the
 * per-label Python loop that paints CLIP text features over seg / obj_detection / ocr_detection label maps on the CPU, 64 fp32
 * channels per pixel) followed by nn.UpsamplingBilinear2d (vit.py:88-90, align_corners=True).
 *   labels [B, Hin, Win] uint8 (255 = background); table fp32 [256, C] per image (`table_batch_stride` elements apart; 0 = one
 *   table shared by the batch, e.g. the fixed COCO / ADE vocabularies): row l is the feature painted over label l.
 *   y [B, Hout, Wout, C] bf16 (NHWC) = bilinear(in-painted image), bit-identical to ph_resize_bilinear_nchw_to_nhwc of the
 *   dense image. */
int ph_inpaint_resize_nhwc(const uint8_t* labels, const float* table, int64_t table_batch_stride, void* y, int B, int C, int Hin,
                           int Win, int Hout, int Wout, hipStream_t stream);

/* 3x3 (pad 1) or 1x1 (pad 0) window gather from an NHWC bf16 map into col[B*Ho*Wo, Kp] with column order
 * (ky, kx, c); optionally applies BatchNorm(scale, shift per channel) + ReLU to every gathered element
 * (vit.py:90-103: conv -> BN -> ReLU -> conv, the normalised map is never written on its own). */
int ph_im2col_nhwc(const void* x, void* col, int B, int H, int W, int C, int ksize, int stride, int Kp,
                   const float* bn_scale, const float* bn_shift, hipStream_t stream);
/* adjoint of ph_im2col_nhwc: dx[b,y,x,c] = sum of dcol entries that read it (gather form, deterministic). */
int ph_col2im_nhwc(const void* dcol, void* dx, int B, int H, int W, int C, int ksize, int stride, int Kp,
                   hipStream_t stream);
/* BatchNorm2d training statistics over the rows of y[M,C] (vit.py:91-101): biased variance for normalisation,
 * running_mean/var update with momentum and UNBIASED variance, scale = gamma*rstd, shift = beta - mean*scale.
 * training == 0: scale/shift from the running statistics, nothing updated. */
int ph_bn_stats(const void* y, int M, int C, const float* gamma, const float* beta, float* running_mean,
                float* running_var, float momentum, float eps, int training, float* mean, float* rstd,
                float* scale, float* shift, int prezeroed, hipStream_t stream);
/* scale and shift must be ONE [2*C] block (shift == scale + C): it doubles as the reduction scratch.  prezeroed != 0: the
 * caller has already zeroed that block (one memset for all BatchNorm layers of a step instead of one per layer). */
/* BatchNorm + ReLU backward.  da = gradient w.r.t. relu(bn(y)).  Two kernels inside:
 * (1) dgamma += sum g*xhat, dbeta += sum g with g = da * [bn(y) > 0];  (2) dy = gamma*rstd*(g - dbeta/M - xhat*dgamma/M).
 * sums: fp32 workspace [2*C], zeroed by the call unless prezeroed != 0. */
int ph_bn_relu_bwd(const void* da, const void* y, void* dy, int M, int C, const float* gamma, const float* beta,
                   const float* mean, const float* rstd, float* dgamma, float* dbeta, float* sums, int prezeroed,
                   hipStream_t stream);

/* Grouped BatchNorm passes for the expert stems (vit.py:92-118: [conv3x3 -> BatchNorm2d -> ReLU] x 4 in each of up to six
 * independent stems): the same-index layers of all stems in ONE launch.
 *   ph_bn_apply_relu_grouped: a = relu(bn(y)); sums = fp64 [PH_COLSTAT_SLABS][2][C] per-channel sum / sum of squares of y (from the conv GEMM's
 *     epilogue, ph_gemm_args.col_stats: fp64 so that E[x^2] - E[x]^2 is exact to rounding and independent of the atomics' order); writes stats[4][C] = mean, rstd, scale, shift for the backward; train mode also updates
 *     running_mean / running_var (momentum, unbiased variance) like nn.BatchNorm2d.  Eval mode normalises with the running stats.
 *   ph_bn_relu_bwd_grouped: `a` carries dA (gradient w.r.t. the ReLU output), sums = fp32 [2][C], ZERO on entry (receives
 *     sum g, sum g*xhat); dy = gradient w.r.t. the conv output; dgamma / dbeta (optional) are accumulated. */
#define PH_BN_GROUP_MAX 8
typedef struct {
  const void* y; void* a; void* dy; int64_t M; int C;
  const float* gamma; const float* beta; float* running_mean; float* running_var;
  float* stats; void* sums; float* dgamma; float* dbeta;
} ph_bn_item;
int ph_bn_apply_relu_grouped(const ph_bn_item* items, int n, float momentum, float eps, int training, hipStream_t stream);
int ph_bn_relu_bwd_grouped(const ph_bn_item* items, int n, hipStream_t stream);
/* tokens[b, off + t, :] = feat[b*G + t, :] + pos[t, :] (+ inst_emb[table[inst[b, nearest(t)]], :])
 * (vit.py:141-159).  inst: int64 [B, E, E] instance map (nearest down-sampling to g x g), table: int32[256]. */
int ph_tokens_finalize(const void* feat, const float* pos, void* tokens, int B, int G, int D, int tok_per_batch,
                       int tok_off, const int64_t* inst, int E, int g, const int32_t* table, const float* inst_emb,
                       hipStream_t stream);
/* backward of the above: dfeat (bf16 [B*G, D]) = dtokens slice; dpos[t,:] += sum_b; dinst_emb[row,:] += ... */
int ph_tokens_finalize_bwd(const void* dtokens, void* dfeat, float* dpos, int B, int G, int D, int tok_per_batch,
                           int tok_off, const int64_t* inst, int E, int g, const int32_t* table, float* dinst_emb,
                           hipStream_t stream);
/* out[i,:] = sum_t w[i,t] * in[idx[i,t],:]  (bicubic positional-embedding re-grid, utils.py:34-44) and adjoint */
int ph_gather_taps(const float* in, float* out, const int32_t* idx, const float* w, int n_out, int taps, int D,
                   hipStream_t stream);
int ph_scatter_taps(const float* dout, float* din, const int32_t* idx, const float* w, int n_out, int taps, int D,
                    hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Decoder embeddings (roberta.py:38-45,66-76): position ids = cumsum(ids != pad)*(ids != pad) + pad,
 * word + token_type[0] + position -> LayerNorm -> dropout.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  const int64_t* ids; int B, T, H; int pad_id;
  const float* word; const float* pos; const float* type;     /* fp32 master tables */
  const float* gamma; const float* beta; float eps;
  void* out;                      /* bf16 [B*T, H] */
  void* xhat;                     /* bf16 [B*T, H] normalised pre-affine value, saved for backward */
  float* rstd;                    /* fp32 [B*T] */
  float drop_p; const uint64_t* drop_seed; uint32_t drop_stream;
  void* out_f32;                  /* optional fp32 copy of `out` */
} ph_embed_fwd_args;
int ph_embed_fwd(const ph_embed_fwd_args* args, hipStream_t stream);
typedef struct {
  ph_embed_fwd_args f;
  const void* dout;               /* bf16 [B*T, H] */
  float* dword; float* dpos; float* dtype; float* dgamma; float* dbeta;   /* fp32, accumulated with atomics */
} ph_embed_bwd_args;
int ph_embed_bwd(const ph_embed_bwd_args* args, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Shifted label-smoothed cross entropy over bf16 logits (roberta.py:381-387):
 * token (b,t) for t < T-1 is scored against labels[b,t+1]; ignore_index -100;
 * loss[b] = sum_t (1-eps)*nll + eps*mean_c(-log p_c).   Backward overwrites `logits` with dlogits (bf16):
 * dloss[b] * (softmax - (1-eps)*onehot - eps/V) for scored tokens, 0 elsewhere (incl. row T-1 and pad columns).
 * ---------------------------------------------------------------------------------------------- */
int ph_ce_fwd(const void* logits, int ld, const int64_t* labels, int B, int T, int V, float eps, float* loss,
              float* row_lse, float* row_loss /* fp32 [B*T] scratch, fully overwritten: per-token losses; loss[b] is their fixed-order
              sum (no atomics, no memset in front: a captured memset node does not replay correctly on ROCm 7.0, revision 103) */,
              hipStream_t stream);
int ph_ce_bwd(void* logits, int ld, const int64_t* labels, int B, int T, int V, int Vpad, float eps,
              const float* row_lse, const float* dloss, hipStream_t stream);
/* out[r][j] = softmax(logits[r, :V])[ids[j]] (fp32 [rows][n]) over bf16 logit rows `ld` elements apart: the first-token probabilities of the
 * answer candidates of inference='rank' (prismer_caption.py:70, prismer_vqa.py:51: softmax(dim=1).index_select(1, first tokens)). */
int ph_softmax_gather_bf16(const void* logits, int64_t ld, int rows, int V, const int64_t* ids, int n, float* out, hipStream_t stream);

/* ------------------------------------------------------------------------------------------------
 * Optimizer and small utilities.
 * ---------------------------------------------------------------------------------------------- */
/* torch.optim.AdamW step (train_caption.py:111-112,133) over a flat fp32 range; also refreshes the bf16 shadow.
 * lr and step are read from device memory (hipGraph replays see new values): hyper[0] = lr, hyper[1] = bias
 * correction1 = 1-b1^t, hyper[2] = bias correction2 = 1-b2^t. grad_scale multiplies g first (1/world). */
int ph_adamw(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const float* hyper, float beta1,
             float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, hipStream_t stream);
/* zero_grad != 0: g is overwritten with zeros after it has been read (the next step's optimizer.zero_grad(), without a
 * separate pass over the 1 GB gradient buffers) */
/* same with a keep bitmap (bit c of word c/32 set: the 1024 gradients [1024 c, 1024 c + 1024) are NOT zeroed): ranges whose
 * producer overwrites them in the next step -- the single-writer weight gradients of the native training step -- skip the
 * zero store here and the read-modify-write in the producing GEMM.  keep_bitmap == NULL: identical to ph_adamw. */
int ph_adamw_keep(float* p, float* g, float* m, float* v, void* p_bf16, int64_t n, const float* hyper, float beta1,
                  float beta2, float eps, float weight_decay, float grad_scale, int zero_grad, const uint32_t* keep_bitmap,
                  hipStream_t stream);
int ph_cast_f32_to_bf16(const float* x, void* y, int64_t n, hipStream_t stream);
int ph_cast_bf16_to_f32(const void* x, float* y, int64_t n, hipStream_t stream);
/* y = bf16(x * scale): the gradient pack of the bf16 exchange payload, pre-scaled by 1/world BEFORE the rounding (round 3) */
int ph_scale_cast_f32_to_bf16(const float* x, void* y, int64_t n, float scale, hipStream_t stream);
/* out[n] += sum_m x[m,n]  (bias gradients) */
int ph_colsum_bf16(const void* x, int M, int N, int ld, float* out, hipStream_t stream);
/* the same for n <= PH_GEMM_GROUP_MAX tensors in one launch (the bias gradients that go with ph_gemm_grouped_bf16) */
typedef struct { const void* x; float* out; int M, N, ld; } ph_colsum_item;
int ph_colsum_grouped_bf16(const ph_colsum_item* items, int n, hipStream_t stream);
/* dx = dy * act'(pre)  (bf16, n elements): backward through an activation that is not fused into a GEMM
 * (LM head: dense -> gelu -> LayerNorm, roberta.py:421-425) */
int ph_act_bwd_bf16(const void* dy, const void* pre, void* dx, int64_t n, int act, hipStream_t stream);
/* y = a + b (bf16, n elements) */
int ph_add_bf16(const void* a, const void* b, void* y, int64_t n, hipStream_t stream);
/* generic 2-D strided copy of bf16 rows: dst[r*ldd + c] = src[map(r)*lds + c], c < cols */
int ph_copy_rows_bf16(const void* src, int lds, ph_rowmap src_map, void* dst, int ldd, ph_rowmap dst_map, int rows,
                      int cols, int accumulate, hipStream_t stream);
/* dst[r, :cols] = src[idx[r], :cols] (bf16 rows, idx int32 on the device).  Beam reordering of the self-attention K/V caches
 * of KV-cached decoding -- what transformers' generate does with `_reorder_cache` after each beam step; the reference itself
 * (model/prismer_caption.py:45-50, roberta.py:401-406) keeps no cache and re-runs the whole prefix every step. */
int ph_gather_rows_bf16(const void* src, int64_t lds, const int32_t* idx, void* dst, int64_t ldd, int rows, int cols, hipStream_t stream);
/* conv weight layout changes: w[Cout,Cin,kh,kw] fp32 -> shadow bf16 [Cout, Kp] with column order (ky,kx,c) */
int ph_conv_weight_to_shadow(const float* w, void* shadow, int Cout, int Cin, int ks, int Kp, hipStream_t stream);
/* and the adjoint for gradients: dshadow fp32 [Cout,Kp] (ky,kx,c) -> dw[Cout,Cin,kh,kw] += */
int ph_conv_grad_from_shadow(const float* dshadow, float* dw, int Cout, int Cin, int ks, int Kp, hipStream_t stream);
/* the two conv layout passes for n <= PH_CONV_GROUP_MAX layers in one launch each (24 stem convs + 6 1x1 convs per step):
 * to_shadow: src = fp32 weight [Cout][Cin][ks][ks], dst = bf16 shadow [Cout][Kp];  from_shadow: src = fp32 dshadow, dst = fp32 dw (+=) */
#define PH_CONV_GROUP_MAX 32
typedef struct { const float* src; void* dst; int Cout, Cin, ks, Kp; } ph_conv_layout_item;
int ph_conv_weight_to_shadow_grouped(const ph_conv_layout_item* items, int n, hipStream_t stream);
int ph_conv_grad_from_shadow_grouped(const ph_conv_layout_item* items, int n, hipStream_t stream);
/* Weight operand of the IMPLICIT data gradient of a 3x3 (pad 1) convolution (round 3; replaces dcol = dY . W + col2im):
 *   dst bf16 [Cin][9 * Cout], dst[ci][t * Cout + co] = w[co][ci][ky(t)][kx(t)]
 *   stride 1: t = ty * 3 + tx, (ky, kx) = (2 - ty, 2 - tx): dX = conv3x3(dY, pad 1) with this matrix as the [N = Cin][K] operand
 *   stride 2: the taps of the four parity classes (py, px) of the input pixel back to back -- class (0,0): t = 0; (0,1): t = 1..2;
 *             (1,0): t = 3..4; (1,1): t = 5..8 -- tap (ty, tx) of a class reads dY[a + ty][c + tx] and stands for
 *             ky = py ? (ty ? 0 : 2) : 1, kx = px ? (tx ? 0 : 2) : 1.  Class GEMM: A = dY gathered with a (1 + py) x (1 + px)
 *             window at offset 0, stride 1; B = dst + toff * Cout (ldb = 9 * Cout, K = ntaps * Cout); C = dX through rowmap_*. */
typedef struct { const float* w; void* dst; int Cout, Cin, stride; } ph_conv_dgrad_item;
int ph_conv_dgrad_shadow_grouped(const ph_conv_dgrad_item* items, int n, hipStream_t stream);
/* launches per GEMM kernel class since the last reset (128x128, 64x64, intra-block k split, 256x128 single, 256x128 grouped,
 * grouped 128/64, split-K reduce): out[0..n-1]; returns the number of classes.  Lets a test assert which kernels a program ran. */
int ph_gemm_dispatch_counts(int64_t* out, int n, int reset);
/* Scratch sizing for hosts that own their buffers (`ph_query_workspace(op, dims) -> size_t`): bytes of the scratch /
 * workspace argument of entry point `op` for the given dimensions (the largest amount the entry point can make use of; every
 * workspace is optional or has this exact size).  Returns -1 for an unknown op or a wrong number of dims.
 *   PH_WS_GEMM_SPLITK     dims = {M, N, K}: ph_gemm_args.workspace (split-K partial tiles, up to 256 splits of [M][N rounded to 4] fp32)
 *   PH_WS_LAYERNORM_BWD   dims = {M, D}:    ph_layernorm_bwd_args.partial_ws ([ph_layernorm_bwd_blocks(M)][2][D] fp32)
 *   PH_WS_ATTENTION_BWD   dims = {B, H, Sq}: ph_attn_bwd_args.delta ([B*H*Sq] fp32)
 *   PH_WS_CONV_COLSTATS   dims = {N}:       ph_gemm_args.col_stats ([PH_COLSTAT_SLABS][2][N] fp64, zeroed by the caller) */
enum { PH_WS_GEMM_SPLITK = 0, PH_WS_LAYERNORM_BWD = 1, PH_WS_ATTENTION_BWD = 2, PH_WS_CONV_COLSTATS = 3 };
int64_t ph_query_workspace(int op, const int64_t* dims, int ndims);
/* per-step host scalars written by a kernel whose ARGUMENTS carry them (no copy engine, no pinned staging): dst0[0..n0) and dst1[0..n1)
 * (32-bit words, device memory; dst1 may be NULL with n1 = 0) <- host_words[0..n0+n1), read before the call returns.  Replaces the two
 * pinned host-to-device copies per step of the learning rate / Adam bias corrections (train_caption.py:127, torch.optim.AdamW's step
 * count) and the instance-embedding draws (vit.py:145-147).  n0 + n1 <= PH_STORE_WORDS_MAX. */
#define PH_STORE_WORDS_MAX 320
int ph_store_words(void* dst0, int n0, void* dst1, int n1, const uint32_t* host_words, hipStream_t stream);
/* advance the dropout seed (device-side, graph-replay safe): seed[0] = splitmix(seed[0]) */
int ph_advance_seed(uint64_t* seed, hipStream_t stream);
/* step glue (round 4: the last stock torch kernels inside the captured step).  x[i] += value for n int64 counters -- BatchNorm's
 * num_batches_tracked of all stem layers at once (torch/nn/modules/batchnorm.py semantics, vit.py:88-120) */
int ph_add_i64(int64_t* x, int n, int64_t value, hipStream_t stream);
/* zero fill and flat copy as kernels (16-B aligned, bytes % 16 == 0): the step's accumulator resets and the optimizer-sharding staging copies,
 * so that a captured segment contains no memset / memcpy node (revision 103) */
int ph_fill_zero(void* p, int64_t bytes, hipStream_t stream);
int ph_copy_bytes(void* dst, const void* src, int64_t bytes, hipStream_t stream);
/* out[0] = scale * sum_i x[i] * (weights ? weights[i] : 1): the batch loss -- caption loss.mean() (prismer_caption.py:33) with scale = 1/B,
 * VQA (weights * loss).mean() (prismer_vqa.py:40-41) */
int ph_weighted_sum_f32(const float* x, const float* weights, int n, float scale, float* out, hipStream_t stream);

/* Measurement hooks (bench.py): when enabled, every entry point brackets its launches with HIP events on the launch
 * stream.  ph_prof_collect synchronises and writes, per kernel family f (0 gemm, 1 layernorm, 2 attention fwd,
 * 3 attention bwd, 4 front end, 5 embed+CE, 6 optimizer, 7 misc), out[4f..4f+3] = {ms, algorithmic flops, algorithmic
 * bytes, launches}.  Must stay disabled during hipGraph capture. */
int ph_prof_enable(int on);
int ph_prof_collect(double* out);
int ph_prof_dump(const char* path);   /* CSV: family,ms,flops,description,GEMM kernel class (ph_gemm_dispatch_counts order; -1: none) per recorded call */

/* unit-test probe: exercises ds_read_b64_tr_b16 / MFMA lane layouts on the device (tests/test_kernels_gpu.py) */
int ph_probe_layouts(const void* in_bf16, float* out, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* PRISMER_HIP_H_ */
