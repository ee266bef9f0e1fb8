/*
 * selftok_hip.h -- C ABI of libselftok_hip.so, the MI355X (gfx950) kernels of the Selftok
 * encode/decode hot path.
 *
 * The reference (selftok-team/SelftokTokenizer) has no native/FFI layer: its operator API is the
 * Python class mimogpt.infer.SelftokPipeline, and every device op is a stock PyTorch call.  Each
 * entry point below therefore cites the reference *PyTorch call site* it replaces (file:line under
 * the reference tree); INTEGRATION.md shows the ctypes stub a maintainer would add at that site.
 *
 * Conventions: plain pointers to DEVICE memory owned by the caller (e.g. torch tensors'
 * data_ptr()), explicit sizes, a hipStream_t to launch on.  Every function returns 0 on success,
 * SELFTOK_EINVAL (-1) for a bad argument, SELFTOK_EHIP (-2) for a HIP launch error;
 * selftok_last_error() returns a thread-local message.  Nothing allocates, nothing synchronises,
 * nothing keeps state between calls (no environment variables are read; the only memo is the per-device
 * occupancy of a kernel, a constant): all entry points are re-entrant and graph-capturable.
 */
#ifndef SELFTOK_HIP_H
#define SELFTOK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define SELFTOK_OK 0
#define SELFTOK_EINVAL (-1)
#define SELFTOK_EHIP (-2)

/* flags shared by the VQ entry points */
#define SELFTOK_IDS_I32 1     /* ids are int32 (default: int64, the reference's dtype) */
#define SELFTOK_PRENORMED 2   /* z rows are already unit-norm: skip the fused l2norm */
/* launch-shape overrides of the packed (MFMA) path, for tests and tuning only -- ids never depend on them:
 * rows per wave tile (1, 2 or 4 x 32) and the number of code splits (1..64).  0 = choose from N and the device. */
#define SELFTOK_VQ_F16COARSE 8  /* packed path: approximate scores on the f16 matrix cores (3 MFMAs of the 16x-rate pipe per 32x32 scores),
                                  then canonical fp32 re-score of every candidate within the proven error window -> the SAME ids and
                                  top-1 score bits as the fp32 kernels (csrc/vq.hip, vq_f16_kernel) */
#define SELFTOK_VQ_F16COARSE1 16 /* with SELFTOK_VQ_F16COARSE: ONE MFMA per 32x32 scores (hi x hi only) and a 136x wider re-score window
                                   (17 x 2^-14, proven in csrc/vq.hip): 3x fewer MFMAs in the coarse pass, a longer exact re-score; same
                                   ids and score bits.  Pass the same flag to the partial pass and to the finalize. */
#define SELFTOK_VQ_RT(n) (((n) & 0xF) << 8)
#define SELFTOK_VQ_SPLIT(n) (((n) & 0xFF) << 16)

int selftok_version(void);
const char* selftok_last_error(void);

/* ---- VQ nearest-code lookup ---------------------------------------------------------------
 * Replaces VectorQuantize.forward's `x = l2norm(x)` + CosineSimCodebook.forward eval branch
 * (mimogpt/models/selftok/vector_quantize_pytorch.py:854, :561 einsum, :125-143 argmax/one_hot).
 * z [N,16] fp32 = output of project_in (:844); codebook [C,16] fp32 = _codebook.embed[0];
 * ids [N] int64 (or int32); best [N] top-1 score or NULL; workspace >= selftok_vq_workspace_bytes.
 * Bit-exact w.r.t. the reference CPU arithmetic, incl. ties (lowest index) and NaN (first NaN). */
size_t selftok_vq_workspace_bytes(int N, int C);
int selftok_vq_encode_f32(const float* z, const float* codebook, void* ids, float* best, void* workspace,
                          int N, int C, int D, int flags, hipStream_t stream);
/* One-time re-layout of the (constant) codebook into MFMA fragment order, C % 32 == 0.  `packed` must hold
 * selftok_vq_packed_bytes(C,D) bytes (= the fp32 image + one metadata line: "non-finite / out-of-fp16-range code present" flags +
 * the fp16 hi/lo image of the 2^7-scaled codebook used by SELFTOK_VQ_F16COARSE). */
size_t selftok_vq_packed_bytes(int C, int D);
int selftok_vq_pack_codebook(const float* codebook, float* packed, int C, int D, hipStream_t stream);
/* Same contract as selftok_vq_encode_f32 on the packed codebook (v_mfma_f32_32x32x2_f32 path). */
int selftok_vq_encode_packed_f32(const float* z, const float* packed, void* ids, float* best, void* workspace,
                                 int N, int C, int D, int flags, hipStream_t stream);

/* The two launches of selftok_vq_encode_packed_f32, separately callable (bench.py times the main kernel alone):
 * partial = per-(code split, wave half, row) candidates (orderable(best)<<32 | winning tile) in `workspace`;
 * finalize = max over candidates + exact slot recovery inside the winning tile (needs z and the packed codebook). */
int selftok_vq_argmax_partial_packed_f32(const float* z, const float* packed, void* workspace, int* nsplit_out,
                                         int N, int C, int D, int flags, hipStream_t stream);
int selftok_vq_finalize_packed(const void* workspace, const float* z, const float* packed, void* ids, float* best,
                               int N, int C, int D, int nsplit, int flags, hipStream_t stream);

/* ---- code gather + LayerNorm(16) ----------------------------------------------------------
 * Replaces quantizer.get_output_from_indices (vector_quantize_pytorch.py:787-809) followed by
 * encoder.final_layer_norm3 (SelftokPipeline.py:236-240; models_ours.py:88).  out [n,16].
 * ln_w/ln_b NULL -> plain gather. */
int selftok_code_gather_ln_f32(const void* ids, const float* codebook, const float* ln_w, const float* ln_b,
                               float* out, int n, int C, int D, float eps, int flags, hipStream_t stream);

/* ---- training-side codebook maintenance --------------------------------------
 * Replaces the one-hot contractions of CosineSimCodebook.forward in training mode (vector_quantize_pytorch.py:583-594:
 * `bins = embed_onehot.sum(1)`, `embed_sum = einsum('h n d, h n c -> h c d', flatten, embed_onehot)`): bins [C] and
 * embed_sum [C,16], zeroed by the caller, receive += 1 and += l2norm(z[row]) at ids[row] (z as for selftok_vq_encode_f32;
 * flags: SELFTOK_IDS_I32, SELFTOK_PRENORMED).  The caller all-reduces both over ranks (:588, :594) and applies the EMA. */
int selftok_vq_ema_accumulate_f32(const float* z, const void* ids, float* bins, float* embed_sum, int N, int C, int D, int flags, hipStream_t stream);
/* timestep_p_over_c [K,C] <- lerp(itself, mean_b one_hot(ids[b,k]), weight)  (:568-578, ema_inplace :66-72) with ids [B,K]
 * (the ids of ALL ranks: gather the ids instead of all-reducing the dense [K,C] mean); C % 4 == 0. */
int selftok_vq_tpc_update_f32(float* tpc, const void* ids, int B, int K, int C, float weight, int flags, hipStream_t stream);
/* Entropy regularisers of VectorQuantize.forward in training (vector_quantize_pytorch.py:1006-1031 with calc_entropy :89-100 and
 * calc_ema_entropy :109-118) without the [B, K, C] probability tensor.  With p = softmax_c(scale * <l2norm(z[b,k]), codebook[c]>)
 * (scale = 10, :1007), z [B, K, 16] as for selftok_vq_encode_f32 (flag SELFTOK_PRENORMED honoured), codebook [C, 16] unit-norm rows:
 *   rowstats [B*K, 2] = (1 / sum_c exp(logit), H(p[b,k,:]))                 -> entropy_to_min = mean of column 1
 *   colmean  [K, C]   = mean_b p[b, k, :]   (NULL: skipped)                  -> ap of calc_ema_entropy; its mean over k is calc_entropy's ap
 * selftok_vq_softmax_backward_f32: given g = dF/d(colmean) [K, C] for any scalar F of colmean (the caller differentiates the small
 * [K, C] epilogue), grad_z [B, K, 16] = dF/dz through the softmax, the scores and the l2norm (codebook detached, as
 * `embed.detach()` :559).  workspace: selftok_vq_softmax_workspace_bytes(B*K) bytes, 16-byte aligned, shared by both calls. */
size_t selftok_vq_softmax_workspace_bytes(int N);
int selftok_vq_softmax_stats_f32(const float* z, const float* codebook, float* rowstats, float* colmean, void* workspace, int B, int K, int C, int D,
                                 float scale, int flags, hipStream_t stream);
int selftok_vq_softmax_backward_f32(const float* z, const float* codebook, const float* rowstats, const float* g_colmean, float* grad_z, void* workspace,
                                    int B, int K, int C, int D, float scale, int flags, hipStream_t stream);

/* ---- fused residual + LayerNorm + adaLN modulate ---------------------------------------------
 *   x' = x + gate*y ;  n = LN(x') * (1 + scale) + shift          (LN: no affine, eps)
 * Replaces modulate()/gate() and the nn.LayerNorm calls around them:
 *   encoder  DualBlock.forward               mimogpt/models/selftok/modules.py:321-326 (+ :29-37)
 *   decoder  DismantledBlock.pre_attention   mimogpt/models/selftok/sd3/mmdit.py:472-483
 *            DismantledBlock.post_attention  mimogpt/models/selftok/sd3/mmdit.py:485-496
 *            FinalLayer.forward              mimogpt/models/selftok/sd3/mmdit.py:641-645
 * x,y,x_out,n_out: [B,T,H] fp32.  shift/scale/gate element (b,t,c) is read at
 * ptr + b*stride_b + t*stride_t + c  (per-token table: stride_b=0; per-sample table: stride_t=0).
 * y==NULL: no residual; n_out==NULL: residual only; shift==scale==NULL: plain LN; gate==NULL: x+y.
 * H in {64,256,512,1024,1536}. */
int selftok_residual_ln_mod_f32(const float* x, const float* y, const float* gate, const float* shift, const float* scale,
                                float* x_out, float* n_out, int B, int T, int H,
                                long mod_stride_b, long mod_stride_t, long gate_stride_b, long gate_stride_t,
                                float eps, hipStream_t stream);
/* Same, with n written as a "split activation" [B*T, H] (see selftok_linear_f16x2_split) for the f16x2 Linear that consumes it;
 * overflow (device int, may be NULL) bit 0 is OR-ed if |n| >= 65504. */
int selftok_residual_ln_mod_split(const float* x, const float* y, const float* gate, const float* shift, const float* scale,
                                  float* x_out, void* n_blk, int* overflow, int B, int T, int H,
                                  long mod_stride_b, long mod_stride_t, long gate_stride_b, long gate_stride_t,
                                  float eps, hipStream_t stream);

/* in-place h = gelu_tanh(h + bias); bias may be NULL.  Replaces Mlp.act after fc1
 * (sd3/other_impls.py:82-90; timm Mlp used at modules.py:109,293). */
int selftok_bias_gelu_f32(float* h, const float* bias, long rows, int cols, hipStream_t stream);
/* out = silu(in) (adaLN_modulation[0], TimestepEmbedder.mlp[1]: modules.py:297; sd3/mmdit.py:151,428). */
int selftok_silu_f32(const float* in, float* out, long n, hipStream_t stream);
/* out[b,:] = in[b,:] + table[:]  (per_sample floats per b): PatchEmbed bias + cropped pos-embed
 * (models_ours.py:211-214; sd3/mmdit.py:1000), context pos-embed (sd3/mmdit.py:1026). */
int selftok_add_rows_f32(const float* in, const float* table, float* out, int B, long per_sample, hipStream_t stream);
/* out[n,dim] = [cos(t*t_scale*f), sin(...)] ; freqs[dim/2] from the host
 * (TimestepEmbedder.timestep_embedding: models.py:56-74; sd3/mmdit.py:156-175). */
int selftok_timestep_embed_f32(const float* t, const float* freqs, float* out, int n, int dim, float t_scale, hipStream_t stream);
/* x [B,C,H,W] -> patches [B,(H/2)(W/2),4C], feature = c*4+p*2+q: PatchEmbed's k=2,s=2 conv as a GEMM
 * (sd3/mmdit.py:66-75). */
int selftok_patchify_f32(const float* x, float* out, int B, int C, int H, int W, hipStream_t stream);
/* fused unpatchify (sd3/mmdit.py:898-916) + CFG mix v=u+s(c-u) (sd3/rectified_flow.py:289) +
 * Euler step x_out = x - dt*v (sd3/rectified_flow.py:301-304).  y_* [B,hp*wp,4C]; y_uncond NULL: no CFG;
 * x_out NULL: only v_out; v_out NULL: only x_out. */
int selftok_unpatchify_cfg_euler_f32(const float* y_cond, const float* y_uncond, const float* x, float* x_out, float* v_out,
                                     int B, int C, int hp, int wp, float dt, float cfg_scale, hipStream_t stream);
/* RMSNorm (modules.py:73-95; only with qk_norm='rms', unused by the shipped configs). */
int selftok_rmsnorm_f32(const float* x, const float* w, float* out, long rows, int dim, float eps, hipStream_t stream);
/* rotary embedding: apply_rotary_emb(freqs, t, scale=) over the rotated slice (mimogpt/utils/rotary_embedding_torch.py:37-53; no call
 * site in the reference).  t, out [rows, dim], freqs [seq, dim], row r uses freqs[r % seq]; interleaved pairs. */
int selftok_rotary_f32(const float* t, const float* freqs, float* out, long rows, int seq, int dim, float scale, hipStream_t stream);

/* ---- fp32-equivalent Linear on the f16 matrix cores ("f16x2 split") ---------------------------------
 * out[M,N] = act(A[M,K] W[N,K]^T + bias[N]),  fp32 in / fp32 out; replaces the fp32 nn.Linear GEMMs of the MMDiT
 * blocks (qkv / proj / fc1 / fc2: sd3/mmdit.py:291-297, 485-496; sd3/other_impls.py:65-90; F.linear in the reference).
 * Every operand is split x = x0 + x1 2^-11 into two fp16 values and a.w ~= a0 w0 + 2^-11 (a0 w1 + a1 w0) runs on
 * v_mfma_f32_32x32x16_f16 with separate fp32 accumulators for the high and the low terms: error against an fp64
 * product is below that of an fp32 GEMM (tests/test_gemm_gpu.py).  The weight is split once into the kernel's tile
 * order (N % 128 == 0, K % 32 == 0; selftok_linear_f16x2_packed_bytes = 4 N K).  `overflow` (device int, may be NULL)
 * gets bit 0 OR-ed when an activation, bit 1 when a weight, is not below 65504 in magnitude (fp16 range): the result
 * is then invalid and the caller must use its fp32 GEMM.  A row stride lda (floats, multiple of 4), out row stride ldo. */
#define SELFTOK_LINEAR_GELU 1   /* act = GELU(tanh), the Mlp activation (sd3/other_impls.py:82-90) */
size_t selftok_linear_f16x2_packed_bytes(int N, int K);
int selftok_linear_f16x2_pack_weight(const float* W, void* packed, int N, int K, int* overflow, hipStream_t stream);
int selftok_linear_f16x2_f32(const float* A, long lda, const void* packed, const float* bias, float* out, long ldo,
                             int M, int N, int K, int flags, int* overflow, hipStream_t stream);
/* "Split activation": a [rows, K] fp32 tensor (K % 32 == 0) stored as the two fp16 planes of the split, hi = fp16(x),
 * lo = fp16((x - hi) 2^11) -- 4 bytes per element like fp32 -- in 1-KiB chunks of 16 rows x 32 k per plane:
 *     halfs index of element (row, k) of plane p (0 hi, 1 lo) = (((row/16) (K/32) + k/32) 2 + p) 512 + (row%16) 32 + k%32
 * (selftok_split_f16x2_bytes = ceil(rows/16) 16 K 4; rows of the last chunk beyond `rows` are never read for a result).
 * One chunk is one LDS-DMA piece of the consuming Linear.  A kernel that PRODUCES a Linear's input writes this form directly
 * (selftok_residual_ln_mod_split, selftok_attn_f32 with o_blk, the epilogue of selftok_linear_f16x2_split with out_blk), and
 * selftok_linear_f16x2_split stages its activation tiles by LDS-DMA like its weight tiles; results are bit-identical to
 * selftok_linear_f16x2_f32 on the fp32 tensor.  selftok_split_f16x2_f32 is the stand-alone producer.  Pointers 16-byte
 * aligned.  Output: either fp32 `out` (row stride ldo, out_blk = NULL) or a split activation [M, N] (out = NULL).
 * overflow bit 0 as above (raised by whichever kernel rounds a value beyond the fp16 range). */
size_t selftok_split_f16x2_bytes(long rows, int cols);
int selftok_split_f16x2_f32(const float* x, long ld, void* blk, long rows, int cols, int* overflow, hipStream_t stream);
int selftok_linear_f16x2_split(const void* a_blk, const void* packed, const float* bias, float* out, void* out_blk, long ldo,
                               int M, int N, int K, int flags, int* overflow, hipStream_t stream);
/* The same Linear with the residual update of DismantledBlock.post_attention / block_mixing fused into its epilogue
 * (x + gate_msa * attn.proj(...), x + gate_mlp * mlp(...): sd3/mmdit.py:485-496):
 *   out[r, c] = resid[r, c] + gate(r, c) * (A W^T + bias)[r, c],   gate(r, c) at gate + (r / T) gate_stride_b + (r % T) gate_stride_t + c
 * (gate NULL: out = resid + y).  Multiply and add are separate fp32 operations, i.e. the bits selftok_residual_ln_mod_f32 would
 * produce from the stored y; `out` may alias `resid`.  fp32 output only. */
int selftok_linear_f16x2_split_residual(const void* a_blk, const void* packed, const float* bias,
                                        const float* resid, long ldr, const float* gate, long gate_stride_b, long gate_stride_t, int T,
                                        float* out, long ldo, int M, int N, int K, int* overflow, hipStream_t stream);
/* Small-M forms of the two entry points above (one image: M = 256 image rows / <= 513 context rows -- the reference's own
 * configs[0] call, SelftokPipeline.py:224-291 with a batch of one).  A 256 x 128 tile grid then has 12 .. 96 work-groups on 256 CUs and
 * each walks all of K alone; here `ksplit` (2 .. 64, a divisor of K / 32) work-groups share an output tile, each over K / ksplit
 * consecutive k, their fp32 partial sums go to `workspace` (selftok_linear_f16x2_splitk_workspace_bytes = ksplit M N 4, 16-byte
 * aligned) and a second launch adds them in ascending k order (deterministic, independent of scheduling) and runs the same
 * epilogue.  The sum is then rounded at ksplit - 1 more places than the single-pass kernel's: results agree with it to fp32
 * rounding, not bit for bit (tests/test_gemm_gpu.py; the CPU twin models the order exactly).  ksplit == 1 forwards to the single-pass
 * entry point (workspace unused). */
size_t selftok_linear_f16x2_splitk_workspace_bytes(int M, int N, int ksplit);
int selftok_linear_f16x2_split_k(const void* a_blk, const void* packed, const float* bias, float* out, void* out_blk, long ldo,
                                 int M, int N, int K, int flags, int ksplit, void* workspace, int* overflow, hipStream_t stream);
int selftok_linear_f16x2_split_residual_k(const void* a_blk, const void* packed, const float* bias,
                                          const float* resid, long ldr, const float* gate, long gate_stride_b, long gate_stride_t, int T,
                                          float* out, long ldo, int M, int N, int K, int ksplit, void* workspace, int* overflow, hipStream_t stream);

/* ---- two-segment attention with implicit prefix-visibility mask ------------------------------
 * Replaces attention(q,k,v,heads,mask)=SDPA with a materialised bool mask (sd3/other_impls.py:37-45,
 * called from block_mixing sd3/mmdit.py:529-530; mask built at sd3/mmdit.py:1041-1094) and the SDPA calls of
 * DualAttention (modules.py:235-238, 263-266).  fp32 in/out on fp32-input MFMA (head_dim 64) or VALU (16).
 * Row r of a segment: ptr + b*bs + r*rs + head*head_dim.  seg[0] keys j visible iff j <= kvis[b] (kvis NULL: all);
 * seg[1] keys visible to seg[1] rows, and to seg[0] rows iff seg0_sees_seg1.  q==NULL: keys/values only.
 * seg[0] rows beyond kvis[b] are dead in the reference and are not written. */
#define SELFTOK_ATTN_F16X2 1
typedef struct selftok_attn_seg {
    const float* q; const float* k; const float* v; float* o;
    int len;
    long q_rs, k_rs, v_rs, o_rs;   /* row strides, floats */
    long q_bs, k_bs, v_bs, o_bs;   /* batch strides, floats */
} selftok_attn_seg;
typedef struct selftok_attn_desc {
    selftok_attn_seg seg[2];
    int B, H, head_dim;
    const int* kvis;
    int seg0_sees_seg1;
    float scale;
    int mode;            /* 0: fp32-input MFMA (exact fp32 products); SELFTOK_ATTN_F16X2: both contractions as f16x2-split
                            products on the f16 matrix cores (head_dim 64 only; see selftok_linear_f16x2_f32) */
    int* overflow;       /* f16x2 mode: device int, bit 2 is OR-ed if |q|, |k| or |v| >= 65504 (result invalid); may be NULL */
    void* o_blk[2];      /* f16x2 mode, per segment: if non-NULL the segment's output [B * len, H * head_dim] (row = b len + r) is written
                            as a "split activation" for selftok_linear_f16x2_split; seg.o may then be NULL */
} selftok_attn_desc;
int selftok_attn_f32(const selftok_attn_desc* desc, hipStream_t stream);

/* ---- SD3-VAE epilogues (bf16, NCHW) ------------------------------------------------------------
 * GroupNorm(groups,eps,affine)+SiLU (ResnetBlock/norm_out: sd3/sd3_impls.py:244-253,373-375,440-442). */
int selftok_groupnorm_silu_bf16(const void* x, const void* weight, const void* bias, void* out, int B, int C, int HW, int groups,
                                float eps, int apply_silu, hipStream_t stream);
/* `.mode()` (first c_keep of c_in channels) + SD3LatentFormat.process_in + .to(fp32)
 * (SelftokPipeline.py:215-218; sd3/sd3_impls.py:140-141). */
int selftok_latent_process_in(const void* moments_bf16, float* out, int B, int c_in, int c_keep, int HW, float shift, float scale, hipStream_t stream);
/* SD3LatentFormat.process_out + .to(bf16) (SelftokPipeline.py:285-287; sd3/sd3_impls.py:143-144). */
int selftok_latent_process_out(const float* z, void* out_bf16, long n, float shift, float scale, hipStream_t stream);
/* norm_ip(recons,-1,1) in place (SelftokPipeline.py:135-137,290). */
int selftok_clamp01_bf16(void* img, long n, hipStream_t stream);

/* ---- SD3-VAE convolutions and GroupNorm, channels-last (round 3; csrc/conv.hip) -----------------------------------------------
 * The bf16 convolutions of the VAE (ResnetBlock conv1/conv2/nin_shortcut, Downsample, Upsample, conv_in, conv_out:
 * sd3/sd3_impls.py:228-262, 286-318, 340-456) as one implicit-GEMM kernel on the bf16 matrix cores with the reference's CPU
 * arithmetic: fp32 accumulation of bf16 products, bias added inside it, ONE rounding to bf16.
 *   x [B, H, W, Cin] bf16 (Cin % 8 == 0), out / residual [B, Ho, Wo, ldo] bf16, channels [0, Cstore) written (Cstore % 4 == 0,
 *   Cout <= Cstore <= ldo; channels >= Cout are zero), bias [Cout] bf16 or NULL.
 *   ksize 3: padding 1; stride 2 (ksize 3 only) = Downsample's F.pad(x, (0,1,0,1)) + stride-2 convolution, Ho = H / 2.
 *   upsample 1: the input is read as its nearest-neighbour 2x upsample (Upsample.forward), Ho = 2 H; never materialised.
 *   residual: out = bf16(bf16(conv + bias) + residual), the two roundings of `x + h` (ResnetBlock.forward :262).
 * Weights: the checkpoint's [Cout, Cin, k, k] bf16 tensor packed once by selftok_conv2d_pack_weight_bf16 into an opaque image of
 * selftok_conv2d_packed_bytes(Cout, Cin, ksize, bn) bytes; bn = output channels per workgroup, 128 (Cout >= 64) or 32 (narrow
 * outputs: encoder conv_out, decoder conv_out); the same bn must be passed to the convolution. */
size_t selftok_conv2d_packed_bytes(int Cout, int Cin, int ksize, int bn);
int selftok_conv2d_pack_weight_bf16(const void* w, void* packed, int Cout, int Cin, int ksize, int bn, hipStream_t stream);
int selftok_conv2d_nhwc_bf16(const void* x, const void* packed, const void* bias, const void* residual, void* out, int B, int H, int W, int Cin, int Cout,
                             int Cstore, int ldo, int ksize, int stride, int upsample, int bn, hipStream_t stream);
/* GroupNorm(groups, eps, affine) [+ SiLU] on [B, HW, C] bf16 (channels-last): selftok_groupnorm_silu_bf16's arithmetic, statistics
 * accumulated in fp64 and reduced in a fixed order (deterministic).  C / 8 must divide 256, (C / groups) % 4 == 0. */
size_t selftok_groupnorm_nhwc_workspace_bytes(int B, int HW, int C);
int selftok_groupnorm_silu_nhwc_bf16(const void* x, const void* weight, const void* bias, void* out, void* workspace, int B, int HW, int C, int groups,
                                     float eps, int apply_silu, hipStream_t stream);

/* ---- SD3-VAE ENCODER in the reference's exact summation orders (round 4; csrc/vae_exact.hip) ------------------------------------
 * `vae.encode(images)[0].mode()` of the reference (SelftokPipeline.py:215; sd3/sd3_impls.py:221-377) runs in bf16 on the CPU and the
 * token ids depend on the exact rounding of every layer.  These entries evaluate every reduction as the SAME SEQUENCE of fp32
 * operations torch-CPU executes (oneDNN's AMX convolution, ATen's GroupNorm / SiLU / flash attention, glibc's expf) -- orders probed
 * in the build container, restated in oracle/vae_exact.c -- so that latents and token ids from pixels equal the reference's bit for
 * bit.  All tensors bf16 channels-last unless noted; fp32 MFMA rate (the order is prescribed, a bf16 MFMA has its own).
 *
 * Convolution.  x [B, H, W, ldx] (channels [0, Cin) used; ldx == Cin unless order 2), w [Cout, k, k, Cin] (the checkpoint's tensor
 * permuted, no packing), bias [Cout], out / residual [B, Ho, Wo, Cout].  ksize 3: padding 1; stride 2 (ksize 3) = Downsample's
 * F.pad(x, (0,1,0,1)) + stride-2 convolution.  residual: out = bf16(bf16(conv + bias) + residual).
 * order = the chunk order of oneDNN's kernel for that layer: 0: 32-channel chunks in (kh, kw, channel-block) order; 3: channel-block
 * major, every block's 9 taps summed privately and then added to the total (a stride-2 layer whose input is >= 102 pixels wide: the 128- and
 * 256-channel Downsample layers at 256 and 320 px, the 128-channel one at 128 px -- oneDNN decides by the layer's width, whatever H, B, C);
 * 1: channel-block major into the ONE running total (round 5: what oneDNN does once a 3x3 layer's bf16 input or output reaches 2^31 bytes -- the two
 * decoder layers around the [64, 256, 256, 256] activation when 64 images are decoded in one call, BASELINE configs[1]);
 * 2: conv_in (Cin = 3): one chunk of 27 elements in (kw, kh, ic) order.  Cin % 32 == 0, Cout % 32 == 0 (orders 0, 3); any row count
 * B*Ho*Wo (the last 64- / 128-row tile may be ragged: 40 x 40 = 1600 rows).
 * order | SELFTOK_VX_UPSAMPLE2X (round 5, the decoder's Upsample: F.interpolate(nearest, x2) + 3x3 convolution, sd3_impls.py:308-311): x is
 * [B, H/2, W/2, Cin] in memory and is read as its nearest-2x upsampled view of size H x W (H, W even, stride 1).  The decoder's layers all
 * use order 0 (tools/probe_cpu_bf16/check_decoder_convs.py); its conv_in (16 channels: one 16-channel chunk per tap) and conv_out (3 output
 * channels) are issued with zero-padded channels (a zero product leaves a chain's bits unchanged). */
#define SELFTOK_VX_UPSAMPLE2X 8
int selftok_vx_conv2d_bf16(const void* x, const void* w, const void* bias, const void* residual, void* out, int B, int H, int W, int ldx, int Cin, int Cout,
                           int ksize, int stride, int order, hipStream_t stream);
/* GroupNorm(groups, eps, affine) [+ SiLU] with ATen's statistics (Welford in 8 fp32 lanes over 16-element vectors, chunks of 16
 * vectors, binary cascade; elements in NCHW order) and y = bf16(fma(rstd * gamma, x, fma(-rstd * gamma, mean, beta))).
 * silu_table: 65536 bf16 entries from selftok_vx_silu_table_bf16, or NULL for no activation.  stats (may be NULL): [B, groups, 2] fp32
 * mean, rstd.  C % 128 == 0, H*W % 16 == 0, power-of-two channels per group, at most 512 cascade nodes per group (every layer of the VAE at
 * 128 / 256 / 320 px: 16 / 4 / 2 aligned chunks are pre-combined per thread where a channel holds a multiple of them, otherwise -- 80 x 80,
 * 16 x 16: an odd number of chunks per channel; 40 x 40: chunks straddle channels -- every chunk's two half-moments are stored and the whole
 * loop is replayed per group). */
size_t selftok_vx_groupnorm_workspace_bytes(int B, int HW, int C);
int selftok_vx_groupnorm_bf16(const void* x, const void* gamma, const void* beta, void* out, void* workspace, const void* silu_table, float* stats, int B, int HW,
                              int C, int groups, double eps, hipStream_t stream);
/* torch-CPU's `SiLU` on every bf16 bit pattern (it is a function of the input alone): table[bits(x)] = bits(silu(x)). */
int selftok_vx_silu_table_bf16(void* table, hipStream_t stream);
/* AttnBlock's scaled_dot_product_attention (sd3_impls.py:274-284), one head of C channels over T tokens, as ATen's CPU flash kernel
 * evaluates it (kv blocks of 512 keys, the last one shorter; running maximum / sum / accumulator rescaled at every block): q, k, v, out
 * [B, T, C] bf16, T % 32 == 0 (256 / 1024 / 1600 tokens = the VAE at 128 / 256 / 320 px), C % 128 == 0. */
size_t selftok_vx_attention_workspace_bytes(int B, int T, int C);
int selftok_vx_attention_bf16(const void* q, const void* k, const void* v, void* out, void* workspace, int B, int T, int C, hipStream_t stream);
/* glibc's expf (what `std::exp(float)` evaluates inside the flash kernel), element-wise; exposed for the parity tests. */
int selftok_vx_expf_f32(const float* x, float* y, long n, hipStream_t stream);

/* ---- fp32 Linear on the fp32-input matrix cores, both operands staged by LDS-DMA (round 6; csrc/gemm_fp32.hip) ----
 * out[m][n] = epilogue(sum_k x[m][k] w[n][k]): nn.Linear / F.linear of the MMDiT joint blocks (mimogpt/models/selftok/sd3/mmdit.py:266-307 qkv / proj,
 * :413-419 + sd3/other_impls.py:82-90 Mlp fc1 / fc2) and the wide Linears of the Q-Former (modules.py:186-199, 293).  x rows at stride ldx (16-byte aligned),
 * w [N][K] contiguous, N % 128 == 0, K % 32 == 0.  flags:
 *   SELFTOK_LINEAR_MKL_ORDER  the summation order of torch-CPU's MKL sgemm (K <= 384 or K >= 768: sequential fmaf chains per K-block of 384,
 *                             out = ((bias + c0) + c1) + ...): bit-identical to selftok_ex_linear_f32, the kernel of gemm='exact'.  Without it the order is
 *                             free: ONE k-ascending chain per output over the whole K (tail tiles: a few, see below), bias added last -- a candidate for gemm='fp32',
 *                             not wired in (the tuned library kernels win at 197 of the step's 204 shapes).
 *   SELFTOK_LINEAR_GELU       out = GELU_tanh(out), ATen / Sleef arithmetic (as SELFTOK_EX_GELU); a second launch over `out`: needs ldo == N and no res / gate
 *                             (the Mlp's fc1 -> act, sd3/other_impls.py:82-90, has neither), SELFTOK_EINVAL otherwise
 *   SELFTOK_LINEAR_BIAS_LAST  as SELFTOK_EX_BIAS_LAST
 *   SELFTOK_LINEAR_SPLIT(n)   tools / tests: force the tail split to n units (0: planned)
 * res / gate / res_mod / gate_mod / aliasing: as selftok_ex_linear_f32.
 * workspace: the tiles left over after the last full round of 256 tiles (one per CU and tile time) are computed as several K-range units whose raw sums go through
 * `workspace` and are added in K order by a second kernel (MKL order: one plane per K-block, so the result is unchanged bit for bit).  NULL / too small:
 * no split (correct, the tail round then runs at partial occupancy).  selftok_linear_f32_workspace_bytes = the most any plan for the shape takes. */
#define SELFTOK_LINEAR_BIAS_LAST 2
#define SELFTOK_LINEAR_MKL_ORDER 4
#define SELFTOK_LINEAR_SPLIT(n) (((n) & 0xFF) << 8)
size_t selftok_linear_f32_workspace_bytes(long M, int N, int K, int flags);
int selftok_linear_f32(const float* x, long ldx, const float* w, const float* bias, const float* res, long ldr, int res_mod, const float* gate, long ldg,
                       int gate_mod, float* out, long ldo, long M, int N, int K, int flags, void* workspace, size_t workspace_bytes, hipStream_t stream);

/* ---- the fp32 Q-Former encoder in the reference's exact summation orders (round 5; csrc/encoder_exact.hip, CPU twin oracle/encoder_exact.c) ----
 * Replaces, bit for bit, what torch-CPU executes for `Encoder.forward` (mimogpt/models/selftok/models_ours.py:204-257, 315-343) and
 * `DualBlock` / `DualAttention` (modules.py:165-327): every nn.Linear / F.linear (MKL sgemm: sequential fmaf chains per K-block of 384,
 * two halves for 384 < K < 768), the k2 s2 PatchEmbed convolution (one 64-tap chain = a Linear over the (kh, kw, ic)-ordered patch),
 * nn.LayerNorm (ATen RowwiseMoments), GELU(tanh) / SiLU (Sleef tanhf_u10 / expf_u10), F.scaled_dot_product_attention (ATen's fp32
 * cpu_flash_attention).  Every output row depends on its own input row only: results do not depend on the batch size.
 *
 * selftok_ex_linear_f32: out[m][n] = bias[n] + sum_k x[m][k] w[n][k]   (modules.py:109,186-199,293 ...; F.linear)
 *   x rows at stride ldx (a column slice of a fused projection is fine; 16-byte aligned), w [N][K] contiguous, K % 16 == 0.
 *   flags (`gelu`): bit 0: out = GELU_tanh(out) (timm Mlp fc1 -> act); bit 1 (SELFTOK_EX_BIAS_LAST): out = (sum of the K-blocks) + bias instead of
 *   ((bias + c0) + c1) + ... -- what at::linear computes for a NON-CONTIGUOUS input (matmul on a copy, then add_(bias)): the attention projections of the
 *   MMDiT's joint block read slices of the concatenated attention output (sd3/mmdit.py:537-541, 298-299).  res != NULL: out = res[row(m, res_mod)][n] + (gate ? gate[row(m, gate_mod)][n] * out : out)
 *   with the product and the sum rounded separately (`x + attn`, `q + gate(q_attn, g)`, modules.py:322-326); row(m, d) = m % d for d > 0 (per-token
 *   tables), m / -d for d < 0 (per-sample tables: -d rows per sample, sd3/mmdit.py:485-496 't_emb'), m for d == 0.
 *   out may alias res. */
#define SELFTOK_EX_GELU 1
#define SELFTOK_EX_BIAS_LAST 2
int selftok_ex_linear_f32(const float* x, long ldx, const float* w, const float* bias, const float* res, long ldr, int res_mod, const float* gate,
                          long ldg, int gate_mod, float* out, long ldo, long M, int N, int K, int gelu, hipStream_t stream);
/* nn.LayerNorm(N, eps) [affine gamma / beta or NULL] followed, when shift / scale are given, by the reference's modulate
 * `x * (1 + scale[tok]) + shift[tok]` (modules.py:29-32), tok = row % T (T > 0) or row / -T (T < 0: per-sample tables), table rows at stride ldt.  stats (may be NULL): [rows][2] mean, rstd. */
int selftok_ex_layernorm_mod_f32(const float* x, long ldx, float* out, long ldo, const float* shift, const float* scale, long ldt, int T, const float* gamma,
                                 const float* beta, float* stats, long rows, int N, float eps, hipStream_t stream);
/* The residual update of a DismantledBlock fused into the LayerNorm + modulate that follows it (round 6; sd3/mmdit.py:485-496):
 *   x' = x + gate[row(m, gate_mod)] * (lin + lin_bias)      (lin_bias / gate may be NULL; product and sums separately rounded, in this order)
 *   out = LayerNorm(x') * (1 + scale[tok]) + shift[tok]      (ATen's arithmetic, as selftok_ex_layernorm_mod_f32; no affine)
 * x' goes to x_out (may alias x).  `lin` is the plain output of the block's Linear (selftok_linear_f32 / selftok_ex_linear_f32 without res / gate):
 * the same bits as the Linear's `res + gate * y` epilogue followed by selftok_ex_layernorm_mod_f32. */
int selftok_ex_res_layernorm_mod_f32(const float* x, long ldx, const float* lin, long ldl, const float* lin_bias, const float* gate, long ldg, int gate_mod,
                                     float* x_out, long ldxo, float* out, long ldo, const float* shift, const float* scale, long ldt, int T, long rows, int N, float eps,
                                     hipStream_t stream);
/* element-wise: mode 0 GELU(tanh) (ATen GeluKernelImpl), 1 SiLU, and the building blocks 2 Sleef expf_u10, 3 Sleef tanhf_u10, 4 ATen exp_u20 */
int selftok_ex_unary_f32(const float* x, float* y, long n, int mode, hipStream_t stream);
/* F.scaled_dot_product_attention as ATen's fp32 flash kernel evaluates it.  q [B][Tq][..] rows at stride qs, head h = columns h*D .. h*D+D-1;
 * first key / value segment: Tk1 key SLOTS of which the first valid1 are visible, held in k1 / v1 [B][rows1][..] at stride kvs1 (rows1 >= valid1; the
 * encoder: valid1 == rows1 == Tk1 -- no mask; the MMDiT's prefix-visibility mask `arange(K) <= k`, sd3/mmdit.py:1041-1094: Tk1 = K, valid1 = k + 1 --
 * masked keys keep their position in the kv blocks of 512 and in MKL's K-blocks, contribute exp = 0 and 0 * v, and are never read); optional second
 * segment (Tk2 rows, stride kvs2) that follows the first (`torch.cat([k, query_k], dim=2)`, modules.py:250-251; the image tokens of the joint
 * attention, sd3/mmdit.py:519-536); out [B][Tq][H*D].  workspace >= selftok_ex_attention_workspace_bytes(B, H, Tq, Tk1 + Tk2, D). */
size_t selftok_ex_attention_workspace_bytes(int B, int H, int Tq, int Tk, int D);
int selftok_ex_attention_f32(const float* q, long qs, const float* k1, const float* v1, long kvs1, int Tk1, int valid1, int rows1, const float* k2, const float* v2,
                             long kvs2, int Tk2, float* out, void* workspace, int B, int H, int Tq, int D, hipStream_t stream);
/* selftok_ex_attention_f32 in ONE kernel (round 6): the scores are computed twice per kv block of 512 keys (once for the block maximum ATen takes before it
 * exponentiates, once for the probabilities) instead of being written to and re-read from a [B H, Tq, Tk] fp32 workspace; same bits.  head_dim 64, Tk1 % 64 == 0,
 * Tk2 % 64 == 0, (Tk1 + Tk2) % 512 == 0 or <= 384 (selftok_ex_attention_fused_supported != 0) -- every attention of the MMDiT and the Q-Former's query attention
 * at 256 x 256; other shapes (head_dim 16, the 320 px key counts): the entry above. */
int selftok_ex_attention_fused_supported(int Tk1, int Tk2, int D);
int selftok_ex_attention_fused_f32(const float* q, long qs, const float* k1, const float* v1, long kvs1, int Tk1, int valid1, int rows1, const float* k2, const float* v2,
                                   long kvs2, int Tk2, float* out, int B, int H, int Tq, int D, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SELFTOK_HIP_H */
