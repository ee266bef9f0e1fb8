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
 * nothing keeps state between calls: all entry points are re-entrant and graph-capturable.
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
/* One-time re-layout of the (constant) codebook into MFMA fragment order, C % 32 == 0. */
int selftok_vq_pack_codebook(const float* codebook, float* packed, int C, int D, hipStream_t stream);
/* Same contract as selftok_vq_encode_f32 on the packed codebook (v_mfma_f32_32x32x2_f32 path). */
int selftok_vq_encode_packed_f32(const float* z, const float* packed, void* ids, float* best, void* workspace,
                                 int N, int C, int D, int flags, hipStream_t stream);

/* ---- code gather + LayerNorm(16) ----------------------------------------------------------
 * Replaces quantizer.get_output_from_indices (vector_quantize_pytorch.py:787-809) followed by
 * encoder.final_layer_norm3 (SelftokPipeline.py:236-240; models_ours.py:88).  out [n,16].
 * ln_w/ln_b NULL -> plain gather. */
int selftok_code_gather_ln_f32(const void* ids, const float* codebook, const float* ln_w, const float* ln_b,
                               float* out, int n, int C, int D, float eps, int flags, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SELFTOK_HIP_H */
