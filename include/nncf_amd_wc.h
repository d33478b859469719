/*
 * nncf_amd_wc.h -- C ABI of libnncf_amd_wc.so: the MI355X (gfx950) implementation of NNCF's
 * weight-compression lowering hot path.
 *
 * Conventions
 *  - every pointer argument is a DEVICE pointer unless its name ends in `_host`;
 *  - `stream` is a hipStream_t passed as void* (NULL = the default stream); calls are
 *    stream-ordered and asynchronous; no entry point allocates or synchronizes;
 *  - global state: read-only tables and the scratch array of nncf_amd_int_quant_error_static, nothing else: the
 *    multi-tensor (*_batched) kernels keep no counters or queues, so a call may be issued on any number of streams
 *    at once and recorded into a HIP graph and replayed;
 *  - the caller owns every buffer; "nullable" outputs are skipped when NULL;
 *  - return value: NNCF_AMD_OK (0) or a negative nncf_amd_status; nncf_amd_last_error()
 *    gives a thread-local message for the last failure;
 *  - a weight is described as a contiguous [outer, r, inner] array that is reduced along `r`
 *    in groups of `group_size` (group_size == -1 means one group of length r).  This is the
 *    reference's `reshape_weight_for_grouped_quantization` view
 *    (src/nncf/quantization/algorithms/weight_compression/weight_lowering.py:57-81);
 *    torch nn.Linear weights [out, in] reduced along the last axis are (outer=out, r=in, inner=1).
 *    Per-group parameters (scale, zero point) are laid out [outer, r/group_size, inner], i.e. the
 *    reference's keepdims shape [outer, r/gs, 1, inner] flattened.
 *
 * Reference paths are relative to the nncf 3.2.0 checkout; "WL" abbreviates
 * src/nncf/quantization/algorithms/weight_compression/weight_lowering.py.
 */
#ifndef NNCF_AMD_WC_H
#define NNCF_AMD_WC_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    NNCF_AMD_OK = 0,
    NNCF_AMD_ERR_BAD_ARG = -1,    /* null/negative/inconsistent argument                       */
    NNCF_AMD_ERR_GROUP_SIZE = -2, /* r % group_size != 0  (nncf.errors.InvalidGroupSizeError)  */
    NNCF_AMD_ERR_DTYPE = -3,      /* unsupported dtype enum                                     */
    NNCF_AMD_ERR_MODE = -4,       /* unsupported mode enum for this entry point                 */
    NNCF_AMD_ERR_LAUNCH = -5,     /* hipGetLastError() != hipSuccess after a launch             */
    NNCF_AMD_ERR_WORKSPACE = -6,  /* workspace missing or too small                             */
    NNCF_AMD_ERR_UNSUPPORTED = -7 /* valid request this build does not implement (caller error) */
} nncf_amd_status;

typedef enum { NNCF_AMD_F32 = 0, NNCF_AMD_F16 = 1, NNCF_AMD_BF16 = 2 } nncf_amd_dtype;

/* CompressWeightsMode subset on the path (src/nncf/parameters.py:73-120) */
typedef enum {
    NNCF_AMD_INT8_ASYM = 0,
    NNCF_AMD_INT8_SYM = 1,
    NNCF_AMD_INT4_ASYM = 2,
    NNCF_AMD_INT4_SYM = 3,
    NNCF_AMD_NF4 = 4,
    NNCF_AMD_MXFP4 = 5, /* e2m1 values, E8M0 power-of-two scale */
    NNCF_AMD_FP4 = 6    /* e2m1 values, fp32 scale              */
} nncf_amd_mode;

typedef enum { NNCF_AMD_ZP_NONE = 0, NNCF_AMD_ZP_I32 = 1, NNCF_AMD_ZP_F32 = 2 } nncf_amd_zp_dtype;
typedef enum { NNCF_AMD_ERR_MAX_MEAN = 0, NNCF_AMD_ERR_FROBENIUS = 1 } nncf_amd_error_reduction;

/* Library identity / diagnostics. */
int nncf_amd_version(void);               /* major*10000 + minor*100 + patch */
const char* nncf_amd_last_error(void);    /* thread-local, never NULL         */
const char* nncf_amd_status_string(int status);

/*
 * Fused integer lowering: group min/max -> scale[, zero point] -> round -> clip [-> nibble pack] [-> dequantize].
 * Replaces, in one pass over the weight:
 *   calculate_integer_quantization_params   WL:292-326  (asym: quantization/fake_quantize.py:322-350,
 *                                                         sym: _calculate_signed_scale WL:629-649)
 *   _calculate_integer_quantized_weight     WL:652-684
 *   do_integer_quantization                 WL:451-508
 *   do_integer_dequantization               WL:428-448   (when dq_out != NULL -> integer_quantize_dequantize_weight WL:511-554)
 *   pack_int4 / pack_uint4                  src/nncf/torch/quantization/quantize_functions.py:417-461 (when packed_out != NULL)
 *
 *   scale_in  nullable: precomputed scale per group; when given no statistics are computed.
 *   zp_in     nullable: precomputed zero point per group (zp_dtype says int32 or float32, WL:679-680).
 *             For asymmetric modes scale_in and zp_in must be given together (WL:472-477).
 *   codes_out nullable: int8 (sym) / uint8 (asym), one code per element, same layout as weight.
 *   packed_out nullable: 4-bit modes only, inner == 1 only; byte k = (u[2k] & 15) | (u[2k+1] << 4) over the
 *             flattened codes with u = code + 8 for INT4_SYM; n_elements/2 bytes.
 *   scale_out nullable: fp32 [n_groups] (written also when scale_in is given: a verbatim copy).
 *   zp_out    nullable: int32 [n_groups] (asymmetric modes computing their own parameters only).
 *   dq_out    nullable: fp32 dequantized weight, same layout as weight.
 */
int nncf_amd_int_quantize(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                          int64_t group_size, int mode, const float* scale_in, const void* zp_in, int zp_dtype,
                          void* codes_out, uint8_t* packed_out, float* scale_out, int32_t* zp_out, float* dq_out,
                          void* stream);

/*
 * Batched form of nncf_amd_int_quantize for data-free compression of many [rows, cols] weights (inner == 1,
 * no precomputed parameters): one host call, device work queued back-to-back on `stream`.
 * Replaces the sequential per-weight loop of PTWeightCompressionAlgoBackend.transform_model
 * (src/nncf/quantization/algorithms/weight_compression/torch_backend.py:442-477) for the lowering part.
 * All arrays are HOST arrays of length n; pointers inside them are device pointers (entries may be NULL
 * where the single-tensor call allows NULL).
 */
int nncf_amd_int_quantize_batched(int n, const void* const* weights_host, int w_dtype, const int64_t* rows_host,
                                  const int64_t* cols_host, int64_t group_size, int mode,
                                  void* const* codes_out_host, uint8_t* const* packed_out_host,
                                  float* const* scale_out_host, int32_t* const* zp_out_host, void* stream);

/*
 * Integer dequantization from unpacked codes: (int32(q) - zp) * scale.  WL:428-448.
 *   codes: int8 (codes_signed != 0) or uint8; zp nullable (zp_dtype as above).
 */
int nncf_amd_int_dequantize(const void* codes, int codes_signed, int64_t outer, int64_t r, int64_t inner,
                            int64_t group_size, const float* scale, const void* zp, int zp_dtype, float* dq_out,
                            void* stream);

/*
 * Quantization error of integer Q->DQ:  max over channels of mean((dq-w)^2) or Frobenius norm of (dq-w).
 * Replaces get_integer_quantization_error WL:329-374 (mean is taken over the whole reduction axis r, the
 * max over the remaining [outer, inner] channels).  result: one fp32 on the device.
 * workspace: at least nncf_amd_int_quant_error_workspace(outer, inner) bytes.
 */
size_t nncf_amd_int_quant_error_workspace(int64_t outer, int64_t inner);
int nncf_amd_int_quant_error(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                             int64_t group_size, int mode, int reduction, float* result, void* workspace,
                             size_t workspace_bytes, void* stream);

/*
 * The metric of MANY 2-D [rows, cols] weights (reduced along the columns) in two launches per 48 tensors: the scoring
 * loop of the mixed-precision criteria (src/nncf/quantization/algorithms/weight_compression/mixed_precision.py:161-199,
 * :350-361 call get_integer_quantization_error weight by weight).  weights_host / rows_host / cols_host: HOST arrays
 * of n entries; results: fp32 [n] on the device; workspace: nncf_amd_int_quant_error_batched_workspace bytes.
 */
size_t nncf_amd_int_quant_error_batched_workspace(int n, const int64_t* rows_host);
int nncf_amd_int_quant_error_batched(int n, const void* const* weights_host, int w_dtype, const int64_t* rows_host,
                                     const int64_t* cols_host, int64_t group_size, int mode, int reduction,
                                     float* results, void* workspace, size_t workspace_bytes, void* stream);

/*
 * The same metric WITHOUT any caller workspace and without any allocation.  No counterpart in the reference: it is
 * for callers that cannot allocate device memory around the call (HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  The per-channel partial sums live in a 1 MiB device array inside the library (one copy per
 * GPU, held until the process ends), so outer * inner is limited to NNCF_AMD_STATIC_ERR_MAX_CHANNELS; beyond that the
 * call returns NNCF_AMD_ERR_WORKSPACE and nncf_amd_int_quant_error with a caller workspace is the way.  Same
 * kernels, same result bits as nncf_amd_int_quant_error.  All calls on one GPU share that array: calls that overlap
 * on one GPU (different streams) race; calls on one stream, or on different GPUs, do not.  The array's address is
 * looked up for the CURRENT device (hipGetDevice) at the call: the current device must be the one `stream` and the
 * buffers belong to.
 */
#define NNCF_AMD_STATIC_ERR_MAX_CHANNELS 262144
int nncf_amd_int_quant_error_static(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                                    int64_t group_size, int mode, int reduction, float* result, void* stream);

/*
 * Nibble packing, standalone.  src/nncf/torch/quantization/quantize_functions.py:417-473.
 *   pack_u4: in uint8 [n] (n even) -> out uint8 [n/2];   pack_i4: in int8 (+8 bias first).
 *   unpack_u4 / unpack_i4: the inverses (out has 2*n_packed entries).
 */
int nncf_amd_pack_u4(const uint8_t* in, int64_t n, uint8_t* out, void* stream);
int nncf_amd_pack_i4(const int8_t* in, int64_t n, uint8_t* out, void* stream);
int nncf_amd_unpack_u4(const uint8_t* in, int64_t n_packed, uint8_t* out, void* stream);
int nncf_amd_unpack_i4(const uint8_t* in, int64_t n_packed, int8_t* out, void* stream);

/*
 * Fused float-codebook lowering (NF4 / MXFP4 / FP4): group abs-max -> scale (E8M0 power of two for MXFP4)
 * -> normalize -> nearest quantile.  Replaces
 *   calculate_float_quantization_params   WL:84-128
 *   _calculate_normalized_weight          WL:613-626
 *   _calculate_float_quantized_weight     WL:557-586   (NF4: searchsorted on CENTER_OF_NF4_QUANTILES;
 *                                                       e2m1: nearest with ties to the even quantile index)
 *   do_float_quantization / do_float_dequantization / float_quantize_dequantize_weight  WL:131-289
 *   values_out nullable: fp32 quantile VALUES (what the reference's torch/numpy path returns, WL:585-586).
 *   codes_out  nullable: uint8 one 4-bit code per element.  NF4: index into NF4_QUANTILES;
 *              e2m1: OCP FP4 bit pattern (sign<<3 | exp<<1 | mantissa), the reference's single zero -> 0b0000.
 *   packed_out nullable: inner == 1 only; two codes per byte, low nibble = even element.
 *   scale_out  nullable: fp32 per group.   scale_e8m0_out nullable (MXFP4 only): uint8 log2(scale)+127.
 *   dq_out     nullable: fp32 value * scale.
 */
int nncf_amd_float_quantize(const void* weight, int w_dtype, int64_t outer, int64_t r, int64_t inner,
                            int64_t group_size, int mode, const float* scale_in, float* values_out,
                            uint8_t* codes_out, uint8_t* packed_out, float* scale_out, uint8_t* scale_e8m0_out,
                            float* dq_out, void* stream);

/* Batched data-free form (packed 4-bit codes + scales [+ E8M0 bytes]); arrays are HOST arrays of device pointers,
 * as in nncf_amd_int_quantize_batched.  scale_e8m0_out_host may be NULL. */
int nncf_amd_float_quantize_batched(int n, const void* const* weights_host, int w_dtype, const int64_t* rows_host,
                                    const int64_t* cols_host, int64_t group_size, int mode,
                                    uint8_t* const* packed_out_host, float* const* scale_out_host,
                                    uint8_t* const* scale_e8m0_out_host, void* stream);

/*
 * Activation-statistics reducers over x = [batch, tokens, hidden] (contiguous, dtype x_dtype).
 * Replace TensorReducerBase._reduce_out_of_place of
 *   MeanReducer(axes=(0,1))          src/nncf/common/tensor_statistics/collectors.py:512-516
 *   MeanVarianceReducer(axes=(1,))   :519-524      MaxVarianceReducer(axes=(1,))  :527-532
 *   MeanAbsMaxReducer(axes=(1,))     :535-540      HAWQAggregator's sum(x*x)      :929-933
 * mean_out: fp32 [hidden] (mean over batch and tokens; the caller rounds to x's dtype like torch.mean does).
 * stats_out: fp32 [4] = { mean(var(x,1)), max(var(x,1)), mean(max|x| over axis 1), sum(x*x) }, each
 *            nullable-by-flag via `want_mean` / `want_stats`.
 * workspace: nncf_amd_reduce_workspace(batch, tokens, hidden) bytes.
 */
size_t nncf_amd_reduce_workspace(int64_t batch, int64_t tokens, int64_t hidden);
int nncf_amd_reduce_activation_stats(const void* x, int x_dtype, int64_t batch, int64_t tokens, int64_t hidden,
                                     float* mean_out, float* stats_out, void* workspace, size_t workspace_bytes,
                                     void* stream);

/*
 * GPTQ Hessian accumulation  H = beta*H + alpha * X^T X  (fp32, MFMA f32 tiles; full symmetric H is written).
 * Replaces the loop body of GPTQ._calculate_hessian
 * (src/nncf/quantization/algorithms/weight_compression/gptq.py:237-240) with beta = n/(n+b), alpha = 2/(n+b).
 *   x: [tokens, k] row-major (x_dtype), h: [k, k] fp32.
 */
int nncf_amd_hessian_accumulate(float* h, const void* x, int x_dtype, int64_t tokens, int64_t k, float alpha,
                                float beta, void* stream);

/*
 * The same accumulation for a STREAM of calibration chunks (the sample loop of GPTQ._calculate_hessian,
 * gptq.py:225-240): `_upper` updates only the upper triangle of H (no symmetrising pass per chunk; with beta = 1 no
 * scaling pass either), `nncf_amd_hessian_finalize` multiplies H by `scale` and writes the lower triangle from the
 * upper one.  Raw sums with alpha = beta = 1 and one finalize(scale = 2 / n) give the reference's running average
 * (it telescopes to (2 / n) * sum X^T X).
 */
int nncf_amd_hessian_accumulate_upper(float* h, const void* x, int x_dtype, int64_t tokens, int64_t k, float alpha,
                                      float beta, void* stream);
int nncf_amd_hessian_finalize(float* h, int64_t k, float scale, void* stream);

/*
 * The `_upper` accumulation (beta = 1) over a LIST of separately allocated calibration items x_i [tokens_i, k], as
 * GPTQ._calculate_hessian receives them (gptq.py:225-240 loops over `inputs`): one launch per 128 items instead of a
 * concatenation copy + launch (16-bit items, 2048 <= k, k % 8 == 0, 16-byte aligned: the 256x256-tile MFMA kernel walks
 * the list, items are zero-padded to 32-token stages by the buffer bounds check; other shapes run item by item).
 * xs_host / tokens_host are HOST arrays of n entries.  Finish with nncf_amd_hessian_finalize.
 */
int nncf_amd_hessian_accumulate_list(float* h, int n, const void* const* xs_host, const int64_t* tokens_host,
                                     int x_dtype, int64_t k, float alpha, void* stream);

/*
 * One whole search step of scale estimation for a weight (scale_estimation.py:268-360), fused: candidate scale ->
 * quantize -> ideal-scale estimate (`near`) -> quantize-dequantize with `near` -> output error of every group against
 * fp_outs (the [group_size x n_samples] activation slice sits in LDS, the contraction runs on the fp32 MFMA) -> the
 * reference's masked blends of best_diffs / result_scale.
 *   weight: [batch, rows, groups, group_size] in its own dtype (NNCF_AMD_F32 or NNCF_AMD_BF16), the grouped view of
 *   the [rows, groups*group_size] weight, 16-byte aligned;
 *   zero_mask, rden: from nncf_amd_se_zero_mask (the importance of :236-247 as 1 bit per element + 1 / (sum + eps)
 *   per group);  s_per_channel: fp32 [batch, groups*group_size] (process_stats' s);
 *   x: fp32 [batch, groups, group_size, n_samples] (process_stats' X, grouped, :249);
 *   fp_outs: fp32 [batch, groups, rows, n_samples] = matmul(moveaxis(weight, -2, -3), x) (:250): WRITTEN by kind 0,
 *   read by kinds 1 and 2;
 *   scale0 (the min-max scale), zero_point (fp32, NULL for symmetric / float modes), best_diffs, result_scale:
 *   fp32 [batch, rows, groups].
 *   kind 0: near = scale0, best_diffs = diff, result_scale = scale0 (the min-max baseline, :252-262), fp_outs written;
 *   kind 1: candidate = result_scale (rectification step, :268-298); kind 2: candidate = factor * scale0 (grid step,
 *   :300-360); both update best_diffs / result_scale in place.  weight_penalty <= 0 disables the penalty term.
 * Every group size >= 1 (whole-row groups included) and every n_samples >= 1 is served: group_size in {32, 64, 128}
 * with n_samples <= 64 (nncf_amd_se_step_supported == 1: the reference's defaults) by the tiled MFMA kernel, which
 * reads zero_mask / rden; everything else by a kernel that rebuilds the importance itself (zero_mask / rden may be
 * NULL then, nncf_amd_se_zero_mask returns without a launch for such shapes).
 */
int nncf_amd_se_step_supported(int64_t group_size, int64_t n_samples);
int nncf_amd_se_zero_mask(const void* weight, int w_dtype, const float* s_per_channel, const float* scale0,
                          const float* zero_point, int64_t batch, int64_t rows, int64_t groups, int64_t group_size,
                          int mode, uint32_t* zero_mask, float* rden, void* stream);
int nncf_amd_se_step(const void* weight, int w_dtype, const uint32_t* zero_mask, const float* rden,
                     const float* s_per_channel, const float* x, float* fp_outs, const float* scale0,
                     const float* zero_point, float* best_diffs, float* result_scale, int64_t batch, int64_t rows,
                     int64_t groups, int64_t group_size, int64_t n_samples, int mode, int kind, float factor,
                     float zero_scale, float weight_penalty, void* stream);

/*
 * The whole search of scale_estimation.py:236-360 for one weight in one call: nncf_amd_se_zero_mask, then
 * nncf_amd_se_step kind 0, `initial_steps` x kind 1, `scale_steps` x kind 2 with factor 1 - 0.05 k (:301), queued back
 * to back on `stream`.  Same buffers as those two (zero_mask / rden are scratch the call fills itself: on the tiled
 * shapes of the integer modes and e2m1 the kind-0 step writes them while it quantizes with the min-max parameters, so
 * the separate pass over the weight is not launched).
 */
int nncf_amd_se_search(const void* weight, int w_dtype, uint32_t* zero_mask, float* rden, const float* s_per_channel,
                       const float* x, float* fp_outs, const float* scale0, const float* zero_point, float* best_diffs,
                       float* result_scale, int64_t batch, int64_t rows, int64_t groups, int64_t group_size,
                       int64_t n_samples, int mode, int initial_steps, int scale_steps, float zero_scale,
                       float weight_penalty, void* stream);

/*
 * AWQ alpha-grid search (AWQ._data_aware_step, src/nncf/quantization/algorithms/weight_compression/awq.py:297-352):
 * for every outlier group g and every candidate scale vector s = 0..steps-1
 *     wq = gweight_g * scale[g, s];  dq = per-row quantize-dequantize(wq)  (one quantization group per row, :337-341)
 *     partial[g, s, tile] = sum over the rows of a 16-row tile and the samples of |dq @ sacts[g, s] - fp_out[g]|
 * in ONE launch; the host sums the tiles (fixed order) and divides by rows * n_samples to get cur_diff (:345).
 *   weight: fp32, rows of pitch row_pitch; group_offsets (DEVICE int64 [groups]): element offset of the group's first
 *   column in row 0 (+ the expert offset for 3-D weights);  scales: fp32 [groups, steps, group_size] (cur_scale, after
 *   the clamp of :315-335);  sacts = gacts[k][j] / scale[k] (:342) as nncf_amd_awq_prepare writes it, 16-byte aligned,
 *   nncf_amd_awq_sacts_bytes(group_size, n_samples, mode) bytes per (group, step) (-1: bad argument): for group_size in
 *   {32, 64, 128} and n_samples <= 64 the tiled MFMA kernels' layouts over NT = 1 / 2 / 4 sixteen-sample tiles (zero
 *   beyond the samples) -- integer modes and e2m1: the exact three-way bf16 split [3][NT][group_size / 32][4][16] x 8
 *   bf16 in the B-operand order of v_mfma_f32_16x16x32_bf16; NF4: fp32 [group_size, 16, NT] holding sample t*16 + c at
 *   [.., c, t] -- and for every other group size >= 1 / sample count plain fp32 [group_size, n_samples];
 *   fp_out: fp32 [groups, rows, n_samples] (:307);  partial: fp32 [groups, steps, ceil(rows / 16)].
 */
int64_t nncf_amd_awq_sacts_bytes(int64_t group_size, int64_t n_samples, int mode);
/*
 * Preparation of nncf_amd_awq_search for a set of outlier groups, one launch (awq.py:311-342): per group g and alpha
 * step s the candidate scale vector cur = gscale[g] ^ alphas[s] (sqrt for 0.5), the fp16-overflow clamp of :315-335 when
 * the previous matmul is merged (gprev = mean |previous weight| of the group's channels, prev_s = max activation
 * magnitude in front of it, n_prev = its reduction length; gprev NULL: no clamp), and gacts / cur in the layout
 * that search reads for the shape and `mode` (above).  gscale: fp32 [groups, group_size], clipped to [quantile 0.1, 100] (:300-301);
 * gacts: fp32 [groups, group_size, n_samples];  scales: fp32 [groups, steps, group_size];  sacts: see above.
 */
int nncf_amd_awq_prepare(const float* gscale, const float* alphas, const float* gacts, const float* gprev, float prev_s,
                         float n_prev, int groups, int steps, int64_t group_size, int64_t n_samples, int mode,
                         float* scales, float* sacts, void* stream);
int nncf_amd_awq_search(const float* weight, int64_t rows, int64_t row_pitch, const int64_t* group_offsets, int groups,
                        int64_t group_size, const float* scales, int steps, const float* sacts, const float* fp_out,
                        int64_t n_samples, int mode, float* partial, void* stream);

/*
 * Token mean of a LIST of separately allocated activation samples x_i[tokens_i, hidden] (one calibration item each):
 * the per-item MeanReducer of the statistics the data-aware algorithms consume (WCTensorStatistic.mean_values,
 * src/nncf/quantization/algorithms/weight_compression/gptq.py:381-396; MeanReducer collectors.py:512-518), two
 * launches per 64 samples instead of three per sample.  means_out: fp32 [n, hidden] (cast to the activation dtype
 * by the caller, as the reducer returns it).  xs_host / tokens_host are HOST arrays of n entries.
 */
size_t nncf_amd_token_mean_workspace(int n, int64_t max_tokens, int64_t hidden);
int nncf_amd_token_mean_batched(int n, const void* const* xs_host, int x_dtype, const int64_t* tokens_host,
                                int64_t hidden, float* means_out, void* workspace, size_t workspace_bytes,
                                void* stream);

/*
 * GPTQ in-block column sweep (gptq.py:298-361, one block of `count` <= 128 columns starting at column i1).
 * For every row independently: at group starts (re)compute scale[, zp] from `weight` (the not-in-block-updated
 * tensor, gptq.py:315-333), quantize-dequantize the column, propagate err = (w-q)/Hinv[c,c] to the remaining
 * columns of the block.  Writes q into quantized[:, i1:i1+count], err into err_block [rows, count],
 * appends scales/zps at group index (i1+i)/group_size.
 *   weight, quantized: [rows, k] fp32;  hinv: [k, k] fp32 upper Cholesky factor of H^-1;
 *   scales: fp32 [rows, k/group_size]; zps: int32 same shape (asymmetric integer modes) or NULL.
 *   use_given_params != 0: the entries of scales/zps for the groups starting in this block were filled in by
 *   the caller (scale estimation inside GPTQ, gptq.py:319-327) and are used instead of being computed.
 * The trailing update W[:, i2:] -= Err @ Hinv[i1:i2, i2:] (gptq.py:363) is a plain GEMM left to the caller.
 */
int nncf_amd_gptq_block(const float* weight, float* quantized, float* err_block, const float* hinv, int64_t rows,
                        int64_t k, int64_t i1, int64_t count, int64_t group_size, int mode, float* scales,
                        int32_t* zps, int use_given_params, void* stream);

/*
 * The dense products of GPTQ's algebra on the bf16 matrix cores (fp32 GEMMs in the reference: the trailing update
 * W[:, i2:] -= Err @ Hinv[i1:i2, i2:] of gptq.py:363 and the products behind linalg.cholesky / cholesky_inverse of
 * :283-285).  nncf_amd_split3_bf16 writes an fp32 matrix view as its exact three-way bf16 split x = h + m + l, six-fold
 * along the contraction index, which becomes the ROW index of the result: order 0 (left operand) h h h m m l, order 1
 * (right operand) h m l h m h.
 *   contract_cols == 0: src [k, n] -> dst [6k, n] (row 6t + s = slot s of src row t);
 *   contract_cols != 0: src [m, k] -> dst [6k, m] (row 6t + s = slot s of src column t);  negate != 0 splits -x.
 *   src: fp32, row pitch src_pitch;  dst: bf16, row pitch dst_pitch (elements).
 * nncf_amd_gemm_bf16_tn_acc: C[m, n] += P^T Q for bf16 P [t, m] (pitch ldp), Q [t, n] (pitch ldq), fp32 accumulation on
 * the MFMA, one fp32 atomic add per element of C and split of t.  With the two splits of fp32 A [m, k] and B as P and Q
 * (t = 6k) this is C += A B to fp32-GEMM accuracy (hh + hm + hl + mh + mm + lh: what is dropped is below 2^-24 |a||b|).
 *   P, Q: 16-byte aligned, pitches multiples of 8, t * pitch * 2 < 2^32 each (else NNCF_AMD_ERR_UNSUPPORTED: the caller
 *   keeps its fp32 GEMM);  C: fp32, row pitch ldc.
 *   tri: what the kernel may skip for a triangular operand or result (contraction index = row / 6 of P and Q), at the
 *   granularity of its 256-wide tiles -- 0 nothing;  1: C = A T^T with T lower triangular (only contraction indices <=
 *   the column of C contribute);  2: C = A T with T lower triangular (indices >= the column of C);  3: C = T B with T
 *   lower triangular (indices <= the row of C);  4: only the lower triangle of C is wanted (tiles strictly above the
 *   diagonal are not touched).  The skipped terms must be zeros (1-3) or unread by the caller (4).
 */
int nncf_amd_split3_bf16(const float* src, int64_t rows, int64_t cols, int64_t src_pitch, int contract_cols, int order,
                         int negate, void* dst, int64_t dst_pitch, void* stream);
int nncf_amd_gemm_bf16_tn_acc(float* c, int64_t ldc, const void* p, int64_t ldp, const void* q, int64_t ldq, int64_t m,
                              int64_t n, int64_t t, int tri, void* stream);

/*
 * Inverse Cholesky factor of one small SPD block: linv = L^-1 (lower triangle written, the rest left untouched)
 * with a = L L^T; only the lower triangle of `a` is read.  Base case of the blocked factorization behind GPTQ's
 * `hessian_inv` (gptq.py:283-285: linalg.cholesky -> cholesky_inverse -> cholesky(upper=True), i.e. the
 * torch_linalg.py:29-36 calls), one workgroup, n <= 128.
 *   a: fp32, row pitch lda (elements);  linv: fp32, row pitch ldo;  info: int32 on the device; when the block
 *   is not positive definite and *info is still 0 it becomes 1 + the index of the first non-positive pivot
 *   (LAPACK's potrf convention; the caller zeroes it beforehand and reads it once per factorization, so the first
 *   failing block of a blocked factorization is the one reported).
 */
int nncf_amd_cholinv_block(const float* a, int64_t lda, float* linv, int64_t ldo, int64_t n, int32_t* info,
                           void* stream);

/*
 * Decompressor forward: packed codes -> weights in the layer's dtype in one pass.
 * Replaces INT8/INT4{Asymmetric,Symmetric}WeightsDecompressor.forward (src/nncf/torch/quantization/layers.py:1303-1460)
 * = unpack_uint4/unpack_int4 + decompress_asymmetric/decompress_symmetric
 * (src/nncf/torch/quantization/quantize_functions.py:389-473) + reshape + cast, with the reference's arithmetic:
 * d = code - zero_point, v = round_to_scale_dtype(d * scale), out = cast(v).  NF4 / MXFP4 (no torch decompressor in
 * the reference) use the code book value for d.
 *   codes: INT8: 1 byte per element; 4-bit modes: 2 elements per byte, low nibble first (INT4_SYM stored +8).
 *   scale: [n_groups], scale_kind 0 = fp32, 1 = fp16 (what the reference's modules hold), 3 = E8M0 bytes (MXFP4).
 *   zero_point: INT8_ASYM uint8 [n_groups]; INT4_ASYM two per byte (pack_uint4 of the zero points); else NULL.
 *   out: [n_groups * group_size] of out_dtype (nncf_amd_dtype).
 */
int nncf_amd_decompress(const uint8_t* codes, int mode, int64_t n_groups, int64_t group_size, const void* scale,
                        int scale_kind, const uint8_t* zero_point, void* out, int out_dtype, void* stream);

/*
 * Export of per-channel INT8 weights to the GPTQ v2 tensor layout (optional export).
 * Replaces the tensor work of convert_linear_module (src/nncf/experimental/torch/gptqmodel/convertor.py:35-84):
 * decompress + gptqmodel's TritonV2QuantLinear.pack on the CPU (bits = 8, group_size = -1, desc_act = False).
 *   codes   : [out_features, in_features] uint8 (codes_signed = 0) or int8 (1: stored as code + 128, zero point 128)
 *   qweight : [in_features / 4, out_features] int32, byte j of word (i, o) = code of weight[o, 4 i + j]
 * qzeros [1, out_features / 4] int32 is the byte image of the uint8 zero points, scales [1, out_features] fp16 the
 * scale vector: both are views on the host side.  in_features and out_features must be multiples of 4.
 */
int nncf_amd_gptq_pack8(const void* codes, int codes_signed, int64_t out_features, int64_t in_features,
                        int32_t* qweight, void* stream);
/* The forward of that layout: weight[o, k] = RN_fp16((byte k of column o - zero[o]) * scale[o]) cast to out_dtype,
 * [out_features, in_features]; replaces the dequantization inside TritonV2QuantLinear.forward. */
int nncf_amd_gptq_dequant8(const int32_t* qweight, const int32_t* qzeros, const void* scales, int64_t in_features,
                           int64_t out_features, void* weight, int out_dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNCF_AMD_WC_H */
