/*
 * idiff_hip.h -- C ABI of libidiff_hip.so, the gfx950 (MI355X) kernels behind the
 * manifold_dimension hot path of GBATZOLIS/ID-diff.
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer (HBM) unless named h_*; the library never
 *     allocates, frees or synchronises: the caller owns all buffers and passes the
 *     hipStream_t (as void*) the work is enqueued on.  Re-entrant.  The only process-level state is
 *     (a) which devices already carry the per-kernel dynamic-LDS attribute (one bit per device
 *     ordinal, set on first use on that device) and (b) the debug switches of idiff_set_option(),
 *     read from the environment once when the library is loaded -- never on a launch path.
 *   - Return value: 0 on success, otherwise a hipError_t (launch failure) or
 *     IDIFF_EINVAL (1001) for an argument the kernels cannot take; idiff_last_error()
 *     gives a thread-local message.
 *   - "Replaces" names the reference interface (file:line under GBATZOLIS/ID-diff)
 *     that the entry point stands in for.
 */
#ifndef IDIFF_HIP_H
#define IDIFF_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define IDIFF_EINVAL 1001
#define IDIFF_ABI_VERSION 1

int idiff_abi_version(void);
const char *idiff_last_error(void);
/* First 16 hex digits of the sha256 over the sources the library was built from (csrc/build.sh); the Python binding
 * compares it with the tree and refuses a stale libidiff_hip.so. */
const char *idiff_source_stamp(void);
/* "" for the product library.  A diagnostic / A-B build (csrc/build.sh with IDIFF_VARIANT=<name> IDIFF_VARIANT_FLAGS=...: timing-only
 * kernels, stamped kernels, lifted scratch limit) reports its extra compile flags here, is written to libidiff_hip.<name>.so only,
 * and the Python binding loads it only when IDIFF_LIB_VARIANT=<name> asks for it -- never from the production path. */
const char *idiff_variant_flags(void);

/* Debug / A-B switches, named like the environment variables that initialise them at load time:
 * IDIFF_NO_WINOGRAD (3x3 convs on the implicit GEMM), IDIFF_NO_COLSTATS, IDIFF_NO_PIPE, IDIFF_SCALAR_EPILOGUE,
 * IDIFF_DBUF_ONLY, IDIFF_TRIDIAG_ONESTAGE (per-column Householder instead of the two-stage band reduction),
 * IDIFF_UFD_ROWS, IDIFF_CHASE_WAVEFRONT (bulge chasing one launch per wavefront instead of the persistent systolic kernel),
 * IDIFF_CHASE_SPIN_LIMIT (polls before a waiting node of the systolic chase gives up; default 2^24), IDIFF_FAKE_CU_COUNT
 * (CU count used when sizing the systolic chase; tests), IDIFF_SBR_SYNC (band reduction waits for and names every launch on
 * stderr: a fault then names its kernel), IDIFF_SBR_FULL (band reduction keeps both triangles up to date, the round-2 form),
 * IDIFF_NO_SPLIT (contractions of idiff_gemm_f32 / idiff_conv2d_nhwc_f32 on the fp32 matrix cores instead of the
 * split-precision products described there), IDIFF_WINO_SPLIT (opt-in: the split-precision Winograd kernel),
 * IDIFF_SBR_LOOKAHEAD (opt-in: band reduction with the look-ahead -- the bulk of a panel's trailing update on a helper
 * stream beside the next panel's factorisation; 5 % at D = 12288 when the helper gets a hardware queue of its own, 50 %
 * SLOWER when the runtime maps it onto the caller's queue, which happens once a process has made a few streams),
 * IDIFF_NO_WINO43 (3x3 convolutions on the F(2x2,3x3) kernel instead of F(4x4,3x3)), IDIFF_NO_WINO43H (F(4x4,3x3) with its
 * contractions on the fp32 matrix cores instead of fp16 pairs), IDIFF_NO_FUSED_ATTN (idiff_attention256_ok answers 0), IDIFF_NO_WINO1D (idiff_conv2d_wino1d_ok answers 0), IDIFF_NO_PAIRS (idiff_gemm_pairs_ok answers 0: the 1x1
 * projections behind a GroupNorm stay on idiff_gemm_f32's six-product form), IDIFF_PAIRS_MIN_TILES (tests: the number of 128 x 128
 * tiles from which idiff_gemm_pairs_ok answers 1; default 256).
 * Returns the previous value, -1 for an unknown name.  No reference counterpart. */
int idiff_set_option(const char *name, int value);

/* The same switch for launches made from the CALLING host thread only (every launcher reads its switches on the thread that
 * calls it); set = 0 removes the override and the process-wide value applies again.  Used by the fail-soft re-solve of a
 * failed eigensolve, so that selecting a slower solver form for ONE launch cannot change what another host thread launches
 * in the meantime.  Returns 0, -1 for an unknown name.  No reference counterpart. */
int idiff_set_thread_option(const char *name, int value, int set);

/* ------------------------------------------------------------------ native ops (op/) */

/* Replaces the pybind entry `upfirdn2d(input, kernel, up_x, up_y, down_x, down_y, pad_x0, pad_x1,
 * pad_y0, pad_y1)` of op/upfirdn2d.cpp:12-19 (host op op/upfirdn2d_kernel.cu:209-369).
 * x is [major, in_h, in_w, minor] fp32 contiguous, k is [kh, kw] fp32 (NOT flipped: the op is a true
 * convolution, the kernel flips), out is [major, out_h, out_w, minor] with
 * out_h = (in_h*up_y + pad_y0 + pad_y1 - kh)/down_y + 1 (op/upfirdn2d_kernel.cu:237-240).
 * minor = 1 is the NCHW view the reference uses (op/upfirdn2d.py:99); minor = C serves NHWC activations. */
int idiff_upfirdn2d_f32(const float *x, const float *k, float *out, int major, int in_h, int in_w, int minor,
                        int kh, int kw, int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1,
                        int pad_y0, int pad_y1, void *stream);

/* Replaces `fused_bias_act(input, bias, refer, act, grad, alpha, scale)` of op/fused_bias_act.cpp:11-17
 * (kernel op/fused_bias_act_kernel.cu:18-49).  out[i] = f(x[i] + b[(i/step_b) % size_b]) * scale with
 * act*10+grad in {10,11: identity; 12,32: 0; 30: x>0?x:x*alpha; 31: ref>0?x:x*alpha}.  b may be NULL
 * (size_b = 0), ref may be NULL unless grad = 1. */
int idiff_fused_bias_act_f32(const float *x, const float *b, const float *ref, float *out, int64_t n,
                             int step_b, int size_b, int act, int grad, float alpha, float scale, void *stream);

/* The other two dtypes of the reference's native-op dispatch (AT_DISPATCH_FLOATING_TYPES_AND_HALF: op/upfirdn2d_kernel.cu:311,
 * op/fused_bias_act_kernel.cu:79).  Same arguments and geometry as the _f32 entry points; the FIR kernel / bias / ref tensors carry
 * the input's dtype, as the reference's `data_ptr<scalar_t>()` calls require.  f16 pointers are IEEE binary16 (torch.float16).
 *   upfirdn2d:      f16 = fp32 products and accumulation, rounded once to half (the CPU path's arithmetic, op/upfirdn2d.py:159-200;
 *                   the reference's CUDA kernels round per tap in one of two ways depending on the kernel chosen); f64 = fp64
 *                   products and accumulation (its tiled kernels form fp32 products under an fp64 accumulator, .cu:115-116,198).
 *   fused_bias_act: the reference kernel's scalar_t arithmetic operation by operation: alpha and scale are converted to the
 *                   tensor's dtype first (.cu:19), then x = r(x + b), y = x > 0 ? x : r(x * alpha), out = r(y * scale) with r =
 *                   rounding to the dtype. */
int idiff_upfirdn2d_f16(const void *x, const void *k, void *out, int major, int in_h, int in_w, int minor, int kh, int kw,
                        int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void *stream);
int idiff_upfirdn2d_f64(const double *x, const double *k, double *out, int major, int in_h, int in_w, int minor, int kh, int kw,
                        int up_x, int up_y, int down_x, int down_y, int pad_x0, int pad_x1, int pad_y0, int pad_y1, void *stream);
int idiff_fused_bias_act_f16(const void *x, const void *b, const void *ref, void *out, int64_t n, int step_b, int size_b,
                             int act, int grad, float alpha, float scale, void *stream);
int idiff_fused_bias_act_f64(const double *x, const double *b, const double *ref, double *out, int64_t n, int step_b,
                             int size_b, int act, int grad, float alpha, float scale, void *stream);

/* ------------------------------------------------------------------ dense contractions (fp32 in, fp32 out) */

/* Activation codes shared by the epilogues below. */
#define IDIFF_ACT_NONE 0
#define IDIFF_ACT_SILU 1
#define IDIFF_ACT_ELU 2
#define IDIFF_ACT_RELU 3
#define IDIFF_ACT_LRELU 4 /* slope 0.2, models/layers.py:36 */

/* Epilogue applied by idiff_gemm_f32 / idiff_conv2d_nhwc_f32 to every accumulator element (m, n):
 *   v = acc + bias[n] + rowbias[(m / rows_per_group) * ld_rowbias + n]
 *   v = act(v)
 *   v = (v + residual[m * ld_residual + n]) * out_scale          (residual optional)
 *   v = v * rowscale[m / rows_per_group]                          (optional, e.g. -1/std[b]: models/utils.py:266-267)
 *   out[m * ldc + n] = v
 * NULL pointers drop the corresponding term.  This is what lets one launch produce
 * `Conv_0(h) + Dense_0(act(temb))[:, :, None, None]` (models/layerspp.py:256-259) or
 * `(x + Conv_1(h)) / sqrt(2)` (:262-273) without extra passes over HBM. */
typedef struct idiff_epilogue {
  const float *bias;      /* [N] or NULL */
  const float *rowbias;   /* [M / rows_per_group, ld_rowbias] or NULL */
  int64_t ld_rowbias;
  int rows_per_group;     /* e.g. H*W for a per-sample time-embedding bias */
  int act;                /* IDIFF_ACT_* */
  const float *residual;  /* [M, ld_residual] or NULL */
  int64_t ld_residual;
  float out_scale;        /* 1.0f for none */
  const float *rowscale;  /* [M / rows_per_group] or NULL */
  double *colstats;       /* NULL, or [tiles_m, N, 2] fp64: per row-tile column sums (sum, sum of squares) of the stored
                             values, so that the GroupNorm consuming the output needs no statistics pass of its own; only
                             valid when idiff_gemm_colstats_split / idiff_conv2d_colstats_split returns > 0 */
} idiff_epilogue;

/* Batched C[b] = epilogue(A[b] (M x K, row-major, lda) * Bt[b]^T (Bt is N x K, row-major, ldb)).
 * Replaces torch.nn.Linear (models/fcn.py:18-28), NIN / 1x1 conv contractions (models/layers.py:555-564),
 * and the two attention einsums of models/layerspp.py:82-86.  fp32 operands and results.  Arithmetic of the fast path
 * (16-byte aligned K-contiguous operands below 4 GiB): every operand element is cut EXACTLY into three bf16 pieces
 * (8 + 8 + 8 mantissa bits) and a product is formed as the six partial products of weight >= 2^-16 on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation; what is left out is < 2^-23 of the product, the size of one fp32
 * rounding (1.7e-7 against an fp64 contraction where the k-ordered fp32 fma chain of v_mfma_f32_32x32x2_f32 gives
 * 2.0e-7).  IDIFF_NO_SPLIT selects that fp32 chain; operands the fast path does not take always use it.
 * Non-finite operands: the cut forms x - hi(x), so a +-inf element turns into NaN in its lower pieces and every output that
 * element reaches is NaN, where the fp32 chain would give +-inf (or NaN against a zero).  Non-finite stays non-finite -- the
 * drivers refuse a score matrix with any non-finite entry either way -- but inf vs NaN is not preserved on this path.
 * batch strides are in elements; a stride of 0 broadcasts that operand.  The epilogue pointers are
 * shared by all batch entries. */
int idiff_gemm_f32(const float *A, int64_t lda, int64_t strideA, const float *Bt, int64_t ldb, int64_t strideB,
                   float *C, int64_t ldc, int64_t strideC, int M, int N, int K, int batch,
                   const idiff_epilogue *ep, void *stream);

/* The batched contraction of idiff_gemm_f32, C[b] = epilogue(A[b] (M x K) * Bt[b]^T (Bt is N x K)), with every fp32 operand element
 * as a PAIR of fp16 values (hi = fp16(v), lo = fp16(v - hi): 22 significand bits) and a product as lo*hi + hi*lo + hi*hi on
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation -- three matrix instructions where idiff_gemm_f32's exact bf16 cut needs six, and
 * the matrix pipes are what that form runs out of (1.36-1.64x faster on the q / k / v projections of models/layerspp.py:66-98 at
 * BASELINE sizes, scripts/gemm_pairs_probe.py).  fp16 has a 5-bit exponent, so the CALLER vouches for the ranges: one operand is an
 * ACTIVATION, used as it is, and must be of order one -- |x| < 65504 (beyond: +-inf, the outputs NaN), and an element below 0.25
 * carries an absolute error of 2^-25 instead of a relative 2^-22: the output of a GroupNorm is the intended operand; the other is
 * a WEIGHT (Bt, or A when weight_is_a != 0 -- V^T = Wv n^T of the attention block), multiplied by w_scale[0], the power of two that
 * idiff_gemm_pairs_scale_f32 computes once per weight (max |w| -> [2^11, 2^12)), the sums by w_scale[1] = 1 / w_scale[0], exactly.
 * w_scale is a DEVICE pointer to those two floats.  Measured against an fp64 contraction of the same fp32 operands: 1.0e-7 where the
 * six-product form gives 1.3e-7 (a GroupNorm's output x N(0, 1/K) weights; tests/test_hip_ops.py).  idiff_gemm_pairs_ok: 1 when the
 * shape is served (N > 64, K % 4 == 0, at least 256 tiles of 128 x 128 over the batch; 0 under IDIFF_NO_PAIRS / IDIFF_NO_SPLIT /
 * IDIFF_NO_PIPE).  Operands 16-byte aligned, row pitches and batch strides multiples of 4, one batch slice inside 4 GiB.  Same
 * epilogue, same column-statistics layout as idiff_gemm_f32 (idiff_gemm_colstats_split; unbatched only). */
int idiff_gemm_pairs_ok(int M, int N, int K, int batch);
int idiff_gemm_pairs_scale_f32(const float *w, int64_t ldw, int rows, int K, float *w_scale, void *stream);
int idiff_gemm_pairs_f32(const float *A, int64_t lda, int64_t strideA, const float *Bt, int64_t ldb, int64_t strideB,
                         const float *w_scale, int weight_is_a, const float *act_scale, float *C, int64_t ldc, int64_t strideC,
                         int M, int N, int K, int batch, const idiff_epilogue *ep, void *stream);

/* The activation operand of the pair form when it is NOT a GroupNorm's output: act_scale (NULL above = 1) is a device pointer to
 * {s, 1 / s}, s the power of two the activation is multiplied by before its cut, so that its ROOT MEAN SQUARE lands in [0.71, 1.41):
 * the pair then carries 22 bits relative to the tensor's own scale whatever that scale is (elements below rms / 4: absolute error
 * 2^-25 rms), and the range bound becomes max |x| < 65504 rms.  idiff_pairs_act_scale_f32 derives it on the device from the per-tile
 * column sums (sum, sum of squares) that the contraction(s) producing the tensor -- or the two tensors of a concatenation -- left for
 * the GroupNorm that reads them as well (idiff_epilogue.colstats: [B, nsplit, C, 2] fp64); `out`: 8 floats, {s, 1 / s} in front.
 * idiff_gemm_pairs_2src_f32: the two-source contraction of idiff_gemm_2src_f32 ([A1 | A2], the shortcut of a residual block on
 * torch.cat([h, skip]), models/ncsnpp.py:376-385) on pairs, one act_scale for both sources.  K1 % 32 == 0. */
int idiff_pairs_act_scale_f32(const double *ws1, int nsplit1, int C1, const double *ws2, int nsplit2, int C2, int B, int HW, float *out,
                              void *stream);
int idiff_gemm_pairs_2src_f32(const float *A1, const float *A2, int64_t lda, int K1, const float *act_scale, const float *Bt, int64_t ldb,
                              const float *w_scale, float *C, int64_t ldc, int M, int N, int K, const idiff_epilogue *ep, void *stream);

/* The same contraction with A given as two row-major matrices of equal row pitch, A = [A1 (M x K1) | A2 (M x (K - K1))]:
 * the 1x1 shortcut of a residual block whose input is torch.cat([h, skip], dim=1) (models/ncsnpp.py:376-385 with
 * layerspp.py:271-274) without materialising the concatenation or an intermediate partial product.  K1 % 32 == 0. */
int idiff_gemm_2src_f32(const float *A1, const float *A2, int64_t lda, int K1, const float *Bt, int64_t ldb, float *C,
                        int64_t ldc, int M, int N, int K, const idiff_epilogue *ep, void *stream);

/* 2-D convolution, NHWC activations: x [B, H, W, Cin] (Cin % 4 == 0), weights packed as
 * wt [Cout, KH, KW, Cin] (= the reference's [Cout, Cin, KH, KW] nn.Conv2d weight permuted once at load),
 * out [B, OH, OW, Cout] with OH = (H + pad_lo + pad_hi - KH)/stride + 1 (pad_lo on top/left, pad_hi on
 * bottom/right: the non-FIR Downsample pads (0, 1), models/layerspp.py:153-155).  Implicit GEMM with M = B*OH*OW,
 * N = Cout, K = KH*KW*Cin on the same MFMA core as idiff_gemm_f32; zero padding.
 * Replaces F.conv2d behind ddpm_conv3x3 / ddpm_conv1x1 (models/layers.py:100-132) and the stride-2 VALID
 * conv of conv_downsample_2d (models/up_or_down_sampling.py:178).  rows_per_group in the epilogue counts
 * OUTPUT pixels (OH*OW for a per-sample bias). */
int idiff_conv2d_nhwc_f32(const float *x, const float *wt, float *out, int B, int H, int W, int Cin, int Cout,
                          int KH, int KW, int stride, int pad_lo, int pad_hi, const idiff_epilogue *ep, void *stream);

/* Fused GroupNorm statistics: when `rows_per_sample` consecutive output rows form one sample, these return the number of
 * workgroup row-tiles per sample (`nsplit`: epilogue.colstats is then laid out [samples, nsplit, N, 2]) or 0 when the
 * fused statistics are unavailable for the problem (general kernel, operands beyond 4 GiB, tiles straddling samples). */
int idiff_gemm_colstats_split(int M, int N, int K, int64_t lda, int64_t ldb, int rows_per_sample);
int idiff_conv2d_colstats_split(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad_lo,
                                int pad_hi);

/* The same 3x3 / stride 1 / pad 1 convolution (ddpm_conv3x3, models/layers.py:100-116; conv3x3 of
 * models/layerspp.py) by Winograd F(2x2, 3x3): 2.25x fewer multiplications than the implicit GEMM, still fp32
 * throughout (results agree with idiff_conv2d_nhwc_f32 to fp32 rounding, not bit for bit).
 *   idiff_conv2d_winograd_ok       1 when the geometry is served (H, W even, Cin % 8 == 0, Cout % 64 == 0), else 0.
 *   idiff_winograd_weight_floats   size of the transformed filter bank u (16 * Cin * Cout floats).
 *   idiff_winograd_pack_f32        wt [Cout, 3, 3, Cin] (the panel idiff_conv2d_nhwc_f32 takes) -> u, once per layer.
 *   idiff_conv2d_winograd_f32      x [B, H, W, Cin] -> out [B, H, W, Cout], epilogue as above (rows_per_group counts
 *                                  output pixels).
 *   idiff_conv2d_winograd_colstats_split   nsplit of epilogue.colstats ([samples, nsplit, Cout, 2]) or 0 when the
 *                                  fused statistics are unavailable (fewer than 64 output tiles per sample). */
int idiff_conv2d_winograd_ok(int B, int H, int W, int Cin, int Cout);
int64_t idiff_winograd_weight_floats(int Cin, int Cout);
int idiff_winograd_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream);
int idiff_conv2d_winograd_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                              const idiff_epilogue *ep, void *stream);
int idiff_conv2d_winograd_colstats_split(int B, int H, int W, int Cin, int Cout);
/* F(4x4, 3x3): the same convolution with 36 multiplications per 4x4 output tile (2.25 per output; the F(2x2, 3x3) form above
 * spends 4), interpolation points 0, +-2/3, +-3/2, infinity, all arithmetic fp32 on v_mfma_f32_32x32x2_f32 (csrc/winograd43.hip).
 * Per layer 0.8-1.4e-6 against an fp64 convolution where the 2x2 form gives 3-9e-7; measured on the whole nf = 128 NCSN++ before
 * the kernel was written (scripts/f43_emulation.py): rel_err(S) 3.3e-6 against an fp64 network, singular values within 1.4e-5.
 *   idiff_conv2d_winograd43_ok      1 when served: H % 4 == 0, W % 4 == 0, Cin % 8 == 0, Cout % 64 == 0, every tensor within one
 *                                   buffer descriptor (4 GiB), and neither IDIFF_NO_WINOGRAD nor IDIFF_NO_WINO43 set.
 *   idiff_winograd43_weight_floats  size of the transformed filter bank (36 * Cin * Cout floats).
 *   idiff_winograd43_pack_f32       wt [Cout, 3, 3, Cin] -> U = G g G^T (fp64, rounded once), once per layer.
 *   idiff_conv2d_winograd43_f32     x [B, H, W, Cin] -> out [B, H, W, Cout] with the epilogue of idiff_conv2d_nhwc_f32; a per-row-group
 *                                   bias / scale only per image (rows_per_group = H * W).
 *   idiff_conv2d_winograd43_colstats_split   nsplit of epilogue.colstats ([samples, nsplit, Cout, 2]) or 0 when the statistics cannot
 *                                   be produced (a workgroup covers 32 tiles of 4x4 pixels: whole workgroups per sample, or whole
 *                                   samples per workgroup).
 * Replaces F.conv2d behind ddpm_conv3x3 (models/layers.py:119-132) like the 2x2 form. */
int idiff_conv2d_winograd43_ok(int B, int H, int W, int Cin, int Cout);
int idiff_conv2d_winograd43_colstats_split(int B, int H, int W, int Cin, int Cout);
int64_t idiff_winograd43_weight_floats(int Cin, int Cout);
int idiff_winograd43_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream);
int idiff_conv2d_winograd43_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                const idiff_epilogue *ep, void *stream);
/* F(4x4, 3x3) with the 36 contractions on the fp16 matrix cores (v_mfma_f32_32x32x16_f16), each fp32 operand as a PAIR of fp16
 * values hi = fp16(v), lo = fp16(v - hi) and three of the four partial products kept (hi hi + hi lo + lo hi, fp32 accumulation);
 * transforms in fp32 as above.  The filter bank is scaled by a power of two at pack time (undone exactly on the outputs) so that its
 * low parts are normal fp16 numbers; the transformed input is used as is and must stay below 65504 in magnitude (activations below
 * ~2000): beyond that the outputs are NaN.  Same parity bars as the fp32 form; measured (scripts/f43_emulation.py, whole network)
 * per layer 8.0e-7 and rel_err(S) 3.33e-6 where the fp32 contraction gives 7.8e-7 and 3.27e-6.
 *   idiff_conv2d_winograd43h_ok     1 when served: the fp32 form's conditions, Cin % 16 == 0, 32 <= Cin <= 1024, and none of
 *                                   IDIFF_NO_WINOGRAD / IDIFF_NO_WINO43 / IDIFF_NO_WINO43H set.
 *   idiff_winograd43h_weight_floats size of the bank in floats (36 * Cin * Cout for the fp16 pairs + 4 of header).
 *   idiff_winograd43h_pack_f32      wt [Cout, 3, 3, Cin] -> the scaled pairs of U = G g G^T (fp64, rounded once to fp32, then cut).
 *   idiff_conv2d_winograd43h_f32    as idiff_conv2d_winograd43_f32; colstats geometry: idiff_conv2d_winograd43_colstats_split. */
int idiff_conv2d_winograd43h_ok(int B, int H, int W, int Cin, int Cout);
int64_t idiff_winograd43h_weight_floats(int Cin, int Cout);
int idiff_winograd43h_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream);
int idiff_conv2d_winograd43h_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                 const idiff_epilogue *ep, void *stream);

/* The same 3x3 / stride 1 / pad 1 convolution (replaces F.conv2d behind ddpm_conv3x3, models/layers.py:100-132, as the entries above)
 * by Winograd's F(4, 3) along the image rows only -- the three filter rows stay a direct sum: 4.5 multiplications per output instead of
 * 2.25, on fp16 pairs as idiff_conv2d_winograd43h_f32 (same cut, same three partial products, fp32 accumulation; one 6-point transform
 * instead of two: gain 5.4 instead of 29.3, the input may reach |x| ~ 12,000 before a pair overflows to NaN).  Twice the matrix work for
 * half the operand traffic per output: a workgroup of 512 output pixels x 64 channels reads 18 instead of 36 transformed filter slots
 * per K step and every input pixel once instead of 2.25 times, which is what bounds the 2-D form on gfx950 (csrc/wino1d.hip).
 *   idiff_conv2d_wino1d_ok             1 when served: W in {4, 8, 16, 32, 64}, H % 4 == 0 with 512 / W a multiple or a divisor of H (a workgroup takes
 *                                      whole images or a whole part of one), Cin % 16 == 0, 32 <= Cin <= 1024, Cout % 64 == 0, every tensor
 *                                      within one buffer descriptor; 0 under IDIFF_NO_WINOGRAD / IDIFF_NO_WINO43H / IDIFF_NO_WINO1D.
 *   idiff_conv2d_wino1d_colstats_split nsplit of epilogue.colstats ([samples, nsplit, Cout, 2]): H * W / 512 for maps above 512 pixels, else 1
 *                                      (0 under IDIFF_NO_COLSTATS or when the geometry is not served).
 *   idiff_wino1d_weight_floats         size of the bank in floats (18 * Cin * Cout for the fp16 pairs + 4 of header).
 *   idiff_wino1d_pack_f32              wt [Cout, 3, 3, Cin] -> the scaled pairs of U[i][ky] = (G g[ky])[i] (fp64, rounded once to fp32, then cut).
 *   idiff_conv2d_wino1d_f32            x [B, H, W, Cin] -> out [B, H, W, Cout] with the epilogue of idiff_conv2d_nhwc_f32. */
int idiff_conv2d_wino1d_ok(int B, int H, int W, int Cin, int Cout);
int idiff_conv2d_wino1d_colstats_split(int B, int H, int W, int Cin, int Cout);
int64_t idiff_wino1d_weight_floats(int Cin, int Cout);
int idiff_wino1d_pack_f32(const float *wt, float *u, int Cin, int Cout, void *stream);
int idiff_conv2d_wino1d_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                            const idiff_epilogue *ep, void *stream);

/* Split-precision form of the same convolution: the 16 position-wise contractions run on the bf16 matrix cores with every
 * fp32 operand cut exactly into three bf16 pieces and six of the nine partial products kept (fp32 accumulation; what is
 * dropped is < 2^-23 of a product, one fp32 rounding -- the result meets the same parity bars as the fp32 form).
 *   idiff_conv2d_winograd_split_ok       1 when served: the fp32 form's conditions, Cin % 16 == 0, and IDIFF_WINO_SPLIT set
 *                                        (opt-in: measured slower than the fp32 form on the NCSN++ layers).
 *   idiff_winograd_split_weight_floats   size of its filter bank (24 * Cin * Cout floats: three bf16 per transformed weight).
 *   idiff_winograd_pack_split_f32        wt [Cout, 3, 3, Cin] -> that bank, once per layer.
 *   idiff_conv2d_winograd_split_f32      as idiff_conv2d_winograd_f32 with that bank (same epilogue, same colstats split). */
int idiff_conv2d_winograd_split_ok(int B, int H, int W, int Cin, int Cout);
int64_t idiff_winograd_split_weight_floats(int Cin, int Cout);
int idiff_winograd_pack_split_f32(const float *wt, float *u, int Cin, int Cout, void *stream);
int idiff_conv2d_winograd_split_f32(const float *x, const float *u, float *out, int B, int H, int W, int Cin, int Cout,
                                    const idiff_epilogue *ep, void *stream);

/* ------------------------------------------------------------------ normalisation / pointwise (HBM-bound) */

/* GroupNorm statistics over NHWC x [B, HW, C] with G groups: stats[b, g] = {mean, rstd}, biased variance,
 * rstd = 1/sqrt(var + eps) (torch.nn.GroupNorm as used at models/layerspp.py:219,231).  fp64 accumulation.
 * `x2`/`C2` describe an optional second source whose channels are appended to x's (the skip tensor of
 * torch.cat([h, hs.pop()], 1), models/ncsnpp.py:324) so the concatenation is never materialised for the norm
 * (a group may straddle the two sources: 256 + 128 channels in 32 groups of 12).  workspace: >= B * nsplit * (C + C2) * 2 doubles, nsplit as
 * returned by idiff_groupnorm_nsplit. */
int idiff_groupnorm_nsplit(int B, int HW, int C);
int idiff_groupnorm_stats_f32(const float *x, int C, const float *x2, int C2, int B, int HW, int G, float eps,
                              double *workspace, float *stats, void *stream);
/* Same statistics from the per-tile column sums a producing contraction wrote through epilogue.colstats (one workspace
 * per source tensor, [B, nsplit_i, C_i, 2] fp64); ws2 may be NULL.  No pass over the activations. */
int idiff_groupnorm_finalize_f32(const double *ws1, int nsplit1, int C1, const double *ws2, int nsplit2, int C2, int B,
                                 int HW, int G, float eps, float *stats, void *stream);
/* y[b, p, c] = act(n * (1 + mod[b, c]) + mod[b, Ctot + c]) with n = (x - mean) * rstd * gamma[c] + beta[c]; writes the
 * channel-concatenated output [B, HW, Ctot = C+C2].  mod ([B, ld_mod], scale then shift halves) is the scale-shift
 * conditioning of models/BeatGANsblocks.py:258-332 (`h * (1 + scale) + shift` after the norm); NULL skips it. */
int idiff_groupnorm_apply_f32(const float *x, int C, const float *x2, int C2, int B, int HW, int G,
                              const float *stats, const float *gamma, const float *beta, const float *mod,
                              int64_t ld_mod, int act, float *y, void *stream);
/* idiff_groupnorm_finalize_f32 + idiff_groupnorm_apply_f32 in one launch: the statistics are taken from the producers'
 * column sums (same arithmetic, same result) inside the apply kernel.  C + C2 <= 1024, B <= 65535. */
int idiff_groupnorm_apply_colstats_f32(const float *x, int C, const float *x2, int C2, int B, int HW, int G,
                                       const double *ws1, int nsplit1, const double *ws2, int nsplit2, float eps,
                                       const float *gamma, const float *beta, const float *mod, int64_t ld_mod,
                                       int act, float *y, void *stream);

/* Row softmax of x [rows, cols] scaled by `scale` before the exponent (layerspp.py:82-84). In place allowed. */
int idiff_softmax_rows_f32(const float *x, float *y, int64_t rows, int cols, float scale, void *stream);

/* Single-head self-attention over 256 tokens in ONE launch, the [256, 256] logits never leaving the chip (models/layerspp.py:75-91:
 * w = einsum(q, k) * C^-1/2 -> softmax -> einsum(w, v); models/BeatGANsblocks.py:466-491, one head):
 *     out[b, i, :] = sum_j softmax_j(q[b, i] . k[b, j] * scale) v[b, j, :]  (+ bias_v)
 * qk: [B * 256, ld_qk] with q in columns [0, C) and k in [C, 2 C) (the stacked projection the executors already make); vt: V^T as
 * [B, C, 256] (idiff_gemm_*'s weight-times-activation form); out: [B * 256, C].  Both contractions run on fp16 PAIRS (22
 * significand bits, 3 products, fp32 accumulation -- idiff_gemm_pairs_f32's arithmetic), the softmax in fp32.  s_qk / s_v: DEVICE
 * pointers to {s, 1 / s}, the power of two q and k / v are multiplied by before their cut (the caller derives them once per weight:
 * the projections' inputs are GroupNorm outputs); s |q|, s |k|, s |v| must stay below 65504 -- beyond it the outputs are NaN,
 * never finite-and-wrong.  idiff_attention256_ok: 1 for tokens == 256 and C in {128, 256}, 0 otherwise and under
 * IDIFF_NO_FUSED_ATTN / IDIFF_NO_PAIRS / IDIFF_NO_SPLIT (the callers then run the three-launch form on idiff_gemm_f32).
 * All pointers 16-byte aligned, ld_qk a multiple of 4. */
int idiff_attention256_ok(int B, int tokens, int C);
int idiff_attention256_f32(const float *qk, int64_t ld_qk, const float *vt, const float *bias_v, const float *s_qk, const float *s_v,
                           float *out, int B, int tokens, int C, float scale, void *stream);

/* y = act(a * alpha + beta_const) elementwise; covers `2*x - 1` (models/ncsnpp.py:264-266), SiLU/ELU of the
 * time embedding, and -out/std when `rowscale` ([n / inner]) is given: y = act(...) * rowscale[i / inner]. */
int idiff_affine_act_f32(const float *a, float *y, int64_t n, float alpha, float beta_const, int act,
                         const float *rowscale, int64_t inner, void *stream);
/* y = (a + b) * scale (skip-rescale adds, models/ncsnpp.py:303-307). */
int idiff_add_scale_f32(const float *a, const float *b, float *y, int64_t n, float scale, void *stream);
/* Gaussian Fourier features: out[b] = [sin(2*pi*t[b]*W), cos(2*pi*t[b]*W)] (models/layerspp.py:39-41). */
int idiff_fourier_embed_f32(const float *t, const float *W, float *out, int B, int half, void *stream);
/* Sinusoidal timestep embedding, dim even.  mode 0: DDPM (models/layers.py:524-538, [sin, cos], log(max)/(half-1));
 * mode 1: guided-diffusion / BeatGANs (models/BeatGANs_nn.py:107-125, [cos, sin], log(max)/half). */
int idiff_positional_embed_f32(const float *t, float *out, int B, int dim, float max_positions, int mode, void *stream);
/* out[r, :] = cat(a[r, :Ca], b[r, :Cb]) for r < rows (channel concat of NHWC tensors / fcn's cat([x, t])). */
int idiff_concat_cols_f32(const float *a, int Ca, const float *b, int Cb, float *out, int64_t rows, void *stream);
/* Layout changes at the model boundary: NCHW [B, C, HW] <-> NHWC [B, HW, Cpad] (zero-filled pad channels),
 * with an optional affine on the way in (alpha * x + beta) and a per-sample scale on the way out. */
int idiff_nchw_to_nhwc_f32(const float *x, float *y, int B, int C, int HW, int Cpad, float alpha, float beta,
                           void *stream);
int idiff_nhwc_to_nchw_f32(const float *x, float *y, int B, int C, int HW, int Cpad, const float *rowscale,
                           void *stream);
/* Perturbation of one data point into `rows` noisy copies (dim_reduction.py:178-182):
 * out[r, :] = mean_coeff[r] * x[:] + std[r] * z[r, :]; mean_coeff NULL means 1 (VE SDE, sde_lib.py:346). */
int idiff_perturb_f32(const float *x, const float *z, const float *std_, const float *mean_coeff, float *out,
                      int64_t rows, int64_t D, void *stream);
/* Same perturbation with the Gaussian draw generated in the kernel (replaces `z = torch.randn_like(batch)` of
 * dim_reduction.py:181): Philox4x32-10 keyed by `seed`, counter = position of the element in the point's logical
 * [total_rows, D] noise matrix (row0 = first row of this launch), Box-Muller.  The draw for an element does not depend on
 * how rows are cut into launches.  D % 4 == 0.  z_out (optional, [rows, D]) receives the N(0,1) values for tests. */
int idiff_perturb_randn_f32(const float *x, const float *std_, const float *mean_coeff, float *out, int64_t rows,
                            int64_t D, int64_t row0, uint64_t seed, float *z_out, void *stream);
/* Nearest x2 upsample / 2x2 mean downsample of NHWC tensors (naive_upsample_2d / naive_downsample_2d,
 * models/up_or_down_sampling.py:59-69; BeatGANs Upsample/Downsample, models/BeatGANsblocks.py:335-396). */
int idiff_resample2x_nhwc_f32(const float *x, float *y, int B, int H, int W, int C, int up, void *stream);

/* ------------------------------------------------------------------ spectrum of the centred score matrix */

/* Replaces `scores - scores.mean(0)` + `torch.linalg.svd(...)` of dim_reduction.py:193-198 for a batch of P
 * score matrices S[p] (M x D fp32, row-major, contiguous): D singular values per matrix, descending; for M < D the
 * reference's min(M, D) values are the leading M of them (the rest are zeros up to rounding).  Method: fp64 column means -> fp64 Gram of the centred columns on v_mfma_f64_16x16x4 ->
 * tridiagonalisation (fp64, idiff_symtridiag_f64) -> Sturm bisection -> sqrt.  Products of fp32 inputs are exact in
 * fp64, so the squared condition number costs nothing at the 1e-4 tolerance.
 * workspace: idiff_spectrum_workspace_bytes(P, M, D) bytes; sv: [P, D] fp32.
 * `eig_out` (optional, [P, D] fp64) receives the Gram eigenvalues (ascending) for diagnosis. */
int64_t idiff_spectrum_workspace_bytes(int P, int M, int D);
int idiff_spectrum_f32(const float *S, int P, int M, int D, void *workspace, int64_t workspace_bytes, float *sv,
                       double *eig_out, void *stream);
/* The stages, exported for the parity tests and the profiler. */
/* scratch: P * 32 * D doubles (deterministic two-stage column sums). */
int idiff_colmean_f64(const float *S, int P, int M, int D, double *mean, double *scratch, void *stream);
int idiff_centered_gram_f64(const float *S, const double *mean, int P, int M, int D, double *G, void *stream);
/* Row-sharded single point (one point's rows spread over the ranks): the upper triangle of the
 * centred Gram for rows [row0, row1) only (tile-aligned: multiples of 64, or D), so that block b can be all-reduced
 * while block b + 1 is computed; idiff_symmetrize_upper_f64 mirrors the reduced upper triangle afterwards.  G is
 * [D][D] fp64 and should be zeroed first (the part left of a block's diagonal tile is not written). */
int idiff_centered_gram_rows_f64(const float *S, const double *mean, int M, int D, int row0, int row1, double *G,
                                 void *stream);
int idiff_symmetrize_upper_f64(double *G, int D, void *stream);
/* G [P][D][D] symmetric (both triangles), overwritten -> diag/offdiag [P][D] of a similar tridiagonal matrix.
 * D <= 128: one workgroup per matrix in LDS.  Larger D: two-stage -- blocked reduction to a band of half-width 32
 * (CholeskyQR2 + Householder-reconstruction panels, compact-WY rank-64 trailing updates on v_mfma_f64_16x16x4) and
 * bulge chasing on the compact band; if the panels' annihilation residual exceeds 1e-11 ||G||_F the outputs are NaN
 * (never a silently wrong spectrum; IDIFF_TRIDIAG_ONESTAGE selects the unblocked Householder sweep instead).
 * scratch: idiff_symtridiag_scratch_doubles(D) doubles (shared by the P matrices, which are processed in turn). */
int64_t idiff_symtridiag_scratch_doubles(int D);
/* stage 1 alone (D > 128): on return the first D * idiff_symband_ld() doubles of scratch hold the lower band,
 * band[j * ld + k] = B[j + k][j], k <= 32 (the rest of a column is bulge room, zero). */
int idiff_symband_ld(void);
int idiff_symband_f64(double *G, int D, double *scratch, void *stream);
int idiff_symtridiag_f64(double *G, int P, int D, double *diag, double *offdiag, double *scratch, void *stream);
/* Which path idiff_symtridiag_f64 takes for a D x D matrix on the current device: 0 LDS-resident (D <= 128), 1 two-stage
 * with the single-launch systolic bulge chase, 2 two-stage with the wavefront chase (the systolic kernel's
 * ceil(D / 32) mutually waiting workgroups exceed HALF of what the device can hold resident -- asked of the
 * runtime per device -- or IDIFF_CHASE_WAVEFRONT is set), 3 one-stage sweep (IDIFF_TRIDIAG_ONESTAGE). */
int idiff_symtridiag_plan(int D);
int idiff_tridiag_eigvals_f64(const double *diag, const double *offdiag, int P, int D, double *eig, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* IDIFF_HIP_H */
