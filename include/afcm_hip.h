/*
 * afcm_hip.h -- C ABI of libafcm_hip.so: the MI355X (gfx950) kernels behind the
 * `--model stylegan3` generator hot path of zhiyuns/AFCM.
 *
 * This is the drop-in boundary.  Every entry point replaces one function of the reference's
 * native plugins (pybind modules JIT-built by torch_utils/custom_ops.py:59-155); the reference
 * interface each one stands in for is cited above it.  Shorthand:
 *   SG3OPS = models/networks/stylegan3/torch_utils/ops   (in the reference tree)
 *   NET    = models/networks/stylegan3/networks_stylegan3.py
 *
 * Conventions
 *   - plain C: device pointers + sizes; no torch / ATen types.  All tensor pointers are DEVICE
 *     pointers into contiguous NCHW storage; filters are DEVICE float32 arrays as in the
 *     reference (filtered_lrelu.cpp:26).
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream).  Launches are
 *     asynchronous; no call allocates, synchronises or keeps global mutable state, so all
 *     of them are re-entrant across streams and capturable into a hipGraph.  (The reference
 *     is not: global filter buffers g_fbuf/c_fbuf, filtered_lrelu.cu:77-78.)
 *   - return value: 0 = launched; AFCM_E_NOKERNEL (-1) = no specialised kernel for these
 *     parameters (the reference's `return_code = -1`, filtered_lrelu.cpp:52-56 -- the host
 *     falls back to the generic upfirdn2d + act path); AFCM_E_INVALID (-2) = argument check
 *     failed (the reference's TORCH_CHECK); >= 1000 = 1000 + hipError_t of the launch.
 *   - dtype codes: AFCM_F32, AFCM_F16, AFCM_BF16 (bf16 is new capability; the reference
 *     plugin accepts half/float only).  Arithmetic is always fp32 inside the kernels.
 */
#ifndef AFCM_HIP_H
#define AFCM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AFCM_ABI_VERSION 13  /* 13 (r06): + afcm_noop, afcm_pool_blocks_fwd / _bwd, afcm_adam_multi_capturable, afcm_conv2d_wgrad_dots_ld, afcm_mapping_input_bwd_workspace_bytes, afcm_axpy_planes, afcm_l1_partials, afcm_l1_grad, afcm_fc_act_fwd / _bwd, afcm_mapping_input_fwd / _bwd (additions only; see the end of this header for the r06 entry points).  12 (r05): + afcm_conv2d_block_k_ks, afcm_conv2d_pack_weights_bk; the packed layout's K-chunk depends on (dtype, kernel size): 16-bit 3x3 images are [nkc][9][rows_pad][32] for the v_mfma 16x16x32 kernel (the pack / conv entry points keep their signatures).  11 (r04): + afcm_amax_bits, afcm_split16, afcm_conv2d_pack_split, afcm_conv2d_split, afcm_unscale, afcm_plane_dot_parts (additions only).  10 (r04): + afcm_filtered_lrelu_args.clamp_flags (appended), afcm_plane_dot_gated_ld; the runtime getenv switches are gone.  9 (r03): + afcm_affine_bank_*, afcm_modulation_bank_*, afcm_conv2d_pack_bank, afcm_conv2d_stride2 (additions only; every v8 entry point and struct is unchanged) */

enum { AFCM_F32 = 0, AFCM_F16 = 1, AFCM_BF16 = 2 };
enum { AFCM_OK = 0, AFCM_E_NOKERNEL = -1, AFCM_E_INVALID = -2 };
enum { AFCM_SIGNS_NONE = 0, AFCM_SIGNS_WRITE = 1, AFCM_SIGNS_READ = 2 };

/* Library / device introspection. */
int afcm_abi_version(void);
/* Human-readable description of the last AFCM_E_INVALID on this thread. */
const char* afcm_last_error(void);
/* An empty one-wave launch on `stream`: what a launch costs the host and the device's front end with no work behind it
 * (bench.py's `host_calibration`).  No counterpart in the reference (its launches go through ATen). */
int afcm_noop(void* stream);

/* ------------------------------------------------------------------------------------------
 * filtered_lrelu -- fused bias -> zero-insert upsample -> pad/crop -> FIR(fu) * up^2 ->
 *                   * gain -> leaky ReLU -> clamp -> FIR(fd) -> decimate.
 *
 * Replaces plugin `filtered_lrelu(x, fu, fd, b, si, up, down, px0, px1, py0, py1, sx, sy, gain,
 * slope, clamp, flip_filter, writeSigns) -> (y, so, return_code)`
 *   SG3OPS/filtered_lrelu.cpp:16-209, kernels SG3OPS/filtered_lrelu.cu:139-1099.
 *
 * Sign tensor (2 bits per element of the upsampled grid, bit0 = value was negative, value 2 =
 * clamped): uint8 [N, C, sh, swb] with element x of a row in byte x>>2 at bits 2*(x&3) -- the
 * reference's shape and packing (filtered_lrelu.cpp:87-94).  In WRITE mode sx = sy = 0 is
 * required (the reference never writes with an offset: filtered_lrelu.py:115,263).
 * In READ mode element (X, Y) of this call's upsampled grid uses code (X + sx, Y + sy); codes
 * outside the tensor leave the value unchanged (filtered_lrelu.cu:564-571).
 * Separable filters: fuh == 0 / fdh == 0 and fu/fd hold fuw / fdw taps (the reference's
 * "shape [n, 0] indicates separable", filtered_lrelu.cpp:49-50).
 * ---------------------------------------------------------------------------------------- */
typedef struct afcm_filtered_lrelu_args {
    const void* x;          /* [N, C, xh, xw]                                  */
    void*       y;          /* [N, C, yh, yw]   (allocated by the caller)      */
    const void* b;          /* [C] same dtype as x, or NULL                    */
    uint8_t*    signs;      /* [N, C, sh, swb] or NULL                         */
    const float* fu;        /* device, fuw taps (separable) or fuh*fuw         */
    const float* fd;        /* device, fdw taps (separable) or fdh*fdw         */
    int32_t dtype;          /* AFCM_F32 / AFCM_F16 / AFCM_BF16                 */
    int32_t n, c, xh, xw, yh, yw;
    int32_t fuw, fuh, fdw, fdh;     /* fuh/fdh == 0: separable                 */
    int32_t up, down;
    int32_t px0, px1, py0, py1;
    int32_t sx, sy;         /* sign offsets                                    */
    int32_t sh, swb;        /* sign tensor rows / bytes per row                */
    float   gain, slope, clamp;     /* clamp = +inf to disable                 */
    int32_t flip_filter;
    int32_t sign_mode;      /* AFCM_SIGNS_*                                    */
    void*   workspace;      /* NULL, or afcm_filtered_lrelu_workspace_bytes() of device memory filled by
                               afcm_filtered_lrelu_prepare() for THIS (fu, fd, up, down, px0, py0, gain, flip, dtype):
                               enables the matrix-core kernels for 16-bit dtypes                              */
    int32_t sign_layout;    /* 0: row-major 2-bit codes (reference layout); 1: row-quad bytes written by the
                               matrix-core kernels ([N,C,ceil(sh/4),swq]); set by afcm_filtered_lrelu_shapes(),
                               must be passed back unchanged with the sign tensor in READ mode                 */
    int32_t plane_sum_slots;/* set by afcm_filtered_lrelu_shapes(): partial sums per plane the selected kernel emits (0: none) */
    float*  plane_sum;      /* NULL, or fp32 [N*C][plane_sum_slots]: the matrix-core kernels store the sum of every output tile
                               (no atomics); summed over slots and N in a backward call this is the bias gradient
                               (db = dx.sum([0,2,3]), SG3OPS/filtered_lrelu.py:266) without a second pass over dx.        */
    const float* oscale;    /* NULL, or fp32 [N*C]: y = (filtered_lrelu(...) + skip) * oscale[n*C + c].  Matrix-core kernels only.
                               Lets the caller fold per-plane factors that would otherwise be separate passes over the tensor:
                               the NEXT layer's style modulation s[n,i] (NET:46-47, forward) and the demodulation d[n,o]
                               of the conv that produced x (NET:50-52, backward: the op is linear in dy given the codes). */
    const void*  skip;      /* NULL, or [N, C, yh, yw] of x's dtype: the encoder feature added after the activation
                               (x + x_skip, NET:376-377), added before oscale.  Matrix-core kernels only.                */
    const float* oscale2;   /* NULL, or a second fp32 [N*C] factor multiplied with oscale (backward: demodulation x styles) */
    int32_t x_pitch, y_pitch, skip_pitch;
                            /* row pitch of x / y / skip in ELEMENTS; 0 = dense (pitch = width).  A pitched tensor is
                               [N, C, H, pitch] in memory with the first W columns of every row meaningful (MI355X layout of the
                               16-bit activation stream: rows start on 64-byte boundaries).  Only the
                               kernels for which afcm_filtered_lrelu_shapes() reports row_pitch_ok take a pitch; they accept
                               x_pitch / skip_pitch up to width + 128 and y_pitch up to 64 * ceil(yw / 64) (AFCM_E_INVALID beyond),
                               and with y_pitch set they write finite values to every column of y up to the pitch (columns
                               >= yw are padding).                                                                              */
    int32_t row_pitch_ok;   /* set by afcm_filtered_lrelu_shapes(): 1 if the kernel selected for these arguments accepts pitches */
    int32_t* clamp_flags;   /* NULL, or int32 [N*C][plane_sum_slots] (sign-writing calls of the wave kernels, i.e. plane_sum_slots > 0 and
                               no bias operand): slot = 1 if that strip's activations could reach the clamp (it then took the exact
                               per-element path), else 0.  Every slot is written.  A plane whose slots are all 0 went through linear
                               filters around a pure leaky ReLU: there filtered_lrelu is positively homogeneous of degree 1, which
                               the caller may use to derive <dL/dy, y> from <g, z> (afcm_plane_dot_gated_ld).                      */
} afcm_filtered_lrelu_args;

/* Output / sign-tensor geometry for the arguments above (filtered_lrelu.cpp:61-94). Fills yh, yw
 * and, for WRITE mode, sh, swb.  Pure host arithmetic. */
int afcm_filtered_lrelu_shapes(afcm_filtered_lrelu_args* a);
int afcm_filtered_lrelu(const afcm_filtered_lrelu_args* a, void* stream);

/* Matrix-core path (16-bit dtypes, the separable 12/24-tap cases of the generator): the FIR passes run as banded
 * Toeplitz products on v_mfma_f32_16x16x32_{bf16,f16}.  The constant Toeplitz fragments depend only on the layer
 * configuration; build them once per layer with afcm_filtered_lrelu_prepare() into `workspace` (this replaces the
 * reference's per-call setup_filters_kernel + copy to __constant__ memory, filtered_lrelu.cu:87-117, without any
 * global state).  Returns AFCM_E_NOKERNEL when the configuration has no matrix-core kernel. */
int64_t afcm_filtered_lrelu_workspace_bytes(void);
int afcm_filtered_lrelu_prepare(const afcm_filtered_lrelu_args* a, void* stream);

/* In-place gain -> leaky ReLU -> clamp with sign write/read, used by the generic fallback.
 * Replaces plugin `filtered_lrelu_act_(x, si, sx, sy, gain, slope, clamp, writeSigns) -> so`
 *   SG3OPS/filtered_lrelu.cpp:213-290, kernel SG3OPS/filtered_lrelu.cu:1105-1211.
 * signs: uint8 [N, C, h, swb] with swb = ceil16(w)/4 in WRITE mode. */
int afcm_filtered_lrelu_act(void* x, uint8_t* signs, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w,
                            int32_t sh, int32_t swb, int32_t sx, int32_t sy, float gain, float slope, float clamp,
                            int32_t sign_mode, void* stream);

/* ------------------------------------------------------------------------------------------
 * upfirdn2d -- zero-insert upsample, pad/crop, 2-D FIR, decimate.
 * Replaces plugin `upfirdn2d(x, f, upx, upy, downx, downy, padx0, padx1, pady0, pady1, flip, gain) -> y`
 *   SG3OPS/upfirdn2d.cpp:16-98, kernels SG3OPS/upfirdn2d.cu:29-375.
 * f is a device float32 [fh, fw] array (a separable filter is two calls, as in
 * SG3OPS/upfirdn2d.py:244-245).  y: [N, C, yh, yw] with yw = (xw*upx + padx0 + padx1 - fw + downx) / downx.
 * ---------------------------------------------------------------------------------------- */
int afcm_upfirdn2d(void* y, const void* x, const float* f, int32_t dtype, int32_t n, int32_t c, int32_t xh, int32_t xw,
                   int32_t yh, int32_t yw, int32_t fh, int32_t fw, int32_t upx, int32_t upy, int32_t downx, int32_t downy,
                   int32_t padx0, int32_t pady0, int32_t flip, float gain, void* stream);

/* ------------------------------------------------------------------------------------------
 * bias_act -- y = clamp(act(x + b) * gain); grad = 1: first-order backward from saved x/y;
 * grad = 2: second-order.  act: 1 linear, 2 relu, 3 lrelu, 4 tanh, 5 sigmoid, 6 elu, 7 selu,
 * 8 softplus, 9 swish (the reference's cuda_idx, SG3OPS/bias_act.py:21-31).
 * Replaces plugin `bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp) -> y`
 *   SG3OPS/bias_act.cpp:32-90, kernel SG3OPS/bias_act.cu:23-147.
 * x is viewed as [outer, nb, inner] with the bias indexed by the middle dimension
 * (nb = 0 / b = NULL: no bias).  xref / yref / dy may be NULL when the mode does not need them.
 * ---------------------------------------------------------------------------------------- */
int afcm_bias_act(void* y, const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                  int32_t dtype, int64_t numel, int64_t inner, int32_t nb, int32_t grad, int32_t act, float alpha,
                  float gain, float clamp, void* stream);


/* ------------------------------------------------------------------------------------------
 * Dense convolution behind modulated_conv2d / the encoder convs.
 *
 * The reference has no native conv: `modulated_conv2d` (NET:25-64) builds per-sample weights
 * and calls a grouped F.conv2d (cuDNN) through conv2d_gradfix.conv2d (SG3OPS/conv2d_gradfix.py:37-58),
 * and EncoderLayer calls conv2d_gradfix.conv2d directly (NET:505).  These entry points replace that
 * conv call with MFMA implicit-GEMM kernels using the equivalent shared-weight form
 *     y[n,o] = oscale[n,o] * conv(W, x[n]),   x pre-scaled per (n,i) plane by afcm_scale_planes.
 * All of them are cross-correlations (F.conv2d semantics) with stride 1, k in {1,3}, 0 <= pad <= k-1.
 * ---------------------------------------------------------------------------------------- */

/* K-chunk (channels) of the packed weight layout.
 * afcm_conv2d_block_k(dtype): the fp32, 1x1 and stride-2 (afcm_conv2d_stride2) kernels -- 16 for 16-bit, 8 for fp32;
 * afcm_conv2d_block_k_ks(dtype, ks): the stride-1 kernel for this kernel size -- 32 for 16-bit 3x3 (one v_mfma_f32_16x16x32 K step),
 * else as above.  afcm_conv2d[_ld], afcm_conv2d_split and the pack entry points below use this one. */
int afcm_conv2d_block_k(int32_t dtype);
int afcm_conv2d_block_k_ks(int32_t dtype, int32_t ks);

/* Pack fp32 weights w[cout][cin][k][k] into the kernel layout [ceil(cols/BK)][k*k][rows_pad][BK] of
 * `dtype` (BK = afcm_conv2d_block_k_ks(dtype, ks)), zero padded.  mode 0: forward (rows = cout, cols = cin).  mode 1: data gradient
 * (rows = cin, cols = cout, taps flipped).  rows_pad: multiple of 64, >= rows. */
int afcm_conv2d_pack_weights(void* dst, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks,
                             int32_t mode, int32_t rows_pad, void* stream);

/* Both images of one layer in ONE launch (training: the forward packs what its backward will need): dst_fwd as mode 0,
 * dst_dgrad as mode 1 of afcm_conv2d_pack_weights; either may be NULL.  Destinations 32-byte aligned. */
int afcm_conv2d_pack_weights2(void* dst_fwd, void* dst_dgrad, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks,
                              int32_t rows_pad_fwd, int32_t rows_pad_dgrad, void* stream);
/* afcm_conv2d_pack_weights with an explicit K-chunk: block_k = afcm_conv2d_block_k(dtype) (the image afcm_conv2d_stride2 reads) or
 * afcm_conv2d_block_k_ks(dtype, ks). */
int afcm_conv2d_pack_weights_bk(void* dst, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks, int32_t mode,
                                int32_t rows_pad, int32_t block_k, void* stream);

/* y[n, cout, h+2*pad-k+1, w+2*pad-k+1] = oscale[n*cout+o] * sum W * x + obias[o].   oscale / obias (fp32) may be NULL.
 * obias is the layer bias the reference adds at the head of filtered_lrelu (x + b before the padding, NET:371 ->
 * SG3OPS/filtered_lrelu.py:133): the conv output IS the whole unpadded image, so adding it in this epilogue is the same
 * arithmetic and saves the conversion work in filtered_lrelu's input staging.
 * For the data gradient call it with the mode-1 packing, cin/cout swapped and pad' = k-1-pad. */
int afcm_conv2d(void* y, const void* x, const void* wpacked, const float* oscale, const float* obias, int32_t dtype, int32_t n,
                int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t rows_pad, void* stream);
/* The same with row-pitched activations (x: [N, cin, h, x_pitch], y: [N, cout, P, y_pitch] in memory, the first w / Q columns of
 * a row meaningful; 0 = dense).  MI355X layout of the 16-bit activation stream: rows start on 64-byte boundaries.
 * Pitches are taken by the 16-bit 3x3 kernel only.  Columns >= w of x are never used; columns >= Q of y are padding: the kernel
 * writes finite values up to the next multiple of 8 past Q (the tail of a row's last 8-pixel granule) and leaves the rest of the
 * padding UNWRITTEN. */
int afcm_conv2d_ld(void* y, const void* x, const void* wpacked, const float* oscale, const float* obias, int32_t dtype, int32_t n,
                   int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t rows_pad, int32_t x_pitch,
                   int32_t y_pitch, void* stream);

/* Weight gradient dw[cout][cin][k][k] (fp32) = sum_n sum_pixels dy[n,o,p,q] * x[n,i,p+r-pad,q+s-pad].
 * workspace: fp32 [afcm_conv2d_wgrad_splits(...)][cout][cin][k][k]. */
int afcm_conv2d_wgrad_splits(int32_t n, int32_t cout, int32_t cin, int32_t p_rows);
int afcm_conv2d_wgrad(float* dw, float* workspace, const void* dy, const void* x, int32_t dtype, int32_t n, int32_t cin,
                      int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, void* stream);
/* The same with row-pitched operands (0 = dense); 16-bit, 3x3 pad 2 or 1x1 pad 0 only.  Columns >= w of x are never used; of dy,
 * the columns up to the next multiple of 8 past Q must hold FINITE values (they multiply zeros): afcm_conv2d_ld writes exactly those
 * columns, afcm_filtered_lrelu writes the whole pitch. */
int afcm_conv2d_wgrad_ld(float* dw, float* workspace, const void* dy, const void* x, int32_t dtype, int32_t n, int32_t cin,
                         int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t dy_pitch, int32_t x_pitch, void* stream);
/* afcm_conv2d_wgrad_ld that ALSO returns dots[n][cin] = sum_{o, tap} wq[o][i][tap] * dW_n[o][i][tap] (r06), dW_n = image n's share of the weight
 * gradient, wq = wref [cout][cin][ks][ks] (fp32) rounded to the operands' 16-bit type: by bilinearity this is <x[n, i], dx[n, i]> with
 * dx = conv^T(wq, dy) -- for a conv whose input carries a per-plane style factor s (modulated_conv2d, NET:41-63) the gradient of s times s,
 * which the host otherwise takes from a pass over x and dx (afcm_plane_dot_ld).  The K range is split over workgroups WITHIN images; available
 * for the 16-bit granule kernel (3x3 pad 2, 1x1 pad 0) when the split count of afcm_conv2d_wgrad_splits() is a multiple of n <= 64, else
 * AFCM_E_NOKERNEL (nothing launched: call afcm_conv2d_wgrad_ld and afcm_plane_dot_ld).  Same workspace as afcm_conv2d_wgrad_ld. */
int afcm_conv2d_wgrad_dots_ld(float* dw, float* dots, float* workspace, const void* dy, const void* x, const float* wref, int32_t dtype, int32_t n,
                              int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t dy_pitch, int32_t x_pitch,
                              void* stream);

/* y[plane, :] = x[plane, :] * scale[plane] with dtype conversion (style modulation s[n,i] of NET:46-47 and the
 * demodulation d[n,o] of NET:50-52 applied to activations instead of weights).  scale may be NULL (pure cast). */
int afcm_scale_planes(void* y, const void* x, const float* scale, int32_t dtype_in, int32_t dtype_out, int64_t planes,
                      int32_t hw, void* stream);

/* out[plane] = sum_i a[plane,i] * b[plane,i]  (b == NULL: plain sum); fp32 accumulation.  Used for the style /
 * demodulation / bias gradients. */
int afcm_plane_dot(float* out, const void* a, const void* b, int32_t dtype, int64_t planes, int32_t hw, void* stream);
/* The same over planes of h rows x w columns with row pitches (elements; 0 = dense); padding columns are never read. */
int afcm_plane_dot_ld(float* out, const void* a, const void* b, int32_t dtype, int64_t planes, int32_t h, int32_t w,
                      int32_t a_pitch, int32_t b_pitch, void* stream);
/* The demodulation gradient's dot product by homogeneity (fused layer node, NET:41-57 backward): for a plane whose clamp_flags
 * (afcm_filtered_lrelu_args, [planes][slots]) are all 0,  out = out_scale * (gz - next_scale * gskip)  -- <dys, y> = d <dL/dy, y> =
 * d (<g, z> - s_next <g, skip>) by Euler's identity for the degree-1 homogeneous filtered_lrelu -- and neither a nor b is read;
 * a flagged plane gets the real dot product sum a * b, and so does a plane whose two sums cancel below 1/8 of their size
 * (|gz - next_scale gskip| * 8 < |gz| + |next_scale gskip|: a skip branch far larger than the layer's own output would amplify the
 * 16-bit rounding of z by that ratio).  next_scale / gskip may be NULL (1 / 0). */
int afcm_plane_dot_gated_ld(float* out, const void* a, const void* b, int32_t dtype, int64_t planes, int32_t h, int32_t w,
                            int32_t a_pitch, int32_t b_pitch, const int32_t* flags, int32_t slots, const float* out_scale,
                            const float* gz, const float* next_scale, const float* gskip, void* stream);

/* ------------------------------------------------------------------------------------------
 * The small-tensor half of modulated_conv2d (NET:41-57), fp32, forward and exact backward.  The reference runs it as
 * ~35 eager elementwise / reduction / matmul launches per layer and step; these are 1 + 1 + 1 + 2 launches.
 *   weight_norm:  w_hat[o] = w[o] * rsqrt(mean_{i,k} w[o]^2) (NET:42), wsq[o,i] = sum_k w_hat[o,i,k]^2, scale[o] kept for backward.
 *                 bwd: dw from g_hat (dL/dw_hat, may be NULL) and g_wsq (dL/dwsq, may be NULL).
 *   style_coefs:  s_hat = t * rsqrt(mean_{n,i} t^2) (NET:43, whole batch; r[0] keeps the factor),
 *                 d[n,o] = rsqrt(sum_i s_hat[n,i]^2 wsq[o,i] + 1e-8) (NET:50-52 factorised over the shared weights),
 *                 s_eff = s_hat * rsqrt(magnitude[0]) (input gain, NET:346,55-57; magnitude NULL: 1).
 *                 demodulate == 0 (ToRGB, NET:362): s_eff = t * gain only, d / wsq unused.
 *                 bwd: dt from g_s (dL/ds_eff) and g_d (dL/dd); g_wsq (may be NULL) = dL/dwsq.
 *                 workspace: n*cin + n*cout + n*ceil(cin/64) floats.
 * ---------------------------------------------------------------------------------------- */
int afcm_weight_norm_fwd(float* w_hat, float* wsq, float* scale, const float* w, int32_t cout, int32_t cin, int32_t kk, void* stream);
int afcm_weight_norm_bwd(float* dw, const float* g_hat, const float* g_wsq, const float* w_hat, const float* scale, int32_t cout,
                         int32_t cin, int32_t kk, void* stream);
int afcm_style_coefs_fwd(float* s_eff, float* d, float* r, const float* t, const float* wsq, const float* magnitude, int32_t n, int32_t cin,
                         int32_t cout, int32_t demodulate, void* stream);
int afcm_style_coefs_bwd(float* dt, float* g_wsq, float* workspace, const float* g_s, const float* g_d, const float* t, const float* d,
                         const float* wsq, const float* magnitude, const float* r, int32_t n, int32_t cin, int32_t cout, int32_t demodulate,
                         void* stream);

/* Small-tensor tail of a fused layer's backward (afcm_amd/torch_utils/ops/fused_layer.py): from the per-tile plane sums of
 * dys that the transposed filtered_lrelu emitted (psum [n, cout, slots]) and two plane dot products, the gradients of the
 * bias (db [cout] = sum_n ps / d), of the next layer's styles (d_next [n, cout] = <g, z> / s_next) and of the demodulation
 * coefficients (d_out [n, cout] = (<dys, y> - b ps) / d^2).  Any of db / d_next / d_out may be NULL; out_scale, bias may be NULL. */
int afcm_layer_bwd_coefs(float* db, float* d_next, float* d_out, const float* psum, int32_t slots, const float* out_scale,
                         const float* next_scale, const float* bias, const float* gz, const float* dysy, int32_t n, int32_t cout,
                         void* stream);

/* ------------------------------------------------------------------------------------------
 * Generator update: gradient scale (1/world after the all-reduce), the NaN/Inf scrub of the gradients
 * (models/stylegan3_model.py:122-124,132-134: torch.nan_to_num(grad, nan=0, posinf=1e5, neginf=-1e5)) and the Adam step
 * (models/comodgan_model.py:19-20: torch.optim.Adam(lr, betas=(0, 0.99), eps=1e-8)) for EVERY parameter tensor in one
 * launch.  Replaces ~110 nan_to_num launches + torch's multi-tensor Adam passes; arithmetic follows torch's foreach
 * implementation operation by operation.  All tensors fp32.
 * `table` is a DEVICE array of n entries sorted by chunk0 (chunk0 = number of afcm_adam_chunk_elems()-sized chunks of
 * the tensors before this one); total_chunks = chunk0 + chunks of the last entry.  step_size = lr / (1 - beta1^t),
 * bias_correction2_sqrt = sqrt(1 - beta2^t), one_minus_beta* = 1 - beta*: all evaluated on the host in double as torch does
 * (1.f - 0.999f differs from (float)(1.0 - 0.999) by 5e-5 relative).
 * write_grad != 0 stores the scaled + scrubbed gradient back (what the reference leaves in .grad).
 * ---------------------------------------------------------------------------------------- */
typedef struct afcm_adam_entry {
    void*   p;        /* parameter, updated in place     */
    void*   g;        /* gradient                        */
    void*   m;        /* exp_avg, updated in place       */
    void*   v;        /* exp_avg_sq, updated in place    */
    int64_t numel;
    int64_t chunk0;
} afcm_adam_entry;
int32_t afcm_adam_chunk_elems(void);
int afcm_adam_multi(const afcm_adam_entry* table, int32_t n, int64_t total_chunks, float step_size, float beta1, float beta2,
                    float one_minus_beta1, float one_minus_beta2, float bias_correction2_sqrt, float eps, float grad_scale, int32_t scrub, float posinf, float neginf,
                    int32_t write_grad, void* stream);
/* afcm_adam_multi for a step captured into a hipGraph (r06): the step count is a DEVICE float (step_dev[0], advanced by one before the update)
 * and the bias corrections 1 - beta^t are formed on the device from it (in double), so a replayed launch applies the corrections of ITS step;
 * `lr` is the plain learning rate.  Same update arithmetic otherwise. */
int afcm_adam_multi_capturable(const afcm_adam_entry* table_dev, int32_t n, int64_t total_chunks, float* step_dev, float lr, float beta1, float beta2,
                               float eps, float grad_scale, int32_t scrub, float posinf, float neginf, int32_t write_grad, void* stream);

/* ------------------------------------------------------------------------------------------
 * The style affine layers of all SynthesisLayers at once (NET:349-352 `styles = self.affine(cat(w, global_w))`, FullyConnectedLayer
 * NET:69-104; ToRGB's extra factor NET:351 rides in alpha / beta): for layer l
 *     y_l[n][c] = alpha_l * sum_k x_l[n][k] W_l[c][k] + beta_l * b_l[c],   x_l[n] = [ w[n][w_index_l][0:kw] | g[n][0:kg] ]
 * and its gradients (dW_l = alpha_l dy_l^T x_l, db_l = beta_l sum_n dy_l, dws[n][l] / dg[n] = the two halves of alpha_l dy_l W_l, dg summed
 * over the layers).  All fp32, rows 16-byte aligned, kw and kg multiples of 4, kw + kg <= 1536, at most AFCM_AFFINE_MAX layers;
 * AFCM_E_NOKERNEL otherwise (the caller runs the layers one by one).  Replaces 15 cat + 15 GEMM launches forward and 30 GEMMs + 15
 * reductions + 14 accumulations backward by 1 + 3 launches.  No atomics: bit-reproducible.
 * y / dy / dweight / dbias are HOST arrays of `layers` device pointers ([n][cout_l], [cout_l][kw + kg], [cout_l]); a NULL dy_l means
 * zeros, NULL dweight_l / dbias_l are skipped, dweight == dbias == NULL skips the weight pass, dws == NULL the input pass.
 * dws: [n][layers][kw], dg: [n][kg]; workspace: afcm_affine_bank_workspace_bytes() bytes (input pass only).
 * ---------------------------------------------------------------------------------------- */
#define AFCM_AFFINE_MAX 16
typedef struct afcm_affine_bank {
    int32_t layers, n, kw, kg;
    int64_t w_stride_n, w_stride_l;               /* elements between batch rows / between latents of the ws tensor */
    const float* w;                               /* ws: w[n * w_stride_n + w_index_l * w_stride_l + k] */
    const float* g;                               /* [n][kg] or NULL when kg == 0 */
    const float* weight[AFCM_AFFINE_MAX];         /* [cout_l][kw + kg] */
    const float* bias[AFCM_AFFINE_MAX];           /* [cout_l] or NULL */
    int32_t cout[AFCM_AFFINE_MAX];
    int32_t w_index[AFCM_AFFINE_MAX];
    float alpha[AFCM_AFFINE_MAX], beta[AFCM_AFFINE_MAX];
} afcm_affine_bank;
int64_t afcm_affine_bank_workspace_bytes(const afcm_affine_bank* a);
int afcm_affine_bank_fwd(const afcm_affine_bank* a, float* const* y, void* stream);
int afcm_affine_bank_bwd(const afcm_affine_bank* a, const float* const* dy, float* const* dweight, float* const* dbias, float* dws, float* dg,
                         void* workspace, void* stream);

/* ----------------------------------------------------------------------------------------
 * Modulation bank: afcm_weight_norm_* and afcm_style_coefs_* (above) for a LIST of layers in 2 launches forward and 3 backward, whatever
 * the number of layers -- the decoder's 15 SynthesisLayers take 29 + 44 launches one by one (NET:41-57, 346-352 per layer).  Same
 * arithmetic, same kernels' bodies: bit-identical to the per-layer entry points.  Per layer, forward:
 *     demodulate != 0:  w [cout][cin][kk] -> w_hat (same shape), wsq [cout][cin], scale [cout];  t [n][cin] -> s_eff [n][cin], d [n][cout], r [1]
 *     demodulate == 0:  t -> s_eff, r (= 1); w / w_hat / wsq / scale / d unused (the caller convolves with w itself)
 * backward (g_hat / g_s / g_d: gradients of w_hat / s_eff / d, NULL = zeros): dw [cout][cin][kk] (NULL: skipped), dt [n][cin];
 * reads the forward's w_hat, wsq, scale, d, r, t, magnitude; workspace: afcm_modulation_bank_workspace_floats() floats per layer.
 * `layers` is a HOST array of `count` <= AFCM_MODULATION_MAX entries; it is copied into the kernel arguments.
 * ---------------------------------------------------------------------------------------- */
#define AFCM_MODULATION_MAX 16
typedef struct afcm_modulation_layer {
    int32_t cout, cin, kk, demodulate;
    const float* w;
    const float* t;
    const float* magnitude;                       /* [1] (input_gain = rsqrt(magnitude), NET:346) or NULL (gain 1) */
    float* w_hat;
    float* wsq;
    float* scale;
    float* s_eff;
    float* d;
    float* r;
    const float* g_hat;                           /* backward only from here on */
    const float* g_s;
    const float* g_d;
    float* dw;
    float* dt;
    float* workspace;
} afcm_modulation_layer;
int64_t afcm_modulation_bank_workspace_floats(int32_t n, int32_t cin, int32_t cout, int32_t demodulate);
int afcm_modulation_bank_fwd(const afcm_modulation_layer* layers, int32_t count, int32_t n, void* stream);
int afcm_modulation_bank_bwd(const afcm_modulation_layer* layers, int32_t count, int32_t n, void* stream);

/* ----------------------------------------------------------------------------------------
 * afcm_conv2d_pack_weights2 for a LIST of layers in one launch: every conv weight of the generator (the encoder's parameters, the
 * decoder's normalised weights from afcm_modulation_bank_fwd) exists before the first convolution of a step, so its 29 pack launches
 * (8 us each) become two (3x3 kernels; the 1x1 ToRGB keeps its own).  Same images bit for bit.  dst_fwd / dst_dgrad: as
 * afcm_conv2d_pack_weights2 (either may be NULL); `entries` is a HOST array of count <= AFCM_PACK_MAX, copied into the kernel arguments.
 * ---------------------------------------------------------------------------------------- */
#define AFCM_PACK_MAX 32
typedef struct afcm_pack_entry {
    void* dst_fwd;
    void* dst_dgrad;
    const float* w;                               /* [cout][cin][ks][ks] fp32 */
    int32_t cout, cin, rows_pad_fwd, rows_pad_dgrad;
} afcm_pack_entry;
int afcm_conv2d_pack_bank(const afcm_pack_entry* entries, int32_t count, int32_t dtype, int32_t ks, void* stream);

/* ----------------------------------------------------------------------------------------
 * fp32 convolution on the 16-bit matrix pipe by split operands.  The reference's fp32 conv is cuDNN's (SG3OPS/conv2d_gradfix.py:37-58
 * behind NET:25-64); gfx950 multiplies bf16 / f16 16x faster than fp32 (2.5 vs 0.157 PFLOP/s), so an fp32 operand v is written as a
 * sum of 16-bit parts, a = r16(v), b = r16(v - a), c = r16(v - a - b) (exact differences), and the product of two such sums is
 * accumulated term by term in the fp32 MFMA accumulators.
 *   float16, 2 parts, terms (b a') + (a b') + (a a'): 22 significand bits -- with both operands first multiplied by a power of two that
 *     brings their largest magnitude near 2^15 (so that b is a normal number wherever it matters; the callers fold the inverse into
 *     oscale), the result is as accurate as an fp32 dot product of the same length (3e-7 of the output scale at K = 4608).
 *   bfloat16 (no range limit, no scaling): 2 parts / the same 3 terms keep ~16 bits (4e-6 of the output scale), 3 parts / the six
 *     terms of order <= 2 keep all 24 (2e-7).
 *
 * afcm_amax_bits: out[0] = max(out[0], bit pattern of max |scale[plane] * x[plane, :]|) (scale NULL: 1) -- the caller zeroes the word
 *   (or passes one that already holds another tensor's bound to get a joint one).  Non-negative floats order like their bit patterns; a NaN
 *   ranks above inf.  The consumers below turn the word into the power of two g with g * bound in [2^14, 2^15) (g = 1 for a non-finite
 *   bound), on the device: no host round trip between the magnitude pass and the split.
 * afcm_split16: parts[k] (k < nparts in {2, 3}) = part k of g * scale[plane] * x[plane, :] (scale NULL: 1; bound NULL: g = 1, the
 *   bfloat16 case), dtype AFCM_F16 or AFCM_BF16, each a dense tensor of x's shape, part_stride ELEMENTS apart (>= planes * hw, a
 *   multiple of 4).
 * afcm_conv2d_pack_split: the weight image of a split conv -- channel block t of [terms * cin16] holds part (term_wparts >> 4 t) & 15 of
 *   g * w ([cout][cin][3][3] fp32; g from `bound`, the weights' afcm_amax_bits word, or 1), in the layout of afcm_conv2d_pack_weights
 *   (mode 0: forward, rows = cout; mode 1: data gradient, transposed and flipped, rows = cin and cin16 is cout rounded up to 16):
 *   dst [terms * cin16 / 16][9][rows_pad][16].
 * afcm_conv2d_split: y[n][cout][P][Q] fp32 = oscale[n,o] / (g_a g_b) * sum_t conv(W_t, part_{x(t)}) + obias[o], 3x3, stride 1,
 *   0 <= pad <= 2, w even.  x_parts as written by afcm_split16 on the [n][cin][h][w] tensor (it must hold every part a term names); term
 *   t (t < terms <= 8) reads part (term_parts >> 4 t) & 15 and the channel block t of wpacked (afcm_conv2d_pack_split).  bound_a /
 *   bound_b: the two operands' bound words or NULL (g = 1).
 * afcm_unscale: t[i] /= g_a g_b in place (the weight gradient summed from split parts of dy and x: afcm_conv2d_wgrad_ld per term).
 * afcm_plane_dot_parts: out[plane] = <sum_k parts[k][plane, :], b[plane, :]> / g, b fp32 -- afcm_plane_dot for a tensor that only exists as
 *   its split parts (the style gradient <x, dx> when the backward keeps x's parts instead of x).
 * ---------------------------------------------------------------------------------------- */
int afcm_amax_bits(uint32_t* out, const float* x, int64_t planes, int32_t hw, const float* scale, void* stream);
int afcm_split16(void* parts, const float* x, const float* scale, const uint32_t* bound, int32_t dtype, int64_t planes, int32_t hw,
                 int32_t nparts, int64_t part_stride, void* stream);
int afcm_unscale(float* t, int64_t numel, const uint32_t* bound_a, const uint32_t* bound_b, void* stream);
int afcm_plane_dot_parts(float* out, const void* parts, int64_t part_stride, int32_t nparts, const float* b, int32_t dtype, int64_t planes,
                         int32_t hw, const uint32_t* bound, void* stream);
int afcm_conv2d_pack_split(void* dst, const float* w, const uint32_t* bound, int32_t dtype, int32_t cout, int32_t cin, int32_t mode,
                           int32_t rows_pad, int32_t terms, uint32_t term_wparts, void* stream);
int afcm_conv2d_split(float* y, const void* x_parts, const void* wpacked, const float* oscale, const float* obias, int32_t dtype, int32_t n,
                      int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t pad, int32_t rows_pad, int32_t terms, uint32_t term_parts,
                      int64_t part_stride, const uint32_t* bound_a, const uint32_t* bound_b, void* stream);

/* ----------------------------------------------------------------------------------------
 * 3x3 convolution at stride 2 (the discriminator's down-sampling convs, CoModGAN/generator.py:613-692, after the blur): 16-bit x
 * [n][cin][h][w] (dense, w even), weights packed by afcm_conv2d_pack_weights_bk(mode 0, block_k = afcm_conv2d_block_k(dtype)) with rows_pad a multiple of 128, y
 * [n][cout][(h + 2 pad - 3) / 2 + 1][(w + 2 pad - 3) / 2 + 1] = the even rows / columns of afcm_conv2d's result (same 16-bit operands, fp32 accumulation; the two
 * kernels sum their K-chunks in a different order -- 16 channels here, 32 in the stride-1 kernel since r05 -- so results agree within 1 ulp
 * of the 16-bit output, not bit for bit: tests/test_gpu_conv.py holds both to a float64 strided conv), at a quarter of its MFMAs and without the full-resolution intermediate.  Forward only (the gradients are stride-1 convolutions with the
 * zero-stuffed dy: afcm_conv2d / afcm_conv2d_wgrad).
 * ---------------------------------------------------------------------------------------- */
int afcm_conv2d_stride2(void* y, const void* x, const void* wpacked, int32_t dtype, int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t w,
                        int32_t pad, int32_t rows_pad, void* stream);

/* AdaptiveAvgPool2d((4, 4)) of the bottleneck (NET:636,683) for planes that divide evenly (r06): y [planes][4][4] fp32 = block means of x
 * [planes][h][w] (16-bit or fp32; h % 4 == w % 4 == 0, else AFCM_E_NOKERNEL), and its backward dx = gy[block] / block size in x's type. */
int afcm_pool_blocks_fwd(float* y, const void* x, int32_t dtype, int64_t planes, int32_t h, int32_t w, void* stream);
int afcm_pool_blocks_bwd(void* dx, const float* gy, int32_t dtype, int64_t planes, int32_t h, int32_t w, void* stream);

/* The generator's L1 term `criterionL1(fake_B, real_B) * lambda_L1` (models/stylegan3_model.py:107; torch.nn.L1Loss, mean reduction) on fp32 tensors
 * (r06): partials[k] = weight / numel * sum over workgroup k's slice of |a - b| (blocks <= 1024; the caller adds them), and the gradient
 * ga = gout[0] * weight / numel * sign(a - b) with gout a DEVICE scalar (the incoming gradient of the loss). */
int afcm_l1_partials(float* partials, const float* a, const float* b, int64_t numel, int32_t blocks, float weight, void* stream);
int afcm_l1_grad(float* ga, const float* a, const float* b, const float* gout, int64_t numel, float weight, void* stream);

/* y[plane][:] = a[plane][:] + scale[plane] * b[plane][:] (r06), 16-bit tensors of ONE layout, hw elements per plane (hw % 8 == 0, 16-byte aligned
 * bases; else AFCM_E_NOKERNEL), scale [planes] fp32 or NULL (1): the accumulation autograd performs for an encoder feature map that feeds the next
 * encoder layer and a decoder layer's skip input (NET:371-377: `x = x + x_skip`), with the decoder's style factor folded in -- three passes
 * over the planes instead of scale_planes + add's five.  Sum in fp32, one rounding. */
int afcm_axpy_planes(void* y, const void* a, const void* b, const float* scale, int32_t dtype, int64_t planes, int32_t hw, void* stream);

/* ----------------------------------------------------------------------------------------
 * Equalised-learning-rate dense layers as one launch per layer and direction (r06): the mapping network's eight FCs and the
 * bottleneck's `fc_in` -- NET:69-104 (`FullyConnectedLayer.forward`: x @ (w * weight_gain).T + b * bias_gain, then bias_act, which the
 * reference runs as addmm / matmul + plugin `bias_act(x, b, xref, yref, dy, grad, dim, act, alpha, gain, clamp)`, SG3OPS/bias_act.cpp:32-90).
 * All tensors fp32, row-major: x [n][cin], w [cout][cin], b [cout] or NULL, y [n][cout].  act: 0 = linear, 1 = lrelu (slope 0.2, gain
 * sqrt 2: bias_act.py:21-31).  alpha = weight_gain, beta = bias_gain.  n <= 64, cin % 16 == 0 (and cout % 16 == 0 for the backward), else
 * AFCM_E_NOKERNEL (the host keeps the GEMM + bias_act composition).  Exact fp32 arithmetic (v_mfma_f32_16x16x4_f32 = an fmaf chain).
 *   afcm_fc_act_fwd:  y = act(alpha * x w^T + beta * b)
 *   afcm_fc_act_bwd:  with gp = gy * act'(y) (from the saved output y, as bias_act.cu:68-73): dx = alpha * gp w, dw = alpha * gp^T x,
 *                     db = beta * colsum(gp); any of dx / dw / db may be NULL (not wanted).
 * afcm_mapping_input_fwd / _bwd: the mapping network's input stage (NET:143-150): x0 [n][zdim + wdim] = cat(normalize(z),
 * normalize(embed(c))) with normalize(v) = v * rsqrt(mean(v^2) + 1e-8) and embed = the linear FC c_dim -> w_dim (weight ew [wdim][cdim],
 * bias eb, gains alpha / beta); cdim = 0: x0 = normalize(z) only.  The backward returns the embedding layer's weight / bias gradients
 * from gx0 = dL/dx0 (z and c carry none).
 * ---------------------------------------------------------------------------------------- */
int afcm_fc_act_fwd(float* y, const float* x, const float* w, const float* b, int32_t n, int32_t cin, int32_t cout, float alpha, float beta,
                    int32_t act, void* stream);
int afcm_fc_act_bwd(float* dx, float* dw, float* db, const float* gy, const float* y, const float* x, const float* w, int32_t n, int32_t cin,
                    int32_t cout, float alpha, float beta, int32_t act, void* stream);
int afcm_mapping_input_fwd(float* x0, const float* z, const float* c, const float* ew, const float* eb, int32_t n, int32_t zdim, int32_t cdim,
                           int32_t wdim, float alpha, float beta, void* stream);
int64_t afcm_mapping_input_bwd_workspace_bytes(int32_t n, int32_t wdim);
/* workspace: afcm_mapping_input_bwd_workspace_bytes() of device memory whose FIRST WORD IS ZERO before the first launch (the kernel leaves it zero):
 * one row of the intermediate gradient per sample + the ticket by which the last workgroup to finish sums the rows (fixed order: reproducible). */
int afcm_mapping_input_bwd(float* dew, float* deb, const float* gx0, const float* c, const float* ew, const float* eb, int32_t n, int32_t zdim,
                           int32_t cdim, int32_t wdim, float alpha, float beta, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AFCM_HIP_H */
