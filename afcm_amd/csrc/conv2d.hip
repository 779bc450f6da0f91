// Dense KxK convolution for the modulated / encoder convs of the generator (NET:25-64, NET:505) on gfx950 MFMA.
//
// The reference materialises per-sample weights [N,O,I,k,k] and runs a grouped cuDNN conv (NET:46-63).
// Here the mathematically identical factorisation is used (the non-fused branch of CoModGAN/layers.py:56-65):
//     y[n,o] = d[n,o] * conv(W^, s[n,i] * x[n,i])          W^ shared by the whole batch
// so the contraction is one implicit GEMM  D[o][pixel] = sum_{tap,i} W^[tap][o][i] * xs[n][i][pixel+tap]  with
//   A = weights  (MFMA rows  = output channels), pre-packed K-contiguous by conv2d_pack_weights
//   B = activations (MFMA cols = output pixels), NCHW in HBM, transposed to [pixel][channel] while staging
//       into LDS so that every tap is a pure address offset of the same LDS patch (im2col never exists)
//   D = fp32 accumulators in registers; the per-(n,o) scale is applied in the epilogue.
// bf16/f16 use v_mfma_f32_32x32x16_{bf16,f16}; fp32 uses v_mfma_f32_32x32x2_f32 (exact fp32, for the <=1e-3 parity path).
// The same kernel computes the data gradient (weights packed transposed + flipped, pad' = k-1-pad).
#include <stdlib.h>
#include <string.h>

#include <type_traits>

#include "common.h"

#ifndef AFCM_CONV_BM96_PERSIST
#define AFCM_CONV_BM96_PERSIST 1   // the 96-row block as persistent workgroups (two per CU) like the 64-row one: 228 registers, no spills; 3-7 % on the 91-row launches (profiles/r05_conv_bm96_ab.txt)
#endif

namespace afcm {

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(2))) float cf32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 cbf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 cf16x2;

// two fp32 values -> one dword of two 16-bit floats (one v_cvt_pk of exactly this pair, round to nearest even)
template <typename T>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    if constexpr (std::is_same<T, bf16_t>::value) {
        union { cbf16x2 v; unsigned u; } r;
        r.v = __builtin_convertvector((cf32x2){lo, hi}, cbf16x2);
        return r.u;
    } else {
        union { cf16x2 v; unsigned u; } r;
        r.v = __builtin_convertvector((cf32x2){lo, hi}, cf16x2);
        return r.u;
    }
}

template <typename T> struct ConvCfg;
template <> struct ConvCfg<bf16_t> { static constexpr int BK = 16, PITCH = 24; };   // elements; 48-byte rows: conflict-free b128
template <> struct ConvCfg<f16_t>  { static constexpr int BK = 16, PITCH = 24; };
template <> struct ConvCfg<float>  { static constexpr int BK = 8,  PITCH = 9;  };   // 36-byte rows: conflict-free b32

// MFMA shape of the 16-bit 3x3 stride-1 kernel: 1 = v_mfma_f32_16x16x32 (conv2d_fwd16x_kernel, K-chunks of 32 channels), 0 = 32x32x16
// (conv2d_fwd16_kernel, chunks of 16).  -DAFCM_CONV_AB builds both and a debug switch (tools/ab_conv_shape.py): the A/B of r05.
#ifndef AFCM_CONV_X16
#define AFCM_CONV_X16 1
#endif
#ifdef AFCM_CONV_AB
static int g_conv_x16 = AFCM_CONV_X16;
#define AFCM_X16_ON (g_conv_x16 != 0)
#else
#define AFCM_X16_ON (AFCM_CONV_X16 != 0)
#endif
// K-chunk (channels) of the packed weight image for (dtype, kernel size)
static inline int conv_bk(int dtype, int ks);
// conv2d_direct.hip: 16-bit 3x3 convs with at most four input channels (the generator's first layer)
int conv2d_direct_small_cin(const void* x, void* y, const void* wp, const float* oscale, const float* obias, int dtype, int n, int cin, int cout,
                            int h, int w, int pad, int rows_pad, int bk, int ldx, int ldy, hipStream_t st);
#ifndef AFCM_CONV_DIRECT4
#define AFCM_CONV_DIRECT4 1        // 0: the implicit-GEMM kernels for every layer (A/B builds)
#endif

constexpr int kPatchMax = 416;   // LDS patch capacity in pixels
constexpr int kPlaneX16 = 416;    // pixels per channel-group plane of conv2d_fwd16x_kernel (a multiple of 16: planes 256 bytes apart) ...
constexpr int kPatchMaxX16 = kPlaneX16 - 4; // ... of which the patch may use all but the last four (the sink of the staging threads past the plane)
constexpr int kPatchMaxS2 = 704; // ... of the stride-2 kernel (two staging items per thread: <= 1024)
constexpr int kSlots = 256;      // output pixels per workgroup

struct ConvParams {
    const void* x;        // [N, Cin, H, W]
    void* y;              // [N, Cout, P, Q]
    const void* wp;       // packed weights [nkc][KK][Opad][BK]
    const float* oscale;  // [N * Cout] or null
    const float* obias;   // [Cout] or null: y = acc * oscale + obias
    int N, Cin, Cout, H, W, P, Q;
    int ldx, ldy;         // row pitch (elements) of x / y; = W / Q for dense tensors.  conv2d_fwd16_kernel only.
    int pad;
    int TH, TW, PWL, tilesX, tilesY;
    int Opad, nkc;
    int total_blocks;     // conv2d_fwd16x_kernel (persistent workgroups): work items = tiles x images x row blocks; the grid may be smaller
    int o_base;           // conv2d_fwd16x_kernel: first output row of this launch (a layer may be split between the 128- and the 64-row kernel)
    unsigned magicTW;     // ceil(2^32 / TW): j / TW = umulhi(j, magicTW) for the tile-local pixel indices (j < 2^16)
    unsigned magicTX, magicTY, magicN, magicPC;   // ... / tilesX, tilesY, N (block index decode: dividend x divisor < 2^32), / (PWL / 4)
    // split-precision form (conv2d_fwd16_kernel<bf16, BM, true>): x holds `parts` bf16 tensors [N, Cin, H, ldx] part_bytes apart,
    // the K loop runs over terms x nkc_real chunks, term t reads part (term_parts >> 4 t) & 15; y is fp32
    int nkc_real; unsigned magicNK, term_parts; int part_bytes, last_part_bytes;   // last_part_bytes: offset of the highest part any term reads
    const unsigned* bound_a; const unsigned* bound_b;   // magnitude-bound words of the two operands (or null): their power-of-two factors are undone in the epilogue
};
// the power of two g with g * bound in [2^14, 2^15) for a magnitude-bound word (amax_bits_kernel); *inverse = 1 / g
__device__ __forceinline__ float pow2_factor(unsigned bound_bits, float* inverse = nullptr) {
    const float b = __uint_as_float(bound_bits);
    int e = 15;                                              // non-finite bound (a NaN fails the comparison): g = 1
    if (b <= 3.4028234664e38f) frexpf(fmaxf(b, 1e-30f), &e); // b = f * 2^e, f in [0.5, 1)
    if (inverse) *inverse = ldexpf(1.f, e - 15);
    return ldexpf(1.f, 15 - e);
}

__host__ __device__ inline unsigned magic_u32(unsigned d) { return (unsigned)((0x100000000ull + d - 1) / d); }   // 0 for d = 1 (see udiv_magic)
__device__ __forceinline__ unsigned udiv_magic(unsigned n, unsigned magic) { return magic ? __umulhi(n, magic) : n; }

template <typename T, int BM_O, int KS>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 1 : 2)) void conv2d_fwd_kernel(ConvParams p) {
    typedef ConvCfg<T> C;
    constexpr int BK = C::BK, PITCH = C::PITCH, MI = BM_O / 64, KK = KS * KS;
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int EPV = 16 / sizeof(T);              // elements per 16-byte piece
    constexpr int PPR = BK / EPV;                    // pieces per weight row (2)
    constexpr int NPIECES = KK * BM_O * PPR;
    constexpr int NWP = cdiv(NPIECES, 256);
    __shared__ __attribute__((aligned(16))) T lds[(KK * BM_O + kPatchMax) * PITCH];
    T* lds_w = lds;
    T* lds_p = lds + KK * BM_O * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave & 1, wpx = wave >> 1;
    const int r32 = lane & 31, h = lane >> 5;

    int bid = blockIdx.x;
    // integer division runs on the vector pipe even for uniform operands: pin the results to SGPRs, or everything derived
    // from them (image base, buffer descriptor) sits in VGPRs and every buffer load gets a waterfall loop around it
    const int tx = __builtin_amdgcn_readfirstlane(bid % p.tilesX); bid /= p.tilesX;
    const int ty = __builtin_amdgcn_readfirstlane(bid % p.tilesY); bid /= p.tilesY;
    const int n = __builtin_amdgcn_readfirstlane(bid % p.N);
    const int ob = __builtin_amdgcn_readfirstlane(bid / p.N);
    const int y0 = ty * p.TH, x0 = tx * p.TW;
    const int o0 = ob * BM_O;
    const int PH = p.TH + KS - 1, PWL = p.PWL;
    const int xorg = (x0 - p.pad) & ~1;
    const int xoff = (x0 - p.pad) - xorg;

    // fragment bases (element offsets into LDS)
    int bbase[4], pyv[4], pxv[4];
#pragma unroll
    for (int ti = 0; ti < 4; ti++) {
        const int j = wpx * 128 + ti * 32 + r32;
        int py = j / p.TW, px = j - py * p.TW;
        const bool valid = j < p.TH * p.TW;
        if (!valid) { py = 0; px = 0; }
        pyv[ti] = valid ? y0 + py : p.P;             // invalid slots fall outside the image -> never stored
        pxv[ti] = x0 + px;
        bbase[ti] = (py * PWL + px + xoff) * PITCH + h * (F32 ? 1 : 8);
    }
    int abase[MI];
#pragma unroll
    for (int mi = 0; mi < MI; mi++) abase[mi] = (wo * (BM_O / 2) + mi * 32 + r32) * PITCH + h * (F32 ? 1 : 8);

    f32x16 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ti = 0; ti < 4; ti++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ti][e] = 0.f;

    // ---- staging descriptors -------------------------------------------------------------------------------
    // weights: 16-byte pieces, straight copies.  piece j = tid + 256*i -> (tap, o, half-row); 256 is a multiple of the
    // pieces per tap, so both addresses advance by a constant per i.
    constexpr int PPT = BM_O * PPR;                  // pieces per tap (256 or 128)
    constexpr int TPI = 256 / PPT;                   // taps advanced per i
    const int wq = tid % PPT, wtap0 = tid / PPT;
    const int wsrc0 = ((wtap0 * p.Opad) + o0 + wq / PPR) * BK + (wq % PPR) * EPV;
    const int wdst0 = (wtap0 * BM_O + wq / PPR) * PITCH + (wq % PPR) * EPV;
    const int wsrc_step = TPI * p.Opad * BK, wdst_step = TPI * BM_O * PITCH;
    const size_t wchunk = (size_t)KK * p.Opad * BK;
    // patch: one item = 4 pixels x 8 channels
    // lanes 0-31 / 32-63 of a wave take the two channel groups of the same 32 pixel groups: one load instruction then reads
    // two contiguous runs (one per channel) instead of two interleaved streams
    const int cg = F32 ? 0 : (tid >> 5) & 1, pg = F32 ? tid : (tid & 31) + 32 * (tid >> 6);
    const int pcols = PWL >> 2;
    const int prow = pg / pcols, pcol4 = pg - prow * pcols;
    const bool pvalid = prow < PH;
    const int iy = y0 - p.pad + prow, ix = xorg + 4 * pcol4;
    const bool rowok = pvalid && (unsigned)iy < (unsigned)p.H;
    const T* xn = (const T*)p.x + (size_t)n * p.Cin * p.H * p.W;
    const long long pix_off = (long long)(rowok ? iy : 0) * p.W + ix;
    const int pdst = (prow * PWL + 4 * pcol4) * PITCH + cg * 8;

    unsigned wreg[NWP][4];
    unsigned preg[8][F32 ? 4 : 2];
    // 16-bit patch loads are raw buffer loads of 8 bytes (4 pixels of one channel, 4-byte aligned): the row / column
    // validity of a thread never changes, so it is baked into the offset (out of range -> zeros, no branches); a group
    // that straddles the image border keeps its valid half through the and-masks below.
    constexpr unsigned kOob = 0x80000000u;
    const bool d0ok = rowok && (unsigned)ix < (unsigned)p.W, d1ok = rowok && (unsigned)(ix + 2) < (unsigned)p.W;
    const unsigned pmask0 = d0ok ? ~0u : 0u, pmask1 = d1ok ? ~0u : 0u;
    const bool any_partial = __builtin_amdgcn_ballot_w64(d0ok != d1ok) != 0;          // wave-uniform
    // a group whose first half lies left of the image loads from its second half instead (never touch bytes before a row 0)
    const bool lshift = !d0ok && d1ok;
    const bool any_lshift = __builtin_amdgcn_ballot_w64(lshift) != 0;
    const unsigned pvoff = (d0ok || d1ok) ? (unsigned)(((long long)cg * 8 * p.H * p.W + pix_off + (lshift ? 2 : 0)) * (long long)sizeof(T)) : kOob;
    const long long img_bytes = (long long)p.Cin * p.H * p.W * (long long)sizeof(T);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (int)(img_bytes > 0x7fffffffll ? 0x7fffffffll : img_bytes), 0x00020000);
    const int hw2 = p.H * p.W * (int)sizeof(T);

    auto issue_loads = [&](int kc) __attribute__((always_inline)) {
        const T* wsrcp = (const T*)p.wp + (size_t)kc * wchunk;
#pragma unroll
        for (int i = 0; i < NWP; i++)
            if ((NPIECES % 256 == 0) || tid + i * 256 < NPIECES) {
                const uint4 t = *(const uint4*)(wsrcp + wsrc0 + i * wsrc_step);
                wreg[i][0] = t.x; wreg[i][1] = t.y; wreg[i][2] = t.z; wreg[i][3] = t.w;
            }
        if constexpr (F32) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int ch = kc * BK + cg * 8 + c;
                const bool chok = rowok && ch < p.Cin;
                const T* src = xn + ((long long)ch * p.H * p.W + pix_off);
#pragma unroll
                for (int e = 0; e < 4; e++) preg[c][e] = (chok && (unsigned)(ix + e) < (unsigned)p.W) ? *(const unsigned*)(src + e) : 0u;
            }
        } else {
            // channels past Cin only exist in the last chunk: they read whatever follows (or zeros past the image) and are
            // cleared below; everything else needs no per-load work: the chunk's channel offset rides in the scalar offset
            const bool tailchunk = (kc + 1) * BK > p.Cin;
#pragma unroll
            for (int c = 0; c < 8; c++) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, pvoff, (kc * BK + c) * hw2, 0);
                preg[c][0] = v.x; preg[c][1] = v.y;
            }
            if (tailchunk) {
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if (kc * BK + cg * 8 + c >= p.Cin) { preg[c][0] = 0u; preg[c][1] = 0u; }
            }
        }
    };
    auto write_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NWP; i++)
            if ((NPIECES % 256 == 0) || tid + i * 256 < NPIECES) {
                if (F32) {
                    unsigned* d = (unsigned*)(lds_w + wdst0 + i * wdst_step);
                    d[0] = wreg[i][0]; d[1] = wreg[i][1]; d[2] = wreg[i][2]; d[3] = wreg[i][3];
                } else {
                    *(uint4*)(lds_w + wdst0 + i * wdst_step) = make_uint4(wreg[i][0], wreg[i][1], wreg[i][2], wreg[i][3]);
                }
            }
        if (pvalid) {
            if (F32) {
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    unsigned* d = (unsigned*)(lds_p + pdst + e * PITCH);
#pragma unroll
                    for (int c = 0; c < 8; c++) d[c] = preg[c][e];
                }
            } else {
                if (any_lshift) {
#pragma unroll
                    for (int c = 0; c < 8; c++) preg[c][1] = lshift ? preg[c][0] : preg[c][1];
                }
                if (any_partial) {
#pragma unroll
                    for (int c = 0; c < 8; c++) { preg[c][0] &= pmask0; preg[c][1] &= pmask1; }
                }
#pragma unroll
                for (int e = 0; e < 4; e++) {
                    const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                    uint4 v;
                    v.x = __builtin_amdgcn_perm(preg[1][e >> 1], preg[0][e >> 1], sel);
                    v.y = __builtin_amdgcn_perm(preg[3][e >> 1], preg[2][e >> 1], sel);
                    v.z = __builtin_amdgcn_perm(preg[5][e >> 1], preg[4][e >> 1], sel);
                    v.w = __builtin_amdgcn_perm(preg[7][e >> 1], preg[6][e >> 1], sel);
                    *(uint4*)(lds_p + pdst + e * PITCH) = v;
                }
            }
        }
    };

    issue_loads(0);
    write_lds();
    __syncthreads();
    for (int kc = 0; kc < p.nkc; kc++) {
        if (kc + 1 < p.nkc) issue_loads(kc + 1);
#pragma unroll
        for (int r = 0; r < KS; r++)
#pragma unroll
            for (int s = 0; s < KS; s++) {
                const int tap = r * KS + s;
                const int tapoff = (r * PWL + s) * PITCH;
                if constexpr (F32) {
#pragma unroll
                    for (int k2 = 0; k2 < BK / 2; k2++) {
                        float a[MI], b[4];
#pragma unroll
                        for (int mi = 0; mi < MI; mi++) a[mi] = lds_w[tap * BM_O * PITCH + abase[mi] + 2 * k2];
#pragma unroll
                        for (int ti = 0; ti < 4; ti++) b[ti] = lds_p[bbase[ti] + tapoff + 2 * k2];
#pragma unroll
                        for (int mi = 0; mi < MI; mi++)
#pragma unroll
                            for (int ti = 0; ti < 4; ti++)
                                acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ti], acc[mi][ti], 0, 0, 0);
                    }
                } else {
                    typedef typename std::conditional<std::is_same<T, bf16_t>::value, bf16x8, f16x8>::type frag_t;
                    frag_t a[MI], b[4];
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) a[mi] = *(const frag_t*)(lds_w + tap * BM_O * PITCH + abase[mi]);
#pragma unroll
                    for (int ti = 0; ti < 4; ti++) b[ti] = *(const frag_t*)(lds_p + bbase[ti] + tapoff);
#pragma unroll
                    for (int mi = 0; mi < MI; mi++)
#pragma unroll
                        for (int ti = 0; ti < 4; ti++) {
                            if constexpr (std::is_same<T, bf16_t>::value)
                                acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ti], acc[mi][ti], 0, 0, 0);
                            else
                                acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[ti], acc[mi][ti], 0, 0, 0);
                        }
                }
            }
        __syncthreads();
        if (kc + 1 < p.nkc) {
            write_lds();
            __syncthreads();
        }
    }

    // ---- epilogue: D[row = channel][col = pixel]; row = (reg&3) + 8*(reg>>2) + 4*h within the 32x32 tile.
    T* yn = (T*)p.y + (size_t)n * p.Cout * p.P * p.Q;
    // per output-row block: all per-channel scales and biases first (clamped index, no branch around the loads: one wait
    // instead of a round trip per row), then the stores
    int poff[4];                                     // pixel offset inside a plane, -1: not stored
#pragma unroll
    for (int ti = 0; ti < 4; ti++) poff[ti] = (pyv[ti] < p.P && pxv[ti] < p.Q) ? pyv[ti] * p.Q + pxv[ti] : -1;
    const float* osn = p.oscale ? p.oscale + (size_t)n * p.Cout : nullptr;
    const int pq = p.P * p.Q;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        float sc[16], ob[16];
        const int obase = o0 + wo * (BM_O / 2) + mi * 32 + 4 * h;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) { sc[reg] = 1.f; ob[reg] = 0.f; }
        if (osn != nullptr) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) sc[reg] = osn[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
        }
        if (p.obias != nullptr) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) ob[reg] = p.obias[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
        }
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int o = obase + (reg & 3) + 8 * (reg >> 2);
            if (o < p.Cout) {
                T* yo = yn + (size_t)o * pq;
#pragma unroll
                for (int ti = 0; ti < 4; ti++)
                    if (poff[ti] >= 0) yo[poff[ti]] = from_f32<T>(acc[mi][ti][reg] * sc[reg] + ob[reg]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit 3x3 forward / data-gradient kernel, r02 structure.  Same tile (BM_O channels x 256 pixels, 4 waves as 2(o) x
// 2(pixel halves), 2 workgroups per CU), different operand paths:
//   * weights never touch LDS: the packed layout [kc][tap][Opad][16] already IS the A-fragment image (lane (r32, h) ->
//     16 bytes at row r32, half h; a wave reads one contiguous 1 KB), so every wave loads its fragments straight from
//     L2 into a 3-tap-deep register ring, three taps ahead of their MFMAs.  That removes 9 of the 13 staging pieces per
//     thread and K-chunk, the 36 staging VGPRs, a third of the LDS reads and 55 KB of LDS per workgroup;
//   * the activation patch (transposed NCHW -> [pixel][channel] while staging) is double-buffered in the freed LDS:
//     ONE barrier per K-chunk instead of two, and the transposing ds_writes sit among the MFMAs of the running chunk;
//   * the patch writes are conflict-free: a thread owns 4 consecutive pixels (192 bytes apart from its neighbour's, a
//     4-way conflict when every lane writes its pixel e at step e), so lane i writes pixel (e + (i >> 2)) & 3 at step e;
//   * XCD-aware tile order: neighbouring tiles of one image (shared halos, same weights) stay on one XCD's L2.
// Measured bounds (ablation builds, whole-generator conv bench, baseline 0.92 / 0.96 PF/s fwd / dgrad): weight fragments served
// from L1 +2 %; no barrier +0 %; B fragments read once per chunk +5 %; patch written later in the chunk +0 %; the four
// transposing ds_write_b128 removed (loads and permutes kept) +19 %; the whole activation path removed +38 %.  So the
// register -> LDS transpose that NCHW forces is the limiter; LDS-DMA + ds_read_b64_tr_b16 cannot replace it because the
// transposing read ignores the low three address bits (tools/ubench/tr_align_probe.hip) and the tap columns shift by
// 1 and 2 pixels.
#ifdef AFCM_CONV_STAMPS        // diagnostic build only: shader-clock stamps per workgroup (entry, K loop start, K loop end, exit)
__device__ unsigned long long afcm_conv_stamps_buf[4 * 65536];
__device__ unsigned long long afcm_conv_bar_buf[4 * 65536];    // per wave: cycles spent at the K loop's barriers
__device__ unsigned long long afcm_conv_rt_buf[4 * 65536];     // the 100 MHz constant clock at the same four points: shader clock = d cycles / d ticks x 100 MHz
#define AFCM_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 65536) { afcm_conv_stamps_buf[4 * blockIdx.x + (k)] = __builtin_readcyclecounter(); \
                                                                          afcm_conv_rt_buf[4 * blockIdx.x + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
// inside the prologue of conv2d_fwd16x_kernel (wave 0): 0 = requests issued, 1 = all of them returned (an explicit vmcnt(0)), 2 = patch written
__device__ unsigned long long afcm_conv_pro_buf[4 * 65536];
#define AFCM_STAMP_P(k) do { if ((k) == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); \
                             if (threadIdx.x == 0 && blockIdx.x < 65536) afcm_conv_pro_buf[4 * blockIdx.x + (k)] = __builtin_readcyclecounter(); } while (0)
// (the persistent kernel stamps per TILE: slot = the work item; a workgroup's first tile carries the prologue)
#define AFCM_STAMP_I(k, it) do { if (threadIdx.x == 0 && (it) < 65536) { afcm_conv_stamps_buf[4 * (it) + (k)] = __builtin_readcyclecounter(); \
                                                                          afcm_conv_rt_buf[4 * (it) + (k)] = __builtin_amdgcn_s_memrealtime(); } } while (0)
#else
#define AFCM_STAMP(k) do { } while (0)
#define AFCM_STAMP_P(k) do { } while (0)
#define AFCM_STAMP_I(k, it) do { } while (0)
#endif
template <typename T, int BM_O, bool SPLIT = false>
__global__ __launch_bounds__(256, 2) void conv2d_fwd16_kernel(ConvParams p) {
    static_assert(sizeof(T) == 2, "16-bit types only");
    typedef ConvCfg<T> C;
    typedef typename std::conditional<SPLIT, float, T>::type TO;      // output element
    constexpr int KS = 3, KK = 9, BK = C::BK, PITCH = C::PITCH, MI = BM_O / 64;
    // RING: taps the weight fragments run ahead; NP: patch register sets (2 = the patch of chunk k + 2 requested at the top of chunk
    // k); BD: taps the B fragments run ahead.  r04 built and measured the deeper forms on the 64-row blocks (RING 9 at two waves per
    // SIMD or with 9-11 spilled registers at three, NP 2, BD 2, the transposing write at tap 7 / 8): 0-8 % slower on every layer
    // (profiles/r04_conv_ring_depth.txt) -- the chunk does not wait for any ONE of these round trips.
    constexpr int RING = 3, NP = 1, BD = 1;
    typedef typename std::conditional<std::is_same<T, bf16_t>::value, bf16x8, f16x8>::type frag_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    __shared__ __attribute__((aligned(16))) T lds[2 * kPatchMax * PITCH + 4 * PITCH];      // + a sink for lanes outside the patch

    AFCM_STAMP(0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave & 1, wpx = wave >> 1;
    const int r32 = lane & 31, h = lane >> 5;

    int bid = blockIdx.x;
    {
        const int total = gridDim.x;
        bid = xcd_order(bid, total);
    }
    // block index -> (tile x, tile y, image, row block): multiplications by host-made reciprocals on the scalar unit (an integer division,
    // even of uniform operands, is ~30 vector instructions, and everything derived from a VGPR result -- image base, buffer
    // descriptor -- would sit in VGPRs with a waterfall loop around every buffer load)
    unsigned q0 = udiv_magic((unsigned)bid, p.magicTX);
    const int tx = __builtin_amdgcn_readfirstlane(bid - (int)q0 * p.tilesX);
    unsigned q1 = udiv_magic(q0, p.magicTY);
    const int ty = __builtin_amdgcn_readfirstlane((int)q0 - (int)q1 * p.tilesY);
    unsigned q2 = udiv_magic(q1, p.magicN);
    const int n = __builtin_amdgcn_readfirstlane((int)q1 - (int)q2 * p.N);
    const int ob = __builtin_amdgcn_readfirstlane((int)q2);
    const int y0 = ty * p.TH, x0 = tx * p.TW;
    const int o0 = ob * BM_O;
    const int PH = p.TH + KS - 1, PWL = p.PWL;
    const int xorg = (x0 - p.pad) & ~1;
    const int xoff = (x0 - p.pad) - xorg;

    f32x16 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ti = 0; ti < 4; ti++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ti][e] = 0.f;

    // ---- A fragments: straight from the packed weights
    const T* wlane = (const T*)p.wp + (size_t)(o0 + wo * (BM_O / 2) + r32) * BK + h * 8;
    const size_t wtap = (size_t)p.Opad * BK;                       // elements per tap
    auto load_a = [&](int kc, int tap, int mi) __attribute__((always_inline)) {
        return *(const frag_t*)(wlane + ((size_t)kc * KK + tap) * wtap + mi * 32 * BK);
    };

    // ---- patch staging (one item = 4 pixels x 8 channels), as in conv2d_fwd_kernel
    const int cg = (tid >> 5) & 1, pg = (tid & 31) + 32 * (tid >> 6);
    const int pcols = PWL >> 2;
    const int prow = (int)udiv_magic((unsigned)pg, p.magicPC), pcol4 = pg - prow * pcols;
    const bool pvalid = prow < PH;
    const int iy = y0 - p.pad + prow, ix = xorg + 4 * pcol4;
    const bool rowok = pvalid && (unsigned)iy < (unsigned)p.H;
    const T* xn = (const T*)p.x + (size_t)n * p.Cin * p.H * p.ldx;
    const long long pix_off = (long long)(rowok ? iy : 0) * p.ldx + ix;
    const int pdst = (prow * PWL + 4 * pcol4) * PITCH + cg * 8;
    constexpr unsigned kOob = 0x80000000u;
    const bool d0ok = rowok && (unsigned)ix < (unsigned)p.W, d1ok = rowok && (unsigned)(ix + 2) < (unsigned)p.W;
    const unsigned pmask0 = d0ok ? ~0u : 0u, pmask1 = d1ok ? ~0u : 0u;
    const bool lshift = !d0ok && d1ok;                                                // never touch bytes before a row 0
    const unsigned pvoff = (d0ok || d1ok) ? (unsigned)(((long long)cg * 8 * p.H * p.ldx + pix_off + (lshift ? 2 : 0)) * 2ll) : kOob;
    // (split form: one descriptor from this image in part 0 to its end in the highest part read -- the part offset rides in the scalar
    // offset, and the range check covers vector + scalar offset (tools/ubench/buffer_range_probe.hip).  It has to end exactly there: the
    // 8-byte loads of a row's last columns run up to 4 bytes past the row, which for the last row of the last image of the last part is
    // past the allocation; host: that end lies below 2^31 bytes)
    const long long img_bytes = (long long)p.Cin * p.H * p.ldx * 2ll + (SPLIT ? (long long)p.last_part_bytes : 0ll);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (int)(img_bytes > 0x7fffffffll ? 0x7fffffffll : img_bytes), 0x00020000);
    const int hw2 = p.H * p.ldx * 2;

    unsigned preg[NP][8][2];
    // Channels past Cin exist only in the last chunk of a layer whose Cin is not a multiple of 16.  In the plain form the descriptor
    // ends with the image and the range check (vector + scalar offset) zeroes them; in the split form the descriptor runs on into the
    // next part, so those lanes get the out-of-range vector offset.  Decided here, at the chunk boundary, to keep the tap loop free of
    // branches.
    // Branch-free, and issued on EVERY chunk (past the last one with the out-of-range offset: zeros, no memory traffic): a
    // conditional issue makes the compiler's s_waitcnt for the weight ring assume the path without these eight loads, and on
    // the path with them that count waits for all eight -- a full memory round trip exposed at the top of every chunk.
    auto issue_patch = [&](int kc, bool live, auto set_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
        int kcr = kc, sbase = 0;                           // chunk inside its term, byte offset of the term's part (scalar unit)
        if constexpr (SPLIT) {
            const int term = (int)udiv_magic((unsigned)kc, p.magicNK);
            kcr = kc - term * p.nkc_real;
            sbase = (int)((p.term_parts >> (4 * term)) & 15u) * p.part_bytes;
        }
        const int cbase = kcr * BK + cg * 8;
        const int climit = live ? p.Cin : 0;              // one scalar select; `live && ...` per load comes back as branches
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const unsigned off = (cbase + c < climit) ? pvoff : kOob;
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, off, sbase + (kcr * BK + c) * hw2, 0);
            preg[SET][c][0] = v.x; preg[SET][c][1] = v.y;
        }
    };
    // Branch-free on purpose: any branch here (even a wave-uniform one) cuts the tap loop into basic blocks, the ~60
    // transpose instructions then run as one serial block with no MFMA in flight (measured: the register -> LDS transpose
    // cost 38 % of the kernel that way).  Channels past Cin are zeroed by issue_patch; lanes outside the patch
    // write to a sink; edge masks are applied unconditionally.
    auto write_patch = [&](int kc, T* dstbuf, int bufbase, auto set_c) __attribute__((always_inline)) {
        constexpr int SET = decltype(set_c)::value;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const unsigned lo = lshift ? 0u : (preg[SET][c][0] & pmask0);
            const unsigned hi = (lshift ? preg[SET][c][0] : preg[SET][c][1]) & pmask1;
            preg[SET][c][0] = lo; preg[SET][c][1] = hi;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
            uint4 v;
            v.x = __builtin_amdgcn_perm(preg[SET][1][e >> 1], preg[SET][0][e >> 1], sel);
            v.y = __builtin_amdgcn_perm(preg[SET][3][e >> 1], preg[SET][2][e >> 1], sel);
            v.z = __builtin_amdgcn_perm(preg[SET][5][e >> 1], preg[SET][4][e >> 1], sel);
            v.w = __builtin_amdgcn_perm(preg[SET][7][e >> 1], preg[SET][6][e >> 1], sel);
            *(uint4*)(lds + (pvalid ? bufbase + pdst + e * PITCH : 2 * kPatchMax * PITCH)) = v;
        }
    };

    // Ring depth and patch sets.  vmcnt is ONE in-order counter: a wait for a weight fragment also waits for every load issued
    // before it, so a patch request (an HBM round trip) stalls the first tap whose fragments were requested behind it -- with
    // RING = 3 that is three taps after the request, whatever the distance to the transposing writes.  <RING = 9, NP = 2> (the
    // 64-row blocks: 32 accumulator registers leave room): a chunk's fragments are all requested one chunk ahead and the patch of
    // chunk k + 2 is requested at the top of chunk k into the second register set, so the fragments waited for during chunk k
    // are all OLDER than that request and the patch has a whole chunk to arrive.
    frag_t ar[RING][MI];
    typedef std::integral_constant<int, 0> set0_t;
    typedef std::integral_constant<int, NP - 1> set1_t;
    issue_patch(0, true, set0_t{});
#pragma unroll
    for (int t = 0; t < RING; t++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) ar[t][mi] = load_a(0, t, mi);
    if (NP == 2) issue_patch(p.nkc > 1 ? 1 : 0, p.nkc > 1, set1_t{});
    // (tile-local pixel coordinates of this lane's four B fragments: computed under the first requests' round trip)
    int bbase[4], pyv[4], pxv[4];
#pragma unroll
    for (int ti = 0; ti < 4; ti++) {
        const int j = wpx * 128 + ti * 32 + r32;
        int py = (int)__umulhi((unsigned)j, p.magicTW), px = j - py * p.TW;
        const bool valid = j < p.TH * p.TW;
        if (!valid) { py = 0; px = 0; }
        pyv[ti] = valid ? y0 + py : p.P;             // invalid slots fall outside the image -> never stored
        pxv[ti] = x0 + px;
        bbase[ti] = (py * PWL + px + xoff) * PITCH + h * 8;
    }

    write_patch(0, lds, 0, set0_t{});
    __syncthreads();
    AFCM_STAMP(1);

    const int last = p.nkc - 1;
#ifdef AFCM_CONV_STAMPS
    unsigned long long bar_cycles = 0;
#endif
    // one K-chunk; PAR = kc & 1 (NP = 2: the register set that takes the request of chunk kc + 2; the other one holds chunk kc + 1)
    auto chunk = [&](int kc, auto par_c) __attribute__((always_inline)) {
        constexpr int PAR = decltype(par_c)::value;
        typedef std::integral_constant<int, NP == 2 ? PAR : 0> req_t;
        typedef std::integral_constant<int, NP == 2 ? 1 - PAR : 0> wr_t;
        const T* cur = lds + (kc & 1) * (kPatchMax * PITCH);
        T* nxt = lds + ((kc + 1) & 1) * (kPatchMax * PITCH);
        const bool more = kc < last;
        // B fragments run one tap ahead of their MFMAs in the SAME registers: a tap's MFMAs go pixel-block by pixel-block, and
        // as soon as block ti's fragment has been consumed the next tap's fragment for that block is read into it.  (Read, wait,
        // multiply per tap left ~one LDS round trip exposed per 8 MFMAs with only the other workgroup's wave to cover it.)
        frag_t b[BD][4];
#pragma unroll
        for (int d = 0; d < BD; d++)
#pragma unroll
            for (int ti = 0; ti < 4; ti++) b[d][ti] = *(const frag_t*)(cur + bbase[ti] + ((d / KS) * PWL + (d % KS)) * PITCH);
        if (NP == 2) {
            const bool more2 = kc + 2 <= last;
            issue_patch(more2 ? kc + 2 : kc, more2, req_t{});
        } else {
            issue_patch(kc + (int)more, more, req_t{});
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 4 * BD, 0);          // the first fragments first, all in flight together
        __builtin_amdgcn_sched_group_barrier(0x020, 8, 0);
#pragma unroll
        for (int tap = 0; tap < KK; tap++) {
            const int nr = (tap + BD) / KS, ns = (tap + BD) - nr * KS;
            const int tapoff_n = (nr * PWL + ns) * PITCH;                 // offset of the tap BD ahead (unused on the last BD taps)
            frag_t a[MI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) a[mi] = ar[tap % RING][mi];
            // refill this ring slot with the fragments RING taps ahead (clamped at the end: no branch around a load)
            {
                const int nt = (tap + RING) % KK;
                const int nk = (tap + RING < KK) ? kc : (more ? kc + 1 : kc);
#pragma unroll
                for (int mi = 0; mi < MI; mi++) ar[tap % RING][mi] = load_a(nk, nt, mi);
            }
#pragma unroll
            for (int ti = 0; ti < 4; ti++) {
#pragma unroll
                for (int mi = 0; mi < MI; mi++) {
                    if constexpr (std::is_same<T, bf16_t>::value)
                        acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[tap % BD][ti], acc[mi][ti], 0, 0, 0);
                    else
                        acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[tap % BD][ti], acc[mi][ti], 0, 0, 0);
                }
#ifdef AFCM_CONV_EXPERIMENT_HALFB      // timing experiment only (wrong results): every other tap keeps the previous tap's B fragments
                if (tap + BD < KK && (tap & 1) == 0) b[tap % BD][ti] = *(const frag_t*)(cur + bbase[ti] + tapoff_n);
#else
                if (tap + BD < KK) b[tap % BD][ti] = *(const frag_t*)(cur + bbase[ti] + tapoff_n);
#endif
            }
            if (tap == 5) {
                // the other buffer (last read one chunk ago); on the last chunk this rewrites stale registers into a buffer
                // nobody reads.  Interleave: one MFMA, then a handful of the transpose's vector instructions.
                write_patch(kc + 1, nxt, ((kc + 1) & 1) * (kPatchMax * PITCH), wr_t{});
                __builtin_amdgcn_sched_group_barrier(0x020, MI, 0);
#pragma unroll
                for (int ti = 0; ti < 4; ti++) {
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 4, 0);
            } else {
                // pin the issue order of the tap: the ring refill first (left alone, the scheduler sinks the loads next to
                // their uses and the prefetch distance collapses), then per pixel block its MFMAs and the read ahead
                __builtin_amdgcn_sched_group_barrier(0x020, MI, 0);
#pragma unroll
                for (int ti = 0; ti < 4; ti++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, MI, 0);
                    if (tap + BD < KK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
#ifdef AFCM_CONV_STAMPS
        const unsigned long long tb0 = __builtin_readcyclecounter();
        __syncthreads();
        bar_cycles += __builtin_readcyclecounter() - tb0;
#else
        __syncthreads();
#endif
    };
    if constexpr (NP == 2) {
        for (int kc = 0; kc < p.nkc; kc += 2) {
            chunk(kc, std::integral_constant<int, 0>{});
            if (kc + 1 >= p.nkc) break;
            chunk(kc + 1, std::integral_constant<int, 1>{});
        }
    } else {
        for (int kc = 0; kc < p.nkc; kc++) chunk(kc, std::integral_constant<int, 0>{});
    }

    AFCM_STAMP(2);
#ifdef AFCM_CONV_STAMPS
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 65536) afcm_conv_bar_buf[4 * blockIdx.x + (threadIdx.x >> 6)] = bar_cycles;
#endif
    // ---- epilogue: D[row = channel][col = pixel]; row = (reg&3) + 8*(reg>>2) + 4*h within the 32x32 tile.
    if (!SPLIT && (p.TW & 7) == 0) {
        // Tile rows that are multiples of 8 pixels: transpose through LDS (the patch buffers are free after the last barrier)
        // and store 8 pixels = 16 bytes per lane.  A lane holds 16 channels of ONE pixel (4 runs of 4 consecutive channels), so
        // it stages [pixel][32 channels] rows with four 8-byte writes per 32x32 tile, and the transposing read
        // (ds_read_b64_tr_b16: a 16-lane group takes a 4-pixel x 16-channel block, lane i receives channel i of the 4 pixels)
        // hands every lane 4 pixels of one channel.
        // Row = 64 bytes = eight 8-byte chunks; chunk c of pixel p lives at c ^ ((p >> 1) & 7): conflict-free for the writes
        // (16 consecutive pixels x one chunk) and for the reads (a 32-lane half = both channel halves of 4 pixels).
        // r04: this block retired ~650 vector instructions per 32-row pass (PMC: 1324 per wave on a 64 -> 64 layer whose K loop
        // needs 436) -- a min + sign-extend + 64-bit add in front of EVERY per-row scale / bias load (32 of them), the swizzled
        // LDS address of every write, a 64-bit pointer and two predicates per store -- and the SIMD issues those at ~4.6 cycles
        // each whatever the number of waves (tools/ubench/issue_mix.hip): on the 64-row layers the vector issue port, not the matrix
        // pipe, was the limit.  Now everything position-dependent is an immediate offset or one of a few registers computed once:
        // scales / biases come as 16-byte buffer loads (range-checked: rows past Cout read 0), the writes use four per-lane
        // addresses + immediates, the stores are buffer stores whose byte offset is one add of two precomputed registers (an
        // out-of-image granule or an out-of-range channel carries a marker that pushes the sum past the descriptor's range).
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        typedef __attribute__((ext_vector_type(4))) float ef32x4;
        typedef __attribute__((ext_vector_type(4))) unsigned eu32x4;
        constexpr int EROW = 64;
        unsigned char* const ebuf = (unsigned char*)lds + wave * (128 * EROW);
        const int pq = p.P * p.ldy;
        // output image n as one buffer (host: Cout * P * ldy * 2 bytes < 2^30, so the markers below cannot wrap back into range)
        const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.y + (size_t)n * p.Cout * pq), 0, p.Cout * pq * 2, 0x00020000);
        const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.oscale ? p.oscale + (size_t)n * p.Cout : (const float*)p.y), 0, p.oscale ? p.Cout * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.obias ? p.obias : (const float*)p.y), 0, p.obias ? p.Cout * 4 : 0, 0x00020000);
        constexpr unsigned kGOut = 0x80000000u, kOOut = 0xc0000000u;
        // per-row scales and biases of BOTH row passes first (rows of a lane's 16 accumulator registers: rowbase + 4 h + (reg & 3) +
        // 8 (reg >> 2): four 16-byte loads each), so that the address arithmetic below runs under their round trip
        const bool has_sc = p.oscale != nullptr, has_ob = p.obias != nullptr;
        ef32x4 sc[MI][4], ob[MI][4];
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const unsigned sboff = (unsigned)((o0 + wo * (BM_O / 2) + mi * 32 + 4 * h) * 4);
#pragma unroll
            for (int k4 = 0; k4 < 4; k4++) {
                sc[mi][k4] = (ef32x4){1.f, 1.f, 1.f, 1.f};
                ob[mi][k4] = (ef32x4){0.f, 0.f, 0.f, 0.f};
            }
            if (has_sc) {
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) sc[mi][k4] = __builtin_bit_cast(ef32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, sboff + 32u * k4, 0, 0));
            }
            if (has_ob) {
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) ob[mi][k4] = __builtin_bit_cast(ef32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, sboff + 32u * k4, 0, 0));
            }
        }
        // read side: lane = (half hh: granule parity, chalf: channel half, i16: channel / address role inside the 16-lane group)
        const int i16 = lane & 15, chalf = (lane >> 4) & 1, hh = lane >> 5;
        const int q4 = i16 >> 2, p4 = i16 & 3;
        unsigned rd_off[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int prow = 8 * hh + 4 * r + q4;                 // + 16 pixels per iteration: (prow >> 1) & 7 does not change
            rd_off[r] = prow * EROW + (((chalf * 4 + p4) ^ ((prow >> 1) & 7)) << 3);
        }
        // write side: pixel 32 ti + r32 (its swizzle (pix >> 1) & 7 does not depend on ti), chunks h + 2 k4
        unsigned wr_off[4];
#pragma unroll
        for (int k4 = 0; k4 < 4; k4++) wr_off[k4] = (unsigned)(r32 * EROW + (((h + 2 * k4) ^ ((r32 >> 1) & 7)) << 3));
        // this lane's 8 granules (8 pixels each, one tile row): byte offset inside a channel plane, or the marker; bit `it` of gfullm =
        // the whole granule fits the row (a pitched row: up to the pitch, columns >= Q are padding)
        unsigned gbyte[8], gfullm = 0;
        int gxv[8];
#pragma unroll
        for (int it = 0; it < 8; it++) {
            const int j0 = wpx * 128 + (2 * it + hh) * 8;
            const int gpy = (int)__umulhi((unsigned)j0, p.magicTW), gpx = j0 - gpy * p.TW;
            const int gy = y0 + gpy, gx = x0 + gpx;
            gbyte[it] = (j0 < p.TH * p.TW && gy < p.P && gx < p.Q) ? (unsigned)((gy * p.ldy + gx) * 2) : kGOut;
            gxv[it] = gx;
            if (gx + 8 <= p.ldy) gfullm |= 1u << it;
        }
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            const int rowbase = o0 + wo * (BM_O / 2) + mi * 32;
#pragma unroll
            for (int ti = 0; ti < 4; ti++) {
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {                   // registers 4 k4 .. 4 k4 + 3 = channels 4 h + 8 k4 + 0..3
                    uint2 w;
                    w.x = pack2<T>(acc[mi][ti][4 * k4 + 0] * sc[mi][k4][0] + ob[mi][k4][0], acc[mi][ti][4 * k4 + 1] * sc[mi][k4][1] + ob[mi][k4][1]);
                    w.y = pack2<T>(acc[mi][ti][4 * k4 + 2] * sc[mi][k4][2] + ob[mi][k4][2], acc[mi][ti][4 * k4 + 3] * sc[mi][k4][3] + ob[mi][k4][3]);
                    *(uint2*)(ebuf + wr_off[k4] + ti * (32 * EROW)) = w;
                }
            }
            // same wave wrote and reads: LDS operations of a wave complete in order, no barrier needed
            const int o = rowbase + chalf * 16 + i16;
            const unsigned obyte = o < p.Cout ? (unsigned)(o * pq * 2) : kOOut;
#pragma unroll
            for (int it = 0; it < 8; it++) {
                union { s16x4 v[2]; eu32x4 q; } u;
#pragma unroll
                for (int r = 0; r < 2; r++)
                    u.v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ebuf + it * (16 * EROW) + rd_off[r]));
                const unsigned off = obyte + gbyte[it];
                if ((gfullm >> it) & 1) {
                    __builtin_amdgcn_raw_buffer_store_b128(u.q, yrs, off, 0, 0);
                } else {                                          // the granule straddles the right edge (even width: whole pairs)
#pragma unroll
                    for (int w2 = 0; w2 < 4; w2++)
                        __builtin_amdgcn_raw_buffer_store_b32(u.q[w2], yrs, (gxv[it] + 2 * w2 < p.Q) ? off + 4u * w2 : kGOut, 0, 0);
                }
            }
        }
        AFCM_STAMP(3);
        return;
    }
    TO* yn = (TO*)p.y + (size_t)n * p.Cout * p.P * p.ldy;
    // per output-row block: all per-channel scales and biases first (clamped index, no branch around the loads: one wait
    // instead of a round trip per row), then the stores
    int poff[4];                                     // pixel offset inside a plane, -1: not stored
#pragma unroll
    for (int ti = 0; ti < 4; ti++) poff[ti] = (pyv[ti] < p.P && pxv[ti] < p.Q) ? pyv[ti] * p.ldy + pxv[ti] : -1;
    const float* osn = p.oscale ? p.oscale + (size_t)n * p.Cout : nullptr;
    const int pq = p.P * p.ldy;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        float sc[16], ob[16];
        const int obase = o0 + wo * (BM_O / 2) + mi * 32 + 4 * h;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) { sc[reg] = 1.f; ob[reg] = 0.f; }
        if (osn != nullptr) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) sc[reg] = osn[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
        }
        if (p.obias != nullptr) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) ob[reg] = p.obias[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
        }
        if constexpr (SPLIT) {
            float ia = 1.f, ib = 1.f;
            if (p.bound_a) pow2_factor(p.bound_a[0], &ia);
            if (p.bound_b) pow2_factor(p.bound_b[0], &ib);
            const float inv = ia * ib;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) sc[reg] *= inv;
        }
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int o = obase + (reg & 3) + 8 * (reg >> 2);
            if (o < p.Cout) {
                TO* yo = yn + (size_t)o * pq;
#pragma unroll
                for (int ti = 0; ti < 4; ti++)
                    if (poff[ti] >= 0) yo[poff[ti]] = from_f32<TO>(acc[mi][ti][reg] * sc[reg] + ob[reg]);
            }
        }
    }
    AFCM_STAMP(3);
}

// ---- r05: the same tile on v_mfma_f32_16x16x32 -------------------------------------------------------------------------------------
// conv2d_fwd16_kernel's >= 256-channel layers hold ~1.5 GHz under their MFMA load (all-zero operands: +23 %, profiles/r04_power_probe.txt),
// and on a clock-limited loop the chip holds a higher clock on the 16x16x32 shape than on 32x32x16 at equal cycles per flop
// (MI355X_MICROARCH.md, DVFS give-back item 7).  Same workgroup tile (BM_O channels x 256 pixels), same wave tile ((BM_O / 2) x 128:
// MO x 8 accumulator tiles of 16 x 16 = the same accumulator registers), same operand bytes per MFMA cycle; what changes:
//   * a K step is 32 channels of one tap: K-chunks of 32 channels (packed weights [kc][tap][Opad][32], a lane's fragment = 16 bytes at
//     row (lane & 15), channel group (lane >> 4); a wave still reads one contiguous 1 KB per fragment), half as many barriers;
//   * the LDS patch is PLANAR: four planes [channel group of 8][pixel][8 channels], a pixel = 16 bytes, planes a multiple of 256 bytes
//     apart.  A B fragment is 16 consecutive pixels x 4 channel groups; ds_read_b128 serves lanes in four groups of 16 that each hold all
//     16 pixel columns with two of the channel groups, so every group reads 16 consecutive 16-byte slots = all 64 banks once, at any
//     pixel offset (taps shift by 1 and 2 pixels) -- no padding (the [pixel][32 channels] row form conflicts for every odd pitch);
//     a plane holds kPlaneX16 = 416 pixels (patch <= 412 + a 4-pixel sink for the staging threads whose group lies past the patch: no predicate);
//   * B fragments live in a ring of four, read four 16-pixel blocks ahead of their MFMAs (one fragment feeds MO MFMAs = 64 cycles);
//     the tap's column offset is the read's immediate, the row offset is added to the block's base register in place: 32 vector adds
//     per chunk of 72 reads;
//   * weight fragments: buffer loads (lane offset + scalar tap offset: no vector address arithmetic), ring of three taps refilled in
//     place after the tap's last MFMA;
//   * staging: a thread owns 4 pixels x 2 items of 8 channels; one register set, item 0 requested at the top of the chunk and written
//     under tap 4, item 1 requested under tap 5 and written under tap 8.  Lanes 2, 3 (mod 4) of a group write their pixel pairs in
//     swapped order (the permute's selector is a register): 2-way instead of 4-way conflicts on the transposing 16-byte writes;
//   * the issue order is the source order: the loop is written as 72 steps (MO MFMAs, the read four steps ahead, a slice of the
//     staging work) with a scheduling barrier after each.  Left to the scheduler (sched_group_barrier pipelines as in
//     conv2d_fwd16_kernel) the MFMAs of different taps were reordered around the reads and every read was waited for at once.
// FASTEPI (r06): tiles whose width is a multiple of 16 pixels on rows whose pitch is a multiple of 8 elements (the 276^2 / 278^2 and 256^2
// planes of the generator: 8 x 32 and 4 x 64 tiles on 288- and 256-element rows).  A 16-pixel block of the tile then lies in ONE tile row,
// so the row and column base of its two 8-pixel granules are wave-uniform: they are computed on the scalar unit and ride in the store's
// scalar offset (the general form computes eight granule coordinates, validity and straddle flags per lane and tile: ~170 of the ~270
// vector instructions of a 64-row epilogue), a granule never straddles a tile row or the pitched image row (no pair path), and the B
// fragment bases of a tile are one vector add each (set_bbyte: 8 instead of ~64).  Same stores, same bytes.
template <typename T, int BM_O, bool SPLIT = false, bool FASTEPI = false>
__global__ __launch_bounds__(256, 2) void conv2d_fwd16x_kernel(ConvParams p) {
    static_assert(sizeof(T) == 2, "16-bit types only");
    static_assert(!(SPLIT && FASTEPI), "the fp32-output epilogue has no fast form");
    typedef typename std::conditional<SPLIT, float, T>::type TO;      // output element
    // B fragments: a ring read BRING 16-pixel blocks ahead of their MFMAs.  MO = 4: four (a block = 64 MFMA cycles); MO = 2: three -- a block
    // is 32 cycles, but the three waves of a SIMD take turns, and the fourth slot is the register that decides between 168 (three waves
    // per SIMD) and 169
    constexpr int KS = 3, KK = 9, BK = 32, MO = BM_O / 32, NT = 8, BRING = MO == 4 ? 4 : 3;
    // weight-fragment ring, in fragments: a chunk's KK * MO fragments (tap-major) cycle through ARING slots.  MO = 4: 9 slots = 2.25 taps
    // (3 taps = 48 registers do not fit beside 128 accumulator registers at two waves per SIMD); MO = 2: 6 slots = 3 taps
    constexpr int ARING = MO == 2 ? 6 : 9;
    static_assert((KK * MO) % ARING == 0 && ARING >= 2 * MO, "static slots; a tap's fragments and the next tap's are live together");
    // bytes of one channel-group plane: kPlaneX16 = 416 pixels.  2 buffers x 4 planes = 53,248 bytes per workgroup.  The patch itself may
    // use kPatchMaxX16 = 412 pixels: the last four are the sink of the staging threads whose pixel group lies past the plane (branch-free
    // staging writes every thread's four pixels).  Occupancy: two workgroups per CU for both block heights.  The 64-row kernel fits three
    // by registers (<= 168) and LDS (159,744 of 163,840 bytes) on paper; the wave counters show ~1.7 alive (SQ_WAVE_CYCLES x 4 = 58 % of
    // the dispatch), a grid of two per CU runs as fast as one of three (profiles/r05_conv_persistent.txt), and FORCING three
    // (__launch_bounds__(256, 3), 51 KB planes) cost 17 spilled registers and 15-20 % on those layers: the register budget stays at two
    // (__launch_bounds__(256, 2)); the persistent 64-row launch still sizes its grid for three per CU (conv_persistent_grid(blocks, 3)) so
    // that a CU that does fit a third workgroup gets one.
    constexpr int PLANE_B = kPlaneX16 * 16;
    constexpr int BUF_B = 4 * PLANE_B;
    static_assert(PLANE_B % 256 == 0 && kPatchMaxX16 % 4 == 0, "planes: a multiple of 256 bytes apart, patch + sink inside");
    typedef typename std::conditional<std::is_same<T, bf16_t>::value, bf16x8, f16x8>::type frag_t;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    __shared__ __attribute__((aligned(256))) unsigned char lds[2 * BUF_B];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave & 1, wpx = wave >> 1;
    const int c16 = lane & 15, g = lane >> 4;
    const int PH = p.TH + KS - 1, PWL = p.PWL;

    // ---- work items.  The workgroup is PERSISTENT (r05): it takes tiles item, item + gridDim.x, ... (the host launches one round of
    // resident workgroups, a multiple of 8: an item's XCD-aware position, xcd_order(), is then the same function of the item index as it
    // was of the hardware block index) and requests the first K-chunk of its NEXT tile during the last K-chunk of the one under way --
    // the staging slots of that chunk used to issue dead loads -- so that a tile starts on a patch that is already in LDS.  What that
    // hides: a new workgroup needed 7-15k cycles from its first instruction to its first barrier (its address set-up is issued
    // in the slots two MFMA-dense older waves leave, then a memory round trip: profiles/r05_conv_prologue_stamps.txt), a quarter of a
    // workgroup's life on the <= 128-channel layers.
    // Only the 64-row kernel is persistent: carrying a second tile's staging state across the K loop costs the 128-row kernel, which
    // sits at 250 of its 256 registers, 19-52 spilled registers in every form tried (its workgroups live 150-200k cycles, the prologue
    // is 3 % of that); the 64-row kernel -- the <= 64-channel and the 181-channel layers, where the prologue is a quarter -- fits in the
    // 168 registers of three waves per SIMD.  A non-persistent launch has one item per workgroup (the host sizes the grid accordingly).
    constexpr bool PERSIST = BM_O == 64 || (BM_O == 96 && AFCM_CONV_BM96_PERSIST);
    struct Tile { int y0, x0, n, o0; };
    auto decode = [&](int it) __attribute__((always_inline)) -> Tile {
        const int bid = xcd_order(it, p.total_blocks);
        // block index -> (tile x, tile y, image, row block): multiplications by host-made reciprocals, results pinned to SGPRs
        const unsigned q0 = udiv_magic((unsigned)bid, p.magicTX);
        const int tx = __builtin_amdgcn_readfirstlane(bid - (int)q0 * p.tilesX);
        const unsigned q1 = udiv_magic(q0, p.magicTY);
        const int ty = __builtin_amdgcn_readfirstlane((int)q0 - (int)q1 * p.tilesY);
        const unsigned q2 = udiv_magic(q1, p.magicN);
        const int n = __builtin_amdgcn_readfirstlane((int)q1 - (int)q2 * p.N);
        const int ob = __builtin_amdgcn_readfirstlane((int)q2);
        return Tile{ty * p.TH, tx * p.TW, n, p.o_base + ob * BM_O};
    };

    f32x4 acc[MO][NT];

    // ---- A fragments: straight from the packed weights (one buffer: the whole image, < 2^31 bytes -- host)
    const int wtap_b = p.Opad * BK * 2;                                    // bytes per tap
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.wp, 0, p.nkc * KK * wtap_b, 0x00020000);
    // ONE lane offset; the row block (o0), the tap and the fragment ride in the scalar offset (as vector offsets the compiler kept four
    // registers per row block in flight: lane offset + 1 KB per fragment)
    const unsigned wlane = (unsigned)(((wo * (BM_O / 2) + c16) * BK + g * 8) * 2);
    auto load_a = [&](int o0s, int kc, int tap, int mo) __attribute__((always_inline)) {
        return __builtin_bit_cast(frag_t, __builtin_amdgcn_raw_buffer_load_b128(wrs, wlane, (kc * KK + tap) * wtap_b + (o0s + mo * 16) * (BK * 2), 0));
    };

    // ---- patch staging: thread = (channel-group parity cg, 4-pixel group pg); item it covers channel group 2 it + cg.
    // Tile-independent geometry first
    const int cg = (tid >> 5) & 1, pg = (tid & 31) + 32 * (tid >> 6);
    const int pcols = PWL >> 2;
    constexpr unsigned kOob = 0x80000000u;
    const long long img_bytes = (long long)p.Cin * p.H * p.ldx * 2ll + (SPLIT ? (long long)p.last_part_bytes : 0ll);
    const int img_records = (int)(img_bytes > 0x7fffffffll ? 0x7fffffffll : img_bytes);
    const int hw2 = p.H * p.ldx * 2;
    // the patch is dense in pixels (PWL = 4 pcols): this thread's pixels are 4 pg .. 4 pg + 3 -- inside the plane whether or not the
    // patch has that row, except for the groups past the plane's end: those write the sink pixels.  Pixel written at step e: e ^ rot
    const int rot = (pg >> 1) & 1;
    const int pgd = 4 * pg < kPatchMaxX16 ? 4 * pg : kPatchMaxX16;
    // steps 0, 2 (+ 32 bytes at step 2): address pdst_a, selector sel_a; steps 1, 3: the other pixel of the pair = pdst_a ^ 16, the other
    // halves = sel_a ^ 0x02020202 (two registers instead of four across the loop)
    const unsigned pdst_a = (unsigned)(pgd * 16 + cg * PLANE_B + rot * 16);
    const unsigned sel_a = rot ? 0x07060302u : 0x05040100u;
    // ... then the state of the tile whose chunks are being STAGED (the tile under way, or the next one during its last chunk)
    unsigned pvoff = kOob, pm_lo = 0, pm_hi = 0;
    bool lshift = false;
    __amdgpu_buffer_rsrc_t xrs = wrs;
    auto stage_tile = [&](const Tile& t) __attribute__((always_inline)) {
        // (the thread's patch coordinates are recomputed from an opaque copy of its index: kept live across the tile loop they -- and
        // everything else the compiler can hoist out of it -- cost the 128-row kernel 52 spilled registers and the 64-row kernel its
        // third workgroup per CU)
        int tid_o = tid;
        asm volatile("" : "+v"(tid_o));
        const int pg_o = (tid_o & 31) + 32 * (tid_o >> 6), cg_o = (tid_o >> 5) & 1;
        const int prow = (int)udiv_magic((unsigned)pg_o, p.magicPC), pcol4 = pg_o - prow * pcols;
        const bool pvalid = prow < PH;
        const int cg = cg_o;
        const int xorg = (t.x0 - p.pad) & ~1;
        const int iy = t.y0 - p.pad + prow, ix = xorg + 4 * pcol4;
        const bool rowok = pvalid && (unsigned)iy < (unsigned)p.H;
        const long long pix_off = (long long)(rowok ? iy : 0) * p.ldx + ix;
        const bool d0ok = rowok && (unsigned)ix < (unsigned)p.W, d1ok = rowok && (unsigned)(ix + 2) < (unsigned)p.W;
        lshift = !d0ok && d1ok;                                                       // never touch bytes before a row 0
        pm_lo = (d0ok && !lshift) ? ~0u : 0u;
        pm_hi = d1ok ? ~0u : 0u;
        pvoff = (d0ok || d1ok) ? (unsigned)(((long long)cg * 8 * p.H * p.ldx + pix_off + (lshift ? 2 : 0)) * 2ll) : kOob;
        // (the descriptor ends with the image -- split form: with the highest part read -- so channels past Cin read zeros in the plain form)
        xrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const T*)p.x + (size_t)t.n * p.Cin * p.H * p.ldx), 0, img_records, 0x00020000);
    };

    // (branch-free, issued on every chunk -- past the last one of the last tile with the out-of-range offset: see conv2d_fwd16_kernel)
    auto issue_patch = [&](unsigned (&pr)[8][2], int kc, bool live, int item) __attribute__((always_inline)) {
        int kcr = kc, sbase = 0;                           // chunk inside its term, byte offset of the term's part (scalar unit)
        if constexpr (SPLIT) {
            const int term = (int)udiv_magic((unsigned)kc, p.magicNK);
            kcr = kc - term * p.nkc_real;
            sbase = (int)((p.term_parts >> (4 * term)) & 15u) * p.part_bytes;
        }
        const int cbase = kcr * BK + item * 16 + cg * 8;
        const int climit = live ? p.Cin : 0;
        const unsigned voff = live ? pvoff : kOob;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            unsigned off = voff;
            if constexpr (SPLIT) off = (cbase + c < climit) ? pvoff : kOob;     // the descriptor runs on into the next part
            const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, off, sbase + (kcr * BK + item * 16 + c) * hw2, 0);
            pr[c][0] = v.x; pr[c][1] = v.y;
        }
    };
    // edge masks of one channel's two dwords (pixels 0, 1 | 2, 3): unconditional, as in conv2d_fwd16_kernel
    auto mask_ch = [&](unsigned (&pr)[8][2], int c) __attribute__((always_inline)) {
        const unsigned lo = pr[c][0] & pm_lo;
        const unsigned hi = (lshift ? pr[c][0] : pr[c][1]) & pm_hi;
        pr[c][0] = lo; pr[c][1] = hi;
    };
    // step e of the transposing write: 8 channels of one pixel = 16 bytes; sbyte: buffer + item planes (scalar)
    auto write_px = [&](unsigned (&pr)[8][2], int e, int sbyte) __attribute__((always_inline)) {
        const unsigned sel = (e & 1) ? (sel_a ^ 0x02020202u) : sel_a;
        u32x4 v;
        v.x = __builtin_amdgcn_perm(pr[1][e >> 1], pr[0][e >> 1], sel);
        v.y = __builtin_amdgcn_perm(pr[3][e >> 1], pr[2][e >> 1], sel);
        v.z = __builtin_amdgcn_perm(pr[5][e >> 1], pr[4][e >> 1], sel);
        v.w = __builtin_amdgcn_perm(pr[7][e >> 1], pr[6][e >> 1], sel);
        *(u32x4*)(lds + ((e & 1) ? (pdst_a ^ 16u) : pdst_a) + (unsigned)sbyte + (e >> 1) * 32) = v;
    };
    // this lane's eight B fragments: bbyte[ti] = byte address of the fragment of the tap ROW under way in the buffer under way (tile-local
    // pixel 128 wpx + 16 ti + lane & 15, channel group lane >> 4) -- walks down the patch rows and over to the other buffer in place;
    // set at the start of a tile (its xoff, the buffer its first chunk is in)
    unsigned bbyte[NT];
    auto set_bbyte = [&](const Tile& t, int buf) __attribute__((always_inline)) {
        const int xoff = (t.x0 - p.pad) & 1;
        int lane_o = tid;                                    // (opaque: see stage_tile; from tid: one register less across the loop than tid AND lane)
        asm volatile("" : "+v"(lane_o));
        lane_o &= 63;
        if constexpr (FASTEPI) {
            // a 16-pixel block lies in one tile row: its row and first column are scalar
            const unsigned lanepart = (unsigned)((lane_o & 15) * 16 + (lane_o >> 4) * PLANE_B);
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
                const int b16 = wpx * 128 + ti * 16;
                int py = (int)__umulhi((unsigned)b16, p.magicTW), px = b16 - py * p.TW;
                if (b16 >= p.TH * p.TW) { py = 0; px = 0; }
                bbyte[ti] = lanepart + (unsigned)((py * PWL + px + xoff) * 16 + buf * BUF_B);
            }
            return;
        }
#pragma unroll
        for (int ti = 0; ti < NT; ti++) {
            const int j = wpx * 128 + ti * 16 + (lane_o & 15);
            int py = (int)__umulhi((unsigned)j, p.magicTW), px = j - py * p.TW;
            if (j >= p.TH * p.TW) { py = 0; px = 0; }
            bbyte[ti] = (unsigned)((py * PWL + px + xoff) * 16 + (lane_o >> 4) * PLANE_B + buf * BUF_B);
        }
    };

    // ---- the first tile of this workgroup: the one exposed prologue
    int item = blockIdx.x;
    Tile S = decode(item);
    AFCM_STAMP_I(0, item);
    stage_tile(S);
    frag_t ar[ARING];
    unsigned preg[8][2];
    {
        unsigned preg1[8][2];                                // the prologue requests both items at once (the accumulators are not live yet)
        issue_patch(preg, 0, true, 0);
        issue_patch(preg1, 0, true, 1);
#pragma unroll
        for (int q = 0; q < ARING; q++) ar[q] = load_a(S.o0, 0, q / MO, q % MO);
        AFCM_STAMP_P(0);
        AFCM_STAMP_P(1);
#pragma unroll
        for (int c = 0; c < 8; c++) { mask_ch(preg, c); mask_ch(preg1, c); }
#pragma unroll
        for (int e = 0; e < 4; e++) { write_px(preg, e, 0); write_px(preg1, e, 2 * PLANE_B); }
        AFCM_STAMP_P(2);
    }
    int cb = 0;                                              // buffer of the chunk under way
    set_bbyte(S, cb);
    __syncthreads();
    AFCM_STAMP_I(1, item);

    const int last = p.nkc - 1;
    const unsigned rowstep = (unsigned)(PWL * 16);
    for (;;) {
        const int item_n = item + (int)gridDim.x;
        const bool has_next = PERSIST && item_n < p.total_blocks;
#pragma unroll
        for (int mo = 0; mo < MO; mo++)
#pragma unroll
            for (int ti = 0; ti < NT; ti++) acc[mo][ti] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int kc = 0; kc < p.nkc; kc++) {
            const bool fin = kc == last;                    // the tile's last chunk stages chunk 0 of the next tile
            int o0_n = S.o0;                                // row block of the chunk after this one
            if (PERSIST && fin) {
                const Tile N = decode(has_next ? item_n : item);     // (decoded where it is needed: four scalars less across the K loop)
                stage_tile(N);
                o0_n = N.o0;
            }
            const int nxt_b = (cb ^ 1) * BUF_B;
            const bool more = !fin || has_next;
            const int knext = fin ? 0 : kc + 1;
            const unsigned bufstep = (unsigned)((cb ? -BUF_B : BUF_B) - 2 * (int)rowstep);     // to tap row 0 of the other buffer
            frag_t b[BRING];
#pragma unroll
            for (int s = 0; s < BRING; s++) b[s] = *(const frag_t*)(lds + bbyte[s]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int tap = 0; tap < KK; tap++) {
#pragma unroll
                for (int ti = 0; ti < NT; ti++) {
                    const int s = tap * NT + ti;
#pragma unroll
                    for (int mo = 0; mo < MO; mo++) {
                        if constexpr (std::is_same<T, bf16_t>::value)
                            acc[mo][ti] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ar[(tap * MO + mo) % ARING], b[s % BRING], acc[mo][ti], 0, 0, 0);
                        else
                            acc[mo][ti] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ar[(tap * MO + mo) % ARING], b[s % BRING], acc[mo][ti], 0, 0, 0);
                    }
                    const int s2 = s + BRING;
                    if (s2 < KK * NT) {
                        const int t2 = s2 / NT, ti2 = s2 % NT;
                        if (t2 > 0 && t2 % KS == 0) bbyte[ti2] += rowstep;                 // first tap of the next patch row
                        b[s % BRING] = *(const frag_t*)(lds + bbyte[ti2] + (t2 % KS) * 16);
                    }
                    // ---- this step's slice of the staging work
                    if (tap == 0 && ti == 0) issue_patch(preg, knext, more, 0);
                    if (tap == 5 && ti == 0) issue_patch(preg, knext, more, 1);
                    if (tap == 4 || tap == 8) {
                        // item 0 (tap 4) / item 1 (tap 8) of the next chunk into the other buffer: masks under blocks 0-3, a pixel under each of 4-7
                        if (ti < 4) { mask_ch(preg, 2 * ti); mask_ch(preg, 2 * ti + 1); }
                        else write_px(preg, ti - 4, nxt_b + (tap == 8 ? 2 * PLANE_B : 0));
                    }
                    if (tap == 8 && ti >= 4) { bbyte[2 * (ti - 4)] += bufstep; bbyte[2 * (ti - 4) + 1] += bufstep; }   // (the chunk's reads are done)
                    if (ti == NT - 1) {
                        // the tap's ring slots take the fragments ARING ahead now that its MFMAs have read them (past the last tile: the same
                        // fragments again -- no branch around a load)
#pragma unroll
                        for (int mo = 0; mo < MO; mo++) {
                            const int q = tap * MO + mo, q2 = q + ARING;
                            ar[q % ARING] = (q2 < KK * MO) ? load_a(S.o0, kc, q2 / MO, q2 % MO) : load_a(o0_n, knext, (q2 - KK * MO) / MO, q2 % MO);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            cb ^= 1;
            __syncthreads();
        }
        AFCM_STAMP_I(2, item);
        // ---- epilogue of tile S: a 16 x 16 tile has its pixel on the lane (col = lane & 15) and channels 4 g .. 4 g + 3 in the 4 registers
        // (the lane id goes through an empty asm: otherwise pixel coordinates computed for the K loop are kept -- spilled -- for the stores
        // below instead of being recomputed)
        int lane_e = tid;
        asm volatile("" : "+v"(lane_e));
        lane_e &= 63;
        const int c16e = lane_e & 15, ge = lane_e >> 4;
        const int y0 = S.y0, x0 = S.x0, n = S.n, o0 = S.o0;
        if constexpr (!SPLIT) {
            // as conv2d_fwd16_kernel: per 32-channel pass stage [pixel][32 channels] rows (64 bytes, 8-byte chunk c of pixel p at
            // c ^ ((p >> 1) & 7)) and read them back transposed; here a lane stages ONE 8-byte chunk per tile (channels 16 (mo & 1) + 4 g ..),
            // 64 pixels at a time: the staging area (4 KB per wave) lies in the patch buffer the last chunk read -- the other one already
            // holds the next tile's first chunk.
            // For EVERY tile width (even): a granule of 8 tile-local pixels that stays inside one tile row and the image goes out as 16
            // bytes, one that runs over a row end (tile widths that are not multiples of 8: the 5 x 50 tiles of the 150-wide planes, 28, 42)
            // as four pixel pairs with their own coordinates.  (r05: the per-element path below took 86k cycles per workgroup on those
            // tiles -- 16 lanes x 2 bytes per run -- against 10k for this one; it remains for fp32 output.)
            typedef __attribute__((ext_vector_type(4))) short s16x4;
            typedef __attribute__((ext_vector_type(4))) unsigned eu32x4;
            constexpr int EROW = 64;
            unsigned char* const ebuf = lds + (cb ^ 1) * BUF_B + wave * (64 * EROW);
            const int pq = p.P * p.ldy;
            const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((T*)p.y + (size_t)n * p.Cout * pq), 0, p.Cout * pq * 2, 0x00020000);
            const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.oscale ? p.oscale + (size_t)n * p.Cout : (const float*)p.y), 0, p.oscale ? p.Cout * 4 : 0, 0x00020000);
            const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.obias ? p.obias : (const float*)p.y), 0, p.obias ? p.Cout * 4 : 0, 0x00020000);
            constexpr unsigned kGOut = 0x80000000u, kOOut = 0xc0000000u;
            const bool has_sc = p.oscale != nullptr, has_ob = p.obias != nullptr;
            f32x4 sc[MO], ob[MO];
#pragma unroll
            for (int mo = 0; mo < MO; mo++) {
                const unsigned sboff = (unsigned)((o0 + wo * (BM_O / 2) + mo * 16 + 4 * ge) * 4);
                sc[mo] = (f32x4){1.f, 1.f, 1.f, 1.f};
                ob[mo] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (has_sc) sc[mo] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(srs, sboff, 0, 0));
                if (has_ob) ob[mo] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(brs, sboff, 0, 0));
            }
            // read side: lane = (half hh: granule parity, chalf: channel half, i16: channel / address role inside the 16-lane group)
            const int i16 = lane_e & 15, chalf = (lane_e >> 4) & 1, hh = lane_e >> 5;
            const int q4 = i16 >> 2, p4 = i16 & 3;
            unsigned rd_off[2];
#pragma unroll
            for (int r = 0; r < 2; r++) {
                const int prw = 8 * hh + 4 * r + q4;                 // + 16 pixels per iteration: (prw >> 1) & 7 does not change
                rd_off[r] = prw * EROW + (((chalf * 4 + p4) ^ ((prw >> 1) & 7)) << 3);
            }
            // write side: pixel 16 ti + c16 (its swizzle (pix >> 1) & 7 does not depend on ti), chunk 4 (mo & 1) + g
            unsigned wr_off[2];
#pragma unroll
            for (int k = 0; k < 2; k++) wr_off[k] = (unsigned)(c16e * EROW + (((4 * k + ge) ^ ((c16e >> 1) & 7)) << 3));
            // gfullm bit `it`: this lane's granule goes out as 16 bytes (one tile row, inside the (pitched) image row) -- or not at all (a
            // granule outside the tile or the image: its offset carries the marker); slow_any bit `it` (wave-uniform): SOME lane's granule of
            // iteration `it` needs the pair-by-pair path.  That path is behind a SCALAR branch: under a per-lane predicate only, its ~40
            // vector instructions per granule (two quarter-rate multiplies per pixel pair) were issued with an empty EXEC mask on every tile
            // -- ~3k issue cycles per wave and tile, a third of what a tile of a 64-channel layer has to issue at all.
            unsigned gbyte[8], gfullm = 0, slow_any = 0;
#pragma unroll
            for (int it = 0; it < (FASTEPI ? 0 : 8); it++) {
                const int j0 = wpx * 128 + (2 * it + hh) * 8;
                const int gpy = (int)__umulhi((unsigned)j0, p.magicTW), gpx = j0 - gpy * p.TW;
                const int gy = y0 + gpy, gx = x0 + gpx;
                const bool in_tile = j0 < p.TH * p.TW && gy < p.P;
                const bool valid = in_tile && gx < p.Q;
                const bool straddle = gpx + 8 > p.TW;                    // runs on into the next tile row (whose pixels may be inside the image when these are not)
                const bool slow = in_tile && (straddle || (valid && gx + 8 > p.ldy));
                gbyte[it] = valid ? (unsigned)((gy * p.ldy + gx) * 2) : kGOut;
                if (!slow) gfullm |= 1u << it;
                if (__builtin_amdgcn_ballot_w64(slow) != 0) slow_any |= 1u << it;
            }
#pragma unroll
            for (int mi = 0; mi < (MO + 1) / 2; mi++) {
                const int rowbase = o0 + wo * (BM_O / 2) + mi * 32;
                const int o = rowbase + chalf * 16 + i16;
                // (MO odd -- the 96-row block: the last pass carries 16 channels, its upper half belongs to the other wave's rows)
                const unsigned obyte = (o < p.Cout && mi * 32 + chalf * 16 < BM_O / 2) ? (unsigned)(o * pq * 2) : kOOut;
#pragma unroll
                for (int half = 0; half < 2; half++) {
#pragma unroll
                    for (int k = 0; k < 2; k++) {
                        const int mo = 2 * mi + k;
                        if (mo >= MO) continue;
#pragma unroll
                        for (int t4 = 0; t4 < 4; t4++) {
                            const int ti = 4 * half + t4;
                            uint2 w;
                            w.x = pack2<T>(acc[mo][ti][0] * sc[mo][0] + ob[mo][0], acc[mo][ti][1] * sc[mo][1] + ob[mo][1]);
                            w.y = pack2<T>(acc[mo][ti][2] * sc[mo][2] + ob[mo][2], acc[mo][ti][3] * sc[mo][3] + ob[mo][3]);
                            *(uint2*)(ebuf + wr_off[k] + t4 * (16 * EROW)) = w;
                        }
                    }
                    // same wave wrote and reads: LDS operations of a wave complete in order, no barrier needed
#pragma unroll
                    for (int i4 = 0; i4 < 4; i4++) {
                        const int it = 4 * half + i4;
                        union { s16x4 v[2]; eu32x4 q; } u;
#pragma unroll
                        for (int r = 0; r < 2; r++)
                            u.v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ebuf + i4 * (16 * EROW) + rd_off[r]));
                        if constexpr (FASTEPI) {
                            // the block's row and first column on the scalar unit; lane half hh takes the second granule (+ 16 bytes)
                            const int b16 = wpx * 128 + 16 * it;
                            const int gpy = (int)__umulhi((unsigned)b16, p.magicTW), gpxs = b16 - gpy * p.TW;
                            const int gy = y0 + gpy, gxs = x0 + gpxs;
                            if (b16 < p.TH * p.TW && gy < p.P && gxs < p.Q) {          // wave-uniform
                                unsigned vo = obyte + 16u * (unsigned)hh;
                                if (gxs + 8 >= p.Q && hh) vo = kGOut;                    // (uniform test first: the image's last granule pair only)
                                __builtin_amdgcn_raw_buffer_store_b128(u.q, yrs, vo, (gy * p.ldy + gxs) * 2, 0);
                            }
                            continue;
                        }
                        const bool full = (gfullm >> it) & 1;
                        // (no branch around the common store: a lane on the pair path sends its 16 bytes out of range)
                        __builtin_amdgcn_raw_buffer_store_b128(u.q, yrs, full ? obyte + gbyte[it] : kGOut, 0, 0);
                        if ((slow_any >> it) & 1) {                       // wave-uniform
                            if (!full) {                                  // the granule runs over the tile row's or the image's right end (even widths: whole pairs)
#pragma unroll
                                for (int w2 = 0; w2 < 4; w2++) {
                                    const int j = wpx * 128 + (2 * it + hh) * 8 + 2 * w2;
                                    const int py = (int)__umulhi((unsigned)j, p.magicTW), px = j - py * p.TW;
                                    const bool ok = j < p.TH * p.TW && y0 + py < p.P && x0 + px < p.Q;
                                    __builtin_amdgcn_raw_buffer_store_b32(u.q[w2], yrs, ok ? obyte + (unsigned)(((y0 + py) * p.ldy + x0 + px) * 2) : kGOut, 0, 0);
                                }
                            }
                        }
                    }
                }
            }
        } else {
            TO* yn = (TO*)p.y + (size_t)n * p.Cout * p.P * p.ldy;
            int poff[NT];                                    // pixel offset inside a plane, -1: not stored
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
                const int j = wpx * 128 + ti * 16 + c16e;
                const int py = (int)__umulhi((unsigned)j, p.magicTW), px = j - py * p.TW;
                poff[ti] = (j < p.TH * p.TW && y0 + py < p.P && x0 + px < p.Q) ? (y0 + py) * p.ldy + x0 + px : -1;
            }
            const float* osn = p.oscale ? p.oscale + (size_t)n * p.Cout : nullptr;
            const int pq = p.P * p.ldy;
            float ia = 1.f, ib = 1.f;
            if (p.bound_a) pow2_factor(p.bound_a[0], &ia);
            if (p.bound_b) pow2_factor(p.bound_b[0], &ib);
            const float inv = ia * ib;
#pragma unroll
            for (int mo = 0; mo < MO; mo++) {
                float sc[4], ob[4];
                const int obase = o0 + wo * (BM_O / 2) + mo * 16 + 4 * ge;
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    sc[reg] = (osn != nullptr ? osn[min(obase + reg, p.Cout - 1)] : 1.f) * inv;
                    ob[reg] = p.obias != nullptr ? p.obias[min(obase + reg, p.Cout - 1)] : 0.f;
                }
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    const int o = obase + reg;
                    if (o < p.Cout) {
                        TO* yo = yn + (size_t)o * pq;
#pragma unroll
                        for (int ti = 0; ti < NT; ti++)
                            if (poff[ti] >= 0) yo[poff[ti]] = from_f32<TO>(acc[mo][ti][reg] * sc[reg] + ob[reg]);
                    }
                }
            }
        }
        AFCM_STAMP_I(3, item);
        if (!has_next) break;
        // the staging area of the epilogue (the buffer the last chunk read) is the one the next tile's first chunk stages INTO
        if (!SPLIT) __syncthreads();
        S = decode(item_n);
        item = item_n;
        set_bbyte(S, cb);
        AFCM_STAMP_I(0, item);
        AFCM_STAMP_I(1, item);
    }
}

// Stride-2 form of conv2d_fwd16_kernel for the discriminator's down-sampling convs (CoModGAN/generator.py:613-692: blur, then a 3x3
// conv at stride 2): the r01/r02 route computed the stride-1 result and decimated it -- four times the MFMAs, a full-resolution
// write and a decimation copy.  Here an output pixel (py, px) reads the patch at (2 py + r, 2 px + s): same packed weights, same
// tap loop, B fragment addresses twice as far apart.  The patch of a tile is ~4x its outputs, so a workgroup takes 128 output
// pixels (two 32-pixel blocks per wave) under a (2 TH + 1) x (2 TW + 2) patch of up to kPatchMaxS2 pixels (two staging items per
// thread), 68 KB of LDS double-buffered: still two workgroups per CU.  Bit-identical to the even pixels of the stride-1 result
// (same K order).  Forward only: the gradients of a strided conv are convolutions with the zero-stuffed dy and keep the stride-1 kernels.
template <typename T, int BM_O>
__global__ __launch_bounds__(256, 2) void conv2d_fwd16s2_kernel(ConvParams p) {
    static_assert(sizeof(T) == 2, "16-bit types only");
    typedef ConvCfg<T> C;
    constexpr int KS = 3, KK = 9, BK = C::BK, PITCH = C::PITCH, MI = BM_O / 64, RING = 3;
    constexpr int NT = 2, PXW = 32 * NT, STRIDE = 2, NITEM = 2, PMAX = kPatchMaxS2;   // 128 output pixels per workgroup, two staging items per thread
    typedef typename std::conditional<std::is_same<T, bf16_t>::value, bf16x8, f16x8>::type frag_t;
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    __shared__ __attribute__((aligned(16))) T lds[2 * PMAX * PITCH + 4 * PITCH];      // + a sink for lanes outside the patch

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave & 1, wpx = wave >> 1;
    const int r32 = lane & 31, h = lane >> 5;

    int bid = blockIdx.x;
    {
        const int total = gridDim.x;
        bid = xcd_order(bid, total);
    }
    // integer division runs on the vector pipe even for uniform operands: pin the results to SGPRs, or everything derived
    // from them (image base, buffer descriptor) sits in VGPRs and every buffer load gets a waterfall loop around it
    const int tx = __builtin_amdgcn_readfirstlane(bid % p.tilesX); bid /= p.tilesX;
    const int ty = __builtin_amdgcn_readfirstlane(bid % p.tilesY); bid /= p.tilesY;
    const int n = __builtin_amdgcn_readfirstlane(bid % p.N);
    const int ob = __builtin_amdgcn_readfirstlane(bid / p.N);
    const int y0 = ty * p.TH, x0 = tx * p.TW;
    const int o0 = ob * BM_O;
    const int PH = (p.TH - 1) * STRIDE + KS, PWL = p.PWL;
    const int xorg = (x0 * STRIDE - p.pad) & ~1;
    const int xoff = (x0 * STRIDE - p.pad) - xorg;

    int bbase[NT], pyv[NT], pxv[NT];
#pragma unroll
    for (int ti = 0; ti < NT; ti++) {
        const int j = wpx * PXW + ti * 32 + r32;
        int py = j / p.TW, px = j - py * p.TW;
        const bool valid = j < p.TH * p.TW;
        if (!valid) { py = 0; px = 0; }
        pyv[ti] = valid ? y0 + py : p.P;             // invalid slots fall outside the image -> never stored
        pxv[ti] = x0 + px;
        bbase[ti] = (py * STRIDE * PWL + px * STRIDE + xoff) * PITCH + h * 8;
    }

    f32x16 acc[MI][NT];
#pragma unroll
    for (int mi = 0; mi < MI; mi++)
#pragma unroll
        for (int ti = 0; ti < NT; ti++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[mi][ti][e] = 0.f;

    // ---- A fragments: straight from the packed weights
    const T* wlane = (const T*)p.wp + (size_t)(o0 + wo * (BM_O / 2) + r32) * BK + h * 8;
    const size_t wtap = (size_t)p.Opad * BK;                       // elements per tap
    auto load_a = [&](int kc, int tap, int mi) __attribute__((always_inline)) {
        return *(const frag_t*)(wlane + ((size_t)kc * KK + tap) * wtap + mi * 32 * BK);
    };

    // ---- patch staging (one item = 4 pixels x 8 channels), as in conv2d_fwd_kernel
    const int cg = (tid >> 5) & 1, pg = (tid & 31) + 32 * (tid >> 6);
    const int pcols = PWL >> 2;
    const T* xn = (const T*)p.x + (size_t)n * p.Cin * p.H * p.ldx;
    constexpr unsigned kOob = 0x80000000u;
    const long long img_bytes = (long long)p.Cin * p.H * p.ldx * 2ll;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)xn, 0, (int)(img_bytes > 0x7fffffffll ? 0x7fffffffll : img_bytes), 0x00020000);
    const int hw2 = p.H * p.ldx * 2;
    bool pvalid[NITEM], lshift[NITEM];
    int pdst[NITEM];
    unsigned pmask0[NITEM], pmask1[NITEM], pvoff[NITEM];
#pragma unroll
    for (int it = 0; it < NITEM; it++) {
        const int item = pg + 128 * it;
        const int prow = item / pcols, pcol4 = item - prow * pcols;
        pvalid[it] = prow < PH;
        const int iy = y0 * STRIDE - p.pad + prow, ix = xorg + 4 * pcol4;
        const bool rowok = pvalid[it] && (unsigned)iy < (unsigned)p.H;
        const long long pix_off = (long long)(rowok ? iy : 0) * p.ldx + ix;
        pdst[it] = (prow * PWL + 4 * pcol4) * PITCH + cg * 8;
        const bool d0ok = rowok && (unsigned)ix < (unsigned)p.W, d1ok = rowok && (unsigned)(ix + 2) < (unsigned)p.W;
        pmask0[it] = d0ok ? ~0u : 0u; pmask1[it] = d1ok ? ~0u : 0u;
        lshift[it] = !d0ok && d1ok;                                                   // never touch bytes before a row 0
        pvoff[it] = (d0ok || d1ok) ? (unsigned)(((long long)cg * 8 * p.H * p.ldx + pix_off + (lshift[it] ? 2 : 0)) * 2ll) : kOob;
    }

    unsigned preg[NITEM][8][2];
    auto issue_patch = [&](int kc, bool live) __attribute__((always_inline)) {
        const int cbase = kc * BK + cg * 8;
        const int climit = live ? p.Cin : 0;
#pragma unroll
        for (int it = 0; it < NITEM; it++)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const unsigned off = (cbase + c < climit) ? pvoff[it] : kOob;
                const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(xrs, off, (kc * BK + c) * hw2, 0);
                preg[it][c][0] = v.x; preg[it][c][1] = v.y;
            }
    };
    auto write_patch = [&](int kc, T* dstbuf, int bufbase) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NITEM; it++) {
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const unsigned lo = lshift[it] ? 0u : (preg[it][c][0] & pmask0[it]);
                const unsigned hi = (lshift[it] ? preg[it][c][0] : preg[it][c][1]) & pmask1[it];
                preg[it][c][0] = lo; preg[it][c][1] = hi;
            }
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const unsigned sel = (e & 1) ? 0x07060302u : 0x05040100u;
                uint4 v;
                v.x = __builtin_amdgcn_perm(preg[it][1][e >> 1], preg[it][0][e >> 1], sel);
                v.y = __builtin_amdgcn_perm(preg[it][3][e >> 1], preg[it][2][e >> 1], sel);
                v.z = __builtin_amdgcn_perm(preg[it][5][e >> 1], preg[it][4][e >> 1], sel);
                v.w = __builtin_amdgcn_perm(preg[it][7][e >> 1], preg[it][6][e >> 1], sel);
                *(uint4*)(lds + (pvalid[it] ? bufbase + pdst[it] + e * PITCH : 2 * PMAX * PITCH)) = v;
            }
        }
    };

    frag_t ar[RING][MI];
    issue_patch(0, true);
#pragma unroll
    for (int t = 0; t < RING; t++)
#pragma unroll
        for (int mi = 0; mi < MI; mi++) ar[t][mi] = load_a(0, t, mi);
    write_patch(0, lds, 0);
    __syncthreads();

    const int last = p.nkc - 1;
    for (int kc = 0; kc < p.nkc; kc++) {
        const T* cur = lds + (kc & 1) * (PMAX * PITCH);
        T* nxt = lds + ((kc + 1) & 1) * (PMAX * PITCH);
        const bool more = kc < last;
        // B fragments run one tap ahead of their MFMAs in the SAME registers: a tap's MFMAs go pixel-block by pixel-block, and
        // as soon as block ti's fragment has been consumed the next tap's fragment for that block is read into it.  (Read, wait,
        // multiply per tap left ~one LDS round trip exposed per 8 MFMAs with only the other workgroup's wave to cover it.)
        frag_t b[NT];
#pragma unroll
        for (int ti = 0; ti < NT; ti++) b[ti] = *(const frag_t*)(cur + bbase[ti]);
        issue_patch(kc + (int)more, more);
        __builtin_amdgcn_sched_group_barrier(0x100, NT, 0);              // tap 0's fragments first, all in flight together
        __builtin_amdgcn_sched_group_barrier(0x020, 8 * NITEM, 0);
#pragma unroll
        for (int tap = 0; tap < KK; tap++) {
            const int nr = (tap + 1) / KS, ns = (tap + 1) - nr * KS;
            const int tapoff_n = (nr * PWL + ns) * PITCH;                 // next tap's offset (unused on the last tap)
            frag_t a[MI];
#pragma unroll
            for (int mi = 0; mi < MI; mi++) a[mi] = ar[tap % RING][mi];
            // refill this ring slot with the fragments three taps ahead (clamped at the end: no branch around a load)
            {
                const int nt = (tap + RING) % KK;
                const int nk = (tap + RING < KK) ? kc : (more ? kc + 1 : kc);
#pragma unroll
                for (int mi = 0; mi < MI; mi++) ar[tap % RING][mi] = load_a(nk, nt, mi);
            }
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
#pragma unroll
                for (int mi = 0; mi < MI; mi++) {
                    if constexpr (std::is_same<T, bf16_t>::value)
                        acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ti], acc[mi][ti], 0, 0, 0);
                    else
                        acc[mi][ti] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[mi], b[ti], acc[mi][ti], 0, 0, 0);
                }
                if (tap + 1 < KK) b[ti] = *(const frag_t*)(cur + bbase[ti] + tapoff_n);
            }
            if (tap == 5) {
                // the other buffer (last read one chunk ago); on the last chunk this rewrites stale registers into a buffer
                // nobody reads.  Interleave: one MFMA, then a handful of the transpose's vector instructions.
                write_patch(kc + 1, nxt, ((kc + 1) & 1) * (PMAX * PITCH));
                __builtin_amdgcn_sched_group_barrier(0x020, MI, 0);
#pragma unroll
                for (int ti = 0; ti < NT; ti++) {
#pragma unroll
                    for (int mi = 0; mi < MI; mi++) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 20, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x200, 4 * NITEM, 0);
            } else {
                // pin the issue order of the tap: the ring refill first (left alone, the scheduler sinks the loads next to
                // their uses and the three-tap prefetch distance collapses), then per pixel block its MFMAs and the read ahead
                __builtin_amdgcn_sched_group_barrier(0x020, MI, 0);
#pragma unroll
                for (int ti = 0; ti < NT; ti++) {
                    __builtin_amdgcn_sched_group_barrier(0x008, MI, 0);
                    if (tap + 1 < KK) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
        }
        __syncthreads();
    }

    // ---- epilogue: D[row = channel][col = pixel]; row = (reg&3) + 8*(reg>>2) + 4*h within the 32x32 tile.
    if ((p.TW & 7) == 0 && (p.Q & 1) == 0) {        // (an odd output width -- possible at stride 2 -- puts rows on odd elements: element stores below)
        // Tile rows that are multiples of 8 pixels: transpose through LDS (the patch buffers are free after the last barrier)
        // and store 8 pixels = 16 bytes per lane.  A lane holds 16 channels of ONE pixel (4 runs of 4 consecutive channels), so
        // it stages [pixel][32 channels] rows with four 8-byte writes per 32x32 tile, and the transposing read
        // (ds_read_b64_tr_b16: a 16-lane group takes a 4-pixel x 16-channel block, lane i receives channel i of the 4 pixels)
        // hands every lane 4 pixels of one channel.  Per thread and 32-channel pass: 32 packed conversions + 16 ds_write_b64 +
        // 16 transposing reads + 8 stores (the first version staged [channel][pixel] with 64 two-byte writes per pass: the
        // epilogue was 12 % of the whole conv time, 35 % on the 64-channel layers).
        // Row = 64 bytes = eight 8-byte chunks; chunk c of pixel p lives at c ^ ((p >> 1) & 7): conflict-free for the writes
        // (16 consecutive pixels x one chunk) and for the reads (a 32-lane half = both channel halves of 4 pixels).
        typedef __attribute__((ext_vector_type(4))) short s16x4;
        constexpr int EROW = 64;
        unsigned char* ebuf = (unsigned char*)lds + wave * (PXW * EROW);
        T* yn = (T*)p.y + (size_t)n * p.Cout * p.P * p.ldy;
        const float* osn = p.oscale ? p.oscale + (size_t)n * p.Cout : nullptr;
        const int pq = p.P * p.ldy;
        // read side: lane = (half hh: granule parity, chalf: channel half, i16: channel / address role inside the 16-lane group)
        const int i16 = lane & 15, chalf = (lane >> 4) & 1, hh = lane >> 5;
        const int q4 = i16 >> 2, p4 = i16 & 3;
        unsigned rd_off[2];
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const int prow = 8 * hh + 4 * r + q4;                 // + 16 pixels per iteration: (prow >> 1) & 7 does not change
            rd_off[r] = prow * EROW + (((chalf * 4 + p4) ^ ((prow >> 1) & 7)) << 3);
        }
        // this lane's 8 granules (8 pixels each, one tile row): plane offset, -1 = outside the image; bit it of gfullm = whole
        int goff[2 * NT];
        unsigned gfullm = 0;
        int gxv[2 * NT];
#pragma unroll
        for (int it = 0; it < 2 * NT; it++) {
            const int j0 = wpx * PXW + (2 * it + hh) * 8;
            const int gpy = (int)__umulhi((unsigned)j0, p.magicTW), gpx = j0 - gpy * p.TW;
            const int gy = y0 + gpy, gx = x0 + gpx;
            goff[it] = (j0 < p.TH * p.TW && gy < p.P && gx < p.Q) ? gy * p.ldy + gx : -1;
            gxv[it] = gx;
            if (gx + 8 <= p.ldy) gfullm |= 1u << it;         // a pitched row has room for the whole granule (columns >= Q: padding)
        }
        const int wr_pix = r32;                                    // + 32 ti
#pragma unroll
        for (int mi = 0; mi < MI; mi++) {
            float sc[16], ob[16];
            const int obase = o0 + wo * (BM_O / 2) + mi * 32 + 4 * h;
#pragma unroll
            for (int reg = 0; reg < 16; reg++) { sc[reg] = 1.f; ob[reg] = 0.f; }
            if (osn != nullptr) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++) sc[reg] = osn[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
            }
            if (p.obias != nullptr) {
#pragma unroll
                for (int reg = 0; reg < 16; reg++) ob[reg] = p.obias[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
            }
#pragma unroll
            for (int ti = 0; ti < NT; ti++) {
                const int pix = ti * 32 + wr_pix;
                const int sw = (pix >> 1) & 7;
#pragma unroll
                for (int k4 = 0; k4 < 4; k4++) {                   // registers 4 k4 .. 4 k4 + 3 = channels 4 h + 8 k4 + 0..3
                    uint2 w;
                    w.x = pack2<T>(acc[mi][ti][4 * k4 + 0] * sc[4 * k4 + 0] + ob[4 * k4 + 0], acc[mi][ti][4 * k4 + 1] * sc[4 * k4 + 1] + ob[4 * k4 + 1]);
                    w.y = pack2<T>(acc[mi][ti][4 * k4 + 2] * sc[4 * k4 + 2] + ob[4 * k4 + 2], acc[mi][ti][4 * k4 + 3] * sc[4 * k4 + 3] + ob[4 * k4 + 3]);
                    *(uint2*)(ebuf + pix * EROW + (((h + 2 * k4) ^ sw) << 3)) = w;
                }
            }
            // same wave wrote and reads: LDS operations of a wave complete in order, no barrier needed
            const int o = o0 + wo * (BM_O / 2) + mi * 32 + chalf * 16 + i16;
            T* const yo = yn + (size_t)min(o, p.Cout - 1) * pq;
#pragma unroll
            for (int it = 0; it < 2 * NT; it++) {
                union { s16x4 v[2]; uint4 q; } u;
#pragma unroll
                for (int r = 0; r < 2; r++)
                    u.v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(ebuf + it * (16 * EROW) + rd_off[r]));
                if (goff[it] >= 0 && o < p.Cout) {
                    T* dst = yo + goff[it];
                    if ((gfullm >> it) & 1) {
                        *(uint4*)dst = u.q;
                    } else {                                      // the granule straddles the right edge (even width: whole pairs)
                        const unsigned vv[4] = {u.q.x, u.q.y, u.q.z, u.q.w};
#pragma unroll
                        for (int w2 = 0; w2 < 4; w2++)
                            if (gxv[it] + 2 * w2 < p.Q) ((unsigned*)dst)[w2] = vv[w2];
                    }
                }
            }
        }
        return;
    }
    T* yn = (T*)p.y + (size_t)n * p.Cout * p.P * p.ldy;
    // per output-row block: all per-channel scales and biases first (clamped index, no branch around the loads: one wait
    // instead of a round trip per row), then the stores
    int poff[NT];                                    // pixel offset inside a plane, -1: not stored
#pragma unroll
    for (int ti = 0; ti < NT; ti++) poff[ti] = (pyv[ti] < p.P && pxv[ti] < p.Q) ? pyv[ti] * p.ldy + pxv[ti] : -1;
    const float* osn = p.oscale ? p.oscale + (size_t)n * p.Cout : nullptr;
    const int pq = p.P * p.ldy;
#pragma unroll
    for (int mi = 0; mi < MI; mi++) {
        float sc[16], ob[16];
        const int obase = o0 + wo * (BM_O / 2) + mi * 32 + 4 * h;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) { sc[reg] = 1.f; ob[reg] = 0.f; }
        if (osn != nullptr) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) sc[reg] = osn[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
        }
        if (p.obias != nullptr) {
#pragma unroll
            for (int reg = 0; reg < 16; reg++) ob[reg] = p.obias[min(obase + (reg & 3) + 8 * (reg >> 2), p.Cout - 1)];
        }
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int o = obase + (reg & 3) + 8 * (reg >> 2);
            if (o < p.Cout) {
                T* yo = yn + (size_t)o * pq;
#pragma unroll
                for (int ti = 0; ti < NT; ti++)
                    if (poff[ti] >= 0) yo[poff[ti]] = from_f32<T>(acc[mi][ti][reg] * sc[reg] + ob[reg]);
            }
        }
    }
}


// ---------------------------------------------------------------------------------------------
// Weight packing: w[O][I][KS][KS] (fp32) -> [nkc][KK][Opad][BK] of T, zero padded.
//   mode 0 (forward):        dst[kc][r*KS+s][o][kk]  = w[o][kc*BK+kk][r][s]
//   mode 1 (data gradient):  roles of O and I swap and taps flip:
//                            dst[kc][r*KS+s][i][kk]  = w[kc*BK+kk][i][KS-1-r][KS-1-s]
template <typename T>
__global__ __launch_bounds__(256) void conv2d_pack_kernel(T* __restrict__ dst, const float* __restrict__ w, int O, int I, int KS,
                                                          int rows, int cols, int rows_pad, int BK, int nkc, int mode) {
    const int KK = KS * KS;
    const long long total = (long long)nkc * KK * rows_pad * BK;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(idx % BK);
        long long t = idx / BK;
        const int row = (int)(t % rows_pad); t /= rows_pad;
        const int tap = (int)(t % KK);
        const int kc = (int)(t / KK);
        const int col = kc * BK + kk;
        float v = 0.f;
        if (row < rows && col < cols) {
            const int r = tap / KS, s = tap - r * KS;
            if (mode == 0) v = w[(((size_t)row * I + col) * KS + r) * KS + s];
            else v = w[(((size_t)col * I + row) * KS + (KS - 1 - r)) * KS + (KS - 1 - s)];
        }
        dst[idx] = from_f32<T>(v);
    }
}

// Same layout from an LDS tile: a workgroup stages w[o0 .. o0+16)[i0 .. i0+64)[all taps] with coalesced loads (every o is one
// contiguous run of 64 * k*k floats) and emits BOTH images from it as 8-element (16-byte for 16-bit types) stores --
//   forward       dst0[kc = i / BK][tap][row = o][i % BK]            16 rows x BK contiguous per (kc, tap)
//   data gradient dst1[kc = o / BK][k*k-1-tap][row = i][o % BK]      64 rows x BK contiguous per (kc, tap)
// -- so the weights are read once for the two images, the index arithmetic is per 8 elements, and the forward and the backward
// image of a layer come out of ONE launch (a null destination skips that image).  The per-element gather kernel above is kept
// as the definition the layout test checks against.
// tile of the pack: TO output x TI input channels, both at least one K-chunk (the data-gradient image chunks the OUTPUT channels)
template <int BK> struct PackTile { static constexpr int TO = BK > 16 ? 32 : 16, TI = BK > 16 ? 32 : 64; };
template <typename T, int KK, int BK>
__device__ __forceinline__ void pack_tile_body(float* tile, int bx, int by, T* __restrict__ dst0, T* __restrict__ dst1, const float* __restrict__ w,
                                               int O, int I, int rows_pad0, int rows_pad1) {
    constexpr int TO = PackTile<BK>::TO, TI = PackTile<BK>::TI, ROW = TI * KK + 1;             // + 1: the 8 channel runs of a store start 9 floats apart
    struct alignas(8 * sizeof(T)) Out { T v[8]; };
    const int i0 = bx * TI, o0 = by * TO;
    {
        // all of a thread's loads in flight before the first LDS write (left as a loop, each load waited for its predecessor:
        // 36 serial round trips = 10 us for any layer size)
        constexpr int NL = TO * TI * KK / 256;
        static_assert(TO * TI * KK % 256 == 0, "tile size");
        float v[NL];
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int e = threadIdx.x + 256 * k;
            const int o = e / (TI * KK), r = e - o * (TI * KK);
            const int i = r / KK;
            v[k] = (o0 + o < O && i0 + i < I) ? w[((size_t)(o0 + o) * I + i0) * KK + r] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < NL; k++) {
            const int e = threadIdx.x + 256 * k;
            const int o = e / (TI * KK), r = e - o * (TI * KK);
            tile[o * ROW + r] = v[k];
        }
    }
    __syncthreads();
    constexpr int G = BK / 8;
    if (dst0 != nullptr && o0 < rows_pad0) {
        // items: (kc_local, tap, o, half); cols (i) are written up to the last started K-chunk only
        const int nkc = cdiv(I, BK);
        constexpr int kcl = TI / BK, NIT = kcl * KK * TO * G;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += 256) {
            const int it = it0 + threadIdx.x;
            if (NIT % 256 != 0 && it >= NIT) break;
            const int half = it % G;
            int t = it / G;
            const int o = t % TO; t /= TO;
            const int tap = t % KK, kc = t / KK;
            const int kcg = i0 / BK + kc;
            if (kcg >= nkc) continue;
            Out v;
#pragma unroll
            for (int c = 0; c < 8; c++) v.v[c] = from_f32<T>(tile[o * ROW + (kc * BK + half * 8 + c) * KK + tap]);
            *(Out*)(dst0 + (((size_t)kcg * KK + tap) * rows_pad0 + o0 + o) * BK + half * 8) = v;
        }
    }
    if (dst1 != nullptr && i0 < rows_pad1) {
        const int nkc = cdiv(O, BK);
        constexpr int kcl = TO / BK, NIT = kcl * KK * TI * G;
#pragma unroll
        for (int it0 = 0; it0 < NIT; it0 += 256) {
            const int it = it0 + threadIdx.x;
            if (NIT % 256 != 0 && it >= NIT) break;
            const int half = it % G;
            int t = it / G;
            const int i = t % TI; t /= TI;
            const int tap = t % KK, kc = t / KK;
            const int kcg = o0 / BK + kc;
            if (kcg >= nkc) continue;
            Out v;
#pragma unroll
            for (int c = 0; c < 8; c++) v.v[c] = from_f32<T>(tile[(kc * BK + half * 8 + c) * ROW + i * KK + (KK - 1 - tap)]);
            *(Out*)(dst1 + (((size_t)kcg * KK + tap) * rows_pad1 + i0 + i) * BK + half * 8) = v;
        }
    }
}

template <typename T, int KK, int BK>
__global__ __launch_bounds__(256) void conv2d_pack_tile_kernel(T* __restrict__ dst0, T* __restrict__ dst1, const float* __restrict__ w, int O,
                                                               int I, int rows_pad0, int rows_pad1) {
    __shared__ float tile[PackTile<BK>::TO * (PackTile<BK>::TI * KK + 1)];
    pack_tile_body<T, KK, BK>(tile, blockIdx.x, blockIdx.y, dst0, dst1, w, O, I, rows_pad0, rows_pad1);
}

// the same for a list of layers in one launch (C ABI afcm_conv2d_pack_bank): the table rides in the kernel arguments, a workgroup
// finds its layer by a scalar scan over the first-block table
struct PackBank {
    int count;
    int blk[AFCM_PACK_MAX + 1];
    int gx[AFCM_PACK_MAX];
    afcm_pack_entry e[AFCM_PACK_MAX];
};
template <typename T, int KK, int BK>
__global__ __launch_bounds__(256) void conv2d_pack_bank_kernel(const PackBank b) {
    __shared__ float tile[PackTile<BK>::TO * (PackTile<BK>::TI * KK + 1)];
    int l = 0;
    while (l + 1 < b.count && (int)blockIdx.x >= b.blk[l + 1]) l++;
    const int loc = blockIdx.x - b.blk[l];
    const afcm_pack_entry& e = b.e[l];
    pack_tile_body<T, KK, BK>(tile, loc % b.gx[l], loc / b.gx[l], (T*)e.dst_fwd, (T*)e.dst_dgrad, e.w, e.cout, e.cin, e.rows_pad_fwd, e.rows_pad_dgrad);
}

// 4 consecutive elements as one vector access (8 B for 16-bit types, 16 B for fp32); planes are 4-element aligned when hw % 4 == 0
template <typename T> struct Vec4 { T v[4]; } __attribute__((aligned(sizeof(T) * 4)));

// y[plane, :] = x[plane, :] * scale[plane]  (dtype conversion fused).  HBM-bound elementwise pass.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void scale_planes_kernel(TO* __restrict__ y, const TI* __restrict__ x, const float* __restrict__ scale,
                                                           long long planes, int hw) {
    const int per = (hw + 3) >> 2;
    const long long total = planes * per;
    const bool vec = (hw & 3) == 0;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long plane = idx / per;
        const int i0 = (int)(idx - plane * per) << 2;
        const float sc = scale ? scale[plane] : 1.f;
        const TI* xp = x + plane * hw + i0;
        TO* yp = y + plane * hw + i0;
        if (vec) {
            const Vec4<TI> in = *(const Vec4<TI>*)xp;
            Vec4<TO> out;
#pragma unroll
            for (int e = 0; e < 4; e++) out.v[e] = from_f32<TO>(to_f32(in.v[e]) * sc);
            *(Vec4<TO>*)yp = out;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (i0 + e < hw) yp[e] = from_f32<TO>(to_f32(xp[e]) * sc);
        }
    }
}

// y[plane, :] = a[plane, :] + scale[plane] * b[plane, :] for 16-bit tensors of one layout (r06): the gradient of an encoder feature map that
// feeds BOTH the next encoder layer (a = that layer's data gradient) and a decoder layer's skip input (b = the decoder layer's incoming
// gradient, scale = the styles its epilogue multiplied the sum by).  The op-by-op form was scale_planes (read b, write s b) + autograd's
// accumulation (read both, write the sum): five passes over a 156 MB plane set where this is three, one rounding instead of two.
// 16-byte vectors; hw % 8 == 0 (host).
template <typename T>
__global__ __launch_bounds__(256) void axpy_planes_kernel(T* __restrict__ y, const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ scale,
                                                          long long planes, int hw) {
    union V16 { uint4 u; T v[8]; };
    const int per = hw >> 3;                                   // 16-byte vectors per plane
    for (long long plane = blockIdx.y; plane < planes; plane += gridDim.y) {
        const float sc = scale ? scale[plane] : 1.f;
        const uint4* ap = (const uint4*)a + plane * per;
        const uint4* bp = (const uint4*)b + plane * per;
        uint4* yp = (uint4*)y + plane * per;
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < per; i += gridDim.x * blockDim.x) {
            V16 av, bv, out;
            av.u = ap[i];
            bv.u = bp[i];
#pragma unroll
            for (int e = 0; e < 8; e++) out.v[e] = from_f32<T>(to_f32(av.v[e]) + sc * to_f32(bv.v[e]));
            yp[i] = out.u;
        }
    }
}

// Magnitude bound of a tensor, for the float16 split (split16_kernel): out[0] = max(out[0], bits of max |scale[plane] * x[plane, :]|).
// Magnitudes compare as their bit patterns; r06: over the FINITE elements only (an inf / NaN element keeps its place in the split's first part
// whatever the factor).  The consumers turn the bound into the power of two g with
// g * bound in [2^14, 2^15) (pow2_factor): float16 parts of g * v cannot overflow (65504) and the second part of every element above
// 2^-18 of the bound is a normal number (below that it is a subnormal: 22 significand bits shrink to 11 at 2^-29 of the bound).  One bound
// per TENSOR, not per plane: the contraction sums over the input planes, so a per-plane factor cannot leave the sum (a per-sample one could
// in forward / data gradient, not in the weight gradient, which sums over samples) -- DESIGN.md section 4.4.
// One atomic per workgroup, and only from workgroups that would raise the value (2048 unconditional atomics on one word cost 50 us).
__global__ __launch_bounds__(256) void amax_bits_kernel(unsigned* __restrict__ out, const float* __restrict__ x, long long planes, int hw,
                                                        const float* __restrict__ scale) {
    __shared__ unsigned red[4];
    unsigned m = 0;
    const long long numel = planes * hw;
    const bool vec = (((uintptr_t)x & 15) == 0) && (hw & 3) == 0;
    const long long n4 = vec ? (numel >> 2) : 0;
    const int per = hw >> 2;                                 // 16-byte groups per plane (vec only)
    const long long stride = (long long)gridDim.x * blockDim.x;
    long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    auto fold = [&](const uint4 v, long long g) {
        // non-finite magnitudes (exponent all ones) do not count: one inf or NaN would otherwise take the power-of-two factor -- and with it
        // 22-bit parts -- away from every finite element of the tensor; the split keeps such an element whole in its first part
        const unsigned ax = v.x & 0x7fffffffu, ay = v.y & 0x7fffffffu, az = v.z & 0x7fffffffu, aw = v.w & 0x7fffffffu;
        unsigned a = max(max(ax < 0x7f800000u ? ax : 0u, ay < 0x7f800000u ? ay : 0u), max(az < 0x7f800000u ? az : 0u, aw < 0x7f800000u ? aw : 0u));
        if (scale) {
            a = __float_as_uint(__uint_as_float(a) * __builtin_fabsf(scale[g / per]));              // |s| max|x| = max|s x|
            if (a >= 0x7f800000u) a = 0u;                                                           // (an overflowing or non-finite factor: as above)
        }
        m = max(m, a);
    };
    // four independent 16-byte loads in flight per lane
    for (; i + 3 * stride < n4; i += 4 * stride) {
        const uint4 v0 = ((const uint4*)x)[i], v1 = ((const uint4*)x)[i + stride], v2 = ((const uint4*)x)[i + 2 * stride], v3 = ((const uint4*)x)[i + 3 * stride];
        fold(v0, i); fold(v1, i + stride); fold(v2, i + 2 * stride); fold(v3, i + 3 * stride);
    }
    for (; i < n4; i += stride) fold(((const uint4*)x)[i], i);
    for (long long j = 4 * n4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; j < numel; j += stride) {
        const float v = x[j] * (scale ? scale[j / hw] : 1.f);
        const unsigned a = __float_as_uint(v) & 0x7fffffffu;
        m = max(m, a < 0x7f800000u ? a : 0u);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(red[0], red[1]), max(red[2], red[3]));
        if (m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
    }
}

// out[plane] = <sum_k parts[k][plane, :], b[plane, :]> / g: the style gradient's dot product <x, dx> of an fp32 conv whose backward kept the
// 16-bit parts of x (split16_kernel) instead of x.  One workgroup per plane; 4 elements per lane and iteration where the plane allows.
template <typename TP>
__global__ __launch_bounds__(256) void plane_dot_parts_kernel(float* __restrict__ out, const TP* __restrict__ parts, long long part_stride, int nparts,
                                                              const float* __restrict__ b, long long planes, int hw, const unsigned* __restrict__ bound) {
    __shared__ float red[4];
    const long long plane = blockIdx.x;
    if (plane >= planes) return;
    const TP* ap = parts + plane * hw;
    const float* bp = b + plane * hw;
    float acc = 0.f;
    if ((hw & 3) == 0 && (part_stride & 3) == 0) {          // 8-byte part loads, 16-byte b loads (the generator's planes: hw % 4 == 0)
        union V8 { uint2 u; TP v[4]; };
        for (int i = threadIdx.x * 4; i < hw; i += blockDim.x * 4) {
            const float4 b0 = *(const float4*)(bp + i);
            const float bv[4] = {b0.x, b0.y, b0.z, b0.w};
            for (int k = 0; k < nparts; k++) {
                V8 a; a.u = *(const uint2*)(ap + k * part_stride + i);
#pragma unroll
                for (int e = 0; e < 4; e++) acc += (float)a.v[e] * bv[e];
            }
        }
    } else {
        for (int i = threadIdx.x; i < hw; i += blockDim.x) {
            float a = 0.f;
            for (int k = 0; k < nparts; k++) a += (float)ap[k * part_stride + i];
            acc += a * bp[i];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float inv = 1.f;
        if (bound) pow2_factor(bound[0], &inv);
        out[plane] = (red[0] + red[1] + red[2] + red[3]) * inv;
    }
}

// t *= 1 / (g_a g_b): the power-of-two factors of two split operands undone (weight gradient of split parts).
__global__ __launch_bounds__(256) void unscale_kernel(float* __restrict__ t, long long numel, const unsigned* __restrict__ bound_a, const unsigned* __restrict__ bound_b) {
    float ia = 1.f, ib = 1.f;
    if (bound_a) pow2_factor(bound_a[0], &ia);
    if (bound_b) pow2_factor(bound_b[0], &ib);
    const float inv = ia * ib;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < numel; i += (long long)gridDim.x * blockDim.x) t[i] *= inv;
}

// The packed image of a split conv's stacked weight parts (afcm_conv2d_split): channel block t (cin16 = 16 nkc_real channels) holds part
// (term_wparts >> 4 t) & 15 of g * w (g: pow2_factor of the bound word, 1 without one) -- part 0 = r16(v), part 1 = r16(v - part 0), ... -- in the layout of conv2d_pack_kernel
// (mode 1: the data gradient's transposed, flipped kernel).
template <typename T>
__global__ __launch_bounds__(256) void conv2d_pack_split_kernel(T* __restrict__ dst, const float* __restrict__ w, const unsigned* __restrict__ bound,
                                                                int O, int I, int rows, int cols, int rows_pad, int nkc_real, int terms,
                                                                unsigned term_wparts, int mode, int BK) {
    constexpr int KS = 3, KK = 9;
    const float gs = bound ? pow2_factor(bound[0]) : 1.f;
    const long long total = (long long)terms * nkc_real * KK * rows_pad * BK;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int kk = (int)(idx % BK);
        long long t = idx / BK;
        const int row = (int)(t % rows_pad); t /= rows_pad;
        const int tap = (int)(t % KK);
        const int kc = (int)(t / KK);
        const int term = kc / nkc_real;
        const int col = (kc - term * nkc_real) * BK + kk;
        float r = 0.f;
        if (row < rows && col < cols) {
            const int rr = tap / KS, ss = tap - rr * KS;
            if (mode == 0) r = w[(((size_t)row * I + col) * KS + rr) * KS + ss];
            else r = w[(((size_t)col * I + row) * KS + (KS - 1 - rr)) * KS + (KS - 1 - ss)];
        }
        r *= gs;
        const int part = (int)((term_wparts >> (4 * term)) & 15u);
        T q = (T)r;
        for (int k = 0; k < part; k++) {
            const float qf = (float)q;
            r = (__builtin_fabsf(qf) <= 3.4028234664e38f) ? r - qf : 0.f;
            q = (T)r;
        }
        dst[idx] = q;
    }
}

// Split-precision operands: v = g * scale[plane] * x (g: pow2_factor of the bound word, 1 without one) as a sum of `parts` 16-bit numbers, v ~ a + b (+ c) with a = r16(v), b = r16(v - a),
// c = r16(v - a - b) (round to nearest even; the differences are exact in fp32).  bfloat16: two parts carry 16 significand bits, three
// carry all 24.  float16: two parts carry 22 bits wherever b is a normal number, i.e. for |v| >= 2^-3; below that the error is at most
// 2^-25 ABSOLUTE, so with g the power of two that brings the tensor's largest magnitude near 2^15 it is 2^-40 of that magnitude.
// parts[k] is a dense tensor of the input's shape, part_stride elements after parts[k - 1].  A non-finite v keeps its class in part a
// and zeros in the others (inf - inf would make NaNs of infinities).
template <typename TP, int PARTS>
__global__ __launch_bounds__(256) void split16_kernel(TP* __restrict__ parts, const float* __restrict__ x, const float* __restrict__ scale,
                                                      const unsigned* __restrict__ bound, long long planes, int hw, long long part_stride) {
    const int per = (hw + 3) >> 2;
    const long long total = planes * per;
    const bool vec = (hw & 3) == 0;
    const float gs = bound ? pow2_factor(bound[0]) : 1.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const long long plane = idx / per;
        const int i0 = (int)(idx - plane * per) << 2;
        const float sc = (scale ? scale[plane] : 1.f) * gs;
        const float* xp = x + plane * hw + i0;
        TP* yp = parts + plane * hw + i0;
        float v[4];
        if (vec) {
            const Vec4<float> in = *(const Vec4<float>*)xp;
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = in.v[e] * sc;
        } else {
#pragma unroll
            for (int e = 0; e < 4; e++) v[e] = (i0 + e < hw) ? xp[e] * sc : 0.f;
        }
        Vec4<TP> out[PARTS];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            float r = v[e];
#pragma unroll
            for (int k = 0; k < PARTS; k++) {
                const TP q = (TP)r;
                out[k].v[e] = q;
                const float qf = (float)q;
                r = (__builtin_fabsf(qf) <= 3.4028234664e38f) ? r - qf : 0.f;      // inf / nan: nothing left for the lower parts
            }
        }
#pragma unroll
        for (int k = 0; k < PARTS; k++) {
            if (vec) {
                *(Vec4<TP>*)(yp + k * part_stride) = out[k];
            } else {
#pragma unroll
                for (int e = 0; e < 4; e++)
                    if (i0 + e < hw) yp[k * part_stride + e] = out[k].v[e];
            }
        }
    }
}

// Per-plane reductions: out[plane] = sum a*b (or sum a when b == null).  One workgroup per plane, fp32 accumulate, 16-byte
// loads from the plane's first 16-byte boundary on (both operands share the plane offset; the host checks the base pointers).
template <typename T>
__global__ __launch_bounds__(256) void plane_dot_kernel(float* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                                        long long planes, int hw) {
    constexpr int E = 16 / (int)sizeof(T);
    union V16 { uint4 u; T v[E]; };
    __shared__ float part[4];
    const long long plane = blockIdx.x;
    if (plane >= planes) return;
    const long long off = plane * hw;
    const T* ap = a + off;
    const T* bp = b ? b + off : nullptr;
    int head = (int)((E - (off % E)) % E);
    if (head > hw) head = hw;
    const int nv = (hw - head) / E;
    float s0 = 0.f, s1 = 0.f;
    {   // ragged ends: fewer than 2E elements in total
        const int tail0 = head + nv * E;
        int i = -1;
        if ((int)threadIdx.x < head) i = threadIdx.x;
        else if ((int)threadIdx.x - head < hw - tail0) i = tail0 + (int)threadIdx.x - head;
        if (i >= 0) s0 = to_f32(ap[i]) * (bp ? to_f32(bp[i]) : 1.f);
    }
    const uint4* av = (const uint4*)(ap + head);
    const uint4* bv = bp ? (const uint4*)(bp + head) : nullptr;
    int i = threadIdx.x;
    if (bv) {
        // four 16-byte loads per operand in flight (128 B per lane) before the first use
        for (; i + 768 < nv; i += 1024) {
            V16 a0, a1, a2, a3, b0, b1, b2, b3;
            a0.u = av[i]; a1.u = av[i + 256]; a2.u = av[i + 512]; a3.u = av[i + 768];
            b0.u = bv[i]; b1.u = bv[i + 256]; b2.u = bv[i + 512]; b3.u = bv[i + 768];
#pragma unroll
            for (int e = 0; e < E; e++) {
                s0 = fmaf(to_f32(a0.v[e]), to_f32(b0.v[e]), s0); s1 = fmaf(to_f32(a1.v[e]), to_f32(b1.v[e]), s1);
                s0 = fmaf(to_f32(a2.v[e]), to_f32(b2.v[e]), s0); s1 = fmaf(to_f32(a3.v[e]), to_f32(b3.v[e]), s1);
            }
        }
    }
    for (; i + 256 < nv; i += 512) {
        V16 a0, a1, b0, b1;
        a0.u = av[i]; a1.u = av[i + 256];
        if (bv) {
            b0.u = bv[i]; b1.u = bv[i + 256];
#pragma unroll
            for (int e = 0; e < E; e++) { s0 = fmaf(to_f32(a0.v[e]), to_f32(b0.v[e]), s0); s1 = fmaf(to_f32(a1.v[e]), to_f32(b1.v[e]), s1); }
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) { s0 += to_f32(a0.v[e]); s1 += to_f32(a1.v[e]); }
        }
    }
    if (i < nv) {
        V16 a0, b0;
        a0.u = av[i];
        if (bv) {
            b0.u = bv[i];
#pragma unroll
            for (int e = 0; e < E; e++) s0 = fmaf(to_f32(a0.v[e]), to_f32(b0.v[e]), s0);
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) s0 += to_f32(a0.v[e]);
        }
    }
    float s = s0 + s1;
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) s += __shfl_down(s, off2, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) out[plane] = part[0] + part[1] + part[2] + part[3];
}

// Small planes (<= 16 KB): one WAVE per plane, four planes per workgroup -- a 36^2 plane is 180 16-byte pieces, so a
// 256-thread workgroup per plane leaves most lanes without a load and the launch is bound by workgroup turnover.
template <typename T>
__global__ __launch_bounds__(256) void plane_dot_wave_kernel(float* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                                             long long planes, int hw) {
    constexpr int E = 16 / (int)sizeof(T);
    union V16 { uint4 u; T v[E]; };
    const int lane = threadIdx.x & 63;
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const long long off = plane * hw;
    const T* ap = a + off;
    const T* bp = b ? b + off : nullptr;
    int head = (int)((E - (off % E)) % E);
    if (head > hw) head = hw;
    const int nv = (hw - head) / E;
    float s0 = 0.f, s1 = 0.f;
    {   // ragged ends: fewer than 2E <= 16 elements in total
        const int tail0 = head + nv * E;
        int i = -1;
        if (lane < head) i = lane;
        else if (lane - head < hw - tail0) i = tail0 + lane - head;
        if (i >= 0) s0 = to_f32(ap[i]) * (bp ? to_f32(bp[i]) : 1.f);
    }
    const uint4* av = (const uint4*)(ap + head);
    const uint4* bv = bp ? (const uint4*)(bp + head) : nullptr;
    int i = lane;
    for (; i + 64 < nv; i += 128) {
        V16 a0, a1, b0, b1;
        a0.u = av[i]; a1.u = av[i + 64];
        if (bv) {
            b0.u = bv[i]; b1.u = bv[i + 64];
#pragma unroll
            for (int e = 0; e < E; e++) { s0 = fmaf(to_f32(a0.v[e]), to_f32(b0.v[e]), s0); s1 = fmaf(to_f32(a1.v[e]), to_f32(b1.v[e]), s1); }
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) { s0 += to_f32(a0.v[e]); s1 += to_f32(a1.v[e]); }
        }
    }
    if (i < nv) {
        V16 a0, b0;
        a0.u = av[i];
        if (bv) {
            b0.u = bv[i];
#pragma unroll
            for (int e = 0; e < E; e++) s0 = fmaf(to_f32(a0.v[e]), to_f32(b0.v[e]), s0);
        } else {
#pragma unroll
            for (int e = 0; e < E; e++) s0 += to_f32(a0.v[e]);
        }
    }
    float s = s0 + s1;
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) s += __shfl_down(s, off2, 64);
    if (lane == 0) out[plane] = s;
}

// ---------------------------------------------------------------------------------------------
// Weight gradient: dW[o][i][r][s] = sum_n sum_{p,q} dy[n,o,p,q] * x[n,i,p+r-pad,q+s-pad]   (inputs already scaled per plane)
// GEMM with K = pixels: both operands are K-contiguous in NCHW, so the LDS images are plain row copies and the
// 3 column shifts of a tap row come from ONE 5-dword read per row (shift 0: dwords 0-3, shift 2: dwords 1-4,
// shift 1: v_alignbyte of neighbours).  One workgroup = 64 o x 64 i x all taps; 4 waves as 2(o) x 2(i), each
// wave holds KK accumulator tiles of 32x32.  K is split over workgroups by output row; partial sums go to
// a workspace [split][O][I][KK] and are summed by wgrad_reduce_kernel.
constexpr int kWgKQ = 64;        // pixels of one output row per K macro-step

// Row-pitched operands (planes [h][ld], the first w columns of a row meaningful -- the 16-bit activation layout of DESIGN.md
// section 3): 16-byte vectors within a row; the last vector of a row is shifted back to END at column w (no access past the row,
// the re-read elements are selected out), so nothing depends on what the padding holds.  PER_WAVE: one wave per plane, four
// planes per workgroup (small planes, as plane_dot_wave_kernel).  Needs w >= 16 / sizeof(T).
// Gate of the "dot product by homogeneity" (afcm_plane_dot_gated_ld): a plane none of whose strips could reach the clamp takes
//     out = osc (gz - nsc gsk)                     (nsc, gsk may be NULL: 1, 0)
// and its wave / workgroup leaves without touching a or b; a flagged plane gets the real dot product, and so does a plane whose two
// sums cancel to less than 1/8 of their size (|skip| >> |F(y)| amplifies the rounding of z by that ratio).
struct PlaneGate {
    const int* flags;     // [planes][slots], NULL: no gate
    int slots;
    const float *osc, *gz, *nsc, *gsk;
};
template <typename T, bool PER_WAVE>
__global__ __launch_bounds__(256) void plane_dot_rows_kernel(float* __restrict__ out, const T* __restrict__ a, const T* __restrict__ b,
                                                             long long planes, int h, int w, int lda, int ldb, PlaneGate gate) {
    constexpr int E = 16 / (int)sizeof(T);
    union V16 { uint4 u; T v[E]; };
    __shared__ float part[4];
    const int nthr = PER_WAVE ? 64 : 256;
    const int t = PER_WAVE ? (int)(threadIdx.x & 63) : (int)threadIdx.x;
    const long long plane = PER_WAVE ? (long long)blockIdx.x * 4 + (threadIdx.x >> 6) : (long long)blockIdx.x;
    if (plane >= planes) return;                               // PER_WAVE: a whole wave leaves (no barrier below in that mode)
    if (gate.flags != nullptr) {
        int any = 0;                                           // (wave- / workgroup-uniform: every thread reads the same words)
        for (int i = 0; i < gate.slots; i++) any |= gate.flags[plane * gate.slots + i];
        if (!any) {
            // ... unless the two sums cancel: each carries the 16-bit rounding of z / skip (2^-9 relative in bf16), so their difference is
            // trusted down to 1/8 of their size (a skip branch that dwarfs this layer's own output); beyond that the plane takes the real
            // dot product like a flagged one
            const float zz = gate.gz[plane], kk = gate.gsk ? (gate.nsc ? gate.nsc[plane] : 1.f) * gate.gsk[plane] : 0.f;
            const float diff = zz - kk;
            if (!(gate.gsk != nullptr && fabsf(diff) * 8.f < fabsf(zz) + fabsf(kk))) {
                if (t == 0) out[plane] = gate.osc[plane] * diff;
                return;
            }
        }
    }
    const int nvec = (w + E - 1) / E, total = h * nvec;
    const unsigned magic = (unsigned)((0x100000000ull + (unsigned)nvec - 1) / (unsigned)nvec);
    const T* ap = a + plane * h * lda;
    const T* bp = b ? b + plane * h * ldb : nullptr;
    float s0 = 0.f, s1 = 0.f;
    for (int i0 = t; i0 < total; i0 += 4 * nthr) {
        V16 av[4], bv[4];
        int skip[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int i = i0 + u * nthr;
            const int ic = i < total ? i : total - 1;
            const int row = (int)__umulhi((unsigned)ic, magic), col = (ic - row * nvec) * E;
            const int colc = col + E <= w ? col : w - E;
            skip[u] = i < total ? col - colc : E;              // leading elements that an earlier vector already counted (E: none live)
            av[u].u = *(const uint4*)(ap + (size_t)row * lda + colc);
            if (bp) bv[u].u = *(const uint4*)(bp + (size_t)row * ldb + colc);
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
            for (int e = 0; e < E; e++) {
                const float x = e >= skip[u] ? to_f32(av[u].v[e]) : 0.f;
                const float y = bp ? to_f32(bv[u].v[e]) : 1.f;
                if (u & 1) s1 = fmaf(x, e >= skip[u] ? y : 0.f, s1); else s0 = fmaf(x, e >= skip[u] ? y : 0.f, s0);
            }
        }
    }
    float s = s0 + s1;
#pragma unroll
    for (int off2 = 32; off2 > 0; off2 >>= 1) s += __shfl_down(s, off2, 64);
    if (PER_WAVE) {
        if ((threadIdx.x & 63) == 0) out[plane] = s;
    } else {
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) out[plane] = part[0] + part[1] + part[2] + part[3];
    }
}


struct WgradParams {
    const void* dy;   // [N, O, P, Q]
    const void* x;    // [N, I, H, W]
    float* part;      // [splits][O][I][KK]
    int N, O, I, H, W, P, Q, pad;
    int lddy, ldx;                // row pitch (elements) of dy / x; = Q / W for dense tensors.  conv2d_wgrad16g_kernel only.
    int splits, steps_per_split;  // K macro-steps = N * rowgroups * qchunks
    // conv2d_wgrad16g_kernel: two classes of 64 x 64 tiles.  The tiles (obk < fo, ib < fi) are FULL: `splits` workgroups of
    // `steps_per_split` steps each; the others have at most 32 live rows or columns -- two of their four wave quadrants multiply nothing and
    // are skipped -- and get `splits_p` workgroups of `steps_per_split_p` steps (fewer, longer shares: a step costs them ~0.6 of a full
    // tile's).  No partial class: fo / fi = the tile counts, splits_p = splits.
    int fo, fi, splits_p, steps_per_split_p;
    int qchunks;                  // ceil(Q / kWgKQ)
    int rowgroups;                // ceil(P / R)
    // conv2d_wgrad16g_kernel, r06: > 0 = splits per IMAGE (splits = N * splits_img): a split never crosses an image, so the slabs of image n are
    // its own weight gradient dW_n -- what the per-plane dot products <x[n, i], dx[n, i]> are read from (wgrad_reduce_dots_kernel)
    int splits_img;
};

// R = output rows per K macro-step (2 for 16-bit: halves the barriers and re-uses the overlapping input rows).
template <typename T, int KS, int R, int XOFF>
__global__ __launch_bounds__(512, (sizeof(T) == 4 ? 1 : 2)) void conv2d_wgrad_kernel(WgradParams p) {
    constexpr bool F32 = sizeof(T) == 4;
    constexpr int KK = KS * KS;
    constexpr int EPD = F32 ? 1 : 2;               // elements per staged dword
    constexpr int PDY = kWgKQ + 8;                 // 72 elements: 16-bit rows of 144 B (odd x 16 B)
    constexpr int PX = kWgKQ + 24;                 // 88 elements: 176 B rows (odd x 16 B)
    constexpr int XW = kWgKQ + 8;                  // staged x columns per row (shifts 0..KS-1, +1 alignment, rounded)
    constexpr int XR = R + KS - 1;                 // staged x rows per channel
    constexpr int DPR_DY = kWgKQ / EPD, DPR_X = XW / EPD;
    constexpr int LDS_ONE = 64 * R * PDY + 64 * XR * PX;
    constexpr int NBUF = (LDS_ONE * (int)sizeof(T) * 2 <= 150 * 1024) ? 2 : 1;     // double-buffer when it fits the 160 KB LDS
    __shared__ __attribute__((aligned(16))) T lds[NBUF * LDS_ONE];
    T* lds_dy = lds;
    T* lds_x = lds + 64 * R * PDY;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = wave & 1, wi = (wave >> 1) & 1, th = wave >> 2;     // th: which half of the chunk's pixels this wave accumulates
    const int r32 = lane & 31, h = lane >> 5;
    constexpr int NACC = KK;

    int bid = blockIdx.x;
    const int split = bid % p.splits; bid /= p.splits;
    const int ib = bid % cdiv(p.I, 64);
    const int obk = bid / cdiv(p.I, 64);
    const int o0 = obk * 64, i0 = ib * 64;

    f32x16 acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    // ---- staging maps.  Rows of 32 dwords are spread as (row = tid/32 + 8*i, dword = tid%32), so the row-dependent
    // parts of an address advance by a constant per i; the 4-dword tail of every x row is a second small map.
    constexpr int NC = DPR_DY / 32;                 // 32-dword column groups per row (1 for 16-bit, 2 for fp32)
    constexpr int TAILD = DPR_X - 32 * NC;          // dwords of the x-row tail (4 / 8)
    static_assert(DPR_DY % 32 == 0 && TAILD > 0 && TAILD <= 8, "staging maps assume 64-pixel chunks");
    constexpr int NDY = (64 * R) / 16;              // dy rows (o * R + rr) per thread
    constexpr int NXM = (64 * XR) / 16;             // x rows (ic * XR + r), main 32*NC dwords
    constexpr int NXT = cdiv(64 * XR * TAILD, 512); // x tail
    const int rb = tid >> 5, dlane = tid & 31;
    unsigned rdy[NDY][NC], rxm[NXM][NC], rxt[NXT];
    // Loads are raw buffer loads: an invalid element (padding row / column, channel past the end) gets the byte offset
    // kOob >= num_records and reads as zero -- no branches, one v_cndmask per load.  The step-dependent part of every address
    // is wave-uniform and lives in the buffer base; the per-thread byte offsets below never change.
    constexpr unsigned kOob = 0x80000000u;
    constexpr bool ROWSAME = (16 % R == 0) && (16 % XR == 0);      // row-in-step index of a thread is the same for all its loads
    unsigned dyoff[NDY], xoff_[NXM], xtoff[NXT];
    int dyr[NDY], xr_[NXM], xtr[NXT], xtc[NXT];
#pragma unroll
    for (int i = 0; i < NDY; i++) {
        const int row = rb + 16 * i;
        const int o = o0 + row / R;
        dyr[i] = row % R;
        dyoff[i] = o < p.O ? (unsigned)(((o * p.P + dyr[i]) * p.Q + dlane * EPD) * (int)sizeof(T)) : kOob;
    }
#pragma unroll
    for (int i = 0; i < NXM; i++) {
        const int row = rb + 16 * i;
        const int ic = i0 + row / XR;
        xr_[i] = row % XR;
        xoff_[i] = ic < p.I ? (unsigned)(((ic * p.H + xr_[i]) * p.W + dlane * EPD) * (int)sizeof(T)) : kOob;
    }
#pragma unroll
    for (int i = 0; i < NXT; i++) {
        const int j = tid + i * 512;
        const int row = j / TAILD;
        const int ic = i0 + row / XR;
        xtr[i] = row % XR;
        xtc[i] = (32 * NC + j % TAILD) * EPD;
        xtoff[i] = (row < 64 * XR && ic < p.I) ? (unsigned)(((ic * p.H + xtr[i]) * p.W + xtc[i]) * (int)sizeof(T)) : kOob;
    }

    const int steps_per_img = p.rowgroups * p.qchunks;
    const int s0 = split * p.steps_per_split;
    const int s1 = min(s0 + p.steps_per_split, p.N * steps_per_img);
    // (image, row group, column chunk) of the next step to load; steps are loaded in order, so this advances by carries
    int ld_n = s0 / steps_per_img;
    int ld_rg = (s0 - ld_n * steps_per_img) / p.qchunks;
    int ld_qc = s0 - ld_n * steps_per_img - ld_rg * p.qchunks;

    auto issue_loads = [&]() __attribute__((always_inline)) {
        const int prow0 = ld_rg * R, q0 = ld_qc * kWgKQ;
        const int xorg = (q0 - p.pad) & ~1;
        // uniform bases: everything that does not depend on the lane (may point before the tensor for padding rows: those
        // elements are never fetched)
        const T* dyb = (const T*)p.dy + (size_t)ld_n * p.O * p.P * p.Q + (size_t)prow0 * p.Q + q0;
        const T* xb = (const T*)p.x + (long long)ld_n * p.I * p.H * p.W + (long long)(prow0 - p.pad) * p.W + xorg;
        const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc((void*)dyb, 0, kOob, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc((void*)xb, 0, kOob, 0x00020000);
        unsigned dyrows = 0, xrows = 0;               // validity of the R / XR rows of this step
#pragma unroll
        for (int r = 0; r < R; r++) dyrows |= (unsigned)(prow0 + r < p.P) << r;
#pragma unroll
        for (int r = 0; r < XR; r++) xrows |= (unsigned)((unsigned)(prow0 + r - p.pad) < (unsigned)p.H) << r;
#pragma unroll
        for (int c = 0; c < NC; c++) {
            const int dcol = (dlane + 32 * c) * EPD;
            const bool dcok = q0 + dcol < p.Q;
            const bool xcok = (unsigned)(xorg + dcol) < (unsigned)p.W;
            // validity as an offset mask: 0 or kOob, OR-ed into the byte offset (kept arithmetic so that no branch is formed)
            const unsigned dym0 = (unsigned)!(dcok && ((dyrows >> dyr[0]) & 1)) << 31, xm0 = (unsigned)!(xcok && ((xrows >> xr_[0]) & 1)) << 31;
#pragma unroll
            for (int i = 0; i < NDY; i++) {
                const unsigned m = ROWSAME ? dym0 : (unsigned)!(dcok && ((dyrows >> dyr[i]) & 1)) << 31;
                rdy[i][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_dy, (dyoff[i] + 32 * c * 4) | m, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NXM; i++) {
                const unsigned m = ROWSAME ? xm0 : (unsigned)!(xcok && ((xrows >> xr_[i]) & 1)) << 31;
                rxm[i][c] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, (xoff_[i] + 32 * c * 4) | m, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < NXT; i++) {
            const unsigned m = (unsigned)!(((xrows >> xtr[i]) & 1) && (unsigned)(xorg + xtc[i]) < (unsigned)p.W) << 31;
            rxt[i] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, xtoff[i] | m, 0, 0);
        }
        if (++ld_qc == p.qchunks) {
            ld_qc = 0;
            if (++ld_rg == p.rowgroups) { ld_rg = 0; ld_n++; }
        }
    };
    auto write_lds = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NDY; i++)
#pragma unroll
            for (int c = 0; c < NC; c++) *(unsigned*)(lds_dy + (dyr[i] * 64 + (rb + 16 * i) / R) * PDY + (dlane + 32 * c) * EPD) = rdy[i][c];
#pragma unroll
        for (int i = 0; i < NXM; i++)
#pragma unroll
            for (int c = 0; c < NC; c++) *(unsigned*)(lds_x + (xr_[i] * 64 + (rb + 16 * i) / XR) * PX + (dlane + 32 * c) * EPD) = rxm[i][c];
#pragma unroll
        for (int i = 0; i < NXT; i++) {
            const int j = tid + i * 512;
            if (j / TAILD < 64 * XR) *(unsigned*)(lds_x + (xtr[i] * 64 + (j / TAILD) / XR) * PX + xtc[i]) = rxt[i];
        }
    };

    // Pipeline.  Double-buffered (16-bit): while a wave runs the MFMAs of step s from buffer s&1, the others may already be
    // writing step s+1 into the other buffer and have step s+2's global loads in flight: one barrier per step.
    // Single-buffered (fp32): write / barrier / compute / barrier.
    if (s0 < s1) {
        issue_loads();
        write_lds();
        if (NBUF == 2 && s0 + 1 < s1) issue_loads();
    }
    __syncthreads();
    for (int step = s0; step < s1; step++) {
        constexpr int xoff = XOFF;                  // (q0 - pad) & 1 with q0 a multiple of 64: launch-wide constant
        if (NBUF == 2) {
            if (step + 1 < s1) {
                lds_dy = lds + ((step + 1 - s0) & 1) * LDS_ONE;
                lds_x = lds_dy + 64 * R * PDY;
                write_lds();                        // step+1 (its loads were issued one step ago)
                if (step + 2 < s1) issue_loads();
            }
            lds_dy = lds + ((step - s0) & 1) * LDS_ONE;
            lds_x = lds_dy + 64 * R * PDY;
        } else {
            if (step > s0) {
                __syncthreads();
                write_lds();
                __syncthreads();
            }
            if (step + 1 < s1) issue_loads();
        }
        // Both wave halves accumulate every tap; they split the 64 pixels of the chunk (th 0: first 32, th 1: last 32), so a
        // staged x row is read once per 16 pixels and feeds all KS shifts x R rows x KS tap rows.
        if constexpr (F32) {
#pragma unroll
            for (int rr = 0; rr < R; rr++)
#pragma unroll 4
                for (int kq = 0; kq < kWgKQ / 4; kq++) {
                    const int k2 = th * (kWgKQ / 4) + kq;
                    const float a = lds_dy[(rr * 64 + wo * 32 + r32) * PDY + 2 * k2 + h];
#pragma unroll
                    for (int t = 0; t < KK; t++) {
                        const int r = t / KS, sft = t - r * KS;
                        const float b = lds_x[((rr + r) * 64 + wi * 32 + r32) * PX + 2 * k2 + h + sft + xoff];
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
                    }
                }
        } else {
            typedef typename std::conditional<std::is_same<T, bf16_t>::value, bf16x8, f16x8>::type frag_t;
            const T* dy_w = lds_dy + (wo * 32 + r32) * PDY + th * (kWgKQ / 2) + 8 * h;
            const T* x_w = lds_x + (wi * 32 + r32) * PX + th * (kWgKQ / 2) + 8 * h;
#pragma unroll
            for (int kq = 0; kq < kWgKQ / 32; kq++) {
                frag_t a[R];
#pragma unroll
                for (int rr = 0; rr < R; rr++) a[rr] = *(const frag_t*)(dy_w + rr * 64 * PDY + kq * 16);
                // the staged rows of this lane's channel: each read once, shifted variants built in registers
#pragma unroll
                for (int xr = 0; xr < XR; xr++) {
                    const unsigned* src = (const unsigned*)(x_w + xr * 64 * PX + kq * 16);
                    const uint4 lo = *(const uint4*)src;
                    const uint4 hi = *(const uint4*)(src + 4);        // 16-byte read (conflict-free); only .x/.y are used
                    const unsigned d[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
#pragma unroll
                    for (int sft = 0; sft < KS; sft++) {
                        union { unsigned u[4]; frag_t f; } b;
                        const int sh = sft + xoff;                            // element shift in [0, 3], compile-time
#pragma unroll
                        for (int w = 0; w < 4; w++) {
                            const unsigned e0 = d[w], e1 = d[w + 1], e2 = d[w + 2];
                            const unsigned odd_lo = __builtin_amdgcn_alignbyte(e1, e0, 2);
                            const unsigned odd_hi = __builtin_amdgcn_alignbyte(e2, e1, 2);
                            b.u[w] = (sh == 0) ? e0 : (sh == 1) ? odd_lo : (sh == 2) ? e1 : odd_hi;
                        }
#pragma unroll
                        for (int rr = 0; rr < R; rr++) {
                            const int r = xr - rr;
                            const int t = r * KS + sft;
                            if (r >= 0 && r < KS) {
                                if constexpr (std::is_same<T, bf16_t>::value)
                                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr], b.f, acc[t], 0, 0, 0);
                                else
                                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rr], b.f, acc[t], 0, 0, 0);
                            }
                        }
                    }
                }
            }
        }
        if (NBUF == 2) __syncthreads();            // everyone done with buffer step&1 and step+1 fully written
    }
    // ---- add the two pixel halves through LDS (the staging buffers are free now): th 1 parks its accumulators, th 0 adds
    // them and writes the partial tile D[row = o][col = i].
    if (NBUF == 1) __syncthreads();
    {
        float* red = (float*)lds;
        constexpr int TPR_CAP = (int)((size_t)NBUF * LDS_ONE * sizeof(T) / (4 * 16 * 64 * sizeof(float)));     // taps per round
        constexpr int TPR = TPR_CAP < KK ? TPR_CAP : KK;
        static_assert(TPR >= 1, "LDS too small for the half-sum");
        const int wv4 = wave & 3;
#pragma unroll
        for (int t0 = 0; t0 < KK; t0 += TPR) {
            if (t0 > 0) __syncthreads();
            if (th == 1) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < KK; t++)
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) red[((wv4 * TPR + (t - t0)) * 16 + reg) * 64 + lane] = acc[t][reg];
            }
            __syncthreads();
            if (th == 0) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < KK; t++)
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) acc[t][reg] += red[((wv4 * TPR + (t - t0)) * 16 + reg) * 64 + lane];
            }
        }
    }
    if (th == 0) {
        float* out = p.part + (size_t)split * p.O * p.I * KK;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int o = o0 + wo * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int i = i0 + wi * 32 + r32;
            if (o < p.O && i < p.I) {
                float* dst = out + ((size_t)o * p.I + i) * KK;
#pragma unroll
                for (int t = 0; t < KK; t++) dst[t] = acc[t][reg];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit weight gradient, LDS-DMA staged.  Same tiling as conv2d_wgrad_kernel (64 o x 64 i x all taps per workgroup, R = 2
// output rows x 64 pixels per K step, 8 waves = 2(o) x 2(i) x 2(pixel halves)), but the operands go HBM -> LDS directly
// (buffer_load_dword ... lds): no staging VGPRs, no ds_write pass, no per-load VALU.  One wave instruction fills 64
// consecutive LDS dwords = two 128-byte rows (64 pixels of two channels), so rows cannot be padded; bank conflicts are
// avoided by an XOR swizzle of the 16-byte granules, applied on the SOURCE address of the load and again on the read:
//     granule g of row r sits at physical granule g ^ ((r >> 1) & 7)            (conflict-free for ds_read_b128's lane groups)
// LDS image of one step (NBUF of them in a ring):
//     dy  [rr 0..1][o 0..63][128 B]                                             16 KB
//     x   [xr 0..XR-1] { main [ch 0..63][128 B] (cols 0..63), tail [ch 0..63][16 B] (cols 64..71) }   XR x 9 KB
// Every piece is predicated by the buffer descriptor: rows outside the image get num_records = 0, channels past the end
// fall behind num_records, columns past the end get the out-of-range offset bit -- all of them read as zero.
typedef __attribute__((ext_vector_type(4))) int i32x4;

template <int LO, int HI, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        static_for<LO + 1, HI>(f);
    }
}

__device__ __forceinline__ void lds_dma_dword(i32x4 rsrc, unsigned voff, unsigned lds_addr) {
    // M0 carries the wave-uniform LDS destination; lane l lands at lds_addr + 4*l
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc));
}

// 16 bytes per lane: lane l lands at lds_addr + 16*l
__device__ __forceinline__ void lds_dma_b128(i32x4 rsrc, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" : : "s"(lds_addr), "v"(voff), "s"(rsrc));
}

__device__ __forceinline__ i32x4 make_rsrc(const void* base, int num_records) {
    // the descriptor is wave-uniform by construction; readfirstlane pins it to SGPRs for the "s" asm operand
    const unsigned long long a = (unsigned long long)base;
    i32x4 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32) & 0xffff);       // stride 0
    r.z = __builtin_amdgcn_readfirstlane(num_records);
    r.w = 0x00020000;
    return r;
}

template <typename T, int KS, int XOFF, int NBUF>
__global__ __launch_bounds__(512, 1) void conv2d_wgrad16_kernel(WgradParams p) {
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int R = 2, KK = KS * KS, XR = R + KS - 1;
    constexpr int ROWB = 128;                       // bytes of one staged row (64 pixels)
    constexpr int DY_BYTES = R * 64 * ROWB;
    constexpr int XBLK = 64 * ROWB + 64 * 16;       // one staged x row of all 64 channels: main + tail
    constexpr int BUF = DY_BYTES + XR * XBLK;
    constexpr bool TAIL = KS > 1;
    constexpr int NTAILP = TAIL ? (4 * XR) / 8 : 0; // tail pieces per wave
    static_assert(!TAIL || (4 * XR) % 8 == 0, "tail pieces must divide over the 8 waves");
    constexpr int NPIECE = 8 + 4 * XR + NTAILP;     // LDS-DMA instructions per wave and step
    __shared__ __attribute__((aligned(256))) char lds[NBUF * BUF];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wo = wave & 1, wi = (wave >> 1) & 1, th = wave >> 2;     // th: which half of the chunk's pixels this wave accumulates
    const int r32 = lane & 31, h = lane >> 5;

    // XCD-aware block order: the 8 XCDs take workgroups round-robin, so give XCD x a contiguous range of logical ids;
    // logical id = split-major, i.e. one XCD's L2 sees all (o, i) tiles of the same pixels.
    const int tiles_i = cdiv(p.I, 64), tiles = tiles_i * cdiv(p.O, 64);
    int bid = blockIdx.x;
    const int total = tiles * p.splits;
    bid = xcd_order(bid, total);
    const int split = bid / tiles;
    const int tile = bid - split * tiles;
    const int ib = tile % tiles_i, obk = tile / tiles_i;
    const int o0 = obk * 64, i0 = ib * 64;

    f32x16 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    // ---- load maps.  Wave w owns the row pairs {w, w+8, w+16, w+24} of every 64-row block, so its swizzle key
    // ((row >> 1) & 7) == w is a constant and one per-lane offset serves all of its pieces.
    const int half = lane >> 5, slot = lane & 31;
    const int cdw = ((((slot >> 2) ^ wave) & 7) << 2) | (slot & 3);          // logical dword column of this lane's LDS slot
    const unsigned lp_dy = (unsigned)((2 * wave + half) * p.P * p.Q * 2 + cdw * 4);
    const unsigned lp_x = (unsigned)((2 * wave + half) * p.H * p.W * 2 + cdw * 4);
    const int trow = lane >> 2, tdw = lane & 3;                               // tail piece: 16 channels x 4 dwords
    const unsigned lp_t = (unsigned)(trow * p.H * p.W * 2 + (32 + tdw) * 4);

    const int steps_per_img = p.rowgroups * p.qchunks;
    const int s0 = split * p.steps_per_split;
    const int s1 = min(s0 + p.steps_per_split, p.N * steps_per_img);
    int ld_n = s0 / steps_per_img;
    int ld_rg = (s0 - ld_n * steps_per_img) / p.qchunks;
    int ld_qc = s0 - ld_n * steps_per_img - ld_rg * p.qchunks;
    int ld_buf = 0;

    // Loads of one step: begin_loads() latches the step's uniform state, issue_piece<I>() issues LDS-DMA instruction I of the
    // wave's NPIECE.  The pieces are spread between the MFMAs of the previous step's compute: a dword load occupies the
    // address unit for 16 cycles, so a burst of 26 x 8 waves would stall every wave at the head of the step.
    int c_prow0 = 0, c_q0 = 0, c_xorg = 0, c_live = 0;
    unsigned c_bufa = 0, v_dy = 0, v_x = 0, v_t = 0;
    const T* c_dyn = nullptr;
    const T* c_xn = nullptr;
    const int pq = p.P * p.Q, hw = p.H * p.W;
    auto begin_loads = [&](bool live) __attribute__((always_inline)) {
        c_live = live ? -1 : 0;                                   // a dead step still issues its pieces (with 0 records)
        c_prow0 = ld_rg * R; c_q0 = ld_qc * kWgKQ;
        c_xorg = (c_q0 - p.pad) & ~1;
        c_bufa = lds0 + ld_buf * BUF;
        // per-lane column validity -> offset masks
        v_dy = lp_dy | ((unsigned)!(c_q0 + 2 * cdw < p.Q) << 31);
        v_x = lp_x | ((unsigned)!((unsigned)(c_xorg + 2 * cdw) < (unsigned)p.W) << 31);
        v_t = lp_t | ((unsigned)!((unsigned)(c_xorg + 64 + 2 * tdw) < (unsigned)p.W) << 31);
        c_dyn = (const T*)p.dy + (size_t)ld_n * p.O * pq;
        c_xn = (const T*)p.x + (size_t)ld_n * p.I * hw;
        if (live) {
            if (++ld_qc == p.qchunks) {
                ld_qc = 0;
                if (++ld_rg == p.rowgroups) { ld_rg = 0; ld_n++; }
            }
        }
        if (++ld_buf == NBUF) ld_buf = 0;
    };
    auto issue_piece = [&](auto idx) __attribute__((always_inline)) {
        constexpr int I = decltype(idx)::value;
        if constexpr (I < 8) {
            constexpr int rr = I >> 2, mm = I & 3;
            const int ch0 = o0 + 16 * mm, row = c_prow0 + rr;
            const int inimg = row * p.Q + c_q0;                                // element offset of the piece origin inside a channel
            int nr = ((p.O - ch0) * pq - inimg) * 2;
            nr = (row < p.P && nr > 0) ? (nr & c_live) : 0;
            lds_dma_dword(make_rsrc(c_dyn + (long long)ch0 * pq + inimg, nr), v_dy, c_bufa + rr * (64 * ROWB) + (wave + 8 * mm) * 256);
        } else if constexpr (I < 8 + 4 * XR) {
            constexpr int m = I - 8, xr = m >> 2, mm = m & 3;
            const int ch0 = i0 + 16 * mm, row = c_prow0 - p.pad + xr;
            const int inimg = row * p.W + c_xorg;
            int nr = ((p.I - ch0) * hw - inimg) * 2;
            nr = ((unsigned)row < (unsigned)p.H && nr > 0) ? (nr & c_live) : 0;
            lds_dma_dword(make_rsrc(c_xn + (long long)ch0 * hw + inimg, nr), v_x, c_bufa + DY_BYTES + xr * XBLK + (wave + 8 * mm) * 256);
        } else {
            constexpr int u = I - 8 - 4 * XR;
            const int q = wave + 8 * u;
            const int xr = q >> 2, t = q & 3;
            const int ch0 = i0 + 16 * t, row = c_prow0 - p.pad + xr;
            const int inimg = row * p.W + c_xorg;
            int nr = ((p.I - ch0) * hw - inimg) * 2;
            nr = ((unsigned)row < (unsigned)p.H && nr > 0) ? (nr & c_live) : 0;
            lds_dma_dword(make_rsrc(c_xn + (long long)ch0 * hw + inimg, nr), v_t, c_bufa + DY_BYTES + xr * XBLK + 64 * ROWB + t * 256);
        }
    };
    // pieces [lo, hi) as one unrolled run
    auto issue_range = [&](auto lo, auto hi) __attribute__((always_inline)) {
        constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
        static_for<LO, HI>([&](auto i) __attribute__((always_inline)) { issue_piece(i); });
    };

    // ---- fragment read offsets inside a buffer (swizzled); the two 16-pixel groups of this wave's half need their own
    const int rowA = wo * 32 + r32, rowB = wi * 32 + r32;
    const int fA = (rowA >> 1) & 7, fB = (rowB >> 1) & 7;
    unsigned a_off[2], xlo_off[2], xhi_off[2];
#pragma unroll
    for (int kq = 0; kq < 2; kq++) {
        const int g = th * 4 + kq * 2 + h;
        a_off[kq] = rowA * ROWB + ((g ^ fA) << 4);
        xlo_off[kq] = DY_BYTES + rowB * ROWB + ((g ^ fB) << 4);
        xhi_off[kq] = (g + 1 < 8) ? DY_BYTES + rowB * ROWB + (((g + 1) ^ fB) << 4) : DY_BYTES + 64 * ROWB + rowB * 16;
    }

    // ---- pipeline: NBUF-1 steps of loads in flight; a step's loads are waited for (counted vmcnt) before the barrier that
    // precedes its use.
#pragma unroll
    for (int i = 0; i < NBUF - 1; i++) {
        begin_loads(s0 + i < s1);
        issue_range(std::integral_constant<int, 0>{}, std::integral_constant<int, NPIECE>{});
    }
    if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NPIECE));
    else asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    int cbuf = 0;
    for (int step = s0; step < s1; step++) {
        constexpr int xoff = XOFF;
        begin_loads(step + NBUF - 1 < s1);                 // into the buffer everyone left at the last barrier
        const char* buf = lds + cbuf * BUF;
        typedef typename std::conditional<std::is_same<T, bf16_t>::value, bf16x8, f16x8>::type frag_t;
        static_for<0, 2>([&](auto kqc) __attribute__((always_inline)) {
            constexpr int kq = decltype(kqc)::value;
            frag_t a[R];
#pragma unroll
            for (int rr = 0; rr < R; rr++) a[rr] = *(const frag_t*)(buf + a_off[kq] + rr * (64 * ROWB));
            static_for<0, XR>([&](auto xrc) __attribute__((always_inline)) {
                constexpr int xr = decltype(xrc)::value;
                constexpr int it = kq * XR + xr, NIT = 2 * XR;
                issue_range(std::integral_constant<int, (it * NPIECE) / NIT>{}, std::integral_constant<int, ((it + 1) * NPIECE) / NIT>{});
                const uint4 lo = *(const uint4*)(buf + xlo_off[kq] + xr * XBLK);
                uint4 hi = lo;
                if (TAIL) {
                    hi = *(const uint4*)(buf + xhi_off[kq] + xr * XBLK);
                    asm volatile("" : : "v"(hi.y), "v"(hi.z), "v"(hi.w));     // keep the read a full (conflict-free) b128
                }
                const unsigned d[6] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y};
#pragma unroll
                for (int sft = 0; sft < KS; sft++) {
                    union { unsigned u[4]; frag_t f; } b;
                    const int sh = sft + xoff;                            // element shift in [0, 3], compile-time
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        const unsigned e0 = d[w], e1 = d[w + 1], e2 = d[(w + 2) % 6];
                        const unsigned odd_lo = __builtin_amdgcn_alignbyte(e1, e0, 2);
                        const unsigned odd_hi = __builtin_amdgcn_alignbyte(e2, e1, 2);
                        b.u[w] = (sh == 0) ? e0 : (sh == 1) ? odd_lo : (sh == 2) ? e1 : odd_hi;
                    }
#pragma unroll
                    for (int rr = 0; rr < R; rr++) {
                        const int r = xr - rr;
                        const int t = r * KS + sft;
                        if (r >= 0 && r < KS) {
                            if constexpr (std::is_same<T, bf16_t>::value)
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr], b.f, acc[t], 0, 0, 0);
                            else
                                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rr], b.f, acc[t], 0, 0, 0);
                        }
                    }
                }
            });
        });
        // the next step's loads (issued NBUF-2 iterations ago, or just now when NBUF == 2) must have landed
        if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NPIECE));
        else asm volatile("s_waitcnt vmcnt(0)");
        __syncthreads();
        if (++cbuf == NBUF) cbuf = 0;
    }
    // ---- add the two pixel halves through LDS (the ring is free now): th 1 parks its accumulators, th 0 adds them and
    // writes the partial tile D[row = o][col = i].
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    {
        float* red = (float*)lds;
        constexpr int TPR_CAP = (NBUF * BUF) / (4 * 16 * 64 * (int)sizeof(float));     // taps per round
        constexpr int TPR = TPR_CAP < KK ? TPR_CAP : KK;
        static_assert(TPR >= 1, "LDS too small for the half-sum");
        const int wv4 = wave & 3;
#pragma unroll
        for (int t0 = 0; t0 < KK; t0 += TPR) {
            if (t0 > 0) __syncthreads();
            if (th == 1) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < KK; t++)
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) red[((wv4 * TPR + (t - t0)) * 16 + reg) * 64 + lane] = acc[t][reg];
            }
            __syncthreads();
            if (th == 0) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < KK; t++)
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) acc[t][reg] += red[((wv4 * TPR + (t - t0)) * 16 + reg) * 64 + lane];
            }
        }
    }
    if (th == 0) {
        float* out = p.part + (size_t)split * p.O * p.I * KK;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            const int o = o0 + wo * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
            const int i = i0 + wi * 32 + r32;
            if (o < p.O && i < p.I) {
                float* dst = out + ((size_t)o * p.I + i) * KK;
#pragma unroll
                for (int t = 0; t < KK; t++) dst[t] = acc[t][reg];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// 16-bit weight gradient, 16-byte LDS-DMA pieces.  The address unit spends about the same time on a wave instruction
// whatever its width, and conv2d_wgrad16_kernel is bound by exactly that (208 dword pieces per step); this variant moves
// the same bytes in 56 pieces of 16 B per lane.  A lane fetches one granule = 8 pixels from a 4-byte aligned address, so
// validity is per granule, not per pixel:
//   * x is staged from column q0 - 8: the granule left of the image is dropped whole (pad = 2: taps reach back 2 columns);
//   * the granule that straddles the right edge of x brings the head of the next row: the wave that loaded it zeroes those
//     pixels in LDS before the barrier;
//   * dy beyond its right edge then multiplies zeros of x (its columns q >= Q pair with x columns >= W), whatever it holds.
// Supported: KS = 3 with pad = 2 (the generator's convs) and KS = 1 with pad = 0; everything else takes the dword kernel.
// LDS image of one step:   dy [rr 0..1][o 64][8 granules]   x main [xr][ch 64][8 granules]   x tail [ch>>3][xr][ch&7][1 granule]
// with granule g of row r at slot g ^ ((r >> 1) & 7); B fragments of tap column s are the 5-dword window
// (granule g).d3, (granule g+1).d0..3 shifted by s.

// SMALL (both tensors below 2^31 bytes): ONE buffer descriptor per tensor for the whole kernel; a piece's position is a 32-bit
// offset added to the lane offsets and its validity (row outside the image, dead step) an OR mask on bit 31.  The general form
// rebuilds a 128-bit descriptor per piece -- 64-bit base, exact record count, validity select, three v_readfirstlane --
// ~45 scalar instructions per piece, 310 per K step of 36 MFMAs: the wave's own instruction stream, not the matrix pipe, set
// the step time (PMC r01e: MFMA pipe 49 % busy, 8.7 SALU per MFMA).
#ifndef AFCM_CONV_BM96
#define AFCM_CONV_BM96 1           // 65 .. 96 output rows on one 96-row block of conv2d_fwd16x_kernel (0: a 128-row block; A/B builds)
#endif
#ifndef AFCM_CONV_BM96_192
#define AFCM_CONV_BM96_192 0       // experiment: 129 .. 192 rows as two 96-row blocks instead of 128 + 64 (measured: the same, profiles/r05_conv_bm96_ab.txt)
#endif
#ifndef AFCM_CONV_MIXED
#define AFCM_CONV_MIXED 1          // 128 k + (1 .. 64) output rows: 128-row kernel + one 64-row block (0: 64-row blocks only; A/B builds)
#endif
#ifndef AFCM_WGRAD_LATE
#define AFCM_WGRAD_LATE 1          // begin_loads behind the first iteration's MFMAs: 8.55 -> 8.28 ms in the step (profiles/r05_wgrad_late_ab.txt); 0: at the step's top
#endif
#ifndef AFCM_WGRAD_NBUF
#define AFCM_WGRAD_NBUF 3          // LDS ring depth of conv2d_wgrad16g_kernel (2: measured in profiles/r04_wgrad_ring.txt)
#endif
// X16 (r05): the same tile on v_mfma_f32_16x16x32 -- a wave's 32 (o) x 32 (i) tile is 2 x 2 tiles of 16 x 16 per tap (the same 144
// accumulator registers), a K step is 32 pixels = FOUR granules, one per 16-lane group: wave th takes pixels 32 th .. 32 th + 31 of the
// chunk.  ds_read_b128 serves the lanes in groups that hold all 16 rows with TWO neighbouring granules (G, G + 1), so the swizzle is
// slot = granule ^ (((row >> 1) & 3) << 1): both row sets of a group take the even XOR values once, G and G + 1 differ in bit 0 (G even)
// or flip bits that keep the even set (G odd: the hi half of the x windows): 16 distinct slots for every read (the (row >> 1) & 7 form
// of the 32x32x16 kernel is conflict-free only when all lanes of a group read the SAME granule).  A/B of the shapes: profiles/r05_*.
template <typename T, int KS, int NBUF, bool SMALL, bool X16 = false>
__global__ __launch_bounds__(512, 1) void conv2d_wgrad16g_kernel(WgradParams p) {
    static_assert(sizeof(T) == 2, "16-bit types only");
    constexpr int R = 2, KK = KS * KS, XR = R + KS - 1;
    auto swz = [](int row) __attribute__((always_inline)) { return X16 ? (((row >> 1) & 3) << 1) : ((row >> 1) & 7); };
    constexpr int ROWB = 128;                       // bytes of one staged row (64 pixels)
    constexpr int DY_BYTES = R * 64 * ROWB;
    constexpr int XMAIN = XR * 64 * ROWB;
    constexpr bool TAIL = KS > 1;
    constexpr int XTAIL = TAIL ? 8 * XR * 8 * 16 : 0;               // [ch>>3][xr][ch&7][16 B]
    constexpr int BUF = DY_BYTES + XMAIN + XTAIL;
    constexpr int NPIECE = R + XR + (TAIL ? 1 : 0);                 // LDS-DMA instructions per wave and step
    constexpr int XLEAD = TAIL ? 8 : 0;                             // x is staged from column q0 - XLEAD
    constexpr unsigned kOob = 0x80000000u;
    static_assert(XR * 8 <= 64, "tail piece: one lane per (xr, channel)");
    __shared__ __attribute__((aligned(1024))) char lds[NBUF * BUF];
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // th: which half of the chunk's pixels this wave accumulates; (wo, wi): its 32 x 32 quadrant of the tile.  The two waves of a SIMD
    // (wave, wave + 4) take DIAGONALLY OPPOSITE quadrants: when a tile's rows 32.. or columns 32.. lie outside the matrix (the last tile of a
    // 91-channel operand: 27 live rows), every SIMD then holds one live and one idle wave instead of two SIMDs holding both
    const int th = wave >> 2, wo = (wave & 1) ^ th, wi = ((wave >> 1) & 1) ^ th;
    const int r32 = lane & 31, h = lane >> 5;

    const int tiles_i = cdiv(p.I, 64), tiles_o = cdiv(p.O, 64);
    int bid = blockIdx.x;
    const int n_full = p.fo * p.fi, total = n_full * p.splits + (tiles_i * tiles_o - n_full) * p.splits_p;
    bid = xcd_order(bid, total);          // XCD x: contiguous logical ids (split-major within a class)
    // integer division runs on the vector pipe even for uniform operands: pin the results to SGPRs, or every per-step address
    // and descriptor computation derived from them runs as 64-bit VALU code + v_readfirstlane (measured: ~130 vector
    // instructions per step beside the 36 MFMAs)
    int split, obk, ib, my_steps;
    if (bid < n_full * p.splits) {
        split = __builtin_amdgcn_readfirstlane(bid / n_full);
        const int tile = bid - split * n_full;                                       // index in the fo x fi grid of full tiles
        obk = __builtin_amdgcn_readfirstlane(tile / p.fi);
        ib = tile - obk * p.fi;
        my_steps = p.steps_per_split;
    } else {
        // partial tiles: the column ib = fi (when fi < tiles_i), all tiles_o rows of it, then the row obk = fo (fo < tiles_o), ib < fi
        const int rel = bid - n_full * p.splits, n_part = tiles_i * tiles_o - n_full;
        split = __builtin_amdgcn_readfirstlane(rel / n_part);
        const int j = rel - split * n_part;
        const int ncol = p.fi < tiles_i ? tiles_o : 0;
        if (j < ncol) { obk = j; ib = p.fi; } else { obk = p.fo; ib = j - ncol; }
        my_steps = p.steps_per_split_p;
    }
    const int o0 = obk * 64, i0 = ib * 64;

    // 32x32x16: one 32 x 32 tile per tap, element e = MFMA register e; 16x16x32: element 4 (2 ob2 + ib2) + reg of the (ob2, ib2) 16 x 16 tile
    f32x16 acc[KK];
#pragma unroll
    for (int t = 0; t < KK; t++)
#pragma unroll
        for (int e = 0; e < 16; e++) acc[t][e] = 0.f;

    // ---- load maps.  Wave w stages rows (channels) 8w..8w+7 of both operands: lane = (row, slot) of an 8-row piece.
    const int prow = 8 * wave + (lane >> 3), pslot = lane & 7;
    const int pg = pslot ^ swz(prow);                                            // logical granule that lives in this slot
    const int pq = p.P * p.lddy, hw = p.H * p.ldx;                                // plane strides (rows by pitch)
    // p.W also feeds per-lane offsets, so the compiler keeps it in a VGPR and then evaluates the (uniform) row addresses of the
    // x pieces on the vector pipe; an explicit scalar copy keeps them on the SALU
    const int Ws = __builtin_amdgcn_readfirstlane(p.ldx), Qs = __builtin_amdgcn_readfirstlane(p.lddy);
    const unsigned lp_dy = (o0 + prow < p.O) ? (unsigned)(prow * pq * 2 + pg * 16) : kOob;
    const unsigned lp_x = (i0 + prow < p.I) ? (unsigned)(prow * hw * 2 + pg * 16) : kOob;
    const int txr = lane >> 3, trow = 8 * wave + (lane & 7);                     // tail piece: lane = (xr, row), lanes >= 8 XR idle
    const unsigned lp_t = (i0 + trow < p.I) ? (unsigned)((trow * hw + txr * p.ldx + 64) * 2) : kOob;
    const long long dy_bytes = (long long)p.N * p.O * pq * 2, x_bytes = (long long)p.N * p.I * hw * 2;
    // SMALL: the two descriptors of the kernel (records = the tensor's bytes: a granule straddling its end reads zeros there)
    const i32x4 rs_dy = make_rsrc(p.dy, SMALL ? (int)dy_bytes : 0), rs_x = make_rsrc(p.x, SMALL ? (int)x_bytes : 0);
    unsigned c_dy32 = 0, c_x32 = 0;                                              // byte offsets of (image n, channel o0 / i0)

    const int steps_per_img = p.rowgroups * p.qchunks;
    int s0 = split * my_steps;
    int s1 = min(s0 + my_steps, p.N * steps_per_img);
    if (p.splits_img > 0) {                                                      // image-aligned shares (my_steps = ceil(steps_per_img / splits_img))
        const int img = __builtin_amdgcn_readfirstlane(split / p.splits_img), j = split - img * p.splits_img;
        s0 = img * steps_per_img + j * my_steps;
        s1 = min(s0 + my_steps, (img + 1) * steps_per_img);
        if (s1 < s0) s1 = s0;                                                    // (a share past the image's last step: zeros)
    }
    // a quadrant wholly outside the matrix: its waves only issue their share of the loads
    const bool quad_dead = o0 + wo * 32 >= p.O || i0 + wi * 32 >= p.I;
    int ld_n = __builtin_amdgcn_readfirstlane(s0 / steps_per_img);
    int ld_rg = __builtin_amdgcn_readfirstlane((s0 - ld_n * steps_per_img) / p.qchunks);
    int ld_qc = s0 - ld_n * steps_per_img - ld_rg * p.qchunks;
    int ld_buf = 0;
    int u_qc = ld_qc;                                                            // chunk (of its row pair) of the step being multiplied

    // state of the step being loaded (c_*) and of the one before it (f_*: the step whose x edge is fixed up next)
    int c_prow0 = 0, c_q0 = 0, c_live = 0, f_q0 = 0, f_live = 0;
    unsigned c_bufa = 0, f_bufa = 0, v_dy = 0, v_x = 0, v_t = 0;
    long long c_dyoff = 0, c_xoff = 0;                                           // byte offsets of the image (n) in dy / x
    auto begin_loads = [&](bool live) __attribute__((always_inline)) {
        f_q0 = c_q0; f_live = c_live; f_bufa = c_bufa;
        c_live = live ? -1 : 0;                                                  // a dead step still issues its pieces (0 records)
        c_prow0 = ld_rg * R; c_q0 = ld_qc * kWgKQ;
        c_bufa = ld_buf * BUF;
        const int xorg = c_q0 - XLEAD;
        v_dy = lp_dy | ((unsigned)!(c_q0 + 8 * pg < p.Q) << 31);
        v_x = lp_x | ((unsigned)!((unsigned)(xorg + 8 * pg) < (unsigned)p.W) << 31);
        v_t = lp_t | ((unsigned)!((unsigned)(xorg + 64) < (unsigned)p.W && (unsigned)(c_prow0 - p.pad + txr) < (unsigned)p.H) << 31);
        c_dyoff = (long long)ld_n * p.O * pq * 2;
        c_xoff = (long long)ld_n * p.I * hw * 2;
        if (SMALL) {
            c_dy32 = (unsigned)((ld_n * p.O + o0) * pq) * 2u;
            c_x32 = (unsigned)((ld_n * p.I + i0) * hw) * 2u;
        }
        if (live) {
            if (++ld_qc == p.qchunks) {
                ld_qc = 0;
                if (++ld_rg == p.rowgroups) { ld_rg = 0; ld_n++; }
            }
        }
        if (++ld_buf == NBUF) ld_buf = 0;
    };
    // records = bytes up to the end of the tensor (a straddling granule's dwords beyond it read as zero), 0 for a dead row
    auto records = [&](long long remaining, bool ok) __attribute__((always_inline)) -> int {
        const int r = remaining > 0x7fffffffll ? 0x7fffffff : (int)remaining;
        return (ok && r > 0) ? (r & c_live) : 0;
    };
    auto issue_piece = [&](auto idx) __attribute__((always_inline)) {
        constexpr int I = decltype(idx)::value;
        if constexpr (SMALL) {
            // offset of the piece (scalar) + lane offset; invalid lanes carry bit 31 in v_*, an invalid piece ORs it in for all
            if constexpr (I < R) {
                constexpr int rr = I;
                const int row = c_prow0 + rr;
                const unsigned soff = c_dy32 + (unsigned)(row * Qs + c_q0) * 2u;
                const unsigned sinv = (row < p.P && c_live) ? 0u : kOob;
                lds_dma_b128(rs_dy, ((v_dy & ~kOob) + soff) | (v_dy & kOob) | sinv, lds0 + c_bufa + rr * (64 * ROWB) + wave * 1024);
            } else if constexpr (I < R + XR) {
                constexpr int xr = I - R;
                const int row = c_prow0 - p.pad + xr;
                const unsigned soff = c_x32 + (unsigned)(row * Ws + c_q0 - XLEAD) * 2u;
#ifdef AFCM_WGRAD_EXPERIMENT_HALFX      // timing experiment only (wrong results): the two x rows shared with the previous row pair are not requested.
                                        // CAUTION: its -8 % is the clock, not the bytes -- the rows then multiply zeros (profiles/r04_wgrad_ring.txt)
                const unsigned sinv = ((unsigned)row < (unsigned)p.H && c_live && xr >= 2) ? 0u : kOob;
#else
                const unsigned sinv = ((unsigned)row < (unsigned)p.H && c_live) ? 0u : kOob;
#endif
                lds_dma_b128(rs_x, ((v_x & ~kOob) + soff) | (v_x & kOob) | sinv, lds0 + c_bufa + DY_BYTES + xr * (64 * ROWB) + wave * 1024);
            } else {
                const int row = c_prow0 - p.pad;                                 // lanes add their xr
                const unsigned soff = c_x32 + (unsigned)(row * Ws + c_q0 - XLEAD) * 2u;
                const unsigned sinv = c_live ? 0u : kOob;
                if (lane < 8 * XR)
                    lds_dma_b128(rs_x, ((v_t & ~kOob) + soff) | (v_t & kOob) | sinv, lds0 + c_bufa + DY_BYTES + XMAIN + wave * (XR * 128));
            }
        } else if constexpr (I < R) {
            constexpr int rr = I;
            const int row = c_prow0 + rr;
            const long long off = c_dyoff + ((long long)o0 * pq + row * Qs + c_q0) * 2;
            lds_dma_b128(make_rsrc((const char*)p.dy + off, records(dy_bytes - off, row < p.P)), v_dy,
                         lds0 + c_bufa + rr * (64 * ROWB) + wave * 1024);
        } else if constexpr (I < R + XR) {
            constexpr int xr = I - R;
            const int row = c_prow0 - p.pad + xr;
            const long long off = c_xoff + ((long long)i0 * hw + row * Ws + c_q0 - XLEAD) * 2;
            lds_dma_b128(make_rsrc((const char*)p.x + off, records(x_bytes - off, (unsigned)row < (unsigned)p.H)), v_x,
                         lds0 + c_bufa + DY_BYTES + xr * (64 * ROWB) + wave * 1024);
        } else {
            const int row = c_prow0 - p.pad;                                     // lanes add their xr
            const long long off = c_xoff + ((long long)i0 * hw + row * Ws + c_q0 - XLEAD) * 2;
            if (lane < 8 * XR)
                lds_dma_b128(make_rsrc((const char*)p.x + off, records(x_bytes - off, true)), v_t,
                             lds0 + c_bufa + DY_BYTES + XMAIN + wave * (XR * 128));
        }
    };
    auto issue_range = [&](auto lo, auto hi) __attribute__((always_inline)) {
        constexpr int LO = decltype(lo)::value, HI = decltype(hi)::value;
        static_for<LO, HI>([&](auto i) __attribute__((always_inline)) { issue_piece(i); });
    };
    // zero the pixels right of x's edge inside the granule that straddles it, in the rows this wave loaded (step f_*)
    auto fix_edge = [&]() __attribute__((always_inline)) {
        const int rel = p.W - (f_q0 - XLEAD);                                    // edge column relative to the staged origin
        const int gw = rel >> 3, vw = rel & 7;                                   // granule, valid pixels in it (even)
        if (f_live && vw != 0 && gw >= 0 && gw <= (TAIL ? 8 : 7) && lane < 8 * XR) {
            const int xr = lane >> 3, row = 8 * wave + (lane & 7);
            char* g = lds + f_bufa + DY_BYTES +
                      (gw < 8 ? xr * (64 * ROWB) + row * ROWB + ((gw ^ swz(row)) << 4) : XMAIN + wave * (XR * 128) + lane * 16);
#pragma unroll
            for (int d = 1; d < 4; d++)
                if (2 * d >= vw) *(unsigned*)(g + 4 * d) = 0u;
        }
    };

    // ---- fragment read offsets inside a buffer (swizzled)
    // X16: index 0 / 1 = the 16-row block (ob2 for dy, ib2 for x); lane = (row c16, granule G = 4 th + (lane >> 4)) -- rows 16 apart share
    // the swizzle, so block 1 is block 0 + 16 rows (the tail granule of x: + 2 channel octets)
    const int c16 = lane & 15, g4 = lane >> 4;
    const int rowA = wo * 32 + (X16 ? c16 : r32), rowB = wi * 32 + (X16 ? c16 : r32);
    const int fA = swz(rowA), fB = swz(rowB);
    unsigned a_off[2], x0_off[2], x1_off[2][XR];
#pragma unroll
    for (int kq = 0; kq < 2; kq++) {
        // 32x32x16: 16-pixel groups interleaved over the sibling waves: th 0: 0, 2; th 1: 1, 3
        const int g = X16 ? 4 * th + g4 : kq * 4 + th * 2 + h;
        const int ra = X16 ? rowA + 16 * kq : rowA, rb = X16 ? rowB + 16 * kq : rowB;
        a_off[kq] = ra * ROWB + ((g ^ fA) << 4);
        x0_off[kq] = DY_BYTES + rb * ROWB + ((g ^ fB) << 4);
#pragma unroll
        for (int xr = 0; xr < XR; xr++)
            x1_off[kq][xr] = (g + 1 < 8) ? DY_BYTES + xr * (64 * ROWB) + rb * ROWB + (((g + 1) ^ fB) << 4)
                                         : DY_BYTES + XMAIN + (rb >> 3) * (XR * 128) + xr * 128 + (rb & 7) * 16;
    }

#ifdef AFCM_WGRAD_PRIO           // A/B builds: static priority for one half of the workgroup's waves (MI355X_MICROARCH.md, two waves per SIMD, item 4): 1 = waves 4-7, 2 = waves 0-3
    if ((AFCM_WGRAD_PRIO == 1) == (wave >= 4)) __builtin_amdgcn_s_setprio(1);
#endif
    // ---- pipeline: NBUF-1 steps of loads in flight; a step's loads are waited for (counted vmcnt) before the barrier that
    // precedes its use.
#pragma unroll
    for (int i = 0; i < NBUF - 1; i++) {
        begin_loads(s0 + i < s1);
        issue_range(std::integral_constant<int, 0>{}, std::integral_constant<int, NPIECE>{});
    }
    asm volatile("s_waitcnt vmcnt(0)");
    if (NBUF == 3) {                       // both prologue steps have landed: fix both edges
        fix_edge();
        { const int q = f_q0, l = f_live; const unsigned b = f_bufa; f_q0 = c_q0; f_live = c_live; f_bufa = c_bufa; fix_edge(); f_q0 = q; f_live = l; f_bufa = b; }
    } else {
        f_q0 = c_q0; f_live = c_live; f_bufa = c_bufa;
        fix_edge();
    }
    __syncthreads();
    int cbuf = 0;
    for (int step = s0; step < s1; step++) {
        // LATE (16x16x32, 3x3): the live waves run begin_loads' ~50 scalar / vector instructions after their first iteration's MFMAs
        // instead of between the barrier and the first MFMA of all eight waves at once (their pieces then go out in iterations 1 .. 7)
        constexpr bool LATE = X16 && TAIL && AFCM_WGRAD_LATE;
        if constexpr (!LATE) begin_loads(step + NBUF - 1 < s1);                 // into the buffer everyone left at the last barrier
        const char* buf = lds + cbuf * BUF;
        typedef typename std::conditional<std::is_same<T, bf16_t>::value, bf16x8, f16x8>::type frag_t;
        // 16-pixel groups of the chunk that lie beyond the row's end would multiply zeros (rows of 86, 150, 278 pixels end
        // with 22 pixels of a 64-pixel chunk): a wave skips its dead groups and only issues its share of the next loads.  The
        // groups alternate between the two waves that share a SIMD (th 0 / th 1), so a 22-pixel chunk costs both one group.
        const int vq = quad_dead ? 0 : p.Q - u_qc * kWgKQ;
        if (++u_qc == p.qchunks) u_qc = 0;
        // one x row of one 16-pixel group: the next loads' share, the three shifted B fragments, 3 or 6 MFMAs
        auto row_mfmas = [&](auto kqc, auto xrc, const uint4 lo, const uint4 hi, const frag_t* a) __attribute__((always_inline)) {
            constexpr int kq = decltype(kqc)::value, xr = decltype(xrc)::value;
            constexpr int it = kq * XR + xr, NIT = 2 * XR;
            issue_range(std::integral_constant<int, (it * NPIECE) / NIT>{}, std::integral_constant<int, ((it + 1) * NPIECE) / NIT>{});
            asm volatile("" : : "v"(lo.x), "v"(lo.y), "v"(lo.z));               // keep the read a full (conflict-free) b128
            const unsigned d[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};               // pixels 8g+6 .. 8g+15 of the staged row
#pragma unroll
            for (int sft = 0; sft < KS; sft++) {
                union { unsigned u[4]; frag_t f; } b;
#pragma unroll
                for (int w = 0; w < 4; w++)
                    b.u[w] = (sft == 0) ? d[w] : (sft == 1) ? __builtin_amdgcn_alignbyte(d[w + 1], d[w], 2) : d[w + 1];
#pragma unroll
                for (int rr = 0; rr < R; rr++) {
                    const int r = xr - rr;
                    const int t = r * KS + sft;
                    if (r >= 0 && r < KS) {
                        if constexpr (std::is_same<T, bf16_t>::value)
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr], b.f, acc[t], 0, 0, 0);
                        else
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rr], b.f, acc[t], 0, 0, 0);
                    }
                }
            }
        };
        if constexpr (X16) {
            // one K step of 32 pixels per wave: 8 iterations (x row, 16-channel block of x), each the next loads' share, one window
            // (read one iteration ahead), its KS shifted B fragments and their MFMAs into the (o block, i block) tiles of the taps
            typedef __attribute__((ext_vector_type(4))) float cf32x4;
            auto mma = [&](const frag_t& av, const frag_t& bv, int t, int blk) __attribute__((always_inline)) {
                cf32x4 c = {acc[t][4 * blk + 0], acc[t][4 * blk + 1], acc[t][4 * blk + 2], acc[t][4 * blk + 3]};
                if constexpr (std::is_same<T, bf16_t>::value) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, c, 0, 0, 0);
                else c = __builtin_amdgcn_mfma_f32_16x16x32_f16(av, bv, c, 0, 0, 0);
                acc[t][4 * blk + 0] = c[0]; acc[t][4 * blk + 1] = c[1]; acc[t][4 * blk + 2] = c[2]; acc[t][4 * blk + 3] = c[3];
            };
            if (th * 32 >= vq) {                       // this wave's 32 pixels lie beyond the row's end: only its share of the next loads
                if constexpr (LATE) begin_loads(step + NBUF - 1 < s1);
                issue_range(std::integral_constant<int, 0>{}, std::integral_constant<int, NPIECE>{});
            } else if constexpr (!TAIL) {
                frag_t a[2][R];
#pragma unroll
                for (int ob2 = 0; ob2 < 2; ob2++)
#pragma unroll
                    for (int rr = 0; rr < R; rr++) a[ob2][rr] = *(const frag_t*)(buf + a_off[ob2] + rr * (64 * ROWB));
                static_for<0, 2 * XR>([&](auto itc) __attribute__((always_inline)) {
                    constexpr int it = decltype(itc)::value, xr = it >> 1, ib2 = it & 1, NIT = 2 * XR;
                    issue_range(std::integral_constant<int, (it * NPIECE) / NIT>{}, std::integral_constant<int, ((it + 1) * NPIECE) / NIT>{});
                    const frag_t bv = *(const frag_t*)(buf + x0_off[ib2] + xr * (64 * ROWB));
#pragma unroll
                    for (int ob2 = 0; ob2 < 2; ob2++) mma(a[ob2][xr], bv, 0, 2 * ob2 + ib2);
                });
            } else {
                frag_t a[2][R];
#pragma unroll
                for (int ob2 = 0; ob2 < 2; ob2++)
#pragma unroll
                    for (int rr = 0; rr < R; rr++) a[ob2][rr] = *(const frag_t*)(buf + a_off[ob2] + rr * (64 * ROWB));
                uint4 lo_n = *(const uint4*)(buf + x0_off[0]);
                uint4 hi_n = *(const uint4*)(buf + x1_off[0][0]);
                static_for<0, 2 * XR>([&](auto itc) __attribute__((always_inline)) {
                    constexpr int it = decltype(itc)::value, xr = it >> 1, ib2 = it & 1, NIT = 2 * XR;
                    const uint4 lo = lo_n, hi = hi_n;
                    if constexpr (it + 1 < NIT) {
                        constexpr int xr1 = (it + 1) >> 1, ib1 = (it + 1) & 1;
                        lo_n = *(const uint4*)(buf + x0_off[ib1] + xr1 * (64 * ROWB));
                        hi_n = *(const uint4*)(buf + x1_off[ib1][xr1]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (LATE) {
                        static_assert(!LATE || NPIECE == NIT - 1, "one piece per iteration after the first");
                        if constexpr (it > 0) issue_piece(std::integral_constant<int, it - 1>{});
                    } else
                        issue_range(std::integral_constant<int, (it * NPIECE) / NIT>{}, std::integral_constant<int, ((it + 1) * NPIECE) / NIT>{});
                    asm volatile("" : : "v"(lo.x), "v"(lo.y), "v"(lo.z));               // keep the read a full (conflict-free) b128
                    const unsigned d[5] = {lo.w, hi.x, hi.y, hi.z, hi.w};               // pixels 8G+6 .. 8G+15 of the staged row
#pragma unroll
                    for (int sft = 0; sft < KS; sft++) {
                        union { unsigned u[4]; frag_t f; } bw;
#pragma unroll
                        for (int w = 0; w < 4; w++)
#ifdef AFCM_WGRAD_EXPERIMENT_NOSHIFT    // timing experiment only (wrong results): 1 = no funnel shifts for the middle tap column, 2 = no register copies for the first either
                            bw.u[w] = (sft == 0 && AFCM_WGRAD_EXPERIMENT_NOSHIFT < 2) ? d[w] : d[w + 1];
#else
                            bw.u[w] = (sft == 0) ? d[w] : (sft == 1) ? __builtin_amdgcn_alignbyte(d[w + 1], d[w], 2) : d[w + 1];
#endif
#pragma unroll
                        for (int rr = 0; rr < R; rr++) {
                            const int r = xr - rr;
                            if (r >= 0 && r < KS) {
#pragma unroll
                                for (int ob2 = 0; ob2 < 2; ob2++) mma(a[ob2][rr], bw.f, r * KS + sft, 2 * ob2 + ib2);
                            }
                        }
                    }
                    if constexpr (LATE && it == 0) begin_loads(step + NBUF - 1 < s1);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
        } else if constexpr (!TAIL) {
            static_for<0, 2>([&](auto kqc) __attribute__((always_inline)) {
                constexpr int kq = decltype(kqc)::value;
                if ((kq * 2 + th) * 16 >= vq) {
                    issue_range(std::integral_constant<int, (kq * XR * NPIECE) / (2 * XR)>{}, std::integral_constant<int, ((kq + 1) * XR * NPIECE) / (2 * XR)>{});
                    return;
                }
                frag_t a[R];
#pragma unroll
                for (int rr = 0; rr < R; rr++) a[rr] = *(const frag_t*)(buf + a_off[kq] + rr * (64 * ROWB));
                static_for<0, XR>([&](auto xrc) __attribute__((always_inline)) {
                    constexpr int xr = decltype(xrc)::value;
                    constexpr int it = kq * XR + xr, NIT = 2 * XR;
                    issue_range(std::integral_constant<int, (it * NPIECE) / NIT>{}, std::integral_constant<int, ((it + 1) * NPIECE) / NIT>{});
                    const frag_t b = *(const frag_t*)(buf + x0_off[kq] + xr * (64 * ROWB));
                    constexpr int rr = xr;
                    if constexpr (std::is_same<T, bf16_t>::value) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[rr], b, acc[0], 0, 0, 0);
                    else acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[rr], b, acc[0], 0, 0, 0);
                });
            });
        } else {
            // The x windows are read one row ahead of their MFMAs, also across the two groups (left to itself the scheduler emits
            // read, wait, multiply for every row: eight exposed LDS round trips per step with only the sibling wave to cover
            // them).  The reads ahead are unconditional -- a dead group's window is simply not used.
            frag_t a[2][R];
#pragma unroll
            for (int rr = 0; rr < R; rr++) a[0][rr] = *(const frag_t*)(buf + a_off[0] + rr * (64 * ROWB));
            uint4 lo_n = *(const uint4*)(buf + x0_off[0]);
            uint4 hi_n = *(const uint4*)(buf + x1_off[0][0]);
            static_for<0, 2>([&](auto kqc) __attribute__((always_inline)) {
                constexpr int kq = decltype(kqc)::value;
                if ((kq * 2 + th) * 16 >= vq) {
                    issue_range(std::integral_constant<int, (kq * XR * NPIECE) / (2 * XR)>{}, std::integral_constant<int, ((kq + 1) * XR * NPIECE) / (2 * XR)>{});
                    if constexpr (kq == 0) {
#pragma unroll
                        for (int rr = 0; rr < R; rr++) a[1][rr] = *(const frag_t*)(buf + a_off[1] + rr * (64 * ROWB));
                        lo_n = *(const uint4*)(buf + x0_off[1]);
                        hi_n = *(const uint4*)(buf + x1_off[1][0]);
                    }
                    return;
                }
                static_for<0, XR>([&](auto xrc) __attribute__((always_inline)) {
                    constexpr int xr = decltype(xrc)::value;
                    const uint4 lo = lo_n, hi = hi_n;
                    if constexpr (xr + 1 < XR) {
                        lo_n = *(const uint4*)(buf + x0_off[kq] + (xr + 1) * (64 * ROWB));
                        hi_n = *(const uint4*)(buf + x1_off[kq][xr + 1 < XR ? xr + 1 : xr]);
                    } else if constexpr (kq == 0) {
                        lo_n = *(const uint4*)(buf + x0_off[1]);
                        hi_n = *(const uint4*)(buf + x1_off[1][0]);
                    }
                    if constexpr (kq == 0 && xr == 1) {
#pragma unroll
                        for (int rr = 0; rr < R; rr++) a[1][rr] = *(const frag_t*)(buf + a_off[1] + rr * (64 * ROWB));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    row_mfmas(kqc, xrc, lo, hi, a[kq]);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
        }
        // the next step's loads (issued NBUF-2 iterations ago, or just now when NBUF == 2) must have landed; patch its edge
        if (NBUF == 3) asm volatile("s_waitcnt vmcnt(%0)" : : "n"(NPIECE));
        else { asm volatile("s_waitcnt vmcnt(0)"); f_q0 = c_q0; f_live = c_live; f_bufa = c_bufa; }
        fix_edge();
        __syncthreads();
        if (++cbuf == NBUF) cbuf = 0;
    }
    // ---- add the two pixel halves through LDS (the ring is free now): th 1 parks its accumulators, th 0 adds them and
    // writes the partial tile D[row = o][col = i].
    asm volatile("s_waitcnt vmcnt(0)");
    __syncthreads();
    {
        float* red = (float*)lds;
        constexpr int TPR_CAP = (NBUF * BUF) / (4 * 16 * 64 * (int)sizeof(float));     // taps per round
        constexpr int TPR = TPR_CAP < KK ? TPR_CAP : KK;
        static_assert(TPR >= 1, "LDS too small for the half-sum");
        const int wv4 = wo + 2 * wi;                                     // the quadrant: both pixel halves of it meet in the same slot
#pragma unroll
        for (int t0 = 0; t0 < KK; t0 += TPR) {
            if (t0 > 0) __syncthreads();
            if (th == 1) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < KK; t++)
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) red[((wv4 * TPR + (t - t0)) * 16 + reg) * 64 + lane] = acc[t][reg];
            }
            __syncthreads();
            if (th == 0) {
#pragma unroll
                for (int t = t0; t < t0 + TPR && t < KK; t++)
#pragma unroll
                    for (int reg = 0; reg < 16; reg++) acc[t][reg] += red[((wv4 * TPR + (t - t0)) * 16 + reg) * 64 + lane];
            }
        }
    }
    if (th == 0) {
        float* out = p.part + (size_t)split * p.O * p.I * KK;
#pragma unroll
        for (int reg = 0; reg < 16; reg++) {
            // 16x16x32: element 4 (2 ob2 + ib2) + r of the (ob2, ib2) tile = row 16 ob2 + 4 (lane >> 4) + r, column 16 ib2 + (lane & 15)
            const int o = o0 + wo * 32 + (X16 ? 16 * (reg >> 3) + 4 * g4 + (reg & 3) : (reg & 3) + 8 * (reg >> 2) + 4 * h);
            const int i = i0 + wi * 32 + (X16 ? 16 * ((reg >> 2) & 1) + c16 : r32);
            if (o < p.O && i < p.I) {
                float* dst = out + ((size_t)o * p.I + i) * KK;
#pragma unroll
                for (int t = 0; t < KK; t++) dst[t] = acc[t][reg];
            }
        }
    }
}

// Slabs an element's tile wrote (WgradParams: full tiles `splits`, partial ones `splits_p`; the slabs beyond were never written)
struct WgradSlabs { int I, KK, fo, fi, splits, splits_p; };
__device__ __forceinline__ int slabs_of(const WgradSlabs& w, long long idx) {
    if (w.splits_p == w.splits) return w.splits;
    const int oi = (int)(idx / w.KK), o = oi / w.I, i = oi - o * w.I;
    return ((o >> 6) < w.fo && (i >> 6) < w.fi) ? w.splits : w.splits_p;
}
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ dw, const float* __restrict__ part, long long numel, WgradSlabs w) {
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < numel; idx += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        const int splits = slabs_of(w, idx);
        for (int k = 0; k < splits; k++) s += part[(size_t)k * numel + idx];
        dw[idx] = s;
    }
}

// Same reduction for numel % 4 == 0 with 16-byte loads and the split index spread over the workgroup: 256 threads =
// COLS float4 columns x (256 / COLS) split groups, group partials summed through LDS.  A 64 -> 64 layer has 36,864 outputs and 256
// splits (151 MB of partials): one thread per output is 144 workgroups of serial 4-byte loads on a 256-CU chip.
template <int COLS>
__global__ __launch_bounds__(256) void wgrad_reduce4_kernel(float* __restrict__ dw, const float* __restrict__ part, long long numel4, WgradSlabs w) {
    constexpr int GROUPS = 256 / COLS;
    typedef __attribute__((ext_vector_type(4))) float f32x4v;
    __shared__ f32x4v red[GROUPS][COLS];
    const int col = threadIdx.x % COLS, grp = threadIdx.x / COLS;
    const long long c4 = (long long)blockIdx.x * COLS + col;
    f32x4v s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
    if (c4 < numel4) {
        const f32x4v* src = (const f32x4v*)part + c4;
        // the four elements of a column may belong to tiles of different classes (9 taps per (o, i): columns straddle i and tile borders):
        // common slabs as whole vectors, the rest element by element
        int ne[4];
#pragma unroll
        for (int e = 0; e < 4; e++) ne[e] = slabs_of(w, 4 * c4 + e);
        const int splits = min(min(ne[0], ne[1]), min(ne[2], ne[3]));
        const int most = max(max(ne[0], ne[1]), max(ne[2], ne[3]));
        for (int k2 = splits + grp; k2 < most; k2 += GROUPS) {
            const f32x4v v = src[(size_t)k2 * numel4];
#pragma unroll
            for (int e = 0; e < 4; e++) s1[e] += k2 < ne[e] ? v[e] : 0.f;
        }
        int k = grp;
        for (; k + 3 * GROUPS < splits; k += 4 * GROUPS) {
            const f32x4v v0 = src[(size_t)k * numel4], v1 = src[(size_t)(k + GROUPS) * numel4];
            const f32x4v v2 = src[(size_t)(k + 2 * GROUPS) * numel4], v3 = src[(size_t)(k + 3 * GROUPS) * numel4];
            s0 += v0; s1 += v1; s0 += v2; s1 += v3;
        }
        for (; k < splits; k += GROUPS) s0 += src[(size_t)k * numel4];
    }
    s0 += s1;
    if (GROUPS > 1) {
        red[grp][col] = s0;
        __syncthreads();
        if (grp == 0 && c4 < numel4) {
#pragma unroll
            for (int g = 1; g < GROUPS; g++) s0 += red[g][col];
            ((f32x4v*)dw)[c4] = s0;
        }
    } else if (c4 < numel4) {
        ((f32x4v*)dw)[c4] = s0;
    }
}


// Slab reduction of an IMAGE-ALIGNED weight gradient (WgradParams::splits_img) that also returns, per image n and input channel i,
//     dots[n][i] = sum_{o, tap} wq[o][i][tap] * dW_n[o][i][tap]          dW_n = the sum of image n's slabs, wq = w rounded to the conv's 16-bit type
// = <x[n, i], dx[n, i]> with dx = conv^T(wq, dy): the contraction <dy_n, conv(wq[:, i], x[n, i])> written from the weight side instead of
// the pixel side.  For the layer below this is <g, z> -- the gradient of the styles its epilogue multiplied z by -- which r01-r05 read from
// g and z themselves: a full pass over both tensors (312-624 MB per 276^2 layer, 50-118 us) for numbers that these slabs already hold
// (27-38 MB, read here anyway).  Differs from the pixel-side dot product only by the 16-bit rounding of the STORED dx.
// One workgroup (16 waves) per input channel i; wave q takes images q, q + 16, ...; lane = output row of a block of 64.
template <typename T, int KK>
__global__ __launch_bounds__(1024) void wgrad_reduce_dots_kernel(float* __restrict__ dw, float* __restrict__ dots, const float* __restrict__ part,
                                                                 const float* __restrict__ w, int N, int O, int I, int splits_img) {
    // 16 waves: wave q takes images q, q + 16, ... (one each at batch 16); lane = output row of a block of 64.  (r06, first form: 4 waves, four
    // images each in turn -- 58 us for the 64 -> 64 layer's 38 MB of slabs where the plain reduction took 7.)
    constexpr int NW = 16;
    __shared__ float red[NW - 1][64][KK];
    const int i = blockIdx.x, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const size_t slab = (size_t)O * I * KK;
    float dotp[4];
#pragma unroll
    for (int k = 0; k < 4; k++) dotp[k] = 0.f;
    for (int ob = 0; ob < O; ob += 64) {
        const int o = ob + lane;
        const bool live = o < O;
        const size_t e0 = ((size_t)(live ? o : O - 1) * I + i) * KK;
        float wq[KK], tot[KK];
#pragma unroll
        for (int t = 0; t < KK; t++) { wq[t] = live ? to_f32(from_f32<T>(w[e0 + t])) : 0.f; tot[t] = 0.f; }
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int n = q + NW * k;
            if (n >= N) break;                                  // (wave-uniform)
            float acc[KK];
#pragma unroll
            for (int t = 0; t < KK; t++) acc[t] = 0.f;
            for (int sp = 0; sp < splits_img; sp++) {
                const float* src = part + (size_t)(n * splits_img + sp) * slab + e0;
#pragma unroll
                for (int t = 0; t < KK; t++) acc[t] += src[t];
            }
            float d = 0.f;
#pragma unroll
            for (int t = 0; t < KK; t++) { tot[t] += acc[t]; d = fmaf(wq[t], acc[t], d); }
            dotp[k] += d;
        }
        if (ob > 0) __syncthreads();                            // (the previous block's partials have been read)
        if (q > 0) {
#pragma unroll
            for (int t = 0; t < KK; t++) red[q - 1][lane][t] = tot[t];
        }
        __syncthreads();
        if (q == 0 && live) {
#pragma unroll
            for (int t = 0; t < KK; t++) {
                float s = tot[t];
#pragma unroll
                for (int k = 0; k < NW - 1; k++) s += red[k][lane][t];
                dw[e0 + t] = s;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int n = q + NW * k;
        if (n >= N) break;
        float d = dotp[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) d += __shfl_xor(d, off);
        if (lane == 0) dots[(size_t)n * I + i] = d;
    }
}

static void choose_tile(int P, int Q, int KS, int* TH, int* TW, int* PWL, int patch_max = kPatchMax) {
    // Tile of TH x TW output pixels with TH*TW <= 256 slots and an LDS patch (TH+KS-1) x round4(TW+KS) <= kPatchMax,
    // chosen to maximise the fraction of useful slots.
    double best = -1;
    constexpr double gran_bonus = 0.03;
    for (int tw = 2; tw <= 128; tw += 2) {
        int th = kSlots / tw;
        if (th > P) th = P;
        for (; th >= 1; th--) {
            const int pwl = round_up(tw + KS, 4);
            if ((th + KS - 1) * pwl > patch_max) continue;
            const double tiles = (double)cdiv(P, th) * cdiv(Q, tw);
            const double util = (double)P * Q / (tiles * kSlots);
            // small preference for wide tiles (longer contiguous runs for loads/stores); rows of whole 8-pixel granules
            // get the 16-byte LDS-transposed epilogue of the 16-bit kernel: worth a few % of slot utilisation
            const double score = util + 1e-4 * tw + ((tw & 7) == 0 ? gran_bonus : 0.0);
            if (score > best) { best = score; *TH = th; *TW = tw; *PWL = pwl; }
            break;
        }
    }
}

static void choose_tile_s2(int P, int Q, int* TH, int* TW, int* PWL) {
    // stride-2 kernel: TH x TW output pixels with TH * TW <= 128 slots under a ((TH - 1) 2 + 3) x round4((TW - 1) 2 + 4) patch <= kPatchMaxS2
    double best = -1;
    for (int tw = 2; tw <= 64; tw += 2) {
        int th = 128 / tw;
        if (th > P) th = P;
        for (; th >= 1; th--) {
            const int pwl = round_up((tw - 1) * 2 + 4, 4);
            if (((th - 1) * 2 + 3) * pwl > kPatchMaxS2) continue;
            const double tiles = (double)cdiv(P, th) * cdiv(Q, tw);
            const double util = (double)P * Q / (tiles * 128);
            const double score = util + 1e-4 * tw + ((tw & 7) == 0 ? 0.03 : 0.0);
            if (score > best) { best = score; *TH = th; *TW = tw; *PWL = pwl; }
            break;
        }
    }
}

// Grid of the persistent conv2d_fwd16x_kernel: one round of resident workgroups (compute units x workgroups per CU, a multiple of 8 so that
// the XCD-aware item order is the same function of the item as of the hardware block index), or every item when there are fewer.
// The compute-unit count is a property of the device, read once per device (speed only: any grid size computes the same result).
static int conv_persistent_grid(long long items, int per_cu) {
    static int cus[16] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int n = (dev >= 0 && dev < 16) ? cus[dev] : 0;
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        if (dev >= 0 && dev < 16) cus[dev] = n;
    }
#ifdef AFCM_CONV_AB
    if (g_conv_x16 >> 4) per_cu = g_conv_x16 >> 4;         // experiment: workgroups per CU of the persistent grid from the debug switch
#endif
    const long long slots = (long long)round_up(n * per_cu, 8);
    return (int)(items < slots ? items : slots);
}

// conv2d_fwd16x_kernel<.., FASTEPI>: tile widths that are multiples of 16 on output rows whose pitch is a multiple of 8 elements
#ifndef AFCM_CONV_FASTEPI
#define AFCM_CONV_FASTEPI 1          // (0: the general epilogue everywhere; A/B builds)
#endif
static bool conv_fast_epilogue(const ConvParams& p) {
    return AFCM_CONV_FASTEPI && (p.TW & 15) == 0 && (p.ldy & 7) == 0 && (p.Q & 1) == 0;
}

// o_base / row_blocks: the launch covers output rows [o_base, o_base + row_blocks * BM_O) (16-bit 3x3 16x16x32 kernel only; 0: all rows)
template <typename T, int BM_O>
static int launch_conv(ConvParams p, int ks, hipStream_t st, int o_base = 0, int row_blocks = 0) {
    const long long blocks = (long long)p.tilesX * p.tilesY * p.N * (row_blocks ? row_blocks : cdiv(p.Cout, BM_O));
    p.o_base = o_base;
    AFCM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv2d: grid of %lld blocks is out of range", blocks);
    dim3 grid((unsigned)blocks), block(256);
    p.total_blocks = (int)blocks;
    if constexpr (sizeof(T) == 2) {
        if (ks == 3) {
#if defined(AFCM_CONV_AB) || AFCM_CONV_X16
            if (AFCM_X16_ON) {
                // (the 64-row kernel is persistent: one round of three workgroups per CU; the 128-row kernel takes one item per workgroup)
                const dim3 g16 = BM_O == 64 ? dim3((unsigned)conv_persistent_grid(blocks, 3)) : grid;
                if (conv_fast_epilogue(p)) hipLaunchKernelGGL((conv2d_fwd16x_kernel<T, BM_O, false, true>), g16, block, 0, st, p);
                else hipLaunchKernelGGL((conv2d_fwd16x_kernel<T, BM_O>), g16, block, 0, st, p);
                return hip_status(hipGetLastError());
            }
#endif
#if defined(AFCM_CONV_AB) || !AFCM_CONV_X16
            hipLaunchKernelGGL((conv2d_fwd16_kernel<T, BM_O>), grid, block, 0, st, p);
#endif
            return hip_status(hipGetLastError());
        }
    }
    if (ks == 3) hipLaunchKernelGGL((conv2d_fwd_kernel<T, BM_O, 3>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv2d_fwd_kernel<T, BM_O, 1>), grid, block, 0, st, p);
    return hip_status(hipGetLastError());
}

}  // namespace afcm

using namespace afcm;

#ifdef AFCM_CONV_STAMPS
extern "C" int afcm_debug_conv_stamps(void* dst, int n_blocks) {
    return hip_status(hipMemcpyFromSymbol(dst, HIP_SYMBOL(afcm_conv_stamps_buf), (size_t)n_blocks * 32, 0, hipMemcpyDeviceToHost));
}
extern "C" int afcm_debug_conv_realtime(void* dst, int n_blocks) {
    return hip_status(hipMemcpyFromSymbol(dst, HIP_SYMBOL(afcm_conv_rt_buf), (size_t)n_blocks * 32, 0, hipMemcpyDeviceToHost));
}
extern "C" int afcm_debug_conv_prologue(void* dst, int n_blocks) {
    return hip_status(hipMemcpyFromSymbol(dst, HIP_SYMBOL(afcm_conv_pro_buf), (size_t)n_blocks * 32, 0, hipMemcpyDeviceToHost));
}
extern "C" int afcm_debug_conv_barrier_cycles(void* dst, int n_blocks) {
    return hip_status(hipMemcpyFromSymbol(dst, HIP_SYMBOL(afcm_conv_bar_buf), (size_t)n_blocks * 32, 0, hipMemcpyDeviceToHost));
}
extern "C" int afcm_debug_conv_stamps_clear() {
    void* a; void* b;
    if (hipGetSymbolAddress(&a, HIP_SYMBOL(afcm_conv_stamps_buf)) != hipSuccess || hipGetSymbolAddress(&b, HIP_SYMBOL(afcm_conv_bar_buf)) != hipSuccess) return AFCM_E_INVALID;
    (void)hipMemset(a, 0, sizeof(afcm_conv_stamps_buf));
    void* c;
    if (hipGetSymbolAddress(&c, HIP_SYMBOL(afcm_conv_pro_buf)) == hipSuccess) (void)hipMemset(c, 0, sizeof(afcm_conv_pro_buf));
    return hip_status(hipMemset(b, 0, sizeof(afcm_conv_bar_buf)));
}
#endif
extern "C" int afcm_conv2d_block_k(int32_t dtype) { return dtype == AFCM_F32 ? ConvCfg<float>::BK : ConvCfg<bf16_t>::BK; }
static inline int afcm::conv_bk(int dtype, int ks) { return (dtype != AFCM_F32 && ks == 3 && AFCM_X16_ON) ? 32 : afcm_conv2d_block_k(dtype); }
extern "C" int afcm_conv2d_block_k_ks(int32_t dtype, int32_t ks) { return conv_bk(dtype, ks); }
#ifdef AFCM_CONV_AB
extern "C" int afcm_debug_conv_variant(int x16) { g_conv_x16 = x16; return AFCM_OK; }
#endif

template <typename T>
static void launch_pack8(void* dst0, void* dst1, const float* w, int cout, int cin, int ks, int rows_pad0, int rows_pad1, int BK, hipStream_t st) {
    // tiles cover the padded row ranges of both images: o up to rows_pad0 (forward rows) and the last started K-chunk of the
    // data-gradient image, i up to rows_pad1 and the forward image's last K-chunk (rows_pad are multiples of 64 >= the extents)
    const int omax = dst0 ? rows_pad0 : round_up(cout, BK), imax = dst1 ? rows_pad1 : round_up(cin, BK);
    constexpr int BKT = ConvCfg<T>::BK;
    const int ti = BK > 16 ? PackTile<32>::TI : PackTile<BKT>::TI, to = BK > 16 ? PackTile<32>::TO : PackTile<BKT>::TO;
    dim3 grid((unsigned)cdiv(imax > cin ? imax : cin, ti), (unsigned)cdiv(omax > cout ? omax : cout, to)), block(256);
    if constexpr (sizeof(T) == 2) {
        if (BK == 32) {           // the 16x16x32 kernel's image (3x3 only)
            hipLaunchKernelGGL((conv2d_pack_tile_kernel<T, 9, 32>), grid, block, 0, st, (T*)dst0, (T*)dst1, w, cout, cin, rows_pad0, rows_pad1);
            return;
        }
    }
    if (ks == 3) hipLaunchKernelGGL((conv2d_pack_tile_kernel<T, 9, BKT>), grid, block, 0, st, (T*)dst0, (T*)dst1, w, cout, cin, rows_pad0, rows_pad1);
    else hipLaunchKernelGGL((conv2d_pack_tile_kernel<T, 1, BKT>), grid, block, 0, st, (T*)dst0, (T*)dst1, w, cout, cin, rows_pad0, rows_pad1);
}

static int pack_weights2_bk(void* dst_fwd, void* dst_dgrad, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks,
                            int32_t rows_pad_fwd, int32_t rows_pad_dgrad, int BK, void* stream);
extern "C" int afcm_conv2d_pack_weights2(void* dst_fwd, void* dst_dgrad, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks,
                                         int32_t rows_pad_fwd, int32_t rows_pad_dgrad, void* stream) {
    return pack_weights2_bk(dst_fwd, dst_dgrad, w, dtype, cout, cin, ks, rows_pad_fwd, rows_pad_dgrad, conv_bk(dtype, ks), stream);
}
extern "C" int afcm_conv2d_pack_weights_bk(void* dst, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks, int32_t mode,
                                           int32_t rows_pad, int32_t block_k, void* stream) {
    AFCM_REQUIRE(dst != nullptr && w != nullptr, "conv2d_pack_weights: null pointer");
    AFCM_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (forward) or 1 (data gradient)");
    AFCM_REQUIRE(block_k == afcm_conv2d_block_k(dtype) || block_k == conv_bk(dtype, ks), "conv2d_pack_weights_bk: K-chunk %d is not one of this dtype's", block_k);
    return mode == 0 ? pack_weights2_bk(dst, nullptr, w, dtype, cout, cin, ks, rows_pad, 0, block_k, stream)
                     : pack_weights2_bk(nullptr, dst, w, dtype, cout, cin, ks, 0, rows_pad, block_k, stream);
}
static int pack_weights2_bk(void* dst_fwd, void* dst_dgrad, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks,
                            int32_t rows_pad_fwd, int32_t rows_pad_dgrad, int BK, void* stream) {
    AFCM_REQUIRE(w != nullptr && (dst_fwd != nullptr || dst_dgrad != nullptr), "conv2d_pack_weights: null pointer");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "dtype must be float32, float16 or bfloat16");
    AFCM_REQUIRE(ks == 1 || ks == 3, "only 1x1 and 3x3 kernels are supported");
    AFCM_REQUIRE(cout > 0 && cin > 0, "conv2d_pack_weights: empty weights");
    AFCM_REQUIRE(dst_fwd == nullptr || (rows_pad_fwd >= cout && rows_pad_fwd % 64 == 0), "rows_pad must be a multiple of 64 covering the rows");
    AFCM_REQUIRE(dst_dgrad == nullptr || (rows_pad_dgrad >= cin && rows_pad_dgrad % 64 == 0), "rows_pad must be a multiple of 64 covering the rows");
    AFCM_REQUIRE((((uintptr_t)dst_fwd | (uintptr_t)dst_dgrad) & 31) == 0, "conv2d_pack_weights: destinations must be 32-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case AFCM_F32: launch_pack8<float>(dst_fwd, dst_dgrad, w, cout, cin, ks, rows_pad_fwd, rows_pad_dgrad, BK, st); break;
        case AFCM_F16: launch_pack8<f16_t>(dst_fwd, dst_dgrad, w, cout, cin, ks, rows_pad_fwd, rows_pad_dgrad, BK, st); break;
        default: launch_pack8<bf16_t>(dst_fwd, dst_dgrad, w, cout, cin, ks, rows_pad_fwd, rows_pad_dgrad, BK, st); break;
    }
    return hip_status(hipGetLastError());
}

extern "C" int afcm_conv2d_pack_weights(void* dst, const float* w, int32_t dtype, int32_t cout, int32_t cin, int32_t ks,
                                        int32_t mode, int32_t rows_pad, void* stream) {
    AFCM_REQUIRE(dst != nullptr && w != nullptr, "conv2d_pack_weights: null pointer");
    AFCM_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (forward) or 1 (data gradient)");
    return mode == 0 ? afcm_conv2d_pack_weights2(dst, nullptr, w, dtype, cout, cin, ks, rows_pad, 0, stream)
                     : afcm_conv2d_pack_weights2(nullptr, dst, w, dtype, cout, cin, ks, 0, rows_pad, stream);
}

template <typename T>
static void launch_pack_bank(const PackBank& b, int blocks, int ks, int BK, hipStream_t st) {
    constexpr int BKT = ConvCfg<T>::BK;
    if constexpr (sizeof(T) == 2) {
        if (BK == 32) {
            hipLaunchKernelGGL((conv2d_pack_bank_kernel<T, 9, 32>), dim3(blocks), dim3(256), 0, st, b);
            return;
        }
    }
    if (ks == 3) hipLaunchKernelGGL((conv2d_pack_bank_kernel<T, 9, BKT>), dim3(blocks), dim3(256), 0, st, b);
    else hipLaunchKernelGGL((conv2d_pack_bank_kernel<T, 1, BKT>), dim3(blocks), dim3(256), 0, st, b);
}

extern "C" int afcm_conv2d_pack_bank(const afcm_pack_entry* entries, int32_t count, int32_t dtype, int32_t ks, void* stream) {
    AFCM_REQUIRE(entries != nullptr && count > 0 && count <= AFCM_PACK_MAX, "conv2d_pack_bank: 1..%d entries", AFCM_PACK_MAX);
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "dtype must be float32, float16 or bfloat16");
    AFCM_REQUIRE(ks == 1 || ks == 3, "only 1x1 and 3x3 kernels are supported");
    const int BK = conv_bk(dtype, ks);
    const int ti = BK > 16 ? PackTile<32>::TI : 64, to = BK > 16 ? PackTile<32>::TO : 16;
    PackBank b;
    b.count = count;
    int tot = 0;
    for (int l = 0; l < count; l++) {
        const afcm_pack_entry& e = entries[l];
        AFCM_REQUIRE(e.w != nullptr && (e.dst_fwd != nullptr || e.dst_dgrad != nullptr) && e.cout > 0 && e.cin > 0, "conv2d_pack_bank: entry %d: null pointer or empty weights", l);
        AFCM_REQUIRE(e.dst_fwd == nullptr || (e.rows_pad_fwd >= e.cout && e.rows_pad_fwd % 64 == 0), "conv2d_pack_bank: entry %d: rows_pad must be a multiple of 64 covering the rows", l);
        AFCM_REQUIRE(e.dst_dgrad == nullptr || (e.rows_pad_dgrad >= e.cin && e.rows_pad_dgrad % 64 == 0), "conv2d_pack_bank: entry %d: rows_pad must be a multiple of 64 covering the rows", l);
        AFCM_REQUIRE((((uintptr_t)e.dst_fwd | (uintptr_t)e.dst_dgrad) & 31) == 0, "conv2d_pack_bank: entry %d: destinations must be 32-byte aligned", l);
        // as launch_pack8: tiles cover the padded row ranges of both images
        const int omax = e.dst_fwd ? e.rows_pad_fwd : round_up(e.cout, BK), imax = e.dst_dgrad ? e.rows_pad_dgrad : round_up(e.cin, BK);
        const int gx = cdiv(imax > e.cin ? imax : e.cin, ti), gy = cdiv(omax > e.cout ? omax : e.cout, to);
        b.e[l] = e;
        b.gx[l] = gx;
        b.blk[l] = tot;
        tot += gx * gy;
    }
    b.blk[count] = tot;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case AFCM_F32: launch_pack_bank<float>(b, tot, ks, BK, st); break;
        case AFCM_F16: launch_pack_bank<f16_t>(b, tot, ks, BK, st); break;
        default: launch_pack_bank<bf16_t>(b, tot, ks, BK, st); break;
    }
    return hip_status(hipGetLastError());
}

extern "C" int afcm_conv2d_stride2(void* y, const void* x, const void* wpacked, int32_t dtype, int32_t n, int32_t cin, int32_t cout, int32_t h,
                                   int32_t w, int32_t pad, int32_t rows_pad, void* stream) {
    AFCM_REQUIRE(y != nullptr && x != nullptr && wpacked != nullptr, "conv2d_stride2: null pointer");
    AFCM_REQUIRE(dtype == AFCM_F16 || dtype == AFCM_BF16, "conv2d_stride2: 16-bit activations only");
    AFCM_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0 && (w % 2) == 0, "conv2d_stride2: empty x or odd width %d", w);
    AFCM_REQUIRE(pad >= 0 && pad <= 2, "padding must be in [0, k-1]");
    AFCM_REQUIRE(rows_pad >= cout && rows_pad % 128 == 0, "conv2d_stride2: rows_pad must be a multiple of 128 covering cout");
    AFCM_REQUIRE(h + 2 * pad >= 3 && w + 2 * pad >= 3, "output must be at least 1x1");
    ConvParams p;
    p.x = x; p.y = y; p.wp = wpacked; p.oscale = nullptr; p.obias = nullptr;
    p.N = n; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w;
    p.P = (h + 2 * pad - 3) / 2 + 1; p.Q = (w + 2 * pad - 3) / 2 + 1;
    p.pad = pad;
    p.ldx = w; p.ldy = p.Q;
    choose_tile_s2(p.P, p.Q, &p.TH, &p.TW, &p.PWL);
    p.tilesX = cdiv(p.Q, p.TW); p.tilesY = cdiv(p.P, p.TH);
    p.magicTW = (unsigned)((0x100000000ull + (unsigned)p.TW - 1) / (unsigned)p.TW);
    p.magicTX = magic_u32((unsigned)p.tilesX); p.magicTY = magic_u32((unsigned)p.tilesY); p.magicN = magic_u32((unsigned)p.N); p.magicPC = magic_u32((unsigned)(p.PWL >> 2));
    p.Opad = rows_pad;
    p.nkc = cdiv(cin, afcm_conv2d_block_k(dtype));
    p.nkc_real = p.nkc; p.magicNK = 0; p.term_parts = 0; p.part_bytes = 0; p.last_part_bytes = 0; p.bound_a = p.bound_b = nullptr; p.total_blocks = 0; p.o_base = 0;
    const long long blocks = (long long)p.tilesX * p.tilesY * n * cdiv(cout, 128);
    AFCM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv2d_stride2: grid of %lld blocks is out of range", blocks);
    AFCM_REQUIRE((long long)cin * h * w * 2ll < (1ll << 31), "conv2d_stride2: image out of range");
    if (dtype == AFCM_F16) hipLaunchKernelGGL((conv2d_fwd16s2_kernel<f16_t, 128>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((conv2d_fwd16s2_kernel<bf16_t, 128>), dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_conv2d(void* y, const void* x, const void* wpacked, const float* oscale, const float* obias, int32_t dtype, int32_t n,
                           int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t rows_pad, void* stream) {
    return afcm_conv2d_ld(y, x, wpacked, oscale, obias, dtype, n, cin, cout, h, w, ks, pad, rows_pad, 0, 0, stream);
}

extern "C" int afcm_conv2d_ld(void* y, const void* x, const void* wpacked, const float* oscale, const float* obias, int32_t dtype, int32_t n,
                              int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t rows_pad, int32_t x_pitch,
                              int32_t y_pitch, void* stream) {
    AFCM_REQUIRE(y != nullptr && x != nullptr && wpacked != nullptr, "conv2d: null pointer");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    AFCM_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "x is empty");
    AFCM_REQUIRE(ks == 1 || ks == 3, "only 1x1 and 3x3 kernels are supported");
    AFCM_REQUIRE(pad >= 0 && pad <= ks - 1, "padding must be in [0, k-1]");
    AFCM_REQUIRE(dtype == AFCM_F32 || (w % 2 == 0), "16-bit conv2d needs an even input width (got %d)", w);
    AFCM_REQUIRE(rows_pad >= cout && rows_pad % 64 == 0, "rows_pad must be a multiple of 64 covering cout");
    ConvParams p;
    p.x = x; p.y = y; p.wp = wpacked; p.oscale = oscale; p.obias = obias;
    p.N = n; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w;
    p.P = h + 2 * pad - ks + 1; p.Q = w + 2 * pad - ks + 1;
    AFCM_REQUIRE(p.P >= 1 && p.Q >= 1, "output must be at least 1x1");
    p.pad = pad;
    p.ldx = x_pitch ? x_pitch : w; p.ldy = y_pitch ? y_pitch : p.Q;
    if (p.ldx != w || p.ldy != p.Q) {
        AFCM_REQUIRE(dtype != AFCM_F32 && ks == 3, "conv2d: row pitches need the 16-bit 3x3 kernel");
        AFCM_REQUIRE(p.ldx >= w && p.ldy >= p.Q && ((p.ldx | p.ldy) & 1) == 0, "conv2d: row pitches %d / %d must be even and cover the widths %d / %d", p.ldx, p.ldy, w, p.Q);
        AFCM_REQUIRE((long long)cout * p.P * p.ldy < (1ll << 30), "conv2d: pitched output image is out of range");
    }
    AFCM_REQUIRE(dtype == AFCM_F32 || ks != 3 || (long long)cout * p.P * p.ldy * 2 < (1ll << 30), "conv2d: 16-bit output image of %lld bytes is out of range (< 2^30)", (long long)cout * p.P * p.ldy * 2);
    if (AFCM_CONV_DIRECT4 && dtype != AFCM_F32 && ks == 3 && cin <= 4 && cout <= 64 && conv_bk(dtype, ks) == 32) {
        // a handful of input channels: the contraction index is (tap column, channel), no channel padding (conv2d_direct.hip; r06)
        return conv2d_direct_small_cin(x, y, wpacked, oscale, obias, dtype, n, cin, cout, h, w, pad, rows_pad, 32, p.ldx, p.ldy, (hipStream_t)stream);
    }
    choose_tile(p.P, p.Q, ks, &p.TH, &p.TW, &p.PWL, (dtype != AFCM_F32 && ks == 3 && AFCM_X16_ON) ? kPatchMaxX16 : kPatchMax);
    p.tilesX = cdiv(p.Q, p.TW); p.tilesY = cdiv(p.P, p.TH);
    p.magicTW = (unsigned)((0x100000000ull + (unsigned)p.TW - 1) / (unsigned)p.TW);
    p.magicTX = magic_u32((unsigned)p.tilesX); p.magicTY = magic_u32((unsigned)p.tilesY); p.magicN = magic_u32((unsigned)p.N); p.magicPC = magic_u32((unsigned)(p.PWL >> 2));
    p.Opad = rows_pad;
    p.nkc = cdiv(cin, conv_bk(dtype, ks));
    p.nkc_real = p.nkc; p.magicNK = 0; p.term_parts = 0; p.part_bytes = 0; p.last_part_bytes = 0; p.bound_a = p.bound_b = nullptr; p.total_blocks = 0; p.o_base = 0;
    hipStream_t st = (hipStream_t)stream;
    // 64-row blocks when they waste fewer padded rows than 128-row blocks
    const bool small = (rows_pad % 128 != 0) || cout <= 64;
    // ... and both when the rows are 128 k + (1 .. 64) (the 181-channel layers: 192 padded rows): the 128-row kernel moves half the
    // pixel-fragment bytes per flop of the 64-row one, so rows [0, 128 k) go to it and only the last 64 to the 64-row kernel -- two
    // launches, disjoint output rows, the same number of passes over x as three 64-row blocks had
    if (AFCM_CONV_BM96 && AFCM_X16_ON && dtype != AFCM_F32 && ks == 3 && ((cout > 64 && cout <= 96) || (AFCM_CONV_BM96_192 && cout > 128 && cout <= 192 && rows_pad >= 192))) {
        // 65 .. 96 output rows (the 91-channel layers): one 96-row block instead of 128 rows of MFMAs for them
        const long long blocks = (long long)p.tilesX * p.tilesY * p.N * cdiv(cout, 96);
        AFCM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv2d: grid of %lld blocks is out of range", blocks);
        p.total_blocks = (int)blocks; p.o_base = 0;
        const dim3 g96((unsigned)(AFCM_CONV_BM96_PERSIST ? conv_persistent_grid(blocks, 2) : blocks));
        if (conv_fast_epilogue(p)) {
            if (dtype == AFCM_F16) hipLaunchKernelGGL((conv2d_fwd16x_kernel<f16_t, 96, false, true>), g96, dim3(256), 0, st, p);
            else hipLaunchKernelGGL((conv2d_fwd16x_kernel<bf16_t, 96, false, true>), g96, dim3(256), 0, st, p);
        } else if (dtype == AFCM_F16) hipLaunchKernelGGL((conv2d_fwd16x_kernel<f16_t, 96>), g96, dim3(256), 0, st, p);
        else hipLaunchKernelGGL((conv2d_fwd16x_kernel<bf16_t, 96>), g96, dim3(256), 0, st, p);
        return hip_status(hipGetLastError());
    }
    if (AFCM_CONV_MIXED && AFCM_X16_ON && dtype != AFCM_F32 && ks == 3 && rows_pad % 128 == 64 && rows_pad > 128) {
        const int big = rows_pad / 128;
        const int rc = dtype == AFCM_F16 ? launch_conv<f16_t, 128>(p, ks, st, 0, big) : launch_conv<bf16_t, 128>(p, ks, st, 0, big);
        if (rc != AFCM_OK) return rc;
        return dtype == AFCM_F16 ? launch_conv<f16_t, 64>(p, ks, st, big * 128, 1) : launch_conv<bf16_t, 64>(p, ks, st, big * 128, 1);
    }
    switch (dtype) {
        case AFCM_F32: return small ? launch_conv<float, 64>(p, ks, st) : launch_conv<float, 128>(p, ks, st);
        case AFCM_F16: return small ? launch_conv<f16_t, 64>(p, ks, st) : launch_conv<f16_t, 128>(p, ks, st);
        default: return small ? launch_conv<bf16_t, 64>(p, ks, st) : launch_conv<bf16_t, 128>(p, ks, st);
    }
}


extern "C" int afcm_split16(void* parts, const float* x, const float* scale, const uint32_t* bound, int32_t dtype, int64_t planes, int32_t hw,
                            int32_t nparts, int64_t part_stride, void* stream) {
    AFCM_REQUIRE(parts != nullptr && x != nullptr && planes > 0 && hw > 0, "split16: empty input");
    AFCM_REQUIRE(dtype == AFCM_BF16 || dtype == AFCM_F16, "split16: parts are bfloat16 or float16");
    AFCM_REQUIRE(nparts == 2 || nparts == 3, "split16: 2 or 3 parts (got %d)", nparts);
    AFCM_REQUIRE(part_stride >= planes * (long long)hw && part_stride % 4 == 0, "split16: part stride %lld must cover the tensor and be a multiple of 4", (long long)part_stride);
    AFCM_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)parts & 7) == 0, "split16: x must be 16-byte, parts 8-byte aligned");
    long long blocks = (planes * ((hw + 3) >> 2) + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
#define AFCM_SPLIT(T, K) hipLaunchKernelGGL((split16_kernel<T, K>), grid, block, 0, st, (T*)parts, x, scale, (const unsigned*)bound, (long long)planes, hw, (long long)part_stride)
    if (dtype == AFCM_BF16) { if (nparts == 2) AFCM_SPLIT(bf16_t, 2); else AFCM_SPLIT(bf16_t, 3); }
    else { if (nparts == 2) AFCM_SPLIT(f16_t, 2); else AFCM_SPLIT(f16_t, 3); }
#undef AFCM_SPLIT
    return hip_status(hipGetLastError());
}

extern "C" int afcm_amax_bits(uint32_t* out, const float* x, int64_t planes, int32_t hw, const float* scale, void* stream) {
    AFCM_REQUIRE(out != nullptr && x != nullptr && planes > 0 && hw > 0, "amax_bits: empty input");
    AFCM_REQUIRE(((uintptr_t)out & 3) == 0 && ((uintptr_t)x & 3) == 0, "amax_bits: misaligned pointer");
    long long blocks = (planes * hw / 16 + 255) / 256;          // >= 4 16-byte groups per lane
    blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
    hipLaunchKernelGGL(amax_bits_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)out, x, (long long)planes, hw, scale);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_plane_dot_parts(float* out, const void* parts, int64_t part_stride, int32_t nparts, const float* b, int32_t dtype, int64_t planes,
                                    int32_t hw, const uint32_t* bound, void* stream) {
    AFCM_REQUIRE(out != nullptr && parts != nullptr && b != nullptr && planes > 0 && hw > 0, "plane_dot_parts: empty input");
    AFCM_REQUIRE(dtype == AFCM_BF16 || dtype == AFCM_F16, "plane_dot_parts: parts are bfloat16 or float16");
    AFCM_REQUIRE(nparts >= 1 && nparts <= 3 && part_stride >= planes * (long long)hw, "plane_dot_parts: 1..3 parts, a stride covering the tensor");
    AFCM_REQUIRE(planes < (1ll << 31), "plane_dot_parts: too many planes");
    AFCM_REQUIRE(((uintptr_t)parts & 7) == 0 && ((uintptr_t)b & 15) == 0, "plane_dot_parts: parts must be 8-byte, b 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AFCM_BF16) hipLaunchKernelGGL((plane_dot_parts_kernel<bf16_t>), dim3((unsigned)planes), dim3(256), 0, st, out, (const bf16_t*)parts, (long long)part_stride, nparts, b, (long long)planes, hw, (const unsigned*)bound);
    else hipLaunchKernelGGL((plane_dot_parts_kernel<f16_t>), dim3((unsigned)planes), dim3(256), 0, st, out, (const f16_t*)parts, (long long)part_stride, nparts, b, (long long)planes, hw, (const unsigned*)bound);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_unscale(float* t, int64_t numel, const uint32_t* bound_a, const uint32_t* bound_b, void* stream) {
    AFCM_REQUIRE(t != nullptr && numel > 0, "unscale: empty input");
    long long blocks = (numel + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(unscale_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, t, (long long)numel, (const unsigned*)bound_a, (const unsigned*)bound_b);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_conv2d_pack_split(void* dst, const float* w, const uint32_t* bound, int32_t dtype, int32_t cout, int32_t cin, int32_t mode,
                                      int32_t rows_pad, int32_t terms, uint32_t term_wparts, void* stream) {
    AFCM_REQUIRE(dst != nullptr && w != nullptr, "conv2d_pack_split: null pointer");
    AFCM_REQUIRE(dtype == AFCM_BF16 || dtype == AFCM_F16, "conv2d_pack_split: parts are bfloat16 or float16");
    AFCM_REQUIRE(mode == 0 || mode == 1, "mode must be 0 (forward) or 1 (data gradient)");
    AFCM_REQUIRE(terms >= 1 && terms <= 8 && cout > 0 && cin > 0, "conv2d_pack_split: 1..8 terms");
    const int rows = mode == 0 ? cout : cin, cols = mode == 0 ? cin : cout;
    AFCM_REQUIRE(rows_pad >= rows && rows_pad % 64 == 0, "rows_pad must be a multiple of 64 covering the rows");
    for (int t = 0; t < terms; t++) AFCM_REQUIRE(((term_wparts >> (4 * t)) & 15u) <= 2, "conv2d_pack_split: parts 0..2");
    const int BK = conv_bk(dtype, 3);
    const int nkc_real = cdiv(cols, BK);
    const long long total = (long long)terms * nkc_real * 9 * rows_pad * BK;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AFCM_BF16) hipLaunchKernelGGL((conv2d_pack_split_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (bf16_t*)dst, w, (const unsigned*)bound, cout, cin, rows, cols, rows_pad, nkc_real, terms, term_wparts, mode, BK);
    else hipLaunchKernelGGL((conv2d_pack_split_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, st, (f16_t*)dst, w, (const unsigned*)bound, cout, cin, rows, cols, rows_pad, nkc_real, terms, term_wparts, mode, BK);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_conv2d_split(float* y, const void* x_parts, const void* wpacked, const float* oscale, const float* obias, int32_t dtype, int32_t n,
                                 int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t pad, int32_t rows_pad, int32_t terms, uint32_t term_parts,
                                 int64_t part_stride, const uint32_t* bound_a, const uint32_t* bound_b, void* stream) {
    const int ks = 3;
    AFCM_REQUIRE(y != nullptr && x_parts != nullptr && wpacked != nullptr, "conv2d_split: null pointer");
    AFCM_REQUIRE(dtype == AFCM_BF16 || dtype == AFCM_F16, "conv2d_split: parts are bfloat16 or float16");
    AFCM_REQUIRE(n > 0 && cin > 0 && cout > 0 && h > 0 && w > 0, "x is empty");
    AFCM_REQUIRE(pad >= 0 && pad <= ks - 1, "padding must be in [0, k-1]");
    AFCM_REQUIRE(w % 2 == 0, "conv2d_split needs an even input width (got %d)", w);
    AFCM_REQUIRE(rows_pad >= cout && rows_pad % 64 == 0, "rows_pad must be a multiple of 64 covering cout");
    AFCM_REQUIRE(terms >= 1 && terms <= 8, "conv2d_split: 1..8 terms (got %d)", terms);
    int max_part = 0;
    for (int t = 0; t < terms; t++) max_part = std::max(max_part, (int)((term_parts >> (4 * t)) & 15u));
    AFCM_REQUIRE(max_part <= 2, "conv2d_split: parts 0..2");
    AFCM_REQUIRE(part_stride >= (long long)n * cin * h * w && (long long)max_part * part_stride * 2 < (1ll << 31) - (long long)cin * h * w * 2,
                 "conv2d_split: part stride %lld out of range", (long long)part_stride);
    ConvParams p;
    p.x = x_parts; p.y = y; p.wp = wpacked; p.oscale = oscale; p.obias = obias;
    p.N = n; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w;
    p.P = h + 2 * pad - ks + 1; p.Q = w + 2 * pad - ks + 1;
    AFCM_REQUIRE(p.P >= 1 && p.Q >= 1, "output must be at least 1x1");
    p.pad = pad;
    p.ldx = w; p.ldy = p.Q;
    choose_tile(p.P, p.Q, ks, &p.TH, &p.TW, &p.PWL, AFCM_X16_ON ? kPatchMaxX16 : kPatchMax);
    p.tilesX = cdiv(p.Q, p.TW); p.tilesY = cdiv(p.P, p.TH);
    p.magicTW = (unsigned)((0x100000000ull + (unsigned)p.TW - 1) / (unsigned)p.TW);
    p.magicTX = magic_u32((unsigned)p.tilesX); p.magicTY = magic_u32((unsigned)p.tilesY); p.magicN = magic_u32((unsigned)p.N); p.magicPC = magic_u32((unsigned)(p.PWL >> 2));
    p.Opad = rows_pad;
    p.nkc_real = cdiv(cin, conv_bk(dtype, 3));
    p.nkc = terms * p.nkc_real;
    p.magicNK = magic_u32((unsigned)p.nkc_real);
    p.term_parts = term_parts;
    p.part_bytes = (int)(part_stride * 2);
    p.last_part_bytes = max_part * p.part_bytes;
    p.bound_a = (const unsigned*)bound_a; p.bound_b = (const unsigned*)bound_b;
    const bool small = (rows_pad % 128 != 0) || cout <= 64;
    const long long blocks = (long long)p.tilesX * p.tilesY * p.N * cdiv(p.Cout, small ? 64 : 128);
    AFCM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "conv2d_split: grid of %lld blocks is out of range", blocks);
    dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
#if defined(AFCM_CONV_AB) || AFCM_CONV_X16
    if (AFCM_X16_ON) {
        p.total_blocks = (int)blocks; p.o_base = 0;
        const dim3 pgrid(small ? (unsigned)conv_persistent_grid(blocks, 3) : (unsigned)blocks);
        if (dtype == AFCM_BF16) {
            if (small) hipLaunchKernelGGL((conv2d_fwd16x_kernel<bf16_t, 64, true>), pgrid, block, 0, st, p);
            else hipLaunchKernelGGL((conv2d_fwd16x_kernel<bf16_t, 128, true>), pgrid, block, 0, st, p);
        } else {
            if (small) hipLaunchKernelGGL((conv2d_fwd16x_kernel<f16_t, 64, true>), pgrid, block, 0, st, p);
            else hipLaunchKernelGGL((conv2d_fwd16x_kernel<f16_t, 128, true>), pgrid, block, 0, st, p);
        }
        return hip_status(hipGetLastError());
    }
#endif
#if defined(AFCM_CONV_AB) || !AFCM_CONV_X16
    if (dtype == AFCM_BF16) {
        if (small) hipLaunchKernelGGL((conv2d_fwd16_kernel<bf16_t, 64, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv2d_fwd16_kernel<bf16_t, 128, true>), grid, block, 0, st, p);
    } else {
        if (small) hipLaunchKernelGGL((conv2d_fwd16_kernel<f16_t, 64, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((conv2d_fwd16_kernel<f16_t, 128, true>), grid, block, 0, st, p);
    }
#endif
    return hip_status(hipGetLastError());
}

// Split count for the weight gradient: enough workgroups to fill the chip, bounded by the K macro-steps.
static int wgrad_rows_per_step(int dtype) { return dtype == AFCM_F32 ? 1 : 2; }

// Workgroups per 64 x 64 tile.  One workgroup per CU is resident (LDS ring), so aim for ONE full round of the 256 CUs and never one
// workgroup more: rounding up (258 workgroups = two rounds) halves the throughput, and every extra split costs a 36 x 64 x 64 x 4 B
// partial tile written and read back (at 768 workgroups the partials of a 64 -> 64 layer were 2/3 of its time).
// two_class (conv2d_wgrad16g_kernel): tiles whose last 32 rows or columns lie outside the matrix (WgradParams::fo / fi) run their steps
// at ~0.6 of a full tile's cost (one live wave per SIMD: 1152 MFMA cycles + the ~600 of barrier and bookkeeping, against 2304 + 600),
// so they get 0.6 of a full tile's workgroups -- the round stays one round and every workgroup ends at about the same time.
// MEASURED (profiles/r05_wgrad_partial_tiles.txt): the skipped quadrants alone are worth 3-6 % on the 91-channel layers (L11 0.33 -> 0.31 ms)
// with the same number of workgroups per tile; the two-class plan on top made them SLOWER (L11 0.31 -> 0.36): a lone wave per SIMD cannot
// hide its own LDS latency, a step costs a partial tile nearer 0.8 than 0.6 of a full one.  Off; the plan and the reduce kernels keep the
// machinery (tests/test_gpu_conv.py ran green with it on).
#ifndef AFCM_WGRAD_TWO_CLASS
#define AFCM_WGRAD_TWO_CLASS 0
#endif
static void wgrad_plan(int n, int cout, int cin, int p_rows, bool two_class, int* fo, int* fi, int* splits, int* splits_p) {
    const int to = cdiv(cout, 64), ti = cdiv(cin, 64), tiles = to * ti;
    const long long ksteps = (long long)n * p_rows;   // upper bound on the macro-steps of any dtype
    auto clampk = [&](int v) { if (v > ksteps) v = (int)ksteps; return v < 1 ? 1 : v; };
    *fo = to; *fi = ti;
    *splits = *splits_p = clampk(256 / tiles);
    if (!two_class || !AFCM_WGRAD_TWO_CLASS) return;
    const int f_o = cout - (to - 1) * 64 <= 32 ? to - 1 : to, f_i = cin - (ti - 1) * 64 <= 32 ? ti - 1 : ti;
    const int nf = f_o * f_i, np = tiles - nf;
    if (nf == 0 || np == 0) return;                   // one class only
    int sf = (int)(256 / (nf + 0.6 * np)), sp = (256 - nf * sf) / np;
    if (sp < 1) { sp = 1; sf = (256 - np) / nf; }
    if (sf < 1) return;
    *fo = f_o; *fi = f_i; *splits = clampk(sf); *splits_p = clampk(sp);
}

extern "C" int afcm_conv2d_wgrad_splits(int32_t n, int32_t cout, int32_t cin, int32_t p_rows) {
    // slabs of the workspace: the most any tile of any kernel writes
    int fo, fi, s0, s0p, s1, s1p;
    wgrad_plan(n, cout, cin, p_rows, false, &fo, &fi, &s0, &s0p);
    wgrad_plan(n, cout, cin, p_rows, true, &fo, &fi, &s1, &s1p);
    return s0 > s1 ? s0 : s1;
}

extern "C" int afcm_conv2d_wgrad(float* dw, float* workspace, const void* dy, const void* x, int32_t dtype, int32_t n, int32_t cin,
                                 int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, void* stream) {
    return afcm_conv2d_wgrad_ld(dw, workspace, dy, x, dtype, n, cin, cout, h, w, ks, pad, 0, 0, stream);
}

static int wgrad_impl(float* dw, float* workspace, const void* dy, const void* x, int32_t dtype, int32_t n, int32_t cin,
                      int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t dy_pitch, int32_t x_pitch, float* dots, const float* wref, void* stream) {
    AFCM_REQUIRE(dw != nullptr && workspace != nullptr && dy != nullptr && x != nullptr, "conv2d_wgrad: null pointer");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    AFCM_REQUIRE(ks == 1 || ks == 3, "only 1x1 and 3x3 kernels are supported");
    AFCM_REQUIRE(pad >= 0 && pad <= ks - 1, "padding must be in [0, k-1]");
    WgradParams p;
    p.dy = dy; p.x = x; p.part = workspace;
    p.N = n; p.O = cout; p.I = cin; p.H = h; p.W = w; p.pad = pad;
    p.P = h + 2 * pad - ks + 1; p.Q = w + 2 * pad - ks + 1;
    AFCM_REQUIRE(p.P >= 1 && p.Q >= 1, "output must be at least 1x1");
    AFCM_REQUIRE(dtype == AFCM_F32 || (w % 2 == 0 && p.Q % 2 == 0), "16-bit conv2d_wgrad needs even widths (got %d, %d)", w, p.Q);
    p.lddy = dy_pitch ? dy_pitch : p.Q; p.ldx = x_pitch ? x_pitch : w;
    const bool pitched = p.lddy != p.Q || p.ldx != w;
    AFCM_REQUIRE(!pitched || (p.lddy >= p.Q && p.ldx >= w && ((p.lddy | p.ldx) & 1) == 0), "conv2d_wgrad: row pitches %d / %d must be even and cover the widths %d / %d", p.lddy, p.ldx, p.Q, w);
    const int R = wgrad_rows_per_step(dtype);
    p.qchunks = cdiv(p.Q, kWgKQ);
    p.rowgroups = cdiv(p.P, R);
    const long long ksteps = (long long)n * p.rowgroups * p.qchunks;
    const bool granule = (ks == 3 && pad == 2) || (ks == 1 && pad == 0);     // 16-byte LDS-DMA pieces; other paddings: 4-byte pieces
    wgrad_plan(n, cout, cin, p.P, dtype != AFCM_F32 && granule, &p.fo, &p.fi, &p.splits, &p.splits_p);
    if (p.splits > ksteps) p.splits = (int)ksteps;
    if (p.splits_p > ksteps) p.splits_p = (int)ksteps;
    p.steps_per_split = (int)((ksteps + p.splits - 1) / p.splits);
    p.steps_per_split_p = (int)((ksteps + p.splits_p - 1) / p.splits_p);
    p.splits_img = 0;
    if (dots != nullptr) {
        // image-aligned shares: the same number of workgroups, each inside one image (the granule kernel, one tile class, a split count that
        // is a multiple of the batch, at most 64 images: what wgrad_reduce_dots_kernel covers) -- else the caller takes its dot products
        // from the tensors themselves
        if (!(dtype != AFCM_F32 && granule) || p.splits != p.splits_p || n > 64 || p.splits % n != 0 || p.splits / n < 1) return AFCM_E_NOKERNEL;
        AFCM_REQUIRE(wref != nullptr, "conv2d_wgrad_dots: the weight tensor the dot products are taken with is missing");
        p.splits_img = p.splits / n;
        const long long per_img = (long long)p.rowgroups * p.qchunks;
        p.steps_per_split = p.steps_per_split_p = (int)((per_img + p.splits_img - 1) / p.splits_img);
    }
    const long long n_full = (long long)p.fo * p.fi;
    const long long blocks = n_full * p.splits + ((long long)cdiv(cout, 64) * cdiv(cin, 64) - n_full) * p.splits_p;
    dim3 grid((unsigned)blocks), block(512);
    hipStream_t st = (hipStream_t)stream;
#define AFCM_WG(T, R) do { if (ks == 3 && (pad & 1) == 0) hipLaunchKernelGGL((conv2d_wgrad_kernel<T, 3, R, 0>), grid, block, 0, st, p); \
                           else if (ks == 3) hipLaunchKernelGGL((conv2d_wgrad_kernel<T, 3, R, 1>), grid, block, 0, st, p); \
                           else hipLaunchKernelGGL((conv2d_wgrad_kernel<T, 1, R, 0>), grid, block, 0, st, p); } while (0)
#define AFCM_WG16(T) do { constexpr int NB = 3; \
                           if (ks == 3 && (pad & 1) == 0) hipLaunchKernelGGL((conv2d_wgrad16_kernel<T, 3, 0, NB>), grid, block, 0, st, p); \
                           else if (ks == 3) hipLaunchKernelGGL((conv2d_wgrad16_kernel<T, 3, 1, NB>), grid, block, 0, st, p); \
                           else hipLaunchKernelGGL((conv2d_wgrad16_kernel<T, 1, 0, NB>), grid, block, 0, st, p); } while (0)
#define AFCM_WG16G_(T, X) do { constexpr int NB = AFCM_WGRAD_NBUF; \
                            if (small && ks == 3) hipLaunchKernelGGL((conv2d_wgrad16g_kernel<T, 3, NB, true, X>), grid, block, 0, st, p); \
                            else if (small) hipLaunchKernelGGL((conv2d_wgrad16g_kernel<T, 1, NB, true, X>), grid, block, 0, st, p); \
                            else if (ks == 3) hipLaunchKernelGGL((conv2d_wgrad16g_kernel<T, 3, NB, false, X>), grid, block, 0, st, p); \
                            else hipLaunchKernelGGL((conv2d_wgrad16g_kernel<T, 1, NB, false, X>), grid, block, 0, st, p); } while (0)
    // MFMA shape (template flag X16; profiles/r05_conv_shape_ab.txt): 16x16x32 holds a higher clock on the large layers (+3 .. 6 %), but its K
    // step is 32 pixels where the 32x32x16 form skips dead 16-pixel groups: rows whose last 64-pixel chunk holds 33 .. 48 pixels (the 38-wide
    // planes of the 36^2 layers) cost it a whole extra step (-12 % there) -- those keep the 32x32x16 form.
    const int q_last = p.Q % 64;
    const bool wg_x16 = AFCM_X16_ON && !(q_last > 32 && q_last <= 48);
#define AFCM_WG16G(T) do { if (wg_x16) AFCM_WG16G_(T, true); else AFCM_WG16G_(T, false); } while (0)
    // tensors below 2 GB: one descriptor per tensor; larger ones: a descriptor per LDS-DMA piece (the general form)
    const bool small = (long long)n * cout * p.P * p.lddy * 2 < (1ll << 31) - 65536 &&
                       (long long)n * cin * h * p.ldx * 2 < (1ll << 31) - 65536;
    // rows by pitch: the 16-byte LDS-DMA kernel only (a granule straddling x's right edge is zeroed in LDS whatever follows it)
    AFCM_REQUIRE(!pitched || (dtype != AFCM_F32 && granule), "conv2d_wgrad: row pitches need the 16-bit granule kernel (3x3 pad 2 or 1x1 pad 0)");
    switch (dtype) {
        case AFCM_F32: AFCM_WG(float, 1); break;
        case AFCM_F16: if (granule) AFCM_WG16G(f16_t); else AFCM_WG16(f16_t); break;
        default: if (granule) AFCM_WG16G(bf16_t); else AFCM_WG16(bf16_t); break;
    }
#undef AFCM_WG16G
#undef AFCM_WG16G_
#undef AFCM_WG16
#undef AFCM_WG
    int rc = hip_status(hipGetLastError());
    if (rc != AFCM_OK) return rc;
    const long long numel = (long long)cout * cin * ks * ks;
    if (p.splits_img > 0) {
        const dim3 rgrid((unsigned)cin), rblock(1024);
#define AFCM_RD(T) do { if (ks == 3) hipLaunchKernelGGL((wgrad_reduce_dots_kernel<T, 9>), rgrid, rblock, 0, st, dw, dots, (const float*)workspace, wref, n, cout, cin, p.splits_img); \
                        else hipLaunchKernelGGL((wgrad_reduce_dots_kernel<T, 1>), rgrid, rblock, 0, st, dw, dots, (const float*)workspace, wref, n, cout, cin, p.splits_img); } while (0)
        if (dtype == AFCM_F16) AFCM_RD(f16_t); else AFCM_RD(bf16_t);
#undef AFCM_RD
        return hip_status(hipGetLastError());
    }
    const WgradSlabs slabs{cin, ks * ks, p.fo, p.fi, p.splits, p.splits_p};
    long long rb = (numel + 255) / 256;
    if (rb > 2048) rb = 2048;
    // splits beyond the last populated one were never launched with work: they still wrote zeros (acc = 0)
    if ((numel & 3) == 0 && (((uintptr_t)dw | (uintptr_t)workspace) & 15) == 0) {
        const long long n4 = numel / 4;
        if (p.splits >= 64) hipLaunchKernelGGL(wgrad_reduce4_kernel<16>, dim3((unsigned)cdiv(n4, 16)), dim3(256), 0, st, dw, (const float*)workspace, n4, slabs);
        else if (p.splits >= 8) hipLaunchKernelGGL(wgrad_reduce4_kernel<64>, dim3((unsigned)cdiv(n4, 64)), dim3(256), 0, st, dw, (const float*)workspace, n4, slabs);
        else hipLaunchKernelGGL(wgrad_reduce4_kernel<256>, dim3((unsigned)cdiv(n4, 256)), dim3(256), 0, st, dw, (const float*)workspace, n4, slabs);
    } else {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)rb), dim3(256), 0, st, dw, (const float*)workspace, numel, slabs);
    }
    return hip_status(hipGetLastError());
}

extern "C" int afcm_conv2d_wgrad_ld(float* dw, float* workspace, const void* dy, const void* x, int32_t dtype, int32_t n, int32_t cin,
                                    int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t dy_pitch, int32_t x_pitch, void* stream) {
    return wgrad_impl(dw, workspace, dy, x, dtype, n, cin, cout, h, w, ks, pad, dy_pitch, x_pitch, nullptr, nullptr, stream);
}

extern "C" int afcm_conv2d_wgrad_dots_ld(float* dw, float* dots, float* workspace, const void* dy, const void* x, const float* wref, int32_t dtype,
                                         int32_t n, int32_t cin, int32_t cout, int32_t h, int32_t w, int32_t ks, int32_t pad, int32_t dy_pitch,
                                         int32_t x_pitch, void* stream) {
    AFCM_REQUIRE(dots != nullptr, "conv2d_wgrad_dots: dots must be non-null");
    return wgrad_impl(dw, workspace, dy, x, dtype, n, cin, cout, h, w, ks, pad, dy_pitch, x_pitch, dots, wref, stream);
}

extern "C" int afcm_scale_planes(void* y, const void* x, const float* scale, int32_t dtype_in, int32_t dtype_out, int64_t planes,
                                 int32_t hw, void* stream) {
    AFCM_REQUIRE(y != nullptr && x != nullptr && planes > 0 && hw > 0, "scale_planes: empty input");
    long long blocks = (planes * ((hw + 3) >> 2) + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
#define AFCM_SP(TI, TO) hipLaunchKernelGGL((scale_planes_kernel<TI, TO>), grid, block, 0, st, (TO*)y, (const TI*)x, scale, (long long)planes, hw)
    if (dtype_in == AFCM_F32 && dtype_out == AFCM_F32) AFCM_SP(float, float);
    else if (dtype_in == AFCM_F32 && dtype_out == AFCM_BF16) AFCM_SP(float, bf16_t);
    else if (dtype_in == AFCM_F32 && dtype_out == AFCM_F16) AFCM_SP(float, f16_t);
    else if (dtype_in == AFCM_BF16 && dtype_out == AFCM_BF16) AFCM_SP(bf16_t, bf16_t);
    else if (dtype_in == AFCM_F16 && dtype_out == AFCM_F16) AFCM_SP(f16_t, f16_t);
    else if (dtype_in == AFCM_BF16 && dtype_out == AFCM_F32) AFCM_SP(bf16_t, float);
    else if (dtype_in == AFCM_F16 && dtype_out == AFCM_F32) AFCM_SP(f16_t, float);
    else { set_error("scale_planes: unsupported dtype pair %d -> %d", dtype_in, dtype_out); return AFCM_E_INVALID; }
#undef AFCM_SP
    return hip_status(hipGetLastError());
}

extern "C" int afcm_axpy_planes(void* y, const void* a, const void* b, const float* scale, int32_t dtype, int64_t planes, int32_t hw, void* stream) {
    AFCM_REQUIRE(y != nullptr && a != nullptr && b != nullptr && planes > 0 && hw > 0, "axpy_planes: empty input");
    AFCM_REQUIRE(dtype == AFCM_F16 || dtype == AFCM_BF16, "axpy_planes: 16-bit tensors only");
    if ((hw & 7) != 0 || ((((uintptr_t)y | (uintptr_t)a | (uintptr_t)b)) & 15) != 0) return AFCM_E_NOKERNEL;
    const int per = hw >> 3;
    // grid.y = planes (one scale per workgroup), grid.x = 1 KB-per-thread-block slices of a plane: >= 4 vectors per thread on the large planes
    const dim3 grid((unsigned)((per + 1023) / 1024), (unsigned)(planes < 65535 ? planes : 65535));
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AFCM_BF16) hipLaunchKernelGGL((axpy_planes_kernel<bf16_t>), grid, dim3(256), 0, st, (bf16_t*)y, (const bf16_t*)a, (const bf16_t*)b, scale, (long long)planes, hw);
    else hipLaunchKernelGGL((axpy_planes_kernel<f16_t>), grid, dim3(256), 0, st, (f16_t*)y, (const f16_t*)a, (const f16_t*)b, scale, (long long)planes, hw);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_plane_dot(float* out, const void* a, const void* b, int32_t dtype, int64_t planes, int32_t hw, void* stream) {
    AFCM_REQUIRE(out != nullptr && a != nullptr && planes > 0 && hw > 0, "plane_dot: empty input");
    AFCM_REQUIRE(planes < (1ll << 31), "plane_dot: too many planes");
    AFCM_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 15) == 0, "plane_dot: operands must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int esize = dtype == AFCM_F32 ? 4 : 2;
    const bool per_wave = (long long)hw * esize <= 16384;
    dim3 grid((unsigned)(per_wave ? (planes + 3) / 4 : planes)), block(256);
#define AFCM_PD(T) do { \
        if (per_wave) hipLaunchKernelGGL((plane_dot_wave_kernel<T>), grid, block, 0, st, out, (const T*)a, (const T*)b, (long long)planes, hw); \
        else hipLaunchKernelGGL((plane_dot_kernel<T>), grid, block, 0, st, out, (const T*)a, (const T*)b, (long long)planes, hw); } while (0)
    switch (dtype) {
        case AFCM_F32: AFCM_PD(float); break;
        case AFCM_F16: AFCM_PD(f16_t); break;
        case AFCM_BF16: AFCM_PD(bf16_t); break;
        default: set_error("plane_dot: bad dtype"); return AFCM_E_INVALID;
    }
#undef AFCM_PD
    return hip_status(hipGetLastError());
}

static int plane_dot_rows(float* out, const void* a, const void* b, int32_t dtype, int64_t planes, int32_t h, int32_t w,
                          int32_t a_pitch, int32_t b_pitch, const PlaneGate& gate, void* stream) {
    AFCM_REQUIRE(out != nullptr && a != nullptr && planes > 0 && h > 0 && w > 0, "plane_dot: empty input");
    AFCM_REQUIRE(planes < (1ll << 31), "plane_dot: too many planes");
    const int esize = dtype == AFCM_F32 ? 4 : 2;
    const int lda = a_pitch ? a_pitch : w, ldb = b_pitch ? b_pitch : w;
    AFCM_REQUIRE(lda >= w && ldb >= w, "plane_dot: row pitches %d / %d are below the width %d", lda, ldb, w);
    AFCM_REQUIRE(w >= 16 / esize, "plane_dot: rows of %d elements are shorter than one 16-byte vector", w);
    AFCM_REQUIRE((((uintptr_t)a | (uintptr_t)b) & 3) == 0 && (esize == 4 || ((w | lda | ldb) & 1) == 0), "plane_dot: rows must start on 4-byte boundaries");
    AFCM_REQUIRE((long long)h * (lda > ldb ? lda : ldb) < (1ll << 31) / 16, "plane_dot: plane is out of range");
    hipStream_t st = (hipStream_t)stream;
    const bool per_wave = (long long)h * w * esize <= 16384;
    dim3 grid((unsigned)(per_wave ? (planes + 3) / 4 : planes)), block(256);
#define AFCM_PDR(T) do { \
        if (per_wave) hipLaunchKernelGGL((plane_dot_rows_kernel<T, true>), grid, block, 0, st, out, (const T*)a, (const T*)b, (long long)planes, h, w, lda, ldb, gate); \
        else hipLaunchKernelGGL((plane_dot_rows_kernel<T, false>), grid, block, 0, st, out, (const T*)a, (const T*)b, (long long)planes, h, w, lda, ldb, gate); } while (0)
    switch (dtype) {
        case AFCM_F32: AFCM_PDR(float); break;
        case AFCM_F16: AFCM_PDR(f16_t); break;
        case AFCM_BF16: AFCM_PDR(bf16_t); break;
        default: set_error("plane_dot: bad dtype"); return AFCM_E_INVALID;
    }
#undef AFCM_PDR
    return hip_status(hipGetLastError());
}

extern "C" int afcm_plane_dot_ld(float* out, const void* a, const void* b, int32_t dtype, int64_t planes, int32_t h, int32_t w,
                                 int32_t a_pitch, int32_t b_pitch, void* stream) {
    return plane_dot_rows(out, a, b, dtype, planes, h, w, a_pitch, b_pitch, PlaneGate{nullptr, 0, nullptr, nullptr, nullptr, nullptr}, stream);
}

extern "C" int afcm_plane_dot_gated_ld(float* out, const void* a, const void* b, int32_t dtype, int64_t planes, int32_t h, int32_t w,
                                       int32_t a_pitch, int32_t b_pitch, const int32_t* flags, int32_t slots, const float* out_scale,
                                       const float* gz, const float* next_scale, const float* gskip, void* stream) {
    AFCM_REQUIRE(flags != nullptr && slots > 0 && out_scale != nullptr && gz != nullptr && b != nullptr, "plane_dot_gated: null pointer");
    return plane_dot_rows(out, a, b, dtype, planes, h, w, a_pitch, b_pitch, PlaneGate{flags, slots, out_scale, gz, next_scale, gskip}, stream);
}
