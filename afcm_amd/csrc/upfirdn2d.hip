// upfirdn2d for gfx950: zero-insert upsample, pad/crop, 2-D FIR, decimate (SG3OPS/upfirdn2d.py:167-211
// is the specification; output size of SG3OPS/upfirdn2d.cpp:35-36).  Generic gather form: the filter
// (with flip and gain folded in) lives in LDS, every lane walks only the taps that hit a real
// sample (polyphase stepping), lanes map to consecutive output columns so reads and writes coalesce.
// Covers every up/down/tap combination incl. the 61-tap separable Gaussian of the loss blur
// (stylegan3_model.py:25-28), which arrives as two 1-D calls (SG3OPS/upfirdn2d.py:244-245).
#include <stdlib.h>

#include "common.h"

namespace afcm {

struct UpfirdnParams {
    void* y;
    const void* x;
    int xw, xh, yw, yh, fw, fh;
    int upx, upy, downx, downy, padx0, pady0;
    int flip;
    float gain;
    long long planes;
};

constexpr int kMaxTaps = 4096;
#ifndef AFCM_UPF_RPT11
#define AFCM_UPF_RPT11 8           // output rows per thread of the (up 1, down 1) row kernel (tuning aid)
#endif
#ifndef AFCM_UPFIRDN_ROWS
#define AFCM_UPFIRDN_ROWS 1        // 16-bit small filters on upfirdn2d_rows_kernel (0: the LDS tile kernel; A/B builds)
#endif

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(UpfirdnParams p, const float* __restrict__ f) {
    __shared__ float taps[kMaxTaps];
    const int nt = p.fw * p.fh;
    // taps[ky*fw+kx] multiplies z[Uy+ky][Ux+kx]; a true convolution unless `flip`.
    for (int i = threadIdx.x; i < nt; i += blockDim.x) taps[i] = (p.flip ? f[i] : f[nt - 1 - i]) * p.gain;
    __syncthreads();
    const int tilesX = (p.yw + 63) >> 6;
    const int tilesY = (p.yh + 3) >> 2;
    const long long nblk = (long long)tilesX * tilesY * p.planes;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int tx = (int)(blk % tilesX);
        const long long t = blk / tilesX;
        const int ty = (int)(t % tilesY);
        const long long plane = t / tilesY;
        const int ox = tx * 64 + lx, oy = ty * 4 + ly;
        if (ox >= p.yw || oy >= p.yh) continue;
        const T* xp = (const T*)p.x + plane * p.xh * p.xw;
        const int Ux = ox * p.downx - p.padx0, Uy = oy * p.downy - p.pady0;
        const int kx0 = pos_mod(-Ux, p.upx), ky0 = pos_mod(-Uy, p.upy);
        const int ix0 = (Ux + kx0) / p.upx, iy0 = (Uy + ky0) / p.upy;   // exact (numerator divisible)
        float acc = 0.f;
        for (int ky = ky0, iy = iy0; ky < p.fh; ky += p.upy, iy++) {
            if ((unsigned)iy >= (unsigned)p.xh) continue;
            const T* row = xp + (size_t)iy * p.xw;
            const float* trow = taps + ky * p.fw;
            for (int kx = kx0, ix = ix0; kx < p.fw; kx += p.upx, ix++)
                if ((unsigned)ix < (unsigned)p.xw) acc = fmaf(trow[kx], to_f32(row[ix]), acc);
        }
        ((T*)p.y)[plane * p.yh * p.yw + (size_t)oy * p.yw + ox] = from_f32<T>(acc);
    }
}

// Small-filter tile kernel: filters of at most 4 x 4 taps with equal factors in x and y, (up, down) in {(1, 1), (1, 2), (2, 1)} --
// the discriminator's [1, 3, 3, 1] blur, its decimating skip path and their transposes (CoModGAN/layers.py:115-162 through
// conv2d_resample; the R1 double backward runs the same three shapes).  One workgroup = 64 x 16 outputs: the input tile is
// staged once into LDS as fp32 (coalesced loads, zero outside the plane), every thread then computes 4 consecutive outputs
// of one row from 16-byte LDS reads with the taps in scalar registers.  The gather kernel above reads every input 16 times
// through 2-byte loads (0.9 TB/s on the 256^2 blur); this one is bound by the staging traffic.
template <int UP, int DOWN> struct UpfTile {
    static constexpr int TOX = 64, TOY = 32;          // 16 x 16 threads, 4 columns x (TOY / 16) rows each
    static constexpr int UW = (TOX - 1) * DOWN + 4, UH = (TOY - 1) * DOWN + 4;       // extent in the zero-inserted domain
    static constexpr int IW = UP == 1 ? UW : UW / 2 + 2, IH = UP == 1 ? UH : UH / 2 + 2;
    static constexpr int IWP = (IW + 3) / 4 * 4 + 4;                                 // row pitch in floats (16-byte rows)
};

template <typename T, int UP, int DOWN>
__global__ __launch_bounds__(256) void upfirdn2d_tile_kernel(UpfirdnParams p, const float* __restrict__ f) {
    typedef UpfTile<UP, DOWN> G;
    __shared__ __attribute__((aligned(16))) float tile[G::IH * G::IWP];
    // taps[ky][kx] multiplies z[Uy+ky][Ux+kx] (true convolution unless `flip`), zero beyond the filter: uniform -> SGPRs
    float t[4][4];
    const int nt = p.fw * p.fh;
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
            const int i = ky * p.fw + kx;
            t[ky][kx] = (ky < p.fh && kx < p.fw) ? (p.flip ? f[i] : f[nt - 1 - i]) * p.gain : 0.f;
        }
    const int tilesX = (p.yw + G::TOX - 1) / G::TOX, tilesY = (p.yh + G::TOY - 1) / G::TOY;
    int bid = blockIdx.x;
    const int tx = bid % tilesX; bid /= tilesX;
    const int ty = bid % tilesY;
    const long long plane = bid / tilesY;
    const int ox0 = tx * G::TOX, oy0 = ty * G::TOY;
    const int Ux0 = ox0 * DOWN - p.padx0, Uy0 = oy0 * DOWN - p.pady0;               // tile origin in the zero-inserted domain
    // first input sample at or after the origin (UP = 2: ceil(U / 2) for negative U too)
    const int ix0 = UP == 1 ? Ux0 : (Ux0 + (Ux0 & 1)) / 2, iy0 = UP == 1 ? Uy0 : (Uy0 + (Uy0 & 1)) / 2;
    const T* xp = (const T*)p.x + plane * p.xh * p.xw;
    {
        // every load of the thread in flight before the first LDS write (as a run-time loop each load waited for the previous one)
        constexpr int NLD = (G::IH * G::IWP + 255) / 256;
        float v[NLD];
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int idx = threadIdx.x + 256 * k;
            const int r = idx / G::IWP, c = idx - r * G::IWP;
            const int iy = iy0 + r, ix = ix0 + c;
            v[k] = 0.f;
            if (idx < G::IH * G::IWP && (unsigned)iy < (unsigned)p.xh && (unsigned)ix < (unsigned)p.xw) v[k] = to_f32(xp[(size_t)iy * p.xw + ix]);
        }
#pragma unroll
        for (int k = 0; k < NLD; k++) {
            const int idx = threadIdx.x + 256 * k;
            if (idx < G::IH * G::IWP) tile[idx] = v[k];
        }
    }
    __syncthreads();
    // a workgroup's life is one memory round trip for the tile plus this loop: two output rows per thread (TOY = 32) halve the
    // number of round trips per output (the 16-row tile ran at 2.3 TB/s with 8 workgroups per CU, each waiting on its one load)
    const int lx4 = threadIdx.x & 15;
#pragma unroll
    for (int hrow = 0; hrow < G::TOY / 16; hrow++) {
    const int ly = (threadIdx.x >> 4) + 16 * hrow;
    const int oy = oy0 + ly, ox = ox0 + 4 * lx4;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    if constexpr (UP == 1) {
        constexpr int NV = 3 * DOWN + 4, NV4 = (NV + 3) / 4;
#pragma unroll
        for (int ky = 0; ky < 4; ky++) {
            const float4* row = (const float4*)(tile + (ly * DOWN + ky) * G::IWP + 4 * lx4 * DOWN);
            float v[NV4 * 4];
#pragma unroll
            for (int q = 0; q < NV4; q++) { const float4 w = row[q]; v[4 * q] = w.x; v[4 * q + 1] = w.y; v[4 * q + 2] = w.z; v[4 * q + 3] = w.w; }
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int kx = 0; kx < 4; kx++) acc[j] = fmaf(t[ky][kx], v[j * DOWN + kx], acc[j]);
        }
    } else {
        // zero insertion: only taps with (U + k) even meet a sample; two per axis for a 4-tap filter
        const int Uy = Uy0 + ly;
#pragma unroll
        for (int a = 0; a < 2; a++) {
            const int ky = ((Uy & 1) ? 1 : 0) + 2 * a;
            const int r = (Uy + ky) / 2 - iy0;                                       // Uy + ky even: exact also below zero
            const float* row = tile + r * G::IWP;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int Ux = Ux0 + 4 * lx4 + j;
#pragma unroll
                for (int b = 0; b < 2; b++) {
                    const int kx = ((Ux & 1) ? 1 : 0) + 2 * b;
                    const float tap = ky == 0 ? (kx == 0 ? t[0][0] : kx == 1 ? t[0][1] : kx == 2 ? t[0][2] : t[0][3])
                                    : ky == 1 ? (kx == 0 ? t[1][0] : kx == 1 ? t[1][1] : kx == 2 ? t[1][2] : t[1][3])
                                    : ky == 2 ? (kx == 0 ? t[2][0] : kx == 1 ? t[2][1] : kx == 2 ? t[2][2] : t[2][3])
                                              : (kx == 0 ? t[3][0] : kx == 1 ? t[3][1] : kx == 2 ? t[3][2] : t[3][3]);
                    acc[j] = fmaf(tap, row[(Ux + kx) / 2 - ix0], acc[j]);
                }
            }
        }
    }
    if (oy < p.yh) {
        T* yp = (T*)p.y + plane * p.yh * p.yw + (size_t)oy * p.yw;
        if ((p.yw & 3) == 0 && ox + 3 < p.yw && (((uintptr_t)p.y) & 15) == 0) {
            // rows of whole 4-element groups: one 8-byte (16-bit types) / 16-byte (fp32) store per lane
            struct alignas(4 * sizeof(T)) Out { T v[4]; } o;
#pragma unroll
            for (int j = 0; j < 4; j++) o.v[j] = from_f32<T>(acc[j]);
            *(Out*)(yp + ox) = o;
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ox + j < p.yw) yp[ox + j] = from_f32<T>(acc[j]);
        }
    }
    }
}

// 16-bit small-filter kernel, no LDS: filters of at most 4 x 4 taps, (up, down) in {(1, 1), (1, 2), (2, 1)}, EVEN widths (rows then start on
// 4-byte boundaries, which is all a 16-byte global access needs).  A thread owns 8 consecutive output columns x RPT output rows: every
// input row it needs arrives as one, two or three 16-byte loads, all in flight before the first product (the tile kernel above moves 2 bytes
// per lane and instruction through LDS: 1.3 TB/s on the discriminator's 256^2 blur); the rows above / below are the neighbouring thread's
// in L1 / L2.  Outputs leave as one 16-byte store per row.  The taps meet every output in the tile kernel's order (ky, then kx), so both
// kernels produce the same bits.
//   XODD: the first input column of a group is odd (UP 1: padx0 odd; UP 2: the zero-inserted origin x0 - padx0 is odd);  YODD (UP 2): the
//   zero-inserted origin row is odd.  Both are uniform over the launch (x0 is a multiple of 8, the first row of a strip a multiple of RPT).
template <int UP, int DOWN> struct UpfRows {
    static constexpr int RPT = (UP == 1 && DOWN == 2) ? 4 : (UP == 1 ? AFCM_UPF_RPT11 : 8);   // output rows per thread
    static constexpr int NIN = UP == 1 ? (RPT - 1) * DOWN + 4 : 6;                      // input rows a thread reads
    static constexpr int NEED = UP == 1 ? 7 * DOWN + 4 : 6;                             // input columns per row (from the group's first)
    static constexpr int NL = (NEED + 1 + 7) / 8;                                       // 16-byte loads per row (+ 1: the odd start)
};

// SMALLF: the filter has fewer than 4 x 4 taps -- slots beyond it are SKIPPED (uniform branches on fh / fw), not multiplied by a zero tap:
// 0 * inf = NaN would poison positions the reference never touches (the one-tap zero-stuffing call of the stride-2 backward
// turned an overflowed dy element into NaN in its three stuffed neighbours as well; a fill + strided copy writes exact zeros there)
template <typename T, int UP, int DOWN, bool XODD, bool YODD, bool SMALLF>
__global__ __launch_bounds__(256) void upfirdn2d_rows_kernel(UpfirdnParams p, const float* __restrict__ f, int ncg, int nstrips, long long total) {
    typedef UpfRows<UP, DOWN> G;
    constexpr int RPT = G::RPT, NIN = G::NIN, NL = G::NL;
    static_assert(sizeof(T) == 2, "16-bit types");
    float t[4][4];
    const int nt = p.fw * p.fh;
#pragma unroll
    for (int ky = 0; ky < 4; ky++)
#pragma unroll
        for (int kx = 0; kx < 4; kx++) {
            const int i = ky * p.fw + kx;
            const float tv = (ky < p.fh && kx < p.fw) ? (p.flip ? f[i] : f[nt - 1 - i]) * p.gain : 0.f;
            t[ky][kx] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, tv)));   // the product runs on the vector pipe: pin the tap to an SGPR
        }
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int cg = (int)(gid % ncg);
    const long long rest = gid / ncg;
    const int strip = (int)(rest % nstrips);
    const long long plane = rest / nstrips;
    const int x0 = cg * 8, oy0 = strip * RPT;

    // first input column / row of the group, and the even column the loads start from
    int c0, iy0;
    if constexpr (UP == 1) {
        c0 = x0 * DOWN - p.padx0; iy0 = oy0 * DOWN - p.pady0;
    } else {
        const int Ux0 = x0 - p.padx0, Uy0 = oy0 - p.pady0;
        c0 = (Ux0 + (XODD ? 1 : 0)) / 2; iy0 = (Uy0 + (YODD ? 1 : 0)) / 2;              // exact: the numerators are even
    }
    const int e = UP == 1 ? (XODD ? 1 : 0) : (c0 & 1);                                  // (UP 2: uniform over the launch as well)
    const int a0 = c0 - e;
    union Row { uint4 q[NL]; T v[NL * 8]; unsigned d[NL * 4]; };
    Row raw[NIN];
    // One code path for every group (a wave of 64 consecutive groups always holds some that touch the plane's left or right edge: as a
    // separate guarded path every wave ran both, and the guarded one set the time).  Buffer loads over the whole tensor: a dword before
    // its first or after its last element reads as zero instead of faulting; columns outside [0, xw) -- which lie in the neighbouring
    // rows -- are masked per 16-bit half with masks made once per thread; a row outside the plane reads the nearest valid one and is
    // masked whole.  Every load is in flight before the first use.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)(p.planes * p.xh * p.xw * 2), 0x00020000);
    const int plane_el = (int)plane * p.xh * p.xw;
#pragma unroll
    for (int r = 0; r < NIN; r++) {
        const int iyc = min(max(iy0 + r, 0), p.xh - 1);
        const unsigned off = (unsigned)((plane_el + iyc * p.xw + a0) * 2);
#pragma unroll
        for (int l = 0; l < NL; l++) raw[r].q[l] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, off + 16u * l, 0, 0));
    }
    if (plane_el + a0 < 0) {
        // the tensor's very first group: row 0 would start at a negative offset, which the range check answers with zeros for the whole
        // vector, not only for the dwords before the buffer -- read it from offset 0 and move the dwords up (a0 = -2 or -4: 1 or 2 dwords)
        const int sh = (-a0) >> 1;
#pragma unroll
        for (int r = 0; r < NIN; r++) {
            if (iy0 + r > 0) continue;                                                   // rows <= 0 all read row 0
            unsigned ld[NL * 4];
#pragma unroll
            for (int l = 0; l < NL; l++) {
                const uint4 q = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(xrs, 16u * l, 0, 0));
                ld[4 * l] = q.x; ld[4 * l + 1] = q.y; ld[4 * l + 2] = q.z; ld[4 * l + 3] = q.w;
            }
#pragma unroll
            for (int k = 0; k < NL * 4; k++) raw[r].d[k] = sh == 1 ? (k >= 1 ? ld[k >= 1 ? k - 1 : 0] : 0u) : (k >= 2 ? ld[k >= 2 ? k - 2 : 0] : 0u);
        }
    }
    if (a0 < 0 || a0 + NL * 8 > p.xw) {                                                  // an edge group: columns outside the row
        unsigned cm[NL * 4];
#pragma unroll
        for (int k = 0; k < NL * 4; k++)
            cm[k] = ((unsigned)(a0 + 2 * k) < (unsigned)p.xw ? 0xffffu : 0u) | ((unsigned)(a0 + 2 * k + 1) < (unsigned)p.xw ? 0xffff0000u : 0u);
#pragma unroll
        for (int r = 0; r < NIN; r++)
#pragma unroll
            for (int k = 0; k < NL * 4; k++) raw[r].d[k] &= cm[k];
    }
#pragma unroll
    for (int r = 0; r < NIN; r++) {
        const unsigned keep = (unsigned)(iy0 + r) < (unsigned)p.xh ? 0xffffffffu : 0u;
#pragma unroll
        for (int k = 0; k < NL * 4; k++) raw[r].d[k] &= keep;
    }
    float acc[RPT][8];
#pragma unroll
    for (int o = 0; o < RPT; o++)
#pragma unroll
        for (int j = 0; j < 8; j++) acc[o][j] = 0.f;
#pragma unroll
    for (int r = 0; r < NIN; r++) {
        float v[NL * 8 - 1];
#pragma unroll
        for (int i = 0; i < NL * 8 - 1; i++) v[i] = e ? to_f32(raw[r].v[i + 1]) : to_f32(raw[r].v[i]);
        if constexpr (UP == 1) {
#pragma unroll
            for (int o = 0; o < RPT; o++) {
                constexpr int dummy = 0; (void)dummy;
                const int ky = r - o * DOWN;                                               // compile-time after unrolling
                if (ky >= 0 && ky < 4 && (!SMALLF || ky < p.fh)) {
#pragma unroll
                    for (int kx = 0; kx < 4; kx++) {
                        if (SMALLF && kx >= p.fw) continue;
#pragma unroll
                        for (int j = 0; j < 8; j++) acc[o][j] = fmaf(t[ky][kx], v[j * DOWN + kx], acc[o][j]);
                    }
                }
            }
        } else {
            // zero insertion: output row o (zero-inserted row Uy0 + o) meets the taps ky = par + 2 a at input row (o + par - qy) / 2 + a
#pragma unroll
            for (int o = 0; o < RPT; o++) {
                const int qy = YODD ? 1 : 0, par = (qy + o) & 1;
#pragma unroll
                for (int a = 0; a < 2; a++) {
                    if ((o + par - qy) / 2 + a != r) continue;
                    const int ky = par + 2 * a;
                    if (SMALLF && ky >= p.fh) continue;
#pragma unroll
                    for (int j = 0; j < 8; j++) {
                        const int qx = XODD ? 1 : 0, parx = (qx + j) & 1;
#pragma unroll
                        for (int b = 0; b < 2; b++) {
                            const int kx = parx + 2 * b;
                            if (SMALLF && kx >= p.fw) continue;
                            acc[o][j] = fmaf(t[ky][kx], v[(j + parx - qx) / 2 + b], acc[o][j]);
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int o = 0; o < RPT; o++) {
        const int oy = oy0 + o;
        if (oy >= p.yh) break;
        T* yp = (T*)p.y + plane * p.yh * p.yw + (size_t)oy * p.yw + x0;
        if (x0 + 8 <= p.yw) {
            union { uint4 q; T v[8]; } out;
#pragma unroll
            for (int j = 0; j < 8; j++) out.v[j] = from_f32<T>(acc[o][j]);
            __builtin_memcpy(__builtin_assume_aligned(yp, 4), &out.q, 16);
        } else {
#pragma unroll
            for (int j = 0; j < 8; j++)
                if (x0 + j < p.yw) yp[j] = from_f32<T>(acc[o][j]);
        }
    }
}

template <typename T>
static bool launch_rows(const UpfirdnParams& p, const float* f, hipStream_t st) {
    if constexpr (sizeof(T) != 2) return false;
    else {
        const bool sq = p.upx == p.upy && p.downx == p.downy && p.fw <= 4 && p.fh <= 4;
        if (!sq || ((p.xw | p.yw) & 1) || (((uintptr_t)p.x | (uintptr_t)p.y) & 3)) return false;
        if (p.planes * p.xh * p.xw * 2 >= (1ll << 31) || p.planes * p.yh * p.yw >= (1ll << 31)) return false;   // 32-bit buffer offsets / element indices
        const int up = p.upx, down = p.downx;
        if (!((up == 1 && (down == 1 || down == 2)) || (up == 2 && down == 1))) return false;
        if (p.padx0 > (up == 1 ? 3 : 6)) return false;                      // the first group starts at most 4 columns left of the plane (kernel: sh <= 2)
        const int rpt = up == 2 ? UpfRows<2, 1>::RPT : down == 2 ? UpfRows<1, 2>::RPT : UpfRows<1, 1>::RPT;
        const int ncg = (p.yw + 7) / 8, nstrips = (p.yh + rpt - 1) / rpt;
        const long long total = (long long)ncg * nstrips * p.planes;
        const long long nblk = (total + 255) / 256;
        if (nblk <= 0 || nblk >= (1ll << 31)) return false;
        dim3 grid((unsigned)nblk), block(256);
        const bool xo = (p.padx0 & 1) != 0, yo = (p.pady0 & 1) != 0;      // x0 is a multiple of 8, a strip's first row a multiple of 4
        const bool smallf = p.fw < 4 || p.fh < 4;
#define AFCM_UPF_ROWS(U, D, XO, YO) do { if (smallf) hipLaunchKernelGGL((upfirdn2d_rows_kernel<T, U, D, XO, YO, true>), grid, block, 0, st, p, f, ncg, nstrips, total); \
        else hipLaunchKernelGGL((upfirdn2d_rows_kernel<T, U, D, XO, YO, false>), grid, block, 0, st, p, f, ncg, nstrips, total); } while (0)
        if (up == 1 && down == 1) { if (xo) AFCM_UPF_ROWS(1, 1, true, false); else AFCM_UPF_ROWS(1, 1, false, false); }
        else if (up == 1) { if (xo) AFCM_UPF_ROWS(1, 2, true, false); else AFCM_UPF_ROWS(1, 2, false, false); }
        else if (xo) { if (yo) AFCM_UPF_ROWS(2, 1, true, true); else AFCM_UPF_ROWS(2, 1, true, false); }
        else { if (yo) AFCM_UPF_ROWS(2, 1, false, true); else AFCM_UPF_ROWS(2, 1, false, false); }
#undef AFCM_UPF_ROWS
        return true;
    }
}

template <typename T>
static bool launch_tile(const UpfirdnParams& p, const float* f, hipStream_t st) {
    const bool sq = p.upx == p.upy && p.downx == p.downy && p.fw <= 4 && p.fh <= 4;
    if (!sq) return false;
    constexpr int TOX = UpfTile<1, 1>::TOX, TOY = UpfTile<1, 1>::TOY;             // the same output tile for every (up, down)
    const long long nblk = (long long)((p.yw + TOX - 1) / TOX) * ((p.yh + TOY - 1) / TOY) * p.planes;
    if (nblk <= 0 || nblk >= (1ll << 31)) return false;
    dim3 grid((unsigned)nblk), block(256);
    if (p.upx == 1 && p.downx == 1) hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, 1, 1>), grid, block, 0, st, p, f);
    else if (p.upx == 1 && p.downx == 2) hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, 1, 2>), grid, block, 0, st, p, f);
    else if (p.upx == 2 && p.downx == 1) hipLaunchKernelGGL((upfirdn2d_tile_kernel<T, 2, 1>), grid, block, 0, st, p, f);
    else return false;
    return true;
}

}  // namespace afcm

using namespace afcm;

extern "C" int afcm_upfirdn2d(void* y, const void* x, const float* f, int32_t dtype, int32_t n, int32_t c, int32_t xh, int32_t xw,
                              int32_t yh, int32_t yw, int32_t fh, int32_t fw, int32_t upx, int32_t upy, int32_t downx, int32_t downy,
                              int32_t padx0, int32_t pady0, int32_t flip, float gain, void* stream) {
    AFCM_REQUIRE(x != nullptr && y != nullptr && f != nullptr, "upfirdn2d: x, y and f must be non-null");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    AFCM_REQUIRE(n > 0 && c > 0 && xh > 0 && xw > 0, "x is empty");
    AFCM_REQUIRE(fh >= 1 && fw >= 1, "f is empty");
    AFCM_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "upsampling and downsampling factors must be at least 1");
    AFCM_REQUIRE(yh >= 1 && yw >= 1, "output must be at least 1x1");
    if ((long long)fh * fw > kMaxTaps) return AFCM_E_NOKERNEL;
    UpfirdnParams p;
    p.y = y; p.x = x; p.xw = xw; p.xh = xh; p.yw = yw; p.yh = yh; p.fw = fw; p.fh = fh;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0;
    p.flip = flip; p.gain = gain; p.planes = (long long)n * c;
    long long nblk = (long long)((yw + 63) >> 6) * ((yh + 3) >> 2) * p.planes;
    if (nblk > 256 * 64) nblk = 256 * 64;
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (AFCM_UPFIRDN_ROWS && dtype != AFCM_F32) {
        const bool done = dtype == AFCM_F16 ? launch_rows<f16_t>(p, f, st) : launch_rows<bf16_t>(p, f, st);
        if (done) return hip_status(hipGetLastError());
    }
    {
        const bool done = dtype == AFCM_F32 ? launch_tile<float>(p, f, st) : dtype == AFCM_F16 ? launch_tile<f16_t>(p, f, st) : launch_tile<bf16_t>(p, f, st);
        if (done) return hip_status(hipGetLastError());
    }
    switch (dtype) {
        case AFCM_F32: hipLaunchKernelGGL((upfirdn2d_kernel<float>), grid, block, 0, st, p, f); break;
        case AFCM_F16: hipLaunchKernelGGL((upfirdn2d_kernel<f16_t>), grid, block, 0, st, p, f); break;
        default: hipLaunchKernelGGL((upfirdn2d_kernel<bf16_t>), grid, block, 0, st, p, f); break;
    }
    return hip_status(hipGetLastError());
}
