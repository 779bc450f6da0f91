// 3x3 convolution of an image with AT MOST FOUR input channels (the generator's first layer, `encoder_0`: 4 -> 64 channels at 276^2,
// NET:505 through conv2d_gradfix.conv2d) -- r06.
//
// The implicit-GEMM kernel (conv2d_fwd16x_kernel) contracts 32 channels per tap and K step: with 4 channels seven eighths of every
// product multiply zeros (79 TFLOP/s = 0.03 of the pipe, 72 us for a layer whose output alone is 158 MB = 26 us of HBM time).  Here the
// contraction index of one MFMA is (tap, channel): v_mfma_f32_16x16x32 with k = 8 g + 4 t + c, where lane group g holds a PAIR of
// horizontally adjacent taps t = 0, 1 (adjacent pixels of the patch = 16 contiguous bytes):
//     MFMA 0:  g0 = row 0 taps (0, 1)   g1 = row 0 taps (2, -)   g2 = row 1 taps (0, 1)   g3 = row 1 taps (2, -)
//     MFMA 1:  g0 = row 2 taps (0, 1)   g1 = row 2 taps (2, -)   g2, g3 = -                       ("-": zero weights)
// two MFMAs per 16 x 16 output tile, 36 of 64 K slots live.  (First form, r06: one 16x16x16 MFMA per filter row -- 48 instead of 32 MFMAs
// per 64 pixels and the older instruction: 53 us.)
//   * LDS patch [row][column][4 channels]: a pixel is 8 bytes, so the B fragment of a 16-pixel group -- lane (g, column): the four
//     channels of the two pixels (column + 2 (g & 1)), + 1 in row (g >> 1) -- is ONE ds_read2_b64, no im2col;
//   * a wave takes 64 consecutive pixels of one output row as four INTERLEAVED groups (group q holds pixels 4 i + q): the four
//     accumulator sets of a lane are then four consecutive pixels of one channel row -- an 8-byte store per lane, 128 contiguous
//     bytes per 16 lanes, with no transposition through LDS;
//   * the twelve weight fragments (3 filter rows x 4 blocks of 16 output channels) stay in registers for the workgroup's life.
// (A weight-gradient kernel in the same spirit was built and measured -- no faster than the general one: docs/experiments/r06_wgrad_direct4.hip.txt.)
// Bound by its output stream: algorithmic bytes = x + y (SURVEY.md 8d prices the conv by flops; this layer's floor is HBM).
#include <type_traits>
#include "flrelu_mfma_common.h"      // pack2<T>: one v_cvt_pk of exactly a pair

namespace afcm {

struct DirectConvParams {
    const void* x; void* y; const void* wp; const float* oscale; const float* obias;
    int N, Cin, Cout, H, W, P, Q, pad, ldx, ldy, Opad, tilesX, tilesY, bk;
};

constexpr int kDcRows = 32, kDcCols = 64;                 // output tile of a workgroup (16 rows: the per-workgroup set-up -- weights, patch, factors -- was half of a wave's instructions)
constexpr int kDcPR = kDcRows + 2;                        // patch rows (columns that hold data: kDcCols + 2)
constexpr int kDcPW = 68;                                 // patch row pitch in pixels (8 bytes each)

template <typename T> struct DcMfma;
template <> struct DcMfma<bf16_t> {
    typedef __attribute__((ext_vector_type(8))) __bf16 frag;
    static __device__ __forceinline__ __attribute__((ext_vector_type(4))) float mma(frag a, frag b, __attribute__((ext_vector_type(4))) float c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    }
};
template <> struct DcMfma<f16_t> {
    typedef __attribute__((ext_vector_type(8))) _Float16 frag;
    static __device__ __forceinline__ __attribute__((ext_vector_type(4))) float mma(frag a, frag b, __attribute__((ext_vector_type(4))) float c) {
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    }
};

template <typename T>
__global__ __launch_bounds__(256, 2) void conv2d_direct4_kernel(DirectConvParams p) {
    typedef DcMfma<T> M;
    typedef typename M::frag frag;
    typedef __attribute__((ext_vector_type(4))) float f32x4;
    __shared__ __attribute__((aligned(16))) unsigned short patch[kDcPR * kDcPW * 4];
    __shared__ __attribute__((aligned(16))) float scs[64], obs[64];                                       // epilogue factors per output channel, staged once per workgroup

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int n = bid / p.tilesY;
    const int y0 = ty * kDcRows, x0 = tx * kDcCols;

    // ---- weights: A[o = 16 ot + l15][k = 8 g + 4 t + c]: MFMA m, lane group g -> filter row 2 m + (g >> 1), taps 2 (g & 1) + t; taps beyond the
    // filter, rows beyond it and channels >= Cin are zero
    frag wa[2][4];
    {
        const unsigned short* wp = (const unsigned short*)p.wp;
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int ot = 0; ot < 4; ot++) {
                union { uint4 u; frag f; unsigned short h[8]; uint2 d[2]; } v;
                v.u = make_uint4(0u, 0u, 0u, 0u);
                const int o = ot * 16 + l15, r = 2 * m + (g >> 1);
                if (r < 3 && o < p.Opad) {
#pragma unroll
                    for (int t = 0; t < 2; t++) {
                        const int sx = 2 * (g & 1) + t;
                        if (sx < 3) v.d[t] = *(const uint2*)(wp + ((size_t)(r * 3 + sx) * p.Opad + o) * p.bk);
                    }
                }
#pragma unroll
                for (int c = 0; c < 8; c++)
                    if ((c & 3) >= p.Cin) v.h[c] = 0;
                wa[m][ot] = v.f;
            }
    }

    // ---- patch: rows y0 - pad .. + 17, columns x0 - pad .. + 65 of the (at most four) input planes, zero outside the image.  Wave c stages
    // channel c: lane = column (64 of them; lanes 0, 1 also take columns 64, 65), one row per step -- the addresses advance by the row pitch
    // (as a flat index over (channel, row, column) every element paid two divisions by constants: 270 instructions per wave)
    {
        // (buffer loads: an element outside the image -- a row above / below it, a column left / right of it, a channel >= Cin -- gets the
        // out-of-range offset and reads as zero: no branch around a load.  As `cond ? row[ix] : 0` every row was a divergent branch.)
        const int c = wave;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)((const unsigned short*)p.x + (size_t)n * p.Cin * p.H * p.ldx), 0,
                                                                              p.Cin * p.H * p.ldx * 2, 0x00020000);
        const int ix = x0 - p.pad + lane, ix2 = x0 - p.pad + 64 + lane;
        const bool cok = c < p.Cin && (unsigned)ix < (unsigned)p.W, cok2 = c < p.Cin && lane < 2 && (unsigned)ix2 < (unsigned)p.W;
        unsigned short v[kDcPR], v2[kDcPR];
#pragma unroll
        for (int pr = 0; pr < kDcPR; pr++) {
            const int iy = y0 - p.pad + pr;
            const bool rok = (unsigned)iy < (unsigned)p.H;
            const unsigned base = (unsigned)((c * p.H + iy) * p.ldx) * 2u;
            v[pr] = __builtin_amdgcn_raw_buffer_load_b16(xrs, (rok && cok) ? base + (unsigned)ix * 2u : 0x80000000u, 0, 0);
            v2[pr] = __builtin_amdgcn_raw_buffer_load_b16(xrs, (rok && cok2) ? base + (unsigned)ix2 * 2u : 0x80000000u, 0, 0);
        }
#pragma unroll
        for (int pr = 0; pr < kDcPR; pr++) {
            patch[(pr * kDcPW + lane) * 4 + c] = v[pr];
            if (lane < 2) patch[(pr * kDcPW + 64 + lane) * 4 + c] = v2[pr];
        }
    }
    if (tid < 64) {
        scs[tid] = (p.oscale != nullptr && tid < p.Cout) ? p.oscale[(size_t)n * p.Cout + tid] : 1.f;
        obs[tid] = (p.obias != nullptr && tid < p.Cout) ? p.obias[tid] : 0.f;
    }
    __syncthreads();

    const int qlim = min(p.ldy, (p.Q + 7) & ~7);                              // columns written: up to the granule past Q (the contract of afcm_conv2d_ld)
    const int plane = p.P * p.ldy;                                            // elements per output channel (Cout planes < 2^30 bytes: host)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc((void*)((unsigned short*)p.y + (size_t)n * p.Cout * plane), 0, p.Cout * plane * 2, 0x00020000);
    // this lane's window inside a patch row pair: row (g >> 1) of the MFMA's two filter rows, pixels + 2 (g & 1) and + 2 (g & 1) + 1 -- except that
    // the slot behind tap 2 (zero weights) re-reads tap 2's pixel: no element outside the 3 x 3 window is ever multiplied (0 x NaN).  Two lane
    // offsets made once (opaque: as `base + select` the compiler read 16 bytes and chose per lane -- 120 vector instructions per 64 pixels)
    const int lrow = g >> 1, lcol = 2 * (g & 1);
    int off_a = (4 * l15 + lcol) * 4, off_b = off_a + ((g & 1) ? 0 : 4);    // elements
    asm volatile("" : "+v"(off_a), "+v"(off_b));
#pragma unroll 1
    for (int i = 0; i < kDcRows / 4; i++) {
        const int pr0 = wave * (kDcRows / 4) + i;
        const int oy = y0 + pr0;
        if (oy >= p.P) break;                                                 // (wave-uniform)
        f32x4 acc[4][4];
#pragma unroll
        for (int m = 0; m < 2; m++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                union { uint2 d[2]; frag f; } b;
                // (MFMA 1, lane groups 2, 3: zero weights -- they re-read the row of groups 0, 1: inside the patch)
                const int prow = pr0 + 2 * m + (m == 1 ? 0 : lrow);
                const unsigned short* src = patch + (prow * kDcPW + q) * 4;
                b.d[0] = *(const uint2*)(src + off_a);
                b.d[1] = *(const uint2*)(src + off_b);
#pragma unroll
                for (int ot = 0; ot < 4; ot++)                                // (the first product starts from a literal zero: no accumulator clearing)
                    acc[ot][q] = M::mma(wa[m][ot], b.f, m == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : acc[ot][q]);
            }
        // stores through a buffer descriptor over this image's output: the lane offset is made once per row, the channel rides in the scalar
        // offset, and rows o >= Cout fall behind the descriptor's end (dropped) -- per-store 64-bit address arithmetic and a branch per channel
        // were 595 vector instructions per 64 pixels (r06, first form)
        const int ox = x0 + 4 * l15;
        const int npx = ox + 4 <= qlim ? 4 : (ox + 2 <= qlim ? 2 : 0);      // (even widths: a dense row may end on a pixel pair)
        const unsigned voff = (unsigned)(((4 * g) * plane + oy * p.ldy + ox) * 2);
        const unsigned v64 = npx == 4 ? voff : 0x80000000u, v32 = npx == 2 ? voff : 0x80000000u;
        const bool any2 = __builtin_amdgcn_ballot_w64(npx == 2) != 0;        // (wave-uniform: the row's last column tile only)
#pragma unroll
        for (int ot = 0; ot < 4; ot++) {
            // this lane's rows o = 16 ot + 4 g + reg: their factors as two 16-byte LDS reads (in registers for the kernel's life they cost the
            // third wave per SIMD)
            const f32x4 sc4 = *(const f32x4*)(scs + ot * 16 + 4 * g), ob4 = *(const f32x4*)(obs + ot * 16 + 4 * g);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
                u32x2 wv;
                wv.x = pack2<T>(__builtin_fmaf(acc[ot][0][reg], sc4[reg], ob4[reg]), __builtin_fmaf(acc[ot][1][reg], sc4[reg], ob4[reg]));
                wv.y = pack2<T>(__builtin_fmaf(acc[ot][2][reg], sc4[reg], ob4[reg]), __builtin_fmaf(acc[ot][3][reg], sc4[reg], ob4[reg]));
                const int soff = (ot * 16 + reg) * plane * 2;
                __builtin_amdgcn_raw_buffer_store_b64(wv, yrs, v64, soff, 0);
                if (any2) __builtin_amdgcn_raw_buffer_store_b32(wv.x, yrs, v32, soff, 0);
            }
        }
    }
}

// host side: called by afcm_conv2d_ld for 16-bit 3x3 convs with cin <= 4, cout <= 64 (declared there)
int conv2d_direct_small_cin(const void* x, void* y, const void* wp, const float* oscale, const float* obias, int dtype, int n, int cin, int cout,
                            int h, int w, int pad, int rows_pad, int bk, int ldx, int ldy, hipStream_t st) {
    DirectConvParams p;
    p.x = x; p.y = y; p.wp = wp; p.oscale = oscale; p.obias = obias;
    p.N = n; p.Cin = cin; p.Cout = cout; p.H = h; p.W = w; p.pad = pad;
    p.P = h + 2 * pad - 2; p.Q = w + 2 * pad - 2;
    p.ldx = ldx; p.ldy = ldy; p.Opad = rows_pad; p.bk = bk;
    p.tilesX = cdiv(p.Q, kDcCols); p.tilesY = cdiv(p.P, kDcRows);
    const long long blocks = (long long)p.tilesX * p.tilesY * n;
    if (blocks <= 0 || blocks >= (1ll << 31)) return AFCM_E_NOKERNEL;
    if (dtype == AFCM_F16) hipLaunchKernelGGL((conv2d_direct4_kernel<f16_t>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv2d_direct4_kernel<bf16_t>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    return hip_status(hipGetLastError());
}


}  // namespace afcm
