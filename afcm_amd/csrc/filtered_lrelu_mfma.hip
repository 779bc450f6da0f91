// filtered_lrelu on the matrix cores, for 16-bit activations (bf16 / f16 storage).
//
// Why MFMA here: on gfx950 the fp32 vector pipe peaks at ~63 T lane-FMA/s (measured, tools/ubench/valu_rate.hip;
// packed and dot2 forms run at half the instruction rate, so there is no faster VALU form), and this op needs
// ~11 G FMA per 256^2 image against 0.62 GB of 16-bit traffic: on the vector pipe it is compute-bound at ~2x the
// HBM time before any overhead.  The four separable FIR passes are banded-Toeplitz matrix products; run as
// v_mfma_f32_16x16x32_{bf16,f16} they cost ~0.25 CU-cycles per output pixel even at ~19-25 % band occupancy.
//
// Dataflow per tile (fp32 accumulation everywhere, 16-bit operands):
//   In (LDS, [row][col])  --A-->  X1 = In * UH        up-FIR along x   D[in-row][ucol]
//   X1 (registers)        --B-->  X2 = UV * X1        up-FIR along y   D[urow][ucol]   (accumulator tile used as
//   act(X2) (registers)   --B-->  X3 = DV * act(X2)   down-FIR along y D[orow][ucol]    the next B operand: no LDS)
//   X3 (LDS, [ucol][orow]) -tr->  Y  = X3 * DH        down-FIR along x D[orow][ocol]   (ds_read_b64_tr_b16)
// UH/UV/DV/DH are constant Toeplitz fragments built once per layer by flrelu_mfma_prepare_kernel.
// Only In and X3 touch LDS; the up^2-times-larger activated intermediate lives in accumulators.
//
// Sign codes use a private "row-quad" layout: one byte = the codes of 4 consecutive rows of one column
// ([N*C][ceil(sh/4)][swq]); forward writes them from the X2 accumulator layout, backward (the same kernel with
// up/down swapped) reads them with a funnel shift for the row offset.
#include <type_traits>

#include "common.h"

namespace afcm {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 mbf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 mf16x8;
typedef __attribute__((ext_vector_type(4))) short s16x4;

struct FlreluMfmaParams {
    const void* x;
    void* y;
    const void* b;
    unsigned char* s;
    const void* ws;        // constant fragments
    float* plane_sum;      // optional fp32 [N*C][tilesX*tilesY]: per-tile sums of this launch's outputs (bias gradient without a second pass)
    int xw, xh, yw, yh, C;
    int px0, py0;
    int tilesX, tilesY;
    float slope, clamp;
    int sx, sy, shq, swq;  // sign tensor: rows of quads, bytes per row
};

constexpr int kFUT = 6;   // taps per polyphase branch of the up filter (filter_size of the model)

// Geometry shared by the kernel, the prepare kernel and the host.
template <int UP, int DOWN, int TOW, int TOH>
struct MfmaGeom {
    static constexpr int FU = kFUT * UP, FD = kFUT * DOWN;
    static constexpr int TUW = (TOW - 1) * DOWN + FD, TUH = (TOH - 1) * DOWN + FD;
    static constexpr int NB = 3, GW = 16 * NB;                 // ucol blocks per group; every group shares one 32-wide input window
    static constexpr int NG = cdiv(cdiv(TUW, 16), NB);         // groups (= waves)
    static constexpr int NVB = cdiv(TUH, 16);                  // urow blocks that carry data
    static constexpr int NOB = TOH / 16, NCB = TOW / 16;
    static constexpr int NDVK = DOWN == 2 ? 2 : 3;             // 32-wide K windows of a down pass
    static constexpr int NPAIR = (DOWN / 2) * (NOB - 1) + NDVK;  // packed X2 tile pairs the down-V pass touches
    static constexpr int NMB = cdiv((16 / UP) * (NVB - 1) + 16 / UP + kFUT, 16);   // X1 row blocks that carry data
    static constexpr int NQ = ((NVB - 1) / UP) + 1;            // packed X1 tile pairs (mb0 = vb / UP)
    static constexpr int IROWS = 16 * NMB;
    static constexpr int IWSTEP = GW / UP;                     // input-window advance per group
    static constexpr int TIW = IWSTEP * (NG - 1) + 32;
    static constexpr int PIN = ((TIW + 7) / 8) * 8 + ((((TIW + 7) / 8) % 2 == 0) ? 8 : 0);   // 16-B units, odd count
    static constexpr int XCOLS = NG * GW;                      // computed upsampled columns
    static constexpr int PX3 = TOH + 4;                        // X3T pitch (elements)
    static constexpr int X3ROWS = cmax(NG * GW, 16 * DOWN * (NCB - 1) + 32 * NDVK);   // the last K window of down-x overhangs
    static constexpr int NFRAG = NB + UP + 2 * NDVK;           // UH[NB] UV[UP] DV[NDVK] DH[NDVK]
    static constexpr int SGN_ROWS = 4 * NVB + 1;               // staged sign quad-rows (READ)
    static constexpr int SGN_WORDS = XCOLS / 4 + 1;            // aligned dwords covering one staged sign row
    static constexpr int SGN_PITCH = 4 * SGN_WORDS + 12;
    static_assert(TOW % 16 == 0 && TOH % 16 == 0 && (TOW * DOWN) % UP == 0 && (TOH * DOWN) % UP == 0, "tile shape");
    static_assert((TOH * DOWN) % 4 == 0, "sign quads");
};

// row index inside a 32-row K window carried by fragment element (g, j): two stacked accumulator tiles
__host__ __device__ __forceinline__ int krow(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

// ---------------------------------------------------------------------------------------------
// Constant fragments.  Layout: [frag][lane][8] elements of T.
template <typename T, int UP, int DOWN, int TOW, int TOH>
__global__ void flrelu_mfma_prepare_kernel(T* __restrict__ ws, const float* __restrict__ fu, const float* __restrict__ fd,
                                           int px0, int py0, int flip, float gain_total) {
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    const int phx = pos_mod(px0, UP), phy = pos_mod(py0, UP);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < G::NFRAG * 64 * 8; idx += gridDim.x * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
        const int l15 = lane & 15, g = lane >> 4;
        float v = 0.f;
        if (frag < G::NB) {
            // UH[nb]: B[k][n], k = 8g + j (input column in the window), n = l15 (ucol 16nb + n of the group)
            const int k = 8 * g + j, urel = 16 * frag + l15;
            const int a = urel % UP, o = (a > phx) ? 1 : 0;
            const int kmin = o ? UP - (a - phx) : phx - a;
            const int jj = k - urel / UP - o;
            if (jj >= 0 && jj < kFUT) {
                const int tap = kmin + UP * jj;
                v = flip ? fu[tap] : fu[G::FU - 1 - tap];
            }
        } else if (frag < G::NB + UP) {
            // UV[var]: A[n][k], n = l15 (urow 16vb + n), k -> input row krow(g, j) of the 32-row window starting at 16*(vb/UP)
            const int var = frag - G::NB, n = l15;
            const int a = n % UP, o = (a > phy) ? 1 : 0;
            const int kmin = o ? UP - (a - phy) : phy - a;
            const int jj = krow(g, j) - (16 / UP) * var - n / UP - o;
            if (jj >= 0 && jj < kFUT) {
                const int tap = kmin + UP * jj;
                v = (flip ? fu[tap] : fu[G::FU - 1 - tap]) * gain_total;
            }
        } else if (frag < G::NB + UP + G::NDVK) {
            // DV[t]: A[n][k], n = l15 (orow), k -> urow 32t + krow(g, j) of the window starting at 16*DOWN*ob
            const int t = frag - G::NB - UP;
            const int kk = 32 * t + krow(g, j) - DOWN * l15;
            if (kk >= 0 && kk < G::FD) v = flip ? fd[kk] : fd[G::FD - 1 - kk];
        } else {
            // DH[t]: B[k][n], k = 8g + j natural (ucol 32t + k of the window starting at 16*DOWN*cb), n = l15 (ocol)
            const int t = frag - G::NB - UP - G::NDVK;
            const int kk = 32 * t + 8 * g + j - DOWN * l15;
            if (kk >= 0 && kk < G::FD) v = flip ? fd[kk] : fd[G::FD - 1 - kk];
        }
        ws[idx] = from_f32<T>(v);
    }
}

template <typename T> struct MfmaOps;
template <> struct MfmaOps<bf16_t> {
    typedef mbf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaOps<f16_t> {
    typedef mf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

template <typename T>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    union { T t[2]; unsigned u; } r;
    r.t[0] = from_f32<T>(lo);
    r.t[1] = from_f32<T>(hi);
    return r.u;
}

// Two accumulator tiles -> one 8-element B fragment: elements 0-3 from `lo`, 4-7 from `hi` (K order = krow()).
template <typename T>
__device__ __forceinline__ typename MfmaOps<T>::frag pack_pair(const f32x4& lo, const f32x4& hi) {
    union { unsigned u[4]; typename MfmaOps<T>::frag f; } r;
    r.u[0] = pack2<T>(lo[0], lo[1]);
    r.u[1] = pack2<T>(lo[2], lo[3]);
    r.u[2] = pack2<T>(hi[0], hi[1]);
    r.u[3] = pack2<T>(hi[2], hi[3]);
    return r.f;
}

template <typename T, int UP, int DOWN, int TOW, int TOH, int SIGN>
__global__ __launch_bounds__((64 * MfmaGeom<UP, DOWN, TOW, TOH>::NG)) void flrelu_mfma_kernel(FlreluMfmaParams p) {
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    typedef MfmaOps<T> M;
    typedef typename M::frag frag;
    constexpr int NT = 64 * G::NG;
    constexpr int SGN_BYTES = (SIGN == AFCM_SIGNS_READ) ? G::SGN_ROWS * G::SGN_PITCH : 0;
    __shared__ __attribute__((aligned(16))) T lds_in[G::IROWS * G::PIN];
    __shared__ __attribute__((aligned(16))) T lds_x3[G::X3ROWS * G::PX3];
    __shared__ __attribute__((aligned(16))) unsigned char lds_sg[SGN_BYTES > 0 ? SGN_BYTES : 16];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    int bid = blockIdx.x;
    const int tx = bid % p.tilesX; bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int plane = bid / p.tilesY;
    const int O0x = tx * TOW, O0y = ty * TOH;
    const int U0x = O0x * DOWN, U0y = O0y * DOWN;
    const int I0x = -floor_div(p.px0 - U0x, UP), I0y = -floor_div(p.py0 - U0y, UP);

    // ---- stage the input tile (+ bias inside the image; zero outside: the bias is added before padding).
    // One item = 8 consecutive columns of one row: 4 (or 5, odd column origin) aligned dword loads, realigned in
    // registers, written with one 16-byte LDS store.  Needs an even plane width (all layers of the model).
    {
        const T* xp = (const T*)p.x + (size_t)plane * p.xh * p.xw;
        const float bias = p.b ? to_f32(((const T*)p.b)[plane % p.C]) : 0.f;
        constexpr int CH = G::PIN / 8;                       // 8-column chunks per staged row (incl. pitch padding)
        constexpr int NIT = cdiv(G::IROWS * CH, NT);
        const int odd = I0x & 1;
        unsigned raw[NIT][5];
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int idx = tid + i * NT;
            const int r = idx / CH, c8 = idx - r * CH;
            const int iy = I0y + r;
            const int ix0 = I0x - odd + 8 * c8;              // even column of the first dword
            const bool rok = idx < G::IROWS * CH && (unsigned)iy < (unsigned)p.xh;
            const T* src = xp + (long long)iy * p.xw + ix0;
#pragma unroll
            for (int w = 0; w < 5; w++)
                raw[i][w] = (rok && (w < 4 || odd) && (unsigned)(ix0 + 2 * w) < (unsigned)p.xw) ? *(const unsigned*)(src + 2 * w) : 0u;
        }
        // rows of X3T beyond the computed columns are read (with zero weights) by the last down-x window: keep them finite
        for (int idx = tid; idx < (G::X3ROWS - G::XCOLS) * G::PX3; idx += NT) lds_x3[G::XCOLS * G::PX3 + idx] = from_f32<T>(0.f);
        if (SIGN == AFCM_SIGNS_READ) {
            // sign window: quad-rows [(U0y+sy)>>2, +SGN_ROWS), columns [U0x+sx, +XCOLS), fetched as aligned dwords
            const unsigned char* sp = p.s + (size_t)plane * p.shq * p.swq;
            const int qy0 = (U0y + p.sy) >> 2, w0 = (U0x + p.sx) >> 2, wpr = p.swq >> 2;
            constexpr int NSW = cdiv(G::SGN_ROWS * G::SGN_WORDS, NT);
            unsigned sv[NSW];
#pragma unroll
            for (int i = 0; i < NSW; i++) {
                const int idx = tid + i * NT;
                const int r = idx / G::SGN_WORDS, c = idx - r * G::SGN_WORDS;
                const int qy = qy0 + r, wi = w0 + c;
                sv[i] = (idx < G::SGN_ROWS * G::SGN_WORDS && (unsigned)qy < (unsigned)p.shq && (unsigned)wi < (unsigned)wpr)
                            ? ((const unsigned*)(sp + (size_t)qy * p.swq))[wi] : 0u;
            }
#pragma unroll
            for (int i = 0; i < NSW; i++) {
                const int idx = tid + i * NT;
                if (idx < G::SGN_ROWS * G::SGN_WORDS) {
                    const int r = idx / G::SGN_WORDS, c = idx - r * G::SGN_WORDS;
                    *(unsigned*)(lds_sg + r * G::SGN_PITCH + 4 * c) = sv[i];
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int idx = tid + i * NT;
            if (idx < G::IROWS * CH) {
                const int r = idx / CH, c8 = idx - r * CH;
                const int iy = I0y + r;
                const int ix0 = I0x + 8 * c8;
                const bool rok = (unsigned)iy < (unsigned)p.xh;
                unsigned o[4];
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const unsigned d = odd ? __builtin_amdgcn_alignbyte(raw[i][w + 1], raw[i][w], 2) : raw[i][w];
                    union { unsigned u; T t[2]; } e;
                    e.u = d;
                    const bool in0 = rok && (unsigned)(ix0 + 2 * w) < (unsigned)p.xw;
                    const bool in1 = rok && (unsigned)(ix0 + 2 * w + 1) < (unsigned)p.xw;
                    o[w] = pack2<T>(in0 ? to_f32(e.t[0]) + bias : 0.f, in1 ? to_f32(e.t[1]) + bias : 0.f);
                }
                *(uint4*)(lds_in + r * G::PIN + 8 * c8) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        }
    }
    __syncthreads();

    const frag* wsf = (const frag*)p.ws;
    auto cfrag = [&](int f) __attribute__((always_inline)) { return wsf[f * 64 + lane]; };
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- phase A: one group of NB ucol blocks per wave: up-x, up-y, activation, down-y, all in registers
    {
        const int Gi = wave;
        frag a_in[G::NMB];
#pragma unroll
        for (int mb = 0; mb < G::NMB; mb++) {
            const T* src = lds_in + (16 * mb + l15) * G::PIN + G::IWSTEP * Gi + 8 * g;
            union { uint2 h[2]; frag f; } t;                   // 8-byte aligned halves (window start is 8-B aligned for UP=4)
            t.h[0] = *(const uint2*)src;
            t.h[1] = *(const uint2*)(src + 4);
            a_in[mb] = t.f;
        }
        frag uv[UP], dv[G::NDVK];
#pragma unroll
        for (int v = 0; v < UP; v++) uv[v] = cfrag(G::NB + v);
#pragma unroll
        for (int t = 0; t < G::NDVK; t++) dv[t] = cfrag(G::NB + UP + t);
        const bool lastX = (tx == p.tilesX - 1), lastY = (ty == p.tilesY - 1);
        // sign-code store: byte (quad-row (U0y + 16vb + 4g)/4, column U0x + ucol); pointer and validity hoisted out of the loops
        unsigned char* sgl = p.s + ((size_t)plane * p.shq + ((U0y >> 2) + g)) * p.swq + U0x + G::GW * Gi + l15;
        unsigned rowmask = 0;
        if (SIGN == AFCM_SIGNS_WRITE) {
#pragma unroll
            for (int vb = 0; vb < G::NVB; vb++)
                rowmask |= (unsigned)(((16 * vb + 4 * g < TOH * DOWN) || lastY) && ((U0y >> 2) + 4 * vb + g) < p.shq) << vb;
        }

#pragma unroll
        for (int nb = 0; nb < G::NB; nb++) {
            const frag uh = cfrag(nb);
            const int ucol = G::GW * Gi + 16 * nb + l15;       // tile-relative upsampled column of this lane
            const bool colown = ((ucol < TOW * DOWN) || lastX) && (U0x + ucol < p.swq);
            // up-x: X1[mb] = In[mb] * UH
            f32x4 x1[G::NMB];
#pragma unroll
            for (int mb = 0; mb < G::NMB; mb++) x1[mb] = M::mma(a_in[mb], uh, zero4);
            frag q[G::NQ];
#pragma unroll
            for (int m = 0; m < G::NQ; m++) q[m] = pack_pair<T>(x1[m], (m + 1 < G::NMB) ? x1[m + 1] : zero4);
            // up-y + activation, packed pairwise for the down-y pass
            frag pr[G::NPAIR];
            f32x4 held = zero4;
#pragma unroll
            for (int vb = 0; vb < 2 * G::NPAIR; vb++) {
                f32x4 x2 = zero4;
                if (vb < G::NVB) {
                    x2 = M::mma(uv[vb % UP], q[vb / UP], zero4);
                    unsigned codes = 0;
                    if (SIGN == AFCM_SIGNS_READ) {
                        // codes of rows Y+sy .. Y+sy+3 at column X+sx: two staged quad bytes, funnel-shifted
                        const int yy = (U0y + p.sy) & 3;       // row offset inside the first staged quad
                        const int qr = (16 * vb + 4 * g + yy) >> 2, sh = ((16 * vb + 4 * g + yy) & 3) << 1;
                        const int coff = (U0x + p.sx) & 3;    // column of the window start inside its first aligned dword
                        const unsigned lo = lds_sg[qr * G::SGN_PITCH + ucol + coff];
                        const unsigned hi = lds_sg[(qr + 1) * G::SGN_PITCH + ucol + coff];
                        codes = ((lo | (hi << 8)) >> sh) & 0xffu;
                    }
                    unsigned wcode = 0;
                    if (SIGN == AFCM_SIGNS_READ) {
                        // spread the four 2-bit codes to one per byte: byte r of `d` = code of row r
                        const unsigned d = (codes * 0x00041041u) & 0x03030303u;
                        if (__builtin_amdgcn_ballot_w64((d & 0x02020202u) != 0) == 0) {
                            // common case, no clamped element in the tile: factor = 1 + bit0 * (slope - 1)
                            const float sm1 = p.slope - 1.f;
                            x2[0] *= fmaf((float)((d >> 0) & 0xffu), sm1, 1.f);
                            x2[1] *= fmaf((float)((d >> 8) & 0xffu), sm1, 1.f);
                            x2[2] *= fmaf((float)((d >> 16) & 0xffu), sm1, 1.f);
                            x2[3] *= fmaf((float)((d >> 24) & 0xffu), sm1, 1.f);
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const unsigned c = codes >> (2 * r);
                                float v = x2[r];
                                if (c & 1u) v *= p.slope;
                                if (c & 2u) v = 0.f;
                                x2[r] = v;
                            }
                        }
                    } else {
                        const float amax = fmaxf(fmaxf(fabsf(x2[0]), fabsf(x2[1])), fmaxf(fabsf(x2[2]), fabsf(x2[3])));
                        if (p.slope <= 1.f && __builtin_amdgcn_ballot_w64(amax > p.clamp) == 0) {
                            // common case: nothing in the tile can reach the clamp (|lrelu(v)| <= |v| for slope <= 1)
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const float v = x2[r];
                                wcode |= (__float_as_uint(v) >> 31) << (2 * r);
                                x2[r] = fmaxf(v, v * p.slope);
                            }
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                float v = x2[r];
                                unsigned c = __float_as_uint(v) >> 31;
                                if (c) v *= p.slope;
                                if (fabsf(v) > p.clamp) { c = 2u; v = (v < 0.f) ? -p.clamp : p.clamp; }
                                wcode |= c << (2 * r);
                                x2[r] = v;
                            }
                        }
                    }
                    if (SIGN == AFCM_SIGNS_WRITE) {
                        if (colown && ((rowmask >> vb) & 1)) sgl[(size_t)(4 * vb) * p.swq + 16 * nb] = (unsigned char)wcode;
                    }
                }
                if (vb & 1) pr[vb >> 1] = pack_pair<T>(held, x2);
                else held = x2;
            }
            // down-y: X3[ob] = sum_t DV[t] * P[(DOWN/2)*ob + t]  ->  LDS X3T[ucol][orow]
#pragma unroll
            for (int ob = 0; ob < G::NOB; ob++) {
                f32x4 x3 = zero4;
#pragma unroll
                for (int t = 0; t < G::NDVK; t++) x3 = M::mma(dv[t], pr[(DOWN / 2) * ob + t], x3);
                uint2 w;
                w.x = pack2<T>(x3[0], x3[1]);
                w.y = pack2<T>(x3[2], x3[3]);
                *(uint2*)(lds_x3 + ucol * G::PX3 + 16 * ob + 4 * g) = w;
            }
        }
    }
    __syncthreads();

    // ---- phase B: down-x through LDS (transposed reads), then store
    {
        frag dh[G::NDVK];
#pragma unroll
        for (int t = 0; t < G::NDVK; t++) dh[t] = cfrag(G::NB + UP + G::NDVK + t);
        const int q4 = l15 >> 2, p4 = l15 & 3;
        // this lane's output pointer at (row O0y + 4g, column O0x + l15); units advance it by constants
        T* ylane = (T*)p.y + (size_t)plane * p.yh * p.yw + (size_t)(O0y + 4 * g) * p.yw + O0x + l15;
        const T* xlane = lds_x3 + (8 * g + q4) * G::PX3 + 4 * p4;
        float psum = 0.f;
        for (int unit = wave; unit < G::NOB * G::NCB; unit += G::NG) {
            const int ob = unit / G::NCB, cb = unit - ob * G::NCB;
            f32x4 acc = zero4;
#pragma unroll
            for (int t = 0; t < G::NDVK; t++) {
                const T* a0 = xlane + (16 * DOWN * cb + 32 * t) * G::PX3 + 16 * ob;
                union { s16x4 h[2]; frag f; } a;
                a.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a0);
                a.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a0 + 4 * G::PX3));
                acc = M::mma(a.f, dh[t], acc);
            }
            const bool colok = O0x + 16 * cb + l15 < p.yw;
            const int rows_left = p.yh - (O0y + 16 * ob + 4 * g);        // rows of this lane's 4 that are inside the image
            T* dst = ylane + (size_t)(16 * ob) * p.yw + 16 * cb;
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (colok && r < rows_left) {
                    const T o = from_f32<T>(acc[r]);
                    dst[(size_t)r * p.yw] = o;
                    psum += to_f32(o);
                }
        }
        if (p.plane_sum != nullptr) {
            // wave reduction -> workgroup reduction through LDS -> one plain store into this tile's slot (no atomics:
            // deterministic, and no same-address contention between the tiles of a plane)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) psum += __shfl_down(psum, off, 64);
            __syncthreads();                                   // phase-B reads of lds_x3 are finished
            float* red = (float*)lds_x3;
            if (lane == 0) red[wave] = psum;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < G::NG; w++) t += red[w];
                p.plane_sum[(size_t)plane * (p.tilesX * p.tilesY) + ty * p.tilesX + tx] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int UP, int DOWN> struct MfmaTile;
template <> struct MfmaTile<2, 2> { static constexpr int TOW = 64, TOH = 32; };
template <> struct MfmaTile<2, 4> { static constexpr int TOW = 32, TOH = 32; };
template <> struct MfmaTile<4, 2> { static constexpr int TOW = 64, TOH = 32; };

template <typename T, int UP, int DOWN>
static int launch_mfma(const afcm_filtered_lrelu_args* a, hipStream_t st) {
    constexpr int TOW = MfmaTile<UP, DOWN>::TOW, TOH = MfmaTile<UP, DOWN>::TOH;
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    FlreluMfmaParams p;
    p.x = a->x; p.y = a->y; p.b = a->b; p.s = a->signs; p.ws = a->workspace; p.plane_sum = a->plane_sum;
    p.xw = a->xw; p.xh = a->xh; p.yw = a->yw; p.yh = a->yh; p.C = a->c;
    p.px0 = a->px0; p.py0 = a->py0;
    p.tilesX = cdiv(a->yw, TOW); p.tilesY = cdiv(a->yh, TOH);
    p.slope = a->slope; p.clamp = a->clamp;
    p.sx = a->sx; p.sy = a->sy; p.shq = a->sh; p.swq = a->swb;
    const long long blocks = (long long)p.tilesX * p.tilesY * a->n * a->c;
    AFCM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "filtered_lrelu: grid of %lld blocks is out of range", blocks);
    dim3 grid((unsigned)blocks), block(64 * G::NG);
    switch (a->sign_mode) {
        case AFCM_SIGNS_NONE: hipLaunchKernelGGL((flrelu_mfma_kernel<T, UP, DOWN, TOW, TOH, AFCM_SIGNS_NONE>), grid, block, 0, st, p); break;
        case AFCM_SIGNS_WRITE: hipLaunchKernelGGL((flrelu_mfma_kernel<T, UP, DOWN, TOW, TOH, AFCM_SIGNS_WRITE>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((flrelu_mfma_kernel<T, UP, DOWN, TOW, TOH, AFCM_SIGNS_READ>), grid, block, 0, st, p); break;
    }
    return hip_status(hipGetLastError());
}

template <typename T, int UP, int DOWN>
static int prepare_mfma(const afcm_filtered_lrelu_args* a, hipStream_t st) {
    constexpr int TOW = MfmaTile<UP, DOWN>::TOW, TOH = MfmaTile<UP, DOWN>::TOH;
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    const float gain_total = (float)a->up * (float)a->up * a->gain;
    hipLaunchKernelGGL((flrelu_mfma_prepare_kernel<T, UP, DOWN, TOW, TOH>), dim3(cdiv(G::NFRAG * 512, 256)), dim3(256), 0, st,
                       (T*)a->workspace, a->fu, a->fd, a->px0, a->py0, a->flip_filter, gain_total);
    return hip_status(hipGetLastError());
}

static int mfma_case(const afcm_filtered_lrelu_args* a) {
    if (a->dtype != AFCM_BF16 && a->dtype != AFCM_F16) return 0;
    if (a->fuh != 0 || a->fdh != 0) return 0;
    if (a->xw & 1) return 0;                   // the staged loads are aligned dword pairs: even plane width (all layers of the model)
    if (a->up == 2 && a->down == 2 && a->fuw == 12 && a->fdw == 12) return 22;
    if (a->up == 2 && a->down == 4 && a->fuw == 12 && a->fdw == 24) return 24;
    if (a->up == 4 && a->down == 2 && a->fuw == 24 && a->fdw == 12) return 42;
    return 0;
}

int flrelu_mfma_supported(const afcm_filtered_lrelu_args* a) { return mfma_case(a) != 0; }

int flrelu_mfma_tiles(const afcm_filtered_lrelu_args* a) {
    switch (mfma_case(a)) {
        case 22: return cdiv(a->yw, MfmaTile<2, 2>::TOW) * cdiv(a->yh, MfmaTile<2, 2>::TOH);
        case 24: return cdiv(a->yw, MfmaTile<2, 4>::TOW) * cdiv(a->yh, MfmaTile<2, 4>::TOH);
        case 42: return cdiv(a->yw, MfmaTile<4, 2>::TOW) * cdiv(a->yh, MfmaTile<4, 2>::TOH);
        default: return 0;
    }
}

int flrelu_mfma_launch(const afcm_filtered_lrelu_args* a, bool prepare, hipStream_t st) {
#define AFCM_MF(T) do { switch (mfma_case(a)) { \
        case 22: return prepare ? prepare_mfma<T, 2, 2>(a, st) : launch_mfma<T, 2, 2>(a, st); \
        case 24: return prepare ? prepare_mfma<T, 2, 4>(a, st) : launch_mfma<T, 2, 4>(a, st); \
        case 42: return prepare ? prepare_mfma<T, 4, 2>(a, st) : launch_mfma<T, 4, 2>(a, st); \
        default: return AFCM_E_NOKERNEL; } } while (0)
    if (a->dtype == AFCM_BF16) AFCM_MF(bf16_t);
    else AFCM_MF(f16_t);
#undef AFCM_MF
}

}  // namespace afcm

extern "C" int64_t afcm_filtered_lrelu_workspace_bytes(void) { return 16 * 64 * 8 * 2; }   // >= NFRAG (<= 13) fragments of 64 x 8 x 2 B
