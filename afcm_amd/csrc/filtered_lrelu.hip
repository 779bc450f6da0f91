// filtered_lrelu for gfx950 (MI355X): bias -> zero-insert upsample -> pad/crop -> separable FIR(fu)
// -> * up^2 * gain -> leaky ReLU -> clamp (+ 2-bit sign codes) -> separable FIR(fd) -> decimate, fused in
// one kernel so the up^2-times-larger intermediate never leaves the CU.
//
// Semantics follow the reference op (SG3OPS/filtered_lrelu.py:121-153; edge rules of
// SG3OPS/filtered_lrelu.cu:264-297, 484-505, 564-571).  The kernel structure is new: wave64
// workgroups, one output tile per workgroup staged through LDS in five register-blocked passes
// (load+bias, up-FIR along x, up-FIR along y + activation + sign codes, down-FIR along x,
// down-FIR along y + store).  Filter taps are expanded into per-phase polyphase tables in LDS by the
// kernel itself -- no global filter buffer, so launches on different streams never interfere.
//
// Polyphase indexing (derivation in DESIGN.md): for a tile whose upsampled origin is U0,
//   d = U0 - px0,  I0 = ceil(d / up),  ph = up*I0 - d  in [0, up)
//   u[U0 + up*m + a] = sum_j F[kmin(a) + up*j] * x[I0 + m + o(a) + j]
//   o(a) = (a > ph),  kmin(a) = o(a) ? up - (a - ph) : ph - a,   F = flip ? fu : reversed(fu)
#include <stdlib.h>
#include <type_traits>
#include "common.h"

namespace afcm {

struct FlreluParams {
    const void* x;
    void* y;
    const void* b;
    unsigned char* s;
    int xw, xh, yw, yh, C;
    int px0, py0;
    int tilesX, tilesY;
    float gain;  // up^2 * gain, formed in fp32 like filtered_lrelu.cu:484
    float slope, clamp;
    int flip;
    int sx, sy, sh, swb;
    float fscale;  // pointwise kernel only: product of the 1x1 filters
    int planes;    // strip kernel only: N * C
};

// ---------------------------------------------------------------------------------------------
// Activation on one element of the upsampled grid.  Returns the 2-bit code in WRITE mode.
template <int SIGN>
__device__ __forceinline__ unsigned act_elem(float& v, float gain, float slope, float clamp, unsigned code_in) {
    v *= gain;
    if (SIGN == AFCM_SIGNS_READ) {
        if (code_in & 1u) v *= slope;
        if (code_in & 2u) v = 0.f;
        return 0u;
    }
    unsigned code = __float_as_uint(v) >> 31;
    if (code) v *= slope;
    if (fabsf(v) > clamp) {
        code = 2u;
        v = (v < 0.f) ? -clamp : clamp;
    }
    return code;
}

// Fetch the packed codes of 4 consecutive elements starting at sign coordinate (X, Y); elements
// outside the tensor read as code 0 (value passes through unchanged).
__device__ __forceinline__ unsigned fetch_codes4(const unsigned char* __restrict__ srow_base, int X, int Y, int sh, int swb) {
    if ((unsigned)Y >= (unsigned)sh) return 0u;
    const unsigned char* row = srow_base + (size_t)Y * swb;
    int b0 = X >> 2;  // arithmetic shift: floor for negative X
    unsigned lo = ((unsigned)b0 < (unsigned)swb) ? row[b0] : 0u;
    unsigned hi = ((unsigned)(b0 + 1) < (unsigned)swb) ? row[b0 + 1] : 0u;
    return ((lo | (hi << 8)) >> ((X & 3) << 1)) & 0xffu;
}

__device__ __forceinline__ int quad_or(int v) {
    // OR-reduce over the 4 lanes of a quad with two DPP quad_perm moves ([1,0,3,2] then [2,3,0,1]).
    v |= __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
    v |= __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
    return v;
}

// ---------------------------------------------------------------------------------------------
template <typename T, int UP, int DOWN, int FUT, int FD, int TOW, int TOH, int RO, int NT, int SIGN>
struct FlreluTile {
    static constexpr int FU = UP * FUT;
    static constexpr int TUW = (TOW - 1) * DOWN + FD;      // upsampled columns the tile's outputs need
    static constexpr int TUH = (TOH - 1) * DOWN + FD;
    static constexpr int TUWP = round_up(TUW, 16);         // computed/pitched width (16 = one sign dword)
    static constexpr int ROWS_C = 8;                       // upsampled rows per stage-C item
    static constexpr int TUHP = round_up(TUH, ROWS_C);
    static constexpr int MB = ROWS_C / UP;                 // input-row steps per stage-C item
    static constexpr int TIW = TUWP / UP + FUT;
    // LDS row pitches are odd multiples of 4 floats (16 B): lanes that walk down consecutive rows at a fixed
    // column then hit 16 distinct 16-byte slots per ds_read_b128 / ds_write_b128 lane group (conflict-free).
    static constexpr int TIWP = odd4(round_up(TIW + 2, 4));     // sIn pitch (+2: stage B reads 12 floats per item)
    static constexpr int PU = odd4(TUWP);                       // upX / upXY pitch
    static constexpr int PD = odd4(TOW);                        // downX pitch
    static constexpr int TIH = TUHP / UP + FUT;
    static constexpr int SZ_A = cmax(TIH * TIWP, TUHP * PU);    // sIn, later upXY
    static constexpr int SZ_B = cmax(TIH * PU, TUH * PD);       // upX, later downX
    static constexpr int NCOEF = 2 * FU + FD;
    // READ mode: the tile's window of the sign tensor, staged as dwords (16 codes each): per row the
    // dwords covering columns [U0x + sx, U0x + sx + TUWP) -- TUWP/16 + 1 of them because sx is arbitrary.
    static constexpr int SGN_W = TUWP / 16 + 1;
    static constexpr int SGN_WORDS = (SIGN == AFCM_SIGNS_READ) ? TUHP * SGN_W : 0;
    static constexpr int LDS_FLOATS = SZ_A + SZ_B + round_up(NCOEF, 4) + SGN_WORDS;
    static_assert(ROWS_C % UP == 0 && FUT % 2 == 0 && TOW % 4 == 0 && TOH % RO == 0, "tile shape");
    static_assert((TOW * DOWN) % 16 == 0, "sign ownership must fall on dword boundaries");
    static_assert(DOWN * (TOW - 4) + round_up(DOWN * 3 + FD, 4) <= PU, "stage D over-read must stay inside the row");
    static_assert(LDS_FLOATS * 4 <= 160 * 1024, "LDS overflow");

    // Stage the sign window into LDS.  Dwords outside the tensor read as 0 (= values pass unchanged).
    static __device__ __forceinline__ void stage_signs(unsigned* __restrict__ sgn, const FlreluParams& p, int plane,
                                                       int U0x, int U0y, int tid) {
        const unsigned* splane = (const unsigned*)(p.s + (size_t)plane * p.sh * p.swb);
        const int wpr = p.swb >> 2;                         // dwords per sign row
        const int w0 = (U0x + p.sx) >> 4;                   // floor: arithmetic shift
        constexpr int NW = cdiv(TUHP * SGN_W, NT);
        unsigned v[NW];
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int idx = tid + i * NT;
            const int r = idx / SGN_W, c = idx - r * SGN_W;
            const int Y = U0y + p.sy + r, wi = w0 + c;
            const bool ok = idx < TUHP * SGN_W && (unsigned)Y < (unsigned)p.sh && (unsigned)wi < (unsigned)wpr;
            v[i] = ok ? splane[(size_t)Y * wpr + wi] : 0u;
        }
#pragma unroll
        for (int i = 0; i < NW; i++) {
            const int idx = tid + i * NT;
            if (idx < TUHP * SGN_W) sgn[idx] = v[i];
        }
    }

    // ---- stage B: up-FIR along x.  One item = one input row x 4 input columns -> 4*UP outputs.
    template <int PH>
    static __device__ __forceinline__ void up_x(const float* __restrict__ sIn, float* __restrict__ upX,
                                                const float* __restrict__ cu, int tid) {
        float c[UP][FUT];
#pragma unroll
        for (int a = 0; a < UP; a++)
#pragma unroll
            for (int j = 0; j < FUT; j++) c[a][j] = cu[a * FUT + j];
        constexpr int NCH = TUWP / (4 * UP);
        constexpr int NIN4 = cdiv(4 + FUT, 4);
        for (int item = tid; item < TIH * NCH; item += NT) {
            const int ch = item / TIH, r = item - ch * TIH;     // consecutive lanes -> consecutive rows
            const float* src = sIn + r * TIWP + 4 * ch;
            float in[NIN4 * 4];
#pragma unroll
            for (int i = 0; i < NIN4; i++) {
                float4 t = *(const float4*)(src + 4 * i);
                in[4 * i] = t.x; in[4 * i + 1] = t.y; in[4 * i + 2] = t.z; in[4 * i + 3] = t.w;
            }
            float out[4 * UP];
#pragma unroll
            for (int mm = 0; mm < 4; mm++)
#pragma unroll
                for (int a = 0; a < UP; a++) {
                    const int o = (a > PH) ? 1 : 0;
                    float acc = 0.f;
#pragma unroll
                    for (int j = 0; j < FUT; j++) acc = fmaf(c[a][j], in[mm + o + j], acc);
                    out[mm * UP + a] = acc;
                }
            float* dst = upX + r * PU + 4 * UP * ch;
#pragma unroll
            for (int q = 0; q < UP; q++) *(float4*)(dst + 4 * q) = make_float4(out[4 * q], out[4 * q + 1], out[4 * q + 2], out[4 * q + 3]);
        }
    }

    // ---- stage C: up-FIR along y + gain + leaky ReLU + clamp + sign codes.  One item = 4 columns x 8 rows.
    template <int PH>
    static __device__ __forceinline__ void up_y_act(const float* __restrict__ upX, float* __restrict__ upXY,
                                                    const float* __restrict__ cu, const unsigned* __restrict__ sgn, int tid,
                                                    const FlreluParams& p, int plane, int U0x, int U0y, bool lastX, bool lastY) {
        float c[UP][FUT];
#pragma unroll
        for (int a = 0; a < UP; a++)
#pragma unroll
            for (int j = 0; j < FUT; j++) c[a][j] = cu[a * FUT + j];
        constexpr int NG = TUWP / 4;
        constexpr int NRB = TUHP / ROWS_C;
        constexpr int NIN = MB + FUT;
        unsigned char* splane = p.s + (size_t)plane * p.sh * p.swb;
        for (int item = tid; item < NG * NRB; item += NT) {
            const int rb = item / NG, g = item - rb * NG;
            float4 in[NIN];
#pragma unroll
            for (int i = 0; i < NIN; i++) in[i] = *(const float4*)(upX + (rb * MB + i) * PU + 4 * g);
            const int X = U0x + 4 * g;
            // READ mode: bit offset of this item's 4 codes inside the staged dword pair
            const int sbit = (((U0x + p.sx) & 15) + 4 * g) * 2;
            const int sw0 = sbit >> 5, sshift = sbit & 31;
#pragma unroll
            for (int mm = 0; mm < MB; mm++)
#pragma unroll
                for (int a = 0; a < UP; a++) {
                    const int o = (a > PH) ? 1 : 0;
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < FUT; j++) {
                        const float w = c[a][j];
                        const float4 v = in[mm + o + j];
                        acc.x = fmaf(w, v.x, acc.x);
                        acc.y = fmaf(w, v.y, acc.y);
                        acc.z = fmaf(w, v.z, acc.z);
                        acc.w = fmaf(w, v.w, acc.w);
                    }
                    const int row = rb * ROWS_C + mm * UP + a;
                    const int Y = U0y + row;
                    unsigned codes = 0;
                    if (SIGN == AFCM_SIGNS_READ) {
                        const unsigned lo = sgn[row * SGN_W + sw0];
                        const unsigned hi = (sw0 + 1 < SGN_W) ? sgn[row * SGN_W + sw0 + 1] : 0u;
                        codes = __builtin_amdgcn_alignbit(hi, lo, sshift) & 0xffu;
                    }
                    unsigned c0 = act_elem<SIGN>(acc.x, p.gain, p.slope, p.clamp, codes);
                    unsigned c1 = act_elem<SIGN>(acc.y, p.gain, p.slope, p.clamp, codes >> 2);
                    unsigned c2 = act_elem<SIGN>(acc.z, p.gain, p.slope, p.clamp, codes >> 4);
                    unsigned c3 = act_elem<SIGN>(acc.w, p.gain, p.slope, p.clamp, codes >> 6);
                    *(float4*)(upXY + row * PU + 4 * g) = acc;
                    if (SIGN == AFCM_SIGNS_WRITE) {
                        // 4 lanes of a quad hold 16 consecutive columns: assemble one dword.
                        int byte = (int)(c0 | (c1 << 2) | (c2 << 4) | (c3 << 6));
                        int word = quad_or(byte << ((g & 3) << 3));
                        const bool ownX = (4 * g < TOW * DOWN) || lastX;
                        const bool ownY = (row < TOH * DOWN) || lastY;
                        if ((g & 3) == 0 && ownX && ownY && (X >> 2) < p.swb && Y < p.sh)
                            *(int*)(splane + (size_t)Y * p.swb + (X >> 2)) = word;
                    }
                }
        }
    }

    // ---- stage D: down-FIR along x.  One item = one upsampled row x 4 outputs.
    static __device__ __forceinline__ void down_x(const float* __restrict__ upXY, float* __restrict__ downX,
                                                  const float* __restrict__ cdl, int tid) {
        float cd[FD];
#pragma unroll
        for (int k = 0; k < FD; k++) cd[k] = cdl[k];
        constexpr int NCD = TOW / 4;
        constexpr int NIN4 = cdiv(DOWN * 3 + FD, 4);
        for (int item = tid; item < TUH * NCD; item += NT) {
            const int ch = item / TUH, r = item - ch * TUH;     // consecutive lanes -> consecutive rows
            const float* src = upXY + r * PU + DOWN * 4 * ch;
            float in[NIN4 * 4];
#pragma unroll
            for (int i = 0; i < NIN4; i++) {
                float4 t = *(const float4*)(src + 4 * i);
                in[4 * i] = t.x; in[4 * i + 1] = t.y; in[4 * i + 2] = t.z; in[4 * i + 3] = t.w;
            }
            float out[4];
#pragma unroll
            for (int t = 0; t < 4; t++) {
                float acc = 0.f;
#pragma unroll
                for (int k = 0; k < FD; k++) acc = fmaf(cd[k], in[DOWN * t + k], acc);
                out[t] = acc;
            }
            *(float4*)(downX + r * PD + 4 * ch) = make_float4(out[0], out[1], out[2], out[3]);
        }
    }

    // ---- stage E: down-FIR along y + store.  One item = 2 output columns x RO output rows.
    static __device__ __forceinline__ void down_y_store(const float* __restrict__ downX, const float* __restrict__ cdl,
                                                        int tid, const FlreluParams& p, int plane, int O0x, int O0y) {
        float cd[FD];
#pragma unroll
        for (int k = 0; k < FD; k++) cd[k] = cdl[k];
        constexpr int NCP = TOW / 2;
        constexpr int NROW = DOWN * (RO - 1) + FD;
        T* yp = (T*)p.y + (size_t)plane * p.yh * p.yw;
        for (int item = tid; item < NCP * (TOH / RO); item += NT) {
            const int rbk = item / NCP, cp = item - rbk * NCP;
            const int p0 = rbk * RO;
            float2 acc[RO];
#pragma unroll
            for (int t = 0; t < RO; t++) acc[t] = make_float2(0.f, 0.f);
#pragma unroll
            for (int i = 0; i < NROW; i++) {
                const float2 v = *(const float2*)(downX + (DOWN * p0 + i) * PD + 2 * cp);
#pragma unroll
                for (int t = 0; t < RO; t++) {
                    const int k = i - DOWN * t;
                    if (k >= 0 && k < FD) {
                        acc[t].x = fmaf(cd[k], v.x, acc[t].x);
                        acc[t].y = fmaf(cd[k], v.y, acc[t].y);
                    }
                }
            }
            const int ox = O0x + 2 * cp;
#pragma unroll
            for (int t = 0; t < RO; t++) {
                const int oy = O0y + p0 + t;
                if (oy < p.yh) {
                    T* dst = yp + (size_t)oy * p.yw + ox;
                    if (sizeof(T) == 4 && ox + 1 < p.yw && ((p.yw & 1) == 0)) {
                        // even plane widths: the pair starts on an 8-byte boundary -- one store instead of two interleaved ones
                        *(float2*)dst = acc[t];
                    } else {
                        if (ox < p.yw) dst[0] = from_f32<T>(acc[t].x);
                        if (ox + 1 < p.yw) dst[1] = from_f32<T>(acc[t].y);
                    }
                }
            }
        }
    }
};

template <typename T, int UP, int DOWN, int FUT, int FD, int TOW, int TOH, int RO, int NT, int SIGN>
__global__ __launch_bounds__(NT) void flrelu_sep_kernel(FlreluParams p, const float* __restrict__ fu,
                                                        const float* __restrict__ fd) {
    typedef FlreluTile<T, UP, DOWN, FUT, FD, TOW, TOH, RO, NT, SIGN> K;
    __shared__ __attribute__((aligned(16))) float lds[K::LDS_FLOATS];
    float* bufA = lds;
    float* bufB = lds + K::SZ_A;
    float* cuX = lds + K::SZ_A + K::SZ_B;
    float* cuY = cuX + K::FU;
    float* cdl = cuY + K::FU;
    unsigned* sgn = (unsigned*)(lds + K::SZ_A + K::SZ_B + round_up(K::NCOEF, 4));

    const int tid = threadIdx.x;
    // XCD-aware order: consecutive logical tiles (neighbours of one plane, shared halos) stay on one XCD / one L2
    int bid = xcd_order(blockIdx.x, gridDim.x);
    const int tx = bid % p.tilesX;
    bid /= p.tilesX;
    const int ty = bid % p.tilesY;
    const int plane = bid / p.tilesY;

    const int O0x = tx * TOW, O0y = ty * TOH;
    const int U0x = O0x * DOWN, U0y = O0y * DOWN;
    const int I0x = -floor_div(p.px0 - U0x, UP), phx = pos_mod(p.px0 - U0x, UP);
    const int I0y = -floor_div(p.py0 - U0y, UP), phy = pos_mod(p.py0 - U0y, UP);

    // polyphase coefficient tables
    if (tid < K::FU) {
        const int a = tid / FUT, j = tid - a * FUT;
        {
            const int kmin = (a > phx) ? UP - (a - phx) : phx - a;
            const int k = kmin + UP * j;
            cuX[tid] = p.flip ? fu[k] : fu[K::FU - 1 - k];
        }
        {
            const int kmin = (a > phy) ? UP - (a - phy) : phy - a;
            const int k = kmin + UP * j;
            cuY[tid] = p.flip ? fu[k] : fu[K::FU - 1 - k];
        }
    }
    if (tid < FD) cdl[tid] = p.flip ? fd[tid] : fd[FD - 1 - tid];

    // stage A: input tile + bias (zero outside the image, without bias: the bias is added before padding).
    // All global loads of the tile are issued back to back before the first LDS write, so the tile
    // pays one HBM round trip, not one per element.
    {
        const T* xp = (const T*)p.x + (size_t)plane * p.xh * p.xw;
        const float bias = p.b ? to_f32(((const T*)p.b)[plane % p.C]) : 0.f;
        constexpr int NLD = cdiv(K::TIH * K::TIWP, NT);
        T raw[NLD];
        bool ok[NLD];
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid + i * NT;
            const int r = idx / K::TIWP, c = idx - r * K::TIWP;
            const int iy = I0y + r, ix = I0x + c;
            ok[i] = (idx < K::TIH * K::TIWP) && (unsigned)ix < (unsigned)p.xw && (unsigned)iy < (unsigned)p.xh;
            raw[i] = ok[i] ? xp[(size_t)iy * p.xw + ix] : from_f32<T>(0.f);
        }
        if (SIGN == AFCM_SIGNS_READ) K::stage_signs(sgn, p, plane, U0x, U0y, tid);
#pragma unroll
        for (int i = 0; i < NLD; i++) {
            const int idx = tid + i * NT;
            if (idx < K::TIH * K::TIWP) bufA[idx] = ok[i] ? to_f32(raw[i]) + bias : 0.f;
        }
    }
    __syncthreads();
    switch (phx) {
        case 0: K::template up_x<0>(bufA, bufB, cuX, tid); break;
        case 1: K::template up_x<1>(bufA, bufB, cuX, tid); break;
        case 2: if (UP > 2) K::template up_x<(UP > 2 ? 2 : 0)>(bufA, bufB, cuX, tid); break;
        default: if (UP > 2) K::template up_x<(UP > 2 ? 3 : 0)>(bufA, bufB, cuX, tid); break;
    }
    __syncthreads();
    const bool lastX = (tx == p.tilesX - 1), lastY = (ty == p.tilesY - 1);
    switch (phy) {
        case 0: K::template up_y_act<0>(bufB, bufA, cuY, sgn, tid, p, plane, U0x, U0y, lastX, lastY); break;
        case 1: K::template up_y_act<1>(bufB, bufA, cuY, sgn, tid, p, plane, U0x, U0y, lastX, lastY); break;
        case 2: if (UP > 2) K::template up_y_act<(UP > 2 ? 2 : 0)>(bufB, bufA, cuY, sgn, tid, p, plane, U0x, U0y, lastX, lastY); break;
        default: if (UP > 2) K::template up_y_act<(UP > 2 ? 3 : 0)>(bufB, bufA, cuY, sgn, tid, p, plane, U0x, U0y, lastX, lastY); break;
    }
    __syncthreads();
    K::down_x(bufA, bufB, cdl, tid);
    __syncthreads();
    K::down_y_store(bufB, cdl, tid, p, plane, O0x, O0y);
}

// ---------------------------------------------------------------------------------------------
// Pointwise form: up = down = 1 with 1x1 filters (the ToRGB layer, NET:369-372) and the in-place
// activation of the generic fallback (filtered_lrelu_act_, filtered_lrelu.cu:1105-1211).
// One thread = 16 consecutive columns of one row = one sign dword.
template <typename T, int SIGN>
__global__ __launch_bounds__(256) void flrelu_pointwise_kernel(FlreluParams p) {
    const int chunks = (p.yw + 15) >> 4;
    const long long total = (long long)p.tilesY * p.yh * chunks;  // tilesY = planes here
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int ch = (int)(idx % chunks);
        const long long t = idx / chunks;
        const int oy = (int)(t % p.yh);
        const int plane = (int)(t / p.yh);
        const T* xp = (const T*)p.x + (size_t)plane * p.xh * p.xw;
        T* yp = (T*)p.y + (size_t)plane * p.yh * p.yw;
        const float bias = p.b ? to_f32(((const T*)p.b)[plane % p.C]) : 0.f;
        const int iy = oy - p.py0;
        const bool rowIn = (unsigned)iy < (unsigned)p.xh;
        unsigned char* splane = (SIGN != AFCM_SIGNS_NONE) ? p.s + (size_t)plane * p.sh * p.swb : nullptr;
        unsigned word = 0;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int ox0 = ch * 16 + q * 4;
            unsigned codes = 0;
            if (SIGN == AFCM_SIGNS_READ) codes = fetch_codes4(splane, ox0 + p.sx, oy + p.sy, p.sh, p.swb);
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const int ox = ox0 + e;
                const int ix = ox - p.px0;
                float v = 0.f;
                if (ox < p.yw && rowIn && (unsigned)ix < (unsigned)p.xw) v = (to_f32(xp[(size_t)iy * p.xw + ix]) + bias) * p.fscale;
                unsigned code = act_elem<SIGN>(v, p.gain, p.slope, p.clamp, codes >> (2 * e));
                word |= code << (2 * (q * 4 + e));
                if (ox < p.yw) yp[(size_t)oy * p.yw + ox] = from_f32<T>(v);
            }
        }
        if (SIGN == AFCM_SIGNS_WRITE && oy < p.sh && ch * 4 < p.swb) *(unsigned*)(splane + (size_t)oy * p.swb + ch * 4) = word;
    }
}

// ---------------------------------------------------------------------------------------------
// Strip kernel (r03, fp32): one WAVE owns a strip of SW output columns x SH output rows of one plane and marches down it one input
// row per step, with every intermediate in registers or in the wave's own 1-3 KB of LDS -- no workgroup barrier, no tile halo in y
// (the tile kernel above recomputes (FD - DOWN) upsampled rows per 20-row tile, 1.55x the useful FMAs at up 2 / down 2 and 4.4x at
// down 4), and an instruction stream close to the arithmetic: profiles/r03_flrelu_fp32_pmc.txt has the tile kernel 75 % VALU-issue-
// bound at 289 vector operations per output where the four FIR passes need 47 packed FMAs.
//   lane l <-> input columns I0x + l, I0x + 64 + l (CPL column blocks).  Per step (input row I0y + it):
//     up-x   the row goes through LDS so that a lane sees its 6 right neighbours: UP upsampled columns per lane, 7 taps each (the
//            phase-dependent one-column offset o(a) of the polyphase form is folded into a 7-tap table with one zero: no selects)
//     up-y   a ring of the last 6 up-x rows in registers (static indices: the step loop is unrolled over the ring period) + the new
//            row -> UP upsampled rows x UP columns, 7 taps each; gain, leaky ReLU, clamp, 2-bit codes (written as whole dwords by
//            the first lane of each 16-column group after a DPP OR-reduction; READ: the row's sign dwords are fetched one step ahead
//            by the first lanes and spread through LDS)
//     down-x the UP activated rows go through LDS; lane j reads the FD taps of output column j (8-byte reads, even / odd taps in the
//            two halves of packed FMAs)
//     down-y scatter form: each new down-x row adds into the FD / DOWN output rows it contributes to (a ring of 6 accumulators,
//            static indices); the accumulator that received its last tap is stored and reset.
//   Signs: a strip owns the SW DOWN upsampled columns of its outputs (a multiple of 16: whole dwords), a segment the SH DOWN rows of
//   its outputs, the last strip / segment the rest.  The last 6 columns have no full tap support: they compute on zero padding, own nothing.
template <int LO, int HI, typename F>
__device__ __forceinline__ void strip_static_for(F&& f) {
    if constexpr (LO < HI) {
        f(std::integral_constant<int, LO>{});
        strip_static_for<LO + 1, HI>(f);
    }
}

template <int UP, int DOWN, int CPL_, int SIGN_>
struct StripGeom {
    static constexpr int FUT = 6, FU = FUT * UP, FD = FUT * DOWN;
    // CPL input columns per lane, in blocks: lane l holds columns l, 64 + l, ... of the strip's 64 CPL (coalesced row loads; the up
    // stages run once per block, the right halo -- 6 columns -- is paid once per strip: 87.5 % of the columns useful at CPL 2, 75 % at 1)
    // The host picks CPL per configuration: 1 for up 2 / down 2 (48-column strips quantise the generator's plane widths better than
    // 112-column ones: enc3 forward 0.91 vs 1.06 ms) and up 4 (registers), 2 for down 4 (56 output lanes instead of 24: 1.26 vs 1.68 ms)
    static constexpr int CPL = CPL_;
    static constexpr int NC = 64 * CPL;                         // input columns of the strip
    static constexpr int SWMAX = (UP * (NC - 6) - FD) / DOWN + 1;                           // outputs with full tap support
    static constexpr int SW = SIGN_ == AFCM_SIGNS_WRITE ? SWMAX / (16 / DOWN) * (16 / DOWN) : SWMAX;   // sign writers: whole dwords per strip
    static constexpr int NU = NC * UP;                          // upsampled columns per row of the strip
    static constexpr int OPL = cdiv(SW, 64);                    // output columns per lane
    static constexpr int PERIOD = (UP == 2 && DOWN == 4) ? 12 : 6;   // steps after which the up-y ring AND the down-y ring repeat
    static constexpr int GS = 16 / UP;                          // lanes per sign dword
    static constexpr int NW = NU / 16 + 1;                      // sign dwords a row's window can touch (READ)
    static_assert(SIGN_ != AFCM_SIGNS_WRITE || (SW * DOWN) % 16 == 0, "sign ownership must fall on dword boundaries");
    static_assert(DOWN * (SW - 1) + FD <= UP * (NC - 6), "the strip's outputs must stay inside the columns with full tap support");
    static_assert((UP * PERIOD) % (DOWN * 6) == 0 && PERIOD % 6 == 0, "ring periods");
    static_assert(NW <= 64, "one lane per sign dword");
};

template <typename T, int UP, int DOWN, int CPL_, int SIGN, bool FASTACT>
__global__ __launch_bounds__(256) void flrelu_strip_kernel(FlreluParams p, const float* __restrict__ fu, const float* __restrict__ fd) {
    typedef StripGeom<UP, DOWN, CPL_, SIGN> G;
    constexpr int FUT = G::FUT, FU = G::FU, FD = G::FD, SW = G::SW, NU = G::NU, NC = G::NC, CPL = G::CPL, OPL = G::OPL, PERIOD = G::PERIOD, GS = G::GS, NW = G::NW;
    __shared__ float s_in[4][NC + 8];                            // input row of the wave + zero pad for the neighbours of the last 6 columns
    __shared__ __attribute__((aligned(16))) float s_u[4][UP][NU];   // the UP activated rows of a step
    __shared__ unsigned s_sg[4][UP][NW + 1];                     // READ: sign dwords of the step's rows

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int SH = cdiv(p.yh, p.tilesY);
    int wt = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (wt >= p.tilesX * p.tilesY * p.planes) return;
    const int tx = wt % p.tilesX; wt /= p.tilesX;
    const int ty = wt % p.tilesY;
    const int plane = wt / p.tilesY;
    const bool lastX = tx == p.tilesX - 1, lastY = ty == p.tilesY - 1;
    const int O0x = tx * SW, O0y = ty * SH;
    const int U0x = O0x * DOWN, U0y = O0y * DOWN;
    const int I0x = -floor_div(p.px0 - U0x, UP), phx = pos_mod(p.px0 - U0x, UP);
    const int I0y = -floor_div(p.py0 - U0y, UP), phy = pos_mod(p.py0 - U0y, UP);

    // 7-tap polyphase tables (uniform: scalar registers): c7[a][t] multiplies row / column (first + t), t = 0..6
    float cx7[UP][7], cy7[UP][7], cd[FD];
#pragma unroll
    for (int a = 0; a < UP; a++) {
        const int ox = (a > phx) ? 1 : 0, oy = (a > phy) ? 1 : 0;
        const int kx = ox ? UP - (a - phx) : phx - a, ky = oy ? UP - (a - phy) : phy - a;
#pragma unroll
        for (int t = 0; t < 7; t++) {
            const int jx = t - ox, jy = t - oy;
            const int ix = kx + UP * (jx < 0 ? 0 : jx > 5 ? 5 : jx), iy = ky + UP * (jy < 0 ? 0 : jy > 5 ? 5 : jy);
            const float vx = p.flip ? fu[ix] : fu[FU - 1 - ix], vy = p.flip ? fu[iy] : fu[FU - 1 - iy];
            cx7[a][t] = (jx >= 0 && jx < FUT) ? vx : 0.f;
            cy7[a][t] = (jy >= 0 && jy < FUT) ? vy : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < FD; k++) cd[k] = p.flip ? fd[k] : fd[FD - 1 - k];
    // the up-x taps as (a, a + 1) pairs in VECTOR registers: with all three tables in the scalar file it overflows (the compiler parked
    // taps in VGPR lanes and read them back every step: 11 of 117 vector instructions per step); pairs keep the packed FMAs
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 cxp[UP / 2][7];
#pragma unroll
    for (int a2 = 0; a2 < UP / 2; a2++)
#pragma unroll
        for (int t = 0; t < 7; t++) {
            cxp[a2][t] = (f32x2){cx7[2 * a2][t], cx7[2 * a2 + 1][t]};
            asm volatile("" : "+v"(cxp[a2][t]));
        }

    float* const in_row = s_in[wave];
    if (lane < 8) in_row[NC + lane] = 0.f;
    if (SIGN == AFCM_SIGNS_READ && lane < UP) s_sg[wave][lane][NW] = 0u;
    const T* const xp = (const T*)p.x + (size_t)plane * p.xh * p.xw;
    T* const yp = (T*)p.y + (size_t)plane * p.yh * p.yw;
    unsigned char* const splane = p.s + (size_t)plane * p.sh * p.swb;
    const float bias = p.b ? to_f32(((const T*)p.b)[plane % p.C]) : 0.f;     // added inside the image only (the padding is zero)

    // rows this wave has to walk: the last tap of its last output row, in WRITE mode of the last segment also the last sign row
    const int SHv = min(SH, p.yh - O0y);
    int qmax = DOWN * (SHv - 1) + FD - 1;
    if (SIGN == AFCM_SIGNS_WRITE && lastY) qmax = max(qmax, p.sh - 1 - U0y);
    const int NIT = qmax / UP + 7;

    // READ: dword window of a sign row and this lane's bit offset inside it (column block c: + 128 UP bits)
    const int w0 = floor_div(U0x + p.sx, 16);
    const int sbit = pos_mod(U0x + p.sx, 16) * 2 + 2 * UP * lane;
    const int wpr = p.swb >> 2;
    auto fetch_signs = [&](int it, unsigned (&sg)[UP]) __attribute__((always_inline)) {
#pragma unroll
        for (int a = 0; a < UP; a++) {
            const int Y = U0y + UP * (it - 6) + a + p.sy, wi = w0 + lane;
            const bool ok = lane < NW && (unsigned)Y < (unsigned)p.sh && (unsigned)wi < (unsigned)wpr;
            sg[a] = ok ? ((const unsigned*)(splane + (size_t)(ok ? Y : 0) * p.swb))[ok ? wi : 0] : 0u;
        }
    };
    // the row is requested one step before its use: clamped address (no branch around the load, nothing waits on it here); validity and
    // the bias are applied when the value is consumed
    bool colok[CPL];
    int colx[CPL];
#pragma unroll
    for (int c = 0; c < CPL; c++) {
        const int ix = I0x + 64 * c + lane;
        colok[c] = (unsigned)ix < (unsigned)p.xw;
        colx[c] = min(max(ix, 0), p.xw - 1);
    }
    auto fetch_input = [&](int it, T (&xv)[CPL]) __attribute__((always_inline)) {
        const int iy = min(max(I0y + it, 0), p.xh - 1);
        const T* row = xp + (size_t)iy * p.xw;
#pragma unroll
        for (int c = 0; c < CPL; c++) xv[c] = row[colx[c]];
    };

    f32x2 ring[6][CPL][UP / 2];                                  // up-x rows: (a, a + 1) column pairs
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
        for (int c = 0; c < CPL; c++)
#pragma unroll
            for (int a2 = 0; a2 < UP / 2; a2++) ring[j][c][a2] = (f32x2){0.f, 0.f};
    float acc[6][OPL];
#pragma unroll
    for (int j = 0; j < 6; j++)
#pragma unroll
        for (int o = 0; o < OPL; o++) acc[j][o] = 0.f;

    T xnext[CPL];
    fetch_input(0, xnext);
    unsigned sgnext[UP];
#pragma unroll
    for (int a = 0; a < UP; a++) sgnext[a] = 0u;
    if (SIGN == AFCM_SIGNS_READ) fetch_signs(6, sgnext);

    for (int base = 0; base < NIT; base += PERIOD) {
        strip_static_for<0, PERIOD>([&](auto phc) __attribute__((always_inline)) {
            constexpr int ph = decltype(phc)::value;
            const int it = base + ph;
            if (it < NIT) {
                // ---- up-x
                float xin[CPL];
                const bool rowok = (unsigned)(I0y + it) < (unsigned)p.xh;
#pragma unroll
                for (int c = 0; c < CPL; c++) xin[c] = (rowok && colok[c]) ? to_f32(xnext[c]) + bias : 0.f;
                fetch_input(it + 1, xnext);
#pragma unroll
                for (int c = 0; c < CPL; c++) in_row[64 * c + lane] = xin[c];
                __builtin_amdgcn_wave_barrier();
                f32x2 R[CPL][UP / 2];
#pragma unroll
                for (int c = 0; c < CPL; c++) {
                    float nb[7];
                    nb[0] = xin[c];
#pragma unroll
                    for (int t = 1; t < 7; t++) nb[t] = in_row[64 * c + lane + t];
#pragma unroll
                    for (int a2 = 0; a2 < UP / 2; a2++) {
                        f32x2 s0 = (f32x2){0.f, 0.f};
#pragma unroll
                        for (int t = 0; t < 7; t++) s0 = __builtin_elementwise_fma(cxp[a2][t], (f32x2){nb[t], nb[t]}, s0);
                        R[c][a2] = s0;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                if (it >= 6) {
                    // ---- up-y: rows m + t, t = 0..5 in ring[(ph + t) % 6], row m + 6 = R;  m = it - 6
                    if (SIGN == AFCM_SIGNS_READ) {
                        unsigned sg[UP];
#pragma unroll
                        for (int a = 0; a < UP; a++) sg[a] = sgnext[a];
                        fetch_signs(it + 1, sgnext);
#pragma unroll
                        for (int a = 0; a < UP; a++)
                            if (lane < NW) s_sg[wave][a][lane] = sg[a];
                        __builtin_amdgcn_wave_barrier();
                    }
                    const int q0 = UP * (it - 6);                 // first upsampled row of the step, relative to U0y
#pragma unroll
                    for (int ay = 0; ay < UP; ay++) {
#pragma unroll
                        for (int c = 0; c < CPL; c++) {
                            float v[UP];
#pragma unroll
                            for (int a2 = 0; a2 < UP / 2; a2++) {
                                f32x2 s0 = (f32x2){0.f, 0.f};
#pragma unroll
                                for (int t = 0; t < 6; t++) s0 = __builtin_elementwise_fma((f32x2){cy7[ay][t], cy7[ay][t]}, ring[(ph + t) % 6][c][a2], s0);
                                s0 = __builtin_elementwise_fma((f32x2){cy7[ay][6], cy7[ay][6]}, R[c][a2], s0);
                                v[2 * a2] = s0.x;
                                v[2 * a2 + 1] = s0.y;
                            }
                            unsigned codes = 0u;
                            if (SIGN == AFCM_SIGNS_READ) {
                                const int sb = sbit + 128 * UP * c;
                                const unsigned lo = s_sg[wave][ay][sb >> 5], hi = s_sg[wave][ay][(sb >> 5) + 1];
                                codes = __builtin_amdgcn_alignbit(hi, lo, sb & 31);
                            }
                            unsigned nib = 0u;
                            if (SIGN != AFCM_SIGNS_READ && FASTACT) {
                                // 0 <= slope <= 1: leaky ReLU = max(v, slope v); the clamp is a select on the compare the code needs anyway
                                // (NOT a med3: v_med3_f32 turns a NaN into -clamp, act_elem and the reference kernel hand it on) -- the
                                // same values as act_elem bit for bit, NaN included, one instruction fewer per element
#pragma unroll
                                for (int a2 = 0; a2 < UP / 2; a2++) {
                                    const f32x2 g2 = (f32x2){v[2 * a2], v[2 * a2 + 1]} * (f32x2){p.gain, p.gain};
                                    const f32x2 t2 = g2 * (f32x2){p.slope, p.slope};
#pragma unroll
                                    for (int e = 0; e < 2; e++) {
                                        const int ax = 2 * a2 + e;
                                        const float w = fmaxf(g2[e], t2[e]);
                                        unsigned code = __float_as_uint(g2[e]) >> 31;
                                        const bool big = fabsf(w) > p.clamp;            // (false for a NaN)
                                        if (big) code = 2u;
                                        v[ax] = big ? __builtin_copysignf(p.clamp, w) : w;
                                        nib |= code << (2 * ax);
                                    }
                                }
                            } else {
#pragma unroll
                                for (int ax = 0; ax < UP; ax++) nib |= act_elem<SIGN>(v[ax], p.gain, p.slope, p.clamp, codes >> (2 * ax)) << (2 * ax);
                            }
                            if (SIGN == AFCM_SIGNS_WRITE) {
                                int word = (int)(nib << ((lane % GS) * 2 * UP));
                                word |= __builtin_amdgcn_mov_dpp(word, 0xB1, 0xF, 0xF, true);              // quad_perm [1,0,3,2]
                                word |= __builtin_amdgcn_mov_dpp(word, 0x4E, 0xF, 0xF, true);              // quad_perm [2,3,0,1]
                                if (GS == 8) word |= __builtin_amdgcn_mov_dpp(word, 0x141, 0xF, 0xF, true);   // row_half_mirror
                                const int col = 64 * c + lane;
                                const int q = q0 + ay, Y = U0y + q, X0 = U0x + UP * col;
                                const bool own = ((UP * col < SW * DOWN) || lastX) && ((q < SH * DOWN) || lastY);
                                if ((lane % GS) == 0 && col + GS <= NC - 6 && own && (X0 >> 2) < p.swb && Y < p.sh)
                                    *(int*)(splane + (size_t)Y * p.swb + (X0 >> 2)) = word;
                            }
                            if constexpr (UP == 2) *(float2*)(&s_u[wave][ay][UP * (64 * c + lane)]) = make_float2(v[0], v[1]);
                            else *(float4*)(&s_u[wave][ay][UP * (64 * c + lane)]) = make_float4(v[0], v[1], v[2], v[3]);
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                    // ---- down-x and down-y
#pragma unroll
                    for (int ay = 0; ay < UP; ay++) {
                        constexpr int QS_BASE = ((UP * (ph - 6)) % (DOWN * 6) + DOWN * 6) % (DOWN * 6);
                        const int qs = (QS_BASE + ay) % (DOWN * 6);       // the row's index modulo the down-y period (compile time after unrolling)
                        float d[OPL];
#pragma unroll
                        for (int o = 0; o < OPL; o++) {
                            const int j = min(lane + 64 * o, SW - 1);
                            const f32x2* src = (const f32x2*)(&s_u[wave][ay][DOWN * j]);
                            f32x2 e2 = (f32x2){0.f, 0.f};                   // even / odd taps in the two halves
#pragma unroll
                            for (int k2 = 0; k2 < FD / 2; k2++) e2 = __builtin_elementwise_fma((f32x2){cd[2 * k2], cd[2 * k2 + 1]}, src[k2], e2);
                            d[o] = e2.x + e2.y;
                        }
#pragma unroll
                        for (int i = 0; i < 6; i++) {
                            const int slot = ((qs / DOWN - i) % 6 + 6) % 6, k = qs % DOWN + DOWN * i;
#pragma unroll
                            for (int o = 0; o < OPL; o++) acc[slot][o] = fmaf(cd[k], d[o], acc[slot][o]);
                        }
                        if (qs % DOWN == DOWN - 1) {
                            const int slot = ((qs / DOWN - 5) % 6 + 6) % 6;
                            const int pr = (q0 + ay - (FD - 1)) / DOWN;     // exact: q - (FD - 1) is a multiple of DOWN here
                            if (q0 + ay >= FD - 1 && pr < SHv) {
#pragma unroll
                                for (int o = 0; o < OPL; o++) {
                                    const int j = lane + 64 * o;
                                    if (j < SW && O0x + j < p.yw) yp[(size_t)(O0y + pr) * p.yw + O0x + j] = from_f32<T>(acc[slot][o]);
                                }
                            }
#pragma unroll
                            for (int o = 0; o < OPL; o++) acc[slot][o] = 0.f;
                        }
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                // the new up-x row replaces the oldest one
#pragma unroll
                for (int c = 0; c < CPL; c++)
#pragma unroll
                    for (int a2 = 0; a2 < UP / 2; a2++) ring[ph % 6][c][a2] = R[c][a2];
            }
        });
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int UP, int DOWN, int FUT, int FD, int TOW, int TOH, int RO, int NT>
static int launch_sep(const afcm_filtered_lrelu_args* a, FlreluParams p, hipStream_t st) {
    p.tilesX = cdiv(a->yw, TOW);
    p.tilesY = cdiv(a->yh, TOH);
    const long long blocks = (long long)p.tilesX * p.tilesY * a->n * a->c;
    AFCM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "filtered_lrelu: grid of %lld blocks is out of range", blocks);
    dim3 grid((unsigned)blocks), block(NT);
    switch (a->sign_mode) {
        case AFCM_SIGNS_NONE:
            hipLaunchKernelGGL((flrelu_sep_kernel<T, UP, DOWN, FUT, FD, TOW, TOH, RO, NT, AFCM_SIGNS_NONE>), grid, block, 0, st, p, a->fu, a->fd);
            break;
        case AFCM_SIGNS_WRITE:
            hipLaunchKernelGGL((flrelu_sep_kernel<T, UP, DOWN, FUT, FD, TOW, TOH, RO, NT, AFCM_SIGNS_WRITE>), grid, block, 0, st, p, a->fu, a->fd);
            break;
        default:
            hipLaunchKernelGGL((flrelu_sep_kernel<T, UP, DOWN, FUT, FD, TOW, TOH, RO, NT, AFCM_SIGNS_READ>), grid, block, 0, st, p, a->fu, a->fd);
            break;
    }
    return hip_status(hipGetLastError());
}

template <typename T, int UP, int DOWN, int CPL>
static int launch_strip(const afcm_filtered_lrelu_args* a, FlreluParams p, hipStream_t st) {
    p.tilesX = a->sign_mode == AFCM_SIGNS_WRITE ? cdiv(a->yw, StripGeom<UP, DOWN, CPL, AFCM_SIGNS_WRITE>::SW) : cdiv(a->yw, StripGeom<UP, DOWN, CPL, AFCM_SIGNS_NONE>::SW);
    constexpr int rows = 96;                                            // output rows per segment (profiles/r03_flrelu_fp32_strip_rows_sweep.txt)
    p.tilesY = a->yh <= rows ? 1 : (a->yh + rows / 2) / rows;
    p.planes = a->n * a->c;
    const long long waves = (long long)p.tilesX * p.tilesY * p.planes;
    AFCM_REQUIRE(waves > 0 && waves < (1ll << 31), "filtered_lrelu: grid of %lld waves is out of range", waves);
    dim3 grid((unsigned)((waves + 3) / 4)), block(256);
    const bool fast = a->slope >= 0.f && a->slope <= 1.f && a->clamp >= 0.f;     // (NaN fails every comparison: general form)
    switch (a->sign_mode) {
        case AFCM_SIGNS_NONE:
            if (fast) hipLaunchKernelGGL((flrelu_strip_kernel<T, UP, DOWN, CPL, AFCM_SIGNS_NONE, true>), grid, block, 0, st, p, a->fu, a->fd);
            else hipLaunchKernelGGL((flrelu_strip_kernel<T, UP, DOWN, CPL, AFCM_SIGNS_NONE, false>), grid, block, 0, st, p, a->fu, a->fd);
            break;
        case AFCM_SIGNS_WRITE:
            if (fast) hipLaunchKernelGGL((flrelu_strip_kernel<T, UP, DOWN, CPL, AFCM_SIGNS_WRITE, true>), grid, block, 0, st, p, a->fu, a->fd);
            else hipLaunchKernelGGL((flrelu_strip_kernel<T, UP, DOWN, CPL, AFCM_SIGNS_WRITE, false>), grid, block, 0, st, p, a->fu, a->fd);
            break;
        default: hipLaunchKernelGGL((flrelu_strip_kernel<T, UP, DOWN, CPL, AFCM_SIGNS_READ, false>), grid, block, 0, st, p, a->fu, a->fd); break;
    }
    return hip_status(hipGetLastError());
}

template <typename T>
static int launch_pointwise(FlreluParams p, int planes, int sign_mode, hipStream_t st) {
    p.tilesY = planes;
    const long long items = (long long)planes * p.yh * ((p.yw + 15) >> 4);
    long long blocks = (items + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    dim3 grid((unsigned)blocks), block(256);
    switch (sign_mode) {
        case AFCM_SIGNS_NONE: hipLaunchKernelGGL((flrelu_pointwise_kernel<T, AFCM_SIGNS_NONE>), grid, block, 0, st, p); break;
        case AFCM_SIGNS_WRITE: hipLaunchKernelGGL((flrelu_pointwise_kernel<T, AFCM_SIGNS_WRITE>), grid, block, 0, st, p); break;
        default: hipLaunchKernelGGL((flrelu_pointwise_kernel<T, AFCM_SIGNS_READ>), grid, block, 0, st, p); break;
    }
    return hip_status(hipGetLastError());
}

template <typename T>
static int dispatch(const afcm_filtered_lrelu_args* a, const FlreluParams& p, hipStream_t st) {
    const bool sep = (a->fuh == 0 && a->fdh == 0);
    if constexpr (sizeof(T) == 4) {
        // fp32: the strip kernel (planes below 2^31 elements; the tile kernel stays for the 16-bit calls with a bias operand)
        const bool strip = sep && (long long)a->xw * a->xh < (1ll << 30) && (long long)a->yw * a->yh < (1ll << 30);
        if (strip && a->up == 2 && a->down == 2 && a->fuw == 12 && a->fdw == 12) return launch_strip<T, 2, 2, 1>(a, p, st);
        if (strip && a->up == 2 && a->down == 4 && a->fuw == 12 && a->fdw == 24) return launch_strip<T, 2, 4, 2>(a, p, st);
        if (strip && a->up == 4 && a->down == 2 && a->fuw == 24 && a->fdw == 12) return launch_strip<T, 4, 2, 1>(a, p, st);
    }
    if (sep && a->up == 2 && a->down == 2 && a->fuw == 12 && a->fdw == 12)
    {
        // Tile height by mode (measured, fp32, batch 16): the sign-writing forward runs 10 % faster on 20-row tiles (53 KB of LDS:
        // three workgroups per CU instead of two cover its five LDS stages), the sign-reading backward 12 % slower (its staged sign
        // window grows with the halo); planes of <= 40 rows take the 20-row tile both ways (36 rows: 40 computed instead of 70).
        const int toh = (a->sign_mode != AFCM_SIGNS_READ || a->yh <= 40) ? 20 : 35;
        if (toh == 20) return launch_sep<T, 2, 2, 6, 12, 64, 20, 5, 384>(a, p, st);
        return launch_sep<T, 2, 2, 6, 12, 64, 35, 5, 384>(a, p, st);
    }
    if (sep && a->up == 2 && a->down == 4 && a->fuw == 12 && a->fdw == 24) {
        // 16-column tiles (46 KB of LDS) only where they also cut the padded columns: planes of <= 40 columns (36 / 38: 48 computed
        // instead of 64).  On the larger planes the 1.31x halo of a 16-column tile costs more than the third workgroup per CU wins.
        const int tow = a->yw <= 40 ? 16 : 32;
        if (tow == 16) return launch_sep<T, 2, 4, 6, 24, 16, 12, 4, 384>(a, p, st);
        return launch_sep<T, 2, 4, 6, 24, 32, 12, 4, 384>(a, p, st);
    }
    if (sep && a->up == 4 && a->down == 2 && a->fuw == 24 && a->fdw == 12) {
        return launch_sep<T, 4, 2, 6, 12, 64, 20, 5, 384>(a, p, st);      // (20-row tiles: 47 KB of LDS, faster than 35 rows in every mode)
    }
    return AFCM_E_NOKERNEL;
}

int flrelu_mfma_supported(const afcm_filtered_lrelu_args* a);
int flrelu_mfma_tiles(const afcm_filtered_lrelu_args* a);
int flrelu_mfma_sign_layout(const afcm_filtered_lrelu_args* a);
int flrelu_mfma_row_pitch_ok(const afcm_filtered_lrelu_args* a);
int flrelu_mfma_launch(const afcm_filtered_lrelu_args* a, bool prepare, hipStream_t st);

}  // namespace afcm

using namespace afcm;

extern "C" int afcm_filtered_lrelu_shapes(afcm_filtered_lrelu_args* a) {
    AFCM_REQUIRE(a != nullptr, "filtered_lrelu: null args");
    AFCM_REQUIRE(a->up >= 1 && a->down >= 1, "up and down must be at least 1");
    AFCM_REQUIRE(a->fuw >= 1 && a->fdw >= 1 && a->fuh >= 0 && a->fdh >= 0, "fu and fd must not be empty");
    const long long fut_w = a->fuw - 1, fut_h = (a->fuh ? a->fuh : a->fuw) - 1;
    const long long fdt_w = a->fdw - 1, fdt_h = (a->fdh ? a->fdh : a->fdw) - 1;
    const long long cw = (long long)a->xw * a->up + (a->px0 + a->px1) - fut_w;
    const long long ch = (long long)a->xh * a->up + (a->py0 + a->py1) - fut_h;
    AFCM_REQUIRE(cw > fdt_w && ch > fdt_h, "upsampled buffer must be at least the size of downsampling filter");
    const long long yw = (cw - fdt_w + (a->down - 1)) / a->down;
    const long long yh = (ch - fdt_h + (a->down - 1)) / a->down;
    AFCM_REQUIRE(yw > 0 && yh > 0 && yw < (1ll << 31) && yh < (1ll << 31), "output must be at least 1x1");
    a->yw = (int)yw;
    a->yh = (int)yh;
    a->plane_sum_slots = (a->workspace != nullptr && flrelu_mfma_supported(a)) ? flrelu_mfma_tiles(a) : 0;
    a->row_pitch_ok = (a->workspace != nullptr && flrelu_mfma_row_pitch_ok(a)) ? 1 : 0;
    if (a->sign_mode == AFCM_SIGNS_WRITE) {
        const long long sw_active = yw * a->down - (a->down - 1) + fdt_w;
        const long long sh = yh * a->down - (a->down - 1) + fdt_h;
        if (a->workspace != nullptr && flrelu_mfma_supported(a)) {
            a->sign_layout = flrelu_mfma_sign_layout(a);         // row-quad bytes (one byte = 4 rows of one column), 2: column-blocked
            a->sh = (int)((sh + 3) >> 2);
            if (a->sign_layout == 2) a->sh = (a->sh + 15) & ~15;   // whole dwords: 4 row blocks of 4 quad-rows (filtered_lrelu_wave.hip)
            a->swb = (int)((sw_active + 15) & ~15ll);
        } else {
            a->sign_layout = 0;
            a->sh = (int)sh;
            a->swb = (int)(((sw_active + 15) & ~15ll) >> 2);
        }
    }
    return AFCM_OK;
}

extern "C" int afcm_filtered_lrelu(const afcm_filtered_lrelu_args* a, void* stream) {
    AFCM_REQUIRE(a != nullptr && a->x && a->y, "filtered_lrelu: x and y must be non-null");
    AFCM_REQUIRE(a->dtype == AFCM_F32 || a->dtype == AFCM_F16 || a->dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    AFCM_REQUIRE(a->n > 0 && a->c > 0 && a->xh > 0 && a->xw > 0, "x is empty");
    AFCM_REQUIRE((long long)a->n * a->c < (1ll << 31), "x is too large");
    afcm_filtered_lrelu_args chk = *a;
    int rc = afcm_filtered_lrelu_shapes(&chk);
    if (rc != AFCM_OK) return rc;
    AFCM_REQUIRE(chk.yh == a->yh && chk.yw == a->yw, "y has shape [%d, %d], expected [%d, %d]", a->yh, a->yw, chk.yh, chk.yw);
    if (a->sign_mode != AFCM_SIGNS_NONE) {
        AFCM_REQUIRE(a->signs != nullptr && a->sh > 0 && a->swb > 0 && (a->swb & 3) == 0, "signs must be a [N,C,sh,4k] uint8 tensor");
        if (a->sign_mode == AFCM_SIGNS_WRITE) {
            AFCM_REQUIRE(a->sx == 0 && a->sy == 0, "sign offsets must be zero when writing signs");
            AFCM_REQUIRE(chk.sh == a->sh && chk.swb == a->swb && chk.sign_layout == a->sign_layout,
                         "signs has shape [%d, %d] layout %d, expected [%d, %d] layout %d", a->sh, a->swb, a->sign_layout, chk.sh, chk.swb, chk.sign_layout);
        }
    }
    hipStream_t st = (hipStream_t)stream;
    const bool mfma = a->workspace != nullptr && flrelu_mfma_supported(a);
    if ((a->x_pitch && a->x_pitch != a->xw) || (a->y_pitch && a->y_pitch != a->yw) || (a->skip_pitch && a->skip_pitch != a->yw)) {
        AFCM_REQUIRE(chk.row_pitch_ok, "filtered_lrelu: the kernel selected for this call takes dense tensors only (row pitches %d / %d / %d)", a->x_pitch, a->y_pitch, a->skip_pitch);
        AFCM_REQUIRE(a->x_pitch == 0 || a->x_pitch >= a->xw, "x_pitch %d is below the width %d", a->x_pitch, a->xw);
        AFCM_REQUIRE(a->y_pitch == 0 || (a->y_pitch >= a->yw && a->y_pitch % 8 == 0), "y_pitch %d must cover the width %d in whole 16-byte pieces", a->y_pitch, a->yw);
        AFCM_REQUIRE(a->skip_pitch == 0 || a->skip_pitch >= a->yw, "skip_pitch %d is below the width %d", a->skip_pitch, a->yw);
        AFCM_REQUIRE(((a->x_pitch | a->y_pitch | a->skip_pitch) & 1) == 0, "row pitches must be even");
    }
    AFCM_REQUIRE(mfma || (a->oscale == nullptr && a->oscale2 == nullptr && a->skip == nullptr), "filtered_lrelu: oscale / skip need the matrix-core kernels (16-bit dtype, prepared workspace)");
    if (a->sign_mode == AFCM_SIGNS_READ)
        AFCM_REQUIRE((a->sign_layout != 0) == mfma, "sign tensor layout %d does not match the kernel family selected for this call", a->sign_layout);
    if (mfma) return flrelu_mfma_launch(a, false, st);

    FlreluParams p;
    p.x = a->x; p.y = a->y; p.b = a->b; p.s = a->signs;
    p.xw = a->xw; p.xh = a->xh; p.yw = a->yw; p.yh = a->yh; p.C = a->c;
    p.px0 = a->px0; p.py0 = a->py0;
    p.tilesX = p.tilesY = 0;
    p.gain = (float)a->up * (float)a->up * a->gain;
    p.slope = a->slope; p.clamp = a->clamp; p.flip = a->flip_filter;
    p.sx = a->sx; p.sy = a->sy; p.sh = a->sh; p.swb = a->swb;
    p.fscale = 1.f;
    p.planes = a->n * a->c;

    // 1x1 filters, no resampling: pointwise kernel.  The two taps are folded into the launch: they are
    // read back on the host only when the caller did not pass NULL (= identity).
    if (a->up == 1 && a->down == 1 && a->fuw == 1 && a->fdw == 1 && a->fuh <= 1 && a->fdh <= 1) {
        if (a->fu != nullptr || a->fd != nullptr) return AFCM_E_NOKERNEL;  // non-identity 1x1 taps: generic path
        switch (a->dtype) {
            case AFCM_F32: return launch_pointwise<float>(p, a->n * a->c, a->sign_mode, st);
            case AFCM_F16: return launch_pointwise<f16_t>(p, a->n * a->c, a->sign_mode, st);
            default: return launch_pointwise<bf16_t>(p, a->n * a->c, a->sign_mode, st);
        }
    }
    AFCM_REQUIRE(a->fu != nullptr && a->fd != nullptr, "fu and fd must be non-null for resampling filters");
    switch (a->dtype) {
        case AFCM_F32: return dispatch<float>(a, p, st);
        case AFCM_F16: return dispatch<f16_t>(a, p, st);
        default: return dispatch<bf16_t>(a, p, st);
    }
}

extern "C" int afcm_filtered_lrelu_prepare(const afcm_filtered_lrelu_args* a, void* stream) {
    AFCM_REQUIRE(a != nullptr && a->workspace != nullptr && a->fu != nullptr && a->fd != nullptr, "filtered_lrelu_prepare: workspace, fu and fd must be non-null");
    if (!flrelu_mfma_supported(a)) return AFCM_E_NOKERNEL;
    return flrelu_mfma_launch(a, true, (hipStream_t)stream);
}

extern "C" int afcm_filtered_lrelu_act(void* x, uint8_t* signs, int32_t dtype, int32_t n, int32_t c, int32_t h, int32_t w,
                                       int32_t sh, int32_t swb, int32_t sx, int32_t sy, float gain, float slope, float clamp,
                                       int32_t sign_mode, void* stream) {
    AFCM_REQUIRE(x != nullptr && n > 0 && c > 0 && h > 0 && w > 0, "x is empty");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    if (sign_mode != AFCM_SIGNS_NONE) {
        AFCM_REQUIRE(signs != nullptr && sh > 0 && swb > 0 && (swb & 3) == 0, "signs must be a [N,C,sh,4k] uint8 tensor");
        if (sign_mode == AFCM_SIGNS_WRITE)
            AFCM_REQUIRE(sx == 0 && sy == 0 && sh == h && swb == (((w + 15) & ~15) >> 2), "signs must be [N,C,%d,%d] with zero offsets when writing", h, ((w + 15) & ~15) >> 2);
    }
    FlreluParams p;
    p.x = x; p.y = x; p.b = nullptr; p.s = signs;
    p.xw = w; p.xh = h; p.yw = w; p.yh = h; p.C = c;
    p.px0 = p.py0 = 0; p.tilesX = p.tilesY = 0;
    p.gain = gain; p.slope = slope; p.clamp = clamp; p.flip = 0;
    p.sx = sx; p.sy = sy; p.sh = sh; p.swb = swb; p.fscale = 1.f; p.planes = n * c;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case AFCM_F32: return launch_pointwise<float>(p, n * c, sign_mode, st);
        case AFCM_F16: return launch_pointwise<f16_t>(p, n * c, sign_mode, st);
        default: return launch_pointwise<bf16_t>(p, n * c, sign_mode, st);
    }
}
