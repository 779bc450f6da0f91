// Shared definitions of the matrix-core filtered_lrelu kernels (filtered_lrelu_mfma.hip: workgroup-tile kernels with LDS staging;
// filtered_lrelu_wave.hip: wave-autonomous kernels, no LDS staging).
#pragma once
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace afcm {

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 mbf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 mf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) short s16x2;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2;

struct FlreluMfmaParams {
    const void* x;
    void* y;
    const void* b;
    unsigned char* s;
    const void* ws;        // constant fragments + mask table
    float* plane_sum;      // optional fp32 [N*C][tilesX*tilesY]: per-tile sums of this launch's outputs (bias gradient without a second pass)
    const float* oscale;   // optional fp32 [N*C]: per-plane factor of the output
    const float* oscale2;  // optional second factor (multiplied)
    const void* skip;      // optional [N*C][yh][yw]: added to the output before the factor
    int xw, xh, yw, yh, C;
    int xld, yld, kld;     // row pitch (elements) of x, y, skip: wave kernels only (the LDS-tile kernels take dense tensors)
    int px0, py0;
    int tilesX, tilesY;
    unsigned magicT, magicP;   // ceil(2^32 / tilesX), ceil(2^32 / (tilesX * tilesY)): block id -> (plane, tile) on the scalar unit
    float slope, clamp;
    int sx, sy, shq, swq;  // sign tensor: rows of quads, bytes per row
    int total_tiles;       // wave kernels: tilesX * tilesY * planes (one wave per tile)
    int oy0, read_aligned; // wave kernels, READ: strips start at output row ty * TOH + oy0 (oy0 <= 0) so that they fall on 16-row blocks of the sign tensor
    int* clamp_flags;      // wave kernels, WRITE / NONE: optional [planes][tilesX * tilesY]: 1 if the strip took the exact (clamp-capable) path
    int st_plain;          // wave kernels: output rows leave as write-back stores instead of non-temporal ones (dense rows wider than 64 columns: launch_wave)
};

constexpr int kFUT = 6;            // taps per polyphase branch of the up filter (filter_size of the model)
constexpr int kWsTable = 16384;    // byte offset of the 256-entry sign-code -> keep-mask table inside the workspace
constexpr int kWsWave = kWsTable + 256 * 8;   // wave kernels: LIN[<= 4] and DH2[<= 3] fragments (1 KB each)
constexpr int kWsWaveFrags = 8;
constexpr int kWsScalars = kWsWave + kWsWaveFrags * 1024;   // floats: [0] = L1 norm bound of the up-y operator (incl. gain)
constexpr int kWsBytes = kWsScalars + 256;

// row index inside a 32-row K window carried by fragment element (g, j): two stacked accumulator tiles
__host__ __device__ __forceinline__ int krow(int g, int j) { return 16 * (j >> 2) + 4 * g + (j & 3); }

template <typename T> struct MfmaOps;
template <> struct MfmaOps<bf16_t> {
    typedef mbf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
};
template <> struct MfmaOps<f16_t> {
    typedef mf16x8 frag;
    static __device__ __forceinline__ f32x4 mma(frag a, frag b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }
};

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;

template <typename T>
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    // an explicit two-element conversion: one v_cvt_pk_bf16_f32 of exactly this pair.  (Scalar casts left to the
    // vectoriser get paired across dwords and re-shuffled with four extra instructions per pair; inline asm is not an
    // option on accumulator values -- the hazard recogniser does not see MFMA -> asm dependencies.)
    if constexpr (std::is_same<T, bf16_t>::value) {
        union { bf16x2 v; unsigned u; } r;
        r.v = __builtin_convertvector((f32x2){lo, hi}, bf16x2);
        return r.u;
    } else {
        union { f16x2 v; unsigned u; } r;
        r.v = __builtin_convertvector((f32x2){lo, hi}, f16x2);
        return r.u;
    }
}

template <typename F>
__device__ __forceinline__ F as_frag(const u32x4& v) {
    union { u32x4 u; F f; } r;
    r.u = v;
    return r.f;
}

// Two accumulator tiles -> one 8-element fragment: elements 0-3 from `lo`, 4-7 from `hi` (K order = krow()).
template <typename T>
__device__ __forceinline__ typename MfmaOps<T>::frag pack_pair(const f32x4& lo, const f32x4& hi) {
    u32x4 r;
    r[0] = pack2<T>(lo[0], lo[1]);
    r[1] = pack2<T>(lo[2], lo[3]);
    r[2] = pack2<T>(hi[0], hi[1]);
    r[3] = pack2<T>(hi[2], hi[3]);
    return as_frag<typename MfmaOps<T>::frag>(r);
}

// relu of two packed 16-bit floats: as signed 16-bit integers every negative float (sign bit set) is below zero
__device__ __forceinline__ unsigned relu_pk(unsigned d) {
    union { unsigned u; s16x2 s; } a, r;
    a.u = d;
    r.s = __builtin_elementwise_max(a.s, (s16x2){0, 0});
    return r.u;
}
// sign bits of two packed 16-bit floats -> bit 0 and bit 16
__device__ __forceinline__ unsigned signs_pk(unsigned d) {
    union { unsigned u; u16x2 s; } a, r;
    a.u = d;
    r.s = a.s >> (u16x2){15, 15};
    return r.u;
}


}  // namespace afcm
