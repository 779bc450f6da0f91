// Shared device/host helpers for libafcm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/afcm_hip.h"

namespace afcm {

// ---- element types -------------------------------------------------------------------------
// Storage types; arithmetic is always fp32.
typedef _Float16 f16_t;
typedef __bf16   bf16_t;

template <typename T> __device__ __forceinline__ float to_f32(T v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v) { return (T)v; }

template <int DT> struct dtype_of;
template <> struct dtype_of<AFCM_F32>  { typedef float  type; };
template <> struct dtype_of<AFCM_F16>  { typedef f16_t  type; };
template <> struct dtype_of<AFCM_BF16> { typedef bf16_t type; };

static inline int dtype_size(int dt) { return dt == AFCM_F32 ? 4 : 2; }

// ---- host-side error reporting ----------------------------------------------------------------
void set_error(const char* fmt, ...);

#define AFCM_REQUIRE(cond, ...)                         \
    do {                                                \
        if (!(cond)) {                                  \
            ::afcm::set_error(__VA_ARGS__);             \
            return AFCM_E_INVALID;                      \
        }                                               \
    } while (0)

static inline int hip_status(hipError_t e) { return e == hipSuccess ? AFCM_OK : 1000 + (int)e; }

constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
constexpr int round_up(int a, int b) { return cdiv(a, b) * b; }

// Workgroup id -> logical work item such that CONSECUTIVE logical items run on ONE XCD (one L2): the hardware deals workgroup
// ids round-robin over the 8 XCDs, so XCD x executes ids x, x + 8, ...  With total = 8 q + r the XCDs below r own q + 1 items:
// XCD x's k-th workgroup takes logical item x q + min(x, r) + k -- a bijection for every grid size (r02 remapped only grids that
// are multiples of 8: the 91- and 181-channel layers, grids of 3276 / 3620 / 6516 workgroups, ran with neighbouring strips on
// different XCDs and fetched their shared halo rows twice: 1.33x the algorithmic HBM bytes against 1.12x on the remapped layers).
// Speed only: nothing may depend on where a workgroup actually runs.
__device__ __forceinline__ int xcd_order(int bid, int total) {
    const int x = bid & 7, k = bid >> 3, q = total >> 3, r = total & 7;
    return x * q + (x < r ? x : r) + k;
}
constexpr int cmax(int a, int b) { return a > b ? a : b; }
// smallest odd multiple of 4 that is >= a (a itself a multiple of 4)
constexpr int odd4(int a) { return ((a / 4) % 2 == 1) ? a : a + 4; }

// floor division / modulo for possibly negative numerators (b > 0)
__host__ __device__ __forceinline__ int floor_div(int a, int b) {
    int q = a / b;
    return (a % b != 0 && (a < 0)) ? q - 1 : q;
}
__host__ __device__ __forceinline__ int pos_mod(int a, int b) {
    int r = a % b;
    return r < 0 ? r + b : r;
}

}  // namespace afcm
