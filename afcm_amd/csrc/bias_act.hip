// bias_act for gfx950: y = clamp(act(x + b) * gain), plus the first- and second-order backward
// evaluated from the saved input / output, exactly the modes of the reference plugin
// (SG3OPS/bias_act.cu:23-147: grad 0 = forward, 1 = dL/dx from dy, 2 = second order).
// HBM-bound elementwise work: 4 elements per lane, 16-byte (fp32) / 8-byte (16-bit) accesses when
// the tensor size allows, grid-stride over at most 8 blocks per CU.
#include "common.h"

namespace afcm {

struct BiasActParams {
    void* y;
    const void* x;
    const void* b;
    const void* xref;
    const void* yref;
    const void* dy;
    long long numel;
    long long inner;
    int nb;
    int grad;
    float alpha, gain, clamp;
};

constexpr float kExpRange = 80.f;
constexpr float kSeluScale = 1.0507009873554804934193349852946f;
constexpr float kSeluAlpha = 1.6732632423543772848170429916717f;

// One activation = three evaluations: value, first derivative factor, second derivative factor.
// `x` is the forwarded quantity of the mode (input, dy, or d_dx); `yy` is the saved output / gain;
// `xr` is the saved (biased) input.  Formulas are the analytic derivatives written in terms of the
// saved output where the reference does so (bias_act.py:21-31 `ref` column).
template <int A>
__device__ __forceinline__ float act_eval(int G, float x, float yy, float xr, float alpha) {
    if (A == 1) return (G <= 1) ? x : 0.f;                                              // linear
    if (A == 2) return (G == 0) ? fmaxf(x, 0.f) : (G == 1 ? (yy > 0.f ? x : 0.f) : 0.f);  // relu
    if (A == 3) {                                                                       // lrelu
        if (G == 0) return x > 0.f ? x : x * alpha;
        if (G == 1) return yy > 0.f ? x : x * alpha;
        return 0.f;
    }
    if (A == 4) {  // tanh
        if (G == 0) {
            if (x < -kExpRange) return -1.f;
            if (x > kExpRange) return 1.f;
            const float e = expf(x), r = 1.f / e;
            return (e - r) / (e + r);
        }
        const float d1 = 1.f - yy * yy;
        return (G == 1) ? x * d1 : x * d1 * (-2.f * yy);
    }
    if (A == 5) {  // sigmoid
        if (G == 0) return (x < -kExpRange) ? 0.f : 1.f / (expf(-x) + 1.f);
        const float d1 = yy * (1.f - yy);
        return (G == 1) ? x * d1 : x * d1 * (1.f - 2.f * yy);
    }
    if (A == 6) {  // elu
        if (G == 0) return x >= 0.f ? x : expf(x) - 1.f;
        if (G == 1) return yy >= 0.f ? x : x * (yy + 1.f);
        return yy >= 0.f ? 0.f : x * (yy + 1.f);
    }
    if (A == 7) {  // selu
        if (G == 0) return x >= 0.f ? kSeluScale * x : (kSeluScale * kSeluAlpha) * (expf(x) - 1.f);
        if (G == 1) return yy >= 0.f ? x * kSeluScale : x * (yy + kSeluScale * kSeluAlpha);
        return yy >= 0.f ? 0.f : x * (yy + kSeluScale * kSeluAlpha);
    }
    if (A == 8) {  // softplus
        if (G == 0) return x > kExpRange ? x : logf(expf(x) + 1.f);
        const float e = expf(-yy);
        return (G == 1) ? x * (1.f - e) : x * e * (1.f - e);
    }
    // swish: derivatives from the saved input
    if (G == 0) return (x < -kExpRange) ? 0.f : x / (expf(-x) + 1.f);
    const float e = expf(xr), d = e + 1.f;
    if (G == 1) return (xr > 0.5f * kExpRange) ? x : x * e * (xr + d) / (d * d);
    return (xr > 0.5f * kExpRange) ? 0.f : x * e * (xr * (2.f - d) + 2.f * d) / (d * d * d);
}

template <int A>
__device__ __forceinline__ float bias_act_elem(const BiasActParams& p, float x, float b, float xr, float yr, float dy) {
    const int G = p.grad;
    if (G == 0) x += b; else xr += b;
    const float yy = (p.gain != 0.f) ? yr / p.gain : 0.f;
    float y = act_eval<A>(G, x, yy, xr, p.alpha);
    if (A == 9 && G != 0) yr = (xr < -kExpRange) ? 0.f : xr / (expf(-xr) + 1.f) * p.gain;  // swish saves x, not y
    y *= p.gain * dy;
    if (p.clamp >= 0.f) {
        if (G == 0) y = fminf(fmaxf(y, -p.clamp), p.clamp);
        else y = (yr > -p.clamp && yr < p.clamp) ? y : 0.f;
    }
    return y;
}

template <typename T, int A>
__global__ __launch_bounds__(256) void bias_act_kernel(BiasActParams p) {
    const T* x = (const T*)p.x;
    const T* b = (const T*)p.b;
    const T* xr = (const T*)p.xref;
    const T* yr = (const T*)p.yref;
    const T* dy = (const T*)p.dy;
    T* y = (T*)p.y;
    const long long nvec = p.numel >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    // The bias index changes every `inner` elements; a 4-vector shares one bias when inner % 4 == 0.
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        const long long i0 = v << 2;
        float xs[4], xrs[4], yrs[4], dys[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            xs[e] = to_f32(x[i0 + e]);
            xrs[e] = xr ? to_f32(xr[i0 + e]) : 0.f;
            yrs[e] = yr ? to_f32(yr[i0 + e]) : 0.f;
            dys[e] = dy ? to_f32(dy[i0 + e]) : 1.f;
        }
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const float bb = b ? to_f32(b[((i0 + e) / p.inner) % p.nb]) : 0.f;
            y[i0 + e] = from_f32<T>(bias_act_elem<A>(p, xs[e], bb, xrs[e], yrs[e], dys[e]));
        }
    }
    // tail
    for (long long i = (nvec << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < p.numel; i += stride) {
        const float bb = b ? to_f32(b[(i / p.inner) % p.nb]) : 0.f;
        y[i] = from_f32<T>(bias_act_elem<A>(p, to_f32(x[i]), bb, xr ? to_f32(xr[i]) : 0.f, yr ? to_f32(yr[i]) : 0.f,
                                            dy ? to_f32(dy[i]) : 1.f));
    }
}

// 16-byte path: V = 16 / sizeof(T) consecutive elements per lane and trip, ONE 64-bit bias-index division per vector (the
// host takes this path only when inner % V == 0, so a vector never straddles two bias entries, numel % V == 0 and every
// operand is 16-byte aligned).  The element kernel above pays a 64-bit division and a 2-byte access per element: 2.3 TB/s on
// the discriminator's 134 MB bf16 activations.
template <typename T, int A>
__global__ __launch_bounds__(256) void bias_act_vec_kernel(BiasActParams p) {
    constexpr int V = 16 / (int)sizeof(T);
    union Vec { uint4 u; T v[V]; };
    const uint4* x = (const uint4*)p.x;
    const uint4* xr = (const uint4*)p.xref;
    const uint4* yr = (const uint4*)p.yref;
    const uint4* dy = (const uint4*)p.dy;
    const T* b = (const T*)p.b;
    uint4* y = (uint4*)p.y;
    const long long nvec = p.numel / V;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
        Vec xv, xrv, yrv, dyv, out;
        xv.u = x[v];
        if (xr) xrv.u = xr[v];
        if (yr) yrv.u = yr[v];
        if (dy) dyv.u = dy[v];
        const float bb = b ? to_f32(b[((v * V) / p.inner) % p.nb]) : 0.f;
#pragma unroll
        for (int e = 0; e < V; e++)
            out.v[e] = from_f32<T>(bias_act_elem<A>(p, to_f32(xv.v[e]), bb, xr ? to_f32(xrv.v[e]) : 0.f, yr ? to_f32(yrv.v[e]) : 0.f,
                                                    dy ? to_f32(dyv.v[e]) : 1.f));
        y[v] = out.u;
    }
}

template <typename T>
static int launch_bias_act(const BiasActParams& p, int act, hipStream_t st) {
    constexpr int V = 16 / (int)sizeof(T);
    const uintptr_t ptrs = (uintptr_t)p.y | (uintptr_t)p.x | (uintptr_t)p.xref | (uintptr_t)p.yref | (uintptr_t)p.dy;
    const bool vec = (p.numel % V == 0) && (ptrs & 15) == 0 && (p.b == nullptr || p.inner % V == 0);
    long long blocks = ((vec ? p.numel / V : (p.numel >> 2)) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 2048) blocks = 2048;
    dim3 grid((unsigned)blocks), block(256);
#define AFCM_BA_CASE(A) case A: if (vec) hipLaunchKernelGGL((bias_act_vec_kernel<T, A>), grid, block, 0, st, p); \
                                else hipLaunchKernelGGL((bias_act_kernel<T, A>), grid, block, 0, st, p); break;
    switch (act) {
        AFCM_BA_CASE(1) AFCM_BA_CASE(2) AFCM_BA_CASE(3) AFCM_BA_CASE(4) AFCM_BA_CASE(5)
        AFCM_BA_CASE(6) AFCM_BA_CASE(7) AFCM_BA_CASE(8) AFCM_BA_CASE(9)
        default: return AFCM_E_INVALID;
    }
#undef AFCM_BA_CASE
    return hip_status(hipGetLastError());
}

}  // namespace afcm

using namespace afcm;

extern "C" int afcm_bias_act(void* y, const void* x, const void* b, const void* xref, const void* yref, const void* dy,
                             int32_t dtype, int64_t numel, int64_t inner, int32_t nb, int32_t grad, int32_t act, float alpha,
                             float gain, float clamp, void* stream) {
    AFCM_REQUIRE(x != nullptr && y != nullptr, "bias_act: x and y must be non-null");
    AFCM_REQUIRE(numel > 0, "x is empty");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    AFCM_REQUIRE(act >= 1 && act <= 9, "unknown activation index %d", act);
    AFCM_REQUIRE(grad >= 0 && grad <= 2, "grad must be 0, 1 or 2");
    AFCM_REQUIRE(b == nullptr || (nb > 0 && inner > 0), "bias needs nb > 0 and inner > 0");
    BiasActParams p;
    p.y = y; p.x = x; p.b = b; p.xref = xref; p.yref = yref; p.dy = dy;
    p.numel = numel; p.inner = inner > 0 ? inner : 1; p.nb = nb > 0 ? nb : 1;
    p.grad = grad; p.alpha = alpha; p.gain = gain; p.clamp = clamp;
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case AFCM_F32: return launch_bias_act<float>(p, act, st);
        case AFCM_F16: return launch_bias_act<f16_t>(p, act, st);
        default: return launch_bias_act<bf16_t>(p, act, st);
    }
}


// ---- block mean of a plane (r06): AdaptiveAvgPool2d((4, 4)) of the bottleneck (NET:636,683) when the plane divides evenly --------------------------
// y[plane][by][bx] = mean of the (h / 4) x (w / 4) block, fp32, from a 16-bit or fp32 x; backward dx = gy[block] / (block size) in x's type.
// The op-by-op form was a cast to fp32, a reshape + mean (46 us for 8192 planes of 36^2) and their three backward launches.
namespace afcm {
template <typename T>
__global__ __launch_bounds__(256) void pool_blocks_fwd_kernel(float* __restrict__ y, const T* __restrict__ x, long long planes, int h, int w) {
    // one wave per plane; lane = (block b = lane & 15, quarter = lane >> 4): a quarter of the block's elements each, then two shuffles
    const long long plane = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (plane >= planes) return;
    const int lane = threadIdx.x & 63, b = lane & 15, sub = lane >> 4;
    const int bh = h >> 2, bw = w >> 2, by = b >> 2, bx = b & 3;
    const T* xp = x + plane * h * w + (size_t)(by * bh) * w + bx * bw;
    float s = 0.f;
    for (int e = sub; e < bh * bw; e += 4) {
        const int r = e / bw, c = e - r * bw;
        s += to_f32(xp[r * w + c]);
    }
    s += __shfl_xor(s, 16);
    s += __shfl_xor(s, 32);
    if (sub == 0) y[plane * 16 + b] = s / (float)(bh * bw);
}
template <typename T>
__global__ __launch_bounds__(256) void pool_blocks_bwd_kernel(T* __restrict__ dx, const float* __restrict__ gy, long long planes, int h, int w) {
    const int bh = h >> 2, bw = w >> 2;
    const float inv = 1.f / (float)(bh * bw);
    const long long total = planes * h * w;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long plane = idx / (h * w);
        const int rem = (int)(idx - plane * (h * w)), r = rem / w, c = rem - r * w;
        dx[idx] = from_f32<T>(gy[plane * 16 + (r / bh) * 4 + c / bw] * inv);
    }
}
}  // namespace afcm

extern "C" int afcm_pool_blocks_fwd(float* y, const void* x, int32_t dtype, int64_t planes, int32_t h, int32_t w, void* stream) {
    using namespace afcm;
    AFCM_REQUIRE(y != nullptr && x != nullptr && planes > 0 && h >= 4 && w >= 4, "pool_blocks: empty input");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    if ((h & 3) || (w & 3)) return AFCM_E_NOKERNEL;
    const dim3 grid((unsigned)((planes + 3) / 4)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AFCM_F32) hipLaunchKernelGGL((pool_blocks_fwd_kernel<float>), grid, block, 0, st, y, (const float*)x, (long long)planes, h, w);
    else if (dtype == AFCM_F16) hipLaunchKernelGGL((pool_blocks_fwd_kernel<f16_t>), grid, block, 0, st, y, (const f16_t*)x, (long long)planes, h, w);
    else hipLaunchKernelGGL((pool_blocks_fwd_kernel<bf16_t>), grid, block, 0, st, y, (const bf16_t*)x, (long long)planes, h, w);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_pool_blocks_bwd(void* dx, const float* gy, int32_t dtype, int64_t planes, int32_t h, int32_t w, void* stream) {
    using namespace afcm;
    AFCM_REQUIRE(dx != nullptr && gy != nullptr && planes > 0 && h >= 4 && w >= 4, "pool_blocks: empty input");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "dx must be float32, float16 or bfloat16");
    if ((h & 3) || (w & 3)) return AFCM_E_NOKERNEL;
    long long blocks = (planes * h * w + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    const dim3 grid((unsigned)blocks), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (dtype == AFCM_F32) hipLaunchKernelGGL((pool_blocks_bwd_kernel<float>), grid, block, 0, st, (float*)dx, gy, (long long)planes, h, w);
    else if (dtype == AFCM_F16) hipLaunchKernelGGL((pool_blocks_bwd_kernel<f16_t>), grid, block, 0, st, (f16_t*)dx, gy, (long long)planes, h, w);
    else hipLaunchKernelGGL((pool_blocks_bwd_kernel<bf16_t>), grid, block, 0, st, (bf16_t*)dx, gy, (long long)planes, h, w);
    return hip_status(hipGetLastError());
}
