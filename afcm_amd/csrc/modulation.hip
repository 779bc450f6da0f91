// The small-tensor half of modulated_conv2d (NET:41-57) as four launches instead of ~35 eager ones per layer and step:
//   weight normalisation   w_hat = w * rsqrt(mean_{i,k} w^2)            (NET:42)      + wsq[o,i] = sum_k w_hat^2
//   style coefficients     s_hat = t * rsqrt(mean_{n,i} t^2)            (NET:43, whole batch)
//                          d[n,o] = rsqrt(sum_i s_hat[n,i]^2 wsq[o,i] + 1e-8)   (NET:50-52, factorised: the per-sample
//                                   weights w[n,o,i,k] = w_hat[o,i,k] s_hat[n,i] are never materialised)
//                          s_eff = s_hat * input_gain                   (NET:55-57)
// and their exact backward passes.  All tensors fp32, sizes <= 512 x 512 x 9: launch-bound work, one workgroup per
// output row / sample, no attempt at bandwidth.
#include "common.h"

namespace afcm {

__device__ __forceinline__ float block_sum(float v, float* red) {
    // 256 threads: wave shuffles, then 4 partials through LDS; every thread returns the total
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// one workgroup per output channel o.  PT > 0: the row (I * KK <= 256 * PT elements) is read ONCE, all loads in flight, and
// stays in registers for the three uses (sum of squares, w_hat, wsq through an LDS copy); PT == 0: any row length, three passes.
template <int PT>
__device__ __forceinline__ void weight_norm_fwd_body(float* red, float* row, int o, float* __restrict__ w_hat, float* __restrict__ wsq,
                                                     float* __restrict__ scale, const float* __restrict__ w, int I, int KK) {
    const int n = I * KK;
    const float* wo = w + (size_t)o * n;
    float ss = 0.f;
    if constexpr (PT > 0) {
        float v[PT];
#pragma unroll
        for (int k = 0; k < PT; k++) { const int j = threadIdx.x + 256 * k; v[k] = j < n ? wo[j] : 0.f; }
#pragma unroll
        for (int k = 0; k < PT; k++) ss += v[k] * v[k];
        ss = block_sum(ss, red);
        const float sc = rsqrtf(ss / (float)n);
        if (threadIdx.x == 0) scale[o] = sc;
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const int j = threadIdx.x + 256 * k;
            const float wh = v[k] * sc;
            row[j] = wh;
            if (j < n) w_hat[(size_t)o * n + j] = wh;
        }
        __syncthreads();
        for (int i = threadIdx.x; i < I; i += 256) {
            float q = 0.f;
            for (int k = 0; k < KK; k++) { const float t = row[i * KK + k]; q += t * t; }
            wsq[(size_t)o * I + i] = q;
        }
    } else {
        for (int j = threadIdx.x; j < n; j += 256) ss += wo[j] * wo[j];
        ss = block_sum(ss, red);
        const float sc = rsqrtf(ss / (float)n);
        if (threadIdx.x == 0) scale[o] = sc;
        for (int j = threadIdx.x; j < n; j += 256) w_hat[(size_t)o * n + j] = wo[j] * sc;
        for (int i = threadIdx.x; i < I; i += 256) {
            float q = 0.f;
            for (int k = 0; k < KK; k++) { const float t = wo[i * KK + k] * sc; q += t * t; }
            wsq[(size_t)o * I + i] = q;
        }
    }
}

template <int PT>
__global__ __launch_bounds__(256) void weight_norm_fwd_kernel(float* __restrict__ w_hat, float* __restrict__ wsq, float* __restrict__ scale,
                                                              const float* __restrict__ w, int I, int KK) {
    __shared__ float red[4];
    __shared__ float row[PT > 0 ? 256 * PT : 1];
    weight_norm_fwd_body<PT>(red, row, blockIdx.x, w_hat, wsq, scale, w, I, KK);
}

// dw = scale * (G - w_hat * mean(G . w_hat)),  G = g_hat + 2 w_hat g_wsq[o,i].  PT as above: operands read once, kept in registers.
template <int PT>
__device__ __forceinline__ void weight_norm_bwd_body(float* red, int o, float* __restrict__ dw, const float* __restrict__ g_hat,
                                                     const float* __restrict__ g_wsq, const float* __restrict__ w_hat,
                                                     const float* __restrict__ scale, int I, int KK) {
    const int n = I * KK;
    const size_t base = (size_t)o * n;
    float dot = 0.f;
    if constexpr (PT > 0) {
        float wh[PT], g[PT];
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const int j = threadIdx.x + 256 * k;
            const bool ok = j < n;
            wh[k] = ok ? w_hat[base + j] : 0.f;
            g[k] = (ok && g_hat) ? g_hat[base + j] : 0.f;
            if (ok && g_wsq) g[k] += 2.f * wh[k] * g_wsq[(size_t)o * I + j / KK];
        }
#pragma unroll
        for (int k = 0; k < PT; k++) dot += g[k] * wh[k];
        dot = block_sum(dot, red);
        const float m = dot / (float)n, sc = scale[o];
#pragma unroll
        for (int k = 0; k < PT; k++) {
            const int j = threadIdx.x + 256 * k;
            if (j < n) dw[base + j] = sc * (g[k] - wh[k] * m);
        }
    } else {
        for (int j = threadIdx.x; j < n; j += 256) {
            const float wh = w_hat[base + j];
            float g = g_hat ? g_hat[base + j] : 0.f;
            if (g_wsq) g += 2.f * wh * g_wsq[(size_t)o * I + j / KK];
            dot += g * wh;
        }
        dot = block_sum(dot, red);
        const float m = dot / (float)n, sc = scale[o];
        for (int j = threadIdx.x; j < n; j += 256) {
            const float wh = w_hat[base + j];
            float g = g_hat ? g_hat[base + j] : 0.f;
            if (g_wsq) g += 2.f * wh * g_wsq[(size_t)o * I + j / KK];
            dw[base + j] = sc * (g - wh * m);
        }
    }
}

template <int PT>
__global__ __launch_bounds__(256) void weight_norm_bwd_kernel(float* __restrict__ dw, const float* __restrict__ g_hat, const float* __restrict__ g_wsq,
                                                              const float* __restrict__ w_hat, const float* __restrict__ scale, int I, int KK) {
    __shared__ float red[4];
    weight_norm_bwd_body<PT>(red, blockIdx.x, dw, g_hat, g_wsq, w_hat, scale, I, KK);
}

// workgroups (n, o-chunk of 32): every workgroup recomputes the batch-wide mean (N * I <= a few thousand elements) and its
// sample's s_hat^2 row; a wave per output channel reads the wsq row coalesced (lanes over i) and reduces by shuffles
__device__ __forceinline__ void style_coefs_fwd_body(float* red, float* s2, int n, int oc, float* __restrict__ s_eff, float* __restrict__ d,
                                                     float* __restrict__ r_out, const float* __restrict__ t, const float* __restrict__ wsq,
                                                     const float* __restrict__ magnitude, int N, int I, int O, int demod) {
    float r = 1.f;
    if (demod) {
        float ss = 0.f;
        float s4[4] = {0.f, 0.f, 0.f, 0.f};
        int j = threadIdx.x;
        for (; j + 768 < N * I; j += 1024) {                       // four independent loads per trip
            const float a0 = t[j], a1 = t[j + 256], a2 = t[j + 512], a3 = t[j + 768];
            s4[0] += a0 * a0; s4[1] += a1 * a1; s4[2] += a2 * a2; s4[3] += a3 * a3;
        }
        for (; j < N * I; j += 256) s4[0] += t[j] * t[j];
        ss = block_sum((s4[0] + s4[1]) + (s4[2] + s4[3]), red);
        r = rsqrtf(ss / (float)(N * I));
    }
    if (n == 0 && oc == 0 && threadIdx.x == 0) r_out[0] = r;
    const float g = magnitude ? rsqrtf(magnitude[0]) : 1.f;           // input_gain = magnitude_ema.rsqrt() (NET:346,55-57)
    for (int i = threadIdx.x; i < I; i += 256) {
        const float sh = t[(size_t)n * I + i] * r;
        if (oc == 0) s_eff[(size_t)n * I + i] = sh * g;
        s2[i] = sh * sh;
    }
    if (!demod) return;
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // the wave's 8 output channels together: 8 independent row loads per i step (one row at a time left each load waiting
    // for the previous row's reduction)
    const int ob = oc * 32 + wave * 8;
    float q[8];
#pragma unroll
    for (int j = 0; j < 8; j++) q[j] = 0.f;
    for (int i = lane; i < I; i += 64) {
        const float sv = s2[i];
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const int o = ob + j < O ? ob + j : O - 1;            // clamped: rows past O are computed and dropped
            q[j] = fmaf(sv, wsq[(size_t)o * I + i], q[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 8; j++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) q[j] += __shfl_xor(q[j], off, 64);
        if (lane == 0 && ob + j < O) d[(size_t)n * O + ob + j] = rsqrtf(q[j] + 1e-8f);
    }
}

__global__ __launch_bounds__(256) void style_coefs_fwd_kernel(float* __restrict__ s_eff, float* __restrict__ d, float* __restrict__ r_out,
                                                              const float* __restrict__ t, const float* __restrict__ wsq,
                                                              const float* __restrict__ magnitude, int N, int I, int O, int demod) {
    __shared__ float red[4];
    extern __shared__ float s2[];                       // s_hat[n, :]^2
    style_coefs_fwd_body(red, s2, blockIdx.x, blockIdx.y, s_eff, d, r_out, t, wsq, magnitude, N, I, O, demod);
}

// phase 1, workgroups (n, i-chunk of 64): Q[n,o] = -1/2 g_d d^3;  G[n,i] = gain g_s + 2 s_hat[n,i] sum_o Q[n,o] wsq[o,i];
// partial[n, chunk] = sum_{i in chunk} G[n,i] s_hat[n,i].  The sum over o is split four ways over the workgroup's waves.
__device__ __forceinline__ void style_coefs_bwd1_body(float (*us)[64], float* q_s, int n, int ib, int IB, float* __restrict__ G,
                                                      float* __restrict__ Q, float* __restrict__ partial, const float* __restrict__ g_s,
                                                      const float* __restrict__ g_d, const float* __restrict__ t, const float* __restrict__ d,
                                                      const float* __restrict__ wsq, const float* __restrict__ magnitude,
                                                      const float* __restrict__ r_in, int I, int O, int demod) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int i = ib * 64 + tx;
    const float r = r_in[0], g = magnitude ? rsqrtf(magnitude[0]) : 1.f;
    if (demod) {
        for (int o = threadIdx.x; o < O; o += 256) {
            const float dd = d[(size_t)n * O + o];
            const float q = g_d ? -0.5f * g_d[(size_t)n * O + o] * dd * dd * dd : 0.f;
            q_s[o] = q;
            if (ib == 0) Q[(size_t)n * O + o] = q;
        }
        __syncthreads();
        float u = 0.f;
        if (i < I) {
#pragma unroll 8
            for (int o = ty; o < O; o += 4) u = fmaf(q_s[o], wsq[(size_t)o * I + i], u);
        }
        us[ty][tx] = u;
        __syncthreads();
    }
    if (ty != 0) return;
    float dot = 0.f;
    if (i < I) {
        const float sh = t[(size_t)n * I + i] * r;
        float gg = g_s ? g * g_s[(size_t)n * I + i] : 0.f;
        if (demod) gg += 2.f * sh * (us[0][tx] + us[1][tx] + us[2][tx] + us[3][tx]);
        G[(size_t)n * I + i] = gg;
        dot = gg * sh;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dot += __shfl_xor(dot, off, 64);
    if (tx == 0) partial[(size_t)n * IB + ib] = dot;
}

__global__ __launch_bounds__(256) void style_coefs_bwd1_kernel(float* __restrict__ G, float* __restrict__ Q, float* __restrict__ partial,
                                                               const float* __restrict__ g_s, const float* __restrict__ g_d,
                                                               const float* __restrict__ t, const float* __restrict__ d, const float* __restrict__ wsq,
                                                               const float* __restrict__ magnitude, const float* __restrict__ r_in, int I, int O, int demod) {
    __shared__ float us[4][64];
    extern __shared__ float q_s[];                      // Q[n, :]
    style_coefs_bwd1_body(us, q_s, blockIdx.x, blockIdx.y, gridDim.y, G, Q, partial, g_s, g_d, t, d, wsq, magnitude, r_in, I, O, demod);
}

// phase 2: workgroups [0, N): dt[n,:] = r (G - s_hat mean(G . s_hat)) (demod) or G (no demod);
//          workgroups [N, N+O): g_wsq[o,i] = sum_n Q[n,o] s_hat[n,i]^2
__device__ __forceinline__ void style_coefs_bwd2_body(int bx, float* __restrict__ dt, float* __restrict__ g_wsq, const float* __restrict__ G,
                                                      const float* __restrict__ Q, const float* __restrict__ partial, const float* __restrict__ t,
                                                      const float* __restrict__ r_in, int N, int I, int O, int demod, int NP) {
    const float r = r_in[0];
    if (bx < N) {
        const int n = bx;
        float m = 0.f;
        if (demod) {
            for (int k = 0; k < NP; k++) m += partial[k];
            m /= (float)(N * I);
        }
        for (int i = threadIdx.x; i < I; i += 256) {
            const float gg = G[(size_t)n * I + i];
            dt[(size_t)n * I + i] = demod ? r * (gg - t[(size_t)n * I + i] * r * m) : gg;
        }
    } else if (demod && g_wsq != nullptr) {
        const int o = bx - N;
        for (int i = threadIdx.x; i < I; i += 256) {
            float acc = 0.f;
            for (int n = 0; n < N; n++) {
                const float sh = t[(size_t)n * I + i] * r;
                acc += Q[(size_t)n * O + o] * sh * sh;
            }
            g_wsq[(size_t)o * I + i] = acc;
        }
    }
}

__global__ __launch_bounds__(256) void style_coefs_bwd2_kernel(float* __restrict__ dt, float* __restrict__ g_wsq, const float* __restrict__ G,
                                                               const float* __restrict__ Q, const float* __restrict__ partial,
                                                               const float* __restrict__ t, const float* __restrict__ r_in, int N, int I, int O, int demod, int NP) {
    style_coefs_bwd2_body((int)blockIdx.x, dt, g_wsq, G, Q, partial, t, r_in, N, I, O, demod, NP);
}

// Small-tensor tail of a fused layer's backward (torch_utils/ops/fused_layer.py), one workgroup (one wave) per output channel:
//   ps[n,o]     = sum over the tile slots of the plane sums the transposed filtered_lrelu emitted (sums of dys)
//   db[o]       = sum_n ps / d                      (bias gradient: dy = dys / d)
//   d_next[n,o] = <g, z> / s_next                   (0 where s_next == 0)
//   d_out[n,o]  = (<dys, y> - b ps) / d^2           (y = d c + b)
__global__ __launch_bounds__(64) void layer_bwd_coefs_kernel(float* __restrict__ db, float* __restrict__ d_next, float* __restrict__ d_out,
                                                             const float* __restrict__ psum, int slots, const float* __restrict__ out_scale,
                                                             const float* __restrict__ next_scale, const float* __restrict__ bias,
                                                             const float* __restrict__ gz, const float* __restrict__ dysy, int N, int O) {
    const int o = blockIdx.x;
    const float b = bias ? bias[o] : 0.f;
    float acc = 0.f;
    for (int n = threadIdx.x; n < N; n += 64) {
        const size_t idx = (size_t)n * O + o;
        float ps = 0.f;
        for (int k = 0; k < slots; k++) ps += psum[idx * slots + k];
        const float d = out_scale ? out_scale[idx] : 1.f;
        acc += ps / d;
        if (d_next) {
            const float ns = next_scale[idx];
            d_next[idx] = ns != 0.f ? gz[idx] / ns : 0.f;
        }
        if (d_out) d_out[idx] = (dysy[idx] - b * ps) / (d * d);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (db && threadIdx.x == 0) db[o] = acc;
}

// ---- the bank: the kernels above over a list of layers in one launch each.  The layer table rides in the kernel arguments (by value);
// blk[] holds the first workgroup of every layer for the launch at hand, a workgroup finds its layer by a scalar scan.
struct ModBank {
    int count, n;
    int blk[AFCM_MODULATION_MAX + 1];
    afcm_modulation_layer L[AFCM_MODULATION_MAX];
};

__device__ __forceinline__ int bank_layer(const ModBank& b, int bid) {
    int l = 0;
    while (l + 1 < b.count && bid >= b.blk[l + 1]) l++;
    return l;
}

__global__ __launch_bounds__(256) void modulation_bank_norm_fwd_kernel(const ModBank b) {
    __shared__ float red[4];
    __shared__ float row[256 * 18];
    const int l = bank_layer(b, blockIdx.x), o = blockIdx.x - b.blk[l];
    const afcm_modulation_layer& L = b.L[l];
    const int n = L.cin * L.kk;
    if (n <= 256 * 4) weight_norm_fwd_body<4>(red, row, o, L.w_hat, L.wsq, L.scale, L.w, L.cin, L.kk);
    else if (n <= 256 * 18) weight_norm_fwd_body<18>(red, row, o, L.w_hat, L.wsq, L.scale, L.w, L.cin, L.kk);
    else weight_norm_fwd_body<0>(red, row, o, L.w_hat, L.wsq, L.scale, L.w, L.cin, L.kk);
}

__global__ __launch_bounds__(256) void modulation_bank_coefs_fwd_kernel(const ModBank b) {
    __shared__ float red[4];
    extern __shared__ float dyn[];
    const int l = bank_layer(b, blockIdx.x), loc = blockIdx.x - b.blk[l];
    const afcm_modulation_layer& L = b.L[l];
    const int ocs = L.demodulate ? (L.cout + 31) / 32 : 1;
    style_coefs_fwd_body(red, dyn, loc / ocs, loc % ocs, L.s_eff, L.d, L.r, L.t, L.wsq, L.magnitude, b.n, L.cin, L.cout, L.demodulate);
}

// workspace of a layer: G [n, cin] | Q [n, cout] | partial [n, ceil(cin / 64)] | g_wsq [cout, cin]   (Q and g_wsq: demodulating layers)
__device__ __forceinline__ void bank_workspace(const afcm_modulation_layer& L, int n, float*& G, float*& Q, float*& partial, float*& g_wsq) {
    G = L.workspace;
    Q = G + (size_t)n * L.cin;
    partial = Q + (size_t)n * (L.demodulate ? L.cout : 0);
    g_wsq = partial + (size_t)n * ((L.cin + 63) / 64);
}

__global__ __launch_bounds__(256) void modulation_bank_coefs_bwd1_kernel(const ModBank b) {
    __shared__ float us[4][64];
    extern __shared__ float dyn[];
    const int l = bank_layer(b, blockIdx.x), loc = blockIdx.x - b.blk[l];
    const afcm_modulation_layer& L = b.L[l];
    const int IB = (L.cin + 63) / 64;
    float *G, *Q, *partial, *g_wsq;
    bank_workspace(L, b.n, G, Q, partial, g_wsq);
    style_coefs_bwd1_body(us, dyn, loc / IB, loc % IB, IB, G, Q, partial, L.g_s, L.g_d, L.t, L.d, L.wsq, L.magnitude, L.r, L.cin, L.cout, L.demodulate);
}

__global__ __launch_bounds__(256) void modulation_bank_coefs_bwd2_kernel(const ModBank b) {
    const int l = bank_layer(b, blockIdx.x), loc = blockIdx.x - b.blk[l];
    const afcm_modulation_layer& L = b.L[l];
    float *G, *Q, *partial, *g_wsq;
    bank_workspace(L, b.n, G, Q, partial, g_wsq);
    style_coefs_bwd2_body(loc, L.dt, g_wsq, G, Q, partial, L.t, L.r, b.n, L.cin, L.cout, L.demodulate, b.n * ((L.cin + 63) / 64));
}

__global__ __launch_bounds__(256) void modulation_bank_norm_bwd_kernel(const ModBank b) {
    __shared__ float red[4];
    const int l = bank_layer(b, blockIdx.x), o = blockIdx.x - b.blk[l];
    const afcm_modulation_layer& L = b.L[l];
    float *G, *Q, *partial, *g_wsq;
    bank_workspace(L, b.n, G, Q, partial, g_wsq);
    const int n = L.cin * L.kk;
    if (n <= 256 * 4) weight_norm_bwd_body<4>(red, o, L.dw, L.g_hat, g_wsq, L.w_hat, L.scale, L.cin, L.kk);
    else if (n <= 256 * 18) weight_norm_bwd_body<18>(red, o, L.dw, L.g_hat, g_wsq, L.w_hat, L.scale, L.cin, L.kk);
    else weight_norm_bwd_body<0>(red, o, L.dw, L.g_hat, g_wsq, L.w_hat, L.scale, L.cin, L.kk);
}

}  // namespace afcm

using namespace afcm;

extern "C" int afcm_layer_bwd_coefs(float* db, float* d_next, float* d_out, const float* psum, int32_t slots, const float* out_scale,
                                    const float* next_scale, const float* bias, const float* gz, const float* dysy, int32_t n, int32_t cout,
                                    void* stream) {
    AFCM_REQUIRE(psum && slots > 0 && n > 0 && cout > 0, "layer_bwd_coefs: bad arguments");
    AFCM_REQUIRE(!d_next || (next_scale && gz), "layer_bwd_coefs: d_next needs next_scale and <g, z>");
    AFCM_REQUIRE(!d_out || (out_scale && dysy), "layer_bwd_coefs: d_out needs out_scale and <dys, y>");
    hipLaunchKernelGGL(layer_bwd_coefs_kernel, dim3(cout), dim3(64), 0, (hipStream_t)stream, db, d_next, d_out, psum, slots, out_scale, next_scale,
                       bias, gz, dysy, n, cout);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_weight_norm_fwd(float* w_hat, float* wsq, float* scale, const float* w, int32_t cout, int32_t cin, int32_t kk, void* stream) {
    AFCM_REQUIRE(w_hat && wsq && scale && w && cout > 0 && cin > 0 && kk > 0, "weight_norm_fwd: bad arguments");
    const int n = cin * kk;
    if (n <= 256 * 4) hipLaunchKernelGGL(weight_norm_fwd_kernel<4>, dim3(cout), dim3(256), 0, (hipStream_t)stream, w_hat, wsq, scale, w, cin, kk);
    else if (n <= 256 * 18) hipLaunchKernelGGL(weight_norm_fwd_kernel<18>, dim3(cout), dim3(256), 0, (hipStream_t)stream, w_hat, wsq, scale, w, cin, kk);
    else hipLaunchKernelGGL(weight_norm_fwd_kernel<0>, dim3(cout), dim3(256), 0, (hipStream_t)stream, w_hat, wsq, scale, w, cin, kk);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_weight_norm_bwd(float* dw, const float* g_hat, const float* g_wsq, const float* w_hat, const float* scale, int32_t cout,
                                    int32_t cin, int32_t kk, void* stream) {
    AFCM_REQUIRE(dw && w_hat && scale && cout > 0 && cin > 0 && kk > 0, "weight_norm_bwd: bad arguments");
    const int n = cin * kk;
    if (n <= 256 * 4) hipLaunchKernelGGL(weight_norm_bwd_kernel<4>, dim3(cout), dim3(256), 0, (hipStream_t)stream, dw, g_hat, g_wsq, w_hat, scale, cin, kk);
    else if (n <= 256 * 18) hipLaunchKernelGGL(weight_norm_bwd_kernel<18>, dim3(cout), dim3(256), 0, (hipStream_t)stream, dw, g_hat, g_wsq, w_hat, scale, cin, kk);
    else hipLaunchKernelGGL(weight_norm_bwd_kernel<0>, dim3(cout), dim3(256), 0, (hipStream_t)stream, dw, g_hat, g_wsq, w_hat, scale, cin, kk);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_style_coefs_fwd(float* s_eff, float* d, float* r, const float* t, const float* wsq, const float* magnitude, int32_t n, int32_t cin,
                                    int32_t cout, int32_t demodulate, void* stream) {
    AFCM_REQUIRE(s_eff && r && t && n > 0 && cin > 0 && (!demodulate || (d && wsq && cout > 0)), "style_coefs_fwd: bad arguments");
    AFCM_REQUIRE(cin <= 16384, "style_coefs_fwd: %d input channels exceed the LDS row", cin);
    hipLaunchKernelGGL(style_coefs_fwd_kernel, dim3(n, demodulate ? (cout + 31) / 32 : 1), dim3(256), cin * sizeof(float), (hipStream_t)stream, s_eff,
                       d, r, t, wsq, magnitude, n, cin, cout, demodulate);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_style_coefs_bwd(float* dt, float* g_wsq, float* workspace, const float* g_s, const float* g_d, const float* t, const float* d,
                                    const float* wsq, const float* magnitude, const float* r, int32_t n, int32_t cin, int32_t cout, int32_t demodulate,
                                    void* stream) {
    AFCM_REQUIRE(dt && workspace && t && r && n > 0 && cin > 0 && (!demodulate || (d && wsq && cout > 0)), "style_coefs_bwd: bad arguments");
    AFCM_REQUIRE(cout <= 16384, "style_coefs_bwd: %d output channels exceed the LDS row", cout);
    // workspace: G [n, cin] | Q [n, cout] | partial [n, ceil(cin / 64)]
    float* G = workspace;
    float* Q = G + (size_t)n * cin;
    float* partial = Q + (size_t)n * (demodulate ? cout : 0);
    hipStream_t st = (hipStream_t)stream;
    const int ib = (cin + 63) / 64;
    hipLaunchKernelGGL(style_coefs_bwd1_kernel, dim3(n, ib), dim3(256), (demodulate ? cout : 1) * sizeof(float), st, G, Q, partial, g_s, g_d, t, d, wsq,
                       magnitude, r, cin, cout, demodulate);
    hipLaunchKernelGGL(style_coefs_bwd2_kernel, dim3(n + ((demodulate && g_wsq) ? cout : 0)), dim3(256), 0, st, dt, g_wsq, G, Q, partial, t, r, n, cin,
                       cout, demodulate, n * ib);
    return hip_status(hipGetLastError());
}

// ---- modulation bank (include/afcm_hip.h) ----
static int bank_check(const afcm_modulation_layer* layers, int32_t count, int32_t n, bool bwd) {
    AFCM_REQUIRE(layers && count > 0 && count <= AFCM_MODULATION_MAX && n > 0, "modulation_bank: bad arguments (1..%d layers)", AFCM_MODULATION_MAX);
    for (int l = 0; l < count; l++) {
        const afcm_modulation_layer& L = layers[l];
        AFCM_REQUIRE(L.cin > 0 && L.cin <= 16384 && L.t && L.r, "modulation_bank: layer %d: styles / r missing or cin outside 1..16384", l);
        if (L.demodulate)
            AFCM_REQUIRE(L.cout > 0 && L.cout <= 16384 && L.kk > 0 && L.w_hat && L.wsq && L.scale && L.d,
                         "modulation_bank: layer %d demodulates: w_hat / wsq / scale / d needed, cout in 1..16384", l);
        if (bwd) AFCM_REQUIRE(L.dt && L.workspace, "modulation_bank: layer %d: dt / workspace missing", l);
        else AFCM_REQUIRE(L.s_eff && (!L.demodulate || L.w), "modulation_bank: layer %d: s_eff / w missing", l);
    }
    return AFCM_OK;
}

extern "C" int64_t afcm_modulation_bank_workspace_floats(int32_t n, int32_t cin, int32_t cout, int32_t demodulate) {
    if (n <= 0 || cin <= 0 || (demodulate && cout <= 0)) return -1;
    return (int64_t)n * cin + (int64_t)n * ((cin + 63) / 64) + (demodulate ? (int64_t)n * cout + (int64_t)cout * cin : 0);
}

extern "C" int afcm_modulation_bank_fwd(const afcm_modulation_layer* layers, int32_t count, int32_t n, void* stream) {
    if (int rc = bank_check(layers, count, n, false)) return rc;
    hipStream_t st = (hipStream_t)stream;
    ModBank b;
    b.count = count;
    b.n = n;
    int max_cin = 1;
    for (int l = 0; l < count; l++) { b.L[l] = layers[l]; max_cin = layers[l].cin > max_cin ? layers[l].cin : max_cin; }
    int tot = 0;
    for (int l = 0; l < count; l++) { b.blk[l] = tot; tot += layers[l].demodulate ? layers[l].cout : 0; }
    b.blk[count] = tot;
    if (tot > 0) hipLaunchKernelGGL(modulation_bank_norm_fwd_kernel, dim3(tot), dim3(256), 0, st, b);
    tot = 0;
    for (int l = 0; l < count; l++) { b.blk[l] = tot; tot += n * (layers[l].demodulate ? (layers[l].cout + 31) / 32 : 1); }
    b.blk[count] = tot;
    hipLaunchKernelGGL(modulation_bank_coefs_fwd_kernel, dim3(tot), dim3(256), max_cin * sizeof(float), st, b);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_modulation_bank_bwd(const afcm_modulation_layer* layers, int32_t count, int32_t n, void* stream) {
    if (int rc = bank_check(layers, count, n, true)) return rc;
    hipStream_t st = (hipStream_t)stream;
    ModBank b;
    b.count = count;
    b.n = n;
    int max_cout = 1;
    for (int l = 0; l < count; l++) { b.L[l] = layers[l]; if (layers[l].demodulate && layers[l].cout > max_cout) max_cout = layers[l].cout; }
    int tot = 0;
    for (int l = 0; l < count; l++) { b.blk[l] = tot; tot += n * ((layers[l].cin + 63) / 64); }
    b.blk[count] = tot;
    hipLaunchKernelGGL(modulation_bank_coefs_bwd1_kernel, dim3(tot), dim3(256), max_cout * sizeof(float), st, b);
    tot = 0;
    for (int l = 0; l < count; l++) { b.blk[l] = tot; tot += n + ((layers[l].demodulate && layers[l].dw) ? layers[l].cout : 0); }
    b.blk[count] = tot;
    hipLaunchKernelGGL(modulation_bank_coefs_bwd2_kernel, dim3(tot), dim3(256), 0, st, b);
    tot = 0;
    for (int l = 0; l < count; l++) { b.blk[l] = tot; tot += (layers[l].demodulate && layers[l].dw) ? layers[l].cout : 0; }
    b.blk[count] = tot;
    if (tot > 0) hipLaunchKernelGGL(modulation_bank_norm_bwd_kernel, dim3(tot), dim3(256), 0, st, b);
    return hip_status(hipGetLastError());
}
