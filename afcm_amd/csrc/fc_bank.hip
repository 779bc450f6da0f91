// Equalised-learning-rate dense layers of the mapping network (NET:69-104, 109-164) and of the bottleneck's `fc_in` (NET:637,684) as ONE
// launch per layer and direction, on the exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32: bit for bit an fp32 fmaf chain).
//
// The framework route costs a layer four launches forward (GEMM, bias_act) and six backward (bias_act', two GEMMs, a column sum, two
// scalar multiplies); with 8 mapping layers + the embedding and its two normalisations that was ~100 of the step's 260 launches outside
// the three hot families, every one a 3-5 us kernel behind a 1.5 us boundary.  The batch is 16-64 rows: a layer is a
// stream over its weight matrix (1-2 MB) with 16 / 32 / 64 columns of work per weight -- latency and launch count, not arithmetic.
//
//   forward   y[n][o] = act(alpha * sum_k x[n][k] w[o][k] + beta * b[o])         act: identity, or lrelu(0.2) * sqrt(2) (bias_act.py:21-31)
//   backward  gp = gy * act'(y)   (from the saved OUTPUT, as bias_act.cu:68-73 does)
//             dx[n][k] = alpha * sum_o gp[n][o] w[o][k],   dw[o][k] = alpha * sum_n gp[n][o] x[n][k],   db[o] = beta * sum_n gp[n][o]
//
// Forward: a workgroup owns 16 output rows; its 4 waves split K, partial tiles meet in LDS.  Backward: a workgroup owns 16 COLUMNS k of the
// weight matrix -- the dx columns and the dw columns it produces need the same operands (all of gp, x[:, k-slab], w[:, k-slab]); its 4
// waves split the rows o.  Lane maps (cdna_hip_programming.md section 3): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// D[i = 4 (lane >> 4) + reg][j = lane & 15].  A 16-byte load gives a lane four consecutive contraction indices: the four components feed
// four MFMAs, whose k slot s = lane >> 4 then stands for index 4 s + component -- the same on both operands, so the sum is complete.
#include "common.h"

namespace afcm {

typedef __attribute__((ext_vector_type(4))) float fcx4;

enum { FC_LINEAR = 0, FC_LRELU = 1 };
constexpr float kFcSlope = 0.2f, kFcGain = 1.41421356237309504880f;

struct FcArgs {
    float* y; const float* x; const float* w; const float* b;
    int n, cin, cout, act;
    float alpha, beta;
};

// NW waves per workgroup split K (4; 16 for long rows: `fc_in`'s 4608-float rows on 64 workgroups x 4 waves kept 2 MB in flight -- 47 us for
// its 18.9 MB)
template <int NT, int NW>
__global__ __launch_bounds__(64 * NW) void fc_act_fwd_kernel(FcArgs p) {
    __shared__ fcx4 red[NW - 1][NT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int o0 = blockIdx.x * 16;
    const int K = p.cin;
    const int orow = min(o0 + l15, p.cout - 1);
    const float* wp = p.w + (size_t)orow * K + 4 * g;
    const float* xp[NT];
    bool live[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = t * 16 + l15;
        live[t] = n < p.n;
        xp[t] = p.x + (size_t)min(n, p.n - 1) * K + 4 * g;
    }
    fcx4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) acc[t] = fcx4{0.f, 0.f, 0.f, 0.f};
    // 16 contraction indices per chunk; wave w takes a CONTIGUOUS quarter of the chunks (r06: interleaved by wave, a row's requests were 64
    // bytes every 256 -- fc_in's 18.9 MB weight matrix streamed at 0.34 TB/s; a wave now walks 512 contiguous bytes of each row per pass)
    const int chunks = K >> 4;
    const int per = (chunks + NW - 1) / NW;
    const int c_end = min(chunks, (wave + 1) * per);
    constexpr int U = 8;                                        // chunks in flight per wave
    int c = wave * per;
    for (; c + U <= c_end; c += U) {
        fcx4 a[U], bv[U][NT];
#pragma unroll
        for (int u = 0; u < U; u++) {
            a[u] = *(const fcx4*)(wp + 16 * (c + u));
#pragma unroll
            for (int t = 0; t < NT; t++) bv[u][t] = *(const fcx4*)(xp[t] + 16 * (c + u));
        }
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const fcx4 bb = live[t] ? bv[u][t] : fcx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < 4; j++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], bb[j], acc[t], 0, 0, 0);
            }
    }
    for (; c < c_end; c++) {
        const fcx4 a = *(const fcx4*)(wp + 16 * c);
#pragma unroll
        for (int t = 0; t < NT; t++) {
            fcx4 bb = *(const fcx4*)(xp[t] + 16 * c);
            if (!live[t]) bb = fcx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bb[j], acc[t], 0, 0, 0);
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; t++) red[wave - 1][t][lane] = acc[t];
    }
    __syncthreads();
    if (wave != 0) return;
    // epilogue: lane holds rows o0 + 4 g + r (r = 0..3) of sample n = 16 t + l15
    const int ob = o0 + 4 * g;
    float bias[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.b != nullptr) {
#pragma unroll
        for (int r = 0; r < 4; r++) bias[r] = ob + r < p.cout ? p.b[ob + r] * p.beta : 0.f;
    }
#pragma unroll
    for (int t = 0; t < NT; t++) {
        const int n = t * 16 + l15;
        fcx4 v = acc[t];
#pragma unroll
        for (int k = 0; k < NW - 1; k++) v += red[k][t][lane];
        fcx4 out;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            float u = v[r] * p.alpha + bias[r];
            if (p.act == FC_LRELU) u = (u > 0.f ? u : u * kFcSlope) * kFcGain;
            out[r] = u;
        }
        if (n >= p.n) continue;
        float* yp = p.y + (size_t)n * p.cout + ob;
        if (ob + 4 <= p.cout && (p.cout & 3) == 0) *(fcx4*)yp = out;
        else {
#pragma unroll
            for (int r = 0; r < 4; r++)
                if (ob + r < p.cout) yp[r] = out[r];
        }
    }
}

struct FcBwdArgs {
    float* dx; float* dw; float* db;
    const float* gy; const float* y; const float* x; const float* w;
    int n, cin, cout, act;
    float alpha, beta;
};

__device__ __forceinline__ float fc_gp(float gy, float y, int act) {
    return act == FC_LRELU ? gy * (y > 0.f ? kFcGain : kFcGain * kFcSlope) : gy;
}

// grid = cin / 16 workgroups (cin % 16 == 0, cout % 16 == 0)
template <int NT>
__global__ __launch_bounds__(256) void fc_act_bwd_kernel(FcBwdArgs p) {
    __shared__ fcx4 red[3][NT][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int k0 = blockIdx.x * 16;
    const int K = p.cin, O = p.cout;
    const int otiles = O >> 4;
    // x[:, k-slab] in the B map of the dw product: B[kk = n][j = k]: sample n = 16 t + 4 g + j, column k0 + l15
    float xb[NT][4];
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int n = 16 * t + 4 * g + j;
            xb[t][j] = n < p.n ? p.x[(size_t)n * K + k0 + l15] : 0.f;
        }
    fcx4 dxa[NT];
#pragma unroll
    for (int t = 0; t < NT; t++) dxa[t] = fcx4{0.f, 0.f, 0.f, 0.f};
    const bool want_db = p.db != nullptr && blockIdx.x == 0;
    // two row tiles per trip, every load of both issued before the first product (as one tile per trip a wave's 8-16 trips were a chain of
    // exposed memory latencies: 15-43 us for a layer)
    struct Tile { float wb[4]; fcx4 gyv[NT], yv[NT]; float gys[NT][4], ys[NT][4]; };
    auto load_tile = [&](int o0, Tile& L) {
#pragma unroll
        for (int j = 0; j < 4; j++) L.wb[j] = p.w[(size_t)(o0 + 4 * g + j) * K + k0 + l15];
#pragma unroll
        for (int t = 0; t < NT; t++) {
            const size_t off = (size_t)min(16 * t + l15, p.n - 1) * O + o0 + 4 * g;
            L.gyv[t] = *(const fcx4*)(p.gy + off);
            L.yv[t] = p.act == FC_LRELU ? *(const fcx4*)(p.y + off) : fcx4{1.f, 1.f, 1.f, 1.f};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const size_t off2 = (size_t)min(16 * t + 4 * g + j, p.n - 1) * O + o0 + l15;
                L.gys[t][j] = p.gy[off2];
                L.ys[t][j] = p.act == FC_LRELU ? p.y[off2] : 1.f;
            }
        }
    };
    auto do_tile = [&](int o0, const Tile& L) {
        // ---- dx: D[i = n][j = k] += A[n][o] B[o][k],  A = gp[n = 16 t + l15][o0 + 4 g + j] (16-byte loads), B = w[o0 + 4 g + j][k0 + l15]
        if (p.dx != nullptr) {
#pragma unroll
            for (int t = 0; t < NT; t++) {
                const int n = 16 * t + l15;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const float a = n < p.n ? fc_gp(L.gyv[t][j], L.yv[t][j], p.act) : 0.f;
                    dxa[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, L.wb[j], dxa[t], 0, 0, 0);
                }
            }
        }
        // ---- dw: D[i = o][j = k] = sum_n A[o][n] B[n][k],  A = gp[n = 16 t + 4 g + j][o0 + l15]
        fcx4 dwa = fcx4{0.f, 0.f, 0.f, 0.f};
        float colsum = 0.f;
#pragma unroll
        for (int t = 0; t < NT; t++)
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = 16 * t + 4 * g + j;
                const float a = n < p.n ? fc_gp(L.gys[t][j], L.ys[t][j], p.act) : 0.f;
                colsum += a;
                dwa = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[t][j], dwa, 0, 0, 0);
            }
        if (p.dw != nullptr) {
#pragma unroll
            for (int r = 0; r < 4; r++) p.dw[(size_t)(o0 + 4 * g + r) * K + k0 + l15] = dwa[r] * p.alpha;
        }
        if (want_db) {
            // colsum: this lane's four samples of column o0 + l15; the other samples sit in the lanes 16, 32, 48 further on
            colsum += __shfl_xor(colsum, 16);
            colsum += __shfl_xor(colsum, 32);
            if (g == 0) p.db[o0 + l15] = colsum * p.beta;
        }
    };
    for (int ot = wave; ot < otiles; ot += 8) {
        Tile L0, L1;
        const bool two = ot + 4 < otiles;                                     // (wave-uniform)
        load_tile(ot * 16, L0);
        if (two) load_tile((ot + 4) * 16, L1);
        do_tile(ot * 16, L0);
        if (two) do_tile((ot + 4) * 16, L1);
    }
    if (p.dx == nullptr) return;
    if (wave > 0) {
#pragma unroll
        for (int t = 0; t < NT; t++) red[wave - 1][t][lane] = dxa[t];
    }
    __syncthreads();
    if (wave != 0) return;
#pragma unroll
    for (int t = 0; t < NT; t++) {
        fcx4 v = dxa[t];
#pragma unroll
        for (int k = 0; k < 3; k++) v += red[k][t][lane];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int n = 16 * t + 4 * g + r;
            if (n < p.n) p.dx[(size_t)n * K + k0 + l15] = v[r] * p.alpha;
        }
    }
}

// ---- the mapping network's input stage (NET:143-150): x0 = cat(normalize(z), normalize(embed(c))) -----------------------------------------
// normalize(v) = v * rsqrt(mean(v^2) + 1e-8) over the feature dimension.  One workgroup per sample.
struct MapInArgs {
    float* x0; const float* z; const float* c; const float* ew; const float* eb;
    int n, zdim, cdim, wdim;
    float alpha, beta;
};

__device__ __forceinline__ float block_sum_256(float v, float* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    __syncthreads();                                            // (red may still be read from a previous call)
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float embed_row(const MapInArgs& p, int n, int o) {
    float e = 0.f;
    for (int j = 0; j < p.cdim; j++) e = fmaf(p.c[(size_t)n * p.cdim + j], p.ew[(size_t)o * p.cdim + j], e);
    return e * p.alpha + (p.eb != nullptr ? p.eb[o] * p.beta : 0.f);
}

__global__ __launch_bounds__(256) void mapping_input_fwd_kernel(MapInArgs p) {
    __shared__ float red[4];
    const int n = blockIdx.x;
    const int stride = p.zdim + (p.cdim > 0 ? p.wdim : 0);
    float s = 0.f;
    for (int k = threadIdx.x; k < p.zdim; k += 256) { const float v = p.z[(size_t)n * p.zdim + k]; s = fmaf(v, v, s); }
    const float rz = rsqrtf(block_sum_256(s, red) / (float)p.zdim + 1e-8f);
    for (int k = threadIdx.x; k < p.zdim; k += 256) p.x0[(size_t)n * stride + k] = p.z[(size_t)n * p.zdim + k] * rz;
    if (p.cdim <= 0) return;
    s = 0.f;
    for (int o = threadIdx.x; o < p.wdim; o += 256) { const float e = embed_row(p, n, o); s = fmaf(e, e, s); }
    const float re = rsqrtf(block_sum_256(s, red) / (float)p.wdim + 1e-8f);
    for (int o = threadIdx.x; o < p.wdim; o += 256) p.x0[(size_t)n * stride + p.zdim + o] = embed_row(p, n, o) * re;
}

// gradient of the embedding layer's weight / bias from g = dL/dx0[:, zdim:]:  with e the embedding, r = rsqrt(mean e^2 + eps), en = e r:
// ge = r (g - en mean(g en));  dew[o][j] = alpha sum_n ge[n][o] c[n][j];  deb[o] = beta sum_n ge[n][o].   (z and c carry no gradient.)
// One workgroup per SAMPLE writes ge[n][:] to a scratch row (phase 1); the last workgroup to finish (a ticket in global memory) sums the
// rows into dew / deb in a fixed order (phase 2).  r06, first form: ONE workgroup walking the samples -- sixteen serial rounds of
// (load, two block reductions, load again) = 42 us for a few KB.
struct MapInBwd { const float* gx0; float* dew; float* deb; float* scratch; unsigned* ticket; };
__global__ __launch_bounds__(256) void mapping_input_bwd_kernel(MapInArgs p, MapInBwd q) {
    __shared__ float red[4];
    __shared__ bool last;
    const int stride = p.zdim + p.wdim;
    const int n = blockIdx.x;
    float s = 0.f, d = 0.f;
    for (int o = threadIdx.x; o < p.wdim; o += 256) {
        const float e = embed_row(p, n, o);
        s = fmaf(e, e, s);
        d = fmaf(q.gx0[(size_t)n * stride + p.zdim + o], e, d);
    }
    const float ms = block_sum_256(s, red) / (float)p.wdim;
    const float r = rsqrtf(ms + 1e-8f);
    const float m = block_sum_256(d, red) * r / (float)p.wdim;              // mean(g en)
    for (int o = threadIdx.x; o < p.wdim; o += 256) {
        const float e = embed_row(p, n, o);
        __hip_atomic_store(q.scratch + (size_t)n * p.wdim + o, r * (q.gx0[(size_t)n * stride + p.zdim + o] - e * r * m), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // hand-off (MI355X_MICROARCH.md, Valid forms): sc1 stores drained by every wave, the workgroup's barrier, one agent-scope ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) last = __hip_atomic_fetch_add(q.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1;
    __syncthreads();
    if (!last) return;
    if (threadIdx.x == 0) __hip_atomic_store(q.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // ready for the next launch
    for (int o = threadIdx.x; o < p.wdim; o += 256) {
        float db = 0.f;
        for (int j = 0; j < p.cdim; j++) {
            float acc = 0.f;
            for (int k = 0; k < p.n; k++)
                acc = fmaf(__hip_atomic_load(q.scratch + (size_t)k * p.wdim + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), p.c[(size_t)k * p.cdim + j], acc);
            q.dew[(size_t)o * p.cdim + j] = p.alpha * acc;
        }
        if (q.deb != nullptr) {
            for (int k = 0; k < p.n; k++) db += __hip_atomic_load(q.scratch + (size_t)k * p.wdim + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            q.deb[o] = p.beta * db;
        }
    }
}

}  // namespace afcm

using namespace afcm;

extern "C" int afcm_fc_act_fwd(float* y, const float* x, const float* w, const float* b, int32_t n, int32_t cin, int32_t cout, float alpha,
                               float beta, int32_t act, void* stream) {
    AFCM_REQUIRE(y != nullptr && x != nullptr && w != nullptr, "fc_act_fwd: y, x and w must be non-null");
    AFCM_REQUIRE(n > 0 && cin > 0 && cout > 0, "fc_act_fwd: empty problem");
    AFCM_REQUIRE(act == FC_LINEAR || act == FC_LRELU, "fc_act_fwd: activation must be 0 (linear) or 1 (lrelu)");
    if (n > 64 || (cin & 15) != 0) return AFCM_E_NOKERNEL;
    AFCM_REQUIRE((((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) == 0, "fc_act_fwd: pointers must be 16-byte aligned");
    FcArgs p{y, x, w, b, n, cin, cout, act, alpha, beta};
    const dim3 grid((unsigned)cdiv(cout, 16));
    hipStream_t st = (hipStream_t)stream;
    const bool wide = cin >= 2048 && n <= 32;                   // long rows: 16 waves share a row block's K range (64 rows: over the 128-register cap)
#define AFCM_FC(NT) do { if (wide) hipLaunchKernelGGL((fc_act_fwd_kernel<NT, 16>), grid, dim3(1024), 0, st, p); \
                         else hipLaunchKernelGGL((fc_act_fwd_kernel<NT, 4>), grid, dim3(256), 0, st, p); } while (0)
    if (n <= 16) AFCM_FC(1);
    else if (n <= 32) AFCM_FC(2);
    else hipLaunchKernelGGL((fc_act_fwd_kernel<4, 4>), grid, dim3(256), 0, st, p);
#undef AFCM_FC
    return hip_status(hipGetLastError());
}

extern "C" int afcm_fc_act_bwd(float* dx, float* dw, float* db, const float* gy, const float* y, const float* x, const float* w, int32_t n,
                               int32_t cin, int32_t cout, float alpha, float beta, int32_t act, void* stream) {
    AFCM_REQUIRE(gy != nullptr && x != nullptr && w != nullptr, "fc_act_bwd: gy, x and w must be non-null");
    AFCM_REQUIRE(act == FC_LINEAR || (act == FC_LRELU && y != nullptr), "fc_act_bwd: activation must be 0 (linear) or 1 (lrelu, with the saved output)");
    AFCM_REQUIRE(n > 0 && cin > 0 && cout > 0, "fc_act_bwd: empty problem");
    if (n > 64 || (cin & 15) != 0 || (cout & 15) != 0) return AFCM_E_NOKERNEL;
    AFCM_REQUIRE((((uintptr_t)gy | (uintptr_t)(y ? y : gy)) & 15) == 0, "fc_act_bwd: gy and y must be 16-byte aligned");
    FcBwdArgs p{dx, dw, db, gy, y, x, w, n, cin, cout, act, alpha, beta};
    const dim3 grid((unsigned)(cin / 16)), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (n <= 16) hipLaunchKernelGGL((fc_act_bwd_kernel<1>), grid, block, 0, st, p);
    else if (n <= 32) hipLaunchKernelGGL((fc_act_bwd_kernel<2>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((fc_act_bwd_kernel<4>), grid, block, 0, st, p);
    return hip_status(hipGetLastError());
}

extern "C" int afcm_mapping_input_fwd(float* x0, const float* z, const float* c, const float* ew, const float* eb, int32_t n, int32_t zdim,
                                      int32_t cdim, int32_t wdim, float alpha, float beta, void* stream) {
    AFCM_REQUIRE(x0 != nullptr && z != nullptr && n > 0 && zdim > 0, "mapping_input_fwd: x0 and z must be non-null and non-empty");
    AFCM_REQUIRE(cdim == 0 || (c != nullptr && ew != nullptr && wdim > 0), "mapping_input_fwd: a conditioning label needs c and the embedding weight");
    MapInArgs p{x0, z, c, ew, eb, n, zdim, cdim, wdim, alpha, beta};
    hipLaunchKernelGGL(mapping_input_fwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, p);
    return hip_status(hipGetLastError());
}

extern "C" int64_t afcm_mapping_input_bwd_workspace_bytes(int32_t n, int32_t wdim) { return ((int64_t)n * wdim + 64) * 4; }

extern "C" int afcm_mapping_input_bwd(float* dew, float* deb, const float* gx0, const float* c, const float* ew, const float* eb, int32_t n,
                                      int32_t zdim, int32_t cdim, int32_t wdim, float alpha, float beta, void* workspace, void* stream) {
    AFCM_REQUIRE(dew != nullptr && gx0 != nullptr && c != nullptr && ew != nullptr, "mapping_input_bwd: dew, gx0, c and ew must be non-null");
    AFCM_REQUIRE(n > 0 && zdim > 0 && cdim > 0 && wdim > 0, "mapping_input_bwd: empty problem");
    AFCM_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 3) == 0, "mapping_input_bwd: workspace of afcm_mapping_input_bwd_workspace_bytes(), its first word zero");
    MapInArgs p{nullptr, nullptr, c, ew, eb, n, zdim, cdim, wdim, alpha, beta};
    // workspace: [ticket (one word, zero between launches: the kernel resets it), 63 words of padding, n x wdim floats]
    MapInBwd q{gx0, dew, deb, (float*)workspace + 64, (unsigned*)workspace};
    hipLaunchKernelGGL(mapping_input_bwd_kernel, dim3((unsigned)n), dim3(256), 0, (hipStream_t)stream, p, q);
    return hip_status(hipGetLastError());
}
