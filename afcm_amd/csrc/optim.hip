// Generator update of the `--model stylegan3` step in ONE launch: gradient averaging factor, the reference's NaN/Inf scrub
// (models/stylegan3_model.py:122-124,132-134: nan -> 0, +inf -> 1e5, -inf -> -1e5) and the Adam update
// (models/comodgan_model.py:19-20: torch.optim.Adam, betas (0, 0.99), eps 1e-8) over every parameter tensor.
// The eager sequence is ~110 nan_to_num launches + 7 multi-tensor passes over 234 MB each; this kernel reads p, g, m, v and
// writes p, m, v once: HBM-bound, 7 x 4 B per parameter.
// Arithmetic mirrors torch's foreach Adam step by step (lerp for the first moment, mul + addcmul for the second,
// sqrt / bias_correction2_sqrt + eps, addcdiv) so results agree to the last bit or two.
#include "common.h"

namespace afcm {

constexpr int kAdamChunk = 16384;      // elements per workgroup

__device__ __forceinline__ float scrub(float g, float posinf, float neginf) {
    if (g != g) return 0.f;
    if (g == __builtin_inff()) return posinf;
    if (g == -__builtin_inff()) return neginf;
    return g;
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const afcm_adam_entry* __restrict__ table, int n, float step_size, float beta1,
                                                         float beta2, float w1, float w2, float bc2_sqrt, float eps, float grad_scale, int do_scrub,
                                                         float posinf, float neginf, int write_grad) {
    // which tensor does this chunk belong to: binary search over the chunk prefix (wave-uniform)
    const long long chunk = blockIdx.x;
    int lo = 0, hi = n - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].chunk0 <= chunk) lo = mid; else hi = mid - 1;
    }
    const afcm_adam_entry e = table[lo];
    const long long base = (chunk - e.chunk0) * kAdamChunk;
    const long long end = min(base + (long long)kAdamChunk, (long long)e.numel);
    float* __restrict__ p = (float*)e.p;
    float* __restrict__ g = (float*)e.g;
    float* __restrict__ m = (float*)e.m;
    float* __restrict__ v = (float*)e.v;
    auto update = [&](float& pp, float& gg, float& mm, float& vv) __attribute__((always_inline)) {
        float gr = gg * grad_scale;
        if (do_scrub) gr = scrub(gr, posinf, neginf);
        gg = gr;
        // torch.lerp(m, g, 1 - beta1)
        mm = (w1 < 0.5f) ? mm + w1 * (gr - mm) : gr - (gr - mm) * (1.f - w1);
        vv = vv * beta2 + (w2 * gr) * gr;
        const float denom = sqrtf(vv) / bc2_sqrt + eps;
        pp = pp + (-step_size) * (mm / denom);
    };
    const bool vec = (((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0;
    if (vec) {
        for (long long i = base + 4 * threadIdx.x; i < end; i += 4 * 256) {
            if (i + 4 <= end) {
                float4 pp = *(float4*)(p + i), gg = *(const float4*)(g + i), mm = *(float4*)(m + i), vv = *(float4*)(v + i);
                update(pp.x, gg.x, mm.x, vv.x); update(pp.y, gg.y, mm.y, vv.y);
                update(pp.z, gg.z, mm.z, vv.z); update(pp.w, gg.w, mm.w, vv.w);
                *(float4*)(p + i) = pp; *(float4*)(m + i) = mm; *(float4*)(v + i) = vv;
                if (write_grad) *(float4*)(g + i) = gg;
            } else {
                for (long long j = i; j < end; j++) {
                    float pp = p[j], gg = g[j], mm = m[j], vv = v[j];
                    update(pp, gg, mm, vv);
                    p[j] = pp; m[j] = mm; v[j] = vv;
                    if (write_grad) g[j] = gg;
                }
            }
        }
    } else {
        for (long long j = base + threadIdx.x; j < end; j += 256) {
            float pp = p[j], gg = g[j], mm = m[j], vv = v[j];
            update(pp, gg, mm, vv);
            p[j] = pp; m[j] = mm; v[j] = vv;
            if (write_grad) g[j] = gg;
        }
    }
}

}  // namespace afcm

extern "C" int32_t afcm_adam_chunk_elems(void) { return afcm::kAdamChunk; }

extern "C" int afcm_adam_multi(const afcm_adam_entry* table_dev, int32_t n, int64_t total_chunks, float step_size, float beta1, float beta2,
                               float one_minus_beta1, float one_minus_beta2, float bias_correction2_sqrt, float eps, float grad_scale, int32_t scrub, float posinf, float neginf,
                               int32_t write_grad, void* stream) {
    using namespace afcm;
    AFCM_REQUIRE(table_dev != nullptr && n > 0, "adam_multi: empty table");
    AFCM_REQUIRE(total_chunks > 0 && total_chunks < (1ll << 31), "adam_multi: %lld chunks is out of range", (long long)total_chunks);
    AFCM_REQUIRE(bias_correction2_sqrt > 0.f, "adam_multi: bias_correction2_sqrt must be positive");
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, n, step_size, beta1,
                       beta2, one_minus_beta1, one_minus_beta2, bias_correction2_sqrt, eps, grad_scale, scrub, posinf, neginf, write_grad);
    return hip_status(hipGetLastError());
}
