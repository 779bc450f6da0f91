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
                                                         float posinf, float neginf, int write_grad, const float* __restrict__ step_dev, float lr) {
    // capturable form (step_dev != NULL): the step count lives on the device (adam_step_inc_kernel advances it), the bias corrections are
    // formed here -- a launch captured into a hipGraph then replays with the right corrections at every step
    if (step_dev != nullptr) {
        const double t = (double)step_dev[0];
        const double bc1 = 1.0 - pow((double)beta1, t), bc2 = 1.0 - pow((double)beta2, t);
        step_size = (float)((double)lr / bc1);
        bc2_sqrt = (float)sqrt(bc2);
    }
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
                // (beta1 = 0, the reference's setting: the first moment is the scrubbed gradient itself -- lerp(m, g, 1) = g for every finite m,
                // and m is finite: it is a scrubbed gradient -- so its old value is not read: 234 of the step's 1,638 MB)
                float4 pp = *(float4*)(p + i), gg = *(const float4*)(g + i), vv = *(float4*)(v + i);
                float4 mm = w1 == 1.f ? make_float4(0.f, 0.f, 0.f, 0.f) : *(float4*)(m + i);
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
                       beta2, one_minus_beta1, one_minus_beta2, bias_correction2_sqrt, eps, grad_scale, scrub, posinf, neginf, write_grad,
                       (const float*)nullptr, 0.f);
    return hip_status(hipGetLastError());
}

namespace afcm { __global__ void adam_step_inc_kernel(float* step) { step[0] += 1.f; } }

extern "C" int afcm_adam_multi_capturable(const afcm_adam_entry* table_dev, int32_t n, int64_t total_chunks, float* step_dev, float lr, float beta1,
                                          float beta2, float eps, float grad_scale, int32_t scrub, float posinf, float neginf, int32_t write_grad,
                                          void* stream) {
    using namespace afcm;
    AFCM_REQUIRE(table_dev != nullptr && n > 0 && step_dev != nullptr, "adam_multi_capturable: empty table or no step counter");
    AFCM_REQUIRE(total_chunks > 0 && total_chunks < (1ll << 31), "adam_multi: %lld chunks is out of range", (long long)total_chunks);
    hipLaunchKernelGGL(adam_step_inc_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev);
    hipLaunchKernelGGL(adam_multi_kernel, dim3((unsigned)total_chunks), dim3(256), 0, (hipStream_t)stream, table_dev, n, 0.f, beta1,
                       beta2, 1.f - beta1, 1.f - beta2, 1.f, eps, grad_scale, scrub, posinf, neginf, write_grad, (const float*)step_dev, lr);
    return hip_status(hipGetLastError());
}


// ---- the generator's L1 term (models/stylegan3_model.py:107: criterionL1(fake_B, real_B) * lambda_L1) ------------------------------------
// torch.nn.L1Loss + the weight is sub, abs, mean, mul forward and four elementwise launches backward over a 4 MB image; here: per-workgroup
// partial sums of weight / numel * |a - b| (the caller adds the <= 256 partials: a fixed order, so the value is reproducible) and
// ga = gout * weight / numel * sign(a - b).
namespace afcm {
__global__ __launch_bounds__(256) void l1_partials_kernel(float* __restrict__ partials, const float* __restrict__ a, const float* __restrict__ b,
                                                          long long numel, float scale) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long long)gridDim.x * 256) s += __builtin_fabsf(a[i] - b[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (red[0] + red[1] + red[2] + red[3]) * scale;
}
__global__ __launch_bounds__(256) void l1_grad_kernel(float* __restrict__ ga, const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ gout, long long numel, float scale) {
    const float gs = gout[0] * scale;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long long)gridDim.x * 256) {
        const float d = a[i] - b[i];
        ga[i] = d > 0.f ? gs : (d < 0.f ? -gs : (d == 0.f ? 0.f : d * gs));     // (a NaN difference stays a NaN, as sign() * g does)
    }
}
}  // namespace afcm

extern "C" int afcm_l1_partials(float* partials, const float* a, const float* b, int64_t numel, int32_t blocks, float weight, void* stream) {
    AFCM_REQUIRE(partials != nullptr && a != nullptr && b != nullptr && numel > 0 && blocks > 0 && blocks <= 1024, "l1_partials: empty input or bad block count");
    hipLaunchKernelGGL(afcm::l1_partials_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, partials, a, b, (long long)numel, weight / (float)numel);
    return afcm::hip_status(hipGetLastError());
}

extern "C" int afcm_l1_grad(float* ga, const float* a, const float* b, const float* gout, int64_t numel, float weight, void* stream) {
    AFCM_REQUIRE(ga != nullptr && a != nullptr && b != nullptr && gout != nullptr && numel > 0, "l1_grad: empty input");
    long long blocks = (numel + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(afcm::l1_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, ga, a, b, gout, (long long)numel, weight / (float)numel);
    return afcm::hip_status(hipGetLastError());
}
