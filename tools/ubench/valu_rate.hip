// VALU issue-rate microbenchmark for gfx950: fp32 FMA vs packed fp32 FMA vs fp16 dot2 (fp32 accumulate) vs packed fp16 FMA.
// Build: hipcc -O3 --offload-arch=gfx950 valu_rate.hip -o valu_rate ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float seed) {
    float a[8]; f2 p[8]; h2 hh[8];
    for (int i = 0; i < 8; i++) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1}; hh[i] = h2{(_Float16)a[i], (_Float16)(a[i] + 1)}; }
    float c = seed * 0.5f; f2 pc = {c, c}; h2 hc = {(_Float16)c, (_Float16)c};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (MODE == 0) a[i] = __builtin_fmaf(a[i], c, c);
                if (MODE == 1) p[i] = __builtin_elementwise_fma(p[i], pc, pc);
                if (MODE == 2) a[i] = __builtin_amdgcn_fdot2(hh[i], hc, a[i], false);
                if (MODE == 3) hh[i] = __builtin_elementwise_fma(hh[i], hc, hc);
                if (MODE == 4) a[i] = __builtin_amdgcn_fdot2_f32_bf16(*(s2*)&hh[i], *(s2*)&hc, a[i], false);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y + (float)hh[i].x + (float)hh[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE> void run(const char* name, double flop_per_inst_lane) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256 * 8, 256>>>(out, 10, 1.0f);
    hipEventRecord(e0);
    k<MODE><<<256 * 8, 256>>>(out, iters, 1.0f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double insts = (double)256 * 8 * 256 * iters * 32;   // per-lane instructions
    printf("%-22s %8.3f ms  %7.2f T lane-inst/s  -> %7.1f TFLOP/s\n", name, ms, insts / ms / 1e9, insts * flop_per_inst_lane / ms / 1e9);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32", 2);
    run<1>("v_pk_fma_f32", 4);
    run<2>("v_dot2_f32_f16", 4);
    run<3>("v_pk_fma_f16", 4);
    run<4>("v_dot2_f32_bf16", 4);
    return 0;
}
