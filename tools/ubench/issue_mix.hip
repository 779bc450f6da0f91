// What one SIMD of gfx950 issues per cycle when W waves each run the same stream of MF x v_mfma_f32_16x16x32_bf16 + NV vector instructions of one
// kind (independent of each other and of the MFMAs): cycles per block per WAVE and per SIMD (= per wave / W) for W = 1, 2, 3 waves per SIMD.
// The filtered_lrelu wave kernels retire ~5 vector instructions per MFMA at 3 waves per SIMD; this says what that mix can cost at best.
// Build: hipcc -O3 --offload-arch=gfx950 issue_mix.hip -o issue_mix.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

// one block = ONE asm statement (the compiler pads separate asm statements with s_nop): operands %0 acc, %1..%8 results, %9 %10 %11 sources, %12 %13 MFMA operands
#define I0(d) "v_cvt_pk_bf16_f32 %" #d ", %9, %10\n"
#define I1(d) "v_pk_max_i16 %" #d ", %9, %10\n"
#define I2(d) "v_perm_b32 %" #d ", %9, %10, %11\n"
#define I3(d) "v_maximum3_f32 %" #d ", %9, |%10|, |%11|\n"
#define I4(d) "v_mov_b32 %" #d ", %9\n"
#define I5(d) "v_fma_f32 %" #d ", %9, %10, %11\n"
#define I6(d) "v_and_or_b32 %" #d ", %9, %10, %11\n"
#define I7(d) "v_pk_lshrrev_b16 %" #d ", 15, %9 op_sel_hi:[0,1]\n"
#define I8(d) "v_lshl_or_b32 %" #d ", %9, 4, %10\n"
#define I9(d) "v_and_b32 %" #d ", %9, %10\n"
#define I10(d) "v_or_b32 %" #d ", %9, %10\n"
#define I11(d) "v_max_f32 %" #d ", %9, %10\n"
#define I12(d) "v_add_f32 %" #d ", %9, %10\n"
#define I13(d) "v_mul_f32 %" #d ", %9, %10\n"
#define I14(d) "v_add_u32 %" #d ", %9, %10\n"
#define I15(d) "v_lshrrev_b32 %" #d ", 6, %9\n"
#define I16(d) "v_max_i32 %" #d ", %9, %10\n"
#define I17(d) "v_bfi_b32 %" #d ", %9, %10, %11\n"
#define I18(d) "v_max3_f32 %" #d ", %9, %10, %11\n"
#define I19(d) "v_alignbyte_b32 %" #d ", %9, %10, 1\n"
#define I20(d) "v_bfe_u32 %" #d ", %9, 8, 8\n"
#define I21(d) "v_cvt_pkrtz_f16_f32 %" #d ", %9, %10\n"
#define I22(d) "v_and_b32 %" #d ", 0x7fff7fff, %9\n"
#define I23(d) "v_max_i16 %" #d ", %9, %10\n"
#define I24(d) "v_pk_add_u16 %" #d ", %9, %10\n"
#define MFMA "v_mfma_f32_16x16x32_bf16 %0, %12, %13, %0\n"
#define OPS : "+v"(acc[u]), "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]), "=&v"(d[4]), "=&v"(d[5]), "=&v"(d[6]), "=&v"(d[7]) : "v"(x), "v"(y), "v"(z), "v"(a), "v"(b)
#define BLOCK(I)                                                                                           \
    do {                                                                                                   \
        if (MF && NV == 0) asm volatile(MFMA OPS);                                                         \
        if (MF && NV == 2) asm volatile(MFMA I(1) I(2) OPS);                                               \
        if (MF && NV == 4) asm volatile(MFMA I(1) I(2) I(3) I(4) OPS);                                     \
        if (MF && NV == 6) asm volatile(MFMA I(1) I(2) I(3) I(4) I(5) I(6) OPS);                           \
        if (MF && NV == 8) asm volatile(MFMA I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) OPS);                 \
        if (!MF && NV == 8) asm volatile(I(1) I(2) I(3) I(4) I(5) I(6) I(7) I(8) OPS);                     \
    } while (0)

template <int KIND, int NV, int MF>
__global__ __launch_bounds__(256) void k(unsigned long long* out, float* sink, int iters, float seed) {
    extern __shared__ unsigned char pad[];
    f32x4 acc[4];
    bf16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (__bf16)(seed + i + threadIdx.x * 0.01f); b[i] = (__bf16)(seed * 0.5f + i); }
    for (int i = 0; i < 4; i++) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float x = seed + threadIdx.x, y = seed * 3.f, z = seed - threadIdx.x;
    float d[8];
    for (int i = 0; i < 8; i++) d[i] = 0.f;
    if (iters < 0) pad[threadIdx.x] = 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            if (KIND == 0) BLOCK(I0);
            if (KIND == 1) BLOCK(I1);
            if (KIND == 2) BLOCK(I2);
            if (KIND == 3) BLOCK(I3);
            if (KIND == 4) BLOCK(I4);
            if (KIND == 5) BLOCK(I5);
            if (KIND == 6) BLOCK(I6);
            if (KIND == 7) BLOCK(I7);
            if (KIND == 8) BLOCK(I8);
            if (KIND == 9) BLOCK(I9);
            if (KIND == 10) BLOCK(I10);
            if (KIND == 11) BLOCK(I11);
            if (KIND == 12) BLOCK(I12);
            if (KIND == 13) BLOCK(I13);
            if (KIND == 14) BLOCK(I14);
            if (KIND == 15) BLOCK(I15);
            if (KIND == 16) BLOCK(I16);
            if (KIND == 17) BLOCK(I17);
            if (KIND == 18) BLOCK(I18);
            if (KIND == 19) BLOCK(I19);
            if (KIND == 20) BLOCK(I20);
            if (KIND == 21) BLOCK(I21);
            if (KIND == 22) BLOCK(I22);
            if (KIND == 23) BLOCK(I23);
            if (KIND == 24) BLOCK(I24);
        }
    }
    asm volatile("s_nop 15\n s_nop 15");
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int i = 0; i < 8; i++) s += d[i];
    for (int u = 0; u < 4; u++) s += acc[u][0] + acc[u][1] + acc[u][2] + acc[u][3];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int KIND, int NV, int MF>
void run(const char* name) {
    const int iters = 20000;
    printf("%-20s MF=%d NV=%d :", name, MF, NV);
    for (int W = 1; W <= 2; W++) {
        const int nwg = 256 * W;
        unsigned long long* out;
        float* sink;
        hipMalloc(&out, nwg * 4 * 8);
        hipMalloc(&sink, nwg * 256 * 4);
        const int lds = W == 1 ? 100 * 1024 : W == 2 ? 70 * 1024 : 48 * 1024;   // W workgroups fit a CU's 160 KB, W + 1 do not
        hipFuncSetAttribute((const void*)k<KIND, NV, MF>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        hipLaunchKernelGGL((k<KIND, NV, MF>), dim3(nwg), dim3(256), lds, 0, out, sink, 100, 1.0f);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, NV, MF>), dim3(nwg), dim3(256), lds, 0, out, sink, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(nwg * 4);
        hipMemcpy(h.data(), out, nwg * 4 * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per_wave = (double)h[h.size() / 2] / (iters * 4.0);   // shader cycles per block (MF MFMA + NV vector instructions), one wave
        printf("  W=%d: %6.1f cyc/block/wave = %5.1f /SIMD (%.2f ms, %.2f GHz)", W, per_wave, per_wave / W, ms, (double)h[h.size() / 2] / (ms * 1e6));
        hipFree(out);
        hipFree(sink);
    }
    printf("\n");
}

int main() {
    // vector instructions alone: 8 per block
    run<0, 8, 0>("v_cvt_pk_bf16_f32");
    run<1, 8, 0>("v_pk_max_i16");
    run<2, 8, 0>("v_perm_b32");
    run<3, 8, 0>("v_maximum3_f32");
    run<4, 8, 0>("v_mov_b32");
    run<5, 8, 0>("v_fma_f32");
    run<6, 8, 0>("v_and_or_b32");
    run<7, 8, 0>("v_pk_lshrrev_b16");
    run<8, 8, 0>("v_lshl_or_b32");
    run<9, 8, 0>("v_and_b32");
    run<10, 8, 0>("v_or_b32");
    run<11, 8, 0>("v_max_f32");
    run<12, 8, 0>("v_add_f32");
    run<13, 8, 0>("v_mul_f32");
    run<14, 8, 0>("v_add_u32");
    run<15, 8, 0>("v_lshrrev_b32");
    run<16, 8, 0>("v_max_i32");
    run<17, 8, 0>("v_bfi_b32");
    run<18, 8, 0>("v_max3_f32");
    run<19, 8, 0>("v_alignbyte_b32");
    run<20, 8, 0>("v_bfe_u32");
    run<21, 8, 0>("v_cvt_pkrtz_f16_f32");
    run<22, 8, 0>("v_and_b32 literal");
    run<23, 8, 0>("v_max_i16");
    run<24, 8, 0>("v_pk_add_u16");
    // the matrix instruction alone
    run<0, 0, 1>("mfma only");
    // one MFMA + NV vector instructions
    run<0, 2, 1>("mfma+cvt_pk");
    run<0, 4, 1>("mfma+cvt_pk");
    run<0, 6, 1>("mfma+cvt_pk");
    run<1, 4, 1>("mfma+pk_max_i16");
    run<2, 4, 1>("mfma+perm");
    run<3, 4, 1>("mfma+maximum3");
    run<4, 4, 1>("mfma+mov");
    run<4, 6, 1>("mfma+mov");
    run<4, 8, 1>("mfma+mov");
    run<0, 8, 1>("mfma+cvt_pk");
    run<5, 4, 1>("mfma+fma");
    run<5, 6, 1>("mfma+fma");
    return 0;
}
