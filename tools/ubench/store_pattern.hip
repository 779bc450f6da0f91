// Store-pattern microbenchmark for the filtered_lrelu output path (gfx950): bf16 planes [PL][H][W], one wave per 32-row strip.
//   rows:   per 64-column group 4 stores of 8 rows x 128 B (the wave kernels' flush: rows of W * 2 bytes, groups not line-aligned)
//   linear: the strip's contiguous bytes in 1 KB pieces (what a fully staged strip could do)
// Build: hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
template <int MODE, int AUX>
__global__ __launch_bounds__(256) void k(unsigned short* y, int PL, int H, int W, int strips) {
    const int lane = threadIdx.x & 63, wt = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (wt >= PL * strips) return;
    const int plane = wt / strips, ty = wt - plane * strips;
    unsigned short* yp = y + (size_t)plane * H * W;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)yp, 0, H * W * 2, 0x00020000);
    const u32x4 v = {(unsigned)wt, (unsigned)lane, 1u, 2u};
    const int RP = W * 2;
    if (MODE == 0) {
        const int groups = (W + 63) / 64;
        for (int q = 0; q < groups; q++)
            for (int j = 0; j < 4; j++) {
                const int row = ty * 32 + 8 * j + (lane >> 3), col = 64 * q + 8 * (lane & 7);
                const unsigned off = (col + 8 <= W) ? (unsigned)(row * RP + col * 2) : 0x80000000u;
                __builtin_amdgcn_raw_buffer_store_b128(v, rs, off, 0, AUX);
            }
    } else {
        const int bytes = 32 * RP, base = ty * bytes;
        for (int o = lane * 16; o < bytes; o += 1024) __builtin_amdgcn_raw_buffer_store_b128(v, rs, (unsigned)(base + o), 0, AUX);
    }
}
template <int MODE, int AUX> void run(const char* name, int PL, int H, int W) {
    unsigned short* y; hipMalloc(&y, (size_t)PL * H * W * 2);
    const int strips = (H + 31) / 32, waves = PL * strips;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) k<MODE, AUX><<<(waves + 3) / 4, 256>>>(y, PL, H, W, strips);
    hipEventRecord(e0);
    for (int i = 0; i < 20; i++) k<MODE, AUX><<<(waves + 3) / 4, 256>>>(y, PL, H, W, strips);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 20;
    printf("%-28s %4dx%4d x %5d planes: %7.3f ms  %7.1f GB/s\n", name, H, W, PL, ms, (double)PL * H * W * 2 / ms / 1e6);
    hipFree(y);
}
int main() {
    for (int W : {276, 256, 148}) {
        const int PL = W == 148 ? 16 * 362 : 16 * 128;
        run<0, 0>("rows", PL, W, W); run<0, 2>("rows nt", PL, W, W);
        run<1, 0>("linear", PL, W, W); run<1, 2>("linear nt", PL, W, W);
    }
    return 0;
}
