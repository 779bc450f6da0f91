// Probe: does ds_read_b64_tr_b16 accept a 2-byte-aligned (odd element) column offset?
// LDS image: 8 rows x 64 columns of 16-bit values v = 100*row + col.  A 16-lane group reads the 4 x 16 block starting at
// column `off`; lane i must receive column off + i of rows 0..3.
// build: hipcc --offload-arch=gfx950 -O2 -o tr_align_probe tr_align_probe.hip ; run: ./tr_align_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void probe(unsigned short* out, int off) {
    __shared__ __attribute__((aligned(16))) unsigned short l[8 * 64];
    for (int i = threadIdx.x; i < 8 * 64; i += 64) l[i] = (unsigned short)(100 * (i / 64) + (i % 64));
    __syncthreads();
    const int lane = threadIdx.x, i16 = lane & 15, q = i16 >> 2, p4 = i16 & 3;
    const unsigned short* a = l + q * 64 + 4 * p4 + off;          // row q, columns off + 4p .. off + 4p + 3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
    for (int e = 0; e < 4; e++) out[lane * 4 + e] = (unsigned short)v[e];
}
int main() {
    unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
    int bad_total = 0;
    for (int off = 0; off < 4; off++) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, off);
        unsigned short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        int bad = 0;
        for (int lane = 0; lane < 16; lane++) for (int e = 0; e < 4; e++) if (h[lane * 4 + e] != 100 * e + off + lane) bad++;
        printf("off %d: %s (lane0: %d %d %d %d, lane1: %d %d %d %d)\n", off, bad ? "MISMATCH" : "ok", h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7]);
        bad_total += bad;
    }
    printf(bad_total ? "UNALIGNED TR READS NOT SUPPORTED (or different semantics)\n" : "UNALIGNED TR READS OK\n");
    return 0;
}
