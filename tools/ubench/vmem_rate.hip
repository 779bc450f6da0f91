// Probe (gfx950): (1) what a vector-memory instruction costs the CU's address path when every access hits the L1 -- by width
// (4 / 8 / 12 / 16 bytes per lane), with a partial exec mask, with out-of-range offsets, with LDS-DMA; (2) do the dwords of a wide
// raw buffer load that STARTS below offset 0 (vector offset -4 / -8 as an unsigned number) wrap back into range one by one?
// Motivation: conv2d_fwd16d_kernel (no-LDS B operands from 4-byte loads) ran at exactly the rate "16 cycles per wave instruction
// whatever its width" predicts.
// Build: hipcc -O3 --offload-arch=gfx950 vmem_rate.hip -o vmem_rate.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(3))) unsigned u32x3;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;

// MODE 0: b32, 1: b64, 2: b96, 3: b128 (lanes contiguous); 4: b32 with 4 of 64 lanes active; 5: b32, every lane out of range;
// 6: b32 gather (lane -> 4 "channels" 1 KB apart x 16 dwords); 7: b128 with 16 of 64 lanes active
template <int MODE>
__global__ __launch_bounds__(256) void rate(const unsigned* src, unsigned* out, int iters) {
    const int lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 8192, 0x00020000);   // 8 KB, every wave: L1-resident
    unsigned acc = 0;
    constexpr int W = MODE == 1 ? 8 : MODE == 2 ? 12 : (MODE == 3 || MODE == 7) ? 16 : 4;
    unsigned off = MODE == 6 ? (unsigned)((lane & 3) * 1024 + (lane >> 2) * 4) : (unsigned)(lane * W);
    // 8..: b32 variants.  8: contiguous, start 4 bytes past a 256-byte boundary; 9: two 32-lane halves in different rows (aligned);
    // 10: four 16-lane groups in different rows (64-byte aligned); 11: as 10, every group 4 bytes past its 64-byte boundary;
    // 12: as 9, halves 4 bytes past; 13: eight 8-lane groups in different rows (32-byte aligned); 14: lanes 8 bytes apart (every
    // other dword); 15: as 10 with the rows 1028 bytes apart (groups start at 0, 4, 8, 12 bytes past a 64-byte boundary)
    if (MODE == 8) off = lane * 4 + 4;
    if (MODE == 9) off = (lane >> 5) * 1024 + (lane & 31) * 4;
    if (MODE == 10) off = (lane >> 4) * 1024 + (lane & 15) * 4;
    if (MODE == 11) off = (lane >> 4) * 1024 + (lane & 15) * 4 + 4;
    if (MODE == 12) off = (lane >> 5) * 1024 + (lane & 31) * 4 + 4;
    if (MODE == 13) off = (lane >> 3) * 512 + (lane & 7) * 4;
    if (MODE == 14) off = lane * 8;
    if (MODE == 15) off = (lane >> 4) * 1028 + (lane & 15) * 4;
    if (MODE == 5) off |= 0x80000000u;
    const bool active = MODE == 4 ? (lane & 15) == 0 : MODE == 7 ? (lane & 3) == 0 : true;
    if (active) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const unsigned o = off + (unsigned)(u * (MODE == 6 ? 64 : 64 * W));
                if (MODE == 1) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, o, 0, 0); acc += v.x ^ v.y; }
                else if (MODE == 2) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, o, 0, 0); acc += v.x ^ v.y ^ v.z; }
                else if (MODE == 3 || MODE == 7) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0); acc += v.x ^ v.y ^ v.z ^ v.w; }
                else acc += __builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0);
            }
            asm volatile("" : "+v"(off));      // keep the loads inside the loop
        }
    }
    if (acc == 0x12345678u) out[threadIdx.x] = acc;
}

__global__ void wrap_probe(const unsigned* src, unsigned* out) {
    // descriptor base = src + 4 dwords: the 16 bytes below it are valid memory, so nothing can fault whatever the hardware does
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)(src + 4), 0, 32, 0x00020000);
    const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, 0xfffffffcu, 0, 0);   // starts 4 bytes below the base
    const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, 0xfffffff8u, 0, 0);   // 8 bytes below
    const u32x3 c = __builtin_amdgcn_raw_buffer_load_b96(rs, 0xfffffffcu, 0, 0);
    const u32x2 d = __builtin_amdgcn_raw_buffer_load_b64(rs, 0xfffffffcu, 0, 0);
    if (threadIdx.x == 0) {
        out[0] = a.x; out[1] = a.y; out[2] = a.z; out[3] = a.w;
        out[4] = b.x; out[5] = b.y; out[6] = b.z; out[7] = b.w;
        out[8] = c.x; out[9] = c.y; out[10] = c.z; out[11] = d.x; out[12] = d.y;
    }
}

template <int MODE>
static void run(const char* name, const unsigned* d, unsigned* o, int bytes_per_lane, int active_lanes) {
    const int iters = 2000, blocks = 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(256), 0, 0, d, o, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    // per CU: 2 workgroups x 4 waves x iters x 8 instructions; clock taken as 2.4 GHz
    const double instr_per_cu = 8.0 * iters * 8, cycles = ms * 1e-3 * 2.4e9;
    printf("%-34s %7.3f ms  %6.1f cycles per wave instruction per CU  %6.1f B/clk/CU\n", name, ms, cycles / instr_per_cu,
           instr_per_cu * bytes_per_lane * active_lanes / cycles);
}

int main() {
    unsigned* h = (unsigned*)malloc(65536);
    for (int i = 0; i < 16384; i++) h[i] = 0x100 + i;
    unsigned *d, *o;
    hipMalloc(&d, 65536); hipMalloc(&o, 4096);
    hipMemcpy(d, h, 65536, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(wrap_probe, dim3(1), dim3(64), 0, 0, d, o);
    unsigned r[13];
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("b128 @ -4: %x %x %x %x   (dwords wrap back one by one -> 0 104 105 106)\n", r[0], r[1], r[2], r[3]);
    printf("b128 @ -8: %x %x %x %x   (-> 0 0 104 105)\n", r[4], r[5], r[6], r[7]);
    printf("b96  @ -4: %x %x %x      (-> 0 104 105)\n", r[8], r[9], r[10]);
    printf("b64  @ -4: %x %x         (-> 0 104)\n", r[11], r[12]);
    run<0>("b32, lanes contiguous", d, o, 4, 64);
    run<1>("b64, lanes contiguous", d, o, 8, 64);
    run<2>("b96, lanes contiguous", d, o, 12, 64);
    run<3>("b128, lanes contiguous", d, o, 16, 64);
    run<4>("b32, 4 of 64 lanes active", d, o, 4, 4);
    run<7>("b128, 16 of 64 lanes active", d, o, 16, 16);
    run<5>("b32, every lane out of range", d, o, 4, 64);
    run<6>("b32, 4 rows x 16 dwords gather", d, o, 4, 64);
    run<8>("b32 contiguous, +4 bytes", d, o, 4, 64);
    run<9>("b32, 2 rows x 32 lanes", d, o, 4, 64);
    run<12>("b32, 2 rows x 32 lanes, +4 bytes", d, o, 4, 64);
    run<10>("b32, 4 rows x 16 lanes", d, o, 4, 64);
    run<11>("b32, 4 rows x 16 lanes, +4 bytes", d, o, 4, 64);
    run<15>("b32, 4 rows x 16 lanes, +0/4/8/12", d, o, 4, 64);
    run<13>("b32, 8 rows x 8 lanes", d, o, 4, 64);
    run<14>("b32, lanes 8 bytes apart", d, o, 4, 64);
    return 0;
}
