// Read-pattern microbenchmark for the wave filtered_lrelu kernels (gfx950): bf16 planes [PL][H][LD], one wave per 32-row strip,
// 4 waves per workgroup = 4 consecutive strips, 3 workgroups per CU (LDS pad), nothing but the loads and a dependent integer
// chain of `spin` steps per group standing for the arithmetic.
//   MODE 0: the kernels' pattern: per group the 48-row x 64-byte window of the strip (3 x buffer_load_dwordx4, lanes = 16 rows x
//           4 chunks), windows advance 32 bytes per group (each byte is requested twice), DEPTH groups ahead of their use
//   MODE 1: whole lines: per 4 groups the 48 rows x 128 bytes the windows advance over (6 loads of 8 rows x 128 B), DEPTH x 4 groups ahead
// Build: hipcc -O3 --offload-arch=gfx950 strip_read.hip -o strip_read.bin ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE, int DEPTH, int LDSKB>
__global__ __launch_bounds__(256) void k(const unsigned short* x, unsigned* out, int PL, int H, int LD, int W, int strips, int spin, int reps) {
    __shared__ unsigned pad[LDSKB * 1024 / 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, wt = blockIdx.x * 4 + wave;
    if (threadIdx.x == 0) pad[0] = 0;
    if (wt >= PL * strips) return;
    const int plane = wt / strips, ty = wt - plane * strips;
    const unsigned short* xp = x + (size_t)plane * H * LD;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, H * LD * 2, 0x00020000);
    const int row0 = ty * 32 - 7;
    unsigned acc = 0;
    for (int rep = 0; rep < reps; rep++)
    if (MODE == 0) {
        const int lrow = lane >> 2, lchk = lane & 3;
        const int ng = (W * 2 + 31) / 32;                              // 32 bytes of advance per group
        auto off = [&](int mb, int gi) { const int r = row0 + 16 * mb + lrow; return (r < 0) ? 0x80000000u : (unsigned)(r * LD * 2 + 32 * gi + 16 * lchk); };
        u32x4 q[DEPTH + 1][3];
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            for (int mb = 0; mb < 3; mb++) q[d][mb] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(mb, d), 0, 0);
        for (int gi = 0; gi < ng; gi++) {
#pragma unroll
            for (int mb = 0; mb < 3; mb++) q[DEPTH][mb] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(mb, gi + DEPTH), 0, 0);
            unsigned v = q[0][0][0] ^ q[0][1][1] ^ q[0][2][2] ^ q[0][0][3] ^ q[0][1][0] ^ q[0][2][1] ^ q[0][0][2] ^ q[0][1][3] ^ q[0][2][0] ^ q[0][0][1] ^ q[0][1][2] ^ q[0][2][3];
            for (int s = 0; s < spin; s++) v = v * 1664525u + 1013904223u;
            acc ^= v;
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
#pragma unroll
                for (int mb = 0; mb < 3; mb++) q[d][mb] = q[d + 1][mb];
        }
    } else if (MODE == 2) {
        // half windows: per group the NEW 32 bytes of each of the 48 rows (the other half of a window is the previous group's): lanes =
        // 32 rows x 2 chunks per instruction, 1.5 instructions per group (issued as 3 per two groups), each byte requested once
        const int lrow = lane >> 1, lchk = lane & 1;
        const int ng = (W * 2 + 31) / 32;
        auto off = [&](int j, int gi) { const int r = row0 + 32 * j + lrow; return (r < 0 || 32 * j + lrow >= 48) ? 0x80000000u : (unsigned)(r * LD * 2 + 32 * gi + 16 * lchk); };
        u32x4 q[DEPTH + 1][2];
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            for (int j = 0; j < 2; j++) q[d][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(j, d), 0, 0);
        for (int gi = 0; gi < ng; gi++) {
#pragma unroll
            for (int j = 0; j < 2; j++) q[DEPTH][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(j, gi + DEPTH), 0, 0);
            unsigned v = q[0][0][0] ^ q[0][1][1] ^ q[0][0][2] ^ q[0][1][3] ^ q[0][1][0] ^ q[0][0][1] ^ q[0][1][2] ^ q[0][0][3];
            for (int s = 0; s < spin; s++) v = v * 1664525u + 1013904223u;
            acc ^= v;
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
#pragma unroll
                for (int j = 0; j < 2; j++) q[d][j] = q[d + 1][j];
        }
    } else if (MODE == 3) {
        // 64-byte pieces, not overlapping: per 2 groups the 48 rows x 64 bytes (3 loads of 16 rows x 64 B), each byte requested once
        const int lrow = lane >> 2, lchk = lane & 3;
        const int ns = (W * 2 + 63) / 64;
        auto off = [&](int mb, int st) { const int r = row0 + 16 * mb + lrow; return (r < 0) ? 0x80000000u : (unsigned)(r * LD * 2 + 64 * st + 16 * lchk); };
        u32x4 q[DEPTH + 1][3];
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            for (int mb = 0; mb < 3; mb++) q[d][mb] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(mb, d), 0, 0);
        for (int st = 0; st < ns; st++) {
#pragma unroll
            for (int mb = 0; mb < 3; mb++) q[DEPTH][mb] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(mb, st + DEPTH), 0, 0);
            unsigned v = q[0][0][0] ^ q[0][1][1] ^ q[0][2][2] ^ q[0][0][3] ^ q[0][1][0] ^ q[0][2][1] ^ q[0][0][2] ^ q[0][1][3] ^ q[0][2][0] ^ q[0][0][1] ^ q[0][1][2] ^ q[0][2][3];
            for (int s = 0; s < 2 * spin; s++) v = v * 1664525u + 1013904223u;
            acc ^= v;
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
#pragma unroll
                for (int mb = 0; mb < 3; mb++) q[d][mb] = q[d + 1][mb];
        }
    } else {
        const int lrow = lane >> 3, lchk = lane & 7;
        const int ns = (W * 2 + 127) / 128;                            // 128 bytes per step = 4 groups
        auto off = [&](int j, int st) { const int r = row0 + 8 * j + lrow; return (r < 0) ? 0x80000000u : (unsigned)(r * LD * 2 + 128 * st + 16 * lchk); };
        u32x4 q[DEPTH + 1][6];
#pragma unroll
        for (int d = 0; d < DEPTH; d++)
            for (int j = 0; j < 6; j++) q[d][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(j, d), 0, 0);
        for (int st = 0; st < ns; st++) {
#pragma unroll
            for (int j = 0; j < 6; j++) q[DEPTH][j] = __builtin_amdgcn_raw_buffer_load_b128(rs, off(j, st + DEPTH), 0, 0);
            unsigned v = 0;
#pragma unroll
            for (int j = 0; j < 6; j++) v ^= q[0][j][0] ^ q[0][j][1] ^ q[0][j][2] ^ q[0][j][3];
            for (int s = 0; s < 4 * spin; s++) v = v * 1664525u + 1013904223u;
            acc ^= v;
#pragma unroll
            for (int d = 0; d < DEPTH; d++)
#pragma unroll
                for (int j = 0; j < 6; j++) q[d][j] = q[d + 1][j];
        }
    }
    if (acc == 0x12345678u) out[wt] = acc + pad[lane];
}

template <int MODE, int DEPTH, int LDSKB> void run(const char* name, int PL, int H, int W, int LD, int spin, unsigned short* x, unsigned* out, int reps = 1) {
    const int strips = (H + 31) / 32, waves = PL * strips;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 2; i++) k<MODE, DEPTH, LDSKB><<<(waves + 3) / 4, 256>>>(x, out, PL, H, LD, W, strips, spin, reps);
    hipEventRecord(e0);
    for (int i = 0; i < 10; i++) k<MODE, DEPTH, LDSKB><<<(waves + 3) / 4, 256>>>(x, out, PL, H, LD, W, strips, spin, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("%-22s depth %d, %d WG/CU, spin %4d, %5d planes x %3d sweeps: %7.3f ms  %7.1f GB/s of plane bytes\n", name, DEPTH, 160 / LDSKB, spin, PL, reps, ms, (double)reps * PL * H * LD * 2 / ms / 1e6);
}
int main(int argc, char** argv) {
    const int PL = 16 * 128, H = 278, W = 278, LD = 288;
    unsigned short* x; unsigned* out;
    hipMalloc(&x, (size_t)PL * H * LD * 2 + (1 << 20)); hipMalloc(&out, PL * 16 * 4);
    hipMemset(x, 1, (size_t)PL * H * LD * 2);
    printf("planes %d x %d x %d (pitch %d) bf16 = %.0f MB; 48 rows are read per 32-row strip\n", PL, H, W, LD, (double)PL * H * LD * 2 / 1e6);
    for (int spin : {0, 60, 100}) {
        run<0, 1, 40>("windows 64B x 16 rows", PL, H, W, LD, spin, x, out);
        run<0, 2, 40>("windows 64B x 16 rows", PL, H, W, LD, spin, x, out);
        run<1, 1, 40>("lines 128B x 8 rows", PL, H, W, LD, spin, x, out);
        run<2, 1, 40>("halves 32B x 32 rows", PL, H, W, LD, spin, x, out);
        run<2, 2, 40>("halves 32B x 32 rows", PL, H, W, LD, spin, x, out);
        run<3, 1, 40>("pieces 64B x 16 rows", PL, H, W, LD, spin, x, out);
        run<3, 2, 40>("pieces 64B x 16 rows", PL, H, W, LD, spin, x, out);
    }
    // the same patterns on a working set that stays in the L2s (8 x 4 MB): 96 planes = 15 MB, swept 40 times by the same waves --
    // what the CU <-> L2 path delivers when HBM is out of the picture
    for (int spin : {0}) {
        run<0, 1, 40>("windows 64B x 16 rows", 96, H, W, LD, spin, x, out, 40); run<0, 1, 40>("windows 64B x 16 rows", 1024, H, W, LD, spin, x, out, 6); run<3, 1, 40>("pieces 64B x 16 rows", 1024, H, W, LD, spin, x, out, 6); run<1, 1, 40>("lines 128B x 8 rows", 1024, H, W, LD, spin, x, out, 6); run<3, 1, 40>("pieces 64B x 16 rows", 384, H, W, LD, spin, x, out, 16);
        run<1, 1, 40>("lines 128B x 8 rows", 96, H, W, LD, spin, x, out, 40);
        run<3, 1, 40>("pieces 64B x 16 rows", 96, H, W, LD, spin, x, out, 40);
        run<3, 2, 40>("pieces 64B x 16 rows", 96, H, W, LD, spin, x, out, 40);
    }
    return 0;
}
