// Probe: are raw buffer loads range-checked per dword?  (a 16-byte load that straddles num_records)
// build: hipcc --offload-arch=gfx950 -O2 -o oob_probe oob_probe.hip ; run: ./oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
__global__ void probe(const unsigned* src, unsigned* out, int records) {
    __shared__ __attribute__((aligned(16))) unsigned l[256];
    __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, records, 0x00020000);
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, 8, 0, 0);     // bytes 8..23
    const u32x2 w = __builtin_amdgcn_raw_buffer_load_b64(r, 16, 0, 0);     // bytes 16..23
    l[threadIdx.x * 4 + 0] = 0xdeadbeef; l[threadIdx.x * 4 + 1] = 0xdeadbeef; l[threadIdx.x * 4 + 2] = 0xdeadbeef; l[threadIdx.x * 4 + 3] = 0xdeadbeef;
    __syncthreads();
    i32x4 rs; rs.x = (int)(unsigned)(unsigned long long)src; rs.y = (int)((unsigned long long)src >> 32) & 0xffff; rs.z = records; rs.w = 0x00020000;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) void*)l;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" : : "s"(lds0), "v"(8u + 4u * 0u), "s"(rs) : "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w; out[4] = w.x; out[5] = w.y;
        out[6] = l[0]; out[7] = l[1]; out[8] = l[2]; out[9] = l[3];
    }
}
int main() {
    unsigned h[16]; for (int i = 0; i < 16; i++) h[i] = 0x100 + i;
    unsigned *d, *o; hipMalloc(&d, 64); hipMalloc(&o, 64); hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 20);
    unsigned r[10]; hipMemcpy(r, o, 40, hipMemcpyDeviceToHost);
    printf("b128@8  records=20: %x %x %x %x   (per-dword check -> 102 103 104 0)\n", r[0], r[1], r[2], r[3]);
    printf("b64@16  records=20: %x %x         (per-dword check -> 104 0)\n", r[4], r[5]);
    printf("lds-dma b128@8    : %x %x %x %x   (per-dword check -> 102 103 104 0)\n", r[6], r[7], r[8], r[9]);
    const int ok = r[0] == 0x102 && r[1] == 0x103 && r[2] == 0x104 && r[3] == 0 && r[4] == 0x104 && r[5] == 0 && r[6] == 0x102 && r[7] == 0x103 && r[8] == 0x104 && r[9] == 0;
    printf(ok ? "PER-DWORD\n" : "NOT PER-DWORD\n");
    return ok ? 0 : 1;
}
