// Which offsets does a raw buffer load's range check cover on gfx950?  One lane, b32 loads from a 2 MB allocation through a
// descriptor with num_records = 4096; the memory holds (byte offset / 4) + 1, so 0 = the load was treated as out of range.
// hipcc -O3 --offload-arch=gfx950 tools/ubench/buffer_range_probe.hip -o /tmp/buffer_range_probe && /tmp/buffer_range_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

struct Case { unsigned v, s; };
__global__ void probe(const unsigned* base, unsigned nr, const Case* cases, int n, unsigned* out) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)nr, 0x00020000);
    for (int i = 0; i < n; i++) {
        const unsigned s = __builtin_amdgcn_readfirstlane(cases[i].s);
        out[i] = __builtin_amdgcn_raw_buffer_load_b32(rs, cases[i].v, s, 0);
    }
}
int main() {
    const size_t bytes = 2u << 20;
    std::vector<unsigned> h(bytes / 4);
    for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)i + 1;
    unsigned *d, *o; Case* c;
    hipMalloc(&d, bytes); hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
    std::vector<Case> cs = {{0, 0}, {4092, 0}, {4096, 0}, {0, 4096}, {0, 8192}, {0x80000, 8192}, {0x80000, 0}, {2048, 1024}, {3068, 1024},
                            {3072, 1024}, {4092, 4}, {4088, 4}, {0x80000, 0x100000}, {0, 0x100000}, {100, 4000}};
    hipMalloc(&c, cs.size() * sizeof(Case)); hipMemcpy(c, cs.data(), cs.size() * sizeof(Case), hipMemcpyHostToDevice);
    hipMalloc(&o, cs.size() * 4);
    probe<<<1, 1>>>(d, 4096, c, (int)cs.size(), o);
    std::vector<unsigned> r(cs.size());
    hipMemcpy(r.data(), o, cs.size() * 4, hipMemcpyDeviceToHost);
    printf("num_records 4096\n%12s %12s %12s  %s\n", "voffset", "soffset", "value", "meaning");
    for (size_t i = 0; i < cs.size(); i++) {
        const unsigned long long sum = (unsigned long long)cs[i].v + cs[i].s;
        printf("%#12x %#12x %12u  %s\n", cs[i].v, cs[i].s, r[i],
               r[i] == 0 ? "zero (out of range)" : (r[i] == (unsigned)(sum & 0xffffffffu) / 4 + 1 ? "memory at base + v + s" : "?"));
    }
    return 0;
}
