// upfirdn2d for gfx950: zero-insert upsample, pad/crop, 2-D FIR, decimate (SG3OPS/upfirdn2d.py:167-211
// is the specification; output size of SG3OPS/upfirdn2d.cpp:35-36).  Generic gather form: the filter
// (with flip and gain folded in) lives in LDS, every lane walks only the taps that hit a real
// sample (polyphase stepping), lanes map to consecutive output columns so reads and writes coalesce.
// Covers every up/down/tap combination incl. the 61-tap separable Gaussian of the loss blur
// (stylegan3_model.py:25-28), which arrives as two 1-D calls (SG3OPS/upfirdn2d.py:244-245).
#include "common.h"

namespace afcm {

struct UpfirdnParams {
    void* y;
    const void* x;
    int xw, xh, yw, yh, fw, fh;
    int upx, upy, downx, downy, padx0, pady0;
    int flip;
    float gain;
    long long planes;
};

constexpr int kMaxTaps = 4096;

template <typename T>
__global__ __launch_bounds__(256) void upfirdn2d_kernel(UpfirdnParams p, const float* __restrict__ f) {
    __shared__ float taps[kMaxTaps];
    const int nt = p.fw * p.fh;
    // taps[ky*fw+kx] multiplies z[Uy+ky][Ux+kx]; a true convolution unless `flip`.
    for (int i = threadIdx.x; i < nt; i += blockDim.x) taps[i] = (p.flip ? f[i] : f[nt - 1 - i]) * p.gain;
    __syncthreads();
    const int tilesX = (p.yw + 63) >> 6;
    const int tilesY = (p.yh + 3) >> 2;
    const long long nblk = (long long)tilesX * tilesY * p.planes;
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int tx = (int)(blk % tilesX);
        const long long t = blk / tilesX;
        const int ty = (int)(t % tilesY);
        const long long plane = t / tilesY;
        const int ox = tx * 64 + lx, oy = ty * 4 + ly;
        if (ox >= p.yw || oy >= p.yh) continue;
        const T* xp = (const T*)p.x + plane * p.xh * p.xw;
        const int Ux = ox * p.downx - p.padx0, Uy = oy * p.downy - p.pady0;
        const int kx0 = pos_mod(-Ux, p.upx), ky0 = pos_mod(-Uy, p.upy);
        const int ix0 = (Ux + kx0) / p.upx, iy0 = (Uy + ky0) / p.upy;   // exact (numerator divisible)
        float acc = 0.f;
        for (int ky = ky0, iy = iy0; ky < p.fh; ky += p.upy, iy++) {
            if ((unsigned)iy >= (unsigned)p.xh) continue;
            const T* row = xp + (size_t)iy * p.xw;
            const float* trow = taps + ky * p.fw;
            for (int kx = kx0, ix = ix0; kx < p.fw; kx += p.upx, ix++)
                if ((unsigned)ix < (unsigned)p.xw) acc = fmaf(trow[kx], to_f32(row[ix]), acc);
        }
        ((T*)p.y)[plane * p.yh * p.yw + (size_t)oy * p.yw + ox] = from_f32<T>(acc);
    }
}

}  // namespace afcm

using namespace afcm;

extern "C" int afcm_upfirdn2d(void* y, const void* x, const float* f, int32_t dtype, int32_t n, int32_t c, int32_t xh, int32_t xw,
                              int32_t yh, int32_t yw, int32_t fh, int32_t fw, int32_t upx, int32_t upy, int32_t downx, int32_t downy,
                              int32_t padx0, int32_t pady0, int32_t flip, float gain, void* stream) {
    AFCM_REQUIRE(x != nullptr && y != nullptr && f != nullptr, "upfirdn2d: x, y and f must be non-null");
    AFCM_REQUIRE(dtype == AFCM_F32 || dtype == AFCM_F16 || dtype == AFCM_BF16, "x must be float32, float16 or bfloat16");
    AFCM_REQUIRE(n > 0 && c > 0 && xh > 0 && xw > 0, "x is empty");
    AFCM_REQUIRE(fh >= 1 && fw >= 1, "f is empty");
    AFCM_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "upsampling and downsampling factors must be at least 1");
    AFCM_REQUIRE(yh >= 1 && yw >= 1, "output must be at least 1x1");
    if ((long long)fh * fw > kMaxTaps) return AFCM_E_NOKERNEL;
    UpfirdnParams p;
    p.y = y; p.x = x; p.xw = xw; p.xh = xh; p.yw = yw; p.yh = yh; p.fw = fw; p.fh = fh;
    p.upx = upx; p.upy = upy; p.downx = downx; p.downy = downy; p.padx0 = padx0; p.pady0 = pady0;
    p.flip = flip; p.gain = gain; p.planes = (long long)n * c;
    long long nblk = (long long)((yw + 63) >> 6) * ((yh + 3) >> 2) * p.planes;
    if (nblk > 256 * 64) nblk = 256 * 64;
    dim3 grid((unsigned)nblk), block(256);
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case AFCM_F32: hipLaunchKernelGGL((upfirdn2d_kernel<float>), grid, block, 0, st, p, f); break;
        case AFCM_F16: hipLaunchKernelGGL((upfirdn2d_kernel<f16_t>), grid, block, 0, st, p, f); break;
        default: hipLaunchKernelGGL((upfirdn2d_kernel<bf16_t>), grid, block, 0, st, p, f); break;
    }
    return hip_status(hipGetLastError());
}
