// All style affine layers of the co-modulated decoder in one launch each way.
//
// Every SynthesisLayer starts with `styles = self.affine(cat(w, global_w))` (NET:349-352; FullyConnectedLayer, NET:69-104): 15 dense
// layers [N, 512 + 1024] x [Cin_l, 1536]^T of an equalised-lr FC, N = the batch (16).  As framework GEMMs that is 15 cat + 15 GEMM
// launches forward and 30 GEMMs + 15 bias reductions + 14 accumulations of the shared input's gradient backward, each 4-10 us of
// launch latency between two convolutions that fill the chip (profiles/r03_bench_kernel_stats.csv: ~0.6 ms per step).  The whole bank
// reads 32 MB of weights: three memory-bound launches.
//   forward   y_l[n][c]  = alpha_l * sum_k x_l[n][k] W_l[c][k] + beta_l * b_l[c]        x_l[n] = [ ws[n][idx_l][0:kw] | g[n][0:kg] ]
//   backward  dW_l[c][k] = alpha_l * sum_n dy_l[n][c] x_l[n][k]        db_l[c] = beta_l * sum_n dy_l[n][c]
//             dws[n][l][k] = alpha_l * sum_c dy_l[n][c] W_l[c][k]  (k < kw)        dg[n][k] = sum_l alpha_l * sum_c dy_l[n][c] W_l[c][kw + k]
// (ToRGB's extra 1 / sqrt(Cin k^2), NET:351, rides in alpha / beta.)  No atomics: the input gradient goes through per-(layer, row block)
// partial sums and a second small kernel, so results are bit-reproducible.
#include "common.h"

namespace afcm {

constexpr int kNB = 16;                 // batch rows per launch (the caller loops over larger batches)
constexpr int kKMax = 1536;             // kw + kg
constexpr int kRowsW = 16;              // output rows per workgroup, forward / weight gradient (4 waves x 4 rows)
constexpr int kRowsX = 64;              // weight rows per wave, input gradient

struct BankArgs {                       // by value in the kernel arguments (~1.2 KB): no device-side tables to keep in step
    afcm_affine_bank a;
    int n0, nb;                         // batch rows [n0, n0 + nb) of this launch
    int blk0[AFCM_AFFINE_MAX + 1];      // first workgroup of each layer
    const float* t0[AFCM_AFFINE_MAX];   // y (forward, written) / dy (backward)
    float* t1[AFCM_AFFINE_MAX];         // dW
    float* t2[AFCM_AFFINE_MAX];         // db
};

__device__ __forceinline__ int find_layer(const BankArgs& A, int blk) {
    int l = 0;
#pragma unroll 1
    while (l + 1 < A.a.layers && A.blk0[l + 1] <= blk) l++;
    return l;
}

// r06: forward and weight gradient on the exact-fp32 matrix instruction (v_mfma_f32_16x16x4_f32, as csrc/fc_bank.hip: lane maps and the
// "one 16-byte load = four contraction indices = four MFMAs" form are described there).  The r03 kernels staged the layer's input
// [16][1536] into 96 KB of LDS in EVERY workgroup (one workgroup per CU, 28 MB of re-reads over the bank), gave a wave four rows in turn
// and reduced sixteen accumulators over the lanes with 96 shuffles per row: 64 us forward, 58 us weight gradient for 29 MB of weights.
typedef __attribute__((ext_vector_type(4))) float abx4;

// x_l[n][k .. k + 3] (k a multiple of 4): the latent's slice for k < kw, the global vector behind it
__device__ __forceinline__ abx4 bank_x4(const BankArgs& A, int l, int n, int k) {
    const float* src = (k < A.a.kw) ? A.a.w + (size_t)(A.n0 + n) * A.a.w_stride_n + (size_t)A.a.w_index[l] * A.a.w_stride_l + k
                                    : A.a.g + (size_t)(A.n0 + n) * A.a.kg + (k - A.a.kw);
    return *(const abx4*)src;
}

__global__ __launch_bounds__(256) void affine_bank_fwd_kernel(BankArgs A) {
    __shared__ abx4 red[3][64];
    const int l = find_layer(A, blockIdx.x);
    const int K = A.a.kw + A.a.kg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const int cout = A.a.cout[l];
    const int row0 = (blockIdx.x - A.blk0[l]) * kRowsW;
    const float* __restrict__ wp = A.a.weight[l] + (size_t)min(row0 + l15, cout - 1) * K + 4 * g;
    const bool live = l15 < A.nb;
    const int nrow = live ? l15 : 0;
    abx4 acc = {0.f, 0.f, 0.f, 0.f};
    // 16 contraction indices per chunk; wave w takes a contiguous quarter of the chunks
    const int chunks = K >> 4, per = (chunks + 3) >> 2;
    const int c_end = min(chunks, (wave + 1) * per);
    constexpr int U = 8;
    int c = wave * per;
    for (; c + U <= c_end; c += U) {
        abx4 a[U], bv[U];
#pragma unroll
        for (int u = 0; u < U; u++) { a[u] = *(const abx4*)(wp + 16 * (c + u)); bv[u] = bank_x4(A, l, nrow, 16 * (c + u) + 4 * g); }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const abx4 bb = live ? bv[u] : abx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u][j], bb[j], acc, 0, 0, 0);
        }
    }
    for (; c < c_end; c++) {
        const abx4 a = *(const abx4*)(wp + 16 * c);
        abx4 bb = bank_x4(A, l, nrow, 16 * c + 4 * g);
        if (!live) bb = abx4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; j++) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bb[j], acc, 0, 0, 0);
    }
    if (wave > 0) red[wave - 1][lane] = acc;
    __syncthreads();
    if (wave != 0 || !live) return;
    acc += red[0][lane] + red[1][lane] + red[2][lane];
    // D[i = 4 g + r][j = l15]: rows row0 + 4 g + r of sample n0 + l15
    float* __restrict__ yl = (float*)A.t0[l];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int row = row0 + 4 * g + r;
        if (row >= cout) continue;
        const float b = A.a.bias[l] ? A.a.bias[l][row] : 0.f;
        yl[(size_t)(A.n0 + l15) * cout + row] = A.a.alpha[l] * acc[r] + A.a.beta[l] * b;
    }
}

// dW_l, db_l: same workgroup map as the forward (16 rows x the whole K, the 96 column tiles of 16 over the 4 waves); `accumulate`: add to
// what is there (second batch half).  D[i = row][j = k] = sum_n A[row][n] B[n][k], A = dy[n = 4 g + j][row0 + l15], B = x[n = 4 g + j][k0 + l15]
__global__ __launch_bounds__(256) void affine_bank_bwd_w_kernel(BankArgs A, int accumulate) {
    const int l = find_layer(A, blockIdx.x);
    const int K = A.a.kw + A.a.kg, kw = A.a.kw;
    const int cout = A.a.cout[l];
    const int row0 = (blockIdx.x - A.blk0[l]) * kRowsW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l15 = lane & 15, g = lane >> 4;
    const float* gy = A.t0[l];
    float* __restrict__ dWl = A.t1[l];
    float* __restrict__ dbl = A.t2[l];
    const float alpha = A.a.alpha[l];
    float a[4];
    float colsum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = 4 * g + j;
        a[j] = (gy != nullptr && n < A.nb && row0 + l15 < cout) ? gy[(size_t)(A.n0 + n) * cout + row0 + l15] : 0.f;
        colsum += a[j];
    }
    if (dbl != nullptr && wave == 0) {
        colsum += __shfl_xor(colsum, 16);
        colsum += __shfl_xor(colsum, 32);
        if (g == 0 && row0 + l15 < cout) dbl[row0 + l15] = A.a.beta[l] * colsum + (accumulate ? dbl[row0 + l15] : 0.f);
    }
    if (dWl == nullptr) return;
    const float* xb[4];                                           // this lane's four samples: row bases of the latent slice and of the global vector
    const float* xg[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int n = min(4 * g + j, A.nb - 1);
        xb[j] = A.a.w + (size_t)(A.n0 + n) * A.a.w_stride_n + (size_t)A.a.w_index[l] * A.a.w_stride_l;
        xg[j] = A.a.g + (size_t)(A.n0 + n) * A.a.kg - kw;
    }
    const int ktiles = K >> 4;
    constexpr int U = 4;
    for (int kt = wave; kt < ktiles; kt += 4 * U) {
        float bv[U][4];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int k = 16 * (kt + 4 * u) + l15;
#pragma unroll
            for (int j = 0; j < 4; j++) bv[u][j] = (kt + 4 * u < ktiles) ? ((k < kw) ? xb[j][k] : xg[j][k]) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (kt + 4 * u >= ktiles) break;                      // (wave-uniform)
            abx4 d = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++) d = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], bv[u][j], d, 0, 0, 0);
            const int k = 16 * (kt + 4 * u) + l15;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = row0 + 4 * g + r;
                if (row >= cout) continue;
                float* dst = dWl + (size_t)row * K + k;
                *dst = alpha * d[r] + (accumulate ? *dst : 0.f);
            }
        }
    }
}

// input gradient, stage 1: one wave per (layer, block of kRowsX weight rows, 256-float chunk of k): partial[blk][n][k]
__global__ __launch_bounds__(64) void affine_bank_bwd_x_kernel(BankArgs A, float* __restrict__ part, int kchunks) {
    __shared__ float sdy[kRowsX][kNB];
    const int rb = blockIdx.x / kchunks, kc = blockIdx.x - rb * kchunks;       // row block (global over layers), k chunk
    const int l = find_layer(A, rb);
    const int K = A.a.kw + A.a.kg, cout = A.a.cout[l];
    const int row0 = (rb - A.blk0[l]) * kRowsX;
    const int lane = threadIdx.x;
    const float* g = A.t0[l];
    for (int i = lane; i < kRowsX * kNB; i += 64) {
        const int r = i / kNB, n = i - r * kNB;
        sdy[r][n] = (g != nullptr && n < A.nb && row0 + r < cout) ? g[(size_t)(A.n0 + n) * cout + row0 + r] : 0.f;
    }
    __syncthreads();
    const int k = kc * 256 + 4 * lane;
    float4 acc[kNB];
#pragma unroll
    for (int n = 0; n < kNB; n++) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < K) {
        const float* __restrict__ W = A.a.weight[l];
        const int rows = min(kRowsX, cout - row0);
#pragma unroll 4
        for (int r = 0; r < rows; r++) {
            const float4 wv = *(const float4*)(W + (size_t)(row0 + r) * K + k);
#pragma unroll
            for (int n = 0; n < kNB; n++) {
                const float s = sdy[r][n];
                acc[n].x += s * wv.x; acc[n].y += s * wv.y; acc[n].z += s * wv.z; acc[n].w += s * wv.w;
            }
        }
        const float alpha = A.a.alpha[l];
#pragma unroll
        for (int n = 0; n < kNB; n++)
            *(float4*)(part + ((size_t)rb * kNB + n) * K + k) = make_float4(alpha * acc[n].x, alpha * acc[n].y, alpha * acc[n].z, alpha * acc[n].w);
    }
}

// stage 2: dws[n][l][k] = sum over the layer's row blocks; dg[n][k] = sum over all layers' row blocks
__global__ __launch_bounds__(256) void affine_bank_bwd_x_reduce_kernel(BankArgs A, const float* __restrict__ part, float* __restrict__ dws, float* __restrict__ dg) {
    const int K = A.a.kw + A.a.kg, kw = A.a.kw;
    const int idx = blockIdx.x * 256 + threadIdx.x;                       // over nb x K
    if (idx >= A.nb * K) return;
    const int n = idx / K, k = idx - n * K;
    if (k < kw) {
        for (int l = 0; l < A.a.layers; l++) {
            float s = 0.f;
            for (int rb = A.blk0[l]; rb < A.blk0[l + 1]; rb++) s += part[((size_t)rb * kNB + n) * K + k];
            dws[((size_t)(A.n0 + n) * A.a.layers + l) * kw + k] = s;
        }
    } else {
        float s = 0.f;
        for (int rb = 0; rb < A.blk0[A.a.layers]; rb++) s += part[((size_t)rb * kNB + n) * K + k];
        dg[(size_t)(A.n0 + n) * A.a.kg + (k - kw)] = s;
    }
}

static int check_bank(const afcm_affine_bank* a) {
    AFCM_REQUIRE(a != nullptr && a->layers >= 1 && a->layers <= AFCM_AFFINE_MAX, "affine_bank: 1..%d layers", AFCM_AFFINE_MAX);
    AFCM_REQUIRE(a->n >= 1 && a->kw >= 4 && a->kg >= 0 && a->w != nullptr && (a->kg == 0 || a->g != nullptr), "affine_bank: empty input");
    for (int l = 0; l < a->layers; l++) AFCM_REQUIRE(a->weight[l] != nullptr && a->cout[l] >= 1, "affine_bank: layer %d has no weight", l);
    return AFCM_OK;
}
// the kernels' shapes: K a multiple of 4 up to kKMax, 16-byte aligned rows
static bool bank_supported(const afcm_affine_bank* a) {
    const int K = a->kw + a->kg;
    if (K > kKMax || (K & 15) || (a->kw & 3) || (a->kg & 3) || (a->w_stride_n & 3) || (a->w_stride_l & 3)) return false;   // (K % 16: the matrix kernels' chunks)
    if (((uintptr_t)a->w | (uintptr_t)a->g) & 15) return false;
    for (int l = 0; l < a->layers; l++)
        if ((uintptr_t)a->weight[l] & 15) return false;
    return true;
}
static void fill_blocks(BankArgs& A, int rows_per_block) {
    A.blk0[0] = 0;
    for (int l = 0; l < A.a.layers; l++) A.blk0[l + 1] = A.blk0[l] + cdiv(A.a.cout[l], rows_per_block);
    for (int l = A.a.layers + 1; l <= AFCM_AFFINE_MAX; l++) A.blk0[l] = A.blk0[A.a.layers];
}

}  // namespace afcm

using namespace afcm;

extern "C" int64_t afcm_affine_bank_workspace_bytes(const afcm_affine_bank* a) {
    if (a == nullptr || a->layers < 1 || a->layers > AFCM_AFFINE_MAX) return 0;
    long long blocks = 0;
    for (int l = 0; l < a->layers; l++) blocks += cdiv(a->cout[l], kRowsX);
    return blocks * kNB * (long long)(a->kw + a->kg) * 4;           // partial sums of the input gradient [row blocks][16][K]
}

extern "C" int afcm_affine_bank_fwd(const afcm_affine_bank* a, float* const* y, void* stream) {
    int rc = check_bank(a);
    if (rc != AFCM_OK) return rc;
    AFCM_REQUIRE(y != nullptr, "affine_bank: null output table");
    if (!bank_supported(a)) return AFCM_E_NOKERNEL;
    hipStream_t st = (hipStream_t)stream;
    BankArgs A = {};
    A.a = *a;
    for (int l = 0; l < a->layers; l++) {
        AFCM_REQUIRE(y[l] != nullptr, "affine_bank: layer %d has no output", l);
        A.t0[l] = y[l];
    }
    fill_blocks(A, kRowsW);
    for (int n0 = 0; n0 < a->n; n0 += kNB) {
        A.n0 = n0; A.nb = a->n - n0 < kNB ? a->n - n0 : kNB;
        hipLaunchKernelGGL(affine_bank_fwd_kernel, dim3(A.blk0[a->layers]), dim3(256), 0, st, A);
    }
    return hip_status(hipGetLastError());
}

extern "C" int afcm_affine_bank_bwd(const afcm_affine_bank* a, const float* const* dy, float* const* dweight, float* const* dbias, float* dws, float* dg,
                                    void* workspace, void* stream) {
    int rc = check_bank(a);
    if (rc != AFCM_OK) return rc;
    AFCM_REQUIRE(dy != nullptr, "affine_bank: null gradient table");
    if (!bank_supported(a)) return AFCM_E_NOKERNEL;
    hipStream_t st = (hipStream_t)stream;
    const int K = a->kw + a->kg;
    BankArgs A = {};
    A.a = *a;
    for (int l = 0; l < a->layers; l++) {
        A.t0[l] = dy[l];                                   // NULL: the layer's styles received no gradient (zeros)
        A.t1[l] = dweight ? dweight[l] : nullptr;
        A.t2[l] = dbias ? dbias[l] : nullptr;
    }
    for (int n0 = 0; n0 < a->n; n0 += kNB) {
        A.n0 = n0; A.nb = a->n - n0 < kNB ? a->n - n0 : kNB;
        if (dweight != nullptr || dbias != nullptr) {
            fill_blocks(A, kRowsW);
            hipLaunchKernelGGL(affine_bank_bwd_w_kernel, dim3(A.blk0[a->layers]), dim3(256), 0, st, A, n0 > 0 ? 1 : 0);
        }
        if (dws != nullptr || dg != nullptr) {
            AFCM_REQUIRE(dws != nullptr && (a->kg == 0 || dg != nullptr) && workspace != nullptr, "affine_bank: dws, dg and the workspace come together");
            fill_blocks(A, kRowsX);
            const int kchunks = cdiv(K, 256);
            hipLaunchKernelGGL(affine_bank_bwd_x_kernel, dim3(A.blk0[a->layers] * kchunks), dim3(64), 0, st, A, (float*)workspace, kchunks);
            hipLaunchKernelGGL(affine_bank_bwd_x_reduce_kernel, dim3(cdiv(A.nb * K, 256)), dim3(256), 0, st, A, (const float*)workspace, dws, dg);
        }
    }
    return hip_status(hipGetLastError());
}
