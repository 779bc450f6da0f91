// filtered_lrelu on the matrix cores, for 16-bit activations (bf16 / f16 storage).
//
// Why MFMA here: on gfx950 the fp32 vector pipe peaks at ~63 T lane-FMA/s (measured, tools/ubench/valu_rate.hip;
// packed and dot2 forms run at half the instruction rate, so there is no faster VALU form), and this op needs
// ~11 G FMA per 256^2 image against 0.62 GB of 16-bit traffic: on the vector pipe it is compute-bound at ~2x the
// HBM time before any overhead.  The four separable FIR passes are banded-Toeplitz matrix products; run as
// v_mfma_f32_16x16x32_{bf16,f16} they cost ~0.25 CU-cycles per output pixel even at ~19-25 % band occupancy.
// What is left for the vector pipe is kept to ~1 instruction per upsampled element (r02):
//   * the leaky ReLU itself runs on the matrix cores:  lrelu(v) = slope*v + (1-slope)*relu(v), so
//         X3 = DV*lrelu(X2) = (slope*DV)*X2 + ((1-slope)*DV)*relu(X2)
//     and relu() of packed 16-bit floats is ONE v_pk_max_i16 against 0 for two elements (sign-magnitude bit patterns
//     order like signed integers below zero).  Backward is the same identity with relu(X2) replaced by X2 AND a
//     16-bit-per-element mask looked up from the saved sign codes (256-entry LDS table).
//   * the clamp (rarely reached: |v| > 256) is detected per 16x16 tile with a wave-uniform ballot; only then does a
//     tile take the exact per-element path.
//   * the input tile is copied HBM -> LDS without conversion (the caller's producer adds the bias; a template flag
//     keeps the fused-bias staging for direct API users), sign codes and outputs leave through LDS as full 16-byte
//     row segments.
//
// Dataflow per tile (fp32 accumulation everywhere, 16-bit operands):
//   In (LDS, [row][col])  --A-->  X1  = In * UH         up-FIR along x    D[in-row][ucol]
//   X1 (registers)        --B-->  X2  = UV * X1         up-FIR along y    D[urow][ucol]    (accumulator tile used as
//   X2, relu(X2) (regs)   --A-->  X3' = X2' * DV'       down-FIR along y  D[ucol][orow]     the next operand: no LDS)
//   X3 (LDS, [orow][ucol]) -B-->  Y'  = DH' * X3'       down-FIR along x  D[ocol][orow]    (lane = 4 consecutive columns)
//   Y (LDS, [orow][ocol])  ---->  16-byte row segments to HBM
// UH/UV/DV/DH are constant Toeplitz fragments built once per layer by flrelu_mfma_prepare_kernel.
// Only In, X3 and Y touch LDS; the up^2-times-larger activated intermediate lives in accumulators.
//
// Sign codes use a private "row-quad" layout: one byte = the codes of 4 consecutive rows of one column
// ([N*C][ceil(sh/4)][swq], code of row r at bits 2r..2r+1: bit0 = negative, bit1 = clamped); forward assembles them
// from the packed X2 words, backward (the same kernel with up/down swapped) reads them with a funnel shift for the
// row offset.
#include "flrelu_mfma_common.h"

namespace afcm {

// Geometry shared by the kernel, the prepare kernel and the host.
template <int UP, int DOWN, int TOW, int TOH>
struct MfmaGeom {
    static constexpr int FU = kFUT * UP, FD = kFUT * DOWN;
    static constexpr int TUW = (TOW - 1) * DOWN + FD, TUH = (TOH - 1) * DOWN + FD;
    static constexpr int NB = 3, GW = 16 * NB;                 // ucol blocks per group; every group shares one 32-wide input window
    static constexpr int NG = cdiv(cdiv(TUW, 16), NB);         // groups (= waves)
    static constexpr int NVB = cdiv(TUH, 16);                  // urow blocks that carry data
    static constexpr int NOB = TOH / 16, NCB = TOW / 16;
    static constexpr int NDVK = DOWN == 2 ? 2 : 3;             // 32-wide K windows of a down pass
    static constexpr int NPAIR = (DOWN / 2) * (NOB - 1) + NDVK;  // packed X2 tile pairs the down-V pass touches
    static constexpr int NMB = cdiv((16 / UP) * (NVB - 1) + 16 / UP + kFUT, 16);   // X1 row blocks that carry data
    static constexpr int NQ = ((NVB - 1) / UP) + 1;            // packed X1 tile pairs (mb0 = vb / UP)
    static constexpr int IROWS = 16 * NMB;
    static constexpr int IWSTEP = GW / UP;                     // input-window advance per group
    static constexpr int TIW = IWSTEP * (NG - 1) + 32;         // staged columns; column 0 = input column I0x - (I0x & 1)
    static constexpr int PIN = ((TIW + 7) / 8) * 8 + ((((TIW + 7) / 8) % 2 == 0) ? 8 : 0);   // 16-B units, odd count
    static constexpr int XCOLS = NG * GW;                      // computed upsampled columns
    static constexpr int X3COLS = cmax(XCOLS, 16 * DOWN * (NCB - 1) + 32 * NDVK);   // the last K window of down-x overhangs
    static constexpr int PX3 = 8 * (cdiv(X3COLS, 8) | 1);      // X3 pitch (elements): an odd number of 16-B units
    static constexpr int POUT = TOW + 8;                       // staged output pitch (elements): odd number of 16-B units
    static constexpr int NFRAG = NB + UP + 3 * NDVK;           // UH[NB] UV[UP] DVs[NDVK] DVr[NDVK] DH[NDVK]
    static constexpr int SGW_ROWS = 4 * NVB;                   // staged sign quad-rows (WRITE)
    static constexpr int SGW_PITCH = XCOLS;                    // bytes; XCOLS is a multiple of 16
    static constexpr int SGN_ROWS = 4 * NVB + 1;               // staged sign quad-rows (READ)
    static constexpr int SGN_WORDS = XCOLS / 4 + 1;            // aligned dwords covering one staged sign row
    static constexpr int SGN_PITCH = 4 * SGN_WORDS + 12;
    static_assert(TOW % 16 == 0 && TOH % 16 == 0 && (TOW * DOWN) % (2 * UP) == 0 && (TOH * DOWN) % UP == 0, "tile shape");
    static_assert((TOH * DOWN) % 4 == 0 && (TOW * DOWN) % 16 == 0, "sign quads / 16-byte sign segments");
    static_assert((GW - 1) / UP + 1 + kFUT + 1 <= 32, "shifted input window must fit the 32-wide K window");
    static_assert(NFRAG * 1024 <= kWsTable, "workspace");
    static_assert((POUT / 8) % 2 == 1 && (PX3 / 8) % 2 == 1, "LDS pitches");
};

// ---------------------------------------------------------------------------------------------
// Constant fragments.  Layout: [frag][lane][8] elements of T, then (byte kWsTable) the keep-mask table.
template <typename T, int UP, int DOWN, int TOW, int TOH>
__global__ void flrelu_mfma_prepare_kernel(T* __restrict__ ws, const float* __restrict__ fu, const float* __restrict__ fd,
                                           int px0, int py0, int flip, float gain_total, float slope, int dshift) {
    // dshift (aligned READ calls of the wave kernels, else 0): the tile's upsampled rows start dshift rows EARLY -- py0 already
    // carries it for the up side (everything there derives from py0 - U0y); the down-y taps move by it here
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    const int phx = pos_mod(px0, UP), phy = pos_mod(py0, UP);
    const int odd = (-floor_div(px0, UP)) & 1;                // the staged tile starts one column early when I0x is odd
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < G::NFRAG * 64 * 8; idx += gridDim.x * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
        const int l15 = lane & 15, g = lane >> 4;
        float v = 0.f;
        if (frag < G::NB) {
            // UH[nb]: B[k][n], k = 8g + j (staged column of the window), n = l15 (ucol 16nb + n of the group)
            const int k = 8 * g + j - odd, urel = 16 * frag + l15;
            const int a = urel % UP, o = (a > phx) ? 1 : 0;
            const int kmin = o ? UP - (a - phx) : phx - a;
            const int jj = k - urel / UP - o;
            if (jj >= 0 && jj < kFUT) {
                const int tap = kmin + UP * jj;
                v = flip ? fu[tap] : fu[G::FU - 1 - tap];
            }
        } else if (frag < G::NB + UP) {
            // UV[var]: A[n][k], n = l15 (urow 16vb + n), k -> input row krow(g, j) of the 32-row window starting at 16*(vb/UP)
            const int var = frag - G::NB, n = l15;
            const int a = n % UP, o = (a > phy) ? 1 : 0;
            const int kmin = o ? UP - (a - phy) : phy - a;
            const int jj = krow(g, j) - (16 / UP) * var - n / UP - o;
            if (jj >= 0 && jj < kFUT) {
                const int tap = kmin + UP * jj;
                v = (flip ? fu[tap] : fu[G::FU - 1 - tap]) * gain_total;
            }
        } else if (frag < G::NB + UP + 2 * G::NDVK) {
            // DVs[t] / DVr[t]: [n][k], n = l15 (orow), k -> urow 32t + krow(g, j) of the window starting at 16*DOWN*ob;
            // scaled by slope (applied to X2) and by 1 - slope (applied to relu(X2))
            const int t2 = frag - G::NB - UP, t = t2 % G::NDVK;
            const int kk = 32 * t + krow(g, j) - DOWN * l15 - dshift;
            if (kk >= 0 && kk < G::FD) v = (flip ? fd[kk] : fd[G::FD - 1 - kk]) * (t2 < G::NDVK ? slope : 1.f - slope);
        } else {
            // DH[t]: [k][n], k = 8g + j natural (ucol 32t + k of the window starting at 16*DOWN*cb), n = l15 (ocol)
            const int t = frag - G::NB - UP - 2 * G::NDVK;
            const int kk = 32 * t + 8 * g + j - DOWN * l15;
            if (kk >= 0 && kk < G::FD) v = flip ? fd[kk] : fd[G::FD - 1 - kk];
        }
        ws[idx] = from_f32<T>(v);
    }
    // keep-mask table: entry c (one sign byte = 4 rows) -> two dwords of 16-bit lanes, all ones where the element is
    // neither negative nor clamped (rows 0,1 | rows 2,3)
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < 256; c += gridDim.x * blockDim.x) {
        unsigned m[2];
        for (int h = 0; h < 2; h++)
            m[h] = (((c >> (4 * h)) & 3) ? 0u : 0xffffu) | (((c >> (4 * h + 2)) & 3) ? 0u : 0xffff0000u);
        unsigned* tab = (unsigned*)((char*)ws + kWsTable);
        tab[2 * c] = m[0];
        tab[2 * c + 1] = m[1];
    }
}

// waves per SIMD the register allocator is asked for.  The 48-row tiles hold half as many accumulator rows again: their live set does
// not fit the 32-row tiles' budgets (the allocator reported 4 / 3 / 2 against 5-6 / 4 / 4 asked, and said so 14 times per build:
// VERDICT r05 #9) -- ask for what such a tile can have
template <int DOWN, int TOH, int SIGN>
constexpr int mfma_tile_occupancy() {
    if (TOH > 32) return DOWN == 2 ? 4 : (SIGN == AFCM_SIGNS_NONE ? 2 : 3);
    return DOWN == 2 ? (SIGN == AFCM_SIGNS_WRITE ? 5 : 6) : 4;
}
template <typename T, int UP, int DOWN, int TOW, int TOH, int SIGN, bool BIAS>
__global__ __launch_bounds__((64 * MfmaGeom<UP, DOWN, TOW, TOH>::NG), (mfma_tile_occupancy<DOWN, TOH, SIGN>())) void flrelu_mfma_kernel(FlreluMfmaParams p) {
    // second bound = waves per SIMD: the 3-wave workgroups (19 KB of LDS) fit 8 to a CU, which needs <= 80 VGPRs -- without
    // the bound the scheduler spends ~90 on overlapping the tiles' MFMAs and two workgroups per CU are lost
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    typedef MfmaOps<T> M;
    typedef typename M::frag frag;
    constexpr int NT = 64 * G::NG;
    // LDS: ONE region (lds_a) is the input tile first, then -- once every wave holds its A fragments -- the sign staging
    // (+ keep-mask table) and the staged output tile; X3 has its own.  ~19 KB per workgroup: 8 workgroups per CU.
    constexpr int SG_BYTES = (SIGN == AFCM_SIGNS_READ) ? G::SGN_ROWS * G::SGN_PITCH + 256 * 8
                           : (SIGN == AFCM_SIGNS_WRITE) ? G::SGW_ROWS * G::SGW_PITCH : 0;
    constexpr int OUT_OFF = (SIGN == AFCM_SIGNS_WRITE) ? SG_BYTES : 0;        // WRITE: codes are copied out while phase B stages Y
    constexpr int A_BYTES = cmax(cmax(G::IROWS * G::PIN * 2, OUT_OFF + TOH * G::POUT * 2), SG_BYTES);
    static_assert(OUT_OFF % 16 == 0, "staged output alignment");
    __shared__ __attribute__((aligned(16))) unsigned char lds_a[A_BYTES];
    __shared__ __attribute__((aligned(16))) T lds_x3[TOH * G::PX3];
    __shared__ unsigned lds_flag[G::NG];
    T* const lds_in = (T*)lds_a;
    unsigned char* const lds_sg = lds_a;
    T* const lds_out = (T*)(lds_a + OUT_OFF);
    const uint2* const lds_tab = (const uint2*)(lds_sg + G::SGN_ROWS * G::SGN_PITCH);   // READ only

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
    // XCD-aware order: consecutive logical tiles (neighbours that share halos) stay on one XCD / one L2
    int bid = blockIdx.x;
    {
        const int total = gridDim.x;
        bid = xcd_order(bid, total);
    }
    // divisions by multiply-high (exact for the grid sizes the host admits): the compiler's uniform integer division is a
    // ~25-instruction float-reciprocal sequence on the VECTOR unit, three of them per wave
    const int plane = p.magicP ? (int)__umulhi((unsigned)bid, p.magicP) : bid;            // magic 0: divisor 1
    const int tile = bid - plane * (p.tilesX * p.tilesY);
    const int ty = p.magicT ? (int)__umulhi((unsigned)tile, p.magicT) : tile;
    const int tx = tile - ty * p.tilesX;
    const int O0x = tx * TOW, O0y = ty * TOH;
    const int U0x = O0x * DOWN, U0y = O0y * DOWN;
    const int I0x = -floor_div(p.px0 - U0x, UP), I0y = -floor_div(p.py0 - U0y, UP);
    const int S0x = I0x - (I0x & 1);                            // first staged column (even: aligned dword pairs)
    const bool lastX = (tx == p.tilesX - 1), lastY = (ty == p.tilesY - 1);
    bool has_clamp = false;                                     // READ: some staged code carries the clamp bit
    constexpr int NSW = (SIGN == AFCM_SIGNS_READ) ? cdiv(G::SGN_ROWS * G::SGN_WORDS, NT) : 1;
    unsigned sv[NSW];                                           // READ: this thread's words of the sign window
    uint4 tabv = make_uint4(0, 0, 0, 0);                        // READ: its piece of the keep-mask table
    // Constant (Toeplitz) fragments: fetched HERE, ahead of the input tile, so that their L2 round trip overlaps the staging
    // loads -- the barriers below are fences, and issued behind them every fragment load was waited for on the spot (one
    // exposed round trip per tile for the up/down-y set, one per column block for UH, one for DH).
    const frag* wsf = (const frag*)p.ws;
    auto cfrag = [&](int f) __attribute__((always_inline)) { return wsf[f * 64 + lane]; };
#ifndef AFCM_FL_PF_TOP
#define AFCM_FL_PF_TOP (SIGN != AFCM_SIGNS_NONE)
#endif
    // (the sign-writing forward kernel is register-bound: at 6 waves per SIMD / 80 VGPRs holding these across the staging phase
    // costs more than the round trip saves, 1630 vs 1710 GB/s; bounded to 5 waves / 96 VGPRs it gains, 1750.  The backward
    // kernel gains at 6 waves, 1712 vs 1602, and loses at 5.)
    constexpr bool PF_TOP = AFCM_FL_PF_TOP;
    frag uv[UP], dvs[G::NDVK], dvr[G::NDVK];
    auto load_const = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int v = 0; v < UP; v++) uv[v] = cfrag(G::NB + v);
#pragma unroll
        for (int t = 0; t < G::NDVK; t++) {
            dvs[t] = cfrag(G::NB + UP + t);
            dvr[t] = cfrag(G::NB + UP + G::NDVK + t);
        }
    };
    if (PF_TOP) load_const();
    frag uh_next;
    if (PF_TOP) uh_next = cfrag(0);

    // ---- stage the input tile: zero outside the image (the bias is added before padding).
    // One item = 8 consecutive columns of one row = one 16-byte load from a 4-byte aligned address.  Rows outside the
    // plane fall outside the buffer descriptor and read as zero; columns outside it are masked in the edge tiles.
    {
        constexpr int CH = G::PIN / 8;                       // 8-column chunks per staged row (incl. pitch padding)
        constexpr int NIT = cdiv(G::IROWS * CH, NT);
        const T* xp = (const T*)p.x + (size_t)plane * p.xh * p.xw;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, p.xh * p.xw * 2, 0x00020000);
        const bool xedge = S0x < 0 || S0x + 8 * CH > p.xw;
        u32x4 raw[NIT];
        if (S0x < 0 && I0y <= 0) {
            // the tile holds the plane's first row with columns left of it: byte offsets would go negative, so this one
            // tile per plane fetches dword by dword with explicit column checks
#pragma unroll
            for (int i = 0; i < NIT; i++) {
                const int idx = tid + i * NT;
                const int r = idx / CH, c8 = idx - r * CH;
                const int iy = I0y + r;
                const int ix0 = S0x + 8 * c8;
                const bool rok = (NIT * NT == G::IROWS * CH || idx < G::IROWS * CH) && (unsigned)iy < (unsigned)p.xh;
#pragma unroll
                for (int w = 0; w < 4; w++) {
                    const bool ok = rok && (unsigned)(ix0 + 2 * w) < (unsigned)p.xw;
                    raw[i][w] = __builtin_amdgcn_raw_buffer_load_b32(rs, ok ? (unsigned)((iy * p.xw + ix0 + 2 * w) * 2) : 0x80000000u, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < NIT; i++) {
                const int idx = tid + i * NT;
                const int r = idx / CH, c8 = idx - r * CH;
                const int iy = I0y + r;
                const int ix0 = S0x + 8 * c8;
                const bool rok = (NIT * NT == G::IROWS * CH || idx < G::IROWS * CH) && (unsigned)iy < (unsigned)p.xh;
                const unsigned off = rok ? (unsigned)((iy * p.xw + ix0) * 2) : 0x80000000u;
                raw[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            }
        }
        if (G::X3COLS > G::XCOLS) {
            // columns of X3 beyond the computed ones are read (with zero weights) by the last down-x window: keep them finite
            constexpr int ZC = (G::X3COLS - G::XCOLS) / 8;
            for (int idx = tid; idx < TOH * ZC; idx += NT)
                *(uint4*)(lds_x3 + (idx / ZC) * G::PX3 + G::XCOLS + 8 * (idx % ZC)) = make_uint4(0, 0, 0, 0);
        }
        if (SIGN == AFCM_SIGNS_READ) {
            // sign window: quad-rows [(U0y+sy)>>2, +SGN_ROWS), columns [U0x+sx, +XCOLS), fetched as aligned dwords
            const unsigned char* sp = p.s + (size_t)plane * p.shq * p.swq;
            const int qy0 = (U0y + p.sy) >> 2, w0 = (U0x + p.sx) >> 2, wpr = p.swq >> 2;
            unsigned any = 0;
            if (qy0 >= 0 && qy0 + G::SGN_ROWS <= p.shq && w0 >= 0 && w0 + G::SGN_WORDS <= wpr) {
                // the whole window lies inside the sign tensor (every tile but the border ones): no per-word predicates
                const unsigned* sbase = (const unsigned*)(sp + (size_t)qy0 * p.swq) + w0;
#pragma unroll
                for (int i = 0; i < NSW; i++) {
                    const int idx = min(tid + i * NT, G::SGN_ROWS * G::SGN_WORDS - 1);
                    const int r = idx / G::SGN_WORDS, c = idx - r * G::SGN_WORDS;
                    sv[i] = sbase[r * wpr + c];
                }
            } else {
#pragma unroll
                for (int i = 0; i < NSW; i++) {
                    const int idx = tid + i * NT;
                    const int r = idx / G::SGN_WORDS, c = idx - r * G::SGN_WORDS;
                    const int qy = qy0 + r, wi = w0 + c;
                    sv[i] = (idx < G::SGN_ROWS * G::SGN_WORDS && (unsigned)qy < (unsigned)p.shq && (unsigned)wi < (unsigned)wpr)
                                ? ((const unsigned*)(sp + (size_t)qy * p.swq))[wi] : 0u;
                }
            }
            // keep-mask table: 2 KB from the workspace.  Both stay in registers until the input tile has been consumed.
            if (tid < 128) tabv = ((const uint4*)((const char*)p.ws + kWsTable))[tid];
#pragma unroll
            for (int i = 0; i < NSW; i++) any |= sv[i];
            const bool wc = __builtin_amdgcn_ballot_w64((any & 0xaaaaaaaau) != 0) != 0;
            if (lane == 0) lds_flag[wave] = wc ? 1u : 0u;
        }
        float bias = 0.f;
        if (BIAS) bias = p.b ? to_f32(((const T*)p.b)[plane % p.C]) : 0.f;
#pragma unroll
        for (int i = 0; i < NIT; i++) {
            const int idx = tid + i * NT;
            if (NIT * NT == G::IROWS * CH || idx < G::IROWS * CH) {
                const int r = idx / CH, c8 = idx - r * CH;
                const int ix0 = S0x + 8 * c8;
                u32x4 v = raw[i];
                if (BIAS) {
                    const bool rok = (unsigned)(I0y + r) < (unsigned)p.xh;
#pragma unroll
                    for (int w = 0; w < 4; w++) {
                        union { unsigned u; T t[2]; } e;
                        e.u = v[w];
                        const bool in = rok && (unsigned)(ix0 + 2 * w) < (unsigned)p.xw;       // even width: pairs are in or out together
                        v[w] = in ? pack2<T>(to_f32(e.t[0]) + bias, to_f32(e.t[1]) + bias) : 0u;
                    }
                } else if (xedge) {
#pragma unroll
                    for (int w = 0; w < 4; w++)
                        if ((unsigned)(ix0 + 2 * w) >= (unsigned)p.xw) v[w] = 0u;
                }
                *(u32x4*)(lds_in + r * G::PIN + 8 * c8) = v;
            }
        }
    }
    __syncthreads();
    // every wave takes its A fragments (input rows x 32-column window) into registers; then lds_a changes hands
    frag a_in[G::NMB];
    {
        const int Gi = wave;
#pragma unroll
        for (int mb = 0; mb < G::NMB; mb++) {
            const T* src = lds_in + (16 * mb + l15) * G::PIN + G::IWSTEP * Gi + 8 * g;
            union { uint2 h[2]; uint4 q; frag f; } t;
            if ((G::IWSTEP * 2) % 16 == 0) {
                t.q = *(const uint4*)src;
            } else {                                            // 8-byte aligned halves (window start is 8-B aligned for UP=4)
                t.h[0] = *(const uint2*)src;
                t.h[1] = *(const uint2*)(src + 4);
            }
            a_in[mb] = t.f;
        }
    }
    if (SIGN != AFCM_SIGNS_NONE) __syncthreads();
    if (SIGN == AFCM_SIGNS_READ) {
#pragma unroll
        for (int i = 0; i < NSW; i++) {
            const int idx = tid + i * NT;
            if (idx < G::SGN_ROWS * G::SGN_WORDS) {
                const int r = idx / G::SGN_WORDS, c = idx - r * G::SGN_WORDS;
                *(unsigned*)(lds_sg + r * G::SGN_PITCH + 4 * c) = sv[i];
            }
        }
        if (tid < 128) ((uint4*)lds_tab)[tid] = tabv;
        unsigned f = 0;
#pragma unroll
        for (int w = 0; w < G::NG; w++) f |= lds_flag[w];
        has_clamp = __builtin_amdgcn_readfirstlane(f) != 0;
        __syncthreads();
    }

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- phase A: one group of NB ucol blocks per wave: up-x, up-y, activation, down-y, all in registers
    {
        const int Gi = wave;
        if (!PF_TOP) load_const();
        // READ: row offset of the sign window inside its first staged quad / column offset inside its first aligned dword
        const int yy = (U0y + p.sy) & 3, coff = (U0x + p.sx) & 3;
        const float cthr = p.clamp / fmaxf(p.slope, 1.f);      // |lrelu(v)| <= max(1, slope) |v|: below cthr nothing clamps
        // per-lane LDS bases, made opaque so they stay in one register each instead of being recomputed per tile
        unsigned sgw_off = g * G::SGW_PITCH + G::GW * Gi + l15;                  // WRITE: + 4 vb rows + 16 nb
        unsigned sgr_off = g * G::SGN_PITCH + G::GW * Gi + l15 + coff;           // READ
        unsigned x3w_off = l15 * G::PX3 + G::GW * Gi + 4 * g;                    // elements; + 16 ob rows + 16 nb
        asm volatile("" : "+v"(sgw_off), "+v"(sgr_off), "+v"(x3w_off));
        unsigned char* const sgw = lds_sg + sgw_off;
        const unsigned char* const sgr = lds_sg + sgr_off;
        T* const x3w = lds_x3 + x3w_off;

#pragma unroll
        for (int nb = 0; nb < G::NB; nb++) {
            const frag uh = PF_TOP ? uh_next : cfrag(nb);
            if (PF_TOP && nb + 1 < G::NB) uh_next = cfrag(nb + 1);                 // one column block ahead
            // up-x: X1[mb] = In[mb] * UH
            f32x4 x1[G::NMB];
#pragma unroll
            for (int mb = 0; mb < G::NMB; mb++) x1[mb] = M::mma(a_in[mb], uh, zero4);
            frag q[G::NQ];
#pragma unroll
            for (int m = 0; m < G::NQ; m++) q[m] = pack_pair<T>(x1[m], (m + 1 < G::NMB) ? x1[m + 1] : zero4);
            // up-y, then the two operands of the down-y pass: X2 itself and relu(X2) (forward) / keep-mask & X2 (backward).
            // Forward: the whole column block first runs the fast path (no clamp) while the largest |X2| is tracked with two
            // v_maximum3 per tile; one wave-uniform test per block decides whether it is redone exactly (rare: |v| > clamp).
            u32x4 pv[G::NPAIR], rv[G::NPAIR];
            auto run_block = [&](auto exact_c) __attribute__((always_inline)) -> float {
                constexpr bool EXACT = decltype(exact_c)::value;
                float amax = 0.f;
#pragma unroll
                for (int vb = 0; vb < 2 * G::NPAIR; vb++) {
                    unsigned d0 = 0, d1 = 0, r0 = 0, r1 = 0;
                    if (vb < G::NVB) {
                        f32x4 x2 = M::mma(uv[vb % UP], q[vb / UP], zero4);
                        if (SIGN == AFCM_SIGNS_READ) {
                            // codes of rows Y+sy .. Y+sy+3 at column X+sx: two staged quad bytes, funnel-shifted
                            const unsigned lo = sgr[(4 * vb) * G::SGN_PITCH + 16 * nb];
                            const unsigned hi = sgr[(4 * vb + 1) * G::SGN_PITCH + 16 * nb];
                            const unsigned codes = __builtin_amdgcn_ubfe(lo | (hi << 8), 2 * yy, 8);
                            if (__builtin_expect(!has_clamp, 1)) {
                                const uint2 keep = lds_tab[codes];
                                d0 = pack2<T>(x2[0], x2[1]);
                                d1 = pack2<T>(x2[2], x2[3]);
                                r0 = d0 & keep.x;
                                r1 = d1 & keep.y;
                            } else {
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    const unsigned c = codes >> (2 * r);
                                    float v = x2[r];
                                    if (c & 1u) v *= p.slope;
                                    if (c & 2u) v = 0.f;
                                    x2[r] = v;
                                }
                                d0 = r0 = pack2<T>(x2[0], x2[1]);     // slope*v + (1-slope)*v
                                d1 = r1 = pack2<T>(x2[2], x2[3]);
                            }
                        } else {
                            unsigned wcode;
                            if (!EXACT) {
                                // NaN-propagating maximum (fmaxf would add a canonicalisation per operand)
                                amax = __builtin_elementwise_maximum(amax, __builtin_elementwise_maximum(__builtin_fabsf(x2[0]), __builtin_fabsf(x2[1])));
                                amax = __builtin_elementwise_maximum(amax, __builtin_elementwise_maximum(__builtin_fabsf(x2[2]), __builtin_fabsf(x2[3])));
                                d0 = pack2<T>(x2[0], x2[1]);
                                d1 = pack2<T>(x2[2], x2[3]);
                                r0 = relu_pk(d0);
                                r1 = relu_pk(d1);
                                const unsigned f = (signs_pk(d1) << 4) | signs_pk(d0);       // bits 0 (row 0), 16 (row 1), 4 (row 2), 20 (row 3)
                                wcode = f | (f >> 14);
                            } else {
                                wcode = 0;
#pragma unroll
                                for (int r = 0; r < 4; r++) {
                                    float v = x2[r];
                                    unsigned c = __float_as_uint(v) >> 31;
                                    if (c) v *= p.slope;
                                    if (fabsf(v) > p.clamp) { c = 2u; v = (v < 0.f) ? -p.clamp : p.clamp; }
                                    wcode |= c << (2 * r);
                                    x2[r] = v;
                                }
                                d0 = r0 = pack2<T>(x2[0], x2[1]);
                                d1 = r1 = pack2<T>(x2[2], x2[3]);
                            }
                            if (SIGN == AFCM_SIGNS_WRITE) sgw[(4 * vb) * G::SGW_PITCH + 16 * nb] = (unsigned char)wcode;
                        }
                    }
                    pv[vb >> 1][2 * (vb & 1)] = d0;
                    pv[vb >> 1][2 * (vb & 1) + 1] = d1;
                    rv[vb >> 1][2 * (vb & 1)] = r0;
                    rv[vb >> 1][2 * (vb & 1) + 1] = r1;
                }
                return amax;
            };
            const float amax_nb = run_block(std::false_type{});
            if (SIGN != AFCM_SIGNS_READ) {
                if (__builtin_expect(__builtin_amdgcn_ballot_w64(!(amax_nb <= cthr)) != 0, 0))    // NaN takes the exact path
                    run_block(std::true_type{});
            }
            // down-y, transposed: X3'[ucol][orow] = sum_t P[(DOWN/2)*ob + t]' * DV[t]'  ->  LDS X3[orow][ucol], 4 columns per lane
#pragma unroll
            for (int ob = 0; ob < G::NOB; ob++) {
                f32x4 x3 = zero4;
#pragma unroll
                for (int t = 0; t < G::NDVK; t++) {
                    x3 = M::mma(as_frag<frag>(pv[(DOWN / 2) * ob + t]), dvs[t], x3);
                    x3 = M::mma(as_frag<frag>(rv[(DOWN / 2) * ob + t]), dvr[t], x3);
                }
                uint2 w;
                w.x = pack2<T>(x3[0], x3[1]);
                w.y = pack2<T>(x3[2], x3[3]);
                *(uint2*)(x3w + (16 * ob) * G::PX3 + 16 * nb) = w;
            }
        }
    }
    frag dh[G::NDVK];                                            // phase B's fragments, requested before the barrier
    if (PF_TOP) {
#pragma unroll
        for (int t = 0; t < G::NDVK; t++) dh[t] = cfrag(G::NB + UP + 2 * G::NDVK + t);
    }
    __syncthreads();

    // ---- sign codes of the region this tile owns: LDS -> HBM in 16-byte row segments
    if (SIGN == AFCM_SIGNS_WRITE) {
        const int q0 = U0y >> 2;
        const int rows = lastY ? min(G::SGW_ROWS, p.shq - q0) : (TOH * DOWN) / 4;
        const int segs = (lastX ? min(G::XCOLS, p.swq - U0x) : TOW * DOWN) >> 4;
        unsigned char* sg = p.s + ((size_t)plane * p.shq + q0) * p.swq + U0x;
        if (!lastX && !lastY) {
            constexpr int SEGS = (TOW * DOWN) / 16, ROWS = (TOH * DOWN) / 4;     // compile-time index split for the common case
#pragma unroll
            for (int i = 0; i < cdiv(ROWS * SEGS, NT); i++) {
                const int idx = tid + i * NT;
                const int r = idx / SEGS, c = idx - r * SEGS;
                if ((ROWS * SEGS) % NT == 0 || idx < ROWS * SEGS)
                    *(uint4*)(sg + (size_t)r * p.swq + 16 * c) = *(const uint4*)(lds_sg + r * G::SGW_PITCH + 16 * c);
            }
        } else {
            for (int idx = tid; idx < rows * segs; idx += NT) {
                const int r = idx / segs, c = idx - r * segs;
                *(uint4*)(sg + (size_t)r * p.swq + 16 * c) = *(const uint4*)(lds_sg + r * G::SGW_PITCH + 16 * c);
            }
        }
    }

    // ---- phase B: down-x, transposed (each lane ends up with 4 consecutive output columns of one row)
    {
        if (!PF_TOP) {
#pragma unroll
            for (int t = 0; t < G::NDVK; t++) dh[t] = cfrag(G::NB + UP + 2 * G::NDVK + t);
        }
        const bool inner = (O0x + TOW <= p.yw) && (O0y + TOH <= p.yh);
        const float osc = (p.oscale ? p.oscale[plane] : 1.f) * (p.oscale2 ? p.oscale2[plane] : 1.f);
        const T* skp = p.skip ? (const T*)p.skip + (size_t)plane * p.yh * p.yw : nullptr;
        float psum = 0.f;
        unsigned x3r_off = l15 * G::PX3 + 8 * g, outw_off = l15 * G::POUT + 4 * g;      // per-lane bases, one register each
        asm volatile("" : "+v"(x3r_off), "+v"(outw_off));
        const T* const x3r = lds_x3 + x3r_off;
        T* const outw = lds_out + outw_off;
        constexpr int NUNIT = G::NOB * G::NCB, UPW = cdiv(NUNIT, G::NG);
#pragma unroll
        for (int ui = 0; ui < UPW; ui++) {
            const int unit = wave + ui * G::NG;                       // wave-uniform
            if (NUNIT % G::NG != 0 && unit >= NUNIT) break;
            const int ob = unit / G::NCB, cb = unit - ob * G::NCB;
            f32x4 acc = zero4;
#pragma unroll
            for (int t = 0; t < G::NDVK; t++) {
                union { uint4 q; frag f; } b;
                b.q = *(const uint4*)(x3r + (16 * ob) * G::PX3 + 16 * DOWN * cb + 32 * t);
                acc = M::mma(dh[t], b.f, acc);
            }
            if (skp != nullptr) {
                // encoder feature of this lane's 4 columns (x + x_skip, NET:376-377); pairs are in or out together (even width)
                const int oy = O0y + 16 * ob + l15, ox = O0x + 16 * cb + 4 * g;
                if (oy < p.yh) {
                    const unsigned* sp2 = (const unsigned*)(skp + (size_t)oy * p.yw + ox);
#pragma unroll
                    for (int w = 0; w < 2; w++)
                        if (ox + 2 * w < p.yw) {
                            union { unsigned u; T t[2]; } e;
                            e.u = sp2[w];
                            acc[2 * w] += to_f32(e.t[0]);
                            acc[2 * w + 1] += to_f32(e.t[1]);
                        }
                }
            }
            if (p.oscale || p.oscale2) acc *= osc;
            uint2 w;
            w.x = pack2<T>(acc[0], acc[1]);
            w.y = pack2<T>(acc[2], acc[3]);
            *(uint2*)(outw + (16 * ob) * G::POUT + 16 * cb) = w;
            if (p.plane_sum != nullptr) {
                if (inner) {
                    psum += (acc[0] + acc[1]) + (acc[2] + acc[3]);
                } else if (O0y + 16 * ob + l15 < p.yh) {
                    const int c0 = O0x + 16 * cb + 4 * g;
#pragma unroll
                    for (int r = 0; r < 4; r++)
                        if (c0 + r < p.yw) psum += acc[r];
                }
            }
        }
        __syncthreads();
        // staged output tile -> HBM, 8 columns (16 bytes) per lane
        {
            constexpr int SEG = TOW / 8;
            T* yp = (T*)p.y + (size_t)plane * p.yh * p.yw;
            if (inner) {
                T* ytile = yp + (size_t)O0y * p.yw + O0x;
#pragma unroll
                for (int i = 0; i < cdiv(TOH * SEG, NT); i++) {
                    const int idx = tid + i * NT;
                    const int r = idx / SEG, c = idx - r * SEG;
                    if ((TOH * SEG) % NT == 0 || idx < TOH * SEG)
                        *(u32x4*)(ytile + (size_t)r * p.yw + 8 * c) = *(const u32x4*)(lds_out + r * G::POUT + 8 * c);
                }
            } else
            for (int idx = tid; idx < TOH * SEG; idx += NT) {
                const int r = idx / SEG, c = idx - r * SEG;
                const int oy = O0y + r, ox = O0x + 8 * c;
                if (oy < p.yh && ox < p.yw) {
                    const u32x4 v = *(const u32x4*)(lds_out + r * G::POUT + 8 * c);
                    T* dst = yp + (size_t)oy * p.yw + ox;
                    if (ox + 8 <= p.yw) {
                        *(u32x4*)dst = v;
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; w++)
                            if (ox + 2 * w < p.yw) ((unsigned*)dst)[w] = v[w];     // even width: pairs are in or out together
                    }
                }
            }
        }
        if (p.plane_sum != nullptr) {
            // wave reduction -> workgroup reduction through LDS -> one plain store into this tile's slot (no atomics:
            // deterministic, and no same-address contention between the tiles of a plane)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) psum += __shfl_down(psum, off, 64);
            float* red = (float*)lds_x3;                       // phase-B reads of lds_x3 finished before the barrier above
            if (lane == 0) red[wave] = psum;
            __syncthreads();
            if (tid == 0) {
                float t = 0.f;
#pragma unroll
                for (int w = 0; w < G::NG; w++) t += red[w];
                p.plane_sum[(size_t)plane * (p.tilesX * p.tilesY) + ty * p.tilesX + tx] = t;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
template <int UP, int DOWN> struct MfmaTile;
template <> struct MfmaTile<2, 2> { static constexpr int TOW = 64, TOH = 32; };
template <> struct MfmaTile<2, 4> { static constexpr int TOW = 32, TOH = 32; };
template <> struct MfmaTile<4, 2> { static constexpr int TOW = 64, TOH = 32; };
// Tall variant for planes of 33..48 output rows (the 36^2 / 38^2 planes of the 256^2 generator): ONE 48-row tile instead of two
// 32-row tiles that are 12 % full in their second row.  The constant fragments do not depend on the tile shape (only on up,
// down and the filters), so both variants share one prepared workspace; the sign layout is tile-independent.
constexpr int kTallTOH = 48;

// Kernel family of a call.  The wave-autonomous kernels (filtered_lrelu_wave.hip) take every matrix-core case without a bias
// operand; they write / read sign layout 2, the LDS-tile kernels below layout 1, so a READ call follows the layout of its tensor.
template <typename T, int UP, int DOWN, int TOW, int TOH>
int launch_wave_tile(const afcm_filtered_lrelu_args* a, FlreluMfmaParams p, hipStream_t st);
template <typename T, int UP, int DOWN>
int prepare_wave(const afcm_filtered_lrelu_args* a, int py0_frag, int dshift, hipStream_t st);

constexpr int kWavePitchSlack = 128;        // elements a row pitch may exceed the plane width by (wave kernels)
static bool wave_family(const afcm_filtered_lrelu_args* a) {
    if (a->sign_mode == AFCM_SIGNS_READ) return a->sign_layout == 2;
    // no bias operand; offsets + out-of-range markers stay below 2^31.  Decided on the plane sizes plus the largest pitch overhead
    // launch_wave() accepts -- NOT on the pitches themselves: afcm_filtered_lrelu_shapes() runs before the caller has chosen them,
    // and the sign layout / strip geometry it reports must be the one the launch uses.
    return a->b == nullptr && (long long)a->xh * (a->xw + kWavePitchSlack) < (1ll << 28) &&
           (long long)a->yh * (a->yw + kWavePitchSlack) < (1ll << 28);
}

// Output rows per strip of the wave kernels: 32; one 48-row strip for the 36^2 / 38^2 planes (up 2 / down 2).  (Measured and
// dropped: 16-row strips for down 4, whose 32-row strips need 240-250 registers = two waves per SIMD: at 16 rows a strip still
// needs 176-199 and computes 1.5x instead of 1.25x its own rows -- forward 1.63 vs 1.89 TB/s over the down-4 layers.)
static int wave_toh(int up, int down, int rows) {
    return (up == 2 && down == 2 && rows > 32 && rows <= kTallTOH) ? kTallTOH : 32;
}

// READ calls of the wave kernels: output rows by which the strips' origin moves up (<= 0) so that every strip's first upsampled
// row, U0y + sy = (ty TOH + oy0) down + sy, is a multiple of 16 = a row block of the sign tensor (kSignsReadAligned in
// filtered_lrelu_wave.hip).  Possible when sy is a multiple of gcd(down, 16) = down; costs at most 16 / down - 1 extra rows on
// top of the plane.  Pure host arithmetic on pitch-independent arguments: shapes() and the launch agree.
// ... and, where sy is not a multiple of `down`, the remaining dshift = (oy0 down + sy) mod 16 < down upsampled rows by which the
// strips' upsampled grid itself starts early: the constant fragments of such a call are prepared with their rows moved by dshift
// (the tiles have 6-12 spare rows: (TOH - 1) down + taps + dshift <= 16 NVB for every shape), so EVERY read call is aligned.
static bool wave_read_origin(const afcm_filtered_lrelu_args* a, int* oy0, int* dshift) {
    *oy0 = 0;
    *dshift = 0;
    if (a->sign_mode != AFCM_SIGNS_READ || a->sign_layout != 2) return false;
    const int m = pos_mod(a->sy, 16);
    *oy0 = -(m / a->down);
    *dshift = m % a->down;
    return true;
}
// rows the strips of a wave launch have to cover
static int wave_rows(const afcm_filtered_lrelu_args* a) {
    int oy0, dshift;
    wave_read_origin(a, &oy0, &dshift);
    return a->yh - oy0;
}

static bool tall_tile(int up, int down, int yh, int sign_mode, bool wave) {
    if (wave) return wave_toh(up, down, yh) == kTallTOH;                    // (callers pass wave_rows())
    (void)down;
    if (up != 2) return false;
    if (yh > 32 && yh <= kTallTOH) return true;
    // The sign-WRITING kernels (forward) also gain on larger planes whenever 48-row tiles cover the plane with no more padded rows
    // than 32-row tiles (276 rows: 6 x 48 = 9 x 32; 84 rows: 2 x 48 = 3 x 32): 7 % fewer halo rows, a third fewer workgroups --
    // enc0..3 forward 0.207 / 0.277 / 0.383 -> 0.181 / 0.250 / 0.337 ms.  The sign-READING kernels lose 5-20 % on the same tiles
    // (their staged sign window and keep-mask table scale with the tile), so the transposed op keeps 32 rows.
    // up to 7 % more padded rows still pay (the 532- and 512-row planes of the 512^2 generator: 576 vs 544, 528 vs 512 rows --
    // filtered_lrelu 10.0 -> 9.8 ms per step there); at 12.5 % (256 rows) the gain is gone
    constexpr int slack = 7;                                                   // extra padded rows tolerated, in percent
    if (sign_mode != AFCM_SIGNS_READ && 100 * cdiv(yh, kTallTOH) * kTallTOH <= (100 + slack) * cdiv(yh, 32) * 32) return true;
    return false;
}

template <typename T, int UP, int DOWN, int TOW, int TOH, int SIGN>
static void launch_one(const FlreluMfmaParams& p, bool bias, dim3 grid, dim3 block, hipStream_t st) {
    if (bias) hipLaunchKernelGGL((flrelu_mfma_kernel<T, UP, DOWN, TOW, TOH, SIGN, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((flrelu_mfma_kernel<T, UP, DOWN, TOW, TOH, SIGN, false>), grid, block, 0, st, p);
}

static int fill_params(const afcm_filtered_lrelu_args* a, FlreluMfmaParams& p, int tilesX, int tilesY) {
    p.x = a->x; p.y = a->y; p.b = a->b; p.s = a->signs; p.ws = a->workspace; p.plane_sum = a->plane_sum;
    p.oscale = a->oscale; p.oscale2 = a->oscale2; p.skip = a->skip;
    p.clamp_flags = nullptr;                    // (wave kernels only: launch_wave sets it)
    p.st_plain = 0;
    p.xw = a->xw; p.xh = a->xh; p.yw = a->yw; p.yh = a->yh; p.C = a->c;
    p.xld = a->x_pitch ? a->x_pitch : a->xw; p.yld = a->y_pitch ? a->y_pitch : a->yw; p.kld = a->skip_pitch ? a->skip_pitch : a->yw;
    p.px0 = a->px0; p.py0 = a->py0;
    p.tilesX = tilesX; p.tilesY = tilesY;
    p.slope = a->slope; p.clamp = a->clamp;
    p.sx = a->sx; p.sy = a->sy; p.shq = a->sh; p.swq = a->swb;
    const long long blocks = (long long)p.tilesX * p.tilesY * a->n * a->c;
    AFCM_REQUIRE(blocks > 0 && blocks < (1ll << 31), "filtered_lrelu: grid of %lld blocks is out of range", blocks);
    // q = umulhi(x, ceil(2^32 / d)) equals x / d whenever x * d < 2^32 (error term x * (M d - 2^32) < x d)
    const long long tpp = (long long)p.tilesX * p.tilesY;
    AFCM_REQUIRE(blocks * tpp < (1ll << 32), "filtered_lrelu: grid of %lld blocks x %lld tiles per plane is out of range", blocks, tpp);
    p.magicT = p.tilesX == 1 ? 0u : (unsigned)(((1ull << 32) + p.tilesX - 1) / p.tilesX);
    p.magicP = tpp == 1 ? 0u : (unsigned)(((1ull << 32) + tpp - 1) / tpp);
    AFCM_REQUIRE((long long)a->xh * p.xld < (1ll << 30), "filtered_lrelu: plane of %d x %d elements is out of range", a->xh, p.xld);
    p.total_tiles = (int)blocks;
    return AFCM_OK;
}

template <typename T, int UP, int DOWN, int TOW, int TOH>
static int launch_mfma_tile(const afcm_filtered_lrelu_args* a, hipStream_t st) {
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    FlreluMfmaParams p;
    const int rc = fill_params(a, p, cdiv(a->yw, TOW), cdiv(a->yh, TOH));
    if (rc != AFCM_OK) return rc;
    dim3 grid((unsigned)p.total_tiles), block(64 * G::NG);
    const bool bias = a->b != nullptr;
    switch (a->sign_mode) {
        case AFCM_SIGNS_NONE: launch_one<T, UP, DOWN, TOW, TOH, AFCM_SIGNS_NONE>(p, bias, grid, block, st); break;
        case AFCM_SIGNS_WRITE: launch_one<T, UP, DOWN, TOW, TOH, AFCM_SIGNS_WRITE>(p, bias, grid, block, st); break;
        default: launch_one<T, UP, DOWN, TOW, TOH, AFCM_SIGNS_READ>(p, bias, grid, block, st); break;
    }
    return hip_status(hipGetLastError());
}

// wave kernels: one strip of wave_toh() output rows spans the plane's width
template <typename T, int UP, int DOWN>
static int launch_wave(const afcm_filtered_lrelu_args* a, hipStream_t st) {
    AFCM_REQUIRE(a->b == nullptr, "filtered_lrelu: sign layout 2 (wave kernels) takes no bias operand");
    AFCM_REQUIRE(a->x_pitch == 0 || (a->x_pitch >= a->xw && a->x_pitch <= a->xw + kWavePitchSlack), "filtered_lrelu: x_pitch %d outside [xw, xw + %d]", a->x_pitch, kWavePitchSlack);
    AFCM_REQUIRE(a->skip_pitch == 0 || (a->skip_pitch >= a->yw && a->skip_pitch <= a->yw + kWavePitchSlack), "filtered_lrelu: skip_pitch %d outside [yw, yw + %d]", a->skip_pitch, kWavePitchSlack);
    // a pitched y is written in whole 64-column groups: the kernel covers columns < 64 * ceil(yw / 64) only, so a larger pitch would
    // leave uninitialised padding behind (the layout's contract is finite padding, include/afcm_hip.h)
    AFCM_REQUIRE(a->y_pitch == 0 || (a->y_pitch >= a->yw && a->y_pitch <= cdiv(a->yw, 64) * 64), "filtered_lrelu: y_pitch %d outside [yw, 64 * ceil(yw / 64) = %d]", a->y_pitch, cdiv(a->yw, 64) * 64);
    const int toh = wave_toh(UP, DOWN, wave_rows(a));
    FlreluMfmaParams p;
    const int rc = fill_params(a, p, 1, cdiv(wave_rows(a), toh));
    if (rc != AFCM_OK) return rc;
    int dshift;
    p.read_aligned = wave_read_origin(a, &p.oy0, &dshift) ? 1 : 0;
    p.py0 += dshift;                 // the fragments were prepared for this origin (prepare_mfma)
    p.sy -= dshift;
    p.clamp_flags = a->sign_mode == AFCM_SIGNS_READ ? nullptr : a->clamp_flags;
    // (r06) dense output rows with a partial second 64-column group, not line-aligned: see flush() in filtered_lrelu_wave.hip
    p.st_plain = (p.yld == p.yw && p.yw > 64 && ((p.yw * 2) & 127) != 0) ? 1 : 0;
    if constexpr (UP == 2 && DOWN == 2) {
        if (toh == kTallTOH) return launch_wave_tile<T, 2, 2, 64, kTallTOH>(a, p, st);
    }
    return launch_wave_tile<T, UP, DOWN, MfmaTile<UP, DOWN>::TOW, 32>(a, p, st);
}

template <typename T, int UP, int DOWN>
static int launch_mfma(const afcm_filtered_lrelu_args* a, hipStream_t st) {
    constexpr int TOW = MfmaTile<UP, DOWN>::TOW, TOH = MfmaTile<UP, DOWN>::TOH;
    if (wave_family(a)) return launch_wave<T, UP, DOWN>(a, st);
    if constexpr (UP == 2) {
        if (tall_tile(UP, DOWN, a->yh, a->sign_mode, false)) return launch_mfma_tile<T, UP, DOWN, TOW, kTallTOH>(a, st);
    }
    return launch_mfma_tile<T, UP, DOWN, TOW, TOH>(a, st);
}

template <typename T, int UP, int DOWN>
static int prepare_mfma(const afcm_filtered_lrelu_args* a, hipStream_t st) {
    constexpr int TOW = MfmaTile<UP, DOWN>::TOW, TOH = MfmaTile<UP, DOWN>::TOH;
    typedef MfmaGeom<UP, DOWN, TOW, TOH> G;
    const float gain_total = (float)a->up * (float)a->up * a->gain;
    // aligned READ calls of the wave kernels: the strips' first upsampled row is (ty TOH + oy0) DOWN - dshift, no longer a multiple
    // of UP: the up-y fragments are built for the phase of py0 measured from THAT row
    int oy0 = 0, dshift = 0;
    if (wave_family(a)) wave_read_origin(a, &oy0, &dshift);
    const int py0_frag = a->py0 + dshift - oy0 * DOWN;
    hipLaunchKernelGGL((flrelu_mfma_prepare_kernel<T, UP, DOWN, TOW, TOH>), dim3(cdiv(G::NFRAG * 512, 256)), dim3(256), 0, st,
                       (T*)a->workspace, a->fu, a->fd, a->px0, py0_frag, a->flip_filter, gain_total, a->slope, dshift);
    const int rc = hip_status(hipGetLastError());
    return rc != AFCM_OK ? rc : prepare_wave<T, UP, DOWN>(a, py0_frag, dshift, st);
}

static int mfma_case(const afcm_filtered_lrelu_args* a) {
    if (a->dtype != AFCM_BF16 && a->dtype != AFCM_F16) return 0;
    if (a->fuh != 0 || a->fdh != 0) return 0;
    if (a->xw & 1) return 0;                   // staged loads and stores move aligned 16-bit pairs: even plane widths
    const long long yw = ((long long)a->xw * a->up + a->px0 + a->px1 - (a->fuw - 1) - (a->fdw - 1) + (a->down - 1)) / a->down;
    if (yw & 1) return 0;
    if (a->up == 2 && a->down == 2 && a->fuw == 12 && a->fdw == 12) return 22;
    if (a->up == 2 && a->down == 4 && a->fuw == 12 && a->fdw == 24) return 24;
    if (a->up == 4 && a->down == 2 && a->fuw == 24 && a->fdw == 12) return 42;
    return 0;
}

int flrelu_mfma_supported(const afcm_filtered_lrelu_args* a) { return mfma_case(a) != 0; }

// sign layout a WRITE call of this configuration produces (1: row-quad bytes, 2: column-blocked row-quad bytes)
int flrelu_mfma_sign_layout(const afcm_filtered_lrelu_args* a) { return wave_family(a) ? 2 : 1; }

// row pitches (x_pitch / y_pitch / skip_pitch): the wave kernels address rows by pitch, the LDS-tile kernels take dense tensors
int flrelu_mfma_row_pitch_ok(const afcm_filtered_lrelu_args* a) { return mfma_case(a) != 0 && wave_family(a); }

int flrelu_mfma_tiles(const afcm_filtered_lrelu_args* a) {
    switch (mfma_case(a)) {
        case 22: return wave_family(a) ? cdiv(wave_rows(a), wave_toh(2, 2, wave_rows(a))) : cdiv(a->yw, MfmaTile<2, 2>::TOW) * cdiv(a->yh, tall_tile(2, 2, a->yh, a->sign_mode, false) ? kTallTOH : MfmaTile<2, 2>::TOH);
        case 24: return wave_family(a) ? cdiv(wave_rows(a), wave_toh(2, 4, wave_rows(a))) : cdiv(a->yw, MfmaTile<2, 4>::TOW) * cdiv(a->yh, tall_tile(2, 4, a->yh, a->sign_mode, false) ? kTallTOH : MfmaTile<2, 4>::TOH);
        case 42: return wave_family(a) ? cdiv(wave_rows(a), wave_toh(4, 2, wave_rows(a))) : cdiv(a->yw, MfmaTile<4, 2>::TOW) * cdiv(a->yh, MfmaTile<4, 2>::TOH);
        default: return 0;
    }
}

int flrelu_mfma_launch(const afcm_filtered_lrelu_args* a, bool prepare, hipStream_t st) {
#define AFCM_MF(T) do { switch (mfma_case(a)) { \
        case 22: return prepare ? prepare_mfma<T, 2, 2>(a, st) : launch_mfma<T, 2, 2>(a, st); \
        case 24: return prepare ? prepare_mfma<T, 2, 4>(a, st) : launch_mfma<T, 2, 4>(a, st); \
        case 42: return prepare ? prepare_mfma<T, 4, 2>(a, st) : launch_mfma<T, 4, 2>(a, st); \
        default: return AFCM_E_NOKERNEL; } } while (0)
    if (a->dtype == AFCM_BF16) AFCM_MF(bf16_t);
    else AFCM_MF(f16_t);
#undef AFCM_MF
}

}  // namespace afcm

extern "C" int64_t afcm_filtered_lrelu_workspace_bytes(void) { return afcm::kWsBytes; }
