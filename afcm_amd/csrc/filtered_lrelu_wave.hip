// filtered_lrelu on the matrix cores, wave-autonomous form (16-bit activations, no bias operand).
//
// The workgroup-tile kernels of filtered_lrelu_mfma.hip are bound by vector-instruction issue (PMC r01e: VALU 72-81 % of the
// slots, MFMA pipe 30 %, HBM traffic = algorithmic): three waves share a 64x32-output tile through LDS, and more than half of
// each wave's ~490 vector instructions are not arithmetic on the tile at all -- staging index math, LDS round trips of the input
// tile / X3 / the staged output / the staged sign codes, copy-out loops, barriers.  Here ONE wave owns a whole output tile and
// nothing is exchanged between waves:
//   * the input rows are loaded from global memory straight into MFMA A fragments (a lane's 8 consecutive columns of one row are
//     one 16-byte load; rows outside the plane fall outside the buffer descriptor and read as zero) -- no LDS staging, no barrier;
//   * X1 (up-x), X2 (up-y), X3 (down-y) never leave the register file: each accumulator tile is the next MFMA's operand with the
//     K order the accumulator layout dictates (krow()), also for the down-x pass (the LDS-tile kernels transposed X3 through LDS);
//   * the linear part of the leaky ReLU is folded into ONE composite operator: lrelu(v) = slope v + (1 - slope) relu(v), and
//         X3 = DV lrelu(X2) = (slope DV UV) X1 + ((1 - slope) DV) relu(X2)
//     -- the first term is a single banded product from X1 (already packed as an operand), so the down-y pass needs NDVK + NLT
//     instead of 2 NDVK MFMAs per block and X2 itself is never an operand;
//   * the clamp (|v| > 256: rare) is excluded per column block from max |X1| and the L1 norm of the up-y operator (two
//     v_maximum3 per X1 tile instead of two per X2 tile); only a flagged block recomputes its X2 tiles on the exact path;
//   * sign codes are written only for the region the tile OWNS (the halo blocks are the neighbours'), as bytes straight from the
//     registers, into a column-blocked row-quad layout (sign_layout 2: [plane][col / 16][quad-row][col % 16], one byte = the codes
//     of 4 rows of one column) in which a tile's stores are 256 contiguous bytes per column block and the transposed op's reads
//     need one address register: quad-row and column-block strides are compile-time offsets;
//   * outputs leave as 8-byte stores from the accumulators (4 consecutive columns per lane).
// Per 64x32-output tile: ~600 vector + 142 matrix instructions on one wave instead of 3 x (~490 + 53).
#include "flrelu_mfma_common.h"

// waves per SIMD the register allocation targets (tuning aids).  Down 2: three (168 registers), except the sign-writing 48-row strips
// (the 36^2 planes): at 168 registers they spill ~70-100 registers to scratch -- encoder_12 forward 0.075 ms, 0.034 ms with two
// waves' budget; the sign-reading 48-row kernel spills 20 and is still 10 % faster at three.
#ifndef AFCM_WAVE_OCC_D2
#define AFCM_WAVE_OCC_D2 3
#endif
#ifndef AFCM_WAVE_OCC_D4
#define AFCM_WAVE_OCC_D4 2
#endif
// cache-policy bits of the output / sign stores (buffer aux: 1 = sc0, 2 = nt, 16 = sc1).  nt: the outputs and codes are written
// once and read by a later kernel; as plain (write-back allocating) stores they slowed the loads queued behind them --
// encoder_1 forward 0.190 -> 0.145 ms with nt
#ifndef AFCM_WAVE_STORE_AUX
#define AFCM_WAVE_STORE_AUX 2
#endif
// ... of the input loads
#ifndef AFCM_WAVE_LOAD_AUX
#define AFCM_WAVE_LOAD_AUX 0
#endif
// ... of the sign-code loads of the transposed op (read once)
#ifndef AFCM_WAVE_SIGNLOAD_AUX
#define AFCM_WAVE_SIGNLOAD_AUX 0
#endif

#if defined(AFCM_WAVE_STAMPS) && !defined(AFCM_WAVE_F16)     // (the bf16 translation unit only: one definition of the symbols)
#define AFCM_WAVE_STAMPS_ON 1
// diagnostic build only (VERDICT r05 #2d): shader-clock and real-time stamps around every strip, summed per kernel variant
// slot = (UP == 4) | (DOWN == 4) << 1 | (sign mode 0..3) << 2 | (TOH == 48) << 4; per slot: {sum of s_memtime deltas (shader cycles), sum of
// s_memrealtime deltas (100 MHz ticks), strips}.  In-loop clock of a variant = cycles / ticks x 100 MHz (MI355X_MICROARCH.md, DVFS item 6).
// The stamps go to a buffer of their own that nothing else reads; no output depends on them.
__device__ unsigned long long afcm_wave_stamp_acc[32][4];
extern "C" int afcm_debug_wave_stamps(void* dst) { return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(afcm_wave_stamp_acc), sizeof(unsigned long long) * 32 * 4); }
extern "C" int afcm_debug_wave_stamps_clear() {
    static unsigned long long zero[32][4] = {};
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(afcm_wave_stamp_acc), zero, sizeof(zero));
}
#endif

namespace afcm {

template <int UP, int DOWN, int TOW, int TOH>
struct WaveGeom {
    static constexpr int FU = kFUT * UP, FD = kFUT * DOWN;
    static constexpr int TUW = (TOW - 1) * DOWN + FD, TUH = (TOH - 1) * DOWN + FD;
    static constexpr int NBG = 2;                               // ucol blocks per group: one 32-wide input window, one packed X3 pair
    static constexpr int NVB = cdiv(TUH, 16);                   // urow blocks
    static constexpr int NOB = TOH / 16, NCB = TOW / 16;
    static constexpr int NDVK = DOWN == 2 ? 2 : 3;              // 32-wide K windows of a down pass
    static constexpr int NPAIR = (DOWN / 2) * (NOB - 1) + NDVK; // packed X2 tile pairs the down-y pass touches
    static constexpr int NMB = cdiv((16 / UP) * (NVB - 1) + 16 / UP + kFUT, 16);   // input row blocks
    static constexpr int NQ = ((NVB - 1) / UP) + 1;             // packed X1 tile pairs
    static constexpr int IWSTEP = 16 * NBG / UP;                // input-window advance per group
    static constexpr int NHIST = NDVK - 1;                      // earlier X3 pairs the down-x pass of a column block reads
    static constexpr int OWN_VB = TOH * DOWN / 16;             // urow blocks whose codes this strip writes
    // composite linear operator (slope DV UV): in-row offsets of an orow block inside its packed X1 pair, K windows
    static constexpr int NLV = (16 * DOWN / UP) % 16 == 0 ? 1 : 2;                  // distinct offsets (0, 8)
    static constexpr int LIN_DI = (15 * DOWN + FD - 1) / UP + 1 + kFUT;            // in-rows (from the block's first) that contribute
    static constexpr int NLT = cdiv(LIN_DI + 8 * (NLV - 1), 32);
    static constexpr int NLIN = NLV * NLT;
    static __host__ __device__ constexpr int lin_q(int ob, int t) { return (16 * DOWN * ob / UP) / 16 + 2 * t; }
    static __host__ __device__ constexpr int lin_f(int ob, int t) { return (((16 * DOWN * ob / UP) % 16) / 8) * NLT + t; }
    static_assert(TOW % 16 == 0 && TOH % 16 == 0 && (TOH * DOWN) % 16 == 0 && (TOW * DOWN) % 16 == 0, "tile shape");
    static_assert((16 * NBG - 1) / UP + 1 + kFUT + 1 <= 32, "shifted input window must fit the 32-wide K window");
    static_assert(lin_q(NOB - 1, NLT - 1) < NQ, "composite operator reaches beyond the packed X1 pairs");
    static_assert(NLIN + NDVK <= kWsWaveFrags, "workspace");
    static_assert((16 * DOWN) % UP == 0, "orow blocks start on input rows");
    // aligned READ calls start the upsampled rows up to DOWN - 1 rows early (dshift): the spare rows of the last blocks / K windows
    static_assert(TUH + DOWN - 1 <= 16 * NVB, "row blocks must cover the tile plus the alignment shift");
    static_assert(15 * DOWN + FD - 1 + DOWN - 1 < 32 * NDVK, "down-y K windows must cover the taps plus the alignment shift");
    static_assert(LIN_DI + 8 * (NLV - 1) + cdiv(DOWN - 1, UP) <= 32 * NLT, "composite operator windows must cover the alignment shift");

};

// ---------------------------------------------------------------------------------------------
// Filter coefficient of the polyphase operators, tile-relative indices (the tile's first upsampled row / column is a multiple of
// UP away from phase `ph`, as in flrelu_mfma_prepare_kernel).
__device__ __forceinline__ float up_coef(const float* fu, int FU, int UP, int ph, int flip, int u, int i) {
    const int a = u % UP, o = (a > ph) ? 1 : 0;
    const int kmin = o ? UP - (a - ph) : ph - a;
    const int jj = i - u / UP - o;
    if (jj < 0 || jj >= kFUT) return 0.f;
    const int tap = kmin + UP * jj;
    return flip ? fu[tap] : fu[FU - 1 - tap];
}
__device__ __forceinline__ float down_coef(const float* fd, int FD, int DOWN, int flip, int n, int urel) {
    const int kk = urel - DOWN * n;
    if (kk < 0 || kk >= FD) return 0.f;
    return flip ? fd[kk] : fd[FD - 1 - kk];
}

// Extra constant fragments of the wave kernels, appended to the workspace of flrelu_mfma_prepare_kernel:
//   LIN[v][t]  B operand [k][n]: n = l15 (orow of the block), k = krow(g, j) -> in-row 32 t + k - 8 v counted from the block's first
//              contributing input row; value = slope * sum_u DV[n][u] * UV[u][in-row]   (UV carries up^2 * gain)
//   DH2[t]     A operand [m][k]: m = l15 (ocol of the block), k = krow(g, j) -> ucol 32 t + k of the window starting at 16 DOWN cb
// and one scalar: an upper bound of the L1 norm of the rows of UV (|X2| <= bound * max |X1|).
template <typename T, int UP, int DOWN>
__global__ void flrelu_wave_prepare_kernel(char* __restrict__ wsb, const float* __restrict__ fu, const float* __restrict__ fd,
                                           int py0, int flip, float gain_total, float slope, int dshift) {
    // dshift: see flrelu_mfma_prepare_kernel (py0 carries it already; the down-y taps of the composite operator move by it)
    typedef WaveGeom<UP, DOWN, 32, 32> G;                      // fragment contents do not depend on the tile shape
    constexpr int FU = kFUT * UP, FD = kFUT * DOWN;
    const int phy = pos_mod(py0, UP);
    T* ws = (T*)(wsb + kWsWave);
    for (int idx = blockIdx.x * blockDim.x + threadIdx.x; idx < (G::NLIN + G::NDVK) * 512; idx += gridDim.x * blockDim.x) {
        const int j = idx & 7, lane = (idx >> 3) & 63, frag = idx >> 9;
        const int l15 = lane & 15, g = lane >> 4;
        float v = 0.f;
        if (frag < G::NLIN) {
            const int var = frag / G::NLT, t = frag % G::NLT;
            const int di = 32 * t + krow(g, j) - 8 * var;
            if (di >= 0) {
                float acc = 0.f;
                for (int kk = 0; kk < FD; kk++) {
                    const int u = DOWN * l15 + kk + dshift;    // upsampled row, from the block's first
                    acc += down_coef(fd, FD, DOWN, flip, l15, u - dshift) * up_coef(fu, FU, UP, phy, flip, u, di);
                }
                v = acc * gain_total * slope;
            }
        } else {
            const int t = frag - G::NLIN;
            v = down_coef(fd, FD, DOWN, flip, l15, 32 * t + krow(g, j));
        }
        ws[idx] = from_f32<T>(v);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        float best = 0.f;
        for (int a = 0; a < UP; a++) {
            float s = 0.f;
            for (int i = 0; i < kFUT + 2; i++) s += fabsf(up_coef(fu, FU, UP, phy, flip, a, i));
            best = fmaxf(best, s);
        }
        // + 2 %: the bound is applied to fp32 X1 while X2 is formed from its 16-bit rounding and 16-bit coefficients
        ((float*)(wsb + kWsScalars))[0] = best * fabsf(gain_total) * 1.02f;
    }
}

// ---------------------------------------------------------------------------------------------
// Sign codes, layout 2 (private to this kernel family): one byte = the 2-bit codes of 4 consecutive rows ("quad-row" q) of one
// column c.  With V = q >> 2 (the 16-row block) and gq = q & 3, the byte lives at
//     ((((c >> 4) * nV4 + (V >> 2)) * 4 + gq) * 16 + (c & 15)) * 4 + (V & 3)          nV4 = quad-rows / 16
// i.e. [plane][column block][V / 4][gq][column in block][V % 4]: lane (g, l15) of the X2 tiles (vb, nb) of a strip owns whole
// dwords -- its codes of 4 consecutive row blocks.
//   forward: one dword store per lane and 4 row blocks: 256 contiguous bytes per store instruction; bounds (plane bottom, row
//            blocks of the next strip) ride in the buffer descriptor of the column block, so no store is predicated;
//   backward: the window starts at an arbitrary (row, column) offset: a lane fetches the dwords of ITS column that cover quad-rows
//            r0 + 4 vb (and r0 + 4 vb + 1 when the window starts inside a quad) -- compile-time offsets from two address registers --
//            and extracts the bytes with v_alignbyte / v_perm.
// The tensor keeps the [N, C, sh, swb] shape of the ABI with sh rounded up to 16 quad-rows.
//
// Loop structure: ONE rolled loop over groups of two column blocks.  An iteration loads its 32-column input window, produces
// the packed X3 pair of its two blocks and runs the down-x pass of the output column block that pair completes; the earlier
// pairs that pass reads are loop-carried registers.  (Fully unrolled, the compiler's latency-driven scheduling stretched live
// ranges over the tile -- 210-240 VGPRs, two waves per SIMD, or spills; a wave spends most of its residency waiting on its own
// MFMA -> convert -> MFMA chains, so the SIMD only fills up with 3-4 of them.)
//
// EPI bits: 1 = per-plane factors (fused layer node), 2 = + encoder skip operand, 4 = per-strip output sums (the backward's bias
// gradient; kept apart from bit 1: the forward kernels of a training step scale but never sum, and the sums cost them registers).
//
// SIGN: AFCM_SIGNS_NONE / WRITE / READ, or kSignsReadAligned (this file only): READ with the strips' origin moved up by p.oy0
// output rows so that every strip's first upsampled row falls on a 16-row block of the sign tensor (the host does that whenever
// (oy0 DOWN + sy) can be made a multiple of 16: every up-2 / down-2 backward of the generator).  The window then starts on a
// quad-row and on a byte of the code dwords: 2-3 dwords per column block instead of 6 (no second "hi" set, no per-tile v_perm /
// v_bfe to re-align quads), a third of the registers for codes -- the general READ kernels retired 8.6 vector instructions per
// MFMA against 5.8 in the forward (profiles/r02_bench_pmc.txt) and were 56 % of the family's time.
constexpr int kSignsReadAligned = 3;
// LDS of one workgroup (4 waves) of flrelu_wave_kernel: keep-mask table, output staging, skip staging, input ring (the arrays declared
// at the top of the kernel), and the workgroups per CU = waves per SIMD that the 160 KB allow.  The register target of a variant
// (__launch_bounds__) is the tuning aid's (AFCM_WAVE_OCC_*) capped by that: asking for more than the LDS admits only made the compiler
// report a missed occupancy target on 14 variants (r04 build log) -- the 48-row strips (66 KB) and the skip variants (56 KB) run two
// workgroups per CU, the down-4 strips (74 KB) two, their skip variants (84 KB) one, whatever the register count.
template <int UP, int DOWN, int TOW, int TOH, int SIGN, int EPI>
constexpr int wave_lds_bytes() {
    typedef WaveGeom<UP, DOWN, TOW, TOH> G;
    constexpr bool RD = SIGN == AFCM_SIGNS_READ || SIGN == kSignsReadAligned;
    return (RD ? 2048 : 8) + 4 * TOH * 144 + ((EPI & 2) ? 4 * TOH * 80 : 16) + 4 * 16 * G::NMB * 144;
}
template <int UP, int DOWN, int TOW, int TOH, int SIGN, int EPI>
constexpr int wave_occupancy() {
    constexpr int want = (DOWN == 2 && (TOH <= 32 || SIGN == AFCM_SIGNS_READ || SIGN == kSignsReadAligned)) ? AFCM_WAVE_OCC_D2 : AFCM_WAVE_OCC_D4;
    constexpr int fit = (160 * 1024) / wave_lds_bytes<UP, DOWN, TOW, TOH, SIGN, EPI>();
    return fit < 1 ? 1 : (fit < want ? fit : want);
}
template <typename T, int UP, int DOWN, int TOW, int TOH, int SIGN, int EPI>
__global__ __launch_bounds__(256, (wave_occupancy<UP, DOWN, TOW, TOH, SIGN, EPI>())) void flrelu_wave_kernel(FlreluMfmaParams p) {
    typedef WaveGeom<UP, DOWN, TOW, TOH> G;
    typedef MfmaOps<T> M;
    typedef typename M::frag frag;
    constexpr bool RD = SIGN == AFCM_SIGNS_READ || SIGN == kSignsReadAligned, RA = SIGN == kSignsReadAligned;
    __shared__ uint2 lds_tab[RD ? 256 : 1];                               // READ: sign byte -> keep masks of its 4 rows
    // Output staging, private to each wave: 64 output columns (4 column blocks) x TOH rows, flushed as full 128-byte row segments.
    // (Stored straight from the accumulators a column block is 32 bytes per row: those reach HBM as partial lines -- measured
    // 1.9x / 2.5x the algorithmic write traffic in forward / backward.)
    constexpr int OPITCH = 144;                                            // bytes per staged row: 128 + 16 (16-byte aligned, spreads banks)
    __shared__ __attribute__((aligned(16))) unsigned char lds_o[4][TOH * OPITCH];
    // Encoder-skip staging (EPI & 2), private to each wave: the skip rows of TWO output column blocks (64 bytes per row) arrive by
    // 16-byte loads in the access lane map, are parked here and read back in the fragment lane map.  (r02 fetched every column
    // block by itself: two 4-byte loads per lane = 32 bytes per row and instruction, and the four visits of a 128-byte line were
    // a whole group apart -- the L1 had long dropped it: profiles/r03: HBM reads of the skip kernels 2.0-2.7x the algorithmic
    // bytes, L12 forward 232 us against 122-145 us for the same plane without a skip.)
    constexpr int KPITCH = 80;                                             // bytes per staged skip row: 64 + 16
    __shared__ __attribute__((aligned(16))) unsigned char lds_k[(EPI & 2) ? 4 : 1][(EPI & 2) ? TOH * KPITCH : 16];
    // Input ring, private to each wave: the strip's 16 NMB input rows x the two most recent 64-byte column PIECES.  The K window of
    // a group is 64 bytes wide but advances by 32 (up 2) or 16 (up 4) bytes: requested as windows (r02) every input byte crossed
    // the L1 -> L2 path two (four) times, and that path -- not HBM, not latency -- is what the kernels ran into: a load-only
    // replica of the access pattern (tools/ubench/strip_read.hip, profiles/r03_strip_read.txt) tops out at 2.5 TB/s of plane bytes
    // with overlapping 64-byte windows, whatever the prefetch depth or occupancy, against 4.6 TB/s for the same instructions on
    // disjoint 64-byte pieces.  So every byte is requested ONCE (one piece per 2 (4) groups, two (four) groups ahead of its first
    // use), parked here, and a group's A fragments are plain 16-byte LDS reads at the window's offset (the ds_bpermute lane-map
    // change is gone with them).  144-byte rows: 16 consecutive rows x 16 bytes hit 64 distinct banks.
    constexpr int IPITCH = 144;
    __shared__ __attribute__((aligned(16))) unsigned char lds_in[4][16 * G::NMB * IPITCH];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4;
#ifdef AFCM_WAVE_EXPERIMENT_LDS         // timing experiment only: LDS nobody uses, to run the same code at fewer waves per SIMD
    __shared__ unsigned char lds_pad[AFCM_WAVE_EXPERIMENT_LDS];
    if (p.total_tiles < 0) lds_pad[tid] = 1;
#endif
    if (RD) {
        if (tid < 128) ((uint4*)lds_tab)[tid] = ((const uint4*)((const char*)p.ws + kWsTable))[tid];
        __syncthreads();
    }
    // XCD-aware order: consecutive logical blocks (neighbouring tiles, shared halos) stay on one XCD / one L2
    int bid = blockIdx.x;
    {
        const int total = gridDim.x;
        bid = xcd_order(bid, total);
    }
    const int wt = bid * 4 + wave;                                       // this wave's strip: TOH output rows x the plane's width
    if (wt >= p.total_tiles) return;
    const int plane = p.magicP ? (int)__umulhi((unsigned)wt, p.magicP) : wt;
    const int tile = wt - plane * (p.tilesX * p.tilesY);
    const int ty = p.magicT ? (int)__umulhi((unsigned)tile, p.magicT) : tile;
    const int tx = tile - ty * p.tilesX;
    const int O0y = ty * TOH + (RA ? p.oy0 : 0);                          // (tilesX = 1: a strip spans the plane; RA: origin moved up, oy0 <= 0)
    const int U0x = 0, U0y = O0y * DOWN;
    (void)tx;
    // column groups: the last output column block, cdiv(yw, 16) - 1, ends on X3 pair (DOWN / 2) cb + NDVK - 1
    const int ncb = (p.yw + 15) >> 4;
    const int ng = (DOWN / 2) * (ncb - 1) + G::NDVK;
    // ... of which only the first ngc have to be COMPUTED (r06): the X3 pair of group gi covers upsampled columns [32 gi, 32 gi + 32), the
    // plane's last output column reads up to column (yw - 1) DOWN + FD - 1, and a WRITE call owes the codes of every column block of
    // the sign tensor.  The K windows of the last output column block reach NDVK pairs whatever the number of columns the plane has
    // in it: with yw = 16 m + 4 (every plane of the 256^2 generator: 36, 52, 84, 148, 276) the last group -- of 4, 5, 7, 11, 19 --
    // produced a pair that only padding columns read.  Those groups now run tail_group(): the down-x pass alone, on a zero pair.
    const int ngc = min(ng, max(((p.yw - 1) * DOWN + G::FD - 1) / 32 + 1, SIGN == AFCM_SIGNS_WRITE ? ((p.swq >> 4) + G::NBG - 1) / G::NBG : 0));
    const int I0x = -floor_div(p.px0 - U0x, UP), I0y = -floor_div(p.py0 - U0y, UP);
    const int S0x = I0x - (I0x & 1);                                      // first input column touched (even: aligned dword pairs)
    const bool lastY = (ty == p.tilesY - 1);

    // constant fragments (registers): UH[NBG] UV[UP] DVr[NDVK] from the LDS-tile kernels' workspace, LIN[NLIN] DH2[NDVK] from ours
    const frag* wsf = (const frag*)p.ws;                                  // UH[3] UV[UP] DVs[NDVK] DVr[NDVK] (DH[NDVK])
    const frag* wsw = (const frag*)((const char*)p.ws + kWsWave);         // LIN[NLIN] DH2[NDVK]
    constexpr int F_UV = 3, F_DVS = 3 + UP, F_DVR = 3 + UP + G::NDVK;
    frag uh[G::NBG], uv[UP], dvr[G::NDVK], lin[G::NLIN], dh[G::NDVK];
#pragma unroll
    for (int v = 0; v < G::NBG; v++) uh[v] = wsf[v * 64 + lane];
#pragma unroll
    for (int v = 0; v < UP; v++) uv[v] = wsf[(F_UV + v) * 64 + lane];
#pragma unroll
    for (int t = 0; t < G::NDVK; t++) dvr[t] = wsf[(F_DVR + t) * 64 + lane];
#pragma unroll
    for (int f = 0; f < G::NLIN; f++) lin[f] = wsw[f * 64 + lane];
#pragma unroll
    for (int t = 0; t < G::NDVK; t++) dh[t] = wsw[(G::NLIN + t) * 64 + lane];
    // no element of the tile can reach the clamp while max |X1| stays below this
    const float cthr1 = p.clamp / (fmaxf(p.slope, 1.f) * ((const float*)((const char*)p.ws + kWsScalars))[0]);

    // ---- input: rows [I0y, +16 NMB) x columns [S0x + IWSTEP gi, +32) per group, straight into A fragments
    const T* xp = (const T*)p.x + (size_t)plane * p.xh * p.xld;
#ifdef AFCM_WAVE_EXPERIMENT_NOLOAD    // timing experiment only: every input load falls outside the descriptor
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, 0, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc((void*)xp, 0, p.xh * p.xld * 2, 0x00020000);
#endif
    // Lane map of the global accesses: an MFMA fragment puts 16 different ROWS on consecutive lanes (lane = 16 g + l15), which the
    // memory pipeline sees as 64 separate 16-byte requests per instruction (measured: the two output stores of a group cost more
    // than all of its arithmetic).  So loads and stores use lane = 4 row + chunk -- 4 consecutive lanes cover 64 (stores: 32)
    // contiguous bytes -- and ds_bpermute moves the dwords between that map and the fragment map: no LDS allocation, no VALU.
    const int lrow = lane >> 2, lchk = lane & 3;
    // byte offset of (row I0y + lrow, column S0x + 8 lchk); rows above the plane give a negative offset = beyond the descriptor's
    // range as an unsigned number, rows below it exceed the record count: both read as zero without a predicate
    const int xoff0 = ((I0y + lrow) * p.xld + S0x + 8 * lchk) * 2;
    const int xrow16 = 32 * p.xld;                                         // 16 rows, bytes
    constexpr unsigned kOut = 0x40000000u;                               // out-of-range marker (planes stay below 2^29 bytes)
    // piece k = input columns [S0x + 32 k, + 32): 16 rows x 64 bytes per instruction (access lane map), loads only -- the data
    // stays in these registers until park_piece() (r02 converted right behind the load, inside the fast / edge branch, and so
    // waited for every window on the spot)
    unsigned char* const ring = lds_in[wave];
    constexpr int ABYTES = 2 * G::IWSTEP;                                  // bytes the window advances per group (32 / 16)
    constexpr int PGRP = 64 / ABYTES;                                      // groups per piece
    static_assert(64 % ABYTES == 0 && PGRP >= 2, "pieces of 64 bytes, at least two groups each");
    auto load_piece = [&](int k, u32x4 (&raw)[G::NMB]) __attribute__((always_inline)) {
        const int c0 = S0x + 32 * k;                                      // wave-uniform
        if (__builtin_expect(c0 >= 0 && c0 + 32 <= p.xw, 1)) {
#pragma unroll
            for (int mb = 0; mb < G::NMB; mb++)
#ifdef AFCM_WAVE_EXPERIMENT_CACHED_LOADS   // timing experiment only (wrong results): every piece is the strip's piece 0 or 1 -- real data, served by the L1 / L2
                raw[mb] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (unsigned)(xoff0 + 64 * (k & 1) + mb * xrow16), 0, AFCM_WAVE_LOAD_AUX);
#else
                raw[mb] = __builtin_amdgcn_raw_buffer_load_b128(rsx, (unsigned)(xoff0 + 64 * k + mb * xrow16), 0, AFCM_WAVE_LOAD_AUX);
#endif
        } else {
            // the piece crosses the left or right edge of the plane: dword by dword, columns outside it read as zero
            // (even plane widths: the two elements of a dword are in or out together; rows outside: as above, except that a
            // negative row offset plus the marker must not wrap back into range)
            const int col = c0 + 8 * lchk;
            unsigned cofs[4];
#pragma unroll
            for (int w = 0; w < 4; w++) cofs[w] = (unsigned)(col + 2 * w) < (unsigned)p.xw ? (unsigned)(64 * k + 4 * w) : kOut;
#pragma unroll
            for (int mb = 0; mb < G::NMB; mb++) {
                const unsigned ro = (I0y + 16 * mb + lrow < 0) ? kOut : (unsigned)(xoff0 + mb * xrow16);
#pragma unroll
                for (int w = 0; w < 4; w++) raw[mb][w] = __builtin_amdgcn_raw_buffer_load_b32(rsx, ro + cofs[w], 0, 0);
            }
        }
    };
    auto park_piece = [&](int k, const u32x4 (&raw)[G::NMB]) __attribute__((always_inline)) {
#pragma unroll
        for (int mb = 0; mb < G::NMB; mb++) *(u32x4*)(ring + (unsigned)((16 * mb + lrow) * IPITCH + (k & 1) * 64 + 16 * lchk)) = raw[mb];
    };
    // A fragments of group gi: lane (g, l15) takes the 16 bytes at window offset 16 g of row l15; ring position = byte mod 128
    auto frags = [&](int gi, frag (&a)[G::NMB]) __attribute__((always_inline)) {
        const unsigned o = (unsigned)(l15 * IPITCH) + ((unsigned)(ABYTES * gi + 16 * g) & 127u);
#pragma unroll
        for (int mb = 0; mb < G::NMB; mb++) a[mb] = *(const frag*)(ring + o + (unsigned)(16 * mb * IPITCH));
    };

    // ---- sign codes (layout 2, see above)
    const int nblk = p.swq >> 4, nV4 = p.shq >> 4;                         // column blocks, groups of 4 row blocks (shq is a multiple of 16)
    const int blkbytes = nV4 * 256;
    unsigned char* const splane = p.s + (size_t)plane * p.shq * p.swq;
    constexpr int NA = cdiv(G::NVB, 4);                                    // dwords of codes per lane and column block
    // WRITE: this strip's row blocks are V = ty * OWN_VB + vb (a multiple of 4 at vb = 0; the tall single-strip tile starts at 0):
    // offset of this lane's first dword inside a column block; dword groups this strip may write (the rest is the next strip's, or
    // below the plane)
    const unsigned sgw_off = (unsigned)((((U0y >> 4) >> 2) * 4 + g) * 64 + l15 * 4);
    const int v4_end = lastY ? nV4 : min(nV4, ((U0y >> 4) + G::OWN_VB) >> 2);
    // READ: lane (g, l15) follows column U0x + sx + l15 and quad-rows r0 + 4 vb (lo) and r0 + 4 vb + 1 (hi), r0 = Q0 + g
    const int yy = (U0y + p.sy) & 3;                                       // row offset inside a quad (0 for the aligned READ kernels)
    const int Q0 = (U0y + p.sy) >> 2;
    int sgr_lo = 0, sgr_hi = 0, sgr_dl = 0, sgr_dh = 0;
    unsigned sgr_sl = 0, sgr_sh = 0;
    __amdgpu_buffer_rsrc_t rss = rsx;
    if (RD) {
        rss = __builtin_amdgcn_make_buffer_rsrc((void*)splane, 0, p.shq * p.swq, 0x00020000);
        const int c = U0x + p.sx + l15, r0 = Q0 + g, r1 = r0 + 1;
        // columns left of the tensor give a negative block = a negative offset, columns right of it a block beyond the last:
        // both fall outside the descriptor and read code 0 ("unchanged", filtered_lrelu.cu:564-571)
        const int cbase = (c >> 4) * nV4 * 256 + (c & 15) * 4;
        sgr_dl = (r0 >> 2) >> 2; sgr_sl = (unsigned)((r0 >> 2) & 3);
        sgr_dh = (r1 >> 2) >> 2; sgr_sh = (unsigned)((r1 >> 2) & 3);
        sgr_lo = cbase + (sgr_dl * 4 + (r0 & 3)) * 64;
        sgr_hi = cbase + (sgr_dh * 4 + (r1 & 3)) * 64;
    }
    // dword groups outside the tensor would alias a neighbouring column block: select them out (wave-uniform fast case)
    const bool rows_inside = Q0 >= 0 && (((Q0 + 4) >> 2) >> 2) + NA + 1 <= nV4;
    // RA: dwords per column block -- code bytes sl .. sl + NVB - 1 of the lane's dword run (sl = (first row block) & 3, wave-uniform)
    constexpr int NL = RA ? cdiv(3 + G::NVB, 4) : NA + 1;
    // READ: the sign dwords of a group's two column blocks, fetched one group ahead like the input window (consumed right after
    // the up-y products: fetched in place, every column block exposed a full memory round trip)
    auto load_signs = [&](int nb, unsigned (&sg)[RA ? 1 : 2][NL]) __attribute__((always_inline)) {
#ifdef AFCM_WAVE_EXPERIMENT_CACHED_LOADS   // timing experiment only (wrong results): the codes of the strip's first two column blocks, again and again
        nb &= 1;
#endif
        const int blo = sgr_lo + nb * blkbytes, bhi = sgr_hi + nb * blkbytes;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            unsigned ol = (unsigned)(blo + 256 * i), oh = (unsigned)(bhi + 256 * i);
            if (!rows_inside) {
                ol = ((unsigned)(sgr_dl + i) < (unsigned)nV4) ? ol : 0x80000000u;
                oh = ((unsigned)(sgr_dh + i) < (unsigned)nV4) ? oh : 0x80000000u;
            }
#ifdef AFCM_WAVE_EXPERIMENT_NOSIGNLOAD  // timing experiment only (wrong results): every code load falls outside the descriptor
            ol |= 0x80000000u;
#endif
            sg[0][i] = __builtin_amdgcn_raw_buffer_load_b32(rss, ol, 0, AFCM_WAVE_SIGNLOAD_AUX);
            if constexpr (!RA) sg[1][i] = (yy != 0) ? __builtin_amdgcn_raw_buffer_load_b32(rss, oh, 0, AFCM_WAVE_SIGNLOAD_AUX) : 0u;
        }
    };

    // ---- output
    T* const yp = (T*)p.y + (size_t)plane * p.yh * p.yld;
#ifdef AFCM_WAVE_EXPERIMENT_NOSTORE   // timing experiment only: every output store falls outside the descriptor
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)yp, 0, 0, 0x00020000);
#else
#ifdef AFCM_WAVE_EXPERIMENT_STORE_ALIAS   // timing experiment only (wrong results): every plane's stores land in plane 0's first 64 KB
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.yh * p.yld * 2, 0x00020000);
#else
    const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc((void*)yp, 0, p.yh * p.yld * 2, 0x00020000);
#endif
#endif
    const bool has_skip = (EPI & 2) && p.skip != nullptr;
    const __amdgpu_buffer_rsrc_t rsk = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(has_skip ? (const T*)p.skip + (size_t)plane * p.yh * p.kld : (const T*)p.x), 0, has_skip ? p.yh * p.kld * 2 : 0, 0x00020000);
    // skip operand: byte offset of (row O0y + lrow, column 8 lchk) (access lane map)
    const int koff0 = ((O0y + lrow) * p.kld + 8 * lchk) * 2;
    const int krow16 = 32 * p.kld;
    const float osc = (EPI & 1) ? (p.oscale ? p.oscale[plane] : 1.f) * (p.oscale2 ? p.oscale2[plane] : 1.f) : 1.f;
    float psum = 0.f;
    unsigned char* const stage = lds_o[wave];
    if (p.yld != p.yw) {
        // a pitched y is written in whole lines up to the pitch: column blocks the strip never produces (past the last started
        // one) must not carry whatever the LDS held -- the padding's contract is "finite" (include/afcm_hip.h)
        for (int i = lane * 16; i < TOH * OPITCH; i += 64 * 16) *(uint4*)(stage + i) = make_uint4(0u, 0u, 0u, 0u);
    }
    const unsigned st_w = (unsigned)(l15 * OPITCH + g * 8);                // fragment lane (g, l15): 4 columns of row l15 (+ 16 ob rows, + 32 B per column block)
    const unsigned st_r = (unsigned)((lane >> 3) * OPITCH + (lane & 7) * 16);   // flush lane: 8 columns (16 B) of row lane / 8 (+ 8 rows per instruction)
    const int fl_row = lane >> 3, fl_chk = lane & 7;

    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // down-x of output column block cb, transposed: Y'[ocol][orow] = DH2 * X3'; a lane holds 4 consecutive output columns of
    // one row.  xw[t] = the X3 pairs of its K windows.  The packed result goes to the staging buffer.
    // encoder feature of output column block cb in the access lane map (columns / rows outside the plane read as zero): requested
    // at the top of the group whose down-x pass consumes it -- fetched in place, every column block waited a full memory round trip
    // (the skip kernels sat in memory waits for 46 % of their residency, the plain ones 23 %)
    // skip rows of output column blocks (cb, cb + 1), cb even: one 16-byte load per lane and 16 rows (access lane map: 4 lanes =
    // 64 contiguous bytes of a row).  Requested one group before the first down-x pass that needs them and parked in LDS at the
    // top of that group (the registers are live for one group in two).  Rows below the plane exceed the record count and read
    // as zero; columns right of the plane hold the next row's head (dense) or padding (pitched): finite, never stored or summed.
    unsigned char* const kstage = lds_k[(EPI & 2) ? wave : 0];
    auto load_skip = [&](int cb, u32x4 (&skw)[G::NOB]) __attribute__((always_inline)) {
#pragma unroll
        for (int ob = 0; ob < G::NOB; ob++)
            skw[ob] = __builtin_amdgcn_raw_buffer_load_b128(rsk, (unsigned)(koff0 + ob * krow16 + 32 * cb), 0, 0);
    };
    auto park_skip = [&](const u32x4 (&skw)[G::NOB]) __attribute__((always_inline)) {
#pragma unroll
        for (int ob = 0; ob < G::NOB; ob++) *(u32x4*)(kstage + (unsigned)((16 * ob + lrow) * KPITCH + 16 * lchk)) = skw[ob];
    };
    auto phase_b = [&](int cb, const u32x4 (&xw)[G::NOB][G::NDVK]) __attribute__((always_inline)) {
        const int slot = cb & 3;
#pragma unroll
        for (int ob = 0; ob < G::NOB; ob++) {
            f32x4 acc = zero4;
#pragma unroll
            for (int t = 0; t < G::NDVK; t++) acc = M::mma(dh[t], as_frag<frag>(xw[ob][t]), acc);
            if (EPI & 2) {
                // encoder feature of this lane's 4 columns (x + x_skip, NET:376-377), fragment lane map, from the parked rows
                const uint2 e = *(const uint2*)(kstage + (unsigned)((16 * ob + l15) * KPITCH + (cb & 1) * 32 + g * 8));
                union { unsigned u; T t[2]; } e0, e1;
                e0.u = e.x;
                e1.u = e.y;
                if (16 * cb + 16 > p.yw) {                       // wave-uniform: the plane's last column block -- columns right of
                    if (16 * cb + 4 * g + 2 > p.yw) e0.u = 0u;   // the plane are the next row's head or padding (anything, NaN included)
                    if (16 * cb + 4 * g + 4 > p.yw) e1.u = 0u;
                }
                acc[0] += to_f32(e0.t[0]);
                acc[1] += to_f32(e0.t[1]);
                acc[2] += to_f32(e1.t[0]);
                acc[3] += to_f32(e1.t[1]);
            }
            if (EPI & 1) acc *= osc;
            if (EPI & 4) {
                if (!(lastY || cb == ncb - 1 || (RA && O0y < 0))) {                            // wave-uniform: every element is inside the plane
                    psum += (acc[0] + acc[1]) + (acc[2] + acc[3]);
                } else {
                    const int fx = 16 * cb + 4 * g;                                            // fragment lane map
                    const bool rin = (unsigned)(O0y + 16 * ob + l15) < (unsigned)p.yh;
                    psum += (rin && fx + 2 <= p.yw ? acc[0] + acc[1] : 0.f) + (rin && fx + 4 <= p.yw ? acc[2] + acc[3] : 0.f);
                }
            }
            uint2 w;
            w.x = pack2<T>(acc[0], acc[1]);
            w.y = pack2<T>(acc[2], acc[3]);
            *(uint2*)(stage + st_w + (unsigned)(ob * 16 * OPITCH + slot * 32)) = w;
        }
    };
    // flush of the staged 64-column group that column block cb completed (every fourth block, and the last one): 8 rows x 128 bytes
    // per store instruction.  Called at the TOP of the group after the one that staged it, ahead of that group's window request:
    // s_waitcnt vmcnt counts loads and stores together, in order, so the wait for a window covers everything older and must
    // name the exact number of younger operations -- with the flush (4 to 16 stores, by path) BEHIND the window request the
    // compiler had to assume the path without it, and on every fourth group the wait for the window also drained the four
    // stores issued a moment before (ISA of the r02 build: s_waitcnt vmcnt(2) at the top of every group).  Now the only younger
    // operations are the group's sign stores: the same count on every path.
    auto flush = [&](int cb) __attribute__((always_inline)) {
        const int slot = cb & 3;
        {
            const int c0 = 64 * (cb >> 2) + 8 * fl_chk;                                         // this lane's first output column
            // wave-uniform: the whole 64-column group lies inside the row (a pitched y: inside the pitch -- its rows start on 128-byte
            // lines and EVERY flush is whole lines; columns >= yw are padding and receive whatever the staging buffer holds: finite
            // values, products of in-range data and zeros)
            const bool full = 64 * (cb >> 2) + 64 <= p.yld;
            const unsigned gofs = (unsigned)(((O0y + fl_row) * p.yld + c0) * 2);
#pragma unroll
            for (int j = 0; j < TOH / 8; j++) {
                const u32x4 v = *(const u32x4*)(stage + st_r + (unsigned)(j * 8 * OPITCH));
                unsigned off = gofs + (unsigned)(j * 16 * p.yld);
#ifdef AFCM_WAVE_EXPERIMENT_STORE_ALIAS  // timing experiment only: all output stores land in one 64 KB window (no HBM write traffic)
                off &= 0xfff0u;
#endif
                if (p.st_plain) {
                    // dense rows wider than one 64-column group (the 84- / 86-wide planes: 168- / 172-byte rows): a row leaves as a 128-byte
                    // piece plus a 40-byte tail of pair stores, neither on a line boundary -- as write-back stores L2 merges them into whole
                    // lines, as non-temporal ones every piece went to memory by itself (encoder_8 forward 140 -> 112 us, backward 171 -> 129)
                    if (full) {
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsy, off, 0, 0);
                    } else {
#pragma unroll
                        for (int w2 = 0; w2 < 4; w2++)
                            __builtin_amdgcn_raw_buffer_store_b32(v[w2], rsy, (c0 + 2 * w2 + 2 <= p.yw && 8 * fl_chk + 2 * w2 < 16 * (slot + 1)) ? off + 4u * w2 : kOut, 0, 0);
                    }
                } else if (full) {
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsy, off, 0, AFCM_WAVE_STORE_AUX);
                } else if (p.yld != p.yw) {
                    // pitched rows end on a 16-byte boundary: the lanes whose 8 columns lie inside the pitch store, the others fall
                    // outside the descriptor
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsy, (c0 + 8 <= p.yld) ? off : kOut, 0, AFCM_WAVE_STORE_AUX);
                } else {
                    // the group crosses the right edge (or is only partly produced): pair by pair (even plane widths)
#pragma unroll
                    for (int w2 = 0; w2 < 4; w2++)
                        __builtin_amdgcn_raw_buffer_store_b32(v[w2], rsy, (c0 + 2 * w2 + 2 <= p.yw && 8 * fl_chk + 2 * w2 < 16 * (slot + 1)) ? off + 4u * w2 : kOut, 0, AFCM_WAVE_STORE_AUX);
                }
            }
        }
    };

    // ---- the strip: groups of two column blocks: up-x, up-y, activation (+ codes), down-y, then the down-x pass the new X3 pair
    // completes.  EXACT: the rare second pass over a strip in which the clamp was reached.  LASTY: the strip also writes the codes of
    // its halo row blocks (every other strip leaves them to the strip below and skips their extraction).
    // One group; the caller alternates two register sets for everything that is carried from one group to the next.
    auto group = [&](int gi, auto exact_c, auto lasty_c, u32x4 (&raw)[G::NMB], unsigned (&sg)[G::NBG][RA ? 1 : 2][NL],
                     u32x4 (&hist)[G::NOB][G::NHIST], u32x4 (&cur)[G::NOB], u32x4 (&skw)[G::NOB], float& amax, unsigned& anyc) __attribute__((always_inline)) {
        constexpr bool EXACT = decltype(exact_c)::value, LASTY = decltype(lasty_c)::value;
        constexpr int NVW = LASTY ? G::NVB : G::OWN_VB;                  // row blocks whose codes this strip writes
        // one group ahead, issued before this group's stores: vmcnt counts in order, so waiting for these loads at the top of
        // the next group does not wait for the (younger) stores
        // the output column block whose last K window is this group's pair: pairs (DOWN / 2) cb ... + NDVK - 1 = gi
        const bool has_b = gi >= G::NDVK - 1 && (gi - (G::NDVK - 1)) % (DOWN / 2) == 0;
        const int cb_b = (gi - (G::NDVK - 1)) / (DOWN / 2);
        // aligned READ: code bytes sl .. sl + NVB - 1 of each column block's dword run, one funnel shift per 4 row blocks -- taken
        // first, so that the registers can take the next group's request below and nothing younger than the codes is waited for
        unsigned four[G::NBG][NA] = {};
        if (RA) {
#pragma unroll
            for (int nbl = 0; nbl < G::NBG; nbl++) {
#pragma unroll
                for (int i = 0; i < NL; i++) anyc |= sg[nbl][0][i];
#pragma unroll
                for (int k = 0; k < NA; k++)
                    four[nbl][k] = __builtin_amdgcn_alignbyte(k + 1 < NL ? sg[nbl][0][k + 1] : 0u, sg[nbl][0][k], sgr_sl);
            }
#ifndef AFCM_WAVE_NO_PIN_SIGNS
            // ... and pinned here: left to the scheduler the funnel shifts sank below the flush, so the wait for the codes (vmcnt is in
            // order and the count must hold on the path without a flush: 0) also drained the four output stores issued a moment
            // before, every fourth group, and the next request left half a group late.  (An empty asm with a memory clobber: the
            // shifts are done above it, every store and load of the group stays below it; sched_barrier alone binds one scheduler)
#pragma unroll
            for (int nbl = 0; nbl < G::NBG; nbl++)
#pragma unroll
                for (int k = 0; k < NA; k++) asm volatile("" : "+v"(four[nbl][k]) : : "memory");
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        // the window of group PGRP (k - 1) + 1 is the first to reach into piece k: park it (requested PGRP groups ago, or before
        // the loop) and request piece k + 1 into the same registers; then this group's fragments; then the stores of the 64-column
        // output group the previous down-x pass may have completed
        if (gi % PGRP == 1) {
            const int k = (gi - 1) / PGRP + 1;
            park_piece(k, raw);
            load_piece(k + 1, raw);                  // (past the plane's right edge: dropped by the descriptor)
        }
        frag a_in[G::NMB];
        frags(gi, a_in);
        {
            const int gp = gi - 1;
            const bool prv_b = gp >= G::NDVK - 1 && (gp - (G::NDVK - 1)) % (DOWN / 2) == 0;
            const int cb_p = (gp - (G::NDVK - 1)) / (DOWN / 2);
            if (prv_b && ((cb_p & 3) == 3 || cb_p == ncb - 1)) flush(cb_p);
        }
        if (EPI & 2) {
            // this group's down-x pass opens a column-block pair: its skip rows were requested a group ago (older than the window
            // that the group waited for) -- park them; the NEXT group opens one: request its rows, ahead of the next window
            if (has_b && (cb_b & 1) == 0) park_skip(skw);
            const int gn = gi + 1;
            const bool nxt_b = gn >= G::NDVK - 1 && (gn - (G::NDVK - 1)) % (DOWN / 2) == 0;
            const int cb_n = (gn - (G::NDVK - 1)) / (DOWN / 2);
            if (nxt_b && (cb_n & 1) == 0 && cb_n < ncb) load_skip(cb_n, skw);
        }
        if (RA) {
            // (unconditional: past the last group the blocks lie outside the tensor and read as zero; a condition here costs
            // register copies that wait for the window just requested)
            load_signs((gi + 1) * G::NBG, sg[0]);
            load_signs((gi + 1) * G::NBG + 1, sg[1]);
#ifndef AFCM_WAVE_NO_PIN_SIGNS
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
#pragma unroll
        for (int nbl = 0; nbl < G::NBG; nbl++) {
            const int nb = gi * G::NBG + nbl;
            // up-x: X1[mb] = In[mb] * UH, packed in pairs as the B operand of up-y / the A operand of the composite operator
            f32x4 x1[G::NMB];
#pragma unroll
            for (int mb = 0; mb < G::NMB; mb++) x1[mb] = M::mma(a_in[mb], uh[nbl], zero4);
            frag q[G::NQ];
#pragma unroll
            for (int m = 0; m < G::NQ; m++) q[m] = pack_pair<T>(x1[m], (m + 1 < G::NMB) ? x1[m + 1] : zero4);
            if (!RD && !EXACT) {
#pragma unroll
                for (int mb = 0; mb < G::NMB; mb++) {
                    amax = __builtin_elementwise_maximum(__builtin_elementwise_maximum(amax, __builtin_fabsf(x1[mb][0])), __builtin_fabsf(x1[mb][1]));
                    amax = __builtin_elementwise_maximum(__builtin_elementwise_maximum(amax, __builtin_fabsf(x1[mb][2])), __builtin_fabsf(x1[mb][3]));
                }
            }
            // READ: the codes of this block's X2 tiles
            unsigned codes[G::NVB];
            if constexpr (RA) {
                // a byte of the funnel-shifted runs (top of the group) each
#pragma unroll
                for (int k = 0; k < NA; k++)
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        if (4 * k + j < G::NVB) codes[4 * k + j] = __builtin_amdgcn_ubfe(four[nbl][k], 8 * j, 8);
            } else if constexpr (RD) {
#pragma unroll
                for (int i = 0; i <= NA; i++) anyc |= sg[nbl][0][i] | sg[nbl][1][i];
#pragma unroll
                for (int k = 0; k < NA; k++) {
                    // the 4 codes of row blocks 4 k .. 4 k + 3: bytes sl .. sl + 3 of the dword pair (k, k + 1)
                    const unsigned al = __builtin_amdgcn_alignbyte(sg[nbl][0][k + 1], sg[nbl][0][k], sgr_sl);
                    const unsigned ah = yy != 0 ? __builtin_amdgcn_alignbyte(sg[nbl][1][k + 1], sg[nbl][1][k], sgr_sh) : 0u;
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        if (4 * k + j < G::NVB) {
                            // quad-rows r0 + 4 vb (low byte) and r0 + 4 vb + 1 (high byte), shifted to the window's first row
                            const unsigned two = __builtin_amdgcn_perm(ah, al, 0x0c0c0000u | ((4u + j) << 8) | (unsigned)j);
                            codes[4 * k + j] = __builtin_amdgcn_ubfe(two, 2 * yy, 8);
                        }
                    }
                }
                // Both register sets are free once the second block's codes are out: request the NEXT group's two blocks here.
                // Loads return in order, so what matters is what a request queues behind and how far away its use is: requested at
                // the top of a group (behind the input window's HBM round trip) with their use half a group away, the codes arrived
                // late at every block -- the transposed kernels waited 42 % of their time on memory against 23 % in the forward
                // kernels.  From here the first block has half a group, the second a whole one: backward pass over the generator's
                // layers 5.45 -> 5.14 ms.  (Each block re-requested right after its own extraction, a whole group ahead for both:
                // 5.76 ms.)
                if (nbl == G::NBG - 1 && gi + 1 < ng) {
                    load_signs(nb + 1, sg[0]);
                    load_signs(nb + 2, sg[1]);
                }
            }
            // WRITE: this block's descriptor: rows below the plane / not owned and blocks beyond the tensor fall outside it and
            // are dropped by the memory pipeline
            __amdgpu_buffer_rsrc_t rsw = rsx;
            if (SIGN == AFCM_SIGNS_WRITE) {
                const int blk = (U0x >> 4) + nb;
#ifdef AFCM_WAVE_EXPERIMENT_STORE_ALIAS   // (... and every block's codes in the tensor's first block)
                rsw = __builtin_amdgcn_make_buffer_rsrc((void*)p.s, 0, blk < nblk ? v4_end * 256 : 0, 0x00020000);
#else
                rsw = __builtin_amdgcn_make_buffer_rsrc((void*)(splane + (size_t)blk * blkbytes), 0, blk < nblk ? v4_end * 256 : 0, 0x00020000);
#endif
            }

            // up-y + activation.  Fast path: relu(X2) (forward) / keep-mask & X2 (backward) is the only operand kept; the linear
            // part comes from X1 through the composite operator.  Exact path: the activated value itself.
            u32x4 rv[G::NPAIR];
            unsigned wc[4 * NA];          // WRITE, exact path: per tile the code byte
            // WRITE, fast path: per PAIR of tiles (2 i, 2 i + 1) the sign bytes (0xff = negative) of rows 0, 1 (sa) and 2, 3 (sb) as
            // [even tile, odd tile][row]: ONE v_perm_b32 with the sign-extending selectors (8 .. 11) per packed dword pair -- r03
            // shifted the sign bits out per dword (two v_pk_lshrrev + a v_lshl_or per tile) and then folded bytes pairwise
            // (perm, shift, or): 19 vector instructions per code dword against 12 now
            unsigned sa[2 * NA], sb[2 * NA], ev0 = 0, ev1 = 0;
#pragma unroll
            for (int i = 0; i < 4 * NA; i++) wc[i] = 0;
#pragma unroll
            for (int i = 0; i < 2 * NA; i++) sa[i] = sb[i] = 0;
#pragma unroll
            for (int vb = 0; vb < 2 * G::NPAIR; vb++) {
                unsigned r0 = 0, r1 = 0;
                if (vb < G::NVB) {
                    f32x4 x2 = M::mma(uv[vb % UP], q[vb / UP], zero4);
                    if (RD) {
                        if (!EXACT) {
#ifdef AFCM_WAVE_EXPERIMENT_NOTABLE     // timing experiment only (wrong results): no keep-mask lookup
                            const uint2 keep = make_uint2(0xffffffffu ^ codes[vb], 0xffffffffu);
#else
                            const uint2 keep = lds_tab[codes[vb]];
#endif
                            r0 = pack2<T>(x2[0], x2[1]) & keep.x;
                            r1 = pack2<T>(x2[2], x2[3]) & keep.y;
                        } else {
#pragma unroll
                            for (int r = 0; r < 4; r++) {
                                const unsigned c = codes[vb] >> (2 * r);
                                float v = x2[r];
                                if (c & 1u) v *= p.slope;
                                if (c & 2u) v = 0.f;
                                x2[r] = v;
                            }
                            r0 = pack2<T>(x2[0], x2[1]);
                            r1 = pack2<T>(x2[2], x2[3]);
                        }
                    } else if (!EXACT) {
                        const unsigned d0 = pack2<T>(x2[0], x2[1]), d1 = pack2<T>(x2[2], x2[3]);
                        r0 = relu_pk(d0);
                        r1 = relu_pk(d1);
                        if (SIGN == AFCM_SIGNS_WRITE && vb < NVW) {
                            // selectors: 8 / 9 = sign of the low / high half of the second operand, 10 / 11 = of the first
                            if ((vb & 1) == 0) {
                                ev0 = d0;
                                ev1 = d1;
                                if (vb == NVW - 1) {
                                    sa[vb >> 1] = __builtin_amdgcn_perm(0u, d0, 0x0c090c08u);
                                    sb[vb >> 1] = __builtin_amdgcn_perm(0u, d1, 0x0c090c08u);
                                }
                            } else {
                                sa[vb >> 1] = __builtin_amdgcn_perm(d0, ev0, 0x0b090a08u);
                                sb[vb >> 1] = __builtin_amdgcn_perm(d1, ev1, 0x0b090a08u);
                            }
                        }
                    } else {
                        unsigned wcode = 0;
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            float v = x2[r];
                            unsigned c = __float_as_uint(v) >> 31;
                            if (c) v *= p.slope;
                            if (fabsf(v) > p.clamp) { c = 2u; v = (v < 0.f) ? -p.clamp : p.clamp; }
                            wcode |= c << (2 * r);
                            x2[r] = v;
                        }
                        r0 = pack2<T>(x2[0], x2[1]);
                        r1 = pack2<T>(x2[2], x2[3]);
                        if (vb < NVW) wc[vb] = wcode;
                    }
                }
                rv[vb >> 1][2 * (vb & 1)] = r0;
                rv[vb >> 1][2 * (vb & 1) + 1] = r1;
            }
            if (SIGN == AFCM_SIGNS_WRITE) {
                // one dword of 4 code bytes per 4 row blocks: 256 contiguous bytes per store instruction
#pragma unroll
                for (int d = 0; d < cdiv(NVW, 4); d++) {
                    unsigned dw;
                    if (!EXACT) {
                        // sa / sb of the tile pairs (2 d, 2 d + 1): [tile][row] -> one dword per row with a byte per tile, then bit 2 r
                        // of every byte from row r (code 1 = negative)
                        const unsigned a0 = sa[2 * d], a1 = sa[2 * d + 1], b0 = sb[2 * d], b1 = sb[2 * d + 1];
                        const unsigned r0 = __builtin_amdgcn_perm(a1, a0, 0x05040100u), r1 = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
                        const unsigned r2 = __builtin_amdgcn_perm(b1, b0, 0x05040100u), r3 = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
                        dw = (r0 & 0x01010101u) | (r1 & 0x04040404u) | (r2 & 0x10101010u) | (r3 & 0x40404040u);
                    } else {
                        const unsigned p01 = __builtin_amdgcn_perm(wc[4 * d + 1], wc[4 * d], 0x0c0c0400u);
                        const unsigned p23 = __builtin_amdgcn_perm(wc[4 * d + 3], wc[4 * d + 2], 0x0c0c0400u);
                        dw = __builtin_amdgcn_perm(p23, p01, 0x05040100u);
                    }
                    __builtin_amdgcn_raw_buffer_store_b32(dw, rsw, sgw_off + 256u * d, 0, AFCM_WAVE_STORE_AUX);
                }
            }

            // down-y, transposed: X3'[ucol][orow]; a lane ends up with 4 consecutive ucols of one orow = its share of the
            // down-x B operand (K order krow())
#pragma unroll
            for (int ob = 0; ob < G::NOB; ob++) {
                f32x4 x3 = zero4;
                if (!EXACT) {
#pragma unroll
                    for (int t = 0; t < G::NLT; t++) x3 = M::mma(q[G::lin_q(ob, t)], lin[G::lin_f(ob, t)], x3);
#pragma unroll
                    for (int t = 0; t < G::NDVK; t++) x3 = M::mma(as_frag<frag>(rv[(DOWN / 2) * ob + t]), dvr[t], x3);
                } else {
                    // rv holds the activated values: (slope DV + (1 - slope) DV) rv = DV rv
#pragma unroll
                    for (int t = 0; t < G::NDVK; t++) {
                        x3 = M::mma(as_frag<frag>(rv[(DOWN / 2) * ob + t]), wsf[(F_DVS + t) * 64 + lane], x3);
                        x3 = M::mma(as_frag<frag>(rv[(DOWN / 2) * ob + t]), dvr[t], x3);
                    }
                }
                cur[ob][2 * nbl] = pack2<T>(x3[0], x3[1]);
                cur[ob][2 * nbl + 1] = pack2<T>(x3[2], x3[3]);
            }
        }
        if (has_b) {
            u32x4 xw[G::NOB][G::NDVK];
#pragma unroll
            for (int ob = 0; ob < G::NOB; ob++) {
#pragma unroll
                for (int h = 0; h < G::NHIST; h++) xw[ob][h] = hist[ob][h];
                xw[ob][G::NDVK - 1] = cur[ob];
            }
            phase_b(cb_b, xw);
        }
    };

    // A group past the last computed one (gi >= ngc): its X3 pair lies wholly beyond the last upsampled column an output of the plane
    // reads (and beyond the sign tensor): the pair is zero, what remains is the bookkeeping of group() around the down-x pass.
    auto tail_group = [&](int gi, u32x4 (&hist)[G::NOB][G::NHIST], u32x4 (&cur)[G::NOB], u32x4 (&skw)[G::NOB]) __attribute__((always_inline)) {
        const bool has_b = gi >= G::NDVK - 1 && (gi - (G::NDVK - 1)) % (DOWN / 2) == 0;
        const int cb_b = (gi - (G::NDVK - 1)) / (DOWN / 2);
        {
            const int gp = gi - 1;
            const bool prv_b = gp >= G::NDVK - 1 && (gp - (G::NDVK - 1)) % (DOWN / 2) == 0;
            const int cb_p = (gp - (G::NDVK - 1)) / (DOWN / 2);
            if (prv_b && ((cb_p & 3) == 3 || cb_p == ncb - 1)) flush(cb_p);
        }
        if (EPI & 2) {
            if (has_b && (cb_b & 1) == 0) park_skip(skw);
            const int gn = gi + 1;
            const bool nxt_b = gn >= G::NDVK - 1 && (gn - (G::NDVK - 1)) % (DOWN / 2) == 0;
            const int cb_n = (gn - (G::NDVK - 1)) / (DOWN / 2);
            if (nxt_b && (cb_n & 1) == 0 && cb_n < ncb) load_skip(cb_n, skw);
        }
#pragma unroll
        for (int ob = 0; ob < G::NOB; ob++) cur[ob] = (u32x4){0u, 0u, 0u, 0u};
        if (has_b) {
            u32x4 xw[G::NOB][G::NDVK];
#pragma unroll
            for (int ob = 0; ob < G::NOB; ob++) {
#pragma unroll
                for (int h = 0; h < G::NHIST; h++) xw[ob][h] = hist[ob][h];
                xw[ob][G::NDVK - 1] = cur[ob];
            }
            phase_b(cb_b, xw);
        }
    };

    auto run_strip = [&](auto exact_c, auto lasty_c) __attribute__((always_inline)) -> bool {
        float amax = 0.f;
        unsigned anyc = 0;
        if (decltype(exact_c)::value) psum = 0.f;
        u32x4 raw[G::NMB];
        unsigned sg[G::NBG][RA ? 1 : 2][NL];
        load_piece(0, raw);
        park_piece(0, raw);                          // (the one exposed round trip of the strip)
        load_piece(1, raw);
        if (SIGN == AFCM_SIGNS_WRITE) {
            // The wait for a piece (park_piece) names the number of YOUNGER memory operations it may leave in flight (vmcnt is one
            // in-order counter for loads and stores): at least the sign stores of the group that ran since.  The loop head is also
            // reachable straight from here, with nothing behind the request -- for the compiler's count, not in fact -- and the
            // stricter state wins: it emitted vmcnt(2 / 1 / 0), i.e. every park also waited for the sign stores issued a moment
            // before.  As many stores behind the first request as a group issues (out of range: dropped by the memory pipeline,
            // counted like any other) make the two states equal.
            constexpr int NVW = decltype(lasty_c)::value ? G::NVB : G::OWN_VB;
#pragma unroll
            for (int i = 0; i < G::NBG * cdiv(NVW, 4); i++) __builtin_amdgcn_raw_buffer_store_b32(0u, rsx, kOut + 256u * i, 0, AFCM_WAVE_STORE_AUX);   // (distinct, non-adjacent offsets: identical or adjacent stores are merged)
        }
        if (RD) {
            load_signs(0, sg[0]);
            load_signs(1, sg[1]);
        }
        u32x4 hist[G::NOB][G::NHIST], cur[G::NOB], skw[G::NOB];
#pragma unroll
        for (int ob = 0; ob < G::NOB; ob++) {
            skw[ob] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
            for (int h = 0; h < G::NHIST; h++) hist[ob][h] = (u32x4){0u, 0u, 0u, 0u};
        }
        // the first down-x pass runs in group NDVK - 1 >= 1: when that is group 1 its skip rows are requested here (group 0 does
        // it itself for later ones)
        static_assert(G::NDVK - 1 >= 1, "the first down-x pass must not be in group 0");
        if constexpr (G::NHIST == 1) {
            // down 2: the pair a group produces is the next group's history -- two groups per iteration with the two register sets
            // trading places instead of a copy per group (8 v_mov), and the parity of gi (which groups park a piece) static
            typedef u32x4 (&as_hist)[G::NOB][1];
            u32x4 alt[G::NOB];
#pragma unroll
            for (int ob = 0; ob < G::NOB; ob++) alt[ob] = hist[ob][0];
#pragma unroll 1
            for (int gi = 0; gi < ngc; gi += 2) {
                group(gi, exact_c, lasty_c, raw, sg, reinterpret_cast<as_hist>(alt), cur, skw, amax, anyc);
                if (gi + 1 >= ngc) break;
                group(gi + 1, exact_c, lasty_c, raw, sg, reinterpret_cast<as_hist>(cur), alt, skw, amax, anyc);
            }
            // (the register sets keep trading places by the parity of the group)
#pragma unroll 1
            for (int gi = ngc; gi < ng; gi++) {
                if ((gi & 1) == 0) tail_group(gi, reinterpret_cast<as_hist>(alt), cur, skw);
                else tail_group(gi, reinterpret_cast<as_hist>(cur), alt, skw);
            }
        } else {
#pragma unroll 1
            for (int gi = 0; gi < ng; gi++) {
                if (gi < ngc) group(gi, exact_c, lasty_c, raw, sg, hist, cur, skw, amax, anyc);
                else tail_group(gi, hist, cur, skw);
#pragma unroll
                for (int ob = 0; ob < G::NOB; ob++) {
#pragma unroll
                    for (int h = 0; h + 1 < G::NHIST; h++) hist[ob][h] = hist[ob][h + 1];
                    hist[ob][G::NHIST - 1] = cur[ob];
                }
            }
        }
        flush(ncb - 1);                                   // the last group completed the last column block
        if (RD) return __builtin_amdgcn_ballot_w64((anyc & 0xaaaaaaaau) != 0) != 0;   // a clamped element in reach
        return __builtin_amdgcn_ballot_w64(!(amax <= cthr1)) != 0;                                        // NaN takes the exact path
    };
#ifdef AFCM_WAVE_STAMPS_ON
    const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // (READ never writes codes: one LASTY variant suffices)
    bool exact;
    if (SIGN == AFCM_SIGNS_WRITE && lastY) {
        exact = run_strip(std::false_type{}, std::true_type{});
        if (__builtin_expect(exact, 0)) run_strip(std::true_type{}, std::true_type{});
    } else {
        exact = run_strip(std::false_type{}, std::false_type{});
        if (__builtin_expect(exact, 0)) run_strip(std::true_type{}, std::false_type{});
    }
#ifdef AFCM_WAVE_STAMPS_ON
    {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        const unsigned long long st_c1 = __builtin_amdgcn_s_memtime(), st_r1 = __builtin_amdgcn_s_memrealtime();
        constexpr int slot = (UP == 4 ? 1 : 0) | (DOWN == 4 ? 2 : 0) | ((SIGN & 3) << 2) | (TOH == 48 ? 16 : 0);
        if (lane == 0) {
            atomicAdd(&afcm_wave_stamp_acc[slot][0], st_c1 - st_c0);
            atomicAdd(&afcm_wave_stamp_acc[slot][1], st_r1 - st_r0);
            atomicAdd(&afcm_wave_stamp_acc[slot][2], 1ull);
        }
    }
#endif
    // optional per-strip flag: the strip's activations could reach the clamp (a plane with no flagged strip is positively homogeneous
    // of degree 1 in its input: the caller derives <dL/dy, y> from <g, z>, afcm_plane_dot_gated_ld).  Every slot is written.
    if (!RD && p.clamp_flags != nullptr && lane == 0) p.clamp_flags[(size_t)plane * (p.tilesX * p.tilesY) + ty * p.tilesX + tx] = exact ? 1 : 0;

    if ((EPI & 4) && p.plane_sum != nullptr) {
        // one plain store into this tile's slot (no atomics: deterministic)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) psum += __shfl_down(psum, off, 64);
        if (lane == 0) p.plane_sum[(size_t)plane * (p.tilesX * p.tilesY) + ty * p.tilesX + tx] = psum;
    }
}

// ---------------------------------------------------------------------------------------------
template <typename T, int UP, int DOWN, int TOW, int TOH>
int launch_wave_tile(const afcm_filtered_lrelu_args* a, FlreluMfmaParams p, hipStream_t st) {
    const long long tiles = p.total_tiles;
    dim3 grid((unsigned)((tiles + 3) / 4)), block(256);
    const bool scale = p.oscale != nullptr || p.oscale2 != nullptr, sums = p.plane_sum != nullptr;
#define AFCM_WAVE_LAUNCH(SIGN) do { \
        if (sums && p.skip != nullptr) hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 7>), grid, block, 0, st, p); \
        else if (sums) hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 5>), grid, block, 0, st, p); \
        else if (p.skip != nullptr) hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 3>), grid, block, 0, st, p); \
        else if (scale) hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 1>), grid, block, 0, st, p); \
        else hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 0>), grid, block, 0, st, p); } while (0)
    // sign-reading calls are the transposed op: there is no encoder skip to add (forward-only operand), so those variants are not built
#define AFCM_WAVE_LAUNCH_RD(SIGN) do { \
        AFCM_REQUIRE(p.skip == nullptr, "filtered_lrelu: a skip operand next to a sign tensor to READ is not supported by the wave kernels"); \
        if (sums) hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 5>), grid, block, 0, st, p); \
        else if (scale) hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 1>), grid, block, 0, st, p); \
        else hipLaunchKernelGGL((flrelu_wave_kernel<T, UP, DOWN, TOW, TOH, SIGN, 0>), grid, block, 0, st, p); } while (0)
    switch (a->sign_mode) {
        case AFCM_SIGNS_NONE: AFCM_WAVE_LAUNCH(AFCM_SIGNS_NONE); break;
        case AFCM_SIGNS_WRITE: AFCM_WAVE_LAUNCH(AFCM_SIGNS_WRITE); break;
        default:
            if (p.read_aligned) AFCM_WAVE_LAUNCH_RD(kSignsReadAligned);
            else AFCM_WAVE_LAUNCH_RD(AFCM_SIGNS_READ);
            break;
    }
#undef AFCM_WAVE_LAUNCH
#undef AFCM_WAVE_LAUNCH_RD
    return hip_status(hipGetLastError());
}

template <typename T, int UP, int DOWN>
int prepare_wave(const afcm_filtered_lrelu_args* a, int py0_frag, int dshift, hipStream_t st) {
    typedef WaveGeom<UP, DOWN, 32, 32> G;
    const float gain_total = (float)a->up * (float)a->up * a->gain;
    hipLaunchKernelGGL((flrelu_wave_prepare_kernel<T, UP, DOWN>), dim3(cdiv((G::NLIN + G::NDVK) * 512, 256)), dim3(256), 0, st,
                       (char*)a->workspace, a->fu, a->fd, py0_frag, a->flip_filter, gain_total, a->slope, dshift);
    return hip_status(hipGetLastError());
}

#define AFCM_WAVE_INST(T, UP, DOWN, TOW, TOH) \
    template int launch_wave_tile<T, UP, DOWN, TOW, TOH>(const afcm_filtered_lrelu_args*, FlreluMfmaParams, hipStream_t);
#define AFCM_WAVE_INST_T(T)           \
    AFCM_WAVE_INST(T, 2, 2, 64, 32)   \
    AFCM_WAVE_INST(T, 2, 2, 64, 48)   \
    AFCM_WAVE_INST(T, 2, 4, 32, 32)   \
    AFCM_WAVE_INST(T, 4, 2, 64, 32)   \
    template int prepare_wave<T, 2, 2>(const afcm_filtered_lrelu_args*, int, int, hipStream_t); \
    template int prepare_wave<T, 2, 4>(const afcm_filtered_lrelu_args*, int, int, hipStream_t); \
    template int prepare_wave<T, 4, 2>(const afcm_filtered_lrelu_args*, int, int, hipStream_t);
// one translation unit per element type (Makefile: filtered_lrelu_wave.o = bf16, filtered_lrelu_wave_f16.o = f16): the two halves
// compile in parallel
#ifdef AFCM_WAVE_F16
AFCM_WAVE_INST_T(f16_t)
#else
AFCM_WAVE_INST_T(bf16_t)
#endif

}  // namespace afcm
