// Panelled TN GEMM on bf16 MFMA: the weight-gradient kernel (SURVEY.md §2b K2 wgrad, K9).
//
//   dW[set][p][n][c] += sum_{r in set} Y[r, n] * X[xrow(r) + shift_p, coff_p + c]
//
// Y = output cotangent (NHWC + zero halo, flat rows), X = saved forward activation.  Because
// Y's halo rows are zero, the sum runs over the FLAT padded row range -- a plain GEMM whose
// reduction dimension is the row index, nine row-shifted panels for a 3x3 filter, no pixel
// decode.  The dual-cotangent backward (g_x and g_a in one pass) is the `set` dimension: both
// cotangent sets read the same saved X (x_set_rows = 0) when the forward was shared (SISS), or
// their own rows (SISS-No-IS).
//
// Both operands arrive row-major with the REDUCTION index on rows, so MFMA fragments need a
// transposed read: tiles are staged [64 rows][128 ch] (256-B rows) by global_load_lds_dwordx4
// and read with ds_read_b64_tr_b16.  16-B chunk index XOR ((row&3)<<2 | (row>>2)&3) -- on the
// DMA source address and on the read -- makes every 32-lane half of a transposed read cover
// all 64 banks exactly once.
// Split-K over row ranges; partial tiles are accumulated with f32 global atomics (no-return
// global_atomic_add_f32; 16 consecutive floats per lane group).
#include "common.h"
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

namespace {

constexpr int BN = 128, BC = 128, BR = 64;   // output tile 128(n) x 128(c); 64 reduction rows / step
constexpr int kYTile = BR * 256;             // 16 KiB
constexpr int kMaxPanels = 9;

struct TNParams {
    const bf16_t* Y; const bf16_t* X; float* dW; const bf16_t* zero_page;
    float* dbias; float* dbias2;          // optional: column sums of Y per set (bias gradients)
    long ldy, ldx, set_stride;
    long bias_stride;                     // floats between the sets of dbias / dbias2 (= set_stride unless siss_gemm_tn_bs says otherwise)
    long x_set_rows;
    int N, C, npanels, nsets, nsplits, rmw;
    int rows_per_set, row_begin, row_end, rows_per_split;
    int shift[kMaxPanels];
    int coff[kMaxPanels];
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)l, 16, 0, 0);
}
__device__ __forceinline__ s16x4_t tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}
__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }

// TAPS = 1: one panel per block (1x1 convs, linears, stride-2 panels with channel offsets).
// TAPS = 3: the three kx taps of one filter row share the block: their X rows are the SAME rows shifted
//           by one, so the X tile is staged once with two extra rows and read at row offsets 0/1/2, and
//           the Y tile is staged once for three products -- a third of the DMA instructions, HBM/L2 bytes
//           and barriers per MFMA, and two thirds of the LDS reads.
template <int TAPS, bool ILV = false>
struct TCfg {
    static constexpr int kWaves = TAPS == 3 ? 8 : 4;              // 3 taps: 8 waves of 64(n) x 32(c) keep 96 acc VGPRs
    static constexpr int kThreads = kWaves * 64;
    static constexpr int kCT = TAPS == 3 ? 2 : 4;                 // 16-wide c-tiles per wave
    static constexpr int kPieces = 16 / kWaves;                   // 4-row DMA pieces per wave and operand
    static constexpr int kXRows = BR + (TAPS == 3 ? 4 : 0);       // 64 (+ one extra 4-row DMA piece)
    static constexpr int kStageBytes = kYTile + kXRows * 256;
    static constexpr int kStages = TAPS == 3 ? (ILV ? 4 : 3) : 2; // 3 taps: one block per CU -> room for a 3-deep ring (4 in the interleaved variant)
    static constexpr int kSmemBytes = kStages * kStageBytes;
};

// The whole product of one block.  bid / nwg: the block's index and the block count of ITS launch -- or, in a grouped launch
// (gemm_tn_grouped_kernel), of its job.
// tid: the thread's index within ITS (virtual) block -- threadIdx.x, or threadIdx.x & 255 for the one-tap halves of a 512-thread block of
// the mixed kernel.  PAIRED (one-tap body only): two virtual blocks share a workgroup, hence its barriers -- every block then runs
// the SAME number of barriers (the full split's step count; the steps it does not have are barrier-only) and never returns early.
// nvalid: the job's REAL block count when nwg was rounded up (the mixed kernel keeps block counts at multiples of 8 so that virtual
// and physical blocks agree on their XCD); logical blocks past it have no work.
template <int TAPS, bool ILV = false, bool PAIRED = false>
__device__ __forceinline__ void tn_body(const TNParams& p, int bid, const int nwg, char* smem, const int tid, const int nvalid = 1 << 30) {
    using C_ = TCfg<TAPS, ILV>;
    constexpr int NP = C_::kPieces, CT = C_::kCT;
    const int lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = TAPS == 3 ? w >> 2 : w >> 1, wc = TAPS == 3 ? w & 3 : w & 1;   // wave tile 64(n) x (CT*16)(c)
    // Grid is 1-D.  Logical order: panel group fastest, then tile, then split, then set -- and each XCD
    // (blocks b, b+8, ... share an L2) gets a CONTIGUOUS run of logical blocks, so the blocks that
    // stream the same Y / X rows run together on one XCD and their re-reads hit in L2
    // (measured before this remap: 5 % L2 hit rate, 9x the operand bytes from HBM).
    const int tiles_c = (p.C + BC - 1) / BC, tiles_n = (p.N + BN - 1) / BN;
    const int ngroups = p.npanels / TAPS;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const bool dead = bid >= nvalid;                   // (a bijection of [0, nwg): the surplus of a rounded-up count decodes to no block)
    if (dead) {
        if (!PAIRED) return;
        bid = 0;                                       // any valid decode: it runs no step, only the workgroup's barriers
    }
    const int pn = (bid % ngroups) * TAPS; bid /= ngroups;
    const int tile = bid % (tiles_c * tiles_n); bid /= tiles_c * tiles_n;
    const int split = bid % p.nsplits;
    const int set = bid / p.nsplits;
    const int tn = tile / tiles_c, tc = tile - tn * tiles_c;
    const int n0 = tn * BN, c0 = tc * BC;
    const int r0 = p.row_begin + split * p.rows_per_split;
    int r1 = r0 + p.rows_per_split; r1 = r1 < p.row_end ? r1 : p.row_end;
    if (!PAIRED && r0 >= r1) return;
    const int steps = (r0 < r1 && !dead) ? (r1 - r0 + BR - 1) / BR : 0;

    // staging: 4 pieces of 4 rows (256 B each) per wave and operand (+ one extra X piece on wave 0)
    const bf16_t* ysrc[NP];
    const bf16_t* xsrc[NP];
    const bf16_t* xsrc_extra = nullptr;
    const bf16_t* zsrc = p.zero_page + (lane & 15) * 8;
    // trow?[j]: the piece's row within the step, or kNever for lanes whose 16-B column chunk lies past the operand's
    // last column (N < 128 / C < 128 tiles): those lanes take the zero page, so no byte past a row's end is ever read.
    constexpr int kNever = 1 << 29;
    int trowy[NP], trowx[NP];
    const long xrow0 = (long)set * p.x_set_rows + r0 + p.shift[pn];
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = (w * NP + j) * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ swz(row);
        trowy[j] = n0 + lc * 8 < p.N ? row : kNever;
        trowx[j] = c0 + lc * 8 < p.C ? row : kNever;
        const long ry = (long)set * p.rows_per_set + r0 + row;
        ysrc[j] = p.Y + ry * p.ldy + n0 + lc * 8;
        xsrc[j] = p.X + (xrow0 + row) * p.ldx + p.coff[pn] + c0 + lc * 8;
    }
    int trow_extra = kNever;
    if (TAPS == 3) {
        const int row = BR + (lane >> 4);
        const int lc = (lane & 15) ^ swz(row);
        xsrc_extra = p.X + (xrow0 + row) * p.ldx + p.coff[pn] + c0 + lc * 8;
        if (c0 + lc * 8 < p.C) trow_extra = row;
    }
    const long ystep = (long)BR * p.ldy, xstep = (long)BR * p.ldx;
    const unsigned smem_a = lds_addr(smem);
    // Fast form of stage() for the steps whose rows all lie inside [r0, r1) of a tile with all 128 + 128 columns (every step but
    // the last one or two of full tiles): wave-uniform 64-bit base per step + per-lane 32-bit offsets fixed for the block, so a
    // piece is s_mov m0 + the load.  The general form below spends, per piece, a compare, two selects against the zero page, a
    // 64-bit add and an m0 save / restore -- ~65 instructions per step in waves that also issue 48 MFMAs and 40 LDS reads.
    const bool full_tile = n0 + BN <= p.N && c0 + BC <= p.C;
    unsigned yoff32[NP], xoff32[NP], xoff32_extra = 0;
    const bf16_t* const ybase0 = p.Y + ((long)set * p.rows_per_set + r0) * p.ldy + n0;
    const bf16_t* const xbase0 = p.X + xrow0 * p.ldx + p.coff[pn] + c0;
#pragma unroll
    for (int j = 0; j < NP; ++j) {
        const int row = (w * NP + j) * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ swz(row);
        yoff32[j] = (unsigned)(((long)row * p.ldy + lc * 8) * 2);
        xoff32[j] = (unsigned)(((long)row * p.ldx + lc * 8) * 2);
    }
    if (TAPS == 3) {
        const int row = BR + (lane >> 4);
        xoff32_extra = (unsigned)(((long)row * p.ldx + ((lane & 15) ^ swz(row)) * 8) * 2);
    }
    auto stage_fast = [&](int buf, int step) {
        const unsigned base = smem_a + buf * C_::kStageBytes + (w * NP * 4) * 256;
        const bf16_t* yb = ybase0 + step * ystep;
        const bf16_t* xb = xbase0 + step * xstep;
#pragma unroll
        for (int j = 0; j < NP; ++j) glds16_saddr(yoff32[j], yb, base + j * 1024);
#pragma unroll
        for (int j = 0; j < NP; ++j) glds16_saddr(xoff32[j], xb, base + kYTile + j * 1024);
        if (TAPS == 3 && w == 0) glds16_saddr(xoff32_extra, xb, smem_a + buf * C_::kStageBytes + kYTile + BR * 256);
    };
    auto stage = [&](int buf, int step) {
        const unsigned base = smem_a + buf * C_::kStageBytes + (w * NP * 4) * 256;
        const int rbase = r0 + step * BR;
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            const bool ok = rbase + trowy[j] < r1;
            glds16_asm(ok ? ysrc[j] + step * ystep : zsrc, base + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < NP; ++j) {
            // an X row matters only if one of the (up to TAPS) Y rows it meets is in range; everything
            // else comes from the zero page (never read past the operand; 0 * garbage could be NaN)
            const bool ok = rbase + trowx[j] - (TAPS - 1) < r1;
            glds16_asm(ok ? xsrc[j] + step * xstep : zsrc, base + kYTile + j * 1024);
        }
        if (TAPS == 3 && w == 0) {
            const bool ok = rbase + trow_extra - (TAPS - 1) < r1;
            glds16_asm(ok ? xsrc_extra + step * xstep : zsrc, smem_a + buf * C_::kStageBytes + kYTile + BR * 256);
        }
    };

    f32x4_t acc[TAPS][4][CT];   // [tap][n-tile][c-tile]
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) acc[t][i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Bias gradient for free: column sums of Y are one more product, Y^T . 1, taken by the waves that own
    // c-tile 0 of panel group 0 (every (set, split, n-tile) exactly once).
    const bool do_bias = p.dbias != nullptr && pn == 0 && tc == 0 && wc == 0;
    f32x4_t bacc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) bacc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16x8_t ones = bf16x8_t{0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};

    // transposed-read addresses: 16-lane group g covers reduction rows 8g..8g+7 of a 32-row
    // k-step in two 4-row blocks (h); lane 4q+pp of the group addresses row q, columns 4pp..4pp+3.
    // For tap t the X rows are shifted by t; the swizzle is a function of the PHYSICAL row, and
    // (row + 32) has the same swizzle, so kk adds a plain 8192 bytes.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int y_off[2], x_off[TAPS][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + q;
        y_off[h] = row * 256 + ((((wn * 8) | (pp >> 1)) ^ swz(row)) << 4) + 8 * (pp & 1);
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int xr = row + t;
            x_off[t][h] = kYTile + xr * 256 + ((((wc * CT * 2) | (pp >> 1)) ^ swz(xr)) << 4) + 8 * (pp & 1);
        }
    }

    auto load_frags = [&](bf16x8_t (&yf)[4], bf16x8_t (&xf)[TAPS][CT], const char* sb, int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s16x4_t a0 = tr_read(sb + ((y_off[0] ^ (i << 5)) + kk * 8192));
            s16x4_t a1 = tr_read(sb + ((y_off[1] ^ (i << 5)) + kk * 8192));
            yf[i] = bf16x8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
        }
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int i = 0; i < CT; ++i) {
                s16x4_t b0 = tr_read(sb + ((x_off[t][0] ^ (i << 5)) + kk * 8192));
                s16x4_t b1 = tr_read(sb + ((x_off[t][1] ^ (i << 5)) + kk * 8192));
                xf[t][i] = bf16x8_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
    };
    auto mma_tap = [&](bf16x8_t (&yf)[4], bf16x8_t (&xf)[TAPS][CT], int t) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j)
                acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[t][j], acc[t][i][j], 0, 0, 0);
    };

    if constexpr (TAPS == 3) {
        // Ring of three stage buffers, one barrier per K-step, and the fragment reads run one HALF-step ahead of
        // the MFMAs in registers -- across the barrier too: the barrier at the bottom of step s comes after every
        // wave's vmcnt(0) for stage s+2, so stage s+1 (landed one barrier earlier) may be read before it.
        // Without this each wave exposed an LDS round trip ~6 times per step, and both waves of a SIMD (same
        // block, same barrier) exposed it at the same time.
        constexpr int SB = C_::kStageBytes;
        bf16x8_t yf[2][4], xf[2][TAPS][CT];
        // Static priority for the second-dispatched half of the block (waves 4-7 share their SIMDs with waves 0-3 and lose every
        // issue arbitration by age: MI355X_MICROARCH.md, 'Two waves per SIMD', item 4)
        if (w >= 4) __builtin_amdgcn_s_setprio(1);
        stage(0, 0);
        if (steps > 1) stage(1, 1);
        if (ILV && steps > 2) stage(2, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load_frags(yf[0], xf[0], smem, 0);
        int buf = 0;
        if constexpr (ILV) {
            // Interleaved variant: the 20 transposed reads of the NEXT half-step are issued one per MFMA under the 24
            // MFMAs of this half-step (sched_group_barrier pipeline), in the order the next half-step consumes them, instead
            // of as one burst between two MFMA bursts -- the two waves of a SIMD are phase-locked by the step barrier, so a
            // burst of reads leaves the matrix pipe idle in BOTH of them.
            // One half-step: 24 MFMAs (taps 2, 1, 0) with ONE transposed read of the next half-step's fragments issued
            // behind each of the first 20, pinned by a scheduling fence per pair.  Read order = consumption order (Y
            // fragments, then taps 2, 1, 0), so the counted LDS wait in front of the next half-step's first MFMA leaves the
            // younger reads in flight.  Past the last step the reads fetch a stale buffer: harmless, never used.
            // Read addresses: the 20 per-lane fragment addresses are held for a PAIR of ring buffers (anchor = buffer 0 or 2; the
            // odd buffer of the pair and the second 32-row half are the instruction's immediate offset -- the loop is unrolled over
            // the ring, so the buffer is a compile-time constant) and rebuilt when the reads move to the other pair: twice per four
            // steps, as (per-lane offset + anchor) ^ tile bits -- 20 vector instructions each (the XOR may follow the add: anchors
            // and the LDS base are multiples of 256 bytes).  Before this every read was preceded by a v_add for the buffer base: 40
            // vector instructions per step beside 48 MFMAs -- with two waves per SIMD the vector issue (an MFMA holds it for 8 of
            // its 16 cycles, a v_add for 4) was as busy as the matrix pipe (rocprofv3: 2.0 vector instructions per MFMA,
            // profiles/r03r_pmc_mix_celeb.txt).
            typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4;
            if (smem_a & 255u) __builtin_trap();
            unsigned aY[2][4], aX[TAPS][2][CT];
            auto set_anchor = [&](unsigned anchor) {
                asm volatile("" : "+s"(anchor));                         // opaque: keeps LICM from holding BOTH anchors' sets in registers
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const unsigned yb = smem_a + anchor + y_off[h];
#pragma unroll
                    for (int i = 0; i < 4; ++i) aY[h][i] = yb ^ (i << 5);
#pragma unroll
                    for (int t = 0; t < TAPS; ++t) {
                        const unsigned xb = smem_a + anchor + x_off[t][h];
#pragma unroll
                        for (int ii = 0; ii < CT; ++ii) aX[t][h][ii] = xb ^ (ii << 5);
                    }
                }
            };
            set_anchor(0);
            // BN_ = ring buffer the NEXT half-step's fragments are read from, KK = its 32-row half
            auto half = [&](auto BN_, auto KK, bf16x8_t (&yc)[4], bf16x8_t (&xc)[TAPS][CT], bf16x8_t (&yn)[4], bf16x8_t (&xn)[TAPS][CT]) {
                constexpr int bn = decltype(BN_)::value, kkn = decltype(KK)::value;
                constexpr int imm = (bn & 1) * SB + kkn * 8192;
                s16x4_t r0;                                              // the first read of the pair in flight
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yc[i], ones, bacc[i], 0, 0, 0);
                }
#pragma unroll
                for (int m = 0; m < 24; ++m) {
                    const int t = 2 - m / 8, i = (m % 8) / CT, j = m % CT;
                    acc[t][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yc[i], xc[t][j], acc[t][i][j], 0, 0, 0);
                    if (m < 20) {
                        // read m: pair (m >> 1) = fragment, m & 1 = its 4-row half; the fragment is assembled the moment its
                        // second half is issued, so that both reads land in the fragment's own registers (no copies)
                        s16x4_t rr;
                        if (m < 8) {
                            rr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(aY[m & 1][m >> 1] + imm));
                        } else {
                            const int q = m - 8, tt = 2 - q / 4, ii = (q % 4) >> 1, h = q & 1;
                            rr = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(aX[tt][h][ii] + imm));
                        }
                        if (!(m & 1)) {
                            r0 = rr;
                        } else {
                            const bf16x8_t f = bf16x8_t{r0[0], r0[1], r0[2], r0[3], rr[0], rr[1], rr[2], rr[3]};
                            if (m < 8) yn[m >> 1] = f;
                            else { const int q = m - 9; xn[2 - q / 4][(q % 4) >> 1] = f; }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // Ring of FOUR stages here: stage s+3 is issued during step s and only has to have landed at the bottom of step
            // s+1, so the wait at the bottom of a step is COUNTED (this step's own DMA stays in flight).  The two waves of a
            // SIMD (w and w + 4) issue their DMA half a step apart: a DMA piece costs 60-180 issue cycles during which the
            // issuing wave feeds no MFMAs -- staggered, the SIMD's other wave keeps the matrix pipe busy meanwhile.
            const bool early = w < 4;
            using std::integral_constant;
            auto step = [&](auto B_, int s) {                            // B_ = the ring buffer of step s (= s mod 4)
                constexpr int b = decltype(B_)::value, b1 = (b + 1) & 3, b3 = (b + 3) & 3;
                const bool more = s + 3 < steps;
                // (all rows of step s + 3 in range: its Y rows and the X rows two past them)
                const bool fast = full_tile && r0 + (s + 4) * BR + 4 <= r1;
                if (more && early) { if (fast) stage_fast(b3, s + 3); else stage(b3, s + 3); }
                __builtin_amdgcn_sched_barrier(0);
                half(integral_constant<int, b>{}, integral_constant<int, 1>{}, yf[0], xf[0], yf[1], xf[1]);
                if (more && !early) { if (fast) stage_fast(b3, s + 3); else stage(b3, s + 3); }
                if constexpr (b & 1) set_anchor((b1 >> 1) * 2 * SB);      // the second half reads the other pair's even buffer
                __builtin_amdgcn_sched_barrier(0);
                half(integral_constant<int, b1>{}, integral_constant<int, 0>{}, yf[1], xf[1], yf[0], xf[0]);
                if (more) {
                    if (w == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP + 1) : "memory");
                    else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NP) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                __builtin_amdgcn_s_barrier();
            };
            for (int s = 0; s < steps; s += 4) {
                step(integral_constant<int, 0>{}, s);
                if (s + 1 < steps) step(integral_constant<int, 1>{}, s + 1);
                if (s + 2 < steps) step(integral_constant<int, 2>{}, s + 2);
                if (s + 3 < steps) step(integral_constant<int, 3>{}, s + 3);
            }
        } else
        for (int s = 0; s < steps; ++s) {
            const int b1 = buf + 1 == 3 ? 0 : buf + 1, b2 = b1 + 1 == 3 ? 0 : b1 + 1;
            if (s + 2 < steps) {
                if (full_tile && r0 + (s + 3) * BR + 4 <= r1) stage_fast(b2, s + 2); else stage(b2, s + 2);
            }
            const char* sb = smem + buf * SB;
            const char* sbn = smem + b1 * SB;
            // half-step kk = 0
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[0][i], ones, bacc[i], 0, 0, 0);
            }
            // taps in REVERSE load order: the first MFMAs need the last-loaded fragment, so the compiler's wait
            // there is lgkmcnt(0) and no older read is outstanding when the next 20 are issued (lgkmcnt counts
            // to 15 only; a capped wait would force some of the NEW reads to land before the MFMAs below)
            mma_tap(yf[0], xf[0], 2);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(yf[1], xf[1], sb, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(yf[0], xf[0], 1);
            mma_tap(yf[0], xf[0], 0);
            __builtin_amdgcn_sched_barrier(0);
            // half-step kk = 1
            if (do_bias) {
#pragma unroll
                for (int i = 0; i < 4; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[1][i], ones, bacc[i], 0, 0, 0);
            }
            mma_tap(yf[1], xf[1], 2);
            __builtin_amdgcn_sched_barrier(0);
            if (s + 1 < steps) load_frags(yf[0], xf[0], sbn, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_tap(yf[1], xf[1], 1);
            mma_tap(yf[1], xf[1], 0);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            buf = b1;
        }
    } else {
        // One barrier per K-step (see gemm_nt.hip): wait own DMA + own LDS reads, barrier, restage, compute.
        // (the loop runs two steps per trip so that the stage buffer -- and with it every fragment read's offset from the 16 per-lane
        // addresses: buffer * 32 KiB + half * 8 KiB -- is an instruction immediate instead of a vector add per read)
        typedef __attribute__((address_space(3))) s16x4_t* lds_s16x4;
        unsigned aY[2][4], aX1[2][CT];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int i = 0; i < 4; ++i) aY[h][i] = smem_a + (y_off[h] ^ (i << 5));
#pragma unroll
            for (int i = 0; i < CT; ++i) aX1[h][i] = smem_a + (x_off[0][h] ^ (i << 5));
        }
        auto frag_at = [&](unsigned a0, unsigned a1, auto IMM) {
            constexpr int imm = decltype(IMM)::value;
            const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(a0 + imm));
            const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(uintptr_t)(a1 + imm));
            return bf16x8_t{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        };
        auto one = [&](auto BUF, int s) {
            constexpr int buf = decltype(BUF)::value;
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + 1 < steps) {
                if (full_tile && r0 + (s + 2) * BR + 4 <= r1) stage_fast(buf ^ 1, s + 1); else stage(buf ^ 1, s + 1);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t yf[4], xf[TAPS][CT];
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    yf[i] = kk ? frag_at(aY[0][i], aY[1][i], std::integral_constant<int, buf * C_::kStageBytes + 8192>{})
                               : frag_at(aY[0][i], aY[1][i], std::integral_constant<int, buf * C_::kStageBytes>{});
#pragma unroll
                for (int i = 0; i < CT; ++i)
                    xf[0][i] = kk ? frag_at(aX1[0][i], aX1[1][i], std::integral_constant<int, buf * C_::kStageBytes + 8192>{})
                                  : frag_at(aX1[0][i], aX1[1][i], std::integral_constant<int, buf * C_::kStageBytes>{});
                if (do_bias) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) bacc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], ones, bacc[i], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < TAPS; ++t) mma_tap(yf, xf, t);
            }
        };
        if (steps > 0) stage(0, 0);
        for (int s = 0; s < steps; s += 2) {
            one(std::integral_constant<int, 0>{}, s);
            if (s + 1 < steps) one(std::integral_constant<int, 1>{}, s + 1);
        }
        if constexpr (PAIRED) {
            // the other virtual block of this workgroup may have more steps (only the last split is short): keep its barriers company
            const int all = p.rows_per_split / BR;
            for (int s = steps; s < all; ++s) {
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
            if (steps == 0) return;
        }
    }

    if (do_bias && (lane & 15) == 0) {      // every column of bacc holds the same sums; lane&15 == 0 keeps column 0
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + i * 16 + g * 4 + r;
                if (n < p.N) {
                    atomicAdd(p.dbias + (long)set * p.bias_stride + n, bacc[i][r]);
                    if (p.dbias2) atomicAdd(p.dbias2 + (long)set * p.bias_stride + n, bacc[i][r]);
                }
            }
    }
    // acc[t][i][j][r]: n = n0 + wn*64 + i*16 + (lane>>4)*4 + r, c = c0 + wc*64 + j*16 + (lane&15)
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        float* out = p.dW + (long)set * p.set_stride + (long)(pn + t) * p.N * p.C;
        if (p.rmw) {
            // one split: this block OWNS the tile -- plain read-add-write (deterministic; float atomics execute at
            // the memory side at ~1.3 TB/s chip-wide and dominate the small layers).  All loads of a tap are issued
            // before the first store (the compiler cannot prove the addresses distinct and would serialise them).
            float old[4][CT][4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const int c = c0 + wc * CT * 16 + j * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + wn * 64 + i * 16 + g * 4 + r;
                        old[i][j][r] = (p.rmw == 1 && n < p.N && c < p.C) ? out[(long)n * p.C + c] : 0.f;   // rmw 2: overwrite
                    }
                }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < CT; ++j) {
                    const int c = c0 + wc * CT * 16 + j * 16 + (lane & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = n0 + wn * 64 + i * 16 + g * 4 + r;
                        if (n < p.N && c < p.C) out[(long)n * p.C + c] = old[i][j][r] + acc[t][i][j][r];
                    }
                }
            continue;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < CT; ++j) {
                const int c = c0 + wc * CT * 16 + j * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int n = n0 + wn * 64 + i * 16 + g * 4 + r;
                    if (n < p.N && c < p.C) atomicAdd(out + (long)n * p.C + c, acc[t][i][j][r]);
                }
            }
    }
}

template <int TAPS, bool ILV = false>
__global__ __launch_bounds__(TCfg<TAPS>::kThreads, 2) void gemm_tn_kernel(const TNParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    tn_body<TAPS, ILV>(p, blockIdx.x, gridDim.x, smem, threadIdx.x);
}

// Several independent products in ONE launch (the low-resolution weight gradients: each of them alone leaves most CUs idle
// and pays a launch's fixed latency; together they fill the chip).  Job j owns blocks [first[j], first[j+1]); first[] are
// multiples of 8 so that a job's blocks keep their XCD (block index mod 8) -- the surplus blocks of a job exit at once.
constexpr int kMaxJobs = 14;
struct TNGroup {
    int njobs;
    int first[kMaxJobs + 1];
    int nwg[kMaxJobs];
    TNParams job[kMaxJobs];
};
static_assert(sizeof(TNGroup) <= 4096, "kernel arguments are limited to 4 KiB");

template <int TAPS>
__global__ __launch_bounds__(TCfg<TAPS>::kThreads, 2) void gemm_tn_grouped_kernel(const TNGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int j = 0;
    while (j + 1 < g.njobs && (int)blockIdx.x >= g.first[j + 1]) ++j;
    const int bid = blockIdx.x - g.first[j];
    if (bid >= g.nwg[j]) return;
    tn_body<TAPS, TAPS == 3>(g.job[j], bid, g.nwg[j], smem, threadIdx.x);      // 3 taps: the interleaved variant (4-deep ring)
}

// The same with a CAPPED grid (siss_gemm_tn_grouped_capped): fewer workgroups than blocks, each walking the block list with the
// grid's stride (a multiple of 8: a workgroup's blocks keep its XCD) -- the launch then occupies at most that many CUs for its
// whole duration and leaves the rest of the chip to whatever runs beside it on another stream.  (A kernel of its own: the loop
// costs the 3-tap body registers -- 152 instead of 44 bytes of scratch per lane -- which the uncapped launches need not pay.)
template <int TAPS>
__global__ __launch_bounds__(TCfg<TAPS>::kThreads, 2) void gemm_tn_grouped_capped_kernel(const TNGroup g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int total = g.first[g.njobs];
    for (int vb = blockIdx.x; vb < total; vb += gridDim.x) {
        int j = 0;
        while (j + 1 < g.njobs && vb >= g.first[j + 1]) ++j;
        const int bid = vb - g.first[j];
        if (bid < g.nwg[j])
            tn_body<TAPS, TAPS == 3>(g.job[j], bid, g.nwg[j], smem, threadIdx.x);
        if (vb + (int)gridDim.x < total) {
            // the next block restages the LDS ring: every wave must be out of this block's reads (and its DMA landed) first
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __syncthreads();
            if (TAPS == 3) __builtin_amdgcn_s_setprio(0);
        }
    }
}

// A fused 3-tap product and a one-tap product in ONE launch of one round of blocks (siss_gemm_tn_pair): blocks [0, n3) run the 3-tap
// body, each of the nphys1 blocks behind them runs TWO virtual one-tap blocks (waves 0-3 and 4-7, a 64-KiB half of the LDS each:
// the residency the one-tap kernel has on its own, two 256-thread blocks per CU).  Why: a one-tap weight gradient at 256 x 256
// (a resnet's 1x1 conv_shortcut: 545 MB of cotangent + 545 MB of input for 140 GFLOP) is HBM-bound when launched alone (263 us
// at 522 TF/s); beside MFMA-bound 3-tap blocks it streams while they compute.
struct TNPair {
    TNParams j3, j1;
    int n3, nphys1;          // physical blocks of the two products (multiples of 8)
    int nv3, nv1;            // their real (virtual, for the one-tap product) block counts
};
static_assert(sizeof(TNPair) <= 4096, "kernel arguments are limited to 4 KiB");

__global__ __launch_bounds__(TCfg<3>::kThreads, 2) void gemm_tn_mixed_kernel(const TNPair g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int b = blockIdx.x;
    if (b < g.n3) {
        tn_body<3, true>(g.j3, b, g.n3, smem, threadIdx.x, g.nv3);
    } else {
        const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);
        // virtual block index lb + half * nphys1: nphys1 and n3 are multiples of 8, so it keeps the physical block's XCD (index mod 8)
        tn_body<1, false, true>(g.j1, (b - g.n3) + half * g.nphys1, 2 * g.nphys1, smem + half * TCfg<1>::kSmemBytes, threadIdx.x & 255, g.nv1);
    }
}

// HOST-side log of the products that OVERWROTE their output (rmw == 2): one record (first float, float count) per set, in launch
// order, drained by siss_gemm_tn_overwrite_log.  The engine learns from it which stretches of the gradient buffer a backward pass
// writes without reading -- those need no zero fill before the next step (UNetEngine.zero_grad(sparse_key=...)) -- and checks every
// such pass against what the fill assumed.  Bounded: with nobody draining it the log stops at kOwMax records and says so.
struct OwRec { const float* ptr; long floats; };
constexpr size_t kOwMax = 1 << 16;
std::mutex g_ow_mu;
std::vector<OwRec> g_ow;
bool g_ow_overflow = false;
void note_overwrite(const TNParams& p) {
    if (p.rmw != 2) return;
    std::lock_guard<std::mutex> lock(g_ow_mu);
    for (int s = 0; s < p.nsets; ++s) {
        if (g_ow.size() >= kOwMax) { g_ow_overflow = true; return; }
        g_ow.push_back({p.dW + (long)s * p.set_stride, (long)p.npanels * p.N * p.C});
    }
}

template <int TAPS>
int launch_tn(const TNParams& p, hipStream_t st) {
    using C_ = TCfg<TAPS>;
    static unsigned char attr_set[kMaxDevices], attr_ilv[kMaxDevices];
    siss_count_dispatch(TAPS == 3 ? SISS_K_TN3 : SISS_K_TN1);
    note_overwrite(p);
    dim3 grid(cdiv(p.N, BN) * cdiv(p.C, BC) * (p.npanels / TAPS) * p.nsets * p.nsplits);
    if constexpr (TAPS == 3) {                             // fragment reads interleaved with the MFMAs: measured +5 % (996 -> 1044, 1057 -> 1112 TF/s)
        constexpr int smem_ilv = TCfg<3, true>::kSmemBytes;
        if (siss_ensure_smem((const void*)gemm_tn_kernel<3, true>, smem_ilv, attr_ilv) != SISS_OK) return SISS_ERR_LAUNCH;
        gemm_tn_kernel<3, true><<<grid, C_::kThreads, smem_ilv, st>>>(p);
    } else {
        if (siss_ensure_smem((const void*)gemm_tn_kernel<TAPS>, C_::kSmemBytes, attr_set) != SISS_OK) return SISS_ERR_LAUNCH;
        gemm_tn_kernel<TAPS><<<grid, C_::kThreads, C_::kSmemBytes, st>>>(p);
    }
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

}  // namespace

// argument list of siss_gemm_tn as a struct (include/siss_hip.h declares the same layout)
struct siss_tn_job {
    const void* Y; long ldy; const void* X; long ldx; float* dW; long set_stride;
    int N, C, npanels, nsets, rows_per_set, row_begin, row_end, nsplits;
    long x_set_rows;
    const void* zero_page; float* dbias; float* dbias2;
    int shifts[9]; int coffs[9];
    long bias_set_stride;                 // floats between the sets of dbias / dbias2; 0 = set_stride (siss_gemm_tn_bs's extra argument)
};

static int tn_setup(const void* Y, long ldy, const void* X, long ldx, float* dW, long set_stride, int N, int C,
                    int npanels, const int* shifts, const int* coffs, int nsets, int rows_per_set,
                    long x_set_rows, int row_begin, int row_end, int nsplits, const void* zero_page,
                    float* dbias, float* dbias2, bool grouped, TNParams& p, bool& fused3_out) {
    SISS_CHECK_ARG(Y && X && dW && shifts && coffs && zero_page);
    SISS_CHECK_ARG(N > 0 && C > 0 && npanels >= 1 && npanels <= kMaxPanels && nsets >= 1);
    SISS_CHECK_ARG(ldy % 8 == 0 && ldx % 8 == 0 && C % 8 == 0);   // N may be ragged (masked at the store)
    SISS_CHECK_ARG(((uintptr_t)Y | (uintptr_t)X | (uintptr_t)zero_page) % 16 == 0 && (uintptr_t)dW % 4 == 0);
    SISS_CHECK_ARG(row_begin >= 0 && row_end > row_begin && row_end <= rows_per_set);
    p.Y = (const bf16_t*)Y; p.X = (const bf16_t*)X; p.dW = dW; p.zero_page = (const bf16_t*)zero_page;
    p.dbias = dbias; p.dbias2 = dbias ? dbias2 : nullptr;
    p.ldy = ldy; p.ldx = ldx; p.set_stride = set_stride; p.bias_stride = set_stride; p.x_set_rows = x_set_rows;
    p.N = N; p.C = C; p.npanels = npanels; p.nsets = nsets;
    p.rows_per_set = rows_per_set; p.row_begin = row_begin; p.row_end = row_end;
    for (int i = 0; i < kMaxPanels; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    for (int i = 0; i < npanels; ++i) SISS_CHECK_ARG(p.coff[i] % 8 == 0);
    // 3x3 filter rows: panels come in triples whose shifts are consecutive rows with equal channel offsets
    bool triples = npanels % 3 == 0;
    for (int g = 0; triples && g < npanels / 3; ++g)
        triples = p.shift[3 * g + 1] == p.shift[3 * g] + 1 && p.shift[3 * g + 2] == p.shift[3 * g] + 2 &&
                  p.coff[3 * g + 1] == p.coff[3 * g] && p.coff[3 * g + 2] == p.coff[3 * g];
    bool fused3 = triples;
    const int rows = row_end - row_begin;
    const bool overwrite = nsplits == -1;                  // one split per tile, dW = product (no read, no zero fill needed)
    if (overwrite) nsplits = 1;
    // -2 (round 6): automatic like 0, and a product that lands on ONE split OVERWRITES its tiles instead of read-add-writing them
    // (the caller knows that this is the first product into dW since it was zeroed: the first micro-batch of a step) -- the
    // low-resolution layers' weight gradients are bound by their output (a 1280 x 1280 x 9 tile set against a 400-row reduction),
    // and the read was half of its traffic.  Products that split still add atomically into the zeroed dW.
    const bool overwrite_if_one = nsplits == -2;
    if (overwrite_if_one) nsplits = 0;
    const bool automatic = nsplits <= 0;
    if (grouped && nsplits <= 0) {
        // a grouped launch fills the chip with OTHER jobs' blocks: no split for occupancy's sake; splits only bound a
        // block's K loop (128 steps of 64 rows), and the fused 3-tap variant is always the better one (X read once).
        // Measured (round 3, same box, 32 / 64 / 128 / 256 steps): SD v1.5 B = 16 146.7 / 143.8 / 142.5 / 143.1 ms,
        // B = 4 56.4 / 55.9 / 55.7 / 56.1 ms, CelebA-HQ 58.40 / 58.29 / 58.30 / 58.31 ms -- every split re-adds a
        // 64 KiB tile through float atomics, and the transformer linears reduce over 65 k rows per set.
        nsplits = cdiv(rows, 128 * BR);
        if (nsplits < 1) nsplits = 1;
    }
    if (nsplits <= 0) {
        // Auto: pick (kernel variant, split count) by a small cost model (us), measured constants:
        //   a block's K-step (64 rows): 1.5 us for the 3-tap kernel (one 8-wave block per CU), 0.6 us for the
        //   one-tap kernel alone on a CU, 1.0 us with a second block beside it;
        //   every split adds the whole dW once more through float atomics (~1.3 TB/s chip-wide), a single split
        //   owns its tile and read-add-writes it with plain accesses.
        // The one-tap variant is only considered for short reductions (the 8x8 / 16x16 layers), where it puts 3x the
        // blocks on the chip without any split; on long reductions it re-reads X three times.
        const double bytes = (double)nsets * npanels * N * C * 4.0;
        const long tiles = (long)cdiv(N, BN) * cdiv(C, BC);
        double best = 1e30;
        int best_ns = 1;
        bool best_f3 = fused3;
        for (int v = 0; v < 2; ++v) {
            const bool f3 = v == 0;
            if (f3 && !fused3) continue;
            if (!f3 && fused3 && rows >= 8192) continue;
            const long base = tiles * (f3 ? npanels / 3 : npanels) * nsets;
            const long slots = f3 ? 256 : 512;
            const int max_ns = rows / 256 > 1 ? rows / 256 : 1;
            for (int ns = 1; ns <= max_ns && ns <= 1024; ++ns) {
                const long blocks = base * ns;
                const long rounds = cdiv(blocks, slots);
                if (rounds > 1 && ns > 1) break;           // never split into a second round
                const double tstep = f3 ? 1.5 : (blocks <= 256 ? 0.6 : 1.0);
                const double t = (double)rounds * cdiv(cdiv(rows, ns), BR) * tstep +
                                 (ns > 1 ? ns * bytes / 1.3e6 : 2.0 * bytes / 4.0e6);
                if (t < best) { best = t; best_ns = ns; best_f3 = f3; }
            }
        }
        nsplits = best_ns;
        fused3 = best_f3;
    }
    p.nsplits = nsplits;
    p.rmw = overwrite ? 2 : (automatic && nsplits == 1 ? (overwrite_if_one ? 2 : 1) : 0);
    int rps = cdiv(rows, nsplits);
    rps = cdiv(rps, BR) * BR;
    p.rows_per_split = rps;
    fused3_out = fused3;
    return SISS_OK;
}

template <int TAPS>
static int launch_tn_group(const TNParams* ps, int n, hipStream_t st, int max_blocks = 0) {
    using C_ = TCfg<TAPS, TAPS == 3>;
    static unsigned char attr_set[kMaxDevices];
    if (siss_ensure_smem((const void*)gemm_tn_grouped_kernel<TAPS>, C_::kSmemBytes, attr_set) != SISS_OK) return SISS_ERR_LAUNCH;
    // Blocks are dispatched in grid order, one resident block per CU: a launch ends with whichever blocks were dispatched last,
    // and a long block (up to 128 K-steps, ~190 us) dispatched near the end leaves most CUs idle behind it.  Longest blocks
    // first (LPT): jobs sorted by descending K-steps per block and dealt round-robin to the launches, so that every launch
    // starts with its longest blocks and drains through its shortest ones (the 8 x 8 levels' ~25-step blocks).
    int order[256];
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order, order + n, [&](int a, int b) { return ps[a].rows_per_split > ps[b].rows_per_split; });
    const int nlaunch = cdiv(n, kMaxJobs);
    for (int l = 0; l < nlaunch; ++l) {
        TNGroup g;
        g.njobs = 0;
        int total = 0;
        for (int i = l; i < n; i += nlaunch) {
            const TNParams& p = ps[order[i]];
            const int j = g.njobs++;
            g.job[j] = p;
            g.first[j] = total;
            g.nwg[j] = cdiv(p.N, BN) * cdiv(p.C, BC) * (p.npanels / TAPS) * p.nsets * p.nsplits;
            total += (g.nwg[j] + 7) & ~7;
            siss_count_dispatch(TAPS == 3 ? SISS_K_TN3 : SISS_K_TN1);
            note_overwrite(p);
        }
        g.first[g.njobs] = total;
        if (max_blocks > 0 && total > max_blocks) {
            static unsigned char attr_capped[kMaxDevices];
            if (siss_ensure_smem((const void*)gemm_tn_grouped_capped_kernel<TAPS>, C_::kSmemBytes, attr_capped) != SISS_OK) return SISS_ERR_LAUNCH;
            gemm_tn_grouped_capped_kernel<TAPS><<<dim3(max_blocks), C_::kThreads, C_::kSmemBytes, st>>>(g);
        } else {
            gemm_tn_grouped_kernel<TAPS><<<dim3(total), C_::kThreads, C_::kSmemBytes, st>>>(g);
        }
    }
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

extern "C" {

// dW must be zeroed (or hold the running sum for gradient accumulation) before the call.
// dbias / dbias2 (optional): dbias[set*set_stride + n] += sum over the set's rows of Y[r][n].
// Rows [row_begin, row_end) of every set are reduced; shifts/coffs are HOST arrays.
// nsplits == 0: choose the kernel variant and the split count here (cost model below); when that lands on one split
// the block that owns a tile read-add-writes it with plain accesses instead of atomics.
// nsplits == -1: one split per tile and dW is OVERWRITTEN with the product (no accumulation: the caller needs no zero
// fill; attention dK / dV).
// nsplits == -2: as 0, but a product that lands on one split overwrites its tiles (the first product into a zeroed dW: no read).
int siss_gemm_tn(const void* Y, long ldy, const void* X, long ldx, float* dW, long set_stride, int N, int C,
                 int npanels, const int* shifts, const int* coffs, int nsets, int rows_per_set,
                 long x_set_rows, int row_begin, int row_end, int nsplits, const void* zero_page,
                 float* dbias, float* dbias2, void* stream) {
    TNParams p;
    bool fused3 = false;
    const int rc = tn_setup(Y, ldy, X, ldx, dW, set_stride, N, C, npanels, shifts, coffs, nsets, rows_per_set, x_set_rows,
                            row_begin, row_end, nsplits, zero_page, dbias, dbias2, false, p, fused3);
    if (rc != SISS_OK) return rc;
    if (fused3) return launch_tn<3>(p, (hipStream_t)stream);
    return launch_tn<1>(p, (hipStream_t)stream);
}

// siss_gemm_tn whose bias gradients have a set stride of their own: dbias[set * bias_set_stride + n] (dW keeps set_stride).  For a
// product that accumulates into a SCRATCH with its own set stride while its bias gradient goes to the flat gradient buffer (the
// phase weight gradients of a sub-pixel upsample convolution).
int siss_gemm_tn_bs(const void* Y, long ldy, const void* X, long ldx, float* dW, long set_stride, int N, int C,
                    int npanels, const int* shifts, const int* coffs, int nsets, int rows_per_set,
                    long x_set_rows, int row_begin, int row_end, int nsplits, const void* zero_page,
                    float* dbias, float* dbias2, long bias_set_stride, void* stream) {
    TNParams p;
    bool fused3 = false;
    const int rc = tn_setup(Y, ldy, X, ldx, dW, set_stride, N, C, npanels, shifts, coffs, nsets, rows_per_set, x_set_rows,
                            row_begin, row_end, nsplits, zero_page, dbias, dbias2, false, p, fused3);
    if (rc != SISS_OK) return rc;
    p.bias_stride = bias_set_stride;
    if (fused3) return launch_tn<3>(p, (hipStream_t)stream);
    return launch_tn<1>(p, (hipStream_t)stream);
}

// Drains the host-side log of overwriting products (nsplits == -1, or -2 landing on one split) launched by this process since the
// last call: up to max_records records of two longs (address of the first float, float count), one per cotangent set, in launch
// order.  Returns the number of records there were (more than max_records: the rest is dropped), or -1 if the log had filled up
// (65536 records without a drain: its content is then incomplete; -2: bad arguments).  out may be null with max_records = 0 (just clear).
long siss_gemm_tn_overwrite_log(long* out, long max_records) {
    if (max_records < 0 || (!out && max_records > 0)) return -2;
    std::lock_guard<std::mutex> lock(g_ow_mu);
    const long n = (long)g_ow.size();
    for (long i = 0; i < n && i < max_records; ++i) { out[2 * i] = (long)(uintptr_t)g_ow[i].ptr; out[2 * i + 1] = g_ow[i].floats; }
    const bool over = g_ow_overflow;
    g_ow.clear();
    g_ow_overflow = false;
    return over ? -1 : n;
}

// Balance of siss_gemm_tn_pair: relative cost of a 64-row K-step of a one-tap virtual block against a 3-tap block's (permille;
// 0 = the built-in default).  A tuning knob for tools/probes and the A/B harness, per process.
static int g_pair_cost_permille = 0;
int siss_gemm_tn_set_pair_cost(int permille) {
    if (permille >= 0) g_pair_cost_permille = permille;
    return g_pair_cost_permille;
}

// ONE launch, ONE round of blocks for two weight-gradient products: `job3` whose panels are 3x3 filter rows (the fused 3-tap
// kernel's shape) and `job1`, a one-panel (1x1 convolution / linear) product -- e.g. a resnet's conv2 and its conv_shortcut, which
// reduce over the same cotangent.  Both are split over row ranges so that together they fill `max_blocks` workgroups (0: one per CU)
// and end at about the same time; partial tiles are added with float atomics (dW must hold the running sum).  jobs: siss_tn_job
// records (nsplits is ignored).  Returns 1 (bad argument) when the shapes are not (3-tap, one-panel): launch them separately then.
int siss_gemm_tn_pair(const void* job3, const void* job1, int max_blocks, void* stream) {
    SISS_CHECK_ARG(job3 && job1 && max_blocks >= 0);
    const siss_tn_job& a = *(const siss_tn_job*)job3;
    const siss_tn_job& b = *(const siss_tn_job*)job1;
    SISS_CHECK_ARG(a.npanels % 3 == 0 && b.npanels == 1);
    int maxb = max_blocks;
    if (maxb == 0) {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return SISS_ERR_LAUNCH;
        maxb = cus;
    }
    SISS_CHECK_ARG(maxb >= 16);
    // blocks: 3-tap  base3 * s3 (rounded up to 8), one-tap ceil(base1 * s1 / 2) rounded up to 8; K-steps per block rows / (64 s)
    const long base3 = (long)cdiv(a.N, BN) * cdiv(a.C, BC) * (a.npanels / 3) * a.nsets;
    const long base1 = (long)cdiv(b.N, BN) * cdiv(b.C, BC) * b.nsets;
    const int rows3 = a.row_end - a.row_begin, rows1 = b.row_end - b.row_begin;
    SISS_CHECK_ARG(rows3 > 0 && rows1 > 0);
    // one-tap K-step (two virtual blocks per CU: one 32-KiB stage each in flight, i.e. bound by the latency of 64 KiB of HBM reads:
    // ~1.4 us) against a 3-tap block's (1.25 us).  Swept at the 256 x 256 shapes (tools/probes/tn_pair.py, us per launch, apart ->
    // paired at 1000 / 1300 / 1500 / 1800 permille): conv2 + conv_shortcut (N 128, C 128 | C 256) 809 -> 800 / 703 / 662 / 657; conv1
    // (C 256) + shortcut 1140 -> 1046 / 1129 / 1084 / 1093; conv2 + conv_out (N 27) 590 -> 610 / 570 / 554 / 564
    const double c1 = (g_pair_cost_permille ? g_pair_cost_permille : 1500) / 1000.0;
    double best = 1e30;
    int s3 = 0, s1 = 0;
    for (int t1 = 1; t1 <= 4096; ++t1) {
        const long phys1 = ((base1 * t1 + 1) / 2 + 7) & ~7L;
        if (phys1 >= maxb) break;
        const long t3 = ((maxb - phys1) & ~7L) / base3;
        if (t3 < 1) break;
        const double cost = std::max((double)cdiv(cdiv(rows3, (int)t3), BR), c1 * cdiv(cdiv(rows1, t1), BR));
        if (cost < best) { best = cost; s3 = (int)t3; s1 = t1; }
    }
    SISS_CHECK_ARG(s3 >= 1 && s1 >= 1);
    TNPair g;
    bool f3 = false, f1 = false;
    int rc = tn_setup(a.Y, a.ldy, a.X, a.ldx, a.dW, a.set_stride, a.N, a.C, a.npanels, a.shifts, a.coffs, a.nsets, a.rows_per_set,
                      a.x_set_rows, a.row_begin, a.row_end, s3, a.zero_page, a.dbias, a.dbias2, false, g.j3, f3);
    if (rc != SISS_OK) return rc;
    rc = tn_setup(b.Y, b.ldy, b.X, b.ldx, b.dW, b.set_stride, b.N, b.C, b.npanels, b.shifts, b.coffs, b.nsets, b.rows_per_set,
                  b.x_set_rows, b.row_begin, b.row_end, s1, b.zero_page, b.dbias, b.dbias2, false, g.j1, f1);
    if (rc != SISS_OK) return rc;
    SISS_CHECK_ARG(f3 && !f1);
    if (a.bias_set_stride) g.j3.bias_stride = a.bias_set_stride;
    if (b.bias_set_stride) g.j1.bias_stride = b.bias_set_stride;
    g.nv3 = (int)(base3 * s3);
    g.nv1 = (int)(base1 * s1);
    g.n3 = (g.nv3 + 7) & ~7;
    g.nphys1 = ((g.nv1 + 1) / 2 + 7) & ~7;
    static unsigned char attr_set[kMaxDevices];
    constexpr int smem = TCfg<3, true>::kSmemBytes > 2 * TCfg<1>::kSmemBytes ? TCfg<3, true>::kSmemBytes : 2 * TCfg<1>::kSmemBytes;
    if (siss_ensure_smem((const void*)gemm_tn_mixed_kernel, smem, attr_set) != SISS_OK) return SISS_ERR_LAUNCH;
    siss_count_dispatch(SISS_K_TN3);
    siss_count_dispatch(SISS_K_TN1);
    siss_count_dispatch(SISS_K_TN_PAIR);
    gemm_tn_mixed_kernel<<<dim3(g.n3 + g.nphys1), TCfg<3>::kThreads, smem, (hipStream_t)stream>>>(g);
    SISS_LAUNCH_RET();
}

// The same product for `njobs` independent problems in ONE launch per kernel variant (job table passed by value as kernel
// arguments: nothing is copied to the device, hipGraph-safe).  jobs: HOST array of siss_tn_job (the argument list of
// siss_gemm_tn as a struct, shifts / coffs inline).  Meant for the low-resolution weight gradients: each of them alone leaves
// most CUs idle and pays a launch's fixed latency.  All operands must stay valid until the launch has run.
static int tn_grouped(const void* jobs, int njobs, int max_blocks, void* stream);
int siss_gemm_tn_grouped(const void* jobs, int njobs, void* stream) { return tn_grouped(jobs, njobs, 0, stream); }
// The same with the launches' grids capped at max_blocks workgroups (a multiple of 8, >= 8; the 3-tap launches hold one CU per
// workgroup, the one-tap launches half a CU): each workgroup walks several blocks.  For running the weight gradients on a side stream
// beside kernels that leave most of the chip idle (the low-resolution part of the backward pass) without taking every CU from them.
int siss_gemm_tn_grouped_capped(const void* jobs, int njobs, int max_blocks, void* stream) {
    SISS_CHECK_ARG(max_blocks >= 8 && max_blocks % 8 == 0);
    return tn_grouped(jobs, njobs, max_blocks, stream);
}
static int tn_grouped(const void* jobs, int njobs, int max_blocks, void* stream) {
    SISS_CHECK_ARG(jobs && njobs > 0 && njobs <= 256);
    const siss_tn_job* js = (const siss_tn_job*)jobs;
    TNParams p3[256], p1[256];
    int n3 = 0, n1 = 0;
    for (int i = 0; i < njobs; ++i) {
        const siss_tn_job& j = js[i];
        TNParams p;
        bool fused3 = false;
        const int rc = tn_setup(j.Y, j.ldy, j.X, j.ldx, j.dW, j.set_stride, j.N, j.C, j.npanels, j.shifts, j.coffs, j.nsets,
                                j.rows_per_set, j.x_set_rows, j.row_begin, j.row_end, j.nsplits, j.zero_page, j.dbias, j.dbias2,
                                true, p, fused3);
        if (rc != SISS_OK) return rc;
        if (j.bias_set_stride) p.bias_stride = j.bias_set_stride;
        if (fused3) p3[n3++] = p; else p1[n1++] = p;
    }
    if (n3) { const int rc = launch_tn_group<3>(p3, n3, (hipStream_t)stream, max_blocks); if (rc != SISS_OK) return rc; }
    if (n1) { const int rc = launch_tn_group<1>(p1, n1, (hipStream_t)stream, 2 * max_blocks); if (rc != SISS_OK) return rc; }
    return SISS_OK;
}


}  // extern "C"
