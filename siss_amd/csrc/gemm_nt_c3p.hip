// Persistent, producer/consumer-specialised 3x3-convolution kernel (fprop and dgrad on the large grids).
//
// What the one-role kernels (gemm_nt.hip, gemm_nt_c3.hip) cannot hide: a wave issues its instructions in order, and
// one global_load_lds piece costs 60-180 issue cycles, so the waves that own the MFMAs stall for the whole DMA
// issue of every K-step (measured: DMA phase + MFMA phase ~ sum, no overlap), and the epilogue of a tile overlaps
// nothing unless another block is resident.  Here one 8-wave block per CU runs ALL its tiles:
//
//   waves 0-3  consumers: a 254x128 output tile as four 128x64 wave tiles (8x4 MFMA accumulators), LDS fragment
//              reads one 16-MFMA block ahead of the MFMAs; at the end of a tile they only add bias / row bias, round
//              to bf16 and park the tile in the LDS slot they just finished reading;
//   waves 4-7  producers: the LDS-DMA of every group, and the rest of the epilogue: they pull the parked tile into
//              registers (which frees the slot for the next DMA), then mask the halo, add the residual and store
//              16 B per lane / 128 B per row while the consumers are already in the next tile's MFMAs.
//              Consumer w and producer w+4 share a SIMD: DMA issue, epilogue VALU and stores overlap MFMA issue.
// (A lone consumer wave doing the whole epilogue cost 40 % of a tile: wave64 VALU ops take 4 cycles each and every
// dependent global load is a full round trip with nobody to hide it.)
//
// Group = (tile, filter row ky, 64-channel chunk kc): as in gemm_nt_c3 the A tile of a group is staged once
// (256 rows x 128 B) beside the three taps' weight tiles and read at row offsets 0/1/2: 80 KiB per group, two slots
// = all 160 KiB of the CU's LDS.  ONE barrier per group (192 MFMAs per consumer wave), shared by all eight waves:
// the producers pass barrier #g only after group g has landed (vmcnt(0)), then issue group g+1 into the slot that
// group g-1 just left.  The group sequence simply continues over tile boundaries: while the consumers run the
// epilogue of tile i, tile i+1's first group is already landing.
//
// Folded 1x1 shortcut (NTParams::A2): a resnet's conv_shortcut(x) + conv2(h) is ONE accumulation -- after the 3 * Kp / 64 groups
// of the 3x3 filter every tile runs K2 / 64 "shortcut groups": the A tile comes from the second tensor (its own row stride),
// staged at the centre row's shift so that the centre tap's read offset (row + 1) lands on the pixel itself, beside ONE weight
// tile in the centre tap's slot; the consumers run the 4 centre-tap blocks of 16 MFMAs.  The 1x1 product's own launch (an
// HBM-bound GEMM: 173 us at 256 x 256, 256 -> 128 channels), its output tensor and the residual read of the 3x3 launch's store
// path (+ 50 us) disappear for K2 / (9 Kp) more MFMA work.
//
// Second product over the same A (NTParams::Cx): the backward of that shortcut.  conv2's dgrad and conv_shortcut's dgrad both read
// the block's output cotangent; the 1x1 one is HBM-bound on its own (it writes a 256-channel tensor for 1/9 of the MFMAs per
// column: 396 us at 256 x 256).  Here it rides as Nx / 128 more column tiles per row tile ("x tiles": Kp / 64 centre-tap groups,
// weights Wx, output Cx with its own row stride) between the 3x3 tiles of the same rows -- its stores drain while the neighbours' MFMAs
// run, and A is fetched from HBM once for both.
#include "nt_common.h"
#include <type_traits>

// Probe build (-DSISS_PROBE, tools/probes/build_probe.sh): in-kernel phase timers (NTParams::dbg) and ablation switches
// (NTParams::ablate); the product build compiles none of it.
#ifdef SISS_PROBE
#define C3P_ABLATE(bit) (p.ablate & (bit))
#else
#define C3P_ABLATE(bit) false
#endif

namespace {

constexpr int P_BM = 256, P_VALID = 254, P_THREADS = 512;
constexpr int P_ABYTES = P_BM * 128;                       // 32,768
constexpr int P_WBYTES = BN * 128;                         // 16,384 per tap
constexpr int P_SLOT = P_ABYTES + 3 * P_WBYTES;            // 81,920: one GROUP (A tile + three weight tiles)
constexpr int P_SMEM = 2 * P_SLOT;                         // 163,840 = all of the CU's LDS
static_assert(P_VALID == kQsTileRows && P_BM / 2 == kQsHalfRows, "qstats geometry is shared with groupnorm.hip");

__device__ __forceinline__ int swz3p(int row) { return row & 6; }
// s_barrier as inline asm: hipcc puts `s_waitcnt vmcnt(0)` in front of the builtin barrier, which would make every
// consumer wave wait for its epilogue STORES (an HBM round trip) before the next tile's first group barrier.
// Everything that must be complete at a barrier here is waited for explicitly.
__device__ __forceinline__ void c3p_barrier() { asm volatile("s_barrier" ::: "memory"); }

struct TileMap {
    int nb, per, b_lo, b_hi, tiles_n, ntiles;
    __device__ __forceinline__ int tile(int k) const { return k * nb + b_lo * per + b_hi; }   // XCD-contiguous
};

// QS: the producers also form the GroupNorm statistics of what they store (NTParams::qstats): the tensor's consumer is a
// GroupNorm, whose statistics pass (one more read of the whole tensor) then disappears.
template <bool QS>
__global__ __launch_bounds__(P_THREADS, 1) void gemm_nt_c3p_kernel(const NTParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    TileMap tm;
    tm.nb = gridDim.x; tm.per = tm.nb >> 3; tm.b_lo = blockIdx.x & 7; tm.b_hi = blockIdx.x >> 3;
    const int tiles_nm = p.N / BN;                         // column tiles of the 3x3 product ...
    const int tiles_nx = p.Cx ? p.Nx / BN : 0;             // ... and of the optional second (1x1, centre-tap) product over the same A
    tm.tiles_n = tiles_nm + tiles_nx;
    tm.ntiles = ((p.M + P_VALID - 1) / P_VALID) * tm.tiles_n;
    int count = 0;
    while (tm.tile(count) < tm.ntiles) ++count;
    const int kchunks = p.Kp / BK;
    const int k2chunks = p.A2 ? p.K2 / BK : 0;             // folded 1x1 shortcut: more K-groups per tile, centre tap only
    const int g3 = 3 * kchunks;                            // groups of the 3x3 filter per tile
    const int gpt = g3 + k2chunks;                         // groups per tile of the 3x3 product (an x tile has kchunks groups)
    // class of a tile: x = it belongs to the second product; groups it runs
    auto tile_is_x = [&](int k) { return tm.tile(k) % tm.tiles_n >= tiles_nm; };
    auto tile_groups = [&](int k) { return tile_is_x(k) ? kchunks : gpt; };
    int G = 0;
    for (int k = 0; k < count; ++k) G += tile_groups(k);
    if (G == 0) return;
    const long wtap = (long)p.N * p.Kp;
    constexpr int SROW = 144;                              // parked tile: 128 B of channels + 16 B pad per row
    const int rpi = p.rows_per_image;
    const int last_img = (p.M - 1) / rpi;

    if (w >= 4) {
        // ------------------------------------------------------------------ producers
        const int pw = w - 4;
        const int wm = pw >> 1, wn = pw & 1;               // the consumer wave whose sub-tile this wave stores
        const int prow = lane >> 3, pc = lane & 7;
        const unsigned smem_a = lds_addr(smem);
        int gk = 0, gky = 0, gkc = 0, issued = 0;          // group cursor: (tile k, filter row ky, K chunk kc)
        // LDS-DMA sources as 32-bit byte offsets from a wave-uniform base (the saddr form of global_load_lds: address = SGPR pair +
        // VGPR offset): per piece the store waves issue s_mov m0 + the load and NO vector arithmetic -- the 64-bit per-lane
        // pointer form cost a v_lshl_add_u64 and an m0 save / restore per piece, on a SIMD they share with an MFMA wave.
        // (m0 is a reserved register the compiler sets itself before each of its own uses; nothing here relies on its value.)
        unsigned aoffs[8], woffs[4], aoffs2[8], woffs2[4];
        bool issue_x = false;                              // the tile whose groups are being ISSUED belongs to the second product
        auto set_tile = [&](int k) {
            const int t = tm.tile(k);
            const int tn = t % tm.tiles_n;
            issue_x = tn >= tiles_nm;
            const int m0 = (t / tm.tiles_n) * P_VALID, n0 = (issue_x ? tn - tiles_nm : tn) * BN;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int row = (pw * 8 + j) * 8 + prow;
                int gr = m0 + row; gr = gr < p.M + 1 ? gr : p.M + 1;
                aoffs[j] = (unsigned)(((long)gr * p.lda + ((pc ^ swz3p(row)) << 3)) * 2);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = (pw * 4 + j) * 8 + prow;
                woffs[j] = (unsigned)(((long)(n0 + row) * p.Kp + ((pc ^ swz3p(row)) << 3)) * 2);
            }
            if (k2chunks && !issue_x) {                    // (uniform) the shortcut operand: other row strides
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int row = (pw * 8 + j) * 8 + prow;
                    int gr = m0 + row; gr = gr < p.M + 1 ? gr : p.M + 1;
                    aoffs2[j] = (unsigned)(((long)gr * p.lda2 + ((pc ^ swz3p(row)) << 3)) * 2);
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = (pw * 4 + j) * 8 + prow;
                    woffs2[j] = (unsigned)(((long)(n0 + row) * p.K2 + ((pc ^ swz3p(row)) << 3)) * 2);
                }
            }
        };
        auto dma = [&](unsigned off, const void* base, unsigned dst) __attribute__((always_inline)) {
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(off), "s"(dst), "s"(base) : "memory");
        };
        auto issue_group = [&]() {
            const unsigned dst = smem_a + (issued & 1) * P_SLOT;
            if (issue_x) {
                // tile of the second product: A at the centre row's shift (the centre tap reads tile row + 1 = the pixel itself),
                // ONE weight tile of Wx in the centre tap's slot; kchunks groups per tile
                const bf16_t* abase = p.A + (long)p.shift[3] * p.lda + p.coff[3] + gkc * BK;
                const bf16_t* wbase = p.Wx + gkc * BK;
#pragma unroll
                for (int j = 0; j < 8; ++j) dma(aoffs[j], abase, dst + (pw * 8 + j) * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) dma(woffs[j], wbase, dst + P_ABYTES + P_WBYTES + (pw * 4 + j) * 1024);
                ++issued;
                if (++gkc == kchunks) { gkc = 0; gky = 0; ++gk; if (gk < count) set_tile(gk); }
                return;
            }
            if (gky < 3) {
                const long aoff = (long)p.shift[3 * gky] * p.lda + p.coff[3 * gky] + gkc * BK;
                const long woff = 3L * gky * wtap + gkc * BK;
                if (!(C3P_ABLATE(2) && issued >= 2)) {
                    const bf16_t* abase = p.A + aoff;
#pragma unroll
                    for (int j = 0; j < 8; ++j) dma(aoffs[j], abase, dst + (pw * 8 + j) * 1024);
#pragma unroll
                    for (int t = 0; t < 3; ++t) {
                        const bf16_t* wbase = p.W + woff + t * wtap;
#pragma unroll
                        for (int j = 0; j < 4; ++j) dma(woffs[j], wbase, dst + P_ABYTES + t * P_WBYTES + (pw * 4 + j) * 1024);
                    }
                }
            } else {
                // shortcut group: A2 staged at the centre ROW's shift (shift[3] = shift[4] - 1), so the centre tap's read offset
                // (tile row + 1) is the pixel itself; one weight tile, in the centre tap's slot
                const bf16_t* abase = p.A2 + (long)p.shift[3] * p.lda2 + gkc * BK;
                const bf16_t* wbase = p.W2 + gkc * BK;
#pragma unroll
                for (int j = 0; j < 8; ++j) dma(aoffs2[j], abase, dst + (pw * 8 + j) * 1024);
#pragma unroll
                for (int j = 0; j < 4; ++j) dma(woffs2[j], wbase, dst + P_ABYTES + P_WBYTES + (pw * 4 + j) * 1024);
            }
            ++issued;
            if (++gkc == (gky < 3 ? kchunks : k2chunks)) {
                gkc = 0;
                if (++gky == (k2chunks ? 4 : 3)) { gky = 0; ++gk; if (gk < count) set_tile(gk); }
            }
        };

        // The parked tile of the previous tile, in registers: lane -> (row = it*8 + lane>>3, 16-B chunk = lane&7).
        u32x4_t ov[16];
        int pend = -1, phalf = 0;                          // tile iteration whose rows `ov` holds (-1: none), next half
        // QS: this lane's running statistics over the rows it stores, [set][sum lo quad, sumsq lo quad, sum hi quad, sumsq hi quad]:
        // its 8 channels are two 4-channel quads; set 0 takes every row, set 1 only the rows of the tile's SECOND image
        float qacc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        // fold the eight row groups of the wave (lane >> 3), then lanes 0..7 write the (half tile, slot) entries:
        // [2 * row tile + wm][slot][N / 4 quads][sum, sumsq]; every entry is written by every launch
        auto fold_tile = [&](int k) __attribute__((always_inline)) {
            const int tl = tm.tile(k);
            const int m0 = (tl / tm.tiles_n) * P_VALID, n0 = (tl % tm.tiles_n) * BN;
            const bool straddle = m0 + P_BM > (m0 / rpi + 1) * rpi;
            float qs[2][4];                                // [slot][sum lo quad, sumsq lo quad, sum hi quad, sumsq hi quad]
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int e = 0; e < 4; ++e) qs[sl][e] = qacc[sl][e];
#pragma unroll
            for (int e = 0; e < 4; ++e) qs[0][e] -= qs[1][e];      // all rows - second image's rows (0 unless straddling)
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl == 1 && !straddle) continue;        // (uniform) slot 1 of a one-image tile is all zero
#pragma unroll
                for (int e = 0; e < 4; ++e) qs[sl][e] = sum_lanes_mod8(qs[sl][e]);
            }
            if (prow == 0) {
                const long ht = (long)(tl / tm.tiles_n) * 2 + wm;
                const int quad = ((n0 + wn * 64) >> 2) + pc * 2;
                float* dst = p.qstats + ((ht * 2) * (p.N >> 2) + quad) * 2;
                *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{qs[0][0], qs[0][1], qs[0][2], qs[0][3]};
                *reinterpret_cast<f32x4_t*>(dst + (p.N >> 2) * 2) = f32x4_t{qs[1][0], qs[1][1], qs[1][2], qs[1][3]};
            }
#pragma unroll
            for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                for (int e = 0; e < 4; ++e) qacc[sl][e] = 0.f;
        };
        // Stores rows it = 8*half .. 8*half+7 of the parked tile (one half per group, so that the producers are
        // back at the next group barrier in time).  Returns the number of store instructions issued when that number
        // is exact (8, or 10 with the statistics rows), else 0.
        auto store_tile = [&](auto has_r, int half) __attribute__((always_inline)) -> int {
            constexpr bool HAS_R = decltype(has_r)::value;
            const int tl = tm.tile(pend);
            const int tn = tl % tm.tiles_n;
            const bool xt = tn >= tiles_nm;                // (uniform) a tile of the second product: its own output tensor
            const int m0 = (tl / tm.tiles_n) * P_VALID, n0 = (xt ? tn - tiles_nm : tn) * BN;
            char* const cbase = reinterpret_cast<char*>(xt ? p.Cx : p.C);
            const unsigned ldc = (unsigned)(xt ? p.ldcx : p.ldc);
            const int img0 = m0 / rpi;                     // a 256-row tile spans at most two images (rpi >= 256)
            const int split = (img0 + 1) * rpi;            // first row of the second image
            const bool straddle = m0 + P_BM > split;       // (uniform) the tile's last rows belong to the next image
            const int ccol = n0 + wn * 64 + pc * 8;
            // Row addresses as 32-bit byte offsets from the (scalar) tensor bases: one multiply per lane and tile half, then a
            // scalar stride per row -- the 64-bit form cost two quarter-rate v_mul_lo_u32 + a v_mad_u64_u32 per stored row in
            // waves that share their SIMD with an MFMA wave.  (siss_launch_gemm_nt_c3p checks that C and R span < 4 GiB.)
            const unsigned row0 = (unsigned)(m0 + wm * 128 + half * 64 + prow);
            const unsigned coff0 = (row0 * ldc + (unsigned)ccol) * 2u, cstep = ldc * 16u;
            u32x4_t res[HAS_R ? 8 : 1];
            if constexpr (HAS_R) {
                // all residual loads, then ONE wait, before the first store: vmcnt counts loads and stores in one
                // in-order queue on gfx9 -- a load waited for after a store would wait for that store's round trip
                const unsigned roff0 = (row0 * (unsigned)p.ldr + (unsigned)ccol) * 2u, rstep = (unsigned)p.ldr * 16u;
                const unsigned rlast = ((unsigned)(p.M - 1) * (unsigned)p.ldr + (unsigned)ccol) * 2u;   // rows past M read the last row
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) {
                    unsigned ro = roff0 + i8 * rstep; ro = ro < rlast ? ro : rlast;
                    res[i8] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.R) + ro);
                }
#pragma unroll
                for (int i8 = 0; i8 < 8; ++i8) asm volatile("" : "+v"(res[i8]));
            }
            // Image-relative pixel coordinates of this lane's first row of the half; its next rows are 8 apart (Wp >= 8: at most one
            // row wrap per step; y == Hp is row 0 of the next image).  The row loop below is branch-free up to the one store
            // predicate: per-row branches (skip / halo / residual) cost more scalar and exec traffic than the arithmetic they saved.
            int py = 1, px = 1;
            if (p.Hp > 0) {
                const int r0i = (int)row0;
                const int rem = r0i - img0 * rpi - (r0i >= split ? rpi : 0);
                py = (int)(((float)rem + 0.5f) * p.inv_wp); px = rem - py * p.Wp;
            }
            const int ylast = p.Hp > 0 ? p.Hp - 1 : 1 << 30, xlast = p.Hp > 0 ? p.Wp - 1 : 1 << 30, wrap = p.Hp > 0 ? p.Wp : 1 << 30;
#pragma unroll
            for (int i8 = 0; i8 < 8; ++i8) {
                const int row = wm * 128 + (half * 8 + i8) * 8 + prow;
                const int r = m0 + row;
                u32x4_t o = half ? ov[8 + i8] : ov[i8];
                if constexpr (HAS_R) {
                    const u32x4_t rr = res[i8];
                    // bf16 + bf16: f32 add, one rounding (the reference's `x + h` under autocast).  (`__bf16` vector arithmetic on the
                    // elements of a u32x4 is miscompiled by hipcc 7.2 -- every element got element 0's sum -- hence the helper.)
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = add_bf16x2(o[e], rr[e]);
                }
                const bool halo = (py == 0) | (py == ylast) | (px == 0) | (px == xlast);
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = halo ? 0u : o[e];
                px += 8;
                if (px >= wrap) { px -= wrap; ++py; }
                if (py > ylast) py = 0;
                if (row >= P_VALID || r >= p.M) continue;
                *reinterpret_cast<u32x4_t*>(cbase + (coff0 + i8 * cstep)) = o;
                if constexpr (QS) {
                    // statistics of the STORED values (halo rows are zero and add nothing) straight from the packed bf16 pairs:
                    // v_dot2c_f32_bf16 (acc += a.lo * b.lo + a.hi * b.hi) against (1, 1) gives the sum, against itself the
                    // sum of squares -- 8 instructions per row, no unpacking.  (Measured: wherever this work is placed -- here,
                    // or in the later groups where the store waves only issue the DMA -- it costs the kernel its own duration:
                    // the fewest instructions win, and that is here, where the value is in registers already.)
                    // Set 0 takes EVERY row; set 1 only the rows of the tile's second image (one tile in ~260 has any: uniform
                    // branch), selected by masks -- a per-lane branch between two accumulator sets makes the compiler index them
                    // dynamically, i.e. park them in scratch memory.  The first image's share is set 0 - set 1.
                    const bf16x2_t ones = __builtin_bit_cast(bf16x2_t, 0x3F803F80u);
                    // (each pair goes through an empty asm into a register of its own: fed with elements of the u32x4 vector,
                    // hipcc 7.2 emitted every dot product against element 0 -- seen in the ISA, caught by test_hip_gn_qstats.py)
                    uint32_t ow[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) { ow[e] = o[e]; asm volatile("" : "+v"(ow[e])); }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const bf16x2_t v = __builtin_bit_cast(bf16x2_t, ow[e]);
                        qacc[0][(e >> 1) * 2] = __builtin_amdgcn_fdot2_f32_bf16(v, ones, qacc[0][(e >> 1) * 2], false);
                        qacc[0][(e >> 1) * 2 + 1] = __builtin_amdgcn_fdot2_f32_bf16(v, v, qacc[0][(e >> 1) * 2 + 1], false);
                    }
                    if (straddle) {
                        const uint32_t mb = r >= split ? 0xffffffffu : 0u;
                        const bf16x2_t mones = __builtin_bit_cast(bf16x2_t, 0x3F803F80u & mb);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const bf16x2_t v = __builtin_bit_cast(bf16x2_t, ow[e]), vm = __builtin_bit_cast(bf16x2_t, ow[e] & mb);
                            qacc[1][(e >> 1) * 2] = __builtin_amdgcn_fdot2_f32_bf16(v, mones, qacc[1][(e >> 1) * 2], false);
                            qacc[1][(e >> 1) * 2 + 1] = __builtin_amdgcn_fdot2_f32_bf16(vm, v, qacc[1][(e >> 1) * 2 + 1], false);
                        }
                    }
                }
            }
            const bool exact = m0 + wm * 128 + half * 64 + 64 <= p.M;   // rows 254/255 only mask lanes of the last instruction
            if constexpr (QS) { if (half == 1) { fold_tile(pend); return exact ? 10 : 0; } }
            return exact ? 8 : 0;
        };

        int gin = 0, ck = 0, cgpt = tile_groups(0);        // consumption cursor: group within tile ck, which has cgpt groups
        set_tile(0);
        issue_group();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        auto store_half = [&]() __attribute__((always_inline)) -> int {
            int counted = 0;
            if (!C3P_ABLATE(1)) {
                if (p.R && !tile_is_x(pend)) store_tile(std::true_type{}, phalf);      // its wait retired the DMA above as well
                else counted = store_tile(std::false_type{}, phalf);
            }
            if (++phalf == 2) { phalf = 0; pend = -1; }
            return counted;
        };
        for (int g = 0; g < G; ++g) {
            c3p_barrier();                                 // #g: group g has landed; consumers are done with group g-1
            if (g + 1 < G) issue_group();                  // -> slot (g+1)&1: last read by group g-1 / parked tile already in `ov`
            int counted = 0;
            if (pend >= 0) counted = store_half();
            // the DMA of group g+1 must have landed before barrier #g+1; the younger stores may stay in flight
            if (counted == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (counted == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (++gin == cgpt) {                           // g was the last group of its tile
                gin = 0;
                c3p_barrier();                             // E : every consumer has finished reading slot g&1
                c3p_barrier();                             // E2: the consumers have parked the tile there
                // (a tile of fewer than two groups would arrive here with a half of the previous tile still in `ov`)
                while (pend >= 0) store_half();
                const char* st = smem + (g & 1) * P_SLOT + pw * (128 * SROW);
#pragma unroll
                for (int it = 0; it < 16; ++it)
                    ov[it] = *reinterpret_cast<const u32x4_t*>(st + (it * 8 + prow) * SROW + pc * 16);
#pragma unroll
                for (int it = 0; it < 16; ++it) asm volatile("" : "+v"(ov[it]));      // in registers before barrier #g+1
                pend = ck;
                if (++ck < count) cgpt = tile_groups(ck);
            }
        }
        while (pend >= 0 && !C3P_ABLATE(1)) store_half();
        return;
    }

    // ---------------------------------------------------------------------- consumers
    const int wm = w >> 1, wn = w & 1;
    const int frow = lane & 15, fq = lane >> 4;
    int a_base[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) a_base[t] = (wm * 128 + frow + t) * 128 + ((fq ^ swz3p(frow + t)) << 4);
    const int w_base = P_ABYTES + (wn * 64 + frow) * 128 + ((fq ^ swz3p(frow)) << 4);

    f32x4_t acc[4][8];   // [n-tile][m-tile]
    bf16x8_t wf[2][4], af[2][4];
    int gg = 0;                                            // global group index (slot = gg & 1)
#ifdef SISS_PROBE
    long long tk[6] = {0, 0, 0, 0, 0, 0};
    long long t0 = clock64();
#define C3P_TICK(i) do { if (p.dbg) { const long long t1 = clock64(); tk[i] += t1 - t0; t0 = t1; } } while (0)
#else
#define C3P_TICK(i) do { } while (0)
#endif
    for (int k = 0; k < count; ++k) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // One group: blocks [B0, B1) of 16 MFMAs, block b = (tap t = b >> 2, k-half kk = (b >> 1) & 1, m-half h = b & 1).  The (4 or
        // 8) fragment reads of block b+1 are issued ONE behind each of the first MFMAs of block b instead of as a burst in front
        // of them -- a consumer wave is alone on its SIMD as far as MFMAs go, so every cycle it spends issuing a burst of reads is
        // a cycle the matrix pipe drains (measured +2.6 %: 1064 -> 1092 TF/s over a step).
        auto run_group = [&](auto b0c, auto b1c) __attribute__((always_inline)) {
            constexpr int B0 = decltype(b0c)::value, B1 = decltype(b1c)::value;
            const char* sl = smem + (gg & 1) * P_SLOT;
            c3p_barrier();                                 // #gg: this group's slot has landed
            if (C3P_ABLATE(4)) return;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                wf[(B0 >> 1) & 1][i] = *reinterpret_cast<const bf16x8_t*>(sl + (w_base ^ (((B0 >> 1) & 1) << 6)) + (B0 >> 2) * P_WBYTES + i * 2048);
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                af[B0 & 1][jj] = *reinterpret_cast<const bf16x8_t*>(sl + (a_base[B0 >> 2] ^ (((B0 >> 1) & 1) << 6)) + ((B0 & 1) * 4 + jj) * 2048);
#pragma unroll
            for (int b = B0; b < B1; ++b) {
                const int h = b & 1;
                const int nb = b + 1, nt = nb >> 2, nkk = (nb >> 1) & 1, nh = nb & 1;
                const bool more = b + 1 < B1;
                const int nreads = more ? (nh == 0 ? 8 : 4) : 0;
#pragma unroll
                for (int m = 0; m < 16; ++m) {
                    const int i = m >> 2, jj = m & 3;
                    acc[i][h * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(b >> 1) & 1][i], af[b & 1][jj],
                                                                                 acc[i][h * 4 + jj], 0, 0, 0);
                    if (m < nreads) {
                        if (m < 4)
                            af[nb & 1][m] = *reinterpret_cast<const bf16x8_t*>(sl + (a_base[nt] ^ (nkk << 6)) + (nh * 4 + m) * 2048);
                        else
                            wf[(nb >> 1) & 1][m - 4] = *reinterpret_cast<const bf16x8_t*>(sl + (w_base ^ (nkk << 6)) + nt * P_WBYTES + (m - 4) * 2048);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        };
        // (separate loops, not one loop with a branch: the accumulators must not meet at a merge point of two code paths; a tile
        // of the second product runs the last loop only: its kchunks centre-tap groups)
        const bool xtile = tile_is_x(k);
        const int n3 = xtile ? 0 : g3, nc = xtile ? kchunks : k2chunks;
        for (int g = 0; g < n3; ++g, ++gg) run_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 12>{});
        for (int g = 0; g < nc; ++g, ++gg) run_group(std::integral_constant<int, 4>{}, std::integral_constant<int, 8>{});   // centre tap
        C3P_TICK(0);

        // ---- consumer half of the epilogue: alpha, bias and the per-image row bias in f32, ONE rounding to bf16
        // (the residual is added after it by the producers: the reference's autocast order), parked in the slot of
        // the tile's last group as four wave-private 128 x 64 images.
        // acc[i][j][r] = channel n0 + wn*64 + i*16 + fq*4 + r of row m0 + wm*128 + j*16 + frow.
        const int tl = tm.tile(k);
        const int tnk = tl % tm.tiles_n;
        const int m0 = (tl / tm.tiles_n) * P_VALID, n0 = (xtile ? tnk - tiles_nm : tnk) * BN;
        const int img0 = m0 / rpi;
        const int img1 = img0 + 1 < last_img ? img0 + 1 : last_img;
        const int split = (img0 + 1) * rpi;
        const int ncol = n0 + wn * 64 + fq * 4;
        // every load of this half is issued here, in front of barrier E, so that its round trip overlaps the wait
        f32x4_t b0[4], b1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            b0[i] = (p.bias && !xtile) ? *reinterpret_cast<const f32x4_t*>(p.bias + ncol + i * 16) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            if (p.bias2 && !xtile) {
                const f32x4_t s2 = *reinterpret_cast<const f32x4_t*>(p.bias2 + ncol + i * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) b0[i][e] += s2[e];
            }
            b1[i] = b0[i];
        }
        if (p.rowbias && !xtile) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4_t r0 = *reinterpret_cast<const f32x4_t*>(p.rowbias + (long)img0 * p.ldrb + ncol + i * 16);
                const f32x4_t r1 = *reinterpret_cast<const f32x4_t*>(p.rowbias + (long)img1 * p.ldrb + ncol + i * 16);
#pragma unroll
                for (int e = 0; e < 4; ++e) { b0[i][e] += r0[e]; b1[i][e] += r1[e]; }
            }
        }
        c3p_barrier();                                     // E: every consumer wave has finished reading the slot
        C3P_TICK(1);
        {
            char* st = smem + ((gg - 1) & 1) * P_SLOT + w * (128 * SROW) + frow * SROW + fq * 8;
            auto park = [&](auto two_images) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const bool second = decltype(two_images)::value && m0 + wm * 128 + j * 16 + frow >= split;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        f32x4_t v = acc[i][j];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] * p.alpha + (second ? b1[i][e] : b0[i][e]);
                        *reinterpret_cast<u32x2_t*>(st + j * 16 * SROW + i * 32) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    }
                }
            };
            // almost every tile lies inside one image: no per-row select between the two images' row biases; a product without
            // bias, row bias and scale (every dgrad launch) only rounds its accumulators
            if (xtile || (!p.bias && !p.bias2 && !p.rowbias && p.alpha == 1.f)) {          // (the second product: plain accumulators)
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f32x4_t v = acc[i][j];
                        *reinterpret_cast<u32x2_t*>(st + j * 16 * SROW + i * 32) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
                    }
            } else if (p.rowbias && m0 + P_BM > split) park(std::true_type{}); else park(std::false_type{});
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        c3p_barrier();                                     // E2: the tile is parked; the producers take it from here
        C3P_TICK(2);
    }
#ifdef SISS_PROBE
    if (p.dbg && blockIdx.x == 0 && lane == 0)
        for (int i = 0; i < 6; ++i) p.dbg[w * 8 + i] = tk[i];
#endif
}

}  // namespace

// Preconditions (checked by siss_gemm_nt): nine panels in three row-consecutive triples, Kp % 64 == 0,
// N % 128 == 0, batch == 1, rows_per_image >= 256.
int siss_launch_gemm_nt_c3p(const void* params, void* stream) {
    const NTParams& p = *reinterpret_cast<const NTParams*>(params);
    static unsigned char attr_set[2][kMaxDevices];
    const bool qs = p.qstats != nullptr;
    // 32-bit byte offsets in the store path (gemm_nt_dispatch sends larger tensors to the one-tile-per-block kernels)
    if ((long)p.M * p.ldc * 2 >= (1L << 32) || (p.R && (long)p.M * p.ldr * 2 >= (1L << 32))) return SISS_ERR_ARG;
    if (p.Cx && ((long)p.M * p.ldcx * 2 >= (1L << 32) || p.Nx % BN || p.Kp < 2 * BK || (long)p.Nx * p.Kp * 2 >= (1L << 32))) return SISS_ERR_ARG;
    using kern_t = void (*)(const NTParams);
    const kern_t kern = qs ? gemm_nt_c3p_kernel<true> : gemm_nt_c3p_kernel<false>;
    if (siss_ensure_smem((const void*)kern, P_SMEM, attr_set[qs ? 1 : 0]) != SISS_OK) return SISS_ERR_LAUNCH;
    siss_count_dispatch(SISS_K_NT_C3P);
    // Grid = the FEWEST blocks (a multiple of 8: XCD runs) that finish in the same number of tile rounds as the full
    // chip: 550 tiles are 3 rounds on 256 CUs (38 CUs with three tiles, 218 with two) and exactly 3 on 184 -- the idle
    // CUs' power goes to the busy ones (measured: the mid-size grids run 4 % faster on 208 CUs than on 256).
    const int maxb = nt_c3p_blocks();
    const long ntiles = (long)((p.M + P_VALID - 1) / P_VALID) * (p.N / BN + (p.Cx ? p.Nx / BN : 0));
    const long rounds = (ntiles + maxb - 1) / maxb;
    int nb = (int)((ntiles + rounds - 1) / rounds);
    nb = (nb + 7) & ~7;
    if (nb > maxb) nb = maxb;
    if (nb < 8) nb = 8;
    kern<<<dim3(nb), P_THREADS, P_SMEM, (hipStream_t)stream>>>(p);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}
