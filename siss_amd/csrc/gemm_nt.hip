// Panelled NT GEMM on bf16 MFMA: the conv3x3 / conv1x1 / linear fprop and dgrad kernel
// (SURVEY.md §2b K2-K4, K7 projections).
//
//   C[r, n] = alpha * sum_p sum_k A[r + shift_p, coff_p + k] * W[p][n][k]  (+ bias[n] + rowbias[img(r)][n] + R[r, n])
//
// Activations are NHWC with a one-pixel zero halo, flattened to rows r = (img, y, x) of C
// contiguous channels.  In that flat space a 3x3 convolution is nine row-shifted GEMM panels
// (shift = (ky-1)*(W+2) + (kx-1)): no im2col, no per-pixel bounds checks -- zero padding comes
// from the halo rows, which every producer keeps at zero (this kernel writes zeros there).
// dgrad is the same kernel with negated shifts and the [tap][ci][co] weight copy.
//
// Tiling (gfx950): 128x128 output tile, BK = 64, 256 threads = 4 waves (2x2), each wave a
// 64x64 sub-tile as 4x4 v_mfma_f32_16x16x32_bf16 accumulators.  Both operands are K-contiguous
// rows, staged global->LDS by global_load_lds_dwordx4 (16 B/lane, 8 rows x 128 B per
// wave-instruction) into a double buffer.  LDS rows are 128 B; the 16-B chunk index is XORed
// with (row>>1)&7 -- applied on the per-lane SOURCE address (the DMA destination is
// lane-linear) and again on the ds_read_b128 address -- which makes every 16-lane read group
// hit 16 distinct 16-B slots of the 256-B bank row (conflict-free).
// The MFMA takes the weight fragment as its A operand so that each lane ends up with four
// CONSECUTIVE output channels of one pixel: the epilogue stages f32 through LDS (528-B padded
// rows) and writes 16-B coalesced bf16 rows.
#include "nt_common.h"
#include <stdlib.h>

namespace {

// Instantiations (BM = 128 rows, 4 waves):
//   STAGES = 1  large grids: single-buffered blocks at 4 per CU (latency hidden by the other three);
//   STAGES = 2  mid-size grids: two double-buffered blocks per CU, both k-halves' fragments in registers;
//   STAGES = 4  grids of at most 256 tiles (the 8x8 .. 32x32 layers): a 4-deep ring, optionally split-K.
//   BNT = 160 (STAGES = 1 only): 128 x 160 tiles for widths that are multiples of 160 but not of 128 (SD v1.5's 320-wide level: two
//   column tiles instead of three, no dead MFMA columns), wave tiles 64 x 80, three blocks per CU.
template <int BM, int NW, int STAGES, int BNT = BN>
struct Cfg {
    static constexpr int kThreads = NW * 64;
    static constexpr int kStageBytes = (BM + BNT) * BK * 2;
    static constexpr int kRing = STAGES * kStageBytes;
    static constexpr int kCRowT = BNT * 2 + 16;
    static constexpr int kSmemBytes = kRing > BM * kCRowT ? kRing : BM * kCRowT;
    static constexpr int kAPieces = BM / 8 / NW;        // 8-row DMA pieces per wave
    static constexpr int kWPieces = BNT / 8 / NW;
    static constexpr int kNT = BNT / 32;                // 16-column n-tiles per wave: 4 (64-wide wave tile) or 5 (80)
    static constexpr int kPerStage = kAPieces + kWPieces;
    static constexpr int kMT = BM / (NW / 2) / 16;      // 16-row m-tiles per wave: 4 (64x64 wave tile) or 8 (128x64)
};

template <int BM, int NW, int STAGES, int BNT = BN>
__global__ __launch_bounds__(NW * 64, (STAGES == 1 && BM == 128 ? (BNT == BN ? 4 : 3) : 2)) void gemm_nt_kernel(const NTParams p) {
    using C_ = Cfg<BM, NW, STAGES, BNT>;
    constexpr int MT = C_::kMT, NTL = C_::kNT;
    static_assert(BNT == BN || STAGES == 1, "the wide tile exists for the large grids only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    // XCD-aware tile order: blocks b, b+8, ... share an L2; give each XCD a contiguous run of
    // tiles (n-tiles of one m-tile adjacent) so shifted A panels and weights hit in L2.
    const int tiles_n = (p.N + BNT - 1) / BNT, tiles_m = (p.M + BM - 1) / BM;
    const int nz = p.nphase ? p.nphase : 1;             // phases: (tile, phase) blocks on grid.x, a tile's phases adjacent
    const int nwg = tiles_n * tiles_m * nz;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    int bz = blockIdx.z;
    if (p.nphase) { bz = bid % nz; bid /= nz; }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BNT;
    const bf16_t* A = p.A + (long)bz * p.strideA;
    const bf16_t* W = p.W + (long)bz * p.strideW;

    // per-lane staging sources (8 rows x 128 B per wave-instruction)
    const bf16_t* asrc[C_::kAPieces];
    const bf16_t* wsrc[C_::kWPieces];
#pragma unroll
    for (int j = 0; j < C_::kAPieces; ++j) {
        const int row = (w * C_::kAPieces + j) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        int gr = m0 + row; gr = gr < p.M ? gr : p.M - 1;
        asrc[j] = A + (long)gr * p.lda + lc * 8;
    }
#pragma unroll
    for (int j = 0; j < C_::kWPieces; ++j) {
        const int row = (w * C_::kWPieces + j) * 8 + (lane >> 3);
        const int lc = (lane & 7) ^ ((row >> 1) & 7);
        int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
        if (BNT == BN && p.gf_y) gn = (n0 >> 1) + (row >> 6) * (p.N >> 1) + (row & 63);   // GEGLU forward: 64 a rows, then the 64 g rows of the same columns
        wsrc[j] = W + (long)gn * p.Kp + lc * 8;
    }
    const int kchunks = p.Kp / BK;
    const long wpanel = (long)p.N * p.Kp;
    // split-K (small grids: the 8x8 / 16x16 layers leave most CUs idle and are bound by the serial K loop):
    // block (tile, blockIdx.y) runs K-steps [s_begin, s_begin + steps) and parks its raw accumulators in a slab
    const int pn0 = p.nphase ? p.ph_p0[bz] : 0, pn1 = p.nphase ? p.ph_p0[bz + 1] : p.npanels;
    int s_begin = 0, steps = (pn1 - pn0) * kchunks;
    if (p.ksplit > 1) {
        const int all = steps;
        s_begin = (int)((long)all * blockIdx.y / p.ksplit);
        steps = (int)((long)all * (blockIdx.y + 1) / p.ksplit) - s_begin;
    }

    // (panel, k-chunk) of the NEXT stage() call; steps are staged in order.  The panel's row shift / channel
    // offset live in the kernel arguments: they are fetched right AFTER a stage's DMA has been issued, so the
    // scalar-load latency hides behind the compute phase instead of sitting between the barrier and the DMA.
    int st_pn = s_begin / kchunks, st_kc = s_begin - st_pn * kchunks;
    st_pn += pn0;
    long a_base = (long)p.shift[st_pn] * p.lda + p.coff[st_pn], w_base = st_pn * wpanel;
    const unsigned smem_a = lds_addr(smem);
    auto stage = [&](int buf, int step) {
        (void)step;
        const long aoff = a_base + st_kc * BK;
        const long woff = w_base + st_kc * BK;
        const unsigned base = smem_a + buf * C_::kStageBytes;
#pragma unroll
        for (int j = 0; j < C_::kAPieces; ++j) glds16_asm(asrc[j] + aoff, base + (w * C_::kAPieces + j) * 1024);
#pragma unroll
        for (int j = 0; j < C_::kWPieces; ++j) glds16_asm(wsrc[j] + woff, base + BM * 128 + (w * C_::kWPieces + j) * 1024);
        if (++st_kc == kchunks) {
            st_kc = 0;
            if (++st_pn < pn1) {
                a_base = (long)p.shift[st_pn] * p.lda + p.coff[st_pn];
                w_base += wpanel;
            }
        }
    };

    f32x4_t acc[NTL][MT];   // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < NTL; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes within a stage), kk = 0; kk = 1 flips chunk bit 2
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[MT], w_off[NTL];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int ra = wm * (MT * 16) + i * 16 + frow;
        a_off[i] = ra * 128 + ((fq ^ ((ra >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int i = 0; i < NTL; ++i) {
        const int rw = wn * (BNT / 2) + i * 16 + frow;
        w_off[i] = BM * 128 + rw * 128 + ((fq ^ ((rw >> 1) & 7)) << 4);
    }

    // Ring of STAGES buffers, ONE barrier per K-step.  At the top of step s every wave waits for its own
    // step-s DMA (a counted vmcnt leaves the younger stages in flight) and for its LDS reads of step s-1;
    // the barrier then makes (a) all of step s visible and (b) buffer (s-1) % STAGES free, which is
    // exactly the buffer the DMA of step s + STAGES - 1 is issued into right after it.
#pragma unroll
    for (int s = 0; s < STAGES - 1; ++s)
        if (s < steps) stage(s, s);
    int buf = 0, nbuf = STAGES - 1;
    for (int s = 0; s < steps; ++s) {
        if (STAGES == 1) {
            // single buffer, two barriers per step: latency is hidden only by the OTHER resident blocks
            // (34 KiB of LDS per block -> 4 blocks per CU)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage(0, s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            // stages s .. s+STAGES-2 have been issued; step s must have landed, the younger ones may fly
            const int rem = steps - 1 - s;
            if (STAGES >= 4 && rem >= 2)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * C_::kPerStage) : "memory");
            else if (STAGES >= 3 && rem >= 1)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C_::kPerStage) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + STAGES - 1 < steps) stage(nbuf, s + STAGES - 1);
        }
        const char* sb = smem + buf * C_::kStageBytes;
        if constexpr (STAGES == 1 && BM == 128) {
            // 4 blocks per CU (128 VGPRs): no room for a second fragment set; the other three blocks hide the reads
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8_t af[MT], wf[NTL];
#pragma unroll
                for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8_t*>(sb + (a_off[i] ^ (kk << 6)));
#pragma unroll
                for (int i = 0; i < NTL; ++i) wf[i] = *reinterpret_cast<const bf16x8_t*>(sb + (w_off[i] ^ (kk << 6)));
#pragma unroll
                for (int i = 0; i < NTL; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // Both k-halves' fragments live in registers: the kk = 1 reads are issued after the first MFMA row of
            // kk = 0 and land under the remaining rows.  MFMA rows run i = 3 .. 0 so that the first row needs the
            // LAST-issued read (the compiler's wait there is lgkmcnt(0); nothing older is outstanding later).
            bf16x8_t af[2][MT], wf[2][4];
#pragma unroll
            for (int i = 0; i < MT; ++i) af[0][i] = *reinterpret_cast<const bf16x8_t*>(sb + a_off[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[0][i] = *reinterpret_cast<const bf16x8_t*>(sb + w_off[i]);
#pragma unroll
            for (int j = 0; j < MT; ++j)
                acc[3][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][3], af[0][j], acc[3][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i) af[1][i] = *reinterpret_cast<const bf16x8_t*>(sb + (a_off[i] ^ 64));
#pragma unroll
            for (int i = 0; i < 4; ++i) wf[1][i] = *reinterpret_cast<const bf16x8_t*>(sb + (w_off[i] ^ 64));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 2; i >= 0; --i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[0][i], af[0][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 3; i >= 0; --i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[1][i], af[1][j], acc[i][j], 0, 0, 0);
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    if (p.ksplit > 1) {
        // accumulator order, 16 B per lane: fully coalesced, read back the same way below
        const int tile = tm * tiles_n + tn;
        static_assert(BNT == BN || STAGES == 1, "split K keeps the 128-wide tile");
        float* base = p.slab + (long)tile * p.ksplit * (BM * BN);
        float* dst = base + (long)blockIdx.y * (BM * BN);
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                *reinterpret_cast<f32x4_t*>(dst + ((i * MT + j) * C_::kThreads + tid) * 4) = acc[i][j];
        return;                                                // gemm_nt_reduce_kernel finishes the tile
    }
    nt_epilogue<BM, C_::kThreads, MT, BNT>(p, acc, smem, m0, n0, bz, tid, wm, wn, frow, fq);
}

// Second half of a split-K product: sum the ksplit partial tiles of one output tile (L2-resident, written a few
// microseconds earlier) and run the normal epilogue (alpha / bias / row bias / residual / halo mask, bf16 rows).
// FOUR blocks per tile (grid.y = m-tile j of every wave's 64 x 64 sub-tile: a quarter of the tile's rows each) -- round 6: one block
// per tile read ksplit x 64 KiB through ONE CU's load path, 10.7 us on average for 51 launches per CelebA-HQ step.
template <int BM, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_nt_reduce_kernel(const NTParams p) {
    constexpr int kThreads = NW * 64, MT = BM / (NW / 2) / 16;
    static_assert(MT == 4 && kThreads == 256, "the row quarters of nt_epilogue's jsel assume 16 rows per store iteration");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int frow = lane & 15, fq = lane >> 4;
    const int j = blockIdx.y;
    f32x4_t a4[4];
    const float* src = p.slab + (long)blockIdx.x * p.ksplit * (BM * BN) + (j * kThreads + tid) * 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) a4[i] = *reinterpret_cast<const f32x4_t*>(src + i * MT * kThreads * 4);
    for (int k = 1; k < p.ksplit; ++k) {
        src += BM * BN;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4_t v = *reinterpret_cast<const f32x4_t*>(src + i * MT * kThreads * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) a4[i][r] += v[r];
        }
    }
    f32x4_t acc[4][MT];                                  // (register arrays take compile-time indices: the quarter goes to slot j by select)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < MT; ++jj) acc[i][jj] = jj == j ? a4[i] : f32x4_t{0.f, 0.f, 0.f, 0.f};
    nt_epilogue<BM, kThreads, MT>(p, acc, smem, tm * BM, tn * BN, 0, tid, wm, wn, frow, fq, true, BM, j);
}

template <int BM, int NW, int STAGES, int BNT = BN>
int launch_nt(const NTParams& p, int batch, hipStream_t st) {
    using C_ = Cfg<BM, NW, STAGES, BNT>;
    static unsigned char attr_set[kMaxDevices];
    if (siss_ensure_smem((const void*)gemm_nt_kernel<BM, NW, STAGES, BNT>, C_::kSmemBytes, attr_set) != SISS_OK) return SISS_ERR_LAUNCH;
    siss_count_dispatch(p.ksplit > 1 ? SISS_K_NT_SPLITK : SISS_K_NT);
    if (BNT != BN) siss_count_dispatch(SISS_K_NT_WIDE);
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BNT) * (p.nphase ? p.nphase : 1), p.ksplit > 1 ? p.ksplit : 1, p.nphase ? 1 : batch);
    gemm_nt_kernel<BM, NW, STAGES, BNT><<<grid, C_::kThreads, C_::kSmemBytes, st>>>(p);
    if (p.ksplit > 1) {
        static unsigned char attr2[kMaxDevices];
        if (siss_ensure_smem((const void*)gemm_nt_reduce_kernel<BM, NW>, BM * kCRow, attr2) != SISS_OK) return SISS_ERR_LAUNCH;
        gemm_nt_reduce_kernel<BM, NW><<<dim3(grid.x, 4), C_::kThreads, BM * kCRow, st>>>(p);
    }
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

#ifndef NT_WIDE_TILES
#define NT_WIDE_TILES 1
#endif
constexpr int g_wide_tiles = NT_WIDE_TILES;     // (probe builds: -DNT_WIDE_TILES=0 keeps the 128-wide tile everywhere)

// split-K workspace handed over by the host (siss_gemm_nt_set_workspace), one per device
float* g_slab_dev[kMaxDevices];
long g_slab_bytes_dev[kMaxDevices];

}  // namespace

int siss_launch_gemm_nt_c3p(const void* params, void* stream);     // gemm_nt_c3p.hip

static int g_c3p_blocks[kMaxDevices];      // per device; 0 = not set yet
int nt_c3p_blocks() {
    const int dev = siss_current_device();
    if (dev < 0) return 256;
    if (g_c3p_blocks[dev] == 0) g_c3p_blocks[dev] = 256;
    return g_c3p_blocks[dev];
}

void* siss_workspace(long* bytes) {
    const int dev = siss_current_device();
    if (dev < 0 || !g_slab_dev[dev]) { *bytes = 0; return nullptr; }
    *bytes = g_slab_bytes_dev[dev];
    return g_slab_dev[dev];
}

static long g_dispatch[SISS_K_COUNT];
void siss_count_dispatch(int k) { if (k >= 0 && k < SISS_K_COUNT) __atomic_fetch_add(&g_dispatch[k], 1L, __ATOMIC_RELAXED); }

extern "C" {

// Optional device workspace for the split-K path of siss_gemm_nt (small grids).  The library never allocates:
// without a workspace (or with one that is too small for a launch) that path is simply not taken.  The buffer
// is used by launches on ONE stream at a time (the partial tiles live from the product kernel to its reduce kernel).
// It must be ZERO-filled by its owner once (its last 4 KiB hold the tiles' arrival counters, which every launch leaves zero).
// The workspace belongs to the CURRENT device (hipGetDevice) -- a second device in the process gets its own.
int siss_gemm_nt_set_workspace(void* ptr, long bytes) {
    SISS_CHECK_ARG((ptr && bytes > 0 && (uintptr_t)ptr % 16 == 0) || (!ptr && bytes == 0));
    const int dev = siss_current_device();
    if (dev < 0) return SISS_ERR_LAUNCH;
    g_slab_dev[dev] = (float*)ptr; g_slab_bytes_dev[dev] = bytes;
    return SISS_OK;
}

// Diagnostics: number of launches dispatched to device kernel `kernel_id` since the last reset (process-wide):
// 0 gemm_nt_kernel, 1 gemm_nt_c3p_kernel, 2 flash_fwd_kernel, 3 flash_bwd_dkdv_kernel + flash_bwd_dq_kernel (one count per siss_flash_attn_bwd), 4 gemm_nt_kernel split-K (+ reduce),
// 5 gemm_tn_kernel<1>, 6 gemm_tn_kernel<3>, 7 GroupNorm slab kernels (forward or backward, small sites),
// 8 GroupNorm forward on the statistics its producing convolution left (no statistics pass),
// 9 attention dK / dV launches that cut the queries into chunks (few key tiles; partials in the workspace + reduce kernel).
// -1 for an unknown id.  siss_dispatch_reset() zeroes them all.  (Tests use these to prove which kernel a case ran on.)
long siss_dispatch_count(int kernel_id) {
    return kernel_id >= 0 && kernel_id < SISS_K_COUNT ? __atomic_load_n(&g_dispatch[kernel_id], __ATOMIC_RELAXED) : -1;
}
// Version of this C ABI: bumped when a struct layout changes or when the Python engines start calling a new entry point
// unconditionally (siss_amd/lib.py refuses an override library -- bench.py --lib, SISS_LIB_PATH -- that is older than it can drive).
// 5: round 5's final build (the first version counted).  6: round 6 (flash_attn32; optional and gated by lib.has: siss_gemm_nt_geglu_bwd,
// siss_zero_ranges + siss_gemm_tn_overwrite_log; nsplits = -2 reads as 0 in an older library).
int siss_abi_version() { return 6; }
int siss_dispatch_reset() {
    for (int k = 0; k < SISS_K_COUNT; ++k) __atomic_store_n(&g_dispatch[k], 0L, __ATOMIC_RELAXED);
    return SISS_OK;
}

/* Number of CUs the persistent 3x3 kernel (gemm_nt_c3p) occupies: a multiple of 8 in [8, 256]; 0 restores the default
   (256).  That kernel takes all 160 KiB of LDS of every CU it runs on for its whole duration,
   so a data-parallel run that overlaps RCCL's all-reduce with the backward may leave a few CUs to the collective
   (SISSStepper.autotune_overlap measures whether that pays on the node).  Returns the value now in effect, -1 for a
   value out of range.  Takes no stream: it only changes how LATER launches are shaped. */
int siss_gemm_nt_set_c3p_blocks(int n) {
    if (n == 0) n = 256;
    if (n < 8 || n > 256 || n % 8) return -1;
    const int dev = siss_current_device();
    if (dev < 0) return -1;
    g_c3p_blocks[dev] = n;
    return n;
}

}  // extern "C"

namespace {
// THE eligibility predicate of the persistent 3x3 kernel (gemm_nt_c3p.hip) for everything but the panel pattern: used by
// gemm_nt_dispatch AND by the host-side siss_conv3x3_*_takes queries, so the two cannot drift apart.  The kernel addresses its
// tensors by 32-bit byte offsets (tensors of 4 GiB and more stay on the generic kernels); ldr / lda2 / ldcx = 0: operand absent.
constexpr long kC3pMinTiles = 256;
bool c3p_eligible(int M, int N, int Kp, int rows_per_image, int Wp, long lda, long ldc, long ldr, long lda2, int K2, long ldcx, int Nx) {
    const long lim = 1L << 32;
    if (M <= 0 || N <= 0 || Kp <= 0 || Kp % BK || N % BN || rows_per_image < 256 || (Wp != 0 && Wp < 8)) return false;
    if ((long)cdiv(M, 128) * cdiv(N, BN) < kC3pMinTiles) return false;
    if (!((long)M * ldc * 2 < lim && (long)(M + 2) * lda * 2 < lim && (long)N * Kp * 2 < lim)) return false;
    if (ldr && !((long)M * ldr * 2 < lim)) return false;
    if (lda2 && !(K2 > 0 && K2 % BK == 0 && (long)(M + 2) * lda2 * 2 < lim && (long)N * K2 * 2 < lim)) return false;
    if (ldcx && !(Nx > 0 && Nx % BN == 0 && Kp >= 2 * BK && (long)M * ldcx * 2 < lim && (long)Nx * Kp * 2 < lim)) return false;
    return true;
}
bool is_conv3x3_pattern(const int* shifts, const int* coffs) {
    for (int g = 0; g < 3; ++g)
        if (!(shifts[3 * g + 1] == shifts[3 * g] + 1 && shifts[3 * g + 2] == shifts[3 * g] + 2 && coffs[3 * g + 1] == coffs[3 * g] &&
              coffs[3 * g + 2] == coffs[3 * g]))
            return false;
    return true;
}

int gemm_nt_dispatch(const void* A, long lda, const void* W, void* C, long ldc, const float* bias,
                     const float* rowbias, long ldrb, const void* R, long ldr, int M, int N, int Kp, int npanels,
                     const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp, float alpha,
                     int batch, long strideA, long strideW, long strideC, const float* rowsub, int mul_r, void* stream,
                     float* qstats = nullptr, int* qstats_written = nullptr, int d2s = 0, const void* A2 = nullptr, long lda2 = 0,
                     const void* W2 = nullptr, int K2 = 0, const float* bias2 = nullptr, const void* Wx = nullptr, void* Cx = nullptr,
                     long ldcx = 0, int Nx = 0, int alpha_cols = 0, const int* phase_p0 = nullptr, const void* gg_h = nullptr,
                     long gg_rows_x = 0, void* gf_y = nullptr) {
    SISS_CHECK_ARG(A && W && C && shifts && coffs);
    SISS_CHECK_ARG(alpha_cols == 0 || (alpha_cols > 0 && alpha_cols % 4 == 0 && npanels == 1 && !mul_r));   // (one-panel products: the generic kernel)
    if (qstats_written) *qstats_written = 0;
    SISS_CHECK_ARG(M > 0 && N > 0 && Kp > 0 && Kp % BK == 0 && npanels >= 1 && npanels <= kMaxPanels);
    SISS_CHECK_ARG(lda % 8 == 0 && ldc % 8 == 0 && (!R || ldr % 8 == 0) && batch >= 1);
    SISS_CHECK_ARG(((uintptr_t)A | (uintptr_t)W | (uintptr_t)C | (uintptr_t)R) % 16 == 0);
    SISS_CHECK_ARG(strideA % 8 == 0 && strideW % 8 == 0 && strideC % 8 == 0);
    SISS_CHECK_ARG(rows_per_image > 0 && (Hp == 0 || (long)Hp * Wp == rows_per_image));
    NTParams p;
    p.A = (const bf16_t*)A; p.W = (const bf16_t*)W; p.C = (bf16_t*)C;
    p.bias = bias; p.rowbias = rowbias; p.R = (const bf16_t*)R;
    p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb;
    p.strideA = strideA; p.strideW = strideW; p.strideC = strideC;
    p.M = M; p.N = N; p.Kp = Kp; p.npanels = npanels;
    p.rows_per_image = rows_per_image; p.Hp = Hp; p.Wp = Wp; p.alpha = alpha; p.alpha_cols = alpha_cols;
    p.inv_wp = Wp > 0 ? 1.0f / (float)Wp : 0.f;
    p.ksplit = 1; p.slab = nullptr; p.qstats = nullptr; p.d2s = d2s;
    p.gg_h = (const bf16_t*)gg_h; p.gg_rows_x = gg_rows_x;
    p.gf_y = (bf16_t*)gf_y;
    SISS_CHECK_ARG(!gf_y || (npanels == 1 && Hp == 0 && !d2s && !R && !rowbias && !rowsub && batch == 1 && !gg_h && N % 128 == 0 &&
                             ldc == N && (uintptr_t)gf_y % 16 == 0 && alpha == 1.f && alpha_cols == 0));
    SISS_CHECK_ARG(!gg_h || (gg_rows_x > 0 && npanels == 1 && Hp == 0 && !d2s && !R && !bias && !rowbias && !rowsub && batch == 1 &&
                             N % 8 == 0 && ldc >= 2L * N && (uintptr_t)gg_h % 16 == 0 && alpha == 1.f));
    p.A2 = (const bf16_t*)A2; p.W2 = (const bf16_t*)W2; p.bias2 = A2 ? bias2 : nullptr; p.lda2 = lda2; p.K2 = A2 ? K2 : 0;
    p.Wx = (const bf16_t*)Wx; p.Cx = (bf16_t*)Cx; p.ldcx = ldcx; p.Nx = Cx ? Nx : 0;
    SISS_CHECK_ARG(!Cx || (Wx && Nx > 0 && Nx % BN == 0 && ldcx % 8 == 0 && ldcx >= Nx && npanels == 9 && !qstats && !A2 &&
                           ((uintptr_t)Wx | (uintptr_t)Cx) % 16 == 0));
    SISS_CHECK_ARG(!A2 || (W2 && K2 > 0 && K2 % BK == 0 && lda2 % 8 == 0 && lda2 >= K2 && !R && npanels == 9 &&
                           ((uintptr_t)A2 | (uintptr_t)W2) % 16 == 0 && (!bias2 || (uintptr_t)bias2 % 16 == 0)));
#ifdef SISS_PROBE
    { const char* e = getenv("SISS_NT_ABLATE"); p.ablate = e ? atoi(e) : 0; }
    { const char* e = getenv("SISS_NT_DEBUG_PTR"); p.dbg = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
#endif
    SISS_CHECK_ARG(d2s == 0 || (d2s >= 1 && d2s <= 4 && Hp > 2 && Wp > 2 && batch == 1 && !rowsub && !mul_r));
    p.nphase = 0;
    for (int i = 0; i < 5; ++i) p.ph_p0[i] = 0;
    if (phase_p0) {                                            // the four planes' products as ONE launch (siss_gemm_nt_d2s_phases)
        SISS_CHECK_ARG(d2s && phase_p0[0] == 0 && phase_p0[4] == npanels && !qstats && !A2 && !Cx);
        for (int i = 0; i < 4; ++i) SISS_CHECK_ARG(phase_p0[i + 1] > phase_p0[i]);
        p.nphase = 4;
        for (int i = 0; i < 5; ++i) p.ph_p0[i] = phase_p0[i];
    }
    p.rowsub = rowsub; p.mul_r = mul_r;
    SISS_CHECK_ARG(!mul_r || (R && Hp == 0));              // the multiplicative epilogue has no halo form
    const int dev_ = siss_current_device();
    float* const g_slab = dev_ >= 0 ? g_slab_dev[dev_] : nullptr;
    const long g_slab_bytes = dev_ >= 0 ? g_slab_bytes_dev[dev_] : 0;
    SISS_CHECK_ARG((!rowbias && Hp == 0) || rows_per_image >= 64);   // <= 3 images per 128/256-row tile
    SISS_CHECK_ARG(N % 8 == 0 && (!rowbias || ldrb % 4 == 0) && (!bias || (uintptr_t)bias % 16 == 0));
    for (int i = 0; i < kMaxPanels; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    for (int i = 0; i < npanels; ++i) SISS_CHECK_ARG(p.coff[i] % 8 == 0);
    // 3x3 filters on grids of at least kC3pMinTiles 128-row tiles: the persistent kernel (gemm_nt_c3p.hip; A tile shared by the three
    // kx taps, DMA / store waves beside the MFMA waves).  Measured sweep of the threshold (round 1, 2048 .. 32): 256 gives the
    // shortest step.  It addresses its tensors by 32-bit byte offsets: tensors of 4 GiB and more stay on the kernels below.
    {
        const bool conv3 = npanels == 9 && batch == 1 && !rowsub && !mul_r && !d2s && is_conv3x3_pattern(p.shift, p.coff);
        if (conv3 && c3p_eligible(M, N, Kp, rows_per_image, Wp, lda, ldc, R ? ldr : 0, A2 ? lda2 : 0, K2, Cx ? ldcx : 0, Nx)) {
            if (qstats && Hp > 0 && ((uintptr_t)qstats % 16) == 0) {
                p.qstats = qstats;                          // only the persistent kernel forms them; the caller is told
                if (qstats_written) *qstats_written = 1;
            }
            return siss_launch_gemm_nt_c3p(&p, stream);
        }
    }
    SISS_CHECK_ARG(!A2 && !Cx);                                // only the persistent kernel folds a shortcut in (callers ask siss_conv3x3_*_takes first)
    // Large grids: single-buffered blocks at 4 per CU (latency hidden by the other three) measured 10-15 %
    // faster than double-buffered blocks at 2 per CU; small grids (< 4 blocks per CU) keep the double buffer.
    const long tiles128 = (long)cdiv(M, 128) * cdiv(N, BN) * batch * (p.nphase ? p.nphase : 1);
    // (more tiles than the double-buffered form keeps resident -- 2 blocks x 256 CUs -- : that form would run a second, partly
    // empty round; A/B of the threshold on one box, 2048 / 1100 / 520 / 384 / 260 tiles: SD v1.5 B = 16 112.5 / 112.8 / 111.3 /
    // 111.5 / 111.8 ms, B = 4 46.0 / 46.9 / 45.8 / 46.2 / 46.3 ms, CelebA-HQ unchanged -- e.g. 8192 x 1280 x K 10240: 640 tiles)
    if (tiles128 > 512) {
        // widths that are multiples of 160 but not of 128 (320, 960): 128 x 160 tiles -- no dead columns, two thirds of the A re-reads
        if (N % 160 == 0 && N % BN != 0 && !p.gg_h && !p.gf_y && g_wide_tiles) return launch_nt<128, 4, 1, 160>(p, batch, (hipStream_t)stream);
        return launch_nt<128, 4, 1>(p, batch, (hipStream_t)stream);
    }
    // at most one block per CU anyway: a 4-deep ring (128 KiB) keeps three K-steps of DMA in flight, so a step
    // costs its MFMA time instead of an L2 round trip (the 8x8 .. 32x32 layers are bound by the serial K loop)
    if (tiles128 <= 256) {
        // Few tiles and a long K loop: split K over up to 8 blocks per tile so that (nearly) every CU holds one
        // block; each block keeps >= 6 K-steps.  Partial tiles go through the host-provided slab.
        const int steps = npanels * (Kp / BK);
        if (batch == 1 && !p.nphase && tiles128 <= 128 && steps >= 12 && g_slab) {
            int S = (int)(256 / tiles128);
            if (S > 8) S = 8;
            if (S > steps / 6) S = steps / 6;
            if (S >= 2 && (long)tiles128 * S * 128 * BN * (long)sizeof(float) <= g_slab_bytes) {
                p.ksplit = S; p.slab = g_slab;
            }
        }
        // 129..256 tiles (the 16x16 layers): two double-buffered blocks per CU, each with half the K loop
        if (batch == 1 && !p.nphase && tiles128 > 128 && steps >= 24 && g_slab &&
            (long)tiles128 * 2 * 128 * BN * (long)sizeof(float) <= g_slab_bytes) {
            p.ksplit = 2; p.slab = g_slab;
            return launch_nt<128, 4, 2>(p, batch, (hipStream_t)stream);
        }
        return launch_nt<128, 4, 4>(p, batch, (hipStream_t)stream);
    }
    return launch_nt<128, 4, 2>(p, batch, (hipStream_t)stream);
}
}  // namespace

extern "C" {

// Flat argument list (ctypes-friendly).  shifts/coffs are HOST arrays of npanels ints.
// Returns SISS_ERR_ARG for shapes the kernel does not cover (Kp % 64, alignment, panel count).
int siss_gemm_nt(const void* A, long lda, const void* W, void* C, long ldc, const float* bias,
                 const float* rowbias, long ldrb, const void* R, long ldr, int M, int N, int Kp, int npanels,
                 const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp, float alpha,
                 int batch, long strideA, long strideW, long strideC, void* stream) {
    return gemm_nt_dispatch(A, lda, W, C, ldc, bias, rowbias, ldrb, R, ldr, M, N, Kp, npanels, shifts, coffs,
                            rows_per_image, Hp, Wp, alpha, batch, strideA, strideW, strideC, nullptr, 0, stream);
}

// siss_gemm_nt (one panel) whose alpha scales only the FIRST alpha_cols output columns (alpha_cols % 4 == 0; the others take 1): the
// query part of a fused q / k / v projection, pre-scaled by softmax_scale * log2(e) for siss_flash_attn_*_merged(..., q_prescaled = 1).
int siss_gemm_nt_alpha_cols(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const void* R, long ldr,
                            int M, int N, int Kp, float alpha, int alpha_cols, void* stream) {
    const int zero = 0;
    return gemm_nt_dispatch(A, lda, W, C, ldc, bias, nullptr, N, R, ldr, M, N, Kp, 1, &zero, &zero, 1, 0, 0, alpha, 1, 0, 0, 0,
                            nullptr, 0, stream, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, alpha_cols);
}

// siss_gemm_nt that may also hand the GroupNorm statistics of its OUTPUT to the consumer.  `qstats` (f32, siss_conv_qstats_words
// floats, 16-B aligned) receives, per (128-row half of a 254-row tile, image slot, 4-channel quad), the sum and the sum of squares
// of the stored bf16 values -- but only when the product is dispatched to the persistent 3x3 kernel (nine 3x3 panels, halo mask,
// large grid); *written (a HOST int) says whether it was (1) or whether the buffer was left untouched (0: the consumer runs its
// own statistics pass).  siss_groupnorm_fwd_qs folds the buffer.
int siss_gemm_nt_qstats(const void* A, long lda, const void* W, void* C, long ldc, const float* bias,
                        const float* rowbias, long ldrb, const void* R, long ldr, int M, int N, int Kp, int npanels,
                        const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp, float alpha,
                        float* qstats, int* written, void* stream) {
    SISS_CHECK_ARG(qstats && written);
    return gemm_nt_dispatch(A, lda, W, C, ldc, bias, rowbias, ldrb, R, ldr, M, N, Kp, npanels, shifts, coffs,
                            rows_per_image, Hp, Wp, alpha, 1, 0, 0, 0, nullptr, 0, stream, qstats, written);
}

// A ResnetBlock2D's tail in ONE product: C = conv3x3(A; W) + conv1x1(A2; W2) + bias + bias2 (+ rowbias) -- the 1x1 shortcut convolution
// (A2: the block's input, row stride lda2, K2 channels, a multiple of 64; W2 [N][K2] bf16; bias2 f32 [N] or null) rides in the 3x3
// product as K2 / 64 more K-groups per tile with an A base of their own (centre tap), instead of a launch of its own whose output
// comes back as the 3x3 product's residual.  Only the persistent 3x3 kernel does this: siss_conv3x3_sc_takes says (host side, no
// launch) whether a product of this shape lands there; when it does not, run the 1x1 product and pass its result as R to siss_gemm_nt.
// qstats / written: as siss_gemm_nt_qstats (both may be null).  Each term is accumulated in f32 and rounded ONCE (the two-launch form
// rounds the shortcut's output to bf16 first).
int siss_conv3x3_sc_takes(int M, int N, int Kp, int K2, int rows_per_image, int Wp, long lda, long ldc, long lda2) {
    return lda2 > 0 && c3p_eligible(M, N, Kp, rows_per_image, Wp, lda, ldc, 0, lda2, K2, 0, 0) ? 1 : 0;
}
int siss_conv3x3_sc(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const float* rowbias, long ldrb,
                    const void* A2, long lda2, const void* W2, int K2, const float* bias2, int M, int N, int Kp, const int* shifts,
                    const int* coffs, int rows_per_image, int Hp, int Wp, float* qstats, int* written, void* stream) {
    SISS_CHECK_ARG(A2 && W2 && shifts && coffs && is_conv3x3_pattern(shifts, coffs));     // (a 3x3 filter's nine panels: layout.conv3x3_panels)
    SISS_CHECK_ARG(siss_conv3x3_sc_takes(M, N, Kp, K2, rows_per_image, Wp, lda, ldc, lda2));
    return gemm_nt_dispatch(A, lda, W, C, ldc, bias, rowbias, ldrb, nullptr, 0, M, N, Kp, 9, shifts, coffs, rows_per_image, Hp, Wp,
                            1.0f, 1, 0, 0, 0, nullptr, 0, stream, qstats, written, 0, A2, lda2, W2, K2, bias2);
}

// The backward of that block tail in ONE product: conv2's dgrad  C = sum_taps A(shifted) . W  (nine panels, W = the transposed tap copies)
// AND the shortcut's dgrad  Cx[r, n] = sum_k A[r, k] . Wx[n][k]  (Wx [Nx][Kp] bf16 = the transposed 1x1 weight, Nx % 128 == 0, Cx rows of
// ldcx elements, halo rows zeroed) -- both read the same cotangent A.  On its own the 1x1 product is HBM-bound (it writes Nx / N times
// the 3x3 product's output); here it runs as extra column tiles of the persistent kernel.  R (optional) is added to C only.
// siss_conv3x3_dgrad_sc_takes: whether a product of this shape lands on that kernel (else: two siss_gemm_nt calls).
// (ldr: the row stride of R, 0 without one -- R is addressed by 32-bit offsets too)
int siss_conv3x3_dgrad_sc_takes(int M, int N, int Kp, int Nx, int rows_per_image, int Wp, long lda, long ldc, long ldcx, long ldr) {
    return ldcx > 0 && c3p_eligible(M, N, Kp, rows_per_image, Wp, lda, ldc, ldr, 0, 0, ldcx, Nx) ? 1 : 0;
}
int siss_conv3x3_dgrad_sc(const void* A, long lda, const void* W, void* C, long ldc, const void* R, long ldr, const void* Wx, void* Cx,
                          long ldcx, int Nx, int M, int N, int Kp, const int* shifts, const int* coffs, int rows_per_image, int Hp,
                          int Wp, void* stream) {
    SISS_CHECK_ARG(Wx && Cx && shifts && coffs && is_conv3x3_pattern(shifts, coffs));
    SISS_CHECK_ARG(siss_conv3x3_dgrad_sc_takes(M, N, Kp, Nx, rows_per_image, Wp, lda, ldc, ldcx, R ? ldr : 0));
    return gemm_nt_dispatch(A, lda, W, C, ldc, nullptr, nullptr, N, R, ldr, M, N, Kp, 9, shifts, coffs, rows_per_image, Hp, Wp,
                            1.0f, 1, 0, 0, 0, nullptr, 0, stream, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, Wx, Cx, ldcx, Nx);
}

// siss_gemm_nt whose rows are the pixels of ONE space-to-depth plane (plane = 2 py + px) of a stride-2 convolution's input: the
// epilogue writes pixel (y, x) of the plane to pixel (2y + py, 2x + px) of the FULL-resolution padded tensor C (row stride ldc;
// (2 Hp - 2) x (2 Wp - 2) padded pixels per image) and adds R (optional, same layout; may be C itself) there -- the
// depth-to-space scatter of the downsample dgrad without a pass of its own.  Four launches (one per plane) cover every
// interior pixel of C exactly once; C's halo is not touched.
int siss_gemm_nt_d2s(const void* A, long lda, const void* W, void* C, long ldc, const void* R, long ldr, int M, int N,
                     int Kp, int npanels, const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp,
                     int plane, void* stream) {
    SISS_CHECK_ARG(plane >= 0 && plane < 4);
    return gemm_nt_dispatch(A, lda, W, C, ldc, nullptr, nullptr, N, R, ldr, M, N, Kp, npanels, shifts, coffs,
                            rows_per_image, Hp, Wp, 1.0f, 1, 0, 0, 0, nullptr, 0, stream, nullptr, nullptr, 1 + plane);
}
// The same with a bias (f32 [N], added before the one rounding): one PHASE of a sub-pixel upsample convolution -- the rows are the
// LOW-resolution pixels, plane = 2 py + px is the phase, the panels its 2x2 taps (siss_upsample_phase_weights), and the epilogue
// writes pixel (y, x) to (2y + py, 2x + px) of the high-resolution output.  Four launches cover every interior pixel of C once.
int siss_gemm_nt_d2s_bias(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, int M, int N, int Kp,
                          int npanels, const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp, int plane,
                          void* stream) {
    SISS_CHECK_ARG(plane >= 0 && plane < 4);
    return gemm_nt_dispatch(A, lda, W, C, ldc, bias, nullptr, N, nullptr, 0, M, N, Kp, npanels, shifts, coffs,
                            rows_per_image, Hp, Wp, 1.0f, 1, 0, 0, 0, nullptr, 0, stream, nullptr, nullptr, 1 + plane);
}

// The FOUR planes of siss_gemm_nt_d2s / siss_gemm_nt_d2s_bias as one launch: plane z runs panels [phase_p0[z], phase_p0[z + 1]) of
// shifts / coffs / W (W holds the planes' panels back to back: [phase_p0[4]][N][Kp]; phase_p0[0] = 0, every plane at least one
// panel, at most 16 in all) and scatters to plane z.  bias and R are optional as in the single-plane entry points.  Each block
// does what the corresponding block of the single-plane launch does (same K order, same epilogue): bitwise the same C -- unless
// that launch is small enough to split K (this one has four times the blocks and never does): then one bf16 rounding apart.  A row
// tile's four planes run as adjacent blocks on one XCD, so the shifted A rows they share come out of L2, and the launch pays one
// ramp and one drain instead of four (the low-resolution sites are four latency-bound launches otherwise).
int siss_gemm_nt_d2s_phases(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const void* R, long ldr,
                            int M, int N, int Kp, const int* phase_p0, const int* shifts, const int* coffs, int rows_per_image,
                            int Hp, int Wp, void* stream) {
    SISS_CHECK_ARG(phase_p0 && phase_p0[4] >= 4 && phase_p0[4] <= kMaxPanels);
    return gemm_nt_dispatch(A, lda, W, C, ldc, bias, nullptr, N, R, ldr, M, N, Kp, phase_p0[4], shifts, coffs,
                            rows_per_image, Hp, Wp, 1.0f, 1, 0, 0, 0, nullptr, 0, stream, nullptr, nullptr, 1, nullptr, 0, nullptr, 0,
                            nullptr, nullptr, nullptr, 0, 0, 0, phase_p0);
}

// The dgrad of GEGLU's output projection with the GEGLU backward in its epilogue (diffusers FeedForward: out = a * gelu(g), [a | g] =
// h = proj(x); the reference differentiates it inside losses/ddpm_deletion_loss.py:24's unet call): acc[r][n] = sum_k A[r][k] W[n][k] is
// d(out); dh [M][2 N] receives  dh[r][n] = acc * gelu(g),  dh[r][N + n] = acc * a * gelu'(g)  with (a, g) = h[r % rows_x][n], h[..][N + n]
// (h: [rows_x][2 N] bf16).  Equal to siss_gemm_nt into a [M][N] tensor followed by siss_geglu_bwd (the cotangent is rounded to bf16
// in between in both forms) without that tensor's write and read: 4 N of 12 N bytes per row of two HBM-bound passes.
int siss_gemm_nt_geglu_bwd(const void* A, long lda, const void* W, void* dh, const void* h, long rows_x, int M, int N, int Kp,
                           void* stream) {
    SISS_CHECK_ARG(h && rows_x > 0);
    static const int zero = 0;
    return gemm_nt_dispatch(A, lda, W, dh, 2L * N, nullptr, nullptr, N, nullptr, 0, M, N, Kp, 1, &zero, &zero, 1, 0, 0, 1.0f, 1, 0, 0, 0,
                            nullptr, 0, stream, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr,
                            h, rows_x);
}

// GEGLU's input projection with the GEGLU forward in its epilogue: h[r][0 : 2F] = A[r] . W^T + bias (W: [2F][Kp], rows 0 .. F-1 the
// value half, F .. 2F-1 the gate half) AND y[r][f] = h[r][f] * gelu(h[r][F + f]) in one launch -- a tile computes 64 value columns and
// the 64 gate columns that go with them.  Equal to siss_gemm_nt into h followed by siss_geglu_fwd (y is formed from the bf16-rounded h
// in both forms) without the second pass's read of h.  F % 64 == 0.
int siss_gemm_nt_geglu_fwd(const void* A, long lda, const void* W, const float* bias, void* h, void* y, int M, int F, int Kp,
                           void* stream) {
    SISS_CHECK_ARG(y && F > 0 && F % 64 == 0);
    static const int zero = 0;
    return gemm_nt_dispatch(A, lda, W, h, 2L * F, bias, nullptr, 0, nullptr, 0, M, 2 * F, Kp, 1, &zero, &zero, 1, 0, 0, 1.0f, 1, 0, 0, 0,
                            nullptr, 0, stream, nullptr, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, nullptr, nullptr, 0, 0, 0, nullptr,
                            nullptr, 0, y);
}

// floats in the `qstats` buffer of a product with M rows and N output channels
long siss_conv_qstats_words(long M, int N) {
    if (M <= 0 || N <= 0 || N % 4) return -1;
    return ((M + kQsTileRows - 1) / kQsTileRows) * 2 * 2 * (long)(N / 4) * 2;
}

// Same product with the attention-backward epilogue  C = R o (alpha * (acc - rowsub[row]))  (R: bf16 [batch][M][N]
// with C's strides, rowsub: f32 [batch][M]):  dS = scale * P o (dO V^T - delta), delta = rowsum(dO o O), in ONE pass --
// the dP matrix is never written and the separate softmax-backward pass over P / dP / dS disappears.
int siss_gemm_nt_mulsub(const void* A, long lda, const void* W, void* C, long ldc, const void* R, long ldr,
                        const float* rowsub, int M, int N, int Kp, float alpha, int batch, long strideA, long strideW,
                        long strideC, void* stream) {
    SISS_CHECK_ARG(R && rowsub);
    static const int zero = 0;
    return gemm_nt_dispatch(A, lda, W, C, ldc, nullptr, nullptr, N, R, ldr, M, N, Kp, 1, &zero, &zero, 1, 0, 0, alpha,
                            batch, strideA, strideW, strideC, rowsub, 1, stream);
}

}  // extern "C"
