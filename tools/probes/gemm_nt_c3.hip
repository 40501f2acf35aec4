// 3x3-convolution kernel for the LARGE grids (fprop and dgrad of the stride-1 convs at 256^2 / 128^2).
//
// Measured on gemm_nt.hip's 128x128 tile (tools/ablate_nt2.sh, 128->128 @256^2): the L1->LDS DMA phase alone
// runs at 59 B/clk/CU (the vL1D path peaks at 64), the MFMA phase alone at 85 % of the MFMA rate, and the two
// do not overlap: DMA writes and fragment reads share the LDS array, and per 128x128x64 product the plain
// tile moves 32 KiB in and reads 64 KiB out (768 LDS cycles against 512 MFMA cycles).  This kernel cuts both:
//
//   * "group" = (filter row ky, 64-channel K chunk).  In the flat padded-row space the three kx taps of a
//     filter row read the SAME rows shifted by 0/1/2, so the A tile is staged ONCE per group (256 rows x 128 B)
//     beside the three taps' weight tiles (3 x 16 KiB): 80 KiB in for six 128x128x64 products = 13.3 KiB per
//     product instead of 32;
//   * 256-row tile, 4 waves, each wave a 128x64 sub-tile (8 x 4 accumulators): 12 ds_read_b128 per 32 MFMAs
//     instead of 16;
//   * two barriers per GROUP (192 MFMAs per wave between them) instead of per K-step; 80 KiB of LDS -> two
//     blocks per CU cover each other's DMA phase and epilogue.
//
// A tile covers 254 output rows: rows m0 .. m0+253 need staged rows m0 .. m0+255 (+ the group's base shift),
// exactly 32 DMA pieces of 8 rows; the MFMA rows 254/255 read past the staged image and are never stored.
//
// LDS rows are 128 B (eight 16-B chunks).  chunk ^= (row & 6) is conflict-free for every ds_read_b128 lane
// group at row shifts 0, 1 and 2 (exhaustive search over the XOR-linear swizzles: tools/swizzle_search.py).
#include "nt_common.h"
#include <stdlib.h>

namespace {

constexpr int G_BM = 256, G_VALID = 254, G_THREADS = 256;
constexpr int G_ABYTES = G_BM * 128;                       // 32,768
constexpr int G_WBYTES = BN * 128;                         // 16,384 per tap
constexpr int G_SMEM = G_ABYTES + 3 * G_WBYTES;            // 81,920 -> 2 blocks / CU  (epilogue image 69,632 fits)
static_assert(G_BM * kCRow <= G_SMEM, "epilogue image must fit in the staging buffer");

__device__ __forceinline__ int swz3(int row) { return row & 6; }

__global__ __launch_bounds__(G_THREADS, 2) void gemm_nt_c3_kernel(const NTParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + G_VALID - 1) / G_VALID;
    const int nwg = tiles_n * tiles_m;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * G_VALID, n0 = tn * BN;
    if (p.ablate >> 8) {
        // stagger probe: the two co-resident blocks of a CU start half a tile apart (dispatch slot parity)
        if ((blockIdx.x >> 8) & 1)
            for (int i = 0; i < (p.ablate >> 8); ++i) __builtin_amdgcn_s_sleep(127);
    }

    // DMA pieces: 8 rows x 128 B per wave-instruction.  A: 32 pieces, wave w takes 8w .. 8w+7.
    // W: 16 pieces per tap, wave w takes 4w .. 4w+3 of each tap.  Lane (row = lane>>3, physical chunk =
    // lane&7) fetches logical chunk phys ^ swz3(row).  Rows past the tensor are clamped into the guard band.
    const int prow = lane >> 3, pc = lane & 7;
    const bf16_t* asrc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int row = (w * 8 + j) * 8 + prow;
        int gr = m0 + row; gr = gr < p.M + 1 ? gr : p.M + 1;
        asrc[j] = p.A + (long)gr * p.lda + ((pc ^ swz3(row)) << 3);
    }
    const bf16_t* wsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = (w * 4 + j) * 8 + prow;
        int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
        wsrc[j] = p.W + (long)gn * p.Kp + ((pc ^ swz3(row)) << 3);
    }
    const int kchunks = p.Kp / BK;
    const long wtap = (long)p.N * p.Kp;

    f32x4_t acc[4][8];   // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // Fragment addresses.  Row (wm*128 + j*16 + frow + t) has swizzle (frow + t) & 6 -- the tile / m-tile offsets
    // are multiples of 16 rows -- so one base per tap plus compile-time offsets (j * 2 KiB) addresses all A
    // fragments; kk = 1 flips chunk bit 2 = byte bit 6.
    const int frow = lane & 15, fq = lane >> 4;
    int a_base[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) a_base[t] = (wm * 128 + frow + t) * 128 + ((fq ^ swz3(frow + t)) << 4);
    const int w_base = G_ABYTES + (wn * 64 + frow) * 128 + ((fq ^ swz3(frow)) << 4);

    // The MFMA phase of a group is 12 blocks of 16 MFMAs: block b = (tap t, k-half kk, m-half h).  The
    // fragments of block b+1 (4 A, and 4 W when it starts a new (t, kk)) are read from LDS BEFORE block b's
    // MFMAs are issued, so a wave that has its SIMD to itself (the co-resident block is in its DMA phase or its
    // epilogue) still issues MFMAs back to back instead of exposing an LDS round trip per block.
    bf16x8_t wf[2][4], af[2][4];
    auto ldA = [&](bf16x8_t (&dst)[4], int t, int kk, int h) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj)
            dst[jj] = *reinterpret_cast<const bf16x8_t*>(smem + (a_base[t] ^ (kk << 6)) + (h * 4 + jj) * 2048);
    };
    auto ldW = [&](bf16x8_t (&dst)[4], int t, int kk) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            dst[i] = *reinterpret_cast<const bf16x8_t*>(smem + (w_base ^ (kk << 6)) + t * G_WBYTES + i * 2048);
    };

    int ky = 0, kc = 0;
    const int ngroups = 3 * kchunks;
    for (int g = 0; g < ((p.ablate & 16) ? 0 : ngroups); ++g) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of the previous group retired
        __builtin_amdgcn_s_barrier();
        if (!((p.ablate & 2) && g >= 1)) {
            const long aoff = (long)p.shift[3 * ky] * p.lda + p.coff[3 * ky] + kc * BK;
            const long woff = 3L * ky * wtap + kc * BK;
#pragma unroll
            for (int j = 0; j < 8; ++j) glds16(asrc[j] + aoff, smem + (w * 8 + j) * 1024);
#pragma unroll
            for (int t = 0; t < 3; ++t)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    glds16(wsrc[j] + woff + t * wtap, smem + G_ABYTES + t * G_WBYTES + (w * 4 + j) * 1024);
        }
        if (++kc == kchunks) { kc = 0; ++ky; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(p.ablate & 4)) {
            ldW(wf[0], 0, 0);
            ldA(af[0], 0, 0, 0);
#pragma unroll
            for (int b = 0; b < 12; ++b) {
                const int h = b & 1;
                // first MFMA row of the block, THEN the next block's fragment reads (so that the compiler's
                // lgkmcnt(0) at the top of the next block waits only for reads that had 12 MFMAs to land)
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    acc[0][h * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(b >> 1) & 1][0], af[b & 1][jj],
                                                                                  acc[0][h * 4 + jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (b + 1 < 12) {
                    const int nb = b + 1, nt = nb >> 2, nkk = (nb >> 1) & 1, nh = nb & 1;
                    ldA(af[nb & 1], nt, nkk, nh);
                    if (nh == 0) ldW(wf[(nb >> 1) & 1], nt, nkk);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 1; i < 4; ++i)
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
                        acc[i][h * 4 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[(b >> 1) & 1][i], af[b & 1][jj],
                                                                                      acc[i][h * 4 + jj], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    nt_epilogue<G_BM, G_THREADS, 8>(p, acc, smem, m0, n0, 0, tid, wm, wn, frow, fq, true, G_VALID);
}

}  // namespace

// Called by siss_gemm_nt() when the panel list is a 3x3 filter (three row-consecutive triples), Kp % 64 == 0,
// batch == 1 and rows_per_image >= 256 (a 256-row tile then spans at most two images).
int siss_launch_gemm_nt_c3(const void* params, void* stream) {
    const NTParams& p = *reinterpret_cast<const NTParams*>(params);
    static unsigned char attr_set[kMaxDevices];
    if (siss_ensure_smem((const void*)gemm_nt_c3_kernel, G_SMEM, attr_set) != SISS_OK) return SISS_ERR_LAUNCH;
    siss_count_dispatch(SISS_K_NT_C3);
    dim3 grid(cdiv(p.M, G_VALID) * cdiv(p.N, BN));
    gemm_nt_c3_kernel<<<grid, G_THREADS, G_SMEM, (hipStream_t)stream>>>(p);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}
