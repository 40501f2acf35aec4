// 3x3-convolution specialisation of the panelled NT GEMM (fprop and dgrad of the stride-1 convs).
//
// gemm_nt.hip treats the nine taps as nine independent K-panels: every tap re-streams its own
// 128-row A tile from L2, and the measured kernel is bound by that data movement (DMA + epilogue alone
// take 2/3 of its time; MFMA alone 1/3).  In the flat padded-row space the three kx taps of one filter
// row read the SAME rows shifted by one, so here a "super-step" = (filter row ky, 32-channel K chunk):
//
//   * A is staged ONCE per super-step as 128 + 2 (+pad) rows x 64 B and read at row offsets 0 / 1 / 2;
//   * the three taps' weight tiles (128 x 64 B each) are staged next to it;
//   * 48 MFMAs per wave between the two barriers (was 32), 1/1.5 of the L2->LDS bytes and DMA
//     instructions per FLOP;
//   * single-buffered, 34 KiB of LDS -> four blocks per CU hide each other's load phases.
//
// LDS rows are 64 B (four 16-B chunks); chunk index XOR ((row >> 2) & 1) << 1 makes every ds_read_b128
// lane group conflict-free for all three row shifts (exhaustive check in DESIGN.md / tools).
#include "nt_common.h"
#include <stdlib.h>

namespace {

constexpr int F_BM = 128, F_BK = 32;
constexpr int F_AROWS = 144;                               // 128 + 2 halo rows, padded to 9 DMA pieces of 16 rows
constexpr int F_ABYTES = F_AROWS * 64;                     // 9,216
constexpr int F_WBYTES = BN * 64;                          // 8,192 per tap
constexpr int F_STAGE = F_ABYTES + 3 * F_WBYTES;           // 33,792
constexpr int F_SMEM = F_STAGE > F_BM * kCRow ? F_STAGE : F_BM * kCRow;   // 34,816 -> 4 blocks / CU
constexpr int F_THREADS = 256;

__device__ __forceinline__ int swz64(int row) { return ((row >> 2) & 1) << 1; }

__global__ __launch_bounds__(F_THREADS, 4) void gemm_nt_conv3_kernel(const NTParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + F_BM - 1) / F_BM;
    const int nwg = tiles_n * tiles_m;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * F_BM, n0 = tn * BN;

    // DMA pieces: 16 rows x 64 B per wave-instruction.  33 pieces per super-step: 9 (A) + 3 x 8 (W);
    // wave w takes pieces w, w+4, ...  Per piece the lane's source is base + row * ld + logical chunk.
    // A piece q covers tile rows 16q .. 16q+15 = global rows m0 + row + shift(ky, kx=0); rows past the
    // tensor are clamped (they only feed discarded outputs) so no read leaves the guard band.
    const bf16_t* asrc[3];   // pieces w, w+4, w+8 (the last only for w == 0)
    const bf16_t* wsrc[6];   // pieces w, w+4, ..., w+20 of the 24 weight pieces
    const int prow = lane >> 2, pc = lane & 3;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int row = (w + 4 * j) * 16 + prow;
        int gr = m0 + row; gr = gr < p.M + 1 ? gr : p.M + 1;
        asrc[j] = p.A + (long)gr * p.lda + ((pc ^ swz64(row)) << 3);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) {
        const int piece = w + 4 * j, tap = piece >> 3, row = (piece & 7) * 16 + prow;
        int gn = n0 + row; gn = gn < p.N ? gn : p.N - 1;
        wsrc[j] = p.W + ((long)tap * p.N + gn) * p.Kp + ((pc ^ swz64(row)) << 3);
    }
    const int kchunks = p.Kp / F_BK;
    const long wrow3 = 3L * p.N * p.Kp;                   // weights advance three taps per filter row

    auto stage = [&](int ky, int kc) {
        const long aoff = (long)p.shift[3 * ky] * p.lda + p.coff[3 * ky] + kc * F_BK;
        const long woff = ky * wrow3 + kc * F_BK;
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16(asrc[j] + aoff, smem + (w + 4 * j) * 1024);
        if (w == 0) glds16(asrc[2] + aoff, smem + 8 * 1024);
#pragma unroll
        for (int j = 0; j < 6; ++j) glds16(wsrc[j] + woff, smem + F_ABYTES + (w + 4 * j) * 1024);
    };

    f32x4_t acc[4][4];   // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment offsets: lane (frow, fq) reads 16 B = k-group fq of row (tile row + tap shift)
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[3][4], w_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int row = wm * 64 + j * 16 + frow + t;
            a_off[t][j] = row * 64 + ((fq ^ swz64(row)) << 4);
        }
        const int rw = wn * 64 + j * 16 + frow;
        w_off[j] = F_ABYTES + rw * 64 + ((fq ^ swz64(rw)) << 4);
    }

    int ky = 0, kc = 0;
    const int nss = 3 * kchunks;
    for (int s = 0; s < nss; ++s) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // my reads of the previous super-step retired
        __builtin_amdgcn_s_barrier();
        if (!((p.ablate & 2) && s >= 1)) stage(ky, kc);
        if (++kc == kchunks) { kc = 0; ++ky; }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (!(p.ablate & 4))
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8_t*>(smem + a_off[t][i]);
                wf[i] = *reinterpret_cast<const bf16x8_t*>(smem + w_off[i] + t * F_WBYTES);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    nt_epilogue<F_BM, F_THREADS>(p, acc, smem, m0, n0, 0, tid, wm, wn, frow, fq);
}

}  // namespace

// Called by siss_gemm_nt() when the panel list is a 3x3 filter (three row-consecutive triples).
int siss_launch_gemm_nt_conv3(const void* params, void* stream) {
    const NTParams& p = *reinterpret_cast<const NTParams*>(params);
    static unsigned char attr_set[kMaxDevices];
    if (siss_ensure_smem((const void*)gemm_nt_conv3_kernel, F_SMEM, attr_set) != SISS_OK) return SISS_ERR_LAUNCH;
    siss_count_dispatch(SISS_K_NT_CONV3);
    dim3 grid(cdiv(p.M, F_BM) * cdiv(p.N, BN));
    gemm_nt_conv3_kernel<<<grid, F_THREADS, F_SMEM, (hipStream_t)stream>>>(p);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}
