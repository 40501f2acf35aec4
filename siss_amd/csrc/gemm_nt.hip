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
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BN = 128, BK = 64;
constexpr int kCRow = BN * 2 + 16;                       // bf16 epilogue row (256 B) + 16 B pad (bank spread, 16-B aligned)
constexpr int kMaxPanels = 9;

struct NTParams {
    const bf16_t* A; const bf16_t* W; bf16_t* C;
    const float* bias; const float* rowbias; const bf16_t* R;
    long lda, ldc, ldr, ldrb;
    long strideA, strideW, strideC;   // per blockIdx.z batch (elements)
    int M, N, Kp, npanels;
    int rows_per_image, Hp, Wp;       // Hp == 0: no halo mask
    float alpha, inv_wp;
    int ablate;
    int shift[kMaxPanels];
    int coff[kMaxPanels];
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)l, 16, 0, 0);
}

// Two instantiations:
//   <256, 8 waves, 3 stages>  big layers: one 512-thread block per CU (2 waves / SIMD), 144 KiB LDS ring,
//                             the DMA of K-step s+2 is issued while step s computes (counted vmcnt), so an
//                             L2 / Infinity-Cache round trip has two compute phases to land;
//   <128, 4 waves, 2 stages>  small layers (few tiles): two 256-thread blocks per CU.
template <int BM, int NW, int STAGES>
struct Cfg {
    static constexpr int kThreads = NW * 64;
    static constexpr int kStageBytes = (BM + BN) * BK * 2;
    static constexpr int kRing = STAGES * kStageBytes;
    static constexpr int kSmemBytes = kRing > BM * kCRow ? kRing : BM * kCRow;
    static constexpr int kAPieces = BM / 8 / NW;        // 8-row DMA pieces per wave
    static constexpr int kWPieces = BN / 8 / NW;
    static constexpr int kPerStage = kAPieces + kWPieces;
};

template <int BM, int NW, int STAGES>
__global__ __launch_bounds__(NW * 64, (STAGES == 1 ? 4 : 2)) void gemm_nt_kernel(const NTParams p) {
    using C_ = Cfg<BM, NW, STAGES>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;

    // XCD-aware tile order: blocks b, b+8, ... share an L2; give each XCD a contiguous run of
    // tiles (n-tiles of one m-tile adjacent) so shifted A panels and weights hit in L2.
    const int tiles_n = (p.N + BN - 1) / BN, tiles_m = (p.M + BM - 1) / BM;
    const int nwg = tiles_n * tiles_m;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;
    const int bz = blockIdx.z;
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
        wsrc[j] = W + (long)gn * p.Kp + lc * 8;
    }
    const int kchunks = p.Kp / BK;
    const int steps = p.npanels * kchunks;
    const long wpanel = (long)p.N * p.Kp;

    int st_pn = 0, st_kc = 0;                  // (panel, k-chunk) of the NEXT stage() call: steps are staged in order
    auto stage = [&](int buf, int step) {
        (void)step;
        const int pn = st_pn, kc = st_kc;
        if (++st_kc == kchunks) { st_kc = 0; ++st_pn; }
        const long aoff = (long)p.shift[pn] * p.lda + p.coff[pn] + kc * BK;
        const long woff = pn * wpanel + kc * BK;
        char* base = smem + buf * C_::kStageBytes;
#pragma unroll
        for (int j = 0; j < C_::kAPieces; ++j) glds16(asrc[j] + aoff, base + (w * C_::kAPieces + j) * 1024);
#pragma unroll
        for (int j = 0; j < C_::kWPieces; ++j) glds16(wsrc[j] + woff, base + BM * 128 + (w * C_::kWPieces + j) * 1024);
    };

    f32x4_t acc[4][4];   // [n-tile][m-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // fragment read offsets (bytes within a stage), kk = 0; kk = 1 flips chunk bit 2
    const int frow = lane & 15, fq = lane >> 4;
    int a_off[4], w_off[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = wm * 64 + i * 16 + frow;
        a_off[i] = ra * 128 + ((fq ^ ((ra >> 1) & 7)) << 4);
        const int rw = wn * 64 + i * 16 + frow;
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
            if (!((p.ablate & 2) && s >= 2)) stage(0, s);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
        } else {
            if (STAGES == 3 && s + 1 < steps)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(C_::kPerStage) : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (s + STAGES - 1 < steps && !((p.ablate & 2) && s >= 2)) stage(nbuf, s + STAGES - 1);
        }
        const char* sb = smem + buf * C_::kStageBytes;
        if (!(p.ablate & 4))
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *reinterpret_cast<const bf16x8_t*>(sb + (a_off[i] ^ (kk << 6)));
                wf[i] = *reinterpret_cast<const bf16x8_t*>(sb + (w_off[i] ^ (kk << 6)));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
        }
        buf = buf + 1 == STAGES ? 0 : buf + 1;
        nbuf = nbuf + 1 == STAGES ? 0 : nbuf + 1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    // ---- epilogue: bf16 tile through LDS (one 34 KiB image), then 16-B coalesced rows ----
    // Registers: acc[i][j][r] = channel n = wn*64 + i*16 + fq*4 + r of pixel m = wm*64 + j*16 + frow.
    // alpha, bias and the per-image row bias (time embedding) are applied in f32 BEFORE the one rounding to
    // bf16; the residual (if any) is added after it, which is exactly the reference's autocast order
    // (conv output is bf16, then `x + h` rounds again).
    const int rpi = p.rows_per_image;
    const int img0 = m0 / rpi;                       // tile rows span at most 3 images (rows_per_image >= 64)
    const int b1 = (img0 + 1) * rpi - m0, b2 = b1 + rpi;
    {
        f32x4_t bias4[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = n0 + wn * 64 + i * 16 + fq * 4;
            bias4[i] = (p.bias && n + 4 <= p.N) ? *reinterpret_cast<const f32x4_t*>(p.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = wm * 64 + j * 16 + frow;
            const float* rb = nullptr;
            if (p.rowbias) rb = p.rowbias + (long)(img0 + (m >= b1) + (m >= b2)) * p.ldrb;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int nl = wn * 64 + i * 16 + fq * 4;
                f32x4_t v = acc[i][j];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = v[r] * p.alpha + bias4[i][r];
                if (rb && n0 + nl + 4 <= p.N) {
                    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(rb + n0 + nl);
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += t[r];
                }
                *reinterpret_cast<u32x2_t*>(smem + m * kCRow + nl * 2) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
            }
        }
    }
    __syncthreads();
    if (p.ablate & 1) return;
    bf16_t* C = p.C + (long)bz * p.strideC;
    const int chunk = tid & 15;           // 8 channels per chunk
    const int nc = n0 + chunk * 8;
    if (nc >= p.N) return;
    constexpr int kRowsPerIt = C_::kThreads / 16;
#pragma unroll 4
    for (int it = 0; it < BM / kRowsPerIt; ++it) {
        const int row = it * kRowsPerIt + (tid >> 4);
        const int r = m0 + row;
        if (r >= p.M) break;
        u32x4_t o = *reinterpret_cast<const u32x4_t*>(smem + row * kCRow + chunk * 16);
        if (p.Hp > 0) {
            const int rem = r - (img0 + (row >= b1) + (row >= b2)) * rpi;
            const int y = (int)(((float)rem + 0.5f) * p.inv_wp), x = rem - y * p.Wp;   // exact: see header note
            if ((y == 0) | (y == p.Hp - 1) | (x == 0) | (x == p.Wp - 1)) o = u32x4_t{0u, 0u, 0u, 0u};
            else if (p.R && nc + 8 <= p.N) {
                const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(p.R + (long)bz * p.strideC + (long)r * p.ldr + nc);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf2(__builtin_bit_cast(float, o[e] << 16) + __builtin_bit_cast(float, rr[e] << 16),
                                    __builtin_bit_cast(float, o[e] & 0xffff0000u) + __builtin_bit_cast(float, rr[e] & 0xffff0000u));
            }
        } else if (p.R && nc + 8 <= p.N) {
            const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(p.R + (long)bz * p.strideC + (long)r * p.ldr + nc);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                o[e] = pack_bf2(__builtin_bit_cast(float, o[e] << 16) + __builtin_bit_cast(float, rr[e] << 16),
                                __builtin_bit_cast(float, o[e] & 0xffff0000u) + __builtin_bit_cast(float, rr[e] & 0xffff0000u));
        }
        bf16_t* dst = C + (long)r * p.ldc + nc;
        if (nc + 8 <= p.N) {
            *reinterpret_cast<u32x4_t*>(dst) = o;
        } else {        // ragged N tail (N % 8 != 0 never happens for channel counts; kept for safety)
            for (int e = 0; e < 8 && nc + e < p.N; ++e) {
                const uint32_t wv = o[e >> 1];
                float v = (e & 1) ? __builtin_bit_cast(float, wv & 0xffff0000u) : __builtin_bit_cast(float, wv << 16);
                if (p.R) v += bf2f(p.R[(long)bz * p.strideC + (long)r * p.ldr + nc + e]);
                dst[e] = f2bf(v);
            }
        }
    }
}

template <int BM, int NW, int STAGES>
int launch_nt(const NTParams& p, int batch, hipStream_t st) {
    using C_ = Cfg<BM, NW, STAGES>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_nt_kernel<BM, NW, STAGES>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, C_::kSmemBytes) != hipSuccess)
            return SISS_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid(cdiv(p.M, BM) * cdiv(p.N, BN), 1, batch);
    gemm_nt_kernel<BM, NW, STAGES><<<grid, C_::kThreads, C_::kSmemBytes, st>>>(p);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

}  // namespace

extern "C" {

// Flat argument list (ctypes-friendly).  shifts/coffs are HOST arrays of npanels ints.
// Returns SISS_ERR_ARG for shapes the kernel does not cover (Kp % 64, alignment, panel count).
int siss_gemm_nt(const void* A, long lda, const void* W, void* C, long ldc, const float* bias,
                 const float* rowbias, long ldrb, const void* R, long ldr, int M, int N, int Kp, int npanels,
                 const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp, float alpha,
                 int batch, long strideA, long strideW, long strideC, void* stream) {
    SISS_CHECK_ARG(A && W && C && shifts && coffs);
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
    p.rows_per_image = rows_per_image; p.Hp = Hp; p.Wp = Wp; p.alpha = alpha;
    p.inv_wp = Wp > 0 ? 1.0f / (float)Wp : 0.f;
    { const char* e = getenv("SISS_NT_ABLATE"); p.ablate = e ? atoi(e) : 0; }
    SISS_CHECK_ARG((!rowbias && Hp == 0) || rows_per_image >= 64);   // <= 3 images per 128/256-row tile
    SISS_CHECK_ARG(N % 8 == 0 && (!rowbias || ldrb % 4 == 0) && (!bias || (uintptr_t)bias % 16 == 0));
    for (int i = 0; i < kMaxPanels; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    for (int i = 0; i < npanels; ++i) SISS_CHECK_ARG(p.coff[i] % 8 == 0);
    // big problems: 256-row tiles (one 8-wave block per CU, 3-stage ring); otherwise 128-row tiles
    const long big_tiles = (long)cdiv(M, 256) * cdiv(N, BN) * batch;
    static int force = -1;
    if (force < 0) { const char* e = getenv("SISS_NT_TILE"); force = e ? atoi(e) : 0; }
    // measured (tools/bench_kernels.py, round 1): the 8-wave / 3-stage variant is 5-15 % SLOWER than two
    // co-resident 4-wave blocks at every CelebA-HQ layer shape, so it is opt-in (SISS_NT_TILE=256)
    const bool big = force == 256 && big_tiles >= 1;
    static int stages = -1;
    if (stages < 0) { const char* e = getenv("SISS_NT_STAGES"); stages = e ? atoi(e) : 0; }
    if (big) return launch_nt<256, 8, 3>(p, batch, (hipStream_t)stream);
    // Large grids: single-buffered blocks at 4 per CU (latency hidden by the other three) measured 10-15 %
    // faster than double-buffered blocks at 2 per CU; small grids (< 4 blocks per CU) keep the double buffer.
    const long tiles128 = (long)cdiv(M, 128) * cdiv(N, BN) * batch;
    if (stages == 1 || (stages == 0 && tiles128 >= 2048)) return launch_nt<128, 4, 1>(p, batch, (hipStream_t)stream);
    return launch_nt<128, 4, 2>(p, batch, (hipStream_t)stream);
}

}  // extern "C"
