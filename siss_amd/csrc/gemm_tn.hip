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

namespace {

constexpr int BN = 128, BC = 128, BR = 64;   // output tile 128(n) x 128(c); 64 reduction rows / step
constexpr int kThreads = 256;
constexpr int kTile = BR * 256;              // 16 KiB per operand tile
constexpr int kStageBytes = 2 * kTile;
constexpr int kSmemBytes = 2 * kStageBytes;
constexpr int kMaxPanels = 9;

struct TNParams {
    const bf16_t* Y; const bf16_t* X; float* dW; const bf16_t* zero_page;
    long ldy, ldx, set_stride;
    long x_set_rows;
    int N, C, npanels, nsets, nsplits;
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

__global__ __launch_bounds__(kThreads, 2) void gemm_tn_kernel(const TNParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = w >> 1, wc = w & 1;
    // Grid is 1-D.  Logical order: panel (tap) fastest, then tile, then split, then set -- and each XCD
    // (blocks b, b+8, ... share an L2) gets a CONTIGUOUS run of logical blocks, so the nine taps that
    // stream the same Y / X rows run together on one XCD and eight of the nine re-reads hit in L2
    // (measured before this remap: 5 % L2 hit rate, 9x the operand bytes from HBM).
    const int tiles_c = (p.C + BC - 1) / BC, tiles_n = (p.N + BN - 1) / BN;
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int pn = bid % p.npanels; bid /= p.npanels;
    const int tile = bid % (tiles_c * tiles_n); bid /= tiles_c * tiles_n;
    const int split = bid % p.nsplits;
    const int set = bid / p.nsplits;
    const int tn = tile / tiles_c, tc = tile - tn * tiles_c;
    const int n0 = tn * BN, c0 = tc * BC;
    const int r0 = p.row_begin + split * p.rows_per_split;
    int r1 = r0 + p.rows_per_split; r1 = r1 < p.row_end ? r1 : p.row_end;
    if (r0 >= r1) return;
    const int steps = (r1 - r0 + BR - 1) / BR;

    // staging: 4 pieces of 4 rows (256 B each) per wave and operand
    const bf16_t* ysrc[4];
    const bf16_t* xsrc[4];
    const bf16_t* zsrc = p.zero_page + (lane & 15) * 8;
    int trow[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = w * 16 + j * 4 + (lane >> 4);
        const int lc = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
        trow[j] = row;
        const long ry = (long)set * p.rows_per_set + r0 + row;
        const long rx = (long)set * p.x_set_rows + r0 + row + p.shift[pn];
        ysrc[j] = p.Y + ry * p.ldy + n0 + lc * 8;
        xsrc[j] = p.X + rx * p.ldx + p.coff[pn] + c0 + lc * 8;
    }
    auto stage = [&](int buf, int step) {
        char* base = smem + buf * kStageBytes + (w * 16) * 256;
        const int rbase = r0 + step * BR;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = rbase + trow[j] < r1;
            glds16(ok ? ysrc[j] + (long)step * BR * p.ldy : zsrc, base + j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = rbase + trow[j] < r1;
            glds16(ok ? xsrc[j] + (long)step * BR * p.ldx : zsrc, base + kTile + j * 1024);
        }
    };

    f32x4_t acc[4][4];   // [n-tile][c-tile]
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // transposed-read addresses: 16-lane group g covers reduction rows 8g..8g+7 of a 32-row
    // k-step in two 4-row blocks (h); lane 4q+pp of the group addresses row q, columns 4pp..4pp+3.
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    int y_off[2], x_off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int row = 8 * g + 4 * h + q;
        const int sw = (q << 2) | ((2 * g + h) & 3);
        y_off[h] = row * 256 + ((((wn * 8) | (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1);
        x_off[h] = kTile + row * 256 + ((((wc * 8) | (pp >> 1)) ^ sw) << 4) + 8 * (pp & 1);
    }

    // One barrier per K-step: the barrier at the top of step s orders (a) every wave's counted wait for
    // its own step-s DMA (RAW on buf) and (b) every wave's last read of buf^1 in step s-1 (WAR for the
    // restage issued right after it).
    stage(0, 0);
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // own DMA landed, own LDS reads retired
        __builtin_amdgcn_s_barrier();
        if (s + 1 < steps) stage(buf ^ 1, s + 1);
        const char* sb = smem + buf * kStageBytes;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t yf[4], xf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                s16x4_t a0 = tr_read(sb + ((y_off[0] ^ (i << 5)) + kk * 8192));
                s16x4_t a1 = tr_read(sb + ((y_off[1] ^ (i << 5)) + kk * 8192));
                s16x4_t b0 = tr_read(sb + ((x_off[0] ^ (i << 5)) + kk * 8192));
                s16x4_t b1 = tr_read(sb + ((x_off[1] ^ (i << 5)) + kk * 8192));
                yf[i] = bf16x8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                xf[i] = bf16x8_t{b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    }

    // acc[i][j][r]: n = n0 + wn*64 + i*16 + (lane>>4)*4 + r, c = c0 + wc*64 + j*16 + (lane&15)
    float* out = p.dW + (long)set * p.set_stride + (long)pn * p.N * p.C;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = c0 + wc * 64 + j * 16 + (lane & 15);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = n0 + wn * 64 + i * 16 + g * 4 + r;
                if (n < p.N && c < p.C) atomicAdd(out + (long)n * p.C + c, acc[i][j][r]);
            }
        }
}

}  // namespace

extern "C" {

// dW must be zeroed (or hold the running sum for gradient accumulation) before the call.
// Rows [row_begin, row_end) of every set are reduced; shifts/coffs are HOST arrays.
int siss_gemm_tn(const void* Y, long ldy, const void* X, long ldx, float* dW, long set_stride, int N, int C,
                 int npanels, const int* shifts, const int* coffs, int nsets, int rows_per_set,
                 long x_set_rows, int row_begin, int row_end, int nsplits, const void* zero_page,
                 void* stream) {
    SISS_CHECK_ARG(Y && X && dW && shifts && coffs && zero_page);
    SISS_CHECK_ARG(N > 0 && C > 0 && npanels >= 1 && npanels <= kMaxPanels && nsets >= 1 && nsplits >= 1);
    SISS_CHECK_ARG(ldy % 8 == 0 && ldx % 8 == 0 && N % 8 == 0 && C % 8 == 0);
    SISS_CHECK_ARG(((uintptr_t)Y | (uintptr_t)X | (uintptr_t)zero_page) % 16 == 0 && (uintptr_t)dW % 4 == 0);
    SISS_CHECK_ARG(row_begin >= 0 && row_end > row_begin && row_end <= rows_per_set);
    TNParams p;
    p.Y = (const bf16_t*)Y; p.X = (const bf16_t*)X; p.dW = dW; p.zero_page = (const bf16_t*)zero_page;
    p.ldy = ldy; p.ldx = ldx; p.set_stride = set_stride; p.x_set_rows = x_set_rows;
    p.N = N; p.C = C; p.npanels = npanels; p.nsets = nsets; p.nsplits = nsplits;
    p.rows_per_set = rows_per_set; p.row_begin = row_begin; p.row_end = row_end;
    int rps = cdiv(row_end - row_begin, nsplits);
    rps = cdiv(rps, BR) * BR;
    p.rows_per_split = rps;
    for (int i = 0; i < kMaxPanels; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    for (int i = 0; i < npanels; ++i) SISS_CHECK_ARG(p.coff[i] % 8 == 0);
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)gemm_tn_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kSmemBytes) != hipSuccess)
            return SISS_ERR_LAUNCH;
        attr_set = true;
    }
    dim3 grid(cdiv(N, BN) * cdiv(C, BC) * npanels * nsets * nsplits);
    gemm_tn_kernel<<<grid, kThreads, kSmemBytes, (hipStream_t)stream>>>(p);
    SISS_LAUNCH_RET();
}

}  // extern "C"
