// Shared pieces of the NT GEMM kernels: parameter block, LDS-DMA helper and the epilogue.
#pragma once
#include "common.h"

namespace {

constexpr int BN = 128, BK = 64;
constexpr int kCRow = BN * 2 + 16;                       // bf16 epilogue row (256 B) + 16 B pad (bank spread, 16-B aligned)
constexpr int kMaxPanels = 16;                           // (16: the (plane, tap) panels of a sub-pixel upsample convolution's dgrad)

struct NTParams {
    const bf16_t* A; const bf16_t* W; bf16_t* C;
    const float* bias; const float* rowbias; const bf16_t* R;
    long lda, ldc, ldr, ldrb;
    long strideA, strideW, strideC;   // per blockIdx.z batch (elements)
    int M, N, Kp, npanels;
    int rows_per_image, Hp, Wp;       // Hp == 0: no halo mask
    float alpha, inv_wp;
    int alpha_cols;                   // 0: alpha scales every output column; else only columns [0, alpha_cols) (the others take 1)
    const float* rowsub;              // optional f32 [batch][M]: subtracted from the accumulator row before alpha
    int mul_r;                        // 1: the epilogue MULTIPLIES by R instead of adding it  (C = R o (alpha (acc - rowsub)))
    int ksplit;                       // > 1: split-K -- gridDim.y blocks per tile write f32 partial tiles to `slab`
    float* slab;                      //      ([tile][split][BM*BN] in accumulator order); gemm_nt_reduce_kernel sums them and runs the epilogue
#ifdef SISS_PROBE                     // probe build only (tools/probes/build_probe.sh): phase timers and ablation switches of gemm_nt_c3p
    long long* dbg;                   // timing probe buffer ($SISS_NT_DEBUG_PTR), normally null
    int ablate;                       // $SISS_NT_ABLATE: 1 no stores, 2 no DMA after the first two groups, 4 no MFMAs
#endif
    const bf16_t* gg_h;               // GEGLU backward in the epilogue (generic kernel, rows without pixel structure): the accumulator is d(out) of
    long gg_rows_x;                   // out = a * gelu(g), h = [a | g] the saved projection ([gg_rows_x][2 N], row r % gg_rows_x); C has 2 N columns:
                                      // C[r][n] = acc * gelu(g), C[r][N + n] = acc * a * gelu'(g) -- the [rows][N] cotangent never reaches HBM
    bf16_t* gf_y;                     // GEGLU forward in the epilogue (generic kernel, 128-wide tiles, rows without pixel structure): W / bias / C are the
                                      // [2 F] projection h = [a | g] (N = 2 F, F % 64 == 0); a tile holds columns [t 64, t 64 + 64) of a AND of g, stores
                                      // both into C = h and y[r][f] = a * gelu(g) (from the bf16-rounded h, as siss_geglu_fwd reads it) into gf_y [M][F]
    int d2s;                          // 0, or 1 + plane: rows are pixels of space-to-depth plane (py, px) = (plane >> 1, plane & 1); the epilogue
                                      // writes (and reads R) at the pixel's place in the FULL-resolution tensor (2 Hp - 2) x (2 Wp - 2) padded
    int nphase;                       // 0, or 2..4 PHASES in one launch (generic kernel only; d2s != 0): phase z runs the panels
    int ph_p0[5];                     // [ph_p0[z], ph_p0[z + 1]) of shift / coff / W and scatters to space-to-depth plane z -- what would
                                      // be nphase launches with d2s = 1 + z; a tile's phases are adjacent blocks of one XCD (shared A rows in L2)
    float* qstats;                    // optional (persistent 3x3 kernel only): per-(half tile, image slot, 4-channel quad) sums and
                                      // sums of squares of the bf16 OUTPUT, [2 * row tiles][2][N / 4][2] f32 -- the GroupNorm that
                                      // consumes the result folds them instead of reading the tensor a second time
    // Optional 1x1 shortcut convolution folded into a 3x3 one (persistent 3x3 kernel only; ResnetBlock2D.conv_shortcut beside conv2):
    //   C += A2[r, 0:K2] . W2[n][0:K2] + bias2[n]   -- K2 / 64 more K-groups per tile with an A base of their own, centre tap only
    const bf16_t* A2; const bf16_t* W2; const float* bias2;
    long lda2;
    int K2;
    // Optional SECOND product over the same A (persistent 3x3 kernel only; a resnet's conv_shortcut dgrad beside conv2's dgrad -- both
    // read the block's output cotangent):  Cx[r, n] = sum_k A[r, k] . Wx[n][k]  (Nx % 128 == 0 columns, row stride ldcx, halo rows
    // zeroed like C's).  It runs as Nx / 128 more column tiles per row tile, centre tap only: an HBM-bound product (it writes
    // Nx / N times the 3x3 product's output for 1/9 of its MFMAs per column) hidden inside an MFMA-bound kernel.
    const bf16_t* Wx; bf16_t* Cx;
    long ldcx;
    int Nx;
    int shift[kMaxPanels];
    int coff[kMaxPanels];
};

typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void gbl_void;

__device__ __forceinline__ void glds16(const void* g, void* l) {
    __builtin_amdgcn_global_load_lds((gbl_void*)g, (lds_void*)l, 16, 0, 0);
}


// Epilogue of a BM x 128 output tile held as acc[i][j] (i = n-tile, j = m-tile of a 64x64 wave sub-tile).
// MT = 16-row m-tiles per wave (4: 64-row wave tile, 8: 128-row wave tile); waves are laid out 2 (n) wide.
// BNT: columns of the tile (128; 160 for the widths that are multiples of 160 but not of 128 -- SD's 320: two tiles instead of three):
// a wave's sub-tile is BNT / 2 columns = BNT / 32 n-tiles, a staged row has BNT / 8 16-B chunks, one per thread and store.
template <int BM, int kThreads, int MT = 4, int BNT = BN>
__device__ __forceinline__ void nt_epilogue(const NTParams& p, f32x4_t (&acc)[BNT / 32][MT], char* smem, int m0, int n0,
                                            int bz, int tid, int wm, int wn, int frow, int fq,
                                            bool writer = true, int vrows = BM, int jsel = -1) {
    constexpr int NTL = BNT / 32, kCRow = BNT * 2 + 16, NCHK = BNT / 8;
    static_assert(BNT == 128 || BNT == 160, "tile widths");
    // jsel >= 0 (split-K reduce kernel: four blocks per tile): only the m-tile j == jsel of every wave is valid in `acc` -- the rows
    // with (row >> 4 & 3) == jsel are staged and stored, the other three quarters of the tile belong to the sibling blocks
    // ---- epilogue: bf16 tile through LDS (one 34 KiB image), then 16-B coalesced rows ----
    // Registers: acc[i][j][r] = channel n = wn*64 + i*16 + fq*4 + r of pixel m = wm*64 + j*16 + frow.
    // alpha, bias and the per-image row bias (time embedding) are applied in f32 BEFORE the one rounding to
    // bf16; the residual (if any) is added after it, which is exactly the reference's autocast order
    // (conv output is bf16, then `x + h` rounds again).
    const int rpi = p.rows_per_image;
    const int img0 = m0 / rpi;                       // tile rows span at most 3 images (rows_per_image >= 64)
    const int b1 = (img0 + 1) * rpi - m0, b2 = b1 + rpi;
    const int last_img = (p.M - 1) / rpi;
    if (writer) {
        f32x4_t bias4[NTL];
#pragma unroll
        for (int i = 0; i < NTL; ++i) {
            int n = n0 + wn * (BNT / 2) + i * 16 + fq * 4;
            if (p.gf_y) n = (n0 >> 1) + wn * (p.N >> 1) + i * 16 + fq * 4;       // (waves wn = 0 hold the a columns, wn = 1 the g columns)
            bias4[i] = (p.bias && n + 4 <= p.N) ? *reinterpret_cast<const f32x4_t*>(p.bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            if (jsel >= 0 && j != jsel) continue;
            const int m = wm * (MT * 16) + j * 16 + frow;
            float rsub = 0.f;
            if (p.rowsub) { int gr = m0 + m; gr = gr < p.M ? gr : p.M - 1; rsub = p.rowsub[(long)bz * p.M + gr]; }
            const float* rb = nullptr;
            if (p.rowbias) {                 // rows past M (tile overhang) must not index past the last image's row
                int img = img0 + (m >= b1) + (m >= b2);
                img = img < last_img ? img : last_img;
                rb = p.rowbias + (long)img * p.ldrb;
            }
#pragma unroll
            for (int i = 0; i < NTL; ++i) {
                const int nl = wn * (BNT / 2) + i * 16 + fq * 4;
                f32x4_t v = acc[i][j];
                const float al = (p.alpha_cols == 0 || n0 + nl < p.alpha_cols) ? p.alpha : 1.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = (v[r] - rsub) * al + bias4[i][r];
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
    bf16_t* C = p.C + (long)bz * p.strideC;
    constexpr int kRowsPerIt = kThreads / NCHK;          // 16, or 12 (of the 256 threads 240 store)
    const int chunk = NCHK == 16 ? (tid & 15) : tid % NCHK;           // 8 channels per chunk
    const int trow = NCHK == 16 ? (tid >> 4) : tid / NCHK;
    int nc = n0 + chunk * 8;
    if (BNT == 128 && p.gf_y) nc = (n0 >> 1) + (chunk >> 3) * (p.N >> 1) + (chunk & 7) * 8;   // chunks 0-7: a columns, 8-15: g columns
    if (nc >= p.N || trow >= kRowsPerIt) return;
    constexpr int kIts = (BM + kRowsPerIt - 1) / kRowsPerIt;
    if (p.Hp == 0 && !p.d2s && nc + 8 <= p.N) {
        // Rows without a pixel structure (linears, attention products): no halo logic, and the output / residual addresses are
        // running pointers (one 64-bit add per row instead of a 64-bit multiply-add) -- these launches have K loops of 5-20 steps,
        // so the epilogue's vector instructions weigh as much as their MFMAs.
        const int row0 = trow;
        bf16_t* dst = C + (long)(m0 + row0) * p.ldc + nc;
        const bf16_t* rsrc = p.R ? p.R + (long)bz * p.strideC + (long)(m0 + row0) * p.ldr + nc : nullptr;
        const long dstep = (long)kRowsPerIt * p.ldc, rstep = (long)kRowsPerIt * p.ldr;
        int rows_left = (p.M - m0 < vrows ? p.M - m0 : vrows) - row0;
        const char* src = smem + row0 * kCRow + chunk * 16;
        if (BNT == 128 && p.gf_y) {
            // GEGLU forward: chunks 0-7 hold 8 value columns (and find the matching gate columns 8 chunks further in the staged row),
            // chunks 8-15 the gate columns; both store h, the value threads also y = a * gelu(g).  (A loop of its own, not unrolled:
            // the four-blocks-per-CU instantiation has no register to spare.)
            bf16_t* ydst = p.gf_y + (long)(m0 + row0) * (p.N >> 1) + (n0 >> 1) + (chunk & 7) * 8;
            const long ystep = (long)kRowsPerIt * (p.N >> 1);
#pragma unroll 1
            for (int it = 0; it < kIts; ++it) {
                if (rows_left <= 0) break;
                if (jsel < 0 || (it & 3) == jsel) {
                    const u32x4_t o = *reinterpret_cast<const u32x4_t*>(src + it * kRowsPerIt * kCRow);
                    *reinterpret_cast<u32x4_t*>(dst) = o;
                    if (chunk < 8) {
                        const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(src + it * kRowsPerIt * kCRow + 128);
                        u32x4_t y;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float a0 = __builtin_bit_cast(float, o[e] << 16), a1 = __builtin_bit_cast(float, o[e] & 0xffff0000u);
                            const float g0 = __builtin_bit_cast(float, gv[e] << 16), g1 = __builtin_bit_cast(float, gv[e] & 0xffff0000u);
                            y[e] = pack_bf2(a0 * gelu_f(g0), a1 * gelu_f(g1));
                        }
                        *reinterpret_cast<u32x4_t*>(ydst) = y;
                    }
                }
                dst += dstep; ydst += ystep; rows_left -= kRowsPerIt;
            }
            return;
        }
#pragma unroll 4
        for (int it = 0; it < kIts; ++it) {
            if (rows_left <= 0) break;
            if (jsel >= 0 && (it & 3) != jsel) {             // (kRowsPerIt == 16 there: iteration it holds the rows of m-tile it & 3)
                dst += dstep; rows_left -= kRowsPerIt;
                if (rsrc) rsrc += rstep;
                continue;
            }
            u32x4_t o = *reinterpret_cast<const u32x4_t*>(src + it * kRowsPerIt * kCRow);
            if (p.gg_h) {
                // (the cotangent is rounded to bf16 first, exactly as the two-launch form stores it)
                const long rx = (long)(m0 + row0 + it * kRowsPerIt) % p.gg_rows_x;
                const u32x4_t av = *reinterpret_cast<const u32x4_t*>(p.gg_h + rx * 2 * p.N + nc);
                const u32x4_t gv = *reinterpret_cast<const u32x4_t*>(p.gg_h + rx * 2 * p.N + p.N + nc);
                u32x4_t oa, og;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float ra[2], rg[2];
#pragma unroll
                    for (int hh = 0; hh < 2; ++hh) {
                        const float d = __builtin_bit_cast(float, hh ? (o[e] & 0xffff0000u) : (o[e] << 16));
                        const float a_ = __builtin_bit_cast(float, hh ? (av[e] & 0xffff0000u) : (av[e] << 16));
                        const float g_ = __builtin_bit_cast(float, hh ? (gv[e] & 0xffff0000u) : (gv[e] << 16));
                        float P, ex;
                        gelu_parts(g_, P, ex);
                        ra[hh] = d * g_ * P;
                        rg[hh] = d * a_ * fmaf(g_ * 0.3989422804014327f, ex, P);
                    }
                    oa[e] = pack_bf2(ra[0], ra[1]);
                    og[e] = pack_bf2(rg[0], rg[1]);
                }
                *reinterpret_cast<u32x4_t*>(dst) = oa;
                *reinterpret_cast<u32x4_t*>(dst + p.N) = og;
                dst += dstep;
                rows_left -= kRowsPerIt;
                continue;
            }
            if (rsrc) {
                const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(rsrc);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float o0 = __builtin_bit_cast(float, o[e] << 16), o1 = __builtin_bit_cast(float, o[e] & 0xffff0000u);
                    const float r0 = __builtin_bit_cast(float, rr[e] << 16), r1 = __builtin_bit_cast(float, rr[e] & 0xffff0000u);
                    o[e] = p.mul_r ? pack_bf2(o0 * r0, o1 * r1) : pack_bf2(o0 + r0, o1 + r1);
                }
                rsrc += rstep;
            }
            *reinterpret_cast<u32x4_t*>(dst) = o;
            dst += dstep;
            rows_left -= kRowsPerIt;
        }
        return;
    }
#pragma unroll 4
    for (int it = 0; it < kIts; ++it) {
        const int row = it * kRowsPerIt + trow;
        const int r = m0 + row;
        if (r >= p.M || row >= vrows || row >= BM) break;
        if (jsel >= 0 && ((row >> 4) & 3) != jsel) continue;
        u32x4_t o = *reinterpret_cast<const u32x4_t*>(smem + row * kCRow + chunk * 16);
        long ro = r;                          // output (and residual) row
        if (p.Hp > 0) {
            const int img = img0 + (row >= b1) + (row >= b2);
            const int rem = r - img * rpi;
            const int y = (int)(((float)rem + 0.5f) * p.inv_wp), x = rem - y * p.Wp;   // exact: see header note
            const bool halo = (y == 0) | (y == p.Hp - 1) | (x == 0) | (x == p.Wp - 1);
            if (p.d2s) {
                // depth-to-space in the epilogue (stride-2 conv dgrad): this plane's pixel (y, x) is pixel (2y + py, 2x + px) of the
                // full-resolution tensor; the plane's halo rows have no place there (its halo is zero already and stays so)
                if (halo) continue;
                const int pl = p.nphase ? bz : p.d2s - 1, wf = 2 * p.Wp - 2;
                ro = (long)img * (2 * p.Hp - 2) * wf + (long)(2 * y - 1 + (pl >> 1)) * wf + (2 * x - 1 + (pl & 1));
            }
            if (halo) o = u32x4_t{0u, 0u, 0u, 0u};
            else if (p.R && nc + 8 <= p.N) {
                const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(p.R + (long)bz * p.strideC + ro * p.ldr + nc);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    o[e] = pack_bf2(__builtin_bit_cast(float, o[e] << 16) + __builtin_bit_cast(float, rr[e] << 16),
                                    __builtin_bit_cast(float, o[e] & 0xffff0000u) + __builtin_bit_cast(float, rr[e] & 0xffff0000u));
            }
        } else if (p.R && nc + 8 <= p.N) {
            const u32x4_t rr = *reinterpret_cast<const u32x4_t*>(p.R + (long)bz * p.strideC + (long)r * p.ldr + nc);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float o0 = __builtin_bit_cast(float, o[e] << 16), o1 = __builtin_bit_cast(float, o[e] & 0xffff0000u);
                const float r0 = __builtin_bit_cast(float, rr[e] << 16), r1 = __builtin_bit_cast(float, rr[e] & 0xffff0000u);
                o[e] = p.mul_r ? pack_bf2(o0 * r0, o1 * r1) : pack_bf2(o0 + r0, o1 + r1);
            }
        }
        bf16_t* dst = C + ro * p.ldc + nc;
        if (nc + 8 <= p.N) {
            *reinterpret_cast<u32x4_t*>(dst) = o;
        } else {        // ragged N tail (N % 8 != 0 never happens for channel counts; kept for safety)
            for (int e = 0; e < 8 && nc + e < p.N; ++e) {
                const uint32_t wv = o[e >> 1];
                float v = (e & 1) ? __builtin_bit_cast(float, wv & 0xffff0000u) : __builtin_bit_cast(float, wv << 16);
                if (p.R) v += bf2f(p.R[(long)bz * p.strideC + ro * p.ldr + nc + e]);
                dst[e] = f2bf(v);
            }
        }
    }
}

}  // namespace

// persistent grid of gemm_nt_c3p (one block per CU, a multiple of 8): gemm_nt.hip owns the setting (siss_gemm_nt_set_c3p_blocks)
int nt_c3p_blocks();
