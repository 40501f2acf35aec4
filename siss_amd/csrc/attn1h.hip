// Fused SINGLE-HEAD spatial self-attention -- diffusers `Attention` with attention_head_dim = None (one head of D = C channels): the
// six attention sites of the CelebA-HQ UNet (D = 512; S = 256 keys at 16 x 16, 64 at 8 x 8), reached from
// losses/ddpm_deletion_loss.py:24 and differentiated twice at delete_celeb.py:691,:702 (here: two cotangent sets against one saved
// forward).  QK^T -> softmax -> .V in ONE kernel, the backward in two (dQ; dK + dV), on bf16 MFMA with LDS-staged tiles.
//
// What it replaces (round 5): per site and step ~26 launches of the materialised form -- three projections' worth of per-sample
// batched GEMMs at 59-152 TF/s, a V / K transpose, row softmax forward / backward over S x S matrices in HBM, four f32 dK / dV
// products + two casts, pad <-> compact copies -- 1.2 ms of the CelebA-HQ step for 0.14 % of its flops.  The work is tiny
// (2.1 GFLOP per site forward at B = 16): these kernels are sized by what a block must STREAM (K and V: 2 S D bytes each), not by MFMA
// time, so the design keeps every operand that only one wave needs out of LDS and streams the shared ones once per product.
//
// Shapes: no online softmax is needed -- all S <= 256 scores of a query stay in registers.  Everything is computed
// TRANSPOSED (as in flash_attn.hip) so that the probabilities feed the second product straight from the accumulator registers:
//   v_mfma_f32_16x16x32_bf16(X, Y): out[x = (lane >> 4) * 4 + r][y = lane & 15], both operands "row, 8 consecutive k per lane".
//   forward / dQ (a block owns 64 queries of one sample, a wave 16):
//       S^T[key][q]  = mfma(K rows, Q rows)                 lane: ONE query (row statistics are per-lane scalars + two xor-shuffles)
//       O^T[d][q]   += mfma(V^T (transposed LDS read), P)   P = the S^T registers, packed; k-slot order of a 32-key step is
//                                                           [sub-tile 2j keys 4g..4g+3 | sub-tile 2j+1 keys 4g..4g+3], and the
//                                                           V^T fragment is fetched in that order (ds_read_b64_tr_b16)
//   dK / dV (a block owns 64 keys, a wave 16): the mirror image with S[q][key] (lane: one key).
// Operand traffic: the 16 rows a wave owns (its queries' Q / dO / O rows, its keys' K / V rows) are loaded from global memory
// straight into MFMA fragment registers (16 B per lane) and never touch LDS; the tensor every wave needs (K, V / Q, dO) streams
// through two 64-row LDS tiles by LDS-DMA (global_load_lds_dwordx4: one 1-KiB row per wave-instruction at D = 512), the next tile
// in flight under the current tile's products, ONE barrier per tile.  Tile image: [64 rows][D] with the 16-B chunk index XOR
// ((row & 7) << 1) -- conflict-free for the row-major ds_read_b128 fragment reads and for the transposed reads (flash_attn.hip swz()).
// The DMA realises the swizzle on the SOURCE side: LDS slot (row, c) is filled from global chunk c ^ swz(row).
//
// Layouts: q / k / v are column windows of the fused projection's compact token rows [B * S][ld]; o and dO live in the padded NHWC
// activation layout of the engine (row = pixel incl. the one-pixel halo: RowMap) so that to_out runs as a 1x1 convolution with the
// residual in its epilogue and no pad <-> compact copy exists; dq / dk / dv are column windows of ONE [nb * S][ldd] cotangent
// buffer (one wgrad job, one three-panel dgrad).  LSE (base 2) is saved by the forward; delta = rowsum(dO o O) is formed by the dQ
// kernel from the rows it loads anyway and read by the dK / dV kernel, which runs after it.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kThreadsA = 256;
constexpr int kPipe = 8;                 // MFMAs per block of the software pipeline below

// compile-time loop: the tile index selects the phase (which product, which register arrays), so every iteration is its own code
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ s16x4_t tr_read(unsigned lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)lds_byte_addr);
}

template <int D> struct AT {
    static constexpr int RS = 2 * D;                 // LDS row stride in bytes (256 / 512 / 1024)
    static constexpr int CH = D / 8;                 // 16-B chunks per row
    static constexpr int KS = D / 32;                // 32-deep k-steps over the head dimension
    static constexpr int DT = D / 16;                // 16-wide d tiles
    static constexpr int TILE = 64 * RS;             // bytes per staged 64-row tile
    static constexpr int PIECES = TILE / 1024 / 4;   // LDS-DMA wave-instructions (1 KiB each) per wave and tile
    __device__ static __forceinline__ int swz(int row) { return (row & 7) << 1; }
    __device__ static __forceinline__ int off(int row, int chunk) { return row * RS + ((chunk ^ swz(row)) << 4); }
};

// N MFMAs whose LDS fragment (ld(i): one ds_read_b128 or two transposed reads) is fetched kPipe MFMAs AHEAD of its use: block b + 1's
// reads are issued one behind each MFMA of block b, into the other half of a double fragment buffer, pinned by a scheduling fence
// per pair.  A wave is alone on its SIMD here (one 4-wave block per CU): left to the compiler every MFMA waited for its own read
// (ds_read; s_waitcnt lgkmcnt(0); v_mfma -- an LDS round trip of ~120 cycles per 16-cycle MFMA: measured 46 us per backward kernel
// at S = 256, D = 512 where the MFMAs are 8 us).  i is a compile-time index (std::integral_constant).
template <int N, typename Ld, typename Mm>
__device__ __forceinline__ void pipe(Ld&& ld, Mm&& mm) {
    constexpr int G = kPipe;
    static_assert(N % G == 0, "whole blocks");
    bf16x8_t f[2][G];
    static_for<0, G>([&](auto ic) { f[0][decltype(ic)::value] = ld(ic); });
    static_for<0, N / G>([&](auto bc) {
        constexpr int b = decltype(bc)::value;
        static_for<0, G>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            mm(std::integral_constant<int, b * G + i>{}, f[b & 1][i]);
            if constexpr (b + 1 < N / G) f[(b + 1) & 1][i] = ld(std::integral_constant<int, (b + 1) * G + i>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    });
}

// token s of image n -> row of the tensor: compact (wp == 0: n * S + s) or padded NHWC with a one-pixel halo
struct RowMap { long rpi; int W, wp; };
__device__ __forceinline__ long tok_row(const RowMap& m, int n, int s, int S) {
    if (m.wp == 0) return (long)n * S + s;
    const int y = s / m.W, x = s - y * m.W;
    return (long)n * m.rpi + (long)(y + 1) * m.wp + x + 1;
}

// Per-lane state of the tile traffic, computed once per kernel (the loops are fully unrolled: whatever is recomputed per use the
// compiler keeps live across the whole kernel -- the first form of this file formed 64-bit source addresses per DMA piece and ended
// with 128 address registers spilled to scratch).
//   DMA: a 1-KiB piece holds 64 / CH whole rows (one at D = 512); its source is a wave-UNIFORM row base (SGPR pair, scalar
//   arithmetic) + a per-lane byte offset that only depends on the piece's parity: `src[i & 1]`.
//   Fragment reads: the XOR swizzle touches the low four bits of the chunk index only, so a read address is one of a few per-lane
//   bases (4 by ks & 3 for the row-major reads, 8 by dt & 7 for the transposed ones, per tile buffer) + a compile-time constant that
//   goes into the instruction's 16-bit offset field.
template <int D> struct Lanes {
    unsigned src[2];
    unsigned rm[2][4];
    unsigned tr[2][8];
    __device__ __forceinline__ void init(unsigned lb, long ld, int wave, int lane) {
        using A = AT<D>;
        constexpr int RPP = 64 / A::CH;                        // rows per DMA piece
        const int lr = lane / A::CH, pc = lane % A::CH;
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            const int row7 = ((par * 4 + wave) * RPP + lr) & 7;
            src[par] = (unsigned)((lr * (int)ld + ((pc ^ (row7 << 1)) << 3)) * 2);
        }
        const int g = lane >> 4, row = lane & 15, q = (lane >> 2) & 3, pp = lane & 3, rr = 4 * g + q;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int m = 0; m < 4; ++m) rm[b][m] = lb + b * A::TILE + row * A::RS + (((m * 4 + g) ^ A::swz(row)) << 4);
#pragma unroll
            for (int m = 0; m < 8; ++m) tr[b][m] = lb + b * A::TILE + rr * A::RS + (((m * 2 + (pp >> 1)) ^ A::swz(rr)) << 4) + 8 * (pp & 1);
        }
    }
};

// Stage tokens [s0, s0 + 64) of image n (columns [0, D) of `base`, row stride ld elements) into tile buffer b.  Padded layouts:
// the 64 / CH tokens of a piece are consecutive pixels of one image row (W % 4 == 0: checked by the launcher).
template <int D>
__device__ __forceinline__ void stage(const Lanes<D>& L, const bf16_t* __restrict__ base, long ld, const RowMap& m, int n, int s0, int S,
                                      unsigned lb, int b, int wave) {
    using A = AT<D>;
    constexpr int RPP = 64 / A::CH;
#pragma unroll
    for (int i = 0; i < A::PIECES; ++i) {
        const int piece = i * 4 + wave;
        glds16_saddr(L.src[i & 1], base + tok_row(m, n, s0 + piece * RPP, S) * ld, lb + b * A::TILE + piece * 1024);
    }
}

typedef __attribute__((address_space(3))) const bf16x8_t* lds_b128_t;
// row-major fragment of tile buffer b: rows sub * 16 + (lane & 15), k = 32 ks + 8 (lane >> 4) .. + 7
template <int D>
__device__ __forceinline__ bf16x8_t frag_rm(const Lanes<D>& L, int b, int sub, int ks) {
    return *(lds_b128_t)(uintptr_t)(L.rm[b][ks & 3] + sub * 16 * AT<D>::RS + (ks >> 2) * 256);
}
// transposed fragment of k-step j (0 / 1 within the 64-row tile), d tile dt: lane (column dt * 16 + (lane & 15), g = lane >> 4) gets
// [rows 32 j + 4 g + 0..3 | rows 32 j + 16 + 4 g + 0..3]
template <int D>
__device__ __forceinline__ bf16x8_t frag_tr(const Lanes<D>& L, int b, int j, int dt) {
    using A = AT<D>;
    const unsigned a = L.tr[b][dt & 7] + 32 * j * A::RS + (dt >> 3) * 256;
    const s16x4_t a0 = tr_read(a), a1 = tr_read(a + 16 * A::RS);
    return bf16x8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
}
__device__ __forceinline__ bf16x8_t pack_pair(const f32x4_t& a, const f32x4_t& b) {
    return __builtin_bit_cast(bf16x8_t, u32x4_t{pack_bf2(a[0], a[1]), pack_bf2(a[2], a[3]), pack_bf2(b[0], b[1]), pack_bf2(b[2], b[3])});
}
__device__ __forceinline__ float group_max(float v) {      // over the four 16-lane groups (same lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}
__device__ __forceinline__ f32x4_t mfma(const bf16x8_t& x, const bf16x8_t& y, const f32x4_t& acc) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, acc, 0, 0, 0);
}
// 1-D grid -> (64-row tile x, image y) such that the NS tiles of an image -- and, in the backward, the cotangent images that share a
// forward image (y, y + B, ...) where the batch allows -- run on ONE XCD (blocks b and b + 8 share an XCD and its 4-MiB L2): every
// block of an image streams the same K / V (Q / dO) rows, 0.4-0.8 MB per image at D = 512.  Without it the 16 blocks of an XCD
// touch 16 different images and the streams come from the Infinity Cache at about half the per-CU rate (measured: 38 us per
// backward kernel at S = 256 where the streaming of 0.77 MB per block accounts for 12 us from L2).
__device__ __forceinline__ void block_to_tile(int NS, int ny, int& x, int& y) {
    const int lin = blockIdx.x;
    if ((ny & 7) == 0) {
        const int xcd = lin & 7, k = lin >> 3;
        x = k % NS;
        y = xcd + 8 * (k / NS);
    } else {
        x = lin % NS;
        y = lin / NS;
    }
}
// all of this wave's LDS-DMA pieces (and private loads) have landed, and -- behind the barrier -- everybody's
__device__ __forceinline__ void tile_ready() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// acc[dt][r] = element (row = this lane's token, column dt * 16 + 4 g + r): 8-B stores
template <int D>
__device__ __forceinline__ void store_rows(bf16_t* __restrict__ dst, const f32x4_t (&acc)[AT<D>::DT], float mul, int lane) {
    bf16_t* p = dst + 4 * (lane >> 4);
#pragma unroll
    for (int dt = 0; dt < AT<D>::DT; ++dt)
        *reinterpret_cast<u32x2_t*>(p + dt * 16) = u32x2_t{pack_bf2(acc[dt][0] * mul, acc[dt][1] * mul), pack_bf2(acc[dt][2] * mul, acc[dt][3] * mul)};
}

// ---------------------------------------------------------------------------------------------------------------- forward
// grid (S / 64, B).  o[row(n, q)][0:D] = softmax_k(scale q . k) v;  lse[n * S + q] = log2 sum_k exp2(c q . k), c = scale log2 e
template <int D, int NS>
__global__ __launch_bounds__(kThreadsA) void attn1h_fwd_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                const bf16_t* __restrict__ v, long ld, bf16_t* __restrict__ o, long ldo,
                                                                RowMap om, float* __restrict__ lse, float c, int B) {
    using A = AT<D>;
    constexpr int S = 64 * NS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, n;
    block_to_tile(NS, B, bx, n);
    const int s0 = bx * 64;
    const RowMap cm{0, 0, 0};
    const unsigned lb = lds_addr(smem);
    Lanes<D> L;
    L.init(lb, ld, wave, lane);
    stage<D>(L, k, ld, cm, n, 0, S, lb, 0, wave);
    const int qs = s0 + wave * 16 + (lane & 15);
    bf16x8_t qf[A::KS];
    {
        const bf16_t* qrow = q + ((long)n * S + qs) * ld + (lane >> 4) * 8;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 32);
    }
    f32x4_t sT[NS * 4];
#pragma unroll
    for (int i = 0; i < NS * 4; ++i) sT[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t oT[A::DT];
#pragma unroll
    for (int i = 0; i < A::DT; ++i) oT[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    bf16x8_t pf[NS * 2];
    static_for<0, 2 * NS>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        tile_ready();
        if constexpr (t + 1 < 2 * NS)
            stage<D>(L, t + 1 < NS ? k : v, ld, cm, n, ((t + 1) % NS) * 64, S, lb, (t + 1) & 1, wave);
        if constexpr (t < NS) {
            pipe<4 * A::KS>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_rm<D>(L, t & 1, i / A::KS, i % A::KS); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                sT[t * 4 + i / A::KS] = mfma(f, qf[i % A::KS], sT[t * 4 + i / A::KS]);
                            });
            if constexpr (t == NS - 1) {
                float mx = -INFINITY;
#pragma unroll
                for (int i = 0; i < NS * 4; ++i) mx = fmaxf(fmaxf(fmaxf(sT[i][0], sT[i][1]), fmaxf(sT[i][2], sT[i][3])), mx);
                mx = group_max(mx);
                const float mc = mx * c;
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < NS * 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sT[i][r] = __builtin_amdgcn_exp2f(sT[i][r] * c - mc); sum += sT[i][r]; }
                sum = group_sum(sum);
                const float inv = 1.f / sum;
                if (lane < 16) lse[(long)n * S + qs] = mc + __builtin_amdgcn_logf(sum);      // v_log_f32 = log2
#pragma unroll
                for (int i = 0; i < NS * 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sT[i][r] *= inv;
#pragma unroll
                for (int j = 0; j < NS * 2; ++j) pf[j] = pack_pair(sT[2 * j], sT[2 * j + 1]);
            }
        } else {
            constexpr int c2 = t - NS;
            pipe<2 * A::DT>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_tr<D>(L, t & 1, i / A::DT, i % A::DT); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                oT[i % A::DT] = mfma(f, pf[c2 * 2 + i / A::DT], oT[i % A::DT]);
                            });
        }
    });
    store_rows<D>(o + tok_row(om, n, qs, S) * ldo, oT, 1.f, lane);
}

// ---------------------------------------------------------------------------------------------------------------- backward: dQ
// grid (S / 64, nb): cotangent sample z against forward sample z % B.  Also writes delta[z * S + q] = sum_d dO[q][d] O[q][d].
template <int D, int NS>
__global__ __launch_bounds__(kThreadsA) void attn1h_bwd_dq_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                   const bf16_t* __restrict__ v, long ld, const bf16_t* __restrict__ o,
                                                                   long ldo, const bf16_t* __restrict__ dO, long lddo, RowMap pm,
                                                                   const float* __restrict__ lse, float* __restrict__ delta,
                                                                   bf16_t* __restrict__ dq, long ldd, int B, int nb, float c, float scale) {
    using A = AT<D>;
    constexpr int S = 64 * NS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, z;
    block_to_tile(NS, nb, bx, z);
    const int n = z % B, s0 = bx * 64;
    const RowMap cm{0, 0, 0};
    const unsigned lb = lds_addr(smem);
    Lanes<D> L;
    L.init(lb, ld, wave, lane);
    stage<D>(L, k, ld, cm, n, 0, S, lb, 0, wave);
    const int qs = s0 + wave * 16 + (lane & 15);
    bf16x8_t qf[A::KS], dof[A::KS];
    const int g8 = (lane >> 4) * 8;
    {
        const bf16_t* qrow = q + ((long)n * S + qs) * ld + g8;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qrow + ks * 32);
    }
    float dl = 0.f;
    const float l2 = lse[(long)n * S + qs];
    f32x4_t sT[NS * 4];
#pragma unroll
    for (int i = 0; i < NS * 4; ++i) sT[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    f32x4_t dqT[A::DT];
    bf16x8_t dsf[NS * 2];
    static_for<0, 3 * NS>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        tile_ready();
        if constexpr (t + 1 < 3 * NS) {
            constexpr int ph = (t + 1) / NS;
            stage<D>(L, ph == 1 ? v : k, ld, cm, n, ((t + 1) % NS) * 64, S, lb, (t + 1) & 1, wave);
        }
        constexpr int c2 = t % NS;
        if constexpr (t < NS) {                       // S^T = K Q^T
            pipe<4 * A::KS>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_rm<D>(L, t & 1, i / A::KS, i % A::KS); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                sT[t * 4 + i / A::KS] = mfma(f, qf[i % A::KS], sT[t * 4 + i / A::KS]);
                            });
            if constexpr (t == NS - 1) {
#pragma unroll
                for (int i = 0; i < NS * 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) sT[i][r] = __builtin_amdgcn_exp2f(sT[i][r] * c - l2);
            }
        } else if constexpr (t < 2 * NS) {            // dP^T = V dO^T ;  dS^T = scale P o (dP^T - delta)
            if constexpr (t == NS) {
                // this wave's dO rows (the query fragments are dead now) and, from them and the O rows, delta = rowsum(dO o O)
                const bf16_t* drow = dO + tok_row(pm, z, qs, S) * lddo + g8;
                const bf16_t* orow = o + tok_row(pm, n, qs, S) * ldo + g8;
#pragma unroll
                for (int ks = 0; ks < A::KS; ++ks) {
                    dof[ks] = *reinterpret_cast<const bf16x8_t*>(drow + ks * 32);
                    const bf16x8_t of = *reinterpret_cast<const bf16x8_t*>(orow + ks * 32);
#pragma unroll
                    for (int e = 0; e < 8; ++e) dl += bf2f((bf16_t)dof[ks][e]) * bf2f((bf16_t)of[e]);
                }
                dl = group_sum(dl);
                if (lane < 16) delta[(long)z * S + qs] = dl;
            }
            f32x4_t dp[4];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) dp[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            pipe<4 * A::KS>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_rm<D>(L, t & 1, i / A::KS, i % A::KS); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                dp[i / A::KS] = mfma(f, dof[i % A::KS], dp[i / A::KS]);
                            });
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) sT[c2 * 4 + sub][r] *= (dp[sub][r] - dl) * scale;
#pragma unroll
            for (int j = 0; j < 2; ++j) dsf[c2 * 2 + j] = pack_pair(sT[c2 * 4 + 2 * j], sT[c2 * 4 + 2 * j + 1]);
        } else {                            // dQ^T += K^T dS^T
            if constexpr (t == 2 * NS) {
#pragma unroll
                for (int i = 0; i < A::DT; ++i) dqT[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            pipe<2 * A::DT>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_tr<D>(L, t & 1, i / A::DT, i % A::DT); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                dqT[i % A::DT] = mfma(f, dsf[c2 * 2 + i / A::DT], dqT[i % A::DT]);
                            });
        }
    });
    store_rows<D>(dq + ((long)z * S + qs) * ldd, dqT, 1.f, lane);
}

// ---------------------------------------------------------------------------------------------------------------- backward: dK, dV
// grid (S / 64, nb): a block owns 64 keys of cotangent sample z.
template <int D, int NS>
__global__ __launch_bounds__(kThreadsA) void attn1h_bwd_dkv_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k,
                                                                    const bf16_t* __restrict__ v, long ld, const bf16_t* __restrict__ dO,
                                                                    long lddo, RowMap pm, const float* __restrict__ lse,
                                                                    const float* __restrict__ delta, bf16_t* __restrict__ dk,
                                                                    bf16_t* __restrict__ dv, long ldd, int B, int nb, float c, float scale) {
    using A = AT<D>;
    constexpr int S = 64 * NS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), g = lane >> 4;
    int bx, z;
    block_to_tile(NS, nb, bx, z);
    const int n = z % B, s0 = bx * 64;
    const RowMap cm{0, 0, 0};
    const unsigned lb = lds_addr(smem);
    Lanes<D> L, Ld;                      // (Ld: the DMA offsets for dO's row stride; its read addresses are L's and fold away)
    L.init(lb, ld, wave, lane);
    Ld.init(lb, lddo, wave, lane);
    stage<D>(L, q, ld, cm, n, 0, S, lb, 0, wave);
    const int key = s0 + wave * 16 + (lane & 15);
    bf16x8_t kv[A::KS];                   // this wave's K rows (first phase), then its V rows (second phase)
    {
        const bf16_t* krow = k + ((long)n * S + key) * ld + g * 8;
#pragma unroll
        for (int ks = 0; ks < A::KS; ++ks) kv[ks] = *reinterpret_cast<const bf16x8_t*>(krow + ks * 32);
    }
    const float* lrow = lse + (long)n * S + 4 * g;
    const float* drow = delta + (long)z * S + 4 * g;
    bf16x8_t pf[NS * 2], dsf[NS * 2];
    f32x4_t acc[A::DT];                  // dV^T, then dK^T
    static_for<0, 3 * NS>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        tile_ready();
        if constexpr (t + 1 < 3 * NS) {
            constexpr int ph = (t + 1) / NS, s1 = ((t + 1) % NS) * 64;
            if constexpr (ph == 1) stage<D>(Ld, dO, lddo, pm, z, s1, S, lb, (t + 1) & 1, wave);
            else stage<D>(L, q, ld, cm, n, s1, S, lb, (t + 1) & 1, wave);
        }
        constexpr int c2 = t % NS;
        if constexpr (t < NS) {                       // S = Q K^T;  P = exp2(c S - lse[q])
            f32x4_t p[4], l4[4];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                p[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                l4[sub] = *reinterpret_cast<const f32x4_t*>(lrow + c2 * 64 + sub * 16);
            }
            pipe<4 * A::KS>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_rm<D>(L, t & 1, i / A::KS, i % A::KS); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                p[i / A::KS] = mfma(f, kv[i % A::KS], p[i / A::KS]);
                            });
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) p[sub][r] = __builtin_amdgcn_exp2f(p[sub][r] * c - l4[sub][r]);
            pf[c2 * 2] = pack_pair(p[0], p[1]);
            pf[c2 * 2 + 1] = pack_pair(p[2], p[3]);
        } else if constexpr (t < 2 * NS) {            // dV^T += dO^T P;  dP = dO V^T;  dS = scale P o (dP - delta[q])
            if constexpr (t == NS) {
                const bf16_t* vrow = v + ((long)n * S + key) * ld + g * 8;
#pragma unroll
                for (int ks = 0; ks < A::KS; ++ks) kv[ks] = *reinterpret_cast<const bf16x8_t*>(vrow + ks * 32);
#pragma unroll
                for (int i = 0; i < A::DT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
            f32x4_t ds[4], d4[4];
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                ds[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
                d4[sub] = *reinterpret_cast<const f32x4_t*>(drow + c2 * 64 + sub * 16);
            }
            pipe<2 * A::DT>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_tr<D>(L, t & 1, i / A::DT, i % A::DT); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                acc[i % A::DT] = mfma(f, pf[c2 * 2 + i / A::DT], acc[i % A::DT]);
                            });
            pipe<4 * A::KS>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_rm<D>(L, t & 1, i / A::KS, i % A::KS); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                ds[i / A::KS] = mfma(f, kv[i % A::KS], ds[i / A::KS]);
                            });
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const bf16x8_t pp = pf[c2 * 2 + (sub >> 1)];
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[sub][r] = bf2f((bf16_t)pp[(sub & 1) * 4 + r]) * (ds[sub][r] - d4[sub][r]) * scale;
            }
            dsf[c2 * 2] = pack_pair(ds[0], ds[1]);
            dsf[c2 * 2 + 1] = pack_pair(ds[2], ds[3]);
            if constexpr (t == 2 * NS - 1) {
                store_rows<D>(dv + ((long)z * S + key) * ldd, acc, 1.f, lane);
#pragma unroll
                for (int i = 0; i < A::DT; ++i) acc[i] = f32x4_t{0.f, 0.f, 0.f, 0.f};
            }
        } else {                            // dK^T += Q^T dS
            pipe<2 * A::DT>([&](auto ic) { constexpr int i = decltype(ic)::value; return frag_tr<D>(L, t & 1, i / A::DT, i % A::DT); },
                            [&](auto ic, const bf16x8_t& f) {
                                constexpr int i = decltype(ic)::value;
                                acc[i % A::DT] = mfma(f, dsf[c2 * 2 + i / A::DT], acc[i % A::DT]);
                            });
        }
    });
    store_rows<D>(dk + ((long)z * S + key) * ldd, acc, 1.f, lane);
}

unsigned char g_smem_done[3][4][3][kMaxDevices];      // [kernel][NS - 1][D index][device]

template <typename K>
int prep(K kernel, int bytes, unsigned char (&done)[kMaxDevices]) {
    return siss_ensure_smem(reinterpret_cast<const void*>(kernel), bytes, done);
}

}  // namespace

#define ATTN1H_DISPATCH(D, NS, ...)                                                    \
    switch ((D) * 8 + (NS)) {                                                          \
        case 128 * 8 + 1: { constexpr int kD = 128, kNS = 1, kDi = 0; __VA_ARGS__; } break;   \
        case 128 * 8 + 2: { constexpr int kD = 128, kNS = 2, kDi = 0; __VA_ARGS__; } break;   \
        case 128 * 8 + 4: { constexpr int kD = 128, kNS = 4, kDi = 0; __VA_ARGS__; } break;   \
        case 256 * 8 + 1: { constexpr int kD = 256, kNS = 1, kDi = 1; __VA_ARGS__; } break;   \
        case 256 * 8 + 2: { constexpr int kD = 256, kNS = 2, kDi = 1; __VA_ARGS__; } break;   \
        case 256 * 8 + 4: { constexpr int kD = 256, kNS = 4, kDi = 1; __VA_ARGS__; } break;   \
        case 512 * 8 + 1: { constexpr int kD = 512, kNS = 1, kDi = 2; __VA_ARGS__; } break;   \
        case 512 * 8 + 2: { constexpr int kD = 512, kNS = 2, kDi = 2; __VA_ARGS__; } break;   \
        case 512 * 8 + 4: { constexpr int kD = 512, kNS = 4, kDi = 2; __VA_ARGS__; } break;   \
        default: return SISS_ERR_ARG;                                                  \
    }

extern "C" {

// 1 when siss_attn1h_fwd / _bwd cover the shape: one head of D in {128, 256, 512} channels, S in {64, 128, 256} tokens
int siss_attn1h_takes(int S, int D) {
    return (D == 128 || D == 256 || D == 512) && (S == 64 || S == 128 || S == 256);     // (padded rows: W % 4 == 0 as well)
}

// o = softmax(scale q k^T) v for B images of S tokens, ONE head of D channels.  q / k / v: compact token rows [B * S][ld] bf16 (column
// windows of the fused projection: pass the three column-offset pointers); o: row stride ldo, in the padded NHWC activation layout
// when W > 0 (image W pixels wide, one-pixel halo: token s -> pixel (s / W, s % W); halo rows are not touched) or compact [B * S]
// rows when W == 0; lse [B * S] f32 (base-2 log-sum-exp of the scaled scores, for the backward).
int siss_attn1h_fwd(const void* q, const void* k, const void* v, long ld, void* o, long ldo, int W, float* lse, int B, int S,
                    int D, float scale, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && lse && B > 0 && B <= (1 << 20) && siss_attn1h_takes(S, D));
    SISS_CHECK_ARG(ld >= D && ldo >= D && ld % 8 == 0 && ldo % 4 == 0 && (W == 0 || (W > 0 && W % 4 == 0 && S % W == 0)));
    SISS_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v) % 16 == 0 && (uintptr_t)o % 8 == 0);
    const RowMap om{W ? (long)(S / W + 2) * (W + 2) : 0, W, W ? W + 2 : 0};
    const float c = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    ATTN1H_DISPATCH(D, S / 64, {
        auto kern = attn1h_fwd_kernel<kD, kNS>;
        const int smem = 2 * AT<kD>::TILE;
        if (prep(kern, smem, g_smem_done[0][kNS - 1][kDi])) return SISS_ERR_LAUNCH;
        kern<<<dim3(kNS * B), kThreadsA, smem, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, (bf16_t*)o, ldo, om, lse, c, B);
    });
    siss_count_dispatch(SISS_K_ATTN1H_FWD);
    SISS_LAUNCH_RET();
}

// dq / dk / dv [nb * S][ldd] (three column-offset pointers into one cotangent buffer) for nb cotangent images dO against the B saved
// images (forward index = cotangent index % B: the two sets of the dual backward share q / k / v / o / lse).  o, dO: padded NHWC
// (W > 0) or compact (W == 0) rows with strides ldo / lddo; delta: scratch [nb * S] f32 (rowsum(dO o O), written by the first of
// the two launches and read by the second).
int siss_attn1h_bwd(const void* q, const void* k, const void* v, long ld, const void* o, long ldo, const void* dO, long lddo, int W,
                    const float* lse, float* delta, void* dq, void* dk, void* dv, long ldd, int nb, int B, int S, int D, float scale,
                    void* stream) {
    SISS_CHECK_ARG(q && k && v && o && dO && lse && delta && dq && dk && dv && B > 0 && nb > 0 && nb % B == 0 && nb <= (1 << 20));
    SISS_CHECK_ARG(siss_attn1h_takes(S, D) && ld >= D && ldo >= D && lddo >= D && ldd >= D);
    SISS_CHECK_ARG(ld % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && ldd % 4 == 0 && (W == 0 || (W > 0 && W % 4 == 0 && S % W == 0)));
    SISS_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)dO) % 16 == 0);
    SISS_CHECK_ARG(((uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 8 == 0);
    const RowMap pm{W ? (long)(S / W + 2) * (W + 2) : 0, W, W ? W + 2 : 0};
    const float c = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
    ATTN1H_DISPATCH(D, S / 64, {
        auto kq = attn1h_bwd_dq_kernel<kD, kNS>;
        auto kkv = attn1h_bwd_dkv_kernel<kD, kNS>;
        const int smem = 2 * AT<kD>::TILE;
        if (prep(kq, smem, g_smem_done[1][kNS - 1][kDi]) || prep(kkv, smem, g_smem_done[2][kNS - 1][kDi])) return SISS_ERR_LAUNCH;
        kq<<<dim3(kNS * nb), kThreadsA, smem, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, (const bf16_t*)o, ldo,
                                                    (const bf16_t*)dO, lddo, pm, lse, delta, (bf16_t*)dq, ldd, B, nb, c, scale);
        kkv<<<dim3(kNS * nb), kThreadsA, smem, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ld, (const bf16_t*)dO, lddo, pm,
                                                     lse, delta, (bf16_t*)dk, (bf16_t*)dv, ldd, B, nb, c, scale);
    });
    siss_count_dispatch(SISS_K_ATTN1H_BWD);
    SISS_LAUNCH_RET();
}

}  // extern "C"
