// Fused multi-head attention, forward and backward, on v_mfma_f32_32x32x16_bf16 + v_mfma_f32_16x16x32_bf16 (round 6): the SD v1.5
// transformer sites (head dims 40 / 80 / 160; 4096 / 1024 / 256 keys in self-attention, 77 in cross-attention; delete_sd.py:977-985
// -> losses/ddpm_deletion_loss.py:24, differentiated twice at delete_sd.py:1040-1060: here one dual-cotangent backward).
//
// Why a second form beside flash_attn.hip's all-16x16x32 kernels: at D = 40 those pad the contraction over the head dim to 64 (37 %
// dead MFMA work in S and dP), hold the SIMD's vector issue for 8 of every 16 matrix cycles, read one LDS fragment per MFMA (a wave
// owns 16 rows) and stage every tile global -> registers -> LDS through the VALU: 441 TF/s algorithmic in loops bound by vector issue.
// Here:
//   * S / dP run on 32x32x16 in 16-deep steps: D + the 8-wide augmented chunk (40 + 8 = 48 = three MFMAs, nothing dead), and a
//     32x32x16 MFMA blocks vector issue for 8 of 32 cycles;
//   * the output products (contraction over 32 keys / queries, head dim as the row index) run on 16x16x32 over 16-wide d tiles
//     (48 rows for D = 40, not 64): P / dS cross from the 32x32 accumulator layout to the 16x16 B-operand layout by four
//     v_permlane16_swap per 32 x 32 block (to16);
//   * a wave owns 32 rows; its own operands (Q / dO rows in the forward / dQ kernels, K / V rows in the dK / dV kernel) live in
//     registers as B fragments, the streamed tensor's fragments (A operands) are read from LDS once per 32 x 32 score block;
//   * tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4), source-side XOR swizzle; pad chunks and rows past the tensor are
//     never moved (EXEC-masked lanes leave the zeros / ones written once per block); next tile in flight, ONE barrier per tile.
// The algorithm is flash_attn.hip's: FlashAttention-2 recompute from the saved base-2 log-sum-exp, everything transposed so that P / dS
// feed the next product from the accumulator registers, the augmented contraction (three bf16 parts of -lse against ones deliver
// s - lse, two parts of -delta against ones deliver dP - delta), delta formed by the dQ kernel.
// Measured on the SD self-attention shape (B 16, 8 heads of 40, 4096 keys): forward 890 -> 580 us, backward 3.9 -> 2.85 ms; the
// backward then sits at the board's power limit (1385 of 1400 W) with the matrix pipe as its first bound (docs/experiments.md, round 6:
// removing the exponentials saves 9 %, removing the output products 35 %; a hand-interleaved software pipeline was slower).
//
//   S^T[key][q] = mfma32(K rows (LDS, row-major b128), Q frag)   lane: ONE query (lane & 31); register r: key (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
//   dQ^T[d][q] += mfma16(K^T (LDS, transposed reads), dS)        lane: query lane & 15 of its 16-query half; register r: d = 16 dt + 4 (lane >> 4) + r
// The dK / dV kernel is the mirror image (lane: one key; S[q][key] = mfma32(Q rows, K frag)).
//
// LDS tile: 64 rows x RB bytes, RB = 128 / 256 / 512 for D <= 56 / 120 / 248 (D / 8 data chunks of 16 B, the augmented chunk, zeros);
// chunk slot = chunk ^ f(row) (low three / four bits).  f is chosen so that the 32-row b128 fragment reads (lane groups {0-3, 12-15,
// 20-27}, {4-11, 16-19, 28-31}) and the transposed reads (per 32 lanes: rows R .. R + 3 and R + 16 .. R + 19, 32 B each) are
// conflict-free: RB = 128: f = (row bit 1) << 2 | (bit 4) << 1 | (bit 2);  RB >= 256: f = (row & 3) << 1 | (bit 4) << 3 | (bit 2).
#include "common.h"
#include "flash32.h"
#include <type_traits>

namespace {

// FA32_ABL (probe builds only, tools/probes/fa32_ablate.sh; wrong results): 1 no exp2, 2 no output products, 3 no softmax arithmetic
#ifndef FA32_ABL
#define FA32_ABL 0
#endif

constexpr int kT = 256;
constexpr uint32_t kOne2 = 0x3F803F80u, kOne1 = 0x00003F80u;       // bf16 (1, 1) / (1, 0)

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) const bf16x8_t* lds_b128_t;
__device__ __forceinline__ s16x4_t tr_read(unsigned a) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(uintptr_t)a);
}
__device__ __forceinline__ bf16x8_t rd128(unsigned a) { return *(lds_b128_t)(uintptr_t)a; }
__device__ __forceinline__ f32x16_t mfma32(const bf16x8_t& a, const bf16x8_t& b, const f32x16_t& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4_t mfma16(const bf16x8_t& a, const bf16x8_t& b, const f32x4_t& c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ uint32_t bf16_bits(float v) { return pack_bf2(v, 0.f) & 0xffffu; }
__device__ __forceinline__ float bf16_val(uint32_t b) { return __builtin_bit_cast(float, b << 16); }
// v -> (hi, mid, lo) bf16 with hi + mid + lo = v to ~2^-24 relative; nparts = 2 drops lo
__device__ __forceinline__ u32x4_t split_bf16(float v, int nparts) {
    const uint32_t h = bf16_bits(v);
    const float r1 = v - bf16_val(h);
    const uint32_t m = bf16_bits(r1);
    const uint32_t l = nparts > 2 ? bf16_bits(r1 - bf16_val(m)) : 0u;
    return u32x4_t{h | (m << 16), l, 0u, 0u};
}
__device__ __forceinline__ bf16x8_t as_frag(u32x4_t v) { return __builtin_bit_cast(bf16x8_t, v); }
__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
constexpr f32x4_t kZero4 = {0.f, 0.f, 0.f, 0.f};

// geometry of the staged tiles for a head dim of NCH 16-B chunks
template <int NCH> struct G {
    static constexpr int CPR = NCH + 1 <= 8 ? 8 : (NCH + 1 <= 16 ? 16 : 32);     // chunk slots per LDS row
    static_assert(NCH + 1 <= 32, "head dim <= 248");
    static constexpr int RB = CPR * 16;                  // bytes per row
    static constexpr int TILE = 64 * RB;                 // bytes per 64-row tile
    static constexpr int KS = (NCH + 2) / 2;             // 16-deep contraction steps over the head dim (+ the augmented chunk)
    static constexpr int DT = (NCH * 8 + 15) / 16;       // 16-wide d tiles of the output products
    static constexpr int RPP = 1024 / RB;                // rows per 1-KiB DMA piece
    static constexpr int PPW = 64 / RPP / 4;             // pieces per wave and tile
    static constexpr int NV = CPR == 8 ? 1 : 2;          // variants of a lane's DMA source (the pieces of a wave differ in row bit 4)
    static constexpr int NA = KS < 8 ? KS : 8;           // per-lane fragment address registers (chunks >= 16 sit 256 B further: immediates)
    static constexpr int NTR = DT < 8 ? DT : 8;
    __device__ static __forceinline__ int fsw(int r) {
        return CPR == 8 ? ((((r >> 1) & 1) << 2) | (((r >> 4) & 1) << 1) | ((r >> 2) & 1))
                        : (((r & 3) << 1) | (((r >> 4) & 1) << 3) | ((r >> 2) & 1));
    }
    // byte offset of logical chunk c of row r within a tile (the XOR touches the low bits of the chunk index only)
    __device__ static __forceinline__ int off(int r, int c) { return r * RB + ((c ^ fsw(r)) << 4); }
    __device__ static constexpr int variant(int j) { return CPR == 8 ? 0 : (CPR == 16 ? (j & 1) : ((j >> 1) & 1)); }   // of piece w + 4 j
};

// A 32 x 32 f32 tile as it leaves v_mfma_f32_32x32x16 (lane: column lane & 31, register r: row (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
// -> the two B fragments of v_mfma_f32_16x16x32 for columns 0-15 (x0) and 16-31 (x1), contraction over all 32 rows: pack the registers
// 0-7 and 8-15 pairwise, then ONE v_permlane16_swap per word pair (it exchanges the odd 16-lane rows of its first operand with the
// even rows of the second): x0 = [lanes 0-15 regs 0-7 | lanes 0-15 regs 8-15 | lanes 32-47 regs 0-7 | lanes 32-47 regs 8-15], i.e.
// the 16-lane group g of x0 holds, for column lane & 15, the rows  16 (g & 1) + 4 (g >> 1) + {0..3, 8..11}  -- frag_tr16's order.
__device__ __forceinline__ void to16(const f32x16_t& x, bf16x8_t& x0, bf16x8_t& x1) {
    u32x4_t lo, hi;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto sw = __builtin_amdgcn_permlane16_swap(pack_bf2(x[2 * i], x[2 * i + 1]), pack_bf2(x[8 + 2 * i], x[8 + 2 * i + 1]), false, false);
        lo[i] = sw[0];
        hi[i] = sw[1];
    }
    x0 = as_frag(lo);
    x1 = as_frag(hi);
}

// per-lane read addresses (bytes from the tile's start; tile bases and the 32-row half are immediates)
template <int NCH> struct Lanes32 {
    using T = G<NCH>;
    unsigned rm[T::NA];       // row-major b128: row (lane & 31), chunk 2 s + h
    unsigned tr[T::NTR];      // transposed (16-wide d tile dt): rows 16 (g & 1) + 4 (g >> 1) + (lane >> 2 & 3), g = lane >> 4; second read: + 8 rows
    __device__ __forceinline__ void init(unsigned lb, int lane) {
        const int h = lane >> 5, r32 = lane & 31;
#pragma unroll
        for (int s = 0; s < T::NA; ++s) rm[s] = lb + T::off(r32, 2 * s + h);
        const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
        const int row = 16 * (g & 1) + 4 * (g >> 1) + q4;          // (fsw ignores row bit 3: the second read shares the swizzle)
#pragma unroll
        for (int dt = 0; dt < T::NTR; ++dt) tr[dt] = lb + T::off(row, dt * 2 + (pp >> 1)) + 8 * (pp & 1);
    }
    // A fragment of mfma32: rows half * 32 + (lane & 31) of the tile at byte `tile_off`, contraction step ks
    __device__ __forceinline__ bf16x8_t rowfrag(int tile_off, int half, int ks) const {
        return rd128(rm[ks & 7] + tile_off + half * 32 * T::RB + (ks >> 3) * 256);
    }
    // A fragment of mfma16 [16 rows d = dt * 16 + (lane & 15)][32 k = the rows of the 32-row half in to16's order]
    __device__ __forceinline__ bf16x8_t trfrag(int tile_off, int half, int dt) const {
        const unsigned a = tr[dt & 7] + tile_off + half * 32 * T::RB + (dt >> 3) * 256;
        const s16x4_t a0 = tr_read(a), a1 = tr_read(a + 8 * T::RB);
        return bf16x8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
    }
};

// LDS-DMA of 64-row tiles: wave w moves the 1-KiB pieces w, w + 4, ... (RPP rows each).  A lane's source chunk and its activity depend
// on the piece only through row bit 4: NV variants, fixed for the kernel.
template <int NCH> struct Dma {
    using T = G<NCH>;
    unsigned off[T::NV];      // this lane's byte offset from the piece's first row
    int lrow;                 // its row within the piece
    bool act[T::NV];          // its chunk is a data chunk
    __device__ __forceinline__ void init(int lane, int w, long ld) {
        constexpr int LPR = T::CPR;                       // lanes per row
        lrow = lane / LPR;
        const int sl = lane % LPR;
#pragma unroll
        for (int v = 0; v < T::NV; ++v) {
            const int row = (w + (T::CPR == 16 ? 4 * v : 8 * v)) * T::RPP + lrow;     // a piece of variant v
            const int c = sl ^ T::fsw(row);
            act[v] = c < NCH;
            off[v] = (unsigned)((lrow * ld + c * 8) * 2);
        }
    }
    // rows [0, nrows) of the tile that starts at row0 (nrows = 64: no row test)
    __device__ __forceinline__ void stage(const bf16_t* __restrict__ row0, long ld, unsigned lds_tile, int w, int nrows) const {
#pragma unroll
        for (int v = 0; v < T::NV; ++v) {
            if (nrows >= 64) {
                if (act[v]) {
#pragma unroll
                    for (int j = 0; j < T::PPW; ++j)
                        if (T::variant(j) == v) glds16_saddr(off[v], row0 + (long)((w + 4 * j) * T::RPP) * ld, lds_tile + (w + 4 * j) * 1024);
                }
            } else {
#pragma unroll
                for (int j = 0; j < T::PPW; ++j)
                    if (T::variant(j) == v && act[v] && (w + 4 * j) * T::RPP + lrow < nrows)
                        glds16_saddr(off[v], row0 + (long)((w + 4 * j) * T::RPP) * ld, lds_tile + (w + 4 * j) * 1024);
            }
        }
    }
};

struct P32 {
    const bf16_t *q, *k, *v, *o, *d_o;
    bf16_t *dq, *dk, *dv;
    long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    const float* lse2;
    float* delta;
    int nBH, BHf, H, D, Sq, Sk;        // cotangent (batch, head) entries, forward entries, heads, head dim, rows
    float scale, kscale, c;            // dQ factor, dK factor, score -> base-2 logit factor (1 when q is pre-scaled)
    int nch, qchunk;                   // dK / dV kernel: query chunks per key block (1: none) and rows per chunk
    float* part;                       // ... their partial tiles [z][chunk][key block * 128 + key][dK: DT * 16 | dV: DT * 16] f32
};

// 1-D grid -> (tile x of nx, entry z): the nx blocks of an entry -- and the cotangent entries that share a forward entry -- run on ONE
// XCD (blocks b and b + 8 share an XCD and its L2): they stream the same K / V (Q / dO) rows.
__device__ __forceinline__ void block_map(int nx, int nBH, int BHf, int& x, int& z) {
    const int lin = blockIdx.x;
    if ((BHf & 7) == 0) {
        const int xcd = lin & 7, idx = lin >> 3, nsets = nBH / BHf;
        x = idx % nx;
        const int j = idx / nx;
        z = (j % nsets) * BHf + (j / nsets) * 8 + xcd;
    } else {
        x = lin % nx;
        z = lin / nx;
    }
}

template <int NCH> __device__ __forceinline__ void lds_zero(char* smem, int tid) {
#pragma unroll
    for (int i = 0; i < 4 * G<NCH>::TILE / 16 / kT; ++i) reinterpret_cast<u32x4_t*>(smem)[i * kT + tid] = u32x4_t{0u, 0u, 0u, 0u};
}
// key (32x32 layout: register r of lane half h, 32-key half `half` of the tile at key k0) >= Sk ?
__device__ __forceinline__ bool key_past(int r, int h, int k0, int half, int Sk) { return k0 + half * 32 + (r & 3) + 8 * (r >> 2) + 4 * h >= Sk; }

// =====================================================================================================================
// dQ = scale * dS K,  dS = P o (dO V^T - delta),  delta = rowsum(dO o O) (written for the dK / dV kernel)
// block: 128 queries of one cotangent (batch, head) entry (a wave: 32); grid: Sq / 128 x nBH blocks (block_map)
// =====================================================================================================================
template <int NCH, bool PRE>
__global__ __launch_bounds__(kT, NCH <= 7 ? 4 : (NCH <= 15 ? 2 : 1)) void fa32_bwd_dq_kernel(P32 a) {
    using T = G<NCH>;
    constexpr int KS = T::KS, DT = T::DT, TILE = T::TILE;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 buffers][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    int qb, z;
    block_map(a.Sq / 128, a.nBH, a.BHf, qb, z);
    const int bz = z / a.H, hh = z - bz * a.H;
    const int zf = z % a.BHf, bf = zf / a.H;
    const int qrow = qb * 128 + w * 32 + r32;
    const unsigned lb = lds_addr(smem);

    // ---- LDS: zeros everywhere, the ones of the augmented chunk in all four tiles (the DMA never touches chunks >= NCH)
    lds_zero<NCH>(smem, tid);
    __syncthreads();
    {
        const int tile = tid >> 6, row = tid & 63;
        *reinterpret_cast<u32x4_t*>(smem + tile * TILE + T::off(row, NCH)) = (tile & 1) ? u32x4_t{kOne2, 0u, 0u, 0u} : u32x4_t{kOne2, kOne1, 0u, 0u};
    }
    __syncthreads();

    const bf16_t* kg = a.k + (long)bf * a.Sk * a.ldk + hh * a.D;
    const bf16_t* vg = a.v + (long)bf * a.Sk * a.ldv + hh * a.D;
    Dma<NCH> dk_, dv_;
    dk_.init(lane, w, a.ldk);
    dv_.init(lane, w, a.ldv);
    const int NT = (a.Sk + 63) / 64;
    auto stage = [&](int t, int b) {
        const int nrows = a.Sk - t * 64;
        dk_.stage(kg + (long)t * 64 * a.ldk, a.ldk, lb + b * 2 * TILE, w, nrows);
        dv_.stage(vg + (long)t * 64 * a.ldv, a.ldv, lb + b * 2 * TILE + TILE, w, nrows);
    };
    stage(0, 0);

    // ---- this wave's 32 queries: Q and dO as B fragments (lane: query r32, chunks 2 s + h), delta from dO and O
    bf16x8_t qf[KS], dof[KS];
    float dl = 0.f;
    {
        const long col = hh * a.D + h * 8;
        const bf16_t* qp = a.q + ((long)bf * a.Sq + qrow) * a.ldq + col;
        const bf16_t* dop = a.d_o + ((long)bz * a.Sq + qrow) * a.lddo + col;
        const bf16_t* op = a.o + ((long)bf * a.Sq + qrow) * a.ldo + col;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool ok = 2 * s + h < NCH;
            qf[s] = ok ? *reinterpret_cast<const bf16x8_t*>(qp + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
            dof[s] = ok ? *reinterpret_cast<const bf16x8_t*>(dop + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
            const u32x4_t of = ok ? *reinterpret_cast<const u32x4_t*>(op + s * 16) : u32x4_t{0u, 0u, 0u, 0u};
            const u32x4_t d4 = __builtin_bit_cast(u32x4_t, dof[s]);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dl += __builtin_bit_cast(float, d4[e] << 16) * __builtin_bit_cast(float, of[e] << 16) +
                      __builtin_bit_cast(float, d4[e] & 0xffff0000u) * __builtin_bit_cast(float, of[e] & 0xffff0000u);
        }
    }
    dl += __shfl_xor(dl, 32, 64);
    if (h == 0) a.delta[(long)z * a.Sq + qrow] = dl;
    const float lse = a.lse2[(long)zf * a.Sq + qrow];
    if (h == (NCH & 1)) {                                        // the augmented chunk NCH sits in step NCH / 2, lane half NCH & 1
        qf[NCH / 2] = as_frag(split_bf16(PRE ? -lse : -lse / a.c, 3));
        dof[NCH / 2] = as_frag(split_bf16(-dl, 2));
    }
    f32x4_t dq[DT][2];                                           // dQ^T[d tile][query half]: lane = query (lane & 15), 4 d per tile
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dq[dt][0] = kZero4; dq[dt][1] = kZero4; }
    Lanes32<NCH> L;
    L.init(lb, lane);

    auto tile = [&](auto bc, int t) {
        constexpr int b = decltype(bc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < NT) stage(t + 1, b ^ 1);
        constexpr int kb = b * 2 * TILE, vb = kb + TILE;
        const bool ragged = t * 64 + 64 > a.Sk;                  // (wave-uniform) the tile holds rows past the last key
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16_t s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s = mfma32(L.rowfrag(kb, half, ks), qf[ks], s);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dp = mfma32(L.rowfrag(vb, half, ks), dof[ks], dp);
            // (the MFMAs subtracted lse and delta) dS without its factor `scale`: applied once to the finished tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (FA32_ABL == 3) { s[r] += dp[r]; continue; }
                s[r] = (FA32_ABL == 1 ? s[r] : __builtin_amdgcn_exp2f(PRE ? s[r] : s[r] * a.c)) * dp[r];
            }
            if (ragged) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[r] = key_past(r, h, t * 64, half, a.Sk) ? 0.f : s[r];
            }
            bf16x8_t ds0, ds1;
            to16(s, ds0, ds1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (FA32_ABL == 2) { dq[dt][0][0] += __builtin_bit_cast(float, (int)ds0[0] | (int)ds1[1]); continue; }
                const bf16x8_t kt = L.trfrag(kb, half, dt);
                dq[dt][0] = mfma16(kt, ds0, dq[dt][0]);
                dq[dt][1] = mfma16(kt, ds1, dq[dt][1]);
            }
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < NT) tile(std::integral_constant<int, 1>{}, t + 1);
    }
    // dQ^T[d][q]: lane (lane & 15) = query within its half, register r = column dt * 16 + 4 (lane >> 4) + r
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const int qr = qb * 128 + w * 32 + qh * 16 + (lane & 15);
        bf16_t* og = a.dq + ((long)bz * a.Sq + qr) * a.lddq + hh * a.D + 4 * (lane >> 4);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (dt * 16 + 4 * (lane >> 4) < a.D)
                *reinterpret_cast<u32x2_t*>(og + dt * 16) =
                    u32x2_t{pack_bf2(dq[dt][qh][0] * a.scale, dq[dt][qh][1] * a.scale), pack_bf2(dq[dt][qh][2] * a.scale, dq[dt][qh][3] * a.scale)};
    }
}

// =====================================================================================================================
// dV = P^T dO,  dK = kscale * dS^T Q
// block: 128 keys of one cotangent (batch, head) entry (a wave: 32) x one chunk of the queries; grid: ceil(Sk / 128) * nch x nBH
// blocks (block_map).  nch > 1 (few key blocks: cross-attention): every block leaves f32 partial tiles in `part`, summed by
// fa32_dkdv_reduce_kernel -- deterministic, no atomics.
// =====================================================================================================================
// KT: 32-key tiles per wave (1: a block owns 128 keys; 2: 256 -- the streamed Q / dO fragments, row-major and transposed, then serve two
// score blocks each: half the LDS reads and half the L2 -> LDS traffic per MFMA; D = 40 only: the accumulators double)
template <int NCH, bool PRE, int KT>
__global__ __launch_bounds__(kT, KT == 2 ? 2 : (NCH <= 7 ? 3 : (NCH <= 15 ? 2 : 1))) void fa32_bwd_dkdv_kernel(P32 a) {
    using T = G<NCH>;
    constexpr int KS = T::KS, DT = T::DT, TILE = T::TILE, BK_ = 128 * KT;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 buffers][Q tile | dO tile]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    int x_, z;
    block_map(((a.Sk + BK_ - 1) / BK_) * a.nch, a.nBH, a.BHf, x_, z);
    const int kb_ = x_ / a.nch, chunk = x_ - kb_ * a.nch;
    const int bz = z / a.H, hh = z - bz * a.H;
    const int zf = z % a.BHf, bf = zf / a.H;
    const int key0 = kb_ * BK_ + w * 32 * KT;                   // this wave's first key
    const unsigned lb = lds_addr(smem);
    lds_zero<NCH>(smem, tid);
    __syncthreads();

    const int q_lo = chunk * a.qchunk;
    const int q_hi = q_lo + a.qchunk < a.Sq ? q_lo + a.qchunk : a.Sq;
    const int NT = (q_hi - q_lo) / 64;                            // (Sq and qchunk are multiples of 64)
    const bf16_t* qg = a.q + ((long)bf * a.Sq + q_lo) * a.ldq + hh * a.D;
    const bf16_t* dog = a.d_o + ((long)bz * a.Sq + q_lo) * a.lddo + hh * a.D;
    Dma<NCH> dq_, ddo_;
    dq_.init(lane, w, a.ldq);
    ddo_.init(lane, w, a.lddo);
    auto stage = [&](int t, int b) {
        dq_.stage(qg + (long)t * 64 * a.ldq, a.ldq, lb + b * 2 * TILE, w, 64);
        ddo_.stage(dog + (long)t * 64 * a.lddo, a.lddo, lb + b * 2 * TILE + TILE, w, 64);
    };
    stage(0, 0);
    // The augmented chunk of the streamed tiles changes per row: wave 0 writes the three parts of -lse of the tile's 64 queries into
    // the Q tile, wave 1 the two parts of -delta into the dO tile (one lane per row), one tile ahead, from a value loaded two ahead.
    const float* aug_src = (w == 0 ? a.lse2 + (long)zf * a.Sq : a.delta + (long)z * a.Sq) + q_lo;
    const float amul = (w == 0 && !PRE) ? -1.f / a.c : -1.f;
    char* const aug_dst = smem + (w == 1 ? TILE : 0) + T::off(lane, NCH);
    float a_nx = 0.f;
    if (w < 2) {
        const float a0 = aug_src[lane];
        if (NT > 1) a_nx = aug_src[64 + lane];
        *reinterpret_cast<u32x4_t*>(aug_dst) = split_bf16(a0 * amul, w == 0 ? 3 : 2);
    }

    // ---- this wave's keys: K and V as B fragments (+ the ones against the augmented columns); keys past Sk: zeros, masked below
    bf16x8_t kf[KT][KS], vf[KT][KS];
    float key_mask[KT];                                          // a lane keeps ONE key per tile
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
        const int krow = key0 + kt * 32 + r32;
        const bool k_ok = krow < a.Sk;
        key_mask[kt] = k_ok ? 1.f : 0.f;
        const long col = hh * a.D + h * 8;
        const bf16_t* kp = a.k + ((long)bf * a.Sk + krow) * a.ldk + col;
        const bf16_t* vp = a.v + ((long)bf * a.Sk + krow) * a.ldv + col;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool ok = k_ok && 2 * s + h < NCH;
            kf[kt][s] = ok ? *reinterpret_cast<const bf16x8_t*>(kp + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
            vf[kt][s] = ok ? *reinterpret_cast<const bf16x8_t*>(vp + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
        }
        if (h == (NCH & 1)) {
            kf[kt][NCH / 2] = as_frag(u32x4_t{kOne2, kOne1, 0u, 0u});
            vf[kt][NCH / 2] = as_frag(u32x4_t{kOne2, 0u, 0u, 0u});
        }
    }
    const bool any_masked = kb_ * BK_ + BK_ > a.Sk;              // (block-uniform)
    f32x4_t dk[KT][DT][2], dv[KT][DT][2];                        // [key tile][d tile][key half]: lane = key (lane & 15), 4 d per tile
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) { dk[kt][dt][0] = kZero4; dk[kt][dt][1] = kZero4; dv[kt][dt][0] = kZero4; dv[kt][dt][1] = kZero4; }
    Lanes32<NCH> L;
    L.init(lb, lane);

    auto tile = [&](auto bc, int t) {
        constexpr int b = decltype(bc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < NT) {
            stage(t + 1, b ^ 1);
            if (w < 2) {
                *reinterpret_cast<u32x4_t*>(aug_dst + (b ^ 1) * 2 * TILE) = split_bf16(a_nx * amul, w == 0 ? 3 : 2);
                if (t + 2 < NT) a_nx = aug_src[(t + 2) * 64 + lane];
            }
        }
        constexpr int qb = b * 2 * TILE, dob = qb + TILE;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8_t qa[KS], doa[KS];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) { qa[ks] = L.rowfrag(qb, half, ks); doa[ks] = L.rowfrag(dob, half, ks); }
            bf16x8_t p0[KT], p1[KT], ds0[KT], ds1[KT];
#pragma unroll
            for (int kt = 0; kt < KT; ++kt) {
                f32x16_t s = zero16(), dp = zero16();
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) s = mfma32(qa[ks], kf[kt][ks], s);
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) dp = mfma32(doa[ks], vf[kt][ks], dp);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    if (FA32_ABL == 3) { dp[r] += s[r]; continue; }
                    s[r] = FA32_ABL == 1 ? s[r] : __builtin_amdgcn_exp2f(PRE ? s[r] : s[r] * a.c);
                    dp[r] *= s[r];
                }
                if (any_masked) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) { s[r] *= key_mask[kt]; dp[r] *= key_mask[kt]; }
                }
                to16(s, p0[kt], p1[kt]);
                to16(dp, ds0[kt], ds1[kt]);
            }
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (FA32_ABL == 2) { dv[0][dt][0][0] += __builtin_bit_cast(float, (int)p0[0][0] | (int)p1[0][1] | (int)ds0[0][2] | (int)ds1[0][3]); continue; }
                const bf16x8_t dot = L.trfrag(dob, half, dt);
                const bf16x8_t qt = L.trfrag(qb, half, dt);
#pragma unroll
                for (int kt = 0; kt < KT; ++kt) {
                    dv[kt][dt][0] = mfma16(dot, p0[kt], dv[kt][dt][0]);
                    dv[kt][dt][1] = mfma16(dot, p1[kt], dv[kt][dt][1]);
                    dk[kt][dt][0] = mfma16(qt, ds0[kt], dk[kt][dt][0]);
                    dk[kt][dt][1] = mfma16(qt, ds1[kt], dk[kt][dt][1]);
                }
            }
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < NT) tile(std::integral_constant<int, 1>{}, t + 1);
    }
#pragma unroll
    for (int kt = 0; kt < KT; ++kt)
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
            const int kr = key0 + kt * 32 + kh * 16 + (lane & 15);
            if (a.part) {                                        // f32 partial tiles, unscaled; every key of the block (padding included)
                float* pg = a.part + (((long)z * a.nch + chunk) * ((a.Sk + BK_ - 1) / BK_) * BK_ + kr) * (2 * DT * 16) + 4 * (lane >> 4);
#pragma unroll
                for (int dt = 0; dt < DT; ++dt) {
                    *reinterpret_cast<f32x4_t*>(pg + dt * 16) = dk[kt][dt][kh];
                    *reinterpret_cast<f32x4_t*>(pg + DT * 16 + dt * 16) = dv[kt][dt][kh];
                }
                continue;
            }
            if (kr >= a.Sk) continue;
            bf16_t* okg = a.dk + ((long)bz * a.Sk + kr) * a.lddk + hh * a.D + 4 * (lane >> 4);
            bf16_t* ovg = a.dv + ((long)bz * a.Sk + kr) * a.lddv + hh * a.D + 4 * (lane >> 4);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
                if (dt * 16 + 4 * (lane >> 4) < a.D) {
                    *reinterpret_cast<u32x2_t*>(okg + dt * 16) =
                        u32x2_t{pack_bf2(dk[kt][dt][kh][0] * a.kscale, dk[kt][dt][kh][1] * a.kscale),
                                pack_bf2(dk[kt][dt][kh][2] * a.kscale, dk[kt][dt][kh][3] * a.kscale)};
                    *reinterpret_cast<u32x2_t*>(ovg + dt * 16) =
                        u32x2_t{pack_bf2(dv[kt][dt][kh][0], dv[kt][dt][kh][1]), pack_bf2(dv[kt][dt][kh][2], dv[kt][dt][kh][3])};
                }
        }
}

// Sum of the query chunks' partial dK / dV: one thread per (z, key, 4 head-dim columns).
__global__ __launch_bounds__(256) void fa32_dkdv_reduce_kernel(P32 a, int DW, long total) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = a.D >> 2;
    const int col = (int)(i % c4) * 4;
    const long r = i / c4;
    const int key = (int)(r % a.Sk);
    const long z = r / a.Sk;
    const long bz = z / a.H;
    const int hh = (int)(z - bz * a.H);
    const long skp = (long)((a.Sk + 127) / 128) * 128;
    f32x4_t k = kZero4, v = kZero4;
    for (int c = 0; c < a.nch; ++c) {
        const float* pg = a.part + ((z * a.nch + c) * skp + key) * (2 * DW) + col;
        const f32x4_t x = *reinterpret_cast<const f32x4_t*>(pg), y = *reinterpret_cast<const f32x4_t*>(pg + DW);
#pragma unroll
        for (int e = 0; e < 4; ++e) { k[e] += x[e]; v[e] += y[e]; }
    }
    const long o = bz * a.Sk + key;
    *reinterpret_cast<u32x2_t*>(a.dk + o * a.lddk + hh * a.D + col) =
        u32x2_t{pack_bf2(k[0] * a.kscale, k[1] * a.kscale), pack_bf2(k[2] * a.kscale, k[3] * a.kscale)};
    *reinterpret_cast<u32x2_t*>(a.dv + o * a.lddv + hh * a.D + col) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}

// =====================================================================================================================
// forward: O = softmax(c' Q K^T) V, LSE2[q] = base-2 log-sum-exp of the scaled scores
// block: 128 queries of one (batch, head) entry (a wave: 32); grid: Sq / 128 x BH blocks (block_map)
// A 64-key tile: S^T of both 32-key halves (2 KS MFMAs 32x32x16), ONE running-maximum update for the 64 keys (v_max3 chains + one
// v_permlane32_swap), p = exp2(s - m), P V on 16x16x32.  The row sum rides in the product where the head dim leaves a pad column
// inside the last d tile (D % 16 == 8: V's column D holds ones, O^T[D][q] accumulates sum_k p with O's rescaling); otherwise it
// is summed on the VALU.  S lives in the 32x32 layout (lane = query lane & 31), O in the 16x16 layout (lane = query lane & 15 of
// its half): the rescale factor crosses between the two by a bpermute, but only in tiles where some row's maximum moved (a
// wave-uniform branch, rare after the first tiles).
// =====================================================================================================================
struct PF {
    const bf16_t *q, *k, *v;
    bf16_t* o;
    long ldq, ldk, ldv, ldo;
    float* lse2;
    int BH, H, D, Sq, Sk;
    float c;                           // score -> base-2 logit factor (1 when q is pre-scaled)
};
__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }
// (r0, r1) = (v of this lane, v of lane ^ 32) in some order
__device__ __forceinline__ void both_halves(float v, float& r0, float& r1) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned u2 = u;
    asm volatile("" : "+v"(u2));                                 // (see sum_lanes_mod8: keeps hipcc from folding the swap of a value with itself)
    const auto r = __builtin_amdgcn_permlane32_swap(u, u2, false, false);
    unsigned a0 = r[0], a1 = r[1];
    asm volatile("" : "+v"(a0), "+v"(a1));
    r0 = __builtin_bit_cast(float, a0);
    r1 = __builtin_bit_cast(float, a1);
}

template <int NCH, bool PRE>
__global__ __launch_bounds__(kT, NCH <= 7 ? 4 : (NCH <= 15 ? 2 : 1)) void fa32_fwd_kernel(PF a) {
    using T = G<NCH>;
    constexpr int KS = T::KS, DT = T::DT, TILE = T::TILE;
    constexpr bool LCOL = (NCH & 1) == 1;                        // a pad column at index D inside the last 16-wide d tile
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 buffers][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    int qb, z;
    block_map(a.Sq / 128, a.BH, a.BH, qb, z);
    const int bz = z / a.H, hh = z - bz * a.H;
    const int qrow = qb * 128 + w * 32 + r32;
    const unsigned lb = lds_addr(smem);
    lds_zero<NCH>(smem, tid);
    __syncthreads();
    if (LCOL && tid < 128) {                                     // V tiles: column D := 1
        const int tile = 1 + 2 * (tid >> 6), row = tid & 63;
        *reinterpret_cast<u32x4_t*>(smem + tile * TILE + T::off(row, NCH)) = u32x4_t{kOne1, 0u, 0u, 0u};
    }
    __syncthreads();
    const bf16_t* kg = a.k + (long)bz * a.Sk * a.ldk + hh * a.D;
    const bf16_t* vg = a.v + (long)bz * a.Sk * a.ldv + hh * a.D;
    Dma<NCH> dk_, dv_;
    dk_.init(lane, w, a.ldk);
    dv_.init(lane, w, a.ldv);
    const int NT = (a.Sk + 63) / 64;
    auto stage = [&](int t, int b) {
        const int nrows = a.Sk - t * 64;
        dk_.stage(kg + (long)t * 64 * a.ldk, a.ldk, lb + b * 2 * TILE, w, nrows);
        dv_.stage(vg + (long)t * 64 * a.ldv, a.ldv, lb + b * 2 * TILE + TILE, w, nrows);
    };
    stage(0, 0);
    bf16x8_t qf[KS];
    {
        const bf16_t* qp = a.q + ((long)bz * a.Sq + qrow) * a.ldq + hh * a.D + h * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            qf[s] = 2 * s + h < NCH ? *reinterpret_cast<const bf16x8_t*>(qp + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
    }
    f32x4_t o[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { o[dt][0] = kZero4; o[dt][1] = kZero4; }
    float m = -INFINITY;                                         // running maximum of the RAW scores of query r32 (both lane halves)
    float lsum = 0.f;                                            // !LCOL: this lane half's share of the row sum
    Lanes32<NCH> L;
    L.init(lb, lane);

    auto tile = [&](auto bc, int t) {
        constexpr int b = decltype(bc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < NT) stage(t + 1, b ^ 1);
        constexpr int kb = b * 2 * TILE, vb = kb + TILE;
        f32x16_t s0 = zero16(), s1 = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s0 = mfma32(L.rowfrag(kb, 0, ks), qf[ks], s0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s1 = mfma32(L.rowfrag(kb, 1, ks), qf[ks], s1);
        if (t * 64 + 64 > a.Sk) {                                // rows past the last key (stale or zero): out of the softmax
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                s0[r] = key_past(r, h, t * 64, 0, a.Sk) ? -INFINITY : s0[r];
                s1[r] = key_past(r, h, t * 64, 1, a.Sk) ? -INFINITY : s1[r];
            }
        }
        float mx = max3(s0[0], s0[1], s0[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = max3(mx, s0[r], s0[r + 1]);
        mx = max3(mx, s0[15], s1[0]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = max3(mx, s1[r], s1[r + 1]);
        mx = fmaxf(mx, s1[15]);
        float mxa, mxb;
        both_halves(mx, mxa, mxb);
        const float m_new = max3(m, mxa, mxb);                   // finite: every tile holds at least one valid key
        if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {      // some row's maximum moved: rescale (m = -inf on the first tile: alpha = 0)
            const float alpha = __builtin_amdgcn_exp2f((m - m_new) * a.c);
            const float a0 = __shfl(alpha, lane & 15, 64), a1 = __shfl(alpha, 16 + (lane & 15), 64);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[dt][0][r] *= a0; o[dt][1][r] *= a1; }
            lsum *= alpha;
            m = m_new;
        }
        const float mc = PRE ? m : m * a.c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(PRE ? s0[r] - mc : fmaf(s0[r], a.c, -mc));
            s1[r] = __builtin_amdgcn_exp2f(PRE ? s1[r] - mc : fmaf(s1[r], a.c, -mc));
        }
        if (!LCOL) {
            float t0 = 0.f, t1 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) { t0 += s0[r]; t1 += s1[r]; }
            lsum += t0 + t1;
        }
        bf16x8_t p00, p01, p10, p11;
        to16(s0, p00, p01);
        to16(s1, p10, p11);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8_t v0 = L.trfrag(vb, 0, dt);
            o[dt][0] = mfma16(v0, p00, o[dt][0]);
            o[dt][1] = mfma16(v0, p01, o[dt][1]);
            const bf16x8_t v1 = L.trfrag(vb, 1, dt);
            o[dt][0] = mfma16(v1, p10, o[dt][0]);
            o[dt][1] = mfma16(v1, p11, o[dt][1]);
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        if (t + 1 < NT) tile(std::integral_constant<int, 1>{}, t + 1);
    }
    // row sums in the O layout (l0 / l1: the query lane & 15 of half 0 / 1) and in the S layout (lq: query r32)
    float l0, l1, lq;
    if (LCOL) {                                                  // O^T[D][q] = d tile DT - 1, row 8 of the tile = lanes 32..47, register 0
        l0 = __shfl(o[DT - 1][0][0], 32 + (lane & 15), 64);
        l1 = __shfl(o[DT - 1][1][0], 32 + (lane & 15), 64);
        lq = (r32 & 16) ? l1 : l0;
    } else {
        float x0, x1;
        both_halves(lsum, x0, x1);
        lq = x0 + x1;
        l0 = __shfl(lq, lane & 15, 64);
        l1 = __shfl(lq, 16 + (lane & 15), 64);
    }
#pragma unroll
    for (int qh = 0; qh < 2; ++qh) {
        const float inv = 1.f / (qh ? l1 : l0);
        const int qr = qb * 128 + w * 32 + qh * 16 + (lane & 15);
        bf16_t* og = a.o + ((long)bz * a.Sq + qr) * a.ldo + hh * a.D + 4 * (lane >> 4);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (dt * 16 + 4 * (lane >> 4) < a.D)
                *reinterpret_cast<u32x2_t*>(og + dt * 16) =
                    u32x2_t{pack_bf2(o[dt][qh][0] * inv, o[dt][qh][1] * inv), pack_bf2(o[dt][qh][2] * inv, o[dt][qh][3] * inv)};
    }
    if (h == 0) a.lse2[(long)z * a.Sq + qrow] = (PRE ? m : m * a.c) + __builtin_amdgcn_logf(lq);
}

// (explicit instantiations: hipcc 7.2 emitted the host-side launch stub of only the first kernel of a family when they were
// instantiated implicitly by the launchers below)
#define FA32_INST(NCH)                                               \
    template __global__ void fa32_bwd_dq_kernel<NCH, true>(P32);     \
    template __global__ void fa32_bwd_dq_kernel<NCH, false>(P32);    \
    template __global__ void fa32_bwd_dkdv_kernel<NCH, true, 1>(P32);   \
    template __global__ void fa32_bwd_dkdv_kernel<NCH, false, 1>(P32);  \
    template __global__ void fa32_fwd_kernel<NCH, true>(PF);         \
    template __global__ void fa32_fwd_kernel<NCH, false>(PF);
FA32_INST(5)
FA32_INST(10)
FA32_INST(20)
#undef FA32_INST
template __global__ void fa32_bwd_dkdv_kernel<5, true, 2>(P32);
template __global__ void fa32_bwd_dkdv_kernel<5, false, 2>(P32);

#ifndef FA32_KT2
#define FA32_KT2 1
#endif
constexpr int g_kt2 = FA32_KT2;      // (probe builds: -DFA32_KT2=0 keeps one 32-key tile per wave in the dK / dV kernel)
bool shape_ok(int D, int Sq, int Sk) { return (D == 40 || D == 80 || D == 160) && Sq % 128 == 0 && Sq >= 128 && Sk >= 1; }

}  // namespace

// Shapes the 32x32 forms take: SD v1.5's head dims (40 / 80 / 160: five / ten / twenty 16-B chunks + the augmented one), whole
// 128-query blocks, any number of keys (rows past the last key are neither moved nor counted), 16-B aligned rows.  Everything
// else stays on flash_attn.hip's kernels.
bool siss_fa32_bwd_takes(const FA32Args& a) { return shape_ok(a.D, a.Sq, a.Sk) && a.nB % a.Bf == 0; }
bool siss_fa32_fwd_takes(const FA32FwdArgs& a) { return shape_ok(a.D, a.Sq, a.Sk); }

int siss_fa32_bwd(const FA32Args& a, void* stream) {
    if (!siss_fa32_bwd_takes(a)) return SISS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    P32 p;
    p.q = (const bf16_t*)a.q; p.k = (const bf16_t*)a.k; p.v = (const bf16_t*)a.v; p.o = (const bf16_t*)a.o; p.d_o = (const bf16_t*)a.d_o;
    p.dq = (bf16_t*)a.dq; p.dk = (bf16_t*)a.dk; p.dv = (bf16_t*)a.dv;
    p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo; p.lddo = a.lddo; p.lddq = a.lddq; p.lddk = a.lddk; p.lddv = a.lddv;
    p.lse2 = a.lse2; p.delta = a.delta;
    p.nBH = a.nB * a.H; p.BHf = a.Bf * a.H; p.H = a.H; p.D = a.D; p.Sq = a.Sq; p.Sk = a.Sk;
    p.scale = a.scale;
    p.kscale = a.pre ? 0.6931471805599453f : a.scale;
    p.c = a.pre ? 1.f : a.scale * 1.4426950408889634f;
    // Few key blocks (cross-attention: one) leave the dK / dV grid at a fraction of the chip: the queries are cut into chunks of at
    // least four 64-row tiles, as many as bring the grid to ~1024 blocks and fit the library workspace (partials + a reduce kernel).
    const int nkb = (a.Sk + 127) / 128, DW = ((a.D + 15) / 16) * 16;
    p.nch = 1; p.qchunk = a.Sq; p.part = nullptr;
    {
        long bytes = 0;
        float* ws = (float*)siss_workspace(&bytes);
        const long base = (long)nkb * p.nBH;
        const int qtiles = a.Sq / 64;
        if (ws && base < 512 && qtiles >= 8) {
            long n = (1024 + base - 1) / base;
            if (n > qtiles / 4) n = qtiles / 4;
            const long per_chunk = (long)p.nBH * nkb * 128 * 2 * DW * (long)sizeof(float);
            if (n * per_chunk > bytes - 4096) n = (bytes - 4096) / per_chunk;
            if (n >= 2) {
                const int tiles_per = (int)((qtiles + n - 1) / n);
                p.nch = (qtiles + tiles_per - 1) / tiles_per;
                p.qchunk = tiles_per * 64;
                p.part = ws;
            }
        }
    }
    const unsigned gq = (unsigned)((long)(a.Sq / 128) * p.nBH), gk = (unsigned)((long)nkb * p.nch * p.nBH);
    const bool kt2 = a.D == 40 && !p.part && a.Sk % 256 == 0 && g_kt2;    // 256 keys per block (two 32-key tiles per wave)
#define FA32_GO(NCH, PRE)                                                                                                  \
    do {                                                                                                                   \
        static unsigned char a1[kMaxDevices], a2[kMaxDevices];                                                             \
        constexpr int smem = 4 * G<NCH>::TILE;                                                                             \
        if (siss_ensure_smem((const void*)fa32_bwd_dq_kernel<NCH, PRE>, smem, a1) != SISS_OK) return SISS_ERR_LAUNCH;      \
        if (siss_ensure_smem((const void*)fa32_bwd_dkdv_kernel<NCH, PRE, 1>, smem, a2) != SISS_OK) return SISS_ERR_LAUNCH; \
        fa32_bwd_dq_kernel<NCH, PRE><<<dim3(gq), kT, smem, st>>>(p);                                                       \
        if (NCH == 5 && kt2) {                                                                                             \
            static unsigned char a3[kMaxDevices];                                                                          \
            if (siss_ensure_smem((const void*)fa32_bwd_dkdv_kernel<5, PRE, 2>, 4 * G<5>::TILE, a3) != SISS_OK) return SISS_ERR_LAUNCH; \
            fa32_bwd_dkdv_kernel<5, PRE, 2><<<dim3(gk / 2), kT, 4 * G<5>::TILE, st>>>(p);                                  \
        } else {                                                                                                           \
            fa32_bwd_dkdv_kernel<NCH, PRE, 1><<<dim3(gk), kT, smem, st>>>(p);                                              \
        }                                                                                                                  \
    } while (0)
#define FA32_GO2(NCH) do { if (a.pre) FA32_GO(NCH, true); else FA32_GO(NCH, false); } while (0)
    if (a.D == 40) FA32_GO2(5); else if (a.D == 80) FA32_GO2(10); else FA32_GO2(20);
#undef FA32_GO2
#undef FA32_GO
    if (p.part) {
        siss_count_dispatch(SISS_K_FLASH_QSPLIT);
        const long total = (long)p.nBH * a.Sk * (a.D >> 2);
        fa32_dkdv_reduce_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(p, 2 * DW / 2, total);
    }
    siss_count_dispatch(SISS_K_FLASH32);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

int siss_fa32_fwd(const FA32FwdArgs& a, void* stream) {
    if (!siss_fa32_fwd_takes(a)) return SISS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    PF p;
    p.q = (const bf16_t*)a.q; p.k = (const bf16_t*)a.k; p.v = (const bf16_t*)a.v; p.o = (bf16_t*)a.o;
    p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo; p.lse2 = a.lse2;
    p.BH = a.B * a.H; p.H = a.H; p.D = a.D; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.pre ? 1.f : a.scale * 1.4426950408889634f;
    const unsigned grid = (unsigned)((long)(a.Sq / 128) * p.BH);
#define FA32_FWD(NCH, PRE)                                                                                                \
    do {                                                                                                                  \
        static unsigned char a1[kMaxDevices];                                                                             \
        constexpr int smem = 4 * G<NCH>::TILE;                                                                            \
        if (siss_ensure_smem((const void*)fa32_fwd_kernel<NCH, PRE>, smem, a1) != SISS_OK) return SISS_ERR_LAUNCH;        \
        fa32_fwd_kernel<NCH, PRE><<<dim3(grid), kT, smem, st>>>(p);                                                       \
    } while (0)
#define FA32_FWD2(NCH) do { if (a.pre) FA32_FWD(NCH, true); else FA32_FWD(NCH, false); } while (0)
    if (a.D == 40) FA32_FWD2(5); else if (a.D == 80) FA32_FWD2(10); else FA32_FWD2(20);
#undef FA32_FWD2
#undef FA32_FWD
    siss_count_dispatch(SISS_K_FLASH32_FWD);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}
