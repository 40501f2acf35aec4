// Fused attention BACKWARD for narrow heads (D <= 48: SD v1.5's 4096-key self-attention sites, D = 40; delete_sd.py:977-985 ->
// losses/ddpm_deletion_loss.py:24, differentiated twice at delete_sd.py:1040-1060) on v_mfma_f32_32x32x16_bf16.
//
// Why a second form beside flash_attn.hip's 16x16x32 kernels (round 6): at D = 40 those pad the contraction over the head dim to
// 64 (two 32-deep steps: 37 % dead MFMA work in S and dP), hold the SIMD's vector issue for 8 of every 16 matrix cycles, read one LDS
// fragment per MFMA (a wave owns 16 rows) and stage every tile global -> registers -> LDS through the VALU: 441 TF/s algorithmic,
// 40 % matrix-pipe utilisation, in loops that are bound by vector issue.  Here:
//   * the contraction runs in 16-deep steps: 40 + the 8-wide augmented chunk = 48 = three 32x32x16 MFMAs, nothing dead;
//   * a 32x32x16 MFMA blocks vector issue for 8 of 32 cycles: per score element the matrix pipe leaves twice the issue slots;
//   * a wave owns 32 rows; its own operands (Q / dO rows in the dQ kernel, K / V rows in the dK / dV kernel) live in registers as
//     B fragments, the streamed tensor's fragments (A operands) are read from LDS once per 32 x 32 score block;
//   * tiles travel global -> LDS by LDS-DMA (global_load_lds_dwordx4), source-side XOR swizzle, pad chunks never moved (EXEC-masked
//     lanes leave the zeros / ones written once per block), the next tile in flight under the current one, ONE barrier per tile.
// Everything else is flash_attn.hip's algorithm: FlashAttention-2 recompute from the saved base-2 log-sum-exp, everything transposed
// so that P / dS feed the next product straight from the accumulator registers, the augmented contraction (three bf16 parts of
// -lse against ones deliver s - lse, two parts of -delta against ones deliver dP - delta), delta formed by the dQ kernel.
//
//   S^T[key][q] = mfma32(K rows (LDS, row-major b128), Q frag)      lane: ONE query (lane & 31), 16 keys of the 32-key block in
//   dQ^T[d][q] += mfma32(K^T (LDS, transposed reads), dS)           registers: key (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// The accumulator registers 8 s .. 8 s + 7, packed pairwise, ARE the B fragment of 16-key step s (k order: key 16 s + 8 (j >> 2)
// + 4 h + (j & 3) for element j of lane half h); the transposed A fragment is fetched in the same order by two ds_read_b64_tr_b16.
// The dK / dV kernel is the mirror image (lane: one key; S[q][key] = mfma32(Q rows, K frag)).
//
// LDS tile: 64 rows x 128 B (eight 16-B chunks: D / 8 data chunks, the augmented chunk, zeros), chunk slot = chunk ^ f(row),
// f(row) = ((row >> 1 & 1) << 2) | (row >> 2 & 3): conflict-free for the 32-row b128 fragment reads (lane groups {0-3, 12-15, 20-27},
// {4-11, 16-19, 28-31}: eight distinct values per row parity) and for the transposed reads (4 rows x 64 B per 32 lanes: rows r and
// r + 2 differ in slot bit 2, rows r and r + 1 in the 128-B half).
#include "common.h"
#include "flash32.h"
#include <type_traits>

namespace {

// FA32_ABL (probe builds only, tools/probes/fa32_ablate.sh; wrong results): 1 no exp2, 2 no output products, 3 no softmax arithmetic
// at all, 4 one staged tile re-read for the whole loop (no DMA, no barrier)
#ifndef FA32_ABL
#define FA32_ABL 0
#endif

constexpr int kT = 256;
constexpr int kTileB = 64 * 128;          // bytes per staged 64-row tile
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
__device__ __forceinline__ int fsw(int r) { return (((r >> 1) & 1) << 2) | (((r >> 4) & 1) << 1) | ((r >> 2) & 1); }
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
// accumulator registers 8 s .. 8 s + 7 (times m) -> the B fragment of 16-row step s
__device__ __forceinline__ bf16x8_t pack8(const f32x16_t& x, int s) {
    return as_frag(u32x4_t{pack_bf2(x[8 * s + 0], x[8 * s + 1]), pack_bf2(x[8 * s + 2], x[8 * s + 3]),
                           pack_bf2(x[8 * s + 4], x[8 * s + 5]), pack_bf2(x[8 * s + 6], x[8 * s + 7])});
}
__device__ __forceinline__ f32x16_t zero16() {
    f32x16_t z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}

struct P32 {
    const bf16_t *q, *k, *v, *o, *d_o;
    bf16_t *dq, *dk, *dv;
    long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
    const float* lse2;
    float* delta;
    int nBH, BHf, H, D, Sq, Sk;        // cotangent (batch, head) entries, forward entries, heads, head dim, rows
    float scale, kscale, c;            // dQ factor, dK factor, score -> base-2 logit factor (1 when q is pre-scaled)
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
template <int KS, int DT> struct Lanes32 {
    unsigned rm[KS];        // row-major b128: row (lane & 31), chunk 2 s + h
    unsigned tr[DT];        // transposed (16-wide d tile dt): rows 16 (g & 1) + 4 (g >> 1) + (lane >> 2 & 3), g = lane >> 4; second read: + 8 rows
    __device__ __forceinline__ void init(unsigned lb, int lane) {
        const int h = lane >> 5, r32 = lane & 31;
#pragma unroll
        for (int s = 0; s < KS; ++s) rm[s] = lb + r32 * 128 + (((2 * s + h) ^ fsw(r32)) << 4);
        const int g = lane >> 4, q4 = (lane >> 2) & 3, pp = lane & 3;
        const int row = 16 * (g & 1) + 4 * (g >> 1) + q4;          // (fsw ignores row bit 3: the second read shares the swizzle)
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) tr[dt] = lb + row * 128 + (((dt * 2 + (pp >> 1)) ^ fsw(row)) << 4) + 8 * (pp & 1);
    }
};
// A fragment of v_mfma_f32_16x16x32 [16 rows d = dt * 16 + (lane & 15)][32 k = the rows of the 32-row half in to16's order]
template <int KS, int DT>
__device__ __forceinline__ bf16x8_t frag_tr16(const Lanes32<KS, DT>& L, int tile_off, int half, int dt) {
    const int imm = tile_off + half * 32 * 128;
    const s16x4_t a0 = tr_read(L.tr[dt] + imm), a1 = tr_read(L.tr[dt] + imm + 8 * 128);
    return bf16x8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
}

// LDS-DMA of one 64-row tile: wave w moves pieces w and w + 4 (8 rows x 128 B each; both have the row parity bit 3 = w & 1, so a
// lane's source chunk and its activity are fixed for the kernel).  src_off: this lane's byte offset from the piece's first row.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ row0, long ld, unsigned src_off, bool active, unsigned lds_tile, int w) {
    if (active) {
        glds16_saddr(src_off, row0 + (long)(8 * w) * ld, lds_tile + w * 1024);
        glds16_saddr(src_off, row0 + (long)(8 * w + 32) * ld, lds_tile + (w + 4) * 1024);
    }
}
template <int NCH>
__device__ __forceinline__ void dma_lane(int lane, int w, long ld, unsigned& off, bool& active) {
    const int lr = lane >> 3, sl = lane & 7;
    const int c = sl ^ fsw(8 * w + lr);                    // (pieces w and w + 4: same row bits 1, 2 and 4)
    active = c < NCH;
    off = (unsigned)((lr * ld + c * 8) * 2);
}

// =====================================================================================================================
// dQ = scale * dS K,  dS = P o (dO V^T - delta),  delta = rowsum(dO o O) (written for the dK / dV kernel)
// block: 128 queries of one cotangent (batch, head) entry (a wave: 32); grid: Sq / 128 x nBH blocks (block_map)
// =====================================================================================================================
template <int NCH, bool PRE>
__global__ __launch_bounds__(kT, 4) void fa32_bwd_dq_kernel(P32 a) {
    constexpr int KS = (NCH + 2) / 2, DT = (NCH * 8 + 15) / 16;
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
#pragma unroll
    for (int i = 0; i < 4 * kTileB / 16 / kT; ++i) reinterpret_cast<u32x4_t*>(smem)[i * kT + tid] = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();
    {
        const int tile = tid >> 6, row = tid & 63;
        *reinterpret_cast<u32x4_t*>(smem + tile * kTileB + row * 128 + ((NCH ^ fsw(row)) << 4)) =
            (tile & 1) ? u32x4_t{kOne2, 0u, 0u, 0u} : u32x4_t{kOne2, kOne1, 0u, 0u};
    }
    __syncthreads();

    const bf16_t* kg = a.k + (long)bf * a.Sk * a.ldk + hh * a.D;
    const bf16_t* vg = a.v + (long)bf * a.Sk * a.ldv + hh * a.D;
    unsigned koff, voff; bool kact, vact;
    dma_lane<NCH>(lane, w, a.ldk, koff, kact);
    dma_lane<NCH>(lane, w, a.ldv, voff, vact);
    stage_tile(kg, a.ldk, koff, kact, lb, w);
    stage_tile(vg, a.ldv, voff, vact, lb + kTileB, w);

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
    for (int dt = 0; dt < DT; ++dt) { dq[dt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dq[dt][1] = dq[dt][0]; }
    Lanes32<KS, DT> L;
    L.init(lb, lane);

    const int NT = a.Sk / 64;
    auto tile = [&](auto bc, int t) {
        constexpr int b = decltype(bc)::value;
        if (FA32_ABL != 4 || t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (t + 1 < NT && (FA32_ABL != 4 || t == 0)) {
            stage_tile(kg + (long)(t + 1) * 64 * a.ldk, a.ldk, koff, kact, lb + (b ^ 1) * 2 * kTileB, w);
            stage_tile(vg + (long)(t + 1) * 64 * a.ldv, a.ldv, voff, vact, lb + (b ^ 1) * 2 * kTileB + kTileB, w);
        }
        constexpr int kb = b * 2 * kTileB, vb = kb + kTileB;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16_t s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s = mfma32(rd128(L.rm[ks] + kb + half * 32 * 128), qf[ks], s);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dp = mfma32(rd128(L.rm[ks] + vb + half * 32 * 128), dof[ks], dp);
            // (the MFMAs subtracted lse and delta) dS without its factor `scale`: applied once to the finished tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (FA32_ABL == 3) { s[r] += dp[r]; continue; }
                s[r] = (FA32_ABL == 1 ? s[r] : __builtin_amdgcn_exp2f(PRE ? s[r] : s[r] * a.c)) * dp[r];
            }
            bf16x8_t ds0, ds1;
            to16(s, ds0, ds1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (FA32_ABL == 2) { dq[dt][0][0] += __builtin_bit_cast(float, (int)ds0[0] | (int)ds1[1]); continue; }
                const bf16x8_t kt = frag_tr16<KS, DT>(L, kb, half, dt);
                dq[dt][0] = mfma16(kt, ds0, dq[dt][0]);
                dq[dt][1] = mfma16(kt, ds1, dq[dt][1]);
            }
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
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
// block: 128 keys of one cotangent (batch, head) entry (a wave: 32); grid: Sk / 128 x nBH blocks (block_map)
// =====================================================================================================================
template <int NCH, bool PRE>
__global__ __launch_bounds__(kT, 3) void fa32_bwd_dkdv_kernel(P32 a) {
    constexpr int KS = (NCH + 2) / 2, DT = (NCH * 8 + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 buffers][Q tile | dO tile]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    int kb_, z;
    block_map(a.Sk / 128, a.nBH, a.BHf, kb_, z);
    const int bz = z / a.H, hh = z - bz * a.H;
    const int zf = z % a.BHf, bf = zf / a.H;
    const int krow = kb_ * 128 + w * 32 + r32;
    const unsigned lb = lds_addr(smem);
#pragma unroll
    for (int i = 0; i < 4 * kTileB / 16 / kT; ++i) reinterpret_cast<u32x4_t*>(smem)[i * kT + tid] = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();

    const bf16_t* qg = a.q + (long)bf * a.Sq * a.ldq + hh * a.D;
    const bf16_t* dog = a.d_o + (long)bz * a.Sq * a.lddo + hh * a.D;
    unsigned qoff, dooff; bool qact, doact;
    dma_lane<NCH>(lane, w, a.ldq, qoff, qact);
    dma_lane<NCH>(lane, w, a.lddo, dooff, doact);
    stage_tile(qg, a.ldq, qoff, qact, lb, w);
    stage_tile(dog, a.lddo, dooff, doact, lb + kTileB, w);
    // The augmented chunk of the streamed tiles changes per row: wave 0 writes the three parts of -lse of the tile's 64 queries into
    // the Q tile, wave 1 the two parts of -delta into the dO tile (one lane per row), one tile ahead, from a value loaded two ahead.
    const float* aug_src = w == 0 ? a.lse2 + (long)zf * a.Sq : a.delta + (long)z * a.Sq;
    const float amul = (w == 0 && !PRE) ? -1.f / a.c : -1.f;
    char* const aug_dst = smem + (w == 1 ? kTileB : 0) + lane * 128 + ((NCH ^ fsw(lane)) << 4);
    const int NT = a.Sq / 64;
    float a_nx = 0.f;
    if (w < 2) {
        const float a0 = aug_src[lane];
        if (NT > 1) a_nx = aug_src[64 + lane];
        *reinterpret_cast<u32x4_t*>(aug_dst) = split_bf16(a0 * amul, w == 0 ? 3 : 2);
    }

    // ---- this wave's 32 keys: K and V as B fragments (+ the ones against the augmented columns)
    bf16x8_t kf[KS], vf[KS];
    {
        const long col = hh * a.D + h * 8;
        const bf16_t* kp = a.k + ((long)bf * a.Sk + krow) * a.ldk + col;
        const bf16_t* vp = a.v + ((long)bf * a.Sk + krow) * a.ldv + col;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const bool ok = 2 * s + h < NCH;
            kf[s] = ok ? *reinterpret_cast<const bf16x8_t*>(kp + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
            vf[s] = ok ? *reinterpret_cast<const bf16x8_t*>(vp + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
        }
        if (h == (NCH & 1)) {
            kf[NCH / 2] = as_frag(u32x4_t{kOne2, kOne1, 0u, 0u});
            vf[NCH / 2] = as_frag(u32x4_t{kOne2, 0u, 0u, 0u});
        }
    }
    f32x4_t dk[DT][2], dv[DT][2];                                // [d tile][key half]: lane = key (lane & 15), 4 d per tile
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { dk[dt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dk[dt][1] = dk[dt][0]; dv[dt][0] = dk[dt][0]; dv[dt][1] = dk[dt][0]; }
    Lanes32<KS, DT> L;
    L.init(lb, lane);

    auto tile = [&](auto bc, int t) {
        constexpr int b = decltype(bc)::value;
        if (FA32_ABL != 4 || t == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (t + 1 < NT && (FA32_ABL != 4 || t == 0)) {
            stage_tile(qg + (long)(t + 1) * 64 * a.ldq, a.ldq, qoff, qact, lb + (b ^ 1) * 2 * kTileB, w);
            stage_tile(dog + (long)(t + 1) * 64 * a.lddo, a.lddo, dooff, doact, lb + (b ^ 1) * 2 * kTileB + kTileB, w);
            if (w < 2) {
                *reinterpret_cast<u32x4_t*>(aug_dst + (b ^ 1) * 2 * kTileB) = split_bf16(a_nx * amul, w == 0 ? 3 : 2);
                if (t + 2 < NT) a_nx = aug_src[(t + 2) * 64 + lane];
            }
        }
        constexpr int qb = b * 2 * kTileB, dob = qb + kTileB;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            f32x16_t s = zero16(), dp = zero16();
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) s = mfma32(rd128(L.rm[ks] + qb + half * 32 * 128), kf[ks], s);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) dp = mfma32(rd128(L.rm[ks] + dob + half * 32 * 128), vf[ks], dp);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (FA32_ABL == 3) { dp[r] += s[r]; continue; }
                s[r] = FA32_ABL == 1 ? s[r] : __builtin_amdgcn_exp2f(PRE ? s[r] : s[r] * a.c);
                dp[r] *= s[r];
            }
            bf16x8_t p0, p1, ds0, ds1;
            to16(s, p0, p1);
            to16(dp, ds0, ds1);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt) {
                if (FA32_ABL == 2) { dv[dt][0][0] += __builtin_bit_cast(float, (int)p0[0] | (int)p1[1] | (int)ds0[2] | (int)ds1[3]); continue; }
                const bf16x8_t dot = frag_tr16<KS, DT>(L, dob, half, dt);
                dv[dt][0] = mfma16(dot, p0, dv[dt][0]);
                dv[dt][1] = mfma16(dot, p1, dv[dt][1]);
                const bf16x8_t qt = frag_tr16<KS, DT>(L, qb, half, dt);
                dk[dt][0] = mfma16(qt, ds0, dk[dt][0]);
                dk[dt][1] = mfma16(qt, ds1, dk[dt][1]);
            }
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
    }
#pragma unroll
    for (int kh = 0; kh < 2; ++kh) {
        const int kr = kb_ * 128 + w * 32 + kh * 16 + (lane & 15);
        bf16_t* okg = a.dk + ((long)bz * a.Sk + kr) * a.lddk + hh * a.D + 4 * (lane >> 4);
        bf16_t* ovg = a.dv + ((long)bz * a.Sk + kr) * a.lddv + hh * a.D + 4 * (lane >> 4);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt)
            if (dt * 16 + 4 * (lane >> 4) < a.D) {
                *reinterpret_cast<u32x2_t*>(okg + dt * 16) =
                    u32x2_t{pack_bf2(dk[dt][kh][0] * a.kscale, dk[dt][kh][1] * a.kscale), pack_bf2(dk[dt][kh][2] * a.kscale, dk[dt][kh][3] * a.kscale)};
                *reinterpret_cast<u32x2_t*>(ovg + dt * 16) =
                    u32x2_t{pack_bf2(dv[dt][kh][0], dv[dt][kh][1]), pack_bf2(dv[dt][kh][2], dv[dt][kh][3])};
            }
    }
}

// =====================================================================================================================
// forward: O = softmax(c' Q K^T) V, LSE2[q] = base-2 log-sum-exp of the scaled scores
// block: 128 queries of one (batch, head) entry (a wave: 32); grid: Sq / 128 x BH blocks (block_map)
// A 64-key tile: S^T of both 32-key halves (6 MFMAs 32x32x16), ONE running-maximum update for the 64 keys (v_max3 chains + one
// v_permlane32_swap), p = exp2(s - m), P V on 16x16x32 (12 MFMAs).  The row sum rides in the product: V's pad column D holds ones,
// so O^T[D][q] accumulates sum_k p -- with O's rescaling -- in the last d tile (D % 16 == 8).  S lives in the 32x32 layout
// (lane = query lane & 31), O in the 16x16 layout (lane = query lane & 15 of its half): the rescale factor crosses between the
// two by a bpermute, but only in tiles where some row's maximum moved (a wave-uniform branch, rare after the first tiles).
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
__device__ __forceinline__ float max_both_halves(float v) {      // max(v of this lane, v of lane ^ 32)
    const unsigned u = __builtin_bit_cast(unsigned, v);
    unsigned u2 = u;
    asm volatile("" : "+v"(u2));                                 // (see sum_lanes_mod8: keeps hipcc from folding the swap of a value with itself)
    const auto r = __builtin_amdgcn_permlane32_swap(u, u2, false, false);     // one result is this lane's value, the other its partner's
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("" : "+v"(r0), "+v"(r1));
    return fmaxf(__builtin_bit_cast(float, r0), __builtin_bit_cast(float, r1));
}

template <int NCH, bool PRE>
__global__ __launch_bounds__(kT, 4) void fa32_fwd_kernel(PF a) {
    constexpr int KS = (NCH + 2) / 2, DT = (NCH * 8 + 15) / 16;
    static_assert((NCH & 1) == 1, "the ones column needs a pad column inside the last 16-wide d tile (D % 16 == 8)");
    extern __shared__ __attribute__((aligned(16))) char smem[];          // [2 buffers][K tile | V tile]
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = lane >> 5, r32 = lane & 31;
    int qb, z;
    block_map(a.Sq / 128, a.BH, a.BH, qb, z);
    const int bz = z / a.H, hh = z - bz * a.H;
    const int qrow = qb * 128 + w * 32 + r32;
    const unsigned lb = lds_addr(smem);
#pragma unroll
    for (int i = 0; i < 4 * kTileB / 16 / kT; ++i) reinterpret_cast<u32x4_t*>(smem)[i * kT + tid] = u32x4_t{0u, 0u, 0u, 0u};
    __syncthreads();
    if (tid < 128) {                                             // V tiles: column D := 1
        const int tile = 1 + 2 * (tid >> 6), row = tid & 63;
        *reinterpret_cast<u32x4_t*>(smem + tile * kTileB + row * 128 + ((NCH ^ fsw(row)) << 4)) = u32x4_t{kOne1, 0u, 0u, 0u};
    }
    __syncthreads();
    const bf16_t* kg = a.k + (long)bz * a.Sk * a.ldk + hh * a.D;
    const bf16_t* vg = a.v + (long)bz * a.Sk * a.ldv + hh * a.D;
    unsigned koff, voff; bool kact, vact;
    dma_lane<NCH>(lane, w, a.ldk, koff, kact);
    dma_lane<NCH>(lane, w, a.ldv, voff, vact);
    stage_tile(kg, a.ldk, koff, kact, lb, w);
    stage_tile(vg, a.ldv, voff, vact, lb + kTileB, w);
    bf16x8_t qf[KS];
    {
        const bf16_t* qp = a.q + ((long)bz * a.Sq + qrow) * a.ldq + hh * a.D + h * 8;
#pragma unroll
        for (int s = 0; s < KS; ++s)
            qf[s] = 2 * s + h < NCH ? *reinterpret_cast<const bf16x8_t*>(qp + s * 16) : as_frag(u32x4_t{0u, 0u, 0u, 0u});
    }
    f32x4_t o[DT][2];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) { o[dt][0] = f32x4_t{0.f, 0.f, 0.f, 0.f}; o[dt][1] = o[dt][0]; }
    float m = -INFINITY;                                         // running maximum of the RAW scores of query r32 (both lane halves)
    Lanes32<KS, DT> L;
    L.init(lb, lane);

    const int NT = a.Sk / 64;
    auto tile = [&](auto bc, int t) {
        constexpr int b = decltype(bc)::value;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 1 < NT) {
            stage_tile(kg + (long)(t + 1) * 64 * a.ldk, a.ldk, koff, kact, lb + (b ^ 1) * 2 * kTileB, w);
            stage_tile(vg + (long)(t + 1) * 64 * a.ldv, a.ldv, voff, vact, lb + (b ^ 1) * 2 * kTileB + kTileB, w);
        }
        constexpr int kb = b * 2 * kTileB, vb = kb + kTileB;
        f32x16_t s0 = zero16(), s1 = zero16();
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s0 = mfma32(rd128(L.rm[ks] + kb), qf[ks], s0);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) s1 = mfma32(rd128(L.rm[ks] + kb + 32 * 128), qf[ks], s1);
        float mx = max3(s0[0], s0[1], s0[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) mx = max3(mx, s0[r], s0[r + 1]);
        mx = max3(mx, s0[15], s1[0]);
#pragma unroll
        for (int r = 1; r < 15; r += 2) mx = max3(mx, s1[r], s1[r + 1]);
        mx = fmaxf(mx, s1[15]);
        const float m_new = fmaxf(m, max_both_halves(mx));
        if (__builtin_amdgcn_ballot_w64(m_new != m) != 0) {      // some row's maximum moved: rescale (m = -inf on the first tile: alpha = 0)
            const float alpha = __builtin_amdgcn_exp2f((m - m_new) * a.c);
            const float a0 = __shfl(alpha, lane & 15, 64), a1 = __shfl(alpha, 16 + (lane & 15), 64);
#pragma unroll
            for (int dt = 0; dt < DT; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) { o[dt][0][r] *= a0; o[dt][1][r] *= a1; }
            m = m_new;
        }
        const float mc = PRE ? m : m * a.c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            s0[r] = __builtin_amdgcn_exp2f(PRE ? s0[r] - mc : fmaf(s0[r], a.c, -mc));
            s1[r] = __builtin_amdgcn_exp2f(PRE ? s1[r] - mc : fmaf(s1[r], a.c, -mc));
        }
        bf16x8_t p00, p01, p10, p11;
        to16(s0, p00, p01);
        to16(s1, p10, p11);
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
            const bf16x8_t v0 = frag_tr16<KS, DT>(L, vb, 0, dt);
            o[dt][0] = mfma16(v0, p00, o[dt][0]);
            o[dt][1] = mfma16(v0, p01, o[dt][1]);
            const bf16x8_t v1 = frag_tr16<KS, DT>(L, vb, 1, dt);
            o[dt][0] = mfma16(v1, p10, o[dt][0]);
            o[dt][1] = mfma16(v1, p11, o[dt][1]);
        }
    };
    for (int t = 0; t < NT; t += 2) {
        tile(std::integral_constant<int, 0>{}, t);
        tile(std::integral_constant<int, 1>{}, t + 1);
    }
    // row sums: O^T[D][q] = d tile DT - 1, row 8 of the tile = lanes 32..47, register 0
    const float l0 = __shfl(o[DT - 1][0][0], 32 + (lane & 15), 64), l1 = __shfl(o[DT - 1][1][0], 32 + (lane & 15), 64);
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
    // lane r32 < 32 (h = 0) keeps query r32's maximum; its row sum sits in l0 / l1 of the lanes whose lane & 15 == r32 & 15
    const float lq = (r32 & 16) ? l1 : l0;
    if (h == 0) a.lse2[(long)z * a.Sq + qrow] = (PRE ? m : m * a.c) + __builtin_amdgcn_logf(lq);
}

// (explicit instantiations: hipcc 7.2 emitted the host-side launch stub of only the first of these four when they were
// instantiated implicitly by the launcher below)
template __global__ void fa32_bwd_dq_kernel<5, true>(P32);
template __global__ void fa32_bwd_dq_kernel<5, false>(P32);
template __global__ void fa32_bwd_dkdv_kernel<5, true>(P32);
template __global__ void fa32_bwd_dkdv_kernel<5, false>(P32);
template __global__ void fa32_fwd_kernel<5, true>(PF);
template __global__ void fa32_fwd_kernel<5, false>(PF);

}  // namespace

// Shapes the 32x32 form takes: head dim 40 (five 16-B chunks + the augmented one = three 16-deep steps), whole 128-row blocks on
// both sides, every key valid, 16-B aligned rows.  Everything else stays on flash_attn.hip's kernels.
bool siss_fa32_bwd_takes(const FA32Args& a) {
    return a.D == 40 && a.Sq % 128 == 0 && a.Sk % 128 == 0 && a.Sq >= 128 && a.Sk >= 128 && a.nB % a.Bf == 0;
}

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
    constexpr int smem = 4 * kTileB;
    const unsigned gq = (unsigned)((long)(a.Sq / 128) * p.nBH), gk = (unsigned)((long)(a.Sk / 128) * p.nBH);
#define FA32_GO(PRE)                                                                                                       \
    do {                                                                                                                   \
        static unsigned char a1[kMaxDevices], a2[kMaxDevices];                                                             \
        if (siss_ensure_smem((const void*)fa32_bwd_dq_kernel<5, PRE>, smem, a1) != SISS_OK) return SISS_ERR_LAUNCH;       \
        if (siss_ensure_smem((const void*)fa32_bwd_dkdv_kernel<5, PRE>, smem, a2) != SISS_OK) return SISS_ERR_LAUNCH;     \
        fa32_bwd_dq_kernel<5, PRE><<<dim3(gq), kT, smem, st>>>(p);                                                        \
        fa32_bwd_dkdv_kernel<5, PRE><<<dim3(gk), kT, smem, st>>>(p);                                                      \
    } while (0)
    if (a.pre) FA32_GO(true); else FA32_GO(false);
#undef FA32_GO
    siss_count_dispatch(SISS_K_FLASH32);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

bool siss_fa32_fwd_takes(const FA32FwdArgs& a) {
    return a.D == 40 && a.Sq % 128 == 0 && a.Sk % 128 == 0 && a.Sq >= 128 && a.Sk >= 128;
}

int siss_fa32_fwd(const FA32FwdArgs& a, void* stream) {
    if (!siss_fa32_fwd_takes(a)) return SISS_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    PF p;
    p.q = (const bf16_t*)a.q; p.k = (const bf16_t*)a.k; p.v = (const bf16_t*)a.v; p.o = (bf16_t*)a.o;
    p.ldq = a.ldq; p.ldk = a.ldk; p.ldv = a.ldv; p.ldo = a.ldo; p.lse2 = a.lse2;
    p.BH = a.B * a.H; p.H = a.H; p.D = a.D; p.Sq = a.Sq; p.Sk = a.Sk;
    p.c = a.pre ? 1.f : a.scale * 1.4426950408889634f;
    constexpr int smem = 4 * kTileB;
    const unsigned grid = (unsigned)((long)(a.Sq / 128) * p.BH);
#define FA32_FWD(PRE)                                                                                                     \
    do {                                                                                                                  \
        static unsigned char a1[kMaxDevices];                                                                             \
        if (siss_ensure_smem((const void*)fa32_fwd_kernel<5, PRE>, smem, a1) != SISS_OK) return SISS_ERR_LAUNCH;          \
        fa32_fwd_kernel<5, PRE><<<dim3(grid), kT, smem, st>>>(p);                                                         \
    } while (0)
    if (a.pre) FA32_FWD(true); else FA32_FWD(false);
#undef FA32_FWD
    siss_count_dispatch(SISS_K_FLASH32_FWD);
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}
