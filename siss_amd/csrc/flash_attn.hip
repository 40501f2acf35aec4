// Fused multi-head attention for the SD UNet's transformer blocks (delete_sd.py:977-985 -> losses/ddpm_deletion_loss.py:24,
// diffusers BasicTransformerBlock attn1 / attn2): QK^T -> softmax -> .V in ONE kernel, and its backward in two, on bf16 MFMA
// with LDS-staged tiles.  The S x S score / probability matrices never touch HBM (at 64 x 64 latents they are 1 GB per site
// and pass in the GEMM + softmax form this replaces: 22 of the 77 ms of the SD-v1.5 step).
//
// Operands are addressed as (base, row stride): the kernels read q / k / v / dO and write o / dq / dk / dv IN PLACE in the layout
// the projections use -- "merged" [B][S][heads * D] rows with the head at column h * D (siss_flash_attn_*_merged) -- or in the
// head-split, zero-padded [B*heads][S_pad][D_pad] form (siss_flash_attn_fwd / _bwd).  Rows past the tensor's end and head-dim
// chunks past D read as zero and are not stored, so the merged form needs no head-split / head-merge copies and no padded
// tensors (round 3: 256 launches and ~4 ms per SD step at B = 16 gone).  `valid_k` masks the padded keys (cross-attention: 77).
// delta = rowsum(dO o O) is formed by the dQ kernel from the tiles it loads anyway (it runs first; the dK / dV kernel reads it).
//
// Everything is computed TRANSPOSED, so that no register shuffle is ever needed between the two products of a tile:
//   v_mfma_f32_16x16x32_bf16(X, Y): out[x = (lane >> 4) * 4 + r][y = lane & 15], both operands "row, 8 consecutive k per lane".
//   forward / dQ kernels (a block owns 64 queries, a wave 16):
//       S^T[key][q]  = mfma(K rows, Q rows)             lane: ONE query, 4 consecutive keys of each 16-key sub-tile
//       O^T[d][q]   += mfma(V^T (transposed LDS read), P)   P as the Y operand = the S^T registers, packed -- the k-slot order of
//                                                        a 32-key step is [sub-tile 2j keys 4g..4g+3 | sub-tile 2j+1 keys 4g..4g+3],
//                                                        and the V^T fragment is fetched in the same order
//     row statistics (max, sum, lse, delta) are per-LANE scalars (a lane keeps one query), reductions are two xor-shuffles.
//   dK/dV kernel (a block owns 64 keys): the mirror image with S[q][key] (lane: one key).
// Online softmax in base 2 (scale * log2(e) folded into the scores); LSE is kept in base-2 units for the backward, which
// recomputes P = exp2(s - lse) per tile (FlashAttention-2: 7 products instead of 5, no S x S traffic).
// LDS tiles: 64 rows x D_pad, 16-B chunk index XOR f(row) -- conflict-free for the row-major ds_read_b128 fragment reads AND
// for the ds_read_b64_tr_b16 transposed reads (see swz()).
#include "common.h"
#include "flash32.h"

namespace {

constexpr int kTQ = 64;                  // rows (queries or keys) per block and per LDS tile
constexpr int kThreadsFA = 256;

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ s16x4_t tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
}

template <int DP> struct FA {
    static constexpr int RS = DP == 64 ? 128 : (DP == 128 ? 256 : 512);   // LDS row stride in bytes (192 is padded to 256 elements)
    static constexpr int KS = DP / 32;                                     // 32-deep k-steps over the head dimension
    static constexpr int DT = DP / 16;                                     // 16-wide d tiles
    static constexpr int TILE = kTQ * RS;                                  // bytes per staged tile
    static constexpr int CH = DP / 8;                                      // logical 16-B chunks per row
    // f(row): XOR on the 16-B chunk index.  Row-major fragment reads (16 rows, one chunk column per quarter wave) need the 16
    // (row, chunk) pairs on 16 distinct 16-B bank groups; transposed reads (8 rows x one 32-B column per half wave) need 8
    // distinct 32-B bank groups.  RS = 128: rows alternate between the two halves of the 256-B bank period.
    // (Checked against the lane groups the LDS really serves -- a ds_read_b128 goes in four groups of 16 NON-contiguous lanes,
    // {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and the same + 32 -- for the row-major reads, the transposed reads (2 x 32 lanes)
    // and the tile stores (8 x 8 lanes): these two forms are conflict-free for all three.  The earlier forms, which also folded
    // row bit 3 into chunk bit 0, made every row-major read 2-way conflicted: SQ_LDS_BANK_CONFLICT was 31-35 % of SQ_LDS_IDX_ACTIVE.)
    __device__ static __forceinline__ int swz(int row) { return RS == 128 ? (row & 6) : ((row & 7) << 1); }
    __device__ static __forceinline__ int off(int row, int chunk) { return row * RS + ((chunk ^ swz(row)) << 4); }
};

// A [64][DP] tile (global row stride DP; the tensors are padded, all 64 rows exist) travels global -> registers -> LDS in two
// halves, so that the loads of tile t+1 are in flight while tile t is being multiplied (one LDS buffer, register prefetch).
template <int DP> struct TileRegs { u32x4_t v[FA<DP>::CH * kTQ / kThreadsFA]; };
// g: first row of the tile (already at the block's batch / head), ld: row stride in elements, rows: rows of the tile that exist
// (<= 64), dch: 16-B chunks of a row that exist (D / 8); everything else reads as zero
template <int DP>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ g, long ld, int rows, int dch, TileRegs<DP>& r, int tid) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        const int row = idx / F::CH, c = idx - row * F::CH;
        r.v[i] = (row < rows && c < dch) ? *reinterpret_cast<const u32x4_t*>(g + (long)row * ld + c * 8) : u32x4_t{0u, 0u, 0u, 0u};
    }
}
// The same for a tile whose 64 rows all exist (every tile but a ragged last one), for the operand widths whose chunk column is one
// per thread (CH divides the block: 64- and 128-wide): no row predicate, ONE column predicate for all of a thread's loads, addresses
// = wave-uniform tile base + per-thread 32-bit offsets fixed for the block.  Lanes whose chunk column is past the head dim load
// nothing and keep what their registers hold: zeros, or the ones of an augmented tile (set once).  The general form above costs, per
// tile and wave, ~16 register zeroings, a 64-bit address build per load and an EXEC region per load -- ~40 of ~160 vector
// instructions in loops that are bound by vector issue (3 waves per SIMD: 28 MFMAs x 8 + 160 x 4 issue cycles against 28 x 16 of
// matrix pipe per tile).
template <int DP> struct TileOffs { unsigned o[FA<DP>::CH * kTQ / kThreadsFA]; bool cok; };
template <int DP> constexpr bool kTileFast = kThreadsFA % FA<DP>::CH == 0;
template <int DP>
__device__ __forceinline__ void tile_offs(TileOffs<DP>& t, long ld, int dch, int tid) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        const int row = idx / F::CH, c = idx - row * F::CH;
        t.o[i] = (unsigned)(((long)row * ld + c * 8) * 2);
    }
    t.cok = tid % F::CH < dch;
}
template <int DP>
__device__ __forceinline__ void tile_load_full(const bf16_t* __restrict__ g, const TileOffs<DP>& t, TileRegs<DP>& r) {
    using F = FA<DP>;
    if (t.cok) {
#pragma unroll
        for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i)
            r.v[i] = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(g) + t.o[i]);
    }
}
template <int DP>
__device__ __forceinline__ void tile_zero(TileRegs<DP>& r) {
#pragma unroll
    for (int i = 0; i < FA<DP>::CH * kTQ / kThreadsFA; ++i) r.v[i] = u32x4_t{0u, 0u, 0u, 0u};
}

// skip_c: a chunk column somebody else writes (the augmented column of a Q / dO tile: dK / dV kernel), -1: none
template <int DP>
__device__ __forceinline__ void tile_store(const TileRegs<DP>& r, char* lds, int tid, int skip_c = -1) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        const int row = idx / F::CH, c = idx - row * F::CH;
        if (c != skip_c) *reinterpret_cast<u32x4_t*>(lds + F::off(row, c)) = r.v[i];
    }
}

// row-major fragment: rows sub*16 + (lane & 15), k = 32 * ks + 8 * (lane >> 4) .. + 7
template <int DP>
__device__ __forceinline__ bf16x8_t frag_rm(const char* lds, int sub, int ks, int lane) {
    using F = FA<DP>;
    return *reinterpret_cast<const bf16x8_t*>(lds + F::off(sub * 16 + (lane & 15), ks * 4 + (lane >> 4)));
}
// transposed fragment for the k-step j (rows = reduction index in the packed order, columns = d tile dt): lane (row' = lane & 15
// = column of the d tile, k-chunk g = lane >> 4) gets [rows (2j)*16 + 4g + 0..3 | rows (2j+1)*16 + 4g + 0..3][column dt*16 + row']
template <int DP>
__device__ __forceinline__ bf16x8_t frag_tr(const char* lds, int j, int dt, int lane) {
    using F = FA<DP>;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
    const int r0 = (2 * j) * 16 + 4 * g + q, r1 = r0 + 16;
    const int c = dt * 2 + (pp >> 1);                                   // logical 16-B chunk holding columns dt*16 + 4pp .. + 3
    const s16x4_t a0 = tr_read(lds + F::off(r0, c) + 8 * (pp & 1));
    const s16x4_t a1 = tr_read(lds + F::off(r1, c) + 8 * (pp & 1));
    return bf16x8_t{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
}
__device__ __forceinline__ bf16x8_t pack_pair(const f32x4_t& a, const f32x4_t& b) {
    const uint32_t w0 = pack_bf2(a[0], a[1]), w1 = pack_bf2(a[2], a[3]), w2 = pack_bf2(b[0], b[1]), w3 = pack_bf2(b[2], b[3]);
    return __builtin_bit_cast(bf16x8_t, u32x4_t{w0, w1, w2, w3});
}
__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32: exp2(-inf) = 0, no denormal fix-up
__device__ __forceinline__ float group_max(float v) {      // over the four 16-lane groups (same lane & 15)
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float group_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Operand addressing shared by the three kernels: tensor element (batch b, head h, row s, column d) lives at
// base + (b * S_rows + s) * ld + h * hoff + d;  merged layout: hoff = D, ld = the projection's row length;  head-split padded
// layout: the launcher passes H = 1, b = the (batch, head) index, ld = D_pad, hoff = 0.
struct FAShape {
    int H;                 // heads per batch entry (blockIdx.y = b * H + h)
    int hoff;              // column offset between heads
    int D;                 // valid head dim (a multiple of 8; chunks past it read as zero and are not stored)
    int Sq, Sk;            // rows that exist per batch entry (queries, keys)
    int Sqp, Skp;          // the same rounded up to 64 (loop bounds; LSE / delta rows per (batch, head))
    int valid_k;           // keys >= valid_k are masked
};
__device__ __forceinline__ bf16x8_t load_frag_or_zero(const bf16_t* p, bool ok) {
    return ok ? *reinterpret_cast<const bf16x8_t*>(p) : __builtin_bit_cast(bf16x8_t, u32x4_t{0u, 0u, 0u, 0u});
}

// ---- augmented contraction (backward kernels, head dims with padding: D < D_pad) ----------------------------------------------
// The per-element arithmetic of the backward is  p = exp2(s c - lse[q]),  dS = p (dP - delta[q]).  Both subtractions are rank-one
// updates of a product, and the contraction dimension is padded anyway (40 -> 64): with three spare columns of the Q operand set to
// a bf16 hi / mid / lo split of -lse[q] / c against ONES in the K operand, and two spare columns of the dO operand set to a split of
// -delta[q] against ones in V, the MFMAs deliver  s - lse / c  and  dP - delta  directly -- two vector instructions per score
// element gone from loops that are VALU-bound (and the per-tile lse / delta loads of the dK / dV kernel with them).  The products
// that read the same tiles transposed (dQ, dK, dV) get garbage only in output columns >= D, which are never stored.
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
constexpr uint32_t kOne2 = 0x3F803F80u, kOne1 = 0x00003F80u;       // bf16 (1, 1) / (1, 0)

// the tile's chunk `dch` of every row := `val` (the ones of an augmented K / V tile)
template <int DP>
__device__ __forceinline__ void tile_set_chunk(TileRegs<DP>& r, int dch, u32x4_t val, int tid) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        if (idx - (idx / F::CH) * F::CH == dch) r.v[i] = val;
    }
}
// the tile's chunk `dch` of row r := split(v[r] * mul) (the -lse / c or -delta columns of an augmented Q / dO tile), in two steps: the
// row values are LOADED with the tile (rowvals_load: the loads join the prefetch in flight under the current tile's products) and
// converted / placed only when the tile goes to LDS (rowvals_apply) -- converting at load time made every wave wait for its whole
// prefetch before the current tile's MFMAs (vmcnt retires in order): 5.6 -> 5.8 ms instead of 4.8
template <int DP> struct RowVals { float v[FA<DP>::CH * kTQ / kThreadsFA]; };
template <int DP>
__device__ __forceinline__ void rowvals_load(RowVals<DP>& rv, int dch, const float* __restrict__ vals, int tid) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        const int row = idx / F::CH;
        rv.v[i] = (idx - row * F::CH == dch) ? vals[row] : 0.f;
    }
}
template <int DP>
__device__ __forceinline__ void rowvals_apply(TileRegs<DP>& r, const RowVals<DP>& rv, int dch, float mul, int nparts, int tid) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        if (idx - (idx / F::CH) * F::CH == dch) r.v[i] = split_bf16(rv.v[i] * mul, nparts);
    }
}

// ====================================================================================================================
// forward: O = softmax(scale Q K^T) V, LSE2[q] = log2 sum_k exp2(scale log2e (q.k))   (base-2 log-sum-exp)
// grid (Sq_pad / 64, B*heads)
// ====================================================================================================================
// DL: the head dimension covers only the first DL 16-wide d tiles of the DP-wide operands (0: all of them) -- the output tiles and
// the 32-deep contraction steps past it are all padding and are not computed (D = 40 in 64: 3 of 4 tiles; 80 in 128: 5 of 8 tiles
// and 3 of 4 steps; 160 in 192: 10 of 12 and 5 of 6).
template <int DP, int DL>
__global__ __launch_bounds__(kThreadsFA) void flash_fwd_kernel(const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K,
                                                               long ldk, const bf16_t* __restrict__ V, long ldv,
                                                               bf16_t* __restrict__ O, long ldo, float* __restrict__ LSE2,
                                                               FAShape sh, float scale_log2) {
    using F = FA<DP>;
    constexpr int DTL = DL ? DL : F::DT, KSL = DL ? (DL * 16 + 31) / 32 : F::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks_ = smem;                       // K tile
    char* vs_ = smem + F::TILE;             // V tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long bh = blockIdx.y;
    const long b = bh / sh.H;
    const int h = (int)(bh - b * sh.H);
    const int dch = sh.D >> 3, valid_k = sh.valid_k, Skp = sh.Skp;
    const int q0 = blockIdx.x * kTQ + w * 16;
    const int qrow = q0 + (lane & 15);
    const bool q_ok = qrow < sh.Sq;
    const bf16_t* qg = Q + (b * sh.Sq + qrow) * ldq + h * sh.hoff + (lane >> 4) * 8;
    bf16x8_t qf[KSL];
#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) qf[ks] = load_frag_or_zero(qg + ks * 32, q_ok && ks * 4 + (lane >> 4) < dch);
    f32x4_t ot[DTL];
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt) ot[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    // Row sums on the matrix pipe: when the head dim ends in the middle of a d tile (D % 16 == 8: SD's 40) that tile has an all-padding
    // column at index D; a ones column there in V makes O^T[D][q] accumulate sum_k p[q][k] -- with the same rescaling as O -- and the
    // 16 adds + the cross-group sum per tile leave these VALU-bound loops.  (The sum is then over the bf16-rounded p the product uses.)
    const bool lcol = (sh.D & 15) == 8 && sh.D < DTL * 16;
    const bf16_t* kg = K + b * sh.Sk * ldk + h * sh.hoff;
    const bf16_t* vg = V + b * sh.Sk * ldv + h * sh.hoff;
    TileRegs<DP> kr, vr;
    TileOffs<DP> ko, vo;
    tile_offs<DP>(ko, ldk, dch, tid); tile_offs<DP>(vo, ldv, dch, tid);
    tile_zero<DP>(kr); tile_zero<DP>(vr);
    // tile at key row k0n -> registers (full tiles: tile_load_full; a ragged last tile: the general form, which also rewrites the pad
    // chunks, so the ones column goes back in)
    auto fetch = [&](int k0n) {
        if (kTileFast<DP> && k0n + kTQ <= sh.Sk) {
            tile_load_full<DP>(kg + (long)k0n * ldk, ko, kr);
            tile_load_full<DP>(vg + (long)k0n * ldv, vo, vr);
        } else {
            tile_load<DP>(kg + (long)k0n * ldk, ldk, sh.Sk - k0n, dch, kr, tid);
            tile_load<DP>(vg + (long)k0n * ldv, ldv, sh.Sk - k0n, dch, vr, tid);
            if (lcol) tile_set_chunk<DP>(vr, dch, u32x4_t{kOne1, 0u, 0u, 0u}, tid);
        }
    };
    if (lcol) tile_set_chunk<DP>(vr, dch, u32x4_t{kOne1, 0u, 0u, 0u}, tid);
    fetch(0);
    for (int k0 = 0; k0 < Skp; k0 += kTQ) {
        __syncthreads();                                        // the previous tile's readers are done
        tile_store<DP>(kr, ks_, tid);
        tile_store<DP>(vr, vs_, tid);
        __syncthreads();
        if (k0 + kTQ < Skp) fetch(k0 + kTQ);                    // next tile: in flight under this tile's products
        f32x4_t st[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            st[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks)
                st[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(ks_, sub, ks, lane), qf[ks], st[sub], 0, 0, 0);
        }
        // The running maximum is kept on the RAW scores (scale > 0: the maximum commutes with the scaling), so that an element
        // costs one fma + one exp2 (p = exp2(s * c - m * c)) instead of a multiply, a subtract and the exp2; the key mask (two more
        // instructions per element) only exists in tiles that contain padded keys (cross attention's last tile) -- a wave-uniform
        // branch.  These loops are VALU-bound at head dims <= 64 (2.4 vector issue slots per MFMA slot before this diet).
        if (k0 + kTQ > valid_k) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + sub * 16 + (lane >> 4) * 4 + r >= valid_k) st[sub][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(st[0][0], st[0][1]), fmaxf(st[0][2], st[0][3]));
#pragma unroll
        for (int sub = 1; sub < 4; ++sub) mx = fmaxf(mx, fmaxf(fmaxf(st[sub][0], st[sub][1]), fmaxf(st[sub][2], st[sub][3])));
        const float m_new = fmaxf(m, group_max(mx));            // finite: the first tile always holds a valid key (valid_k >= 1)
        const float mc = m_new * scale_log2;
        // no row of this wave has a new maximum (most tiles after the first few): nothing to rescale -- a wave-uniform branch
        const bool moved = __builtin_amdgcn_ballot_w64(m_new != m) != 0;
        if (moved) {
            const float alpha = fast_exp2((m - m_new) * scale_log2);    // m = -inf on the first tile: alpha = 0
            l *= alpha;
#pragma unroll
            for (int dt = 0; dt < DTL; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
            m = m_new;
        }
        if (lcol) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) st[sub][r] = fast_exp2(fmaf(st[sub][r], scale_log2, -mc));
        } else {
            float ps = 0.f;
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float p = fast_exp2(fmaf(st[sub][r], scale_log2, -mc)); st[sub][r] = p; ps += p; }
            l += group_sum(ps);
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8_t pf = pack_pair(st[2 * j], st[2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < DTL; ++dt)
                ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(vs_, j, dt, lane), pf, ot[dt], 0, 0, 0);
        }
    }
    if (lcol) {                                                 // O^T[D][q]: d tile D / 16, row 8 = lanes 32..47, r = 0
        float lv = 0.f;
#pragma unroll
        for (int dt = 0; dt < DTL; ++dt) lv = dt == (sh.D >> 4) ? ot[dt][0] : lv;
        l = __shfl(lv, 32 + (lane & 15), 64);
    }
    const float inv = 1.f / l;
    bf16_t* og = O + (b * sh.Sq + qrow) * ldo + h * sh.hoff + (lane >> 4) * 4;
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt)
        if (q_ok && dt * 16 + (lane >> 4) * 4 < sh.D)
            *reinterpret_cast<u32x2_t*>(og + dt * 16) = u32x2_t{pack_bf2(ot[dt][0] * inv, ot[dt][1] * inv), pack_bf2(ot[dt][2] * inv, ot[dt][3] * inv)};
    if ((lane >> 4) == 0) LSE2[bh * sh.Sqp + q0 + lane] = m * scale_log2 + log2f(l);
}

// ====================================================================================================================
// backward, dQ:  dQ = scale * dS K,  dS = P o (dO V^T - delta)        grid (Sq_pad / 64, nB*heads)
// z = cotangent (batch, head) index; the forward tensors (Q, K, V, O, LSE2) are indexed by (batch % Bf, head) (dual-cotangent
// backward).  Og != null: delta[q] = <dO[q], O[q]> is formed here, from fragments laid out like Q's, and WRITTEN to `delta`
// for the dK / dV kernel that follows; Og == null: `delta` is an input.
// ====================================================================================================================
// PRE: q already holds scale * log2(e) * Q (siss_flash_attn_*_merged with q_prescaled): the score needs no multiply before its exp2.
template <int DP, bool AUG, int DL, bool PRE>
__global__ __launch_bounds__(kThreadsFA) void flash_bwd_dq_kernel(const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K,
                                                                  long ldk, const bf16_t* __restrict__ V, long ldv,
                                                                  const bf16_t* __restrict__ Og, long ldo,
                                                                  const bf16_t* __restrict__ dO, long lddo,
                                                                  const float* __restrict__ LSE2, float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dQ, long lddq, int Bf, FAShape sh,
                                                                  float scale, float scale_log2) {
    using F = FA<DP>;
    constexpr int DTL = DL ? DL : F::DT, KSL = DL ? (DL * 16 + (AUG ? 8 : 0) + 31) / 32 : F::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks_ = smem;
    char* vs_ = smem + F::TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long z = blockIdx.y;
    const long bz = z / sh.H;
    const int h = (int)(z - bz * sh.H);
    const long bf = bz % Bf, zf = bf * sh.H + h;
    const int dch = sh.D >> 3, valid_k = sh.valid_k, Skp = sh.Skp;
    const int q0 = blockIdx.x * kTQ + w * 16;
    const int qrow = q0 + (lane & 15);
    const bool q_ok = qrow < sh.Sq;
    const long col = h * sh.hoff + (lane >> 4) * 8;
    bf16x8_t qf[KSL], dof[KSL];
    float dl = 0.f;
#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) {
        const bool ok = q_ok && ks * 4 + (lane >> 4) < dch;
        qf[ks] = load_frag_or_zero(Q + (bf * sh.Sq + qrow) * ldq + col + ks * 32, ok);
        dof[ks] = load_frag_or_zero(dO + (bz * sh.Sq + qrow) * lddo + col + ks * 32, ok);
        if (Og) {
            const bf16x8_t of = load_frag_or_zero(Og + (bf * sh.Sq + qrow) * ldo + col + ks * 32, ok);
            const u32x4_t a = __builtin_bit_cast(u32x4_t, dof[ks]), c = __builtin_bit_cast(u32x4_t, of);
#pragma unroll
            for (int e = 0; e < 4; ++e)
                dl += __builtin_bit_cast(float, a[e] << 16) * __builtin_bit_cast(float, c[e] << 16) +
                      __builtin_bit_cast(float, a[e] & 0xffff0000u) * __builtin_bit_cast(float, c[e] & 0xffff0000u);
        }
    }
    if (Og) {
        dl = group_sum(dl);                                     // the four 16-lane groups hold the four k-chunks of a row
        if ((lane >> 4) == 0) delta[z * sh.Sqp + q0 + lane] = dl;
    } else {
        dl = delta[z * sh.Sqp + qrow];
    }
    const float lse = LSE2[zf * sh.Sqp + qrow];
    if constexpr (AUG) {
        // this lane's row of the Q / dO operands: the pad chunk (columns D .. D+7) carries -lse / c and -delta, split into bf16 parts
        const u32x4_t lq = split_bf16(-lse / scale_log2, 3), ld = split_bf16(-dl, 2);
#pragma unroll
        for (int ks = 0; ks < KSL; ++ks)
            if (ks * 4 + (lane >> 4) == dch) { qf[ks] = __builtin_bit_cast(bf16x8_t, lq); dof[ks] = __builtin_bit_cast(bf16x8_t, ld); }
    }
    f32x4_t dqt[DTL];
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt) dqt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16_t* kg = K + bf * sh.Sk * ldk + h * sh.hoff;
    const bf16_t* vg = V + bf * sh.Sk * ldv + h * sh.hoff;
    TileRegs<DP> kr, vr;
    TileOffs<DP> ko, vo;
    tile_offs<DP>(ko, ldk, dch, tid); tile_offs<DP>(vo, ldv, dch, tid);
    tile_zero<DP>(kr); tile_zero<DP>(vr);
    auto ones_cols = [&]() {
        if constexpr (AUG) { tile_set_chunk<DP>(kr, dch, u32x4_t{kOne2, kOne1, 0u, 0u}, tid); tile_set_chunk<DP>(vr, dch, u32x4_t{kOne2, 0u, 0u, 0u}, tid); }
    };
    // (the ones columns sit in registers no full-tile load touches: set once, and again after a general-form load)
    auto fetch = [&](int k0n) {
        if (kTileFast<DP> && k0n + kTQ <= sh.Sk) {
            tile_load_full<DP>(kg + (long)k0n * ldk, ko, kr);
            tile_load_full<DP>(vg + (long)k0n * ldv, vo, vr);
        } else {
            tile_load<DP>(kg + (long)k0n * ldk, ldk, sh.Sk - k0n, dch, kr, tid);
            tile_load<DP>(vg + (long)k0n * ldv, ldv, sh.Sk - k0n, dch, vr, tid);
            ones_cols();
        }
    };
    ones_cols();
    fetch(0);
    for (int k0 = 0; k0 < Skp; k0 += kTQ) {
        __syncthreads();
        tile_store<DP>(kr, ks_, tid);
        tile_store<DP>(vr, vs_, tid);
        __syncthreads();
        if (k0 + kTQ < Skp) fetch(k0 + kTQ);
        f32x4_t st[4], dp[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            st[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) {
                st[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(ks_, sub, ks, lane), qf[ks], st[sub], 0, 0, 0);
                dp[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(vs_, sub, ks, lane), dof[ks], dp[sub], 0, 0, 0);
            }
        }
        // dS without its factor `scale` (applied once to the finished dQ tile); the key mask only in tiles with padded keys
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if constexpr (AUG) st[sub][r] = fast_exp2(PRE ? st[sub][r] : st[sub][r] * scale_log2) * dp[sub][r];        // (the MFMAs subtracted lse / c and delta)
                else st[sub][r] = fast_exp2(fmaf(st[sub][r], scale_log2, -lse)) * (dp[sub][r] - dl);
            }
        if (k0 + kTQ > valid_k) {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (k0 + sub * 16 + (lane >> 4) * 4 + r >= valid_k) st[sub][r] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8_t dsf = pack_pair(st[2 * j], st[2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < DTL; ++dt)
                dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(ks_, j, dt, lane), dsf, dqt[dt], 0, 0, 0);
        }
    }
    bf16_t* og = dQ + (bz * sh.Sq + qrow) * lddq + h * sh.hoff + (lane >> 4) * 4;
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt)
        if (q_ok && dt * 16 + (lane >> 4) * 4 < sh.D)
            *reinterpret_cast<u32x2_t*>(og + dt * 16) = u32x2_t{pack_bf2(dqt[dt][0] * scale, dqt[dt][1] * scale), pack_bf2(dqt[dt][2] * scale, dqt[dt][3] * scale)};
}

// ====================================================================================================================
// backward, dK / dV:  dV = P^T dO,  dK = scale * dS^T Q             grid (Sk_pad / 64, nB*heads)
// S[q][key] = mfma(Q rows, K rows): a lane keeps ONE key and 4 consecutive queries of each 16-query sub-tile.
// ====================================================================================================================
template <int DP, bool AUG, int DL, bool PRE>
__global__ __launch_bounds__(kThreadsFA) void flash_bwd_dkdv_kernel(const bf16_t* __restrict__ Q, long ldq, const bf16_t* __restrict__ K,
                                                                    long ldk, const bf16_t* __restrict__ V, long ldv,
                                                                    const bf16_t* __restrict__ dO, long lddo,
                                                                    const float* __restrict__ LSE2, const float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dK, long lddk, bf16_t* __restrict__ dV,
                                                                    long lddv, int Bf, FAShape sh, float kscale, float scale_log2,
                                                                    int qchunk, float* __restrict__ part) {
    using F = FA<DP>;
    constexpr int DTL = DL ? DL : F::DT, KSL = DL ? (DL * 16 + (AUG ? 8 : 0) + 31) / 32 : F::KS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long z = blockIdx.y;
    const long bz = z / sh.H;
    const int h = (int)(z - bz * sh.H);
    const long bf = bz % Bf, zf = bf * sh.H + h;
    const int dch = sh.D >> 3, Sqp = sh.Sqp;
    const int k0 = blockIdx.x * kTQ + w * 16;
    const int krow = k0 + (lane & 15);
    const bool k_ok = krow < sh.Sk;
    const float key_lse_off = krow < sh.valid_k ? 0.f : -INFINITY;     // a lane keeps ONE key: its mask is one additive constant
    const long col = h * sh.hoff + (lane >> 4) * 8;
    bf16x8_t kf[KSL], vf[KSL];
#pragma unroll
    for (int ks = 0; ks < KSL; ++ks) {
        const bool ok = k_ok && ks * 4 + (lane >> 4) < dch;
        kf[ks] = load_frag_or_zero(K + (bf * sh.Sk + krow) * ldk + col + ks * 32, ok);
        vf[ks] = load_frag_or_zero(V + (bf * sh.Sk + krow) * ldv + col + ks * 32, ok);
    }
    if constexpr (AUG) {
        // this lane's key row: ones against the -lse / c (three) and -delta (two) columns of the augmented Q / dO tiles
#pragma unroll
        for (int ks = 0; ks < KSL; ++ks)
            if (ks * 4 + (lane >> 4) == dch) {
                kf[ks] = __builtin_bit_cast(bf16x8_t, u32x4_t{kOne2, kOne1, 0u, 0u});
                vf[ks] = __builtin_bit_cast(bf16x8_t, u32x4_t{kOne2, 0u, 0u, 0u});
            }
    }
    const float key_mask = krow < sh.valid_k ? 1.f : 0.f;
    const bool any_masked = blockIdx.x * kTQ + kTQ > sh.valid_k;       // (block-uniform) this block holds padded keys
    f32x4_t dkt[DTL], dvt[DTL];
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt) { dkt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const bf16_t* qg = Q + bf * sh.Sq * ldq + h * sh.hoff;
    const bf16_t* dog = dO + bz * sh.Sq * lddo + h * sh.hoff;
    const float* lseg = LSE2 + zf * Sqp;
    const float* dlg = delta + z * Sqp;
    const float lmul = -1.f / scale_log2;
    TileRegs<DP> qr, dor;
    // Augmented columns of the Q / dO tiles: ONE lane per row -- wave 0 splits -lse / c of the tile's 64 queries into the Q tile's pad
    // chunk, wave 1 -delta into the dO tile's -- straight to LDS beside the tile stores (which skip that chunk).  (Patching the
    // prefetch registers instead ran the split on 8 live lanes of EVERY wave, twice per tile and operand: ~100 of the ~280 vector
    // instructions of a tile in loops that are bound by vector issue.)  The row value is loaded with the tile's prefetch.
    float aug_cur = 0.f, aug_next = 0.f;
    // query range of this block: all of them, or (few keys: cross attention) the blockIdx.z-th chunk of `qchunk` rows, whose
    // partial dK / dV go to `part` in f32 for flash_dkdv_reduce_kernel
    const int qb0 = part ? blockIdx.z * qchunk : 0;
    const int qb1 = part ? (qb0 + qchunk < Sqp ? qb0 + qchunk : Sqp) : Sqp;
    TileOffs<DP> qo, doo;
    tile_offs<DP>(qo, ldq, dch, tid); tile_offs<DP>(doo, lddo, dch, tid);
    tile_zero<DP>(qr); tile_zero<DP>(dor);
    auto fetch = [&](int q0n) {
        if (kTileFast<DP> && q0n + kTQ <= sh.Sq) {
            tile_load_full<DP>(qg + (long)q0n * ldq, qo, qr);
            tile_load_full<DP>(dog + (long)q0n * lddo, doo, dor);
        } else {
            tile_load<DP>(qg + (long)q0n * ldq, ldq, sh.Sq - q0n, dch, qr, tid);
            tile_load<DP>(dog + (long)q0n * lddo, lddo, sh.Sq - q0n, dch, dor, tid);
        }
    };
    fetch(qb0);
    if constexpr (AUG) { if (w == 0) aug_next = lseg[qb0 + lane]; else if (w == 1) aug_next = dlg[qb0 + lane]; }
    // registers -> LDS (+ the augmented columns); `qs_` / `dos_` are the buffers written
    auto to_lds = [&](char* qs_, char* dos_) {
        tile_store<DP>(qr, qs_, tid, AUG ? dch : -1);
        tile_store<DP>(dor, dos_, tid, AUG ? dch : -1);
        if constexpr (AUG) {
            aug_cur = aug_next;
            if (w == 0) *reinterpret_cast<u32x4_t*>(qs_ + F::off(lane, dch)) = split_bf16(aug_cur * lmul, 3);
            else if (w == 1) *reinterpret_cast<u32x4_t*>(dos_ + F::off(lane, dch)) = split_bf16(-aug_cur, 2);
        }
    };
    auto prefetch = [&](int q0n) {
        fetch(q0n);
        if constexpr (AUG) { if (w == 0) aug_next = lseg[q0n + lane]; else if (w == 1) aug_next = dlg[q0n + lane]; }
    };
    // DB (64-wide operands: 4 x 8 KiB of LDS): TWO buffers per operand -- tile t + 1 goes to LDS before tile t's products, ONE barrier
    // per tile (it orders both "everybody is done reading tile t" and "tile t + 1 is visible"); the wider tiles keep one buffer and
    // two barriers (four 16-32 KiB tiles per block would cost a resident block).
    constexpr bool DB = DP == 64;
    char* const qs0 = smem; char* const dos0 = smem + F::TILE;
    if constexpr (DB) {
        to_lds(qs0, dos0);
        if (qb0 + kTQ < qb1) prefetch(qb0 + kTQ);
        __syncthreads();
    }
    int it = 0;
    for (int q0 = qb0; q0 < qb1; q0 += kTQ, ++it) {
        char* qs_ = qs0; char* dos_ = dos0;
        if constexpr (DB) {
            qs_ = qs0 + (it & 1) * 2 * F::TILE; dos_ = dos0 + (it & 1) * 2 * F::TILE;
            if (q0 + kTQ < qb1) {
                to_lds(qs0 + ((it + 1) & 1) * 2 * F::TILE, dos0 + ((it + 1) & 1) * 2 * F::TILE);
                if (q0 + 2 * kTQ < qb1) prefetch(q0 + 2 * kTQ);
            }
        } else {
            __syncthreads();
            to_lds(qs_, dos_);
            __syncthreads();
            if (q0 + kTQ < qb1) prefetch(q0 + kTQ);
        }
        f32x4_t s[4], dp[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            s[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < KSL; ++ks) {
                s[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(qs_, sub, ks, lane), kf[ks], s[sub], 0, 0, 0);
                dp[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(dos_, sub, ks, lane), vf[ks], dp[sub], 0, 0, 0);
            }
        }
        f32x4_t ds[4];
        if constexpr (AUG) {
            // s and dp arrive with lse / c and delta already subtracted (augmented contraction).  A padded QUERY row (past Sq) has
            // q = dO = 0 in LDS but carries its lse / delta columns: its P is finite garbage that only ever multiplies those zero
            // rows in the dV / dK products.  Padded KEYS (blocks past valid_k only): one more multiply by the lane's 0 / 1 mask.
#pragma unroll
            for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float p = fast_exp2(PRE ? s[sub][r] : s[sub][r] * scale_log2);
                    s[sub][r] = p;
                    ds[sub][r] = p * dp[sub][r];                                                    // (scale: once, on the finished dK tile)
                }
            if (any_masked) {
#pragma unroll
                for (int sub = 0; sub < 4; ++sub)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[sub][r] *= key_mask; ds[sub][r] *= key_mask; }
            }
        } else {
#pragma unroll
            for (int sub = 0; sub < 4; ++sub) {
                const int qb = q0 + sub * 16 + (lane >> 4) * 4;
                const f32x4_t lse = *reinterpret_cast<const f32x4_t*>(lseg + qb);
                const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(dlg + qb);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // padded key: exp2(-inf) = 0.  A padded QUERY row (past Sq) has q = dO = 0 in LDS: its P is finite garbage, but
                    // it only ever multiplies those zero rows in the dV / dK products
                    const float p = fast_exp2(fmaf(s[sub][r], scale_log2, key_lse_off - lse[r]));
                    s[sub][r] = p;
                    ds[sub][r] = p * (dp[sub][r] - dl[r]);                                          // (scale: once, on the finished dK tile)
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8_t pf = pack_pair(s[2 * j], s[2 * j + 1]);
            const bf16x8_t dsf = pack_pair(ds[2 * j], ds[2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < DTL; ++dt) {
                dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(dos_, j, dt, lane), pf, dvt[dt], 0, 0, 0);
                dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(qs_, j, dt, lane), dsf, dkt[dt], 0, 0, 0);
            }
        }
        if constexpr (DB) __syncthreads();
    }
    if (part) {
        // [z][chunk][key][dK: DTL * 16 | dV: DTL * 16] f32, unscaled
        float* pg = part + (((long)z * gridDim.z + blockIdx.z) * sh.Skp + krow) * (2 * DTL * 16) + (lane >> 4) * 4;
#pragma unroll
        for (int dt = 0; dt < DTL; ++dt) {
            *reinterpret_cast<f32x4_t*>(pg + dt * 16) = dkt[dt];
            *reinterpret_cast<f32x4_t*>(pg + DTL * 16 + dt * 16) = dvt[dt];
        }
        return;
    }
    bf16_t* okg = dK + (bz * sh.Sk + krow) * lddk + h * sh.hoff + (lane >> 4) * 4;
    bf16_t* ovg = dV + (bz * sh.Sk + krow) * lddv + h * sh.hoff + (lane >> 4) * 4;
#pragma unroll
    for (int dt = 0; dt < DTL; ++dt) {
        if (!(k_ok && dt * 16 + (lane >> 4) * 4 < sh.D)) continue;
        *reinterpret_cast<u32x2_t*>(okg + dt * 16) = u32x2_t{pack_bf2(dkt[dt][0] * kscale, dkt[dt][1] * kscale), pack_bf2(dkt[dt][2] * kscale, dkt[dt][3] * kscale)};
        *reinterpret_cast<u32x2_t*>(ovg + dt * 16) = u32x2_t{pack_bf2(dvt[dt][0], dvt[dt][1]), pack_bf2(dvt[dt][2], dvt[dt][3])};
    }
}

// Sum of the query chunks' partial dK / dV (flash_bwd_dkdv_kernel with `part`): one thread per (z, key, 4 head-dim columns).
__global__ __launch_bounds__(256) void flash_dkdv_reduce_kernel(const float* __restrict__ part, int nchunks, int DW,
                                                               bf16_t* __restrict__ dK, long lddk, bf16_t* __restrict__ dV, long lddv,
                                                               FAShape sh, float scale, long total) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = sh.D >> 2;
    const int col = (int)(i % c4) * 4;
    const long r = i / c4;
    const int key = (int)(r % sh.Sk);
    const long z = r / sh.Sk;
    const long bz = z / sh.H;
    const int h = (int)(z - bz * sh.H);
    f32x4_t k = f32x4_t{0.f, 0.f, 0.f, 0.f}, v = f32x4_t{0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < nchunks; ++c) {
        const float* pg = part + ((z * nchunks + c) * sh.Skp + key) * (2 * DW) + col;
        const f32x4_t a = *reinterpret_cast<const f32x4_t*>(pg), b = *reinterpret_cast<const f32x4_t*>(pg + DW);
#pragma unroll
        for (int e = 0; e < 4; ++e) { k[e] += a[e]; v[e] += b[e]; }
    }
    const long o = (bz * sh.Sk + key);
    *reinterpret_cast<u32x2_t*>(dK + o * lddk + h * sh.hoff + col) = u32x2_t{pack_bf2(k[0] * scale, k[1] * scale), pack_bf2(k[2] * scale, k[3] * scale)};
    *reinterpret_cast<u32x2_t*>(dV + o * lddv + h * sh.hoff + col) = u32x2_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3])};
}

inline int fa_dpad(int D) { return D <= 64 ? 64 : (D <= 128 ? 128 : (D <= 192 ? 192 : 0)); }

struct FwdArgs { const void *q, *k, *v; void* o; long ldq, ldk, ldv, ldo; float* lse2; };
int fa_launch_fwd(const FwdArgs& a, int nbh, const FAShape& sh, float scale, bool pre, void* stream) {
    const dim3 grid(sh.Sqp / kTQ, nbh);
    const float sl2 = pre ? 1.f : scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
#define FA_FWD(DP, DL)                                                                                                   \
    do {                                                                                                                  \
        static unsigned char att[kMaxDevices];                                                                            \
        if (siss_ensure_smem((const void*)flash_fwd_kernel<DP, DL>, 2 * FA<DP>::TILE, att) != SISS_OK) return SISS_ERR_LAUNCH; \
        flash_fwd_kernel<DP, DL><<<grid, kThreadsFA, 2 * FA<DP>::TILE, st>>>((const bf16_t*)a.q, a.ldq, (const bf16_t*)a.k, a.ldk, \
            (const bf16_t*)a.v, a.ldv, (bf16_t*)a.o, a.ldo, a.lse2, sh, sl2);                                              \
    } while (0)
    const int dp = fa_dpad(sh.D);
    // (the live-tile counts instantiated: SD v1.5's head dims 40 / 80 / 160 and anything narrower; everything else runs all tiles)
    if (dp == 64) { if (sh.D <= 48) FA_FWD(64, 3); else FA_FWD(64, 0); }
    else if (dp == 128) { if (sh.D <= 80) FA_FWD(128, 5); else FA_FWD(128, 0); }
    else { if (sh.D <= 160) FA_FWD(192, 10); else FA_FWD(192, 0); }
#undef FA_FWD
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

struct BwdArgs { const void *q, *k, *v, *o, *d_o; void *dq, *dk, *dv; long ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
                 const float* lse2; float* delta; };
// Few key tiles (cross attention: 77 keys = 2 tiles) leave the dK / dV grid at a fraction of the chip -- 128 blocks of 64 query
// tiles each at B = 4.  Then the queries are cut into chunks (grid z), every block leaves its partial tiles in the library
// workspace (siss_gemm_nt_set_workspace; launches on one stream at a time share it) and a small kernel sums them: deterministic,
// no atomics.  Chunks: as many as bring the grid to ~1024 blocks, at least 4 query tiles each, within the workspace.
void fa_qsplit(const FAShape& sh, int nbh, int DW, int& nch, int& qchunk, float*& ws) {
    long bytes = 0;
    ws = (float*)siss_workspace(&bytes);
    const long base = (long)(sh.Skp / kTQ) * nbh;
    const int qtiles = sh.Sqp / kTQ;
    if (!ws || base >= 512 || qtiles < 8 || sh.Sk != sh.valid_k || sh.D % 4) return;
    long n = (1024 + base - 1) / base;
    if (n > qtiles / 4) n = qtiles / 4;
    const long per_chunk = (long)nbh * sh.Skp * 2 * DW * (long)sizeof(float);
    if (n * per_chunk > bytes - 4096) n = (bytes - 4096) / per_chunk;
    if (n < 2) return;
    const int tiles_per = (int)((qtiles + n - 1) / n);
    nch = (qtiles + tiles_per - 1) / tiles_per;
    qchunk = tiles_per * kTQ;
}

// pre: q holds scale * log2(e) * Q.  Scores then are base-2 logits as they come (sl2 = 1; only the augmented kernels have a
// multiply to drop, hence their PRE forms), dQ is still the gradient with respect to the UNSCALED query -- scale * dS K as before --
// and dK = scale * dS^T Q = ln(2) * dS^T q.
int fa_launch_bwd(const BwdArgs& a, int nbh, int Bf, const FAShape& sh, float scale, bool pre, void* stream) {
    const float sl2 = pre ? 1.f : scale * 1.4426950408889634f;
    const float kscale = pre ? 0.6931471805599453f : scale;
    hipStream_t st = (hipStream_t)stream;
    float* ws = nullptr;
    // dQ first: it forms delta = rowsum(dO o O) for its 64 queries (when O is given) and leaves it for the dK / dV kernel
#define FA_BWD(DP, AUG, DL, PRE)                                                                                                   \
    do {                                                                                                                     \
        static unsigned char a1[kMaxDevices], a2[kMaxDevices];                                                               \
        if (siss_ensure_smem((const void*)flash_bwd_dq_kernel<DP, AUG, DL, PRE>, 2 * FA<DP>::TILE, a1) != SISS_OK) return SISS_ERR_LAUNCH; \
        constexpr int smem_kv = (DP == 64 ? 4 : 2) * FA<DP>::TILE;                                                           \
        if (siss_ensure_smem((const void*)flash_bwd_dkdv_kernel<DP, AUG, DL, PRE>, smem_kv, a2) != SISS_OK) return SISS_ERR_LAUNCH; \
        flash_bwd_dq_kernel<DP, AUG, DL, PRE><<<dim3(sh.Sqp / kTQ, nbh), kThreadsFA, 2 * FA<DP>::TILE, st>>>(                       \
            (const bf16_t*)a.q, a.ldq, (const bf16_t*)a.k, a.ldk, (const bf16_t*)a.v, a.ldv, (const bf16_t*)a.o, a.ldo,     \
            (const bf16_t*)a.d_o, a.lddo, a.lse2, a.delta, (bf16_t*)a.dq, a.lddq, Bf, sh, scale, sl2);                      \
        constexpr int DW = (DL ? DL : FA<DP>::DT) * 16;                                                                     \
        int nch = 1, qchunk = sh.Sqp;                                                                                        \
        fa_qsplit(sh, nbh, DW, nch, qchunk, ws);                                                                             \
        flash_bwd_dkdv_kernel<DP, AUG, DL, PRE><<<dim3(sh.Skp / kTQ, nbh, nch), kThreadsFA, smem_kv, st>>>(                     \
            (const bf16_t*)a.q, a.ldq, (const bf16_t*)a.k, a.ldk, (const bf16_t*)a.v, a.ldv, (const bf16_t*)a.d_o, a.lddo,  \
            a.lse2, a.delta, (bf16_t*)a.dk, a.lddk, (bf16_t*)a.dv, a.lddv, Bf, sh, kscale, sl2, qchunk, nch > 1 ? ws : nullptr); \
        if (nch > 1) {                                                                                                       \
            siss_count_dispatch(SISS_K_FLASH_QSPLIT);                                                                        \
            const long total = (long)nbh * sh.Sk * (sh.D >> 2);                                                              \
            flash_dkdv_reduce_kernel<<<dim3((unsigned)((total + 255) / 256)), 256, 0, st>>>(ws, nch, DW, (bf16_t*)a.dk, a.lddk, \
                (bf16_t*)a.dv, a.lddv, sh, kscale, total);                                                                    \
        }                                                                                                                    \
    } while (0)
    const int dp = fa_dpad(sh.D);
    // A head dim of at most 56 (one whole pad chunk in the 64-wide operands: SD's D = 40, the 4096-key sites) takes the
    // augmented-contraction kernels: 5.58 -> 4.98 ms per launch at B = 16.  The wider tiles do NOT: there the loops are less
    // VALU-bound and the extra registers cost a resident wave (dK / dV kernel at D_pad = 128: 232 -> 256 VGPRs, 772 -> 1069 us).
    if (dp == 64 && sh.D <= 48) { if (pre) FA_BWD(64, true, 3, true); else FA_BWD(64, true, 3, false); }
    else if (dp == 64 && sh.D + 8 <= dp) { if (pre) FA_BWD(64, true, 0, true); else FA_BWD(64, true, 0, false); }
    else if (dp == 64) FA_BWD(64, false, 0, false);
    else if (dp == 128) { if (sh.D <= 80) FA_BWD(128, false, 5, false); else FA_BWD(128, false, 0, false); }
    else { if (sh.D <= 160) FA_BWD(192, false, 10, false); else FA_BWD(192, false, 0, false); }
#undef FA_BWD
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

bool fa_shape_ok(int Sqp, int Skp, int Dp, int valid_k) {
    return Sqp > 0 && Skp > 0 && Sqp % kTQ == 0 && Skp % kTQ == 0 && (Dp == 64 || Dp == 128 || Dp == 192) && valid_k > 0 &&
           valid_k <= Skp;
}
inline int up64(int v) { return (v + 63) / 64 * 64; }

}  // namespace

extern "C" {

// O = softmax(scale * Q K^T) V over the first valid_k keys, per (batch, head): Q / O [BH][Sq_pad][D_pad], K / V [BH][Sk_pad][D_pad]
// bf16 (zero padded; Sq_pad, Sk_pad multiples of 64; D_pad 64, 128 or 192).  lse2 [BH][Sq_pad] f32 receives the base-2
// log-sum-exp of the scaled scores (what siss_flash_attn_bwd needs).  Nothing of size S x S is written.
int siss_flash_attn_fwd(const void* q, const void* k, const void* v, void* o, float* lse2, int BH, int Sq_pad, int Sk_pad,
                        int D_pad, int valid_k, float scale, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && lse2 && BH > 0 && fa_shape_ok(Sq_pad, Sk_pad, D_pad, valid_k));
    SISS_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0);
    siss_count_dispatch(SISS_K_FLASH_FWD);
    const FAShape sh{1, 0, D_pad, Sq_pad, Sk_pad, Sq_pad, Sk_pad, valid_k};
    return fa_launch_fwd(FwdArgs{q, k, v, o, D_pad, D_pad, D_pad, D_pad, lse2}, BH, sh, scale, false, stream);
}

// The same on the projections' own layout: q / o rows [B * Sq][ld >= H * D], k / v rows [B * Sk][ld], head h at columns
// [h * D, (h + 1) * D) -- no head-split / head-merge copies and no padded tensors.  D % 8 == 0, D <= 192; every ld % 8 == 0;
// lse2 [B * H][ceil64(Sq)] f32.  Keys: all Sk rows are valid.
// q_prescaled != 0: q holds scale * log2(e) * Q (the projection's epilogue applied the factor before its one rounding to bf16:
// siss_gemm_nt's alpha, siss_gemm_nt_alpha_cols for a fused q / k / v product); o and lse2 mean the same, and the backward still
// returns the gradient with respect to the UNSCALED query -- the score elements then cost one vector multiply less.
int siss_flash_attn_fwd_merged(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, void* o, long ldo,
                               float* lse2, int B, int H, int Sq, int Sk, int D, float scale, int q_prescaled, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && lse2 && B > 0 && H > 0 && Sq > 0 && Sk > 0 && D > 0 && D % 8 == 0 && fa_dpad(D));
    SISS_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && ldq >= (long)H * D && ldk >= (long)H * D &&
                   ldv >= (long)H * D && ldo >= (long)H * D);
    SISS_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) % 16 == 0);
    siss_count_dispatch(SISS_K_FLASH_FWD);
    {   // narrow heads on whole 128-row blocks: the 32x32x16 form (flash_attn32.hip)
        const FA32FwdArgs a32{q, k, v, o, ldq, ldk, ldv, ldo, lse2, B, H, D, Sq, Sk, scale, q_prescaled != 0};
        if (siss_fa32_fwd_takes(a32)) return siss_fa32_fwd(a32, stream);
    }
    const FAShape sh{H, D, D, Sq, Sk, up64(Sq), up64(Sk), Sk};
    return fa_launch_fwd(FwdArgs{q, k, v, o, ldq, ldk, ldv, ldo, lse2}, B * H, sh, scale, q_prescaled != 0, stream);
}

// dQ, dK, dV for nBH cotangent (batch, head) entries against BH forward entries (entry z uses forward entry z % BH: the
// dual-cotangent backward).  dO / dQ [nBH][Sq_pad][D_pad], dK / dV [nBH][Sk_pad][D_pad] bf16; delta [nBH][Sq_pad] f32 =
// rowsum(dO o O) (siss_rowdot); lse2 from the forward.
int siss_flash_attn_bwd(const void* q, const void* k, const void* v, const void* d_o, const float* lse2, const float* delta,
                        void* dq, void* dk, void* dv, int nBH, int BH, int Sq_pad, int Sk_pad, int D_pad, int valid_k,
                        float scale, void* stream) {
    SISS_CHECK_ARG(q && k && v && d_o && lse2 && delta && dq && dk && dv && BH > 0 && nBH >= BH && nBH % BH == 0);
    SISS_CHECK_ARG(fa_shape_ok(Sq_pad, Sk_pad, D_pad, valid_k));
    SISS_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0);
    SISS_CHECK_ARG(((uintptr_t)lse2 | (uintptr_t)delta) % 16 == 0);
    siss_count_dispatch(SISS_K_FLASH_BWD);
    const FAShape sh{1, 0, D_pad, Sq_pad, Sk_pad, Sq_pad, Sk_pad, valid_k};
    const long d = D_pad;
    return fa_launch_bwd(BwdArgs{q, k, v, nullptr, d_o, dq, dk, dv, d, d, d, d, d, d, d, d, lse2, const_cast<float*>(delta)},
                         nBH, BH, sh, scale, false, stream);
}

// The same on the projections' own layout (see siss_flash_attn_fwd_merged): nB cotangent batch entries against B forward ones
// (entry b uses forward entry b % B).  o: the forward's output (delta = rowsum(dO o O) is formed by the dQ kernel and parked in
// `delta`, scratch of nB * H * ceil64(Sq) floats, 16-B aligned).  d_o / dq rows [nB * Sq], dk / dv rows [nB * Sk].
int siss_flash_attn_bwd_merged(const void* q, long ldq, const void* k, long ldk, const void* v, long ldv, const void* o, long ldo,
                               const void* d_o, long lddo, const float* lse2, float* delta, void* dq, long lddq, void* dk,
                               long lddk, void* dv, long lddv, int nB, int B, int H, int Sq, int Sk, int D, float scale,
                               int q_prescaled, void* stream) {
    SISS_CHECK_ARG(q && k && v && o && d_o && lse2 && delta && dq && dk && dv && B > 0 && nB >= B && nB % B == 0 && H > 0);
    SISS_CHECK_ARG(Sq > 0 && Sk > 0 && D > 0 && D % 8 == 0 && fa_dpad(D));
    const long need = (long)H * D;
    SISS_CHECK_ARG(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0 && lddq % 8 == 0 && lddk % 8 == 0 &&
                   lddv % 8 == 0 && ldq >= need && ldk >= need && ldv >= need && ldo >= need && lddo >= need && lddq >= need &&
                   lddk >= need && lddv >= need);
    SISS_CHECK_ARG(((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o | (uintptr_t)d_o | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) % 16 == 0);
    SISS_CHECK_ARG(((uintptr_t)lse2 | (uintptr_t)delta) % 16 == 0);
    siss_count_dispatch(SISS_K_FLASH_BWD);
    {   // narrow heads on whole 128-row blocks (SD's 4096-key self-attention sites): the 32x32x16 form (flash_attn32.hip)
        const FA32Args a32{q, k, v, o, d_o, dq, dk, dv, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, lse2, delta, nB, B, H, D, Sq, Sk, scale, q_prescaled != 0};
        if (siss_fa32_bwd_takes(a32)) return siss_fa32_bwd(a32, stream);
    }
    const FAShape sh{H, D, D, Sq, Sk, up64(Sq), up64(Sk), Sk};
    return fa_launch_bwd(BwdArgs{q, k, v, o, d_o, dq, dk, dv, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv, lse2, delta},
                         nB * H, B, sh, scale, q_prescaled != 0, stream);
}

}  // extern "C"
