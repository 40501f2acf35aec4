// Fused multi-head attention for the SD UNet's transformer blocks (delete_sd.py:977-985 -> losses/ddpm_deletion_loss.py:24,
// diffusers BasicTransformerBlock attn1 / attn2): QK^T -> softmax -> .V in ONE kernel, and its backward in two, on bf16 MFMA
// with LDS-staged tiles.  The S x S score / probability matrices never touch HBM (at 64 x 64 latents they are 1 GB per site
// and pass in the GEMM + softmax form this replaces: 22 of the 77 ms of the SD-v1.5 step).
//
// Operands are the head-split tensors the transformer path already keeps: [B*heads][S_pad][D_pad] bf16, zero padded
// (S_pad % 64 == 0, D_pad in {64, 128, 192}).  `valid_k` masks the padded keys (cross-attention: 77 of 128).
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
    __device__ static __forceinline__ int swz(int row) {
        return RS == 128 ? ((((row >> 1) & 3) << 1) | ((row >> 3) & 1)) : (((row & 7) << 1) | ((row >> 3) & 1));
    }
    __device__ static __forceinline__ int off(int row, int chunk) { return row * RS + ((chunk ^ swz(row)) << 4); }
};

// A [64][DP] tile (global row stride DP; the tensors are padded, all 64 rows exist) travels global -> registers -> LDS in two
// halves, so that the loads of tile t+1 are in flight while tile t is being multiplied (one LDS buffer, register prefetch).
template <int DP> struct TileRegs { u32x4_t v[FA<DP>::CH * kTQ / kThreadsFA]; };
template <int DP>
__device__ __forceinline__ void tile_load(const bf16_t* __restrict__ g, TileRegs<DP>& r, int tid) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        const int row = idx / F::CH, c = idx - row * F::CH;
        r.v[i] = *reinterpret_cast<const u32x4_t*>(g + (long)row * DP + c * 8);
    }
}
template <int DP>
__device__ __forceinline__ void tile_store(const TileRegs<DP>& r, char* lds, int tid) {
    using F = FA<DP>;
#pragma unroll
    for (int i = 0; i < F::CH * kTQ / kThreadsFA; ++i) {
        const int idx = i * kThreadsFA + tid;
        const int row = idx / F::CH, c = idx - row * F::CH;
        *reinterpret_cast<u32x4_t*>(lds + F::off(row, c)) = r.v[i];
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

// ====================================================================================================================
// forward: O = softmax(scale Q K^T) V, LSE2[q] = log2 sum_k exp2(scale log2e (q.k))   (base-2 log-sum-exp)
// grid (Sq_pad / 64, B*heads)
// ====================================================================================================================
template <int DP>
__global__ __launch_bounds__(kThreadsFA) void flash_fwd_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                               const bf16_t* __restrict__ V, bf16_t* __restrict__ O,
                                                               float* __restrict__ LSE2, int Sqp, int Skp, int valid_k,
                                                               float scale_log2) {
    using F = FA<DP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks_ = smem;                       // K tile
    char* vs_ = smem + F::TILE;             // V tile
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long bh = blockIdx.y;
    const int q0 = blockIdx.x * kTQ + w * 16;
    const bf16_t* qg = Q + (bh * Sqp + q0 + (lane & 15)) * DP + (lane >> 4) * 8;
    bf16x8_t qf[F::KS];
#pragma unroll
    for (int ks = 0; ks < F::KS; ++ks) qf[ks] = *reinterpret_cast<const bf16x8_t*>(qg + ks * 32);
    f32x4_t ot[F::DT];
#pragma unroll
    for (int dt = 0; dt < F::DT; ++dt) ot[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const bf16_t* kg = K + bh * Skp * DP;
    const bf16_t* vg = V + bh * Skp * DP;
    TileRegs<DP> kr, vr;
    tile_load<DP>(kg, kr, tid);
    tile_load<DP>(vg, vr, tid);
    for (int k0 = 0; k0 < Skp; k0 += kTQ) {
        __syncthreads();                                        // the previous tile's readers are done
        tile_store<DP>(kr, ks_, tid);
        tile_store<DP>(vr, vs_, tid);
        __syncthreads();
        if (k0 + kTQ < Skp) {                                   // next tile: in flight under this tile's products
            tile_load<DP>(kg + (long)(k0 + kTQ) * DP, kr, tid);
            tile_load<DP>(vg + (long)(k0 + kTQ) * DP, vr, tid);
        }
        f32x4_t st[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            st[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < F::KS; ++ks)
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
        const float m_new = fmaxf(m, group_max(mx));            // finite: every 64-key tile up to valid_k has a valid key... see launcher
        const float alpha = fast_exp2((m - m_new) * scale_log2);    // m = -inf on the first tile: alpha = 0
        const float mc = m_new * scale_log2;
        float ps = 0.f;
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float p = fast_exp2(fmaf(st[sub][r], scale_log2, -mc)); st[sub][r] = p; ps += p; }
        l = l * alpha + group_sum(ps);
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < F::DT; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) ot[dt][r] *= alpha;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8_t pf = pack_pair(st[2 * j], st[2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < F::DT; ++dt)
                ot[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(vs_, j, dt, lane), pf, ot[dt], 0, 0, 0);
        }
    }
    const float inv = 1.f / l;
    bf16_t* og = O + (bh * Sqp + q0 + (lane & 15)) * DP + (lane >> 4) * 4;
#pragma unroll
    for (int dt = 0; dt < F::DT; ++dt)
        *reinterpret_cast<u32x2_t*>(og + dt * 16) = u32x2_t{pack_bf2(ot[dt][0] * inv, ot[dt][1] * inv), pack_bf2(ot[dt][2] * inv, ot[dt][3] * inv)};
    if ((lane >> 4) == 0) LSE2[bh * Sqp + q0 + lane] = m * scale_log2 + log2f(l);
}

// ====================================================================================================================
// backward, dQ:  dQ = scale * dS K,  dS = P o (dO V^T - delta)        grid (Sq_pad / 64, nB*heads)
// z = cotangent batch-head index; the forward tensors (Q, K, V, LSE2) are indexed z % BH (dual-cotangent backward).
// ====================================================================================================================
template <int DP>
__global__ __launch_bounds__(kThreadsFA) void flash_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                  const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                                  const float* __restrict__ LSE2, const float* __restrict__ delta,
                                                                  bf16_t* __restrict__ dQ, int BH, int Sqp, int Skp,
                                                                  int valid_k, float scale, float scale_log2) {
    using F = FA<DP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ks_ = smem;
    char* vs_ = smem + F::TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long z = blockIdx.y, zf = z % BH;
    const int q0 = blockIdx.x * kTQ + w * 16;
    const long qrow = q0 + (lane & 15);
    bf16x8_t qf[F::KS], dof[F::KS];
#pragma unroll
    for (int ks = 0; ks < F::KS; ++ks) {
        qf[ks] = *reinterpret_cast<const bf16x8_t*>(Q + (zf * Sqp + qrow) * DP + (lane >> 4) * 8 + ks * 32);
        dof[ks] = *reinterpret_cast<const bf16x8_t*>(dO + (z * Sqp + qrow) * DP + (lane >> 4) * 8 + ks * 32);
    }
    const float lse = LSE2[zf * Sqp + qrow], dl = delta[z * Sqp + qrow];
    f32x4_t dqt[F::DT];
#pragma unroll
    for (int dt = 0; dt < F::DT; ++dt) dqt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const bf16_t* kg = K + zf * Skp * DP;
    const bf16_t* vg = V + zf * Skp * DP;
    TileRegs<DP> kr, vr;
    tile_load<DP>(kg, kr, tid);
    tile_load<DP>(vg, vr, tid);
    for (int k0 = 0; k0 < Skp; k0 += kTQ) {
        __syncthreads();
        tile_store<DP>(kr, ks_, tid);
        tile_store<DP>(vr, vs_, tid);
        __syncthreads();
        if (k0 + kTQ < Skp) {
            tile_load<DP>(kg + (long)(k0 + kTQ) * DP, kr, tid);
            tile_load<DP>(vg + (long)(k0 + kTQ) * DP, vr, tid);
        }
        f32x4_t st[4], dp[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            st[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < F::KS; ++ks) {
                st[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(ks_, sub, ks, lane), qf[ks], st[sub], 0, 0, 0);
                dp[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(vs_, sub, ks, lane), dof[ks], dp[sub], 0, 0, 0);
            }
        }
        // dS without its factor `scale` (applied once to the finished dQ tile); the key mask only in tiles with padded keys
#pragma unroll
        for (int sub = 0; sub < 4; ++sub)
#pragma unroll
            for (int r = 0; r < 4; ++r) st[sub][r] = fast_exp2(fmaf(st[sub][r], scale_log2, -lse)) * (dp[sub][r] - dl);
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
            for (int dt = 0; dt < F::DT; ++dt)
                dqt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(ks_, j, dt, lane), dsf, dqt[dt], 0, 0, 0);
        }
    }
    bf16_t* og = dQ + (z * Sqp + qrow) * DP + (lane >> 4) * 4;
#pragma unroll
    for (int dt = 0; dt < F::DT; ++dt)
        *reinterpret_cast<u32x2_t*>(og + dt * 16) = u32x2_t{pack_bf2(dqt[dt][0] * scale, dqt[dt][1] * scale), pack_bf2(dqt[dt][2] * scale, dqt[dt][3] * scale)};
}

// ====================================================================================================================
// backward, dK / dV:  dV = P^T dO,  dK = scale * dS^T Q             grid (Sk_pad / 64, nB*heads)
// S[q][key] = mfma(Q rows, K rows): a lane keeps ONE key and 4 consecutive queries of each 16-query sub-tile.
// ====================================================================================================================
template <int DP>
__global__ __launch_bounds__(kThreadsFA) void flash_bwd_dkdv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                    const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                                    const float* __restrict__ LSE2, const float* __restrict__ delta,
                                                                    bf16_t* __restrict__ dK, bf16_t* __restrict__ dV, int BH,
                                                                    int Sqp, int Skp, int valid_k, float scale, float scale_log2) {
    using F = FA<DP>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* qs_ = smem;
    char* dos_ = smem + F::TILE;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const long z = blockIdx.y, zf = z % BH;
    const int k0 = blockIdx.x * kTQ + w * 16;
    const long krow = k0 + (lane & 15);
    const float key_lse_off = krow < valid_k ? 0.f : -INFINITY;     // a lane keeps ONE key: its mask is one additive constant
    bf16x8_t kf[F::KS], vf[F::KS];
#pragma unroll
    for (int ks = 0; ks < F::KS; ++ks) {
        kf[ks] = *reinterpret_cast<const bf16x8_t*>(K + (zf * Skp + krow) * DP + (lane >> 4) * 8 + ks * 32);
        vf[ks] = *reinterpret_cast<const bf16x8_t*>(V + (zf * Skp + krow) * DP + (lane >> 4) * 8 + ks * 32);
    }
    f32x4_t dkt[F::DT], dvt[F::DT];
#pragma unroll
    for (int dt = 0; dt < F::DT; ++dt) { dkt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dvt[dt] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }
    const bf16_t* qg = Q + zf * Sqp * DP;
    const bf16_t* dog = dO + z * Sqp * DP;
    const float* lseg = LSE2 + zf * Sqp;
    const float* dlg = delta + z * Sqp;
    TileRegs<DP> qr, dor;
    tile_load<DP>(qg, qr, tid);
    tile_load<DP>(dog, dor, tid);
    for (int q0 = 0; q0 < Sqp; q0 += kTQ) {
        __syncthreads();
        tile_store<DP>(qr, qs_, tid);
        tile_store<DP>(dor, dos_, tid);
        __syncthreads();
        if (q0 + kTQ < Sqp) {
            tile_load<DP>(qg + (long)(q0 + kTQ) * DP, qr, tid);
            tile_load<DP>(dog + (long)(q0 + kTQ) * DP, dor, tid);
        }
        f32x4_t s[4], dp[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            s[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dp[sub] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < F::KS; ++ks) {
                s[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(qs_, sub, ks, lane), kf[ks], s[sub], 0, 0, 0);
                dp[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_rm<DP>(dos_, sub, ks, lane), vf[ks], dp[sub], 0, 0, 0);
            }
        }
        f32x4_t ds[4];
#pragma unroll
        for (int sub = 0; sub < 4; ++sub) {
            const int qb = q0 + sub * 16 + (lane >> 4) * 4;
            const f32x4_t lse = *reinterpret_cast<const f32x4_t*>(lseg + qb);
            const f32x4_t dl = *reinterpret_cast<const f32x4_t*>(dlg + qb);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = fast_exp2(fmaf(s[sub][r], scale_log2, key_lse_off - lse[r]));     // padded key: exp2(-inf) = 0
                s[sub][r] = p;
                ds[sub][r] = p * (dp[sub][r] - dl[r]);                                              // (scale: once, on the finished dK tile)
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const bf16x8_t pf = pack_pair(s[2 * j], s[2 * j + 1]);
            const bf16x8_t dsf = pack_pair(ds[2 * j], ds[2 * j + 1]);
#pragma unroll
            for (int dt = 0; dt < F::DT; ++dt) {
                dvt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(dos_, j, dt, lane), pf, dvt[dt], 0, 0, 0);
                dkt[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(frag_tr<DP>(qs_, j, dt, lane), dsf, dkt[dt], 0, 0, 0);
            }
        }
    }
    bf16_t* okg = dK + (z * Skp + krow) * DP + (lane >> 4) * 4;
    bf16_t* ovg = dV + (z * Skp + krow) * DP + (lane >> 4) * 4;
#pragma unroll
    for (int dt = 0; dt < F::DT; ++dt) {
        *reinterpret_cast<u32x2_t*>(okg + dt * 16) = u32x2_t{pack_bf2(dkt[dt][0] * scale, dkt[dt][1] * scale), pack_bf2(dkt[dt][2] * scale, dkt[dt][3] * scale)};
        *reinterpret_cast<u32x2_t*>(ovg + dt * 16) = u32x2_t{pack_bf2(dvt[dt][0], dvt[dt][1]), pack_bf2(dvt[dt][2], dvt[dt][3])};
    }
}

bool fa_shape_ok(int Sqp, int Skp, int Dp, int valid_k) {
    return Sqp > 0 && Skp > 0 && Sqp % kTQ == 0 && Skp % kTQ == 0 && (Dp == 64 || Dp == 128 || Dp == 192) && valid_k > 0 &&
           valid_k <= Skp;
}

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
    const dim3 grid(Sq_pad / kTQ, BH);
    const float sl2 = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
#define FA_FWD(DP)                                                                                                       \
    do {                                                                                                                  \
        static unsigned char att[kMaxDevices];                                                                            \
        if (siss_ensure_smem((const void*)flash_fwd_kernel<DP>, 2 * FA<DP>::TILE, att) != SISS_OK) return SISS_ERR_LAUNCH; \
        flash_fwd_kernel<DP><<<grid, kThreadsFA, 2 * FA<DP>::TILE, st>>>((const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, \
                                                                        (bf16_t*)o, lse2, Sq_pad, Sk_pad, valid_k, sl2);  \
    } while (0)
    if (D_pad == 64) FA_FWD(64); else if (D_pad == 128) FA_FWD(128); else FA_FWD(192);
#undef FA_FWD
    SISS_LAUNCH_RET();
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
    const float sl2 = scale * 1.4426950408889634f;
    hipStream_t st = (hipStream_t)stream;
#define FA_BWD(DP)                                                                                                          \
    do {                                                                                                                     \
        static unsigned char a1[kMaxDevices], a2[kMaxDevices];                                                               \
        if (siss_ensure_smem((const void*)flash_bwd_dq_kernel<DP>, 2 * FA<DP>::TILE, a1) != SISS_OK) return SISS_ERR_LAUNCH; \
        if (siss_ensure_smem((const void*)flash_bwd_dkdv_kernel<DP>, 2 * FA<DP>::TILE, a2) != SISS_OK) return SISS_ERR_LAUNCH; \
        flash_bwd_dkdv_kernel<DP><<<dim3(Sk_pad / kTQ, nBH), kThreadsFA, 2 * FA<DP>::TILE, st>>>(                          \
            (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, (bf16_t*)dk, (bf16_t*)dv, \
            BH, Sq_pad, Sk_pad, valid_k, scale, sl2);                                                                        \
        flash_bwd_dq_kernel<DP><<<dim3(Sq_pad / kTQ, nBH), kThreadsFA, 2 * FA<DP>::TILE, st>>>(                            \
            (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, (const bf16_t*)d_o, lse2, delta, (bf16_t*)dq, BH, Sq_pad,  \
            Sk_pad, valid_k, scale, sl2);                                                                                    \
    } while (0)
    if (D_pad == 64) FA_BWD(64); else if (D_pad == 128) FA_BWD(128); else FA_BWD(192);
#undef FA_BWD
    SISS_LAUNCH_RET();
}

}  // extern "C"
