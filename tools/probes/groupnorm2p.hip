// Two-phase, on-chip-resident GroupNorm (+SiLU) forward / backward: every activation byte crosses HBM ONCE.
//
// groupnorm.hip needs two passes over its inputs (statistics, then apply): the statistics of a (sample, group) cover the
// whole image, so a block cannot normalise what it has just read.  At the large sites that second read is 40 % of the
// kernel's HBM traffic (PMC, round 1: 50.2 GB per step against 34.3 GB algorithmic) -- and these kernels are 24 % of the
// step.  Here ONE persistent launch (one 512-thread block per CU, at most 256 blocks, all co-resident) does both phases:
//
//   phase 1   every block loads its pixel run of sample n and KEEPS it on chip (forward: x in registers; backward: x in
//             LDS, both cotangent sets in registers), forms its partial group statistics and adds them to the sample's
//             accumulators in HBM;
//   barrier   per SAMPLE (not per grid): an arrival counter; blocks of other samples are not held up;
//   phase 2   the totals are read back and the block normalises / back-propagates what it still holds, storing y / dx.
//
// A run that is longer than the on-chip capacity (V 16-B vectors per thread and tensor; the 256-channel 256x256 sites)
// keeps its head and re-reads only its tail.
//
// Determinism.  The accumulators are FIXED-POINT (two int64 limbs per statistic: units of 2^-20 and 2^-72, exact for
// any f32 partial below 2^40 in magnitude), added with integer atomics: integer addition is associative, so the totals
// -- and with them y, dx -- are bitwise independent of the arrival order.  Block-level partials are folded in a fixed
// order through LDS (no LDS atomics).  Eight accumulator replicas (block index mod 8: one per XCD) spread the atomics.
//
// Synchronisation is through agent-scope atomics only (they execute at the device's coherence point, past the per-XCD
// L2s): partial adds -> s_waitcnt vmcnt(0) -> arrival add;  waiters poll the counter with atomic loads and read the
// accumulators with atomic loads.  No fences (a release fence would write back the whole L2 of the XCD, which is full of
// the dx rows this very kernel is streaming out).  The LAST block to have read a sample's totals zeroes its accumulators
// and counters, so the workspace is left as it was found: all zero.  Every spin has a bounded iteration count.
#include "common.h"
#include <type_traits>

namespace {

constexpr int kT = 512;                 // threads per block (8 waves: 256 VGPRs per lane), one block per CU
constexpr int kMaxBlocks = 256;
constexpr int kRep = 8;                 // accumulator replicas per sample
constexpr int kMaxG = 32, kMaxC = 1024;
constexpr int kV = 8;                   // 16-B vectors per thread and tensor kept on chip between the phases
constexpr int kMaxStat = 4 * kMaxG;     // backward: 2 sets x G groups x (S1, S2)
constexpr long kSpinLimit = 1L << 21;   // ~ seconds: a barrier that is not met is given up (results are then garbage, the launch still ends)
constexpr int kCtrInts = 4;             // per sample: arrived, done, non-finite flag, pad

struct Shape2 {
    int H, W, C, G, cpg, lpp, ppi, P;
    int ld, ldx;
    int bps, spr, rounds, run_px;       // blocks per sample, samples per round, rounds, pixels per block and sample
};

__device__ __forceinline__ void unpack8(u32x4_t r, float (&v)[8]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[2 * j] = __builtin_bit_cast(float, r[j] << 16);
        v[2 * j + 1] = __builtin_bit_cast(float, r[j] & 0xffff0000u);
    }
}
__device__ __forceinline__ u32x4_t pack8(const float (&v)[8]) {
    return u32x4_t{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
}

// interior pixels p0 + slot, + ppi, ... < p1 of one image; (y, x) kept incrementally
struct Walk {
    int pi, p1, y, x, W, ppi, dy, dx;
    __device__ __forceinline__ Walk(const Shape2& s, int chunk, int slot) {
        const int p0 = chunk * s.run_px;
        p1 = p0 + s.run_px; p1 = p1 < s.P ? p1 : s.P;
        pi = p0 + slot; W = s.W; ppi = s.ppi;
        y = pi / W; x = pi - y * W;
        dy = ppi / W; dx = ppi - dy * W;
    }
    __device__ __forceinline__ bool ok() const { return pi < p1; }
    __device__ __forceinline__ long row() const { return (long)(y + 1) * (W + 2) + (x + 1); }
    __device__ __forceinline__ void next() {
        pi += ppi; y += dy; x += dx;
        if (x >= W) { x -= W; ++y; }
    }
};

// ---- fixed-point accumulation -------------------------------------------------------------------------------------
__device__ __forceinline__ void fx_add(unsigned long long* acc2, int* flag, float v) {
    if (!(fabsf(v) < 1.0e12f)) {         // NaN / inf / absurd: poison the sample's statistics instead of wrapping
        __hip_atomic_fetch_or(flag, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
    }
    const double d = (double)v;
    const long long hi = (long long)(d * 1048576.0);                                   // units of 2^-20 (exact: v has 24 bits)
    const double rem = d - (double)hi * (1.0 / 1048576.0);                              // exact, |rem| < 2^-20
    const long long lo = (long long)(rem * 4722366482869645213696.0);                   // units of 2^-72, |lo| < 2^52
    if (hi) __hip_atomic_fetch_add(acc2, (unsigned long long)hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (lo) __hip_atomic_fetch_add(acc2 + 1, (unsigned long long)lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// total of statistic `t` of one sample over the replicas (exact integers -> one double)
__device__ __forceinline__ double fx_total(const unsigned long long* acc, int nstat, int t, const int* flag) {
    long long hi = 0, lo = 0;
#pragma unroll
    for (int r = 0; r < kRep; ++r) {
        const unsigned long long* p = acc + ((long)r * nstat + t) * 2;
        hi += (long long)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lo += (long long)__hip_atomic_load(p + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    const bool bad = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
    const double tot = (double)hi * (1.0 / 1048576.0) + (double)lo * (1.0 / 4722366482869645213696.0);
    return bad ? __builtin_nan("") : tot;
}

// ---- per-sample barrier -------------------------------------------------------------------------------------------
__device__ __forceinline__ void sample_arrive(int* ctr) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this thread's accumulator atomics have been performed
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void sample_wait(const int* ctr, int target) {
    if (threadIdx.x == 0) {
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++spins < kSpinLimit)
            __builtin_amdgcn_s_sleep(8);
    }
    __syncthreads();
}
// after the block has consumed the totals: the last block of the sample restores the all-zero workspace
__device__ __forceinline__ void sample_done(int* ctr, unsigned long long* acc, int nstat, int bps, int* sh_last) {
    __syncthreads();
    if (threadIdx.x == 0) *sh_last = __hip_atomic_fetch_add(ctr + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == bps - 1;
    __syncthreads();
    if (*sh_last) {
        for (int i = threadIdx.x; i < kRep * nstat * 2; i += kT)
            __hip_atomic_store(acc + i, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (threadIdx.x < 3) __hip_atomic_store(ctr + threadIdx.x, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Deterministic block reduction over the pixel slots: NA arrays of 8 per-lane channel sums -> out[a][c] (all threads see
// the result).  red holds NA * kT * 8 floats.
template <int NA>
__device__ __forceinline__ void reduce_slots(const float (&v)[NA][8], bool active, int slot, int cc, const Shape2& s,
                                             float* red, float (*out)[kMaxC]) {
    if (active) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            float* dst = red + (a * s.ppi + slot) * s.C + cc * 8;
            *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[a][0], v[a][1], v[a][2], v[a][3]};
            *reinterpret_cast<f32x4_t*>(dst + 4) = f32x4_t{v[a][4], v[a][5], v[a][6], v[a][7]};
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NA * s.C; i += kT) {
        const int a = i / s.C, c = i - a * s.C;
        const float* src = red + a * s.ppi * s.C + c;
        float t = 0.f;
        for (int sl = 0; sl < s.ppi; ++sl) t += src[sl * s.C];
        out[a][c] = t;
    }
    __syncthreads();
}

// =====================================================================================================================
// forward
// =====================================================================================================================
template <bool SILU>
__global__ __launch_bounds__(kT, 1) void gn2p_fwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, Shape2 s, float eps,
    int out_compact, bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out, int N,
    int* __restrict__ ctr, unsigned long long* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);                       // [2][kT * 8]
    float (*chs)[kMaxC] = reinterpret_cast<float (*)[kMaxC]>(red + 2 * kT * 8);   // [2][kMaxC]
    __shared__ double tot[2 * kMaxG];
    __shared__ float sh_mean[kMaxG], sh_rstd[kMaxG];
    __shared__ int sh_last;
    const int tid = threadIdx.x;
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    const bool active = slot < s.ppi;
    const int sr = blockIdx.x / s.bps, chunk = blockIdx.x - sr * s.bps;
    const int rep = blockIdx.x & (kRep - 1);
    const int nstat = 2 * s.G;
    const long rpi = (long)(s.H + 2) * (s.W + 2);
    float ga[8], be[8];
    int grp[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = active ? cc * 8 + e : 0;
        ga[e] = gamma[c]; be[e] = beta[c]; grp[e] = c / s.cpg;
    }
    for (int r = 0; r < s.rounds; ++r) {
        const int n = r * s.spr + sr;
        if (n >= N) break;
        int* cn = ctr + (long)n * kCtrInts;
        unsigned long long* an = acc + (long)n * kRep * kMaxStat * 2;
        const bf16_t* base = x + (long)n * rpi * s.ldx + cc * 8;
        // ---------------- phase 1: load the run (head kept in registers), partial statistics
        u32x4_t cx[kV];
        float ab[2][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { ab[0][e] = 0.f; ab[1][e] = 0.f; }
        {
            Walk w(s, chunk, slot);
#pragma unroll
            for (int j = 0; j < kV; ++j) {
                cx[j] = (active && w.ok()) ? *reinterpret_cast<const u32x4_t*>(base + w.row() * s.ldx) : u32x4_t{0u, 0u, 0u, 0u};
                w.next();
            }
            while (active && w.ok()) {              // tail beyond the on-chip capacity: streamed, re-read in phase 2
                const u32x4_t r0 = *reinterpret_cast<const u32x4_t*>(base + w.row() * s.ldx);
                w.next();
                u32x4_t r1 = u32x4_t{0u, 0u, 0u, 0u};
                if (w.ok()) { r1 = *reinterpret_cast<const u32x4_t*>(base + w.row() * s.ldx); w.next(); }
                float v[8], u[8];
                unpack8(r0, v); unpack8(r1, u);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ab[0][e] += v[e] + u[e]; ab[1][e] += v[e] * v[e] + u[e] * u[e]; }
            }
#pragma unroll
            for (int j = 0; j < kV; ++j) {
                float v[8];
                unpack8(cx[j], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { ab[0][e] += v[e]; ab[1][e] += v[e] * v[e]; }
            }
        }
        reduce_slots<2>(ab, active, slot, cc, s, red, chs);
        if (tid < nstat) {
            const int g = tid >> 1;
            const float* src = chs[tid & 1];
            float t = 0.f;
            for (int c = g * s.cpg; c < (g + 1) * s.cpg; ++c) t += src[c];
            fx_add(an + ((long)rep * nstat + tid) * 2, cn + 2, t);
        }
        sample_arrive(cn);
        sample_wait(cn, s.bps);
        if (tid < nstat) tot[tid] = fx_total(an, nstat, tid, cn + 2);
        __syncthreads();
        if (tid < s.G) {
            const double cnt = (double)s.P * s.cpg;
            const double m = tot[2 * tid] / cnt;
            double var = tot[2 * tid + 1] / cnt - m * m;
            var = var > 0 ? var : (var == var ? 0 : var);          // keep a NaN a NaN
            const float rs = (float)(1.0 / sqrt(var + (double)eps));
            sh_mean[tid] = (float)m; sh_rstd[tid] = rs;
            if (chunk == 0) { mean_out[(long)n * s.G + tid] = (float)m; rstd_out[(long)n * s.G + tid] = rs; }
        }
        sample_done(cn, an, nstat, s.bps, &sh_last);            // (its leading barrier publishes sh_mean / sh_rstd)
        // ---------------- phase 2: normalise what is still on chip, then the re-read tail
        if (active) {
            float sc[8], sf[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = sh_rstd[grp[e]] * ga[e]; sf[e] = be[e] - sh_mean[grp[e]] * sc[e]; }
            const long img = (long)n * rpi;
            auto apply = [&](u32x4_t rx, const Walk& w) {
                float v[8];
                unpack8(rx, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float z = v[e] * sc[e] + sf[e]; v[e] = SILU ? silu_f(z) : z; }
                const long orow = out_compact ? (long)n * s.P + w.pi : img + w.row();
                *reinterpret_cast<u32x4_t*>(y + orow * s.ld + cc * 8) = pack8(v);
            };
            Walk w(s, chunk, slot);
#pragma unroll
            for (int j = 0; j < kV; ++j) {
                if (w.ok()) apply(cx[j], w);
                w.next();
            }
            while (w.ok()) {
                const u32x4_t r0 = *reinterpret_cast<const u32x4_t*>(base + w.row() * s.ldx);
                Walk w0 = w;
                w.next();
                if (w.ok()) {
                    const u32x4_t r1 = *reinterpret_cast<const u32x4_t*>(base + w.row() * s.ldx);
                    apply(r0, w0); apply(r1, w);
                    w.next();
                } else {
                    apply(r0, w0);
                }
            }
        }
    }
}

// =====================================================================================================================
// backward
// =====================================================================================================================
struct BwdArgs {
    const bf16_t* dy; const bf16_t* x; const float* gamma; const float* beta; const float* mean; const float* rstd;
    const bf16_t* accum; const bf16_t* accum2; bf16_t* dx; bf16_t* dx2;
    float* dgamma; float* dbeta; float* colsum;
    long colsum_ld, set_stride;
    int split_c, accumulate2, nx, dy_compact, set_images;
};

// LDS (dynamic): xc [kV][kT] u32x4 (64 KiB) | red [2][kT*8] f32 (32 KiB) | chs [4][kMaxC] f32 (16 KiB) | shg [2 sets][2][kMaxC] f32 (16 KiB)
constexpr int kBwdSmem = kV * kT * 16 + 2 * kT * 8 * 4 + 4 * kMaxC * 4 + 4 * kMaxC * 4;
constexpr int kFwdSmem = 2 * kT * 8 * 4 + 2 * kMaxC * 4;

template <bool SILU, int SETS, bool EXTRA>
__global__ __launch_bounds__(kT, 1) void gn2p_bwd_kernel(BwdArgs a, Shape2 s, int* __restrict__ ctr,
                                                         unsigned long long* __restrict__ acc) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u32x4_t* xc = reinterpret_cast<u32x4_t*>(smem_raw);
    float* red = reinterpret_cast<float*>(smem_raw + kV * kT * 16);
    float (*chs)[kMaxC] = reinterpret_cast<float (*)[kMaxC]>(red + 2 * kT * 8);
    float (*shg)[kMaxC] = chs + 4;                                 // [set*2 + {0: dbeta, 1: dgamma}][C]
    __shared__ double tot[kMaxStat];
    __shared__ int sh_last;
    const int tid = threadIdx.x;
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    const bool active = slot < s.ppi;
    const int sr = blockIdx.x / s.bps, chunk = blockIdx.x - sr * s.bps;
    const int rep = blockIdx.x & (kRep - 1);
    const int nstat = SETS * 2 * s.G;
    const long rpi = (long)(s.H + 2) * (s.W + 2);
    for (int i = tid; i < 4 * kMaxC; i += kT) shg[0][i] = 0.f;
    float ga[8], be[8];
    const int c_lane = active ? cc * 8 : 0;
    const int g_lo = c_lane / s.cpg;                               // the lane's 8 channels span groups g_lo and (maybe) g_lo + 1
    unsigned hi_mask = 0;                                          // bit e: channel e belongs to g_lo + 1
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ga[e] = a.gamma[c_lane + e]; be[e] = a.beta[c_lane + e];
        if ((c_lane + e) / s.cpg != g_lo) hi_mask |= 1u << e;
    }
    const int g_hi = (g_lo + 1 < s.G) ? g_lo + 1 : g_lo;
    // output routing (see groupnorm.hip): one tensor, or a channel split into dx (row stride split_c) / dx2
    const bool second = a.dx2 != nullptr && c_lane >= a.split_c;
    bf16_t* const obase = second ? a.dx2 + (c_lane - a.split_c) : a.dx + c_lane;
    const int ostride = a.dx2 ? (second ? s.ld - a.split_c : a.split_c) : s.ld;
    const bool oacc = second && a.accumulate2;
    __syncthreads();

    for (int r = 0; r < s.rounds; ++r) {
        const int n = r * s.spr + sr;
        if (n >= a.nx) break;
        int* cn = ctr + (long)n * kCtrInts;
        unsigned long long* an = acc + (long)n * kRep * kMaxStat * 2;
        const bf16_t* xb = a.x + (long)n * rpi * s.ldx + c_lane;
        const float rs0 = a.rstd[(long)n * s.G + g_lo], rs1 = a.rstd[(long)n * s.G + g_hi];
        const float mr0 = a.mean[(long)n * s.G + g_lo] * rs0, mr1 = a.mean[(long)n * s.G + g_hi] * rs1;
        auto rs = [&](int e) { return ((hi_mask >> e) & 1) ? rs1 : rs0; };
        auto mr = [&](int e) { return ((hi_mask >> e) & 1) ? mr1 : mr0; };
        auto dy_ptr = [&](int k, const Walk& w) {
            const int n2 = k * a.nx + n;
            const long drow = a.dy_compact ? (long)n2 * s.P + w.pi : (long)n2 * rpi + w.row();
            return a.dy + drow * s.ld + c_lane;
        };
        // ---------------- phase 1
        u32x4_t cd[SETS][kV];
        float c12[2 * SETS][8];                                    // [k*2 + 0]: sum dz (dbeta), [k*2 + 1]: sum dz xhat (dgamma)
#pragma unroll
        for (int q = 0; q < 2 * SETS; ++q)
#pragma unroll
            for (int e = 0; e < 8; ++e) c12[q][e] = 0.f;
        auto accumulate = [&](u32x4_t rx, const u32x4_t (&rd)[SETS]) {
            float v[8], xh[8], dsl[8];
            unpack8(rx, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[e] = v[e] * rs(e) - mr(e);
                dsl[e] = SILU ? dsilu_f(xh[e] * ga[e] + be[e]) : 1.f;
            }
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                float d[8];
                unpack8(rd[k], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dz = d[e] * dsl[e]; c12[2 * k][e] += dz; c12[2 * k + 1][e] += dz * xh[e]; }
            }
        };
        {
            Walk w(s, chunk, slot);
            // head: kept on chip, loaded and consumed in batches of 4 vectors per tensor (register pressure)
#pragma unroll
            for (int b = 0; b < kV / 4; ++b) {
                u32x4_t hx[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = active && w.ok();
                    hx[j] = ok ? *reinterpret_cast<const u32x4_t*>(xb + w.row() * s.ldx) : u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
                    for (int k = 0; k < SETS; ++k)
                        cd[k][b * 4 + j] = ok ? *reinterpret_cast<const u32x4_t*>(dy_ptr(k, w)) : u32x4_t{0u, 0u, 0u, 0u};   // dy = 0: no contribution
                    w.next();
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    u32x4_t rd[SETS];
#pragma unroll
                    for (int k = 0; k < SETS; ++k) rd[k] = cd[k][b * 4 + j];
                    accumulate(hx[j], rd);
                    xc[(b * 4 + j) * kT + tid] = hx[j];            // x waits in LDS (lane-linear: conflict-free), dy in registers
                }
            }
            // tail (streamed; re-read in phase 2), one pixel ahead
            if (active && w.ok()) {
                u32x4_t nx_x = *reinterpret_cast<const u32x4_t*>(xb + w.row() * s.ldx), nx_d[SETS];
#pragma unroll
                for (int k = 0; k < SETS; ++k) nx_d[k] = *reinterpret_cast<const u32x4_t*>(dy_ptr(k, w));
                while (w.ok()) {
                    const u32x4_t rx = nx_x;
                    u32x4_t rd[SETS];
#pragma unroll
                    for (int k = 0; k < SETS; ++k) rd[k] = nx_d[k];
                    w.next();
                    if (w.ok()) {
                        nx_x = *reinterpret_cast<const u32x4_t*>(xb + w.row() * s.ldx);
#pragma unroll
                        for (int k = 0; k < SETS; ++k) nx_d[k] = *reinterpret_cast<const u32x4_t*>(dy_ptr(k, w));
                    }
                    accumulate(rx, rd);
                }
            }
        }
        // block reduction, two arrays at a time; per-channel sums also feed the block's dgamma / dbeta accumulators
#pragma unroll
        for (int k = 0; k < SETS; ++k) {
            float two[2][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { two[0][e] = c12[2 * k][e]; two[1][e] = c12[2 * k + 1][e]; }
            reduce_slots<2>(two, active, slot, cc, s, red, chs + 2 * k);
        }
        {
            for (int i = tid; i < 2 * SETS * s.C; i += kT) {       // thread-owned entries: no race, fixed order
                const int q = i / s.C, c = i - q * s.C;
                const int set = ((q >> 1) * a.nx + n) / a.set_images;
                shg[(set & 1) * 2 + (q & 1)][c] += chs[q][c];
            }
            if (tid < nstat) {
                const int which = tid & 1, g = (tid >> 1) % s.G, k = (tid >> 1) / s.G;
                const float* src = chs[2 * k + which];
                float t = 0.f;
                for (int c = g * s.cpg; c < (g + 1) * s.cpg; ++c) t += src[c] * a.gamma[c];
                fx_add(an + ((long)rep * nstat + tid) * 2, cn + 2, t);
            }
        }
        sample_arrive(cn);
        sample_wait(cn, s.bps);
        if (tid < nstat) tot[tid] = fx_total(an, nstat, tid, cn + 2) / ((double)s.P * s.cpg);
        sample_done(cn, an, nstat, s.bps, &sh_last);
        // ---------------- phase 2
        float cs[SETS][8];
#pragma unroll
        for (int k = 0; k < SETS; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) cs[k][e] = 0.f;
        if (active) {
            float m1[SETS][2], m2[SETS][2];
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                m1[k][0] = (float)tot[(k * s.G + g_lo) * 2]; m2[k][0] = (float)tot[(k * s.G + g_lo) * 2 + 1];
                m1[k][1] = (float)tot[(k * s.G + g_hi) * 2]; m2[k][1] = (float)tot[(k * s.G + g_hi) * 2 + 1];
            }
            struct Extra { u32x4_t a[SETS], b[SETS], c[SETS]; };
            auto load_extra = [&](const Walk& w, Extra& o) {
#pragma unroll
                for (int k = 0; k < SETS; ++k) {
                    const long orow = (long)(k * a.nx + n) * rpi + w.row();
                    o.a[k] = a.accum ? *reinterpret_cast<const u32x4_t*>(a.accum + orow * s.ld + c_lane) : u32x4_t{0u, 0u, 0u, 0u};
                    o.b[k] = a.accum2 ? *reinterpret_cast<const u32x4_t*>(a.accum2 + orow * s.ld + c_lane) : u32x4_t{0u, 0u, 0u, 0u};
                    o.c[k] = oacc ? *reinterpret_cast<const u32x4_t*>(obase + orow * ostride) : u32x4_t{0u, 0u, 0u, 0u};
                }
            };
            auto emit = [&](u32x4_t rx, const u32x4_t (&rd)[SETS], const Extra& ex, const Walk& w) {
                float v[8], xh[8], dsl[8];
                unpack8(rx, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = v[e] * rs(e) - mr(e);
                    dsl[e] = (SILU ? dsilu_f(xh[e] * ga[e] + be[e]) : 1.f) * ga[e];
                }
#pragma unroll
                for (int k = 0; k < SETS; ++k) {
                    const long orow = (long)(k * a.nx + n) * rpi + w.row();
                    float d[8], o[8];
                    unpack8(rd[k], d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool h = (hi_mask >> e) & 1;
                        const float t = rs(e) * (d[e] * dsl[e] - (h ? m1[k][1] : m1[k][0]) - xh[e] * (h ? m2[k][1] : m2[k][0]));
                        cs[k][e] += t;
                        o[e] = t;
                    }
                    if constexpr (EXTRA) {
                        float r1[8], r2[8], r3[8];
                        unpack8(ex.a[k], r1); unpack8(ex.b[k], r2); unpack8(ex.c[k], r3);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = o[e] + r1[e] + r2[e] + r3[e];
                    }
                    *reinterpret_cast<u32x4_t*>(obase + orow * ostride) = pack8(o);
                }
            };
            Walk w(s, chunk, slot);
            Extra nxt = {};
            if constexpr (EXTRA) { if (w.ok()) load_extra(w, nxt); }
            // head: x from LDS, dy from registers; the residual tensors one pixel ahead
#pragma unroll
            for (int j = 0; j < kV; ++j) {
                if (w.ok()) {
                    const Extra cur = nxt;
                    const Walk w0 = w;
                    w.next();
                    if constexpr (EXTRA) { if (w.ok()) load_extra(w, nxt); }
                    u32x4_t rd[SETS];
#pragma unroll
                    for (int k = 0; k < SETS; ++k) rd[k] = cd[k][j];
                    emit(xc[j * kT + tid], rd, cur, w0);
                }
            }
            // tail: everything re-read, one pixel ahead
            if (w.ok()) {
                u32x4_t nx_x = *reinterpret_cast<const u32x4_t*>(xb + w.row() * s.ldx), nx_d[SETS];
#pragma unroll
                for (int k = 0; k < SETS; ++k) nx_d[k] = *reinterpret_cast<const u32x4_t*>(dy_ptr(k, w));
                while (w.ok()) {
                    const Extra cur = nxt;
                    const u32x4_t rx = nx_x;
                    u32x4_t rd[SETS];
#pragma unroll
                    for (int k = 0; k < SETS; ++k) rd[k] = nx_d[k];
                    const Walk w0 = w;
                    w.next();
                    if (w.ok()) {
                        nx_x = *reinterpret_cast<const u32x4_t*>(xb + w.row() * s.ldx);
#pragma unroll
                        for (int k = 0; k < SETS; ++k) nx_d[k] = *reinterpret_cast<const u32x4_t*>(dy_ptr(k, w));
                        if constexpr (EXTRA) load_extra(w, nxt);
                    }
                    emit(rx, rd, cur, w0);
                }
            }
        }
        if (a.colsum) {
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                float one[1][8];
#pragma unroll
                for (int e = 0; e < 8; ++e) one[0][e] = cs[k][e];
                reduce_slots<1>(one, active, slot, cc, s, red, chs);
                for (int i = tid; i < s.C; i += kT) atomicAdd(a.colsum + (long)(k * a.nx + n) * a.colsum_ld + i, chs[0][i]);
                __syncthreads();
            }
        }
    }
    // the block's share of dgamma / dbeta, once
    __syncthreads();
    for (int i = tid; i < 4 * s.C; i += kT) {
        const int q = i / s.C, c = i - q * s.C;
        const float v = shg[q][c];
        if (v != 0.f) atomicAdd(((q & 1) ? a.dgamma : a.dbeta) + (long)(q >> 1) * a.set_stride + c, v);
    }
}

// =====================================================================================================================
// "slab" kernels for the small sites (at most 32 x 32 pixels: the 8x8 / 16x16 / 32x32 levels, 82 of the step's 142
// GroupNorm launches).  There a block can hold ALL pixels of a channel slice of whole groups of one sample on chip, so the
// statistics need no other block: ONE launch, no barrier, no atomics for the statistics, every byte read once.  (The
// two-pass form costs 17 us forward / 40 us backward per site here -- two dependent launches of latency-bound kernels.)
// block (z, n): channels [z * Cs, (z + 1) * Cs) of sample n.  Forward: x in registers.  Backward: x and the first
// cotangent set in LDS, the second set in registers.
// =====================================================================================================================
struct SlabShape {
    int H, W, P, C, G, cpg;            // full tensor
    int Cs, Gs, lpp, ppi, V;           // slice: channels, groups, lanes per pixel, pixels per iteration, vectors per lane
    int ld, ldx;
};

struct SlabWalk {                       // pixels slot, slot + ppi, ... of the whole image
    int pi, P, y, x, W, ppi, dy, dx;
    __device__ __forceinline__ SlabWalk(const SlabShape& s, int slot) {
        pi = slot; P = s.P; W = s.W; ppi = s.ppi;
        y = pi / W; x = pi - y * W;
        dy = ppi / W; dx = ppi - dy * W;
    }
    __device__ __forceinline__ bool ok() const { return pi < P; }
    __device__ __forceinline__ long row() const { return (long)(y + 1) * (W + 2) + (x + 1); }
    __device__ __forceinline__ void next() {
        pi += ppi; y += dy; x += dx;
        if (x >= W) { x -= W; ++y; }
    }
};

// NA arrays of 8 per-lane channel sums -> out[a][c], c < Cs, folded over the pixel slots in a fixed order
template <int NA>
__device__ __forceinline__ void slab_reduce(const float (&v)[NA][8], bool active, int slot, int cc, const SlabShape& s,
                                            float* red, float* out /* [NA][Cs] */) {
    if (active) {
#pragma unroll
        for (int a = 0; a < NA; ++a) {
            float* dst = red + (a * s.ppi + slot) * s.Cs + cc * 8;
            *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[a][0], v[a][1], v[a][2], v[a][3]};
            *reinterpret_cast<f32x4_t*>(dst + 4) = f32x4_t{v[a][4], v[a][5], v[a][6], v[a][7]};
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NA * s.Cs; i += kT) {
        const int a = i / s.Cs, c = i - a * s.Cs;
        const float* src = red + a * s.ppi * s.Cs + c;
        float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;             // four independent chains: the walk is latency-bound
        int sl = 0;
        for (; sl + 4 <= s.ppi; sl += 4) {
            t0 += src[sl * s.Cs]; t1 += src[(sl + 1) * s.Cs]; t2 += src[(sl + 2) * s.Cs]; t3 += src[(sl + 3) * s.Cs];
        }
        for (; sl < s.ppi; ++sl) t0 += src[sl * s.Cs];
        out[a * s.Cs + c] = (t0 + t1) + (t2 + t3);
    }
    __syncthreads();
}

constexpr int kSlabV = 8;
constexpr int kSlabMaxCs = 128;

template <bool SILU>
__global__ __launch_bounds__(kT, 1) void gn_slab_fwd_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, SlabShape s, float eps,
    int out_compact, bf16_t* __restrict__ y, float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* red = reinterpret_cast<float*>(smem_raw);               // [2][kT * 8]
    __shared__ float chs[2 * kSlabMaxCs];
    __shared__ float sh_mean[kMaxG], sh_rstd[kMaxG];
    const int tid = threadIdx.x, n = blockIdx.y, c0 = blockIdx.x * s.Cs, g0 = blockIdx.x * s.Gs;
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    const bool active = slot < s.ppi;
    const long rpi = (long)(s.H + 2) * (s.W + 2);
    const int c_lane = c0 + (active ? cc * 8 : 0);
    const bf16_t* base = x + (long)n * rpi * s.ldx + c_lane;
    u32x4_t cx[kSlabV];
    float ab[2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { ab[0][e] = 0.f; ab[1][e] = 0.f; }
    {
        SlabWalk w(s, slot);
#pragma unroll
        for (int j = 0; j < kSlabV; ++j) {
            cx[j] = (active && w.ok()) ? *reinterpret_cast<const u32x4_t*>(base + w.row() * s.ldx) : u32x4_t{0u, 0u, 0u, 0u};
            w.next();
        }
#pragma unroll
        for (int j = 0; j < kSlabV; ++j) {
            float v[8];
            unpack8(cx[j], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { ab[0][e] += v[e]; ab[1][e] += v[e] * v[e]; }
        }
    }
    slab_reduce<2>(ab, active, slot, cc, s, red, chs);
    if (tid < s.Gs) {
        double a = 0, b = 0;
        for (int c = tid * s.cpg; c < (tid + 1) * s.cpg; ++c) { a += chs[c]; b += chs[s.Cs + c]; }
        const double cnt = (double)s.P * s.cpg;
        const double m = a / cnt;
        double var = b / cnt - m * m;
        var = var > 0 ? var : (var == var ? 0 : var);
        const float rs = (float)(1.0 / sqrt(var + (double)eps));
        sh_mean[tid] = (float)m; sh_rstd[tid] = rs;
        mean_out[(long)n * s.G + g0 + tid] = (float)m; rstd_out[(long)n * s.G + g0 + tid] = rs;
    }
    __syncthreads();
    if (!active) return;
    float sc[8], sf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int g = (cc * 8 + e) / s.cpg;
        sc[e] = sh_rstd[g] * gamma[c_lane + e];
        sf[e] = beta[c_lane + e] - sh_mean[g] * sc[e];
    }
    const long img = (long)n * rpi;
    SlabWalk w(s, slot);
#pragma unroll
    for (int j = 0; j < kSlabV; ++j) {
        if (w.ok()) {
            float v[8];
            unpack8(cx[j], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float z = v[e] * sc[e] + sf[e]; v[e] = SILU ? silu_f(z) : z; }
            const long orow = out_compact ? (long)n * s.P + w.pi : img + w.row();
            *reinterpret_cast<u32x4_t*>(y + orow * s.ld + c_lane) = pack8(v);
        }
        w.next();
    }
}

// LDS (dynamic): xs [kSlabV][kT] u32x4 (64 KiB) | d0 [kSlabV][kT] u32x4 (64 KiB) | red [2][kT * 8] f32 (32 KiB -> aliased: see below)
// red is only live between the two phases' register / LDS traffic of the SAME data, so it gets its own 16 KiB: one array at a time.
constexpr int kSlabBwdSmem = 2 * kSlabV * kT * 16 + kT * 8 * 4;

template <bool SILU, int SETS, bool EXTRA>
__global__ __launch_bounds__(kT, 1) void gn_slab_bwd_kernel(BwdArgs a, SlabShape s) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    u32x4_t* xs = reinterpret_cast<u32x4_t*>(smem_raw);
    u32x4_t* d0s = xs + kSlabV * kT;
    float* red = reinterpret_cast<float*>(d0s + kSlabV * kT);     // [kT * 8]
    __shared__ float chs[4 * kSlabMaxCs];                          // [set*2 + {0: sum dz, 1: sum dz xhat}][Cs]
    __shared__ float sh_m[4 * kMaxG];                              // [set*2 + {S1, S2}][Gs] / cnt
    const int tid = threadIdx.x, n = blockIdx.y, c0 = blockIdx.x * s.Cs, g0 = blockIdx.x * s.Gs;
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    const bool active = slot < s.ppi;
    const long rpi = (long)(s.H + 2) * (s.W + 2);
    const int cl = active ? cc * 8 : 0;                            // first channel of the lane inside the slice
    const int c_lane = c0 + cl;
    const int gl_lo = cl / s.cpg;
    unsigned hi_mask = 0;
    float ga[8], be[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ga[e] = a.gamma[c_lane + e]; be[e] = a.beta[c_lane + e];
        if ((cl + e) / s.cpg != gl_lo) hi_mask |= 1u << e;
    }
    const int gl_hi = gl_lo + 1 < s.Gs ? gl_lo + 1 : gl_lo;
    const float rs0 = a.rstd[(long)n * s.G + g0 + gl_lo], rs1 = a.rstd[(long)n * s.G + g0 + gl_hi];
    const float mr0 = a.mean[(long)n * s.G + g0 + gl_lo] * rs0, mr1 = a.mean[(long)n * s.G + g0 + gl_hi] * rs1;
    auto rs = [&](int e) { return ((hi_mask >> e) & 1) ? rs1 : rs0; };
    auto mr = [&](int e) { return ((hi_mask >> e) & 1) ? mr1 : mr0; };
    const bool second = a.dx2 != nullptr && c_lane >= a.split_c;
    bf16_t* const obase = second ? a.dx2 + (c_lane - a.split_c) : a.dx + c_lane;
    const int ostride = a.dx2 ? (second ? s.ld - a.split_c : a.split_c) : s.ld;
    const bool oacc = second && a.accumulate2;
    const bf16_t* xb = a.x + (long)n * rpi * s.ldx + c_lane;
    auto dy_ptr = [&](int k, const SlabWalk& w) {
        const int n2 = k * a.nx + n;
        const long drow = a.dy_compact ? (long)n2 * s.P + w.pi : (long)n2 * rpi + w.row();
        return a.dy + drow * s.ld + c_lane;
    };
    // ---------------- phase 1: load everything once; x and set 0 wait in LDS, set 1 in registers
    u32x4_t cd1[kSlabV];
    float c12[2 * SETS][8];
#pragma unroll
    for (int q = 0; q < 2 * SETS; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) c12[q][e] = 0.f;
    {
        SlabWalk w(s, slot);
#pragma unroll
        for (int b = 0; b < kSlabV / 4; ++b) {
            u32x4_t hx[4], h0[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const bool ok = active && w.ok();
                hx[j] = ok ? *reinterpret_cast<const u32x4_t*>(xb + w.row() * s.ldx) : u32x4_t{0u, 0u, 0u, 0u};
                h0[j] = ok ? *reinterpret_cast<const u32x4_t*>(dy_ptr(0, w)) : u32x4_t{0u, 0u, 0u, 0u};
                if constexpr (SETS == 2) cd1[b * 4 + j] = ok ? *reinterpret_cast<const u32x4_t*>(dy_ptr(1, w)) : u32x4_t{0u, 0u, 0u, 0u};
                w.next();
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8], xh[8], dsl[8], d[8];
                unpack8(hx[j], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = v[e] * rs(e) - mr(e);
                    dsl[e] = SILU ? dsilu_f(xh[e] * ga[e] + be[e]) : 1.f;
                }
                unpack8(h0[j], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float dz = d[e] * dsl[e]; c12[0][e] += dz; c12[1][e] += dz * xh[e]; }
                if constexpr (SETS == 2) {
                    unpack8(cd1[b * 4 + j], d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float dz = d[e] * dsl[e]; c12[2][e] += dz; c12[3][e] += dz * xh[e]; }
                }
                xs[(b * 4 + j) * kT + tid] = hx[j];
                d0s[(b * 4 + j) * kT + tid] = h0[j];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 2 * SETS; ++q) {
        float one[1][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) one[0][e] = c12[q][e];
        slab_reduce<1>(one, active, slot, cc, s, red, chs + q * s.Cs);
    }
    // the block owns its (sample, channel slice): dgamma / dbeta contributions and group statistics come straight from chs
    for (int i = tid; i < 2 * SETS * s.Cs; i += kT) {
        const int q = i / s.Cs, c = i - q * s.Cs;
        const int set = ((q >> 1) * a.nx + n) / a.set_images;
        atomicAdd(((q & 1) ? a.dgamma : a.dbeta) + (long)set * a.set_stride + c0 + c, chs[q * s.Cs + c]);
    }
    if (tid < 2 * SETS * s.Gs) {
        const int q = tid / s.Gs, g = tid - q * s.Gs;
        double t = 0;
        for (int c = g * s.cpg; c < (g + 1) * s.cpg; ++c) t += (double)chs[q * s.Cs + c] * a.gamma[c0 + c];
        sh_m[q * s.Gs + g] = (float)(t / ((double)s.P * s.cpg));
    }
    __syncthreads();
    // ---------------- phase 2
    float cs[SETS][8];
#pragma unroll
    for (int k = 0; k < SETS; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[k][e] = 0.f;
    if (active) {
        float m1[SETS][2], m2[SETS][2];
#pragma unroll
        for (int k = 0; k < SETS; ++k) {
            m1[k][0] = sh_m[(2 * k) * s.Gs + gl_lo]; m2[k][0] = sh_m[(2 * k + 1) * s.Gs + gl_lo];
            m1[k][1] = sh_m[(2 * k) * s.Gs + gl_hi]; m2[k][1] = sh_m[(2 * k + 1) * s.Gs + gl_hi];
        }
        struct Extra { u32x4_t a[SETS], b[SETS], c[SETS]; };
        auto load_extra = [&](const SlabWalk& w, Extra& o) {
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                const long orow = (long)(k * a.nx + n) * rpi + w.row();
                o.a[k] = a.accum ? *reinterpret_cast<const u32x4_t*>(a.accum + orow * s.ld + c_lane) : u32x4_t{0u, 0u, 0u, 0u};
                o.b[k] = a.accum2 ? *reinterpret_cast<const u32x4_t*>(a.accum2 + orow * s.ld + c_lane) : u32x4_t{0u, 0u, 0u, 0u};
                o.c[k] = oacc ? *reinterpret_cast<const u32x4_t*>(obase + orow * ostride) : u32x4_t{0u, 0u, 0u, 0u};
            }
        };
        SlabWalk w(s, slot);
        Extra nxt = {};
        if constexpr (EXTRA) { if (w.ok()) load_extra(w, nxt); }
#pragma unroll
        for (int j = 0; j < kSlabV; ++j) {
            if (w.ok()) {
                const Extra cur = nxt;
                const SlabWalk w0 = w;
                w.next();
                if constexpr (EXTRA) { if (w.ok()) load_extra(w, nxt); }
                float v[8], xh[8], dsl[8];
                unpack8(xs[j * kT + tid], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    xh[e] = v[e] * rs(e) - mr(e);
                    dsl[e] = (SILU ? dsilu_f(xh[e] * ga[e] + be[e]) : 1.f) * ga[e];
                }
#pragma unroll
                for (int k = 0; k < SETS; ++k) {
                    const long orow = (long)(k * a.nx + n) * rpi + w0.row();
                    float d[8], o[8];
                    unpack8(k == 0 ? d0s[j * kT + tid] : cd1[j], d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const bool h = (hi_mask >> e) & 1;
                        const float t = rs(e) * (d[e] * dsl[e] - (h ? m1[k][1] : m1[k][0]) - xh[e] * (h ? m2[k][1] : m2[k][0]));
                        cs[k][e] += t;
                        o[e] = t;
                    }
                    if constexpr (EXTRA) {
                        float r1[8], r2[8], r3[8];
                        unpack8(cur.a[k], r1); unpack8(cur.b[k], r2); unpack8(cur.c[k], r3);
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = o[e] + r1[e] + r2[e] + r3[e];
                    }
                    *reinterpret_cast<u32x4_t*>(obase + orow * ostride) = pack8(o);
                }
            } else {
                w.next();
            }
        }
    }
    if (a.colsum) {                                                // the block owns these (sample, channel) sums entirely
#pragma unroll
        for (int k = 0; k < SETS; ++k) {
            float one[1][8];
#pragma unroll
            for (int e = 0; e < 8; ++e) one[0][e] = cs[k][e];
            slab_reduce<1>(one, active, slot, cc, s, red, chs);
            for (int i = tid; i < s.Cs; i += kT) atomicAdd(a.colsum + (long)(k * a.nx + n) * a.colsum_ld + c0 + i, chs[i]);
            __syncthreads();
        }
    }
}

// slice width for a site, 0 when the slab form does not cover it
int slab_slice(int H, int W, int C, int G, int N, SlabShape& s) {
    if (H <= 0 || W <= 0 || C <= 0 || G <= 0 || G > kMaxG || C % G || C % 8 || N <= 0) return 0;
    const int P = H * W, cpg = C / G;
    if (P > 1024 || cpg < 4) return 0;
    int best = 0;
    for (int k = G; k >= 1; --k) {                                  // slices of k whole groups, widest first
        if (G % k) continue;
        const int Cs = k * cpg;
        if (Cs % 8 || Cs > kSlabMaxCs) continue;
        const int lpp = Cs / 8, ppi = kT / lpp;
        if ((P + ppi - 1) / ppi > kSlabV) continue;
        bool two = true;                                            // a lane's 8 channels: at most two adjacent groups
        for (int cc = 0; cc < lpp; ++cc) two = two && ((cc * 8 + 7) / cpg <= (cc * 8) / cpg + 1);
        if (!two) continue;
        if (Cs < 32) continue;                                      // 64-B row segments at least (measured: 48-B slices of a 768-channel site run 1.6x slower than the two-pass kernels)
        best = Cs;
        if ((long)(C / Cs) * N >= 128) break;                       // enough blocks: keep the widest such slice
    }
    if (!best) return 0;
    s.H = H; s.W = W; s.P = P; s.C = C; s.G = G; s.cpg = cpg;
    s.Cs = best; s.Gs = best / cpg; s.lpp = best / 8; s.ppi = kT / s.lpp; s.V = (P + s.ppi - 1) / s.ppi;
    s.ld = C; s.ldx = C;
    return best;
}

// ---------------------------------------------------------------------------------------------------------------------
bool make_shape2(int H, int W, int C, int G, int N, Shape2& s) {
    if (H <= 0 || W <= 0 || C <= 0 || G <= 0 || G > kMaxG || C % G || C % 8 || C > kMaxC || N <= 0) return false;
    s.H = H; s.W = W; s.C = C; s.G = G; s.cpg = C / G; s.ld = C; s.ldx = C; s.P = H * W;
    if (s.cpg < 4) return false;
    s.lpp = C / 8;
    if (s.lpp > kT) return false;
    s.ppi = kT / s.lpp;
    for (int cc = 0; cc < s.lpp; ++cc)                                // a lane's 8 channels: at most two adjacent groups
        if ((cc * 8 + 7) / s.cpg > (cc * 8) / s.cpg + 1) return false;
    const int iters_all = (s.P + s.ppi - 1) / s.ppi;                  // 16-B vectors per lane for a whole sample on ONE block
    const int need = (iters_all + kV - 1) / kV;                       // blocks per sample that keep a whole sample on chip
    int want = kMaxBlocks / (N < kMaxBlocks ? N : kMaxBlocks);        // ... and enough blocks to cover the chip
    if (want < 1) want = 1;
    if (want > iters_all) want = iters_all;
    int bps = need > want ? need : want;
    if (bps > kMaxBlocks) bps = kMaxBlocks;                           // longer runs: head on chip, tail re-read
    int run = (s.P + bps - 1) / bps;
    run = (run + s.ppi - 1) / s.ppi * s.ppi;
    bps = (s.P + run - 1) / run;
    s.bps = bps; s.run_px = run;
    s.spr = kMaxBlocks / bps; if (s.spr > N) s.spr = N; if (s.spr < 1) s.spr = 1;
    s.rounds = (N + s.spr - 1) / s.spr;
    return true;
}

}  // namespace

// words (floats) of zero-initialised workspace the two-phase kernels need for n samples
long siss_gn2p_words(int n) { return (long)n * (kCtrInts + kRep * kMaxStat * 2 * 2); }

// Returns SISS_OK, an error, or -1 when the shape is not covered (the caller then takes the two-pass kernels).
// ws: zero-filled workspace of siss_gn2p_words(N) floats, 16-B aligned; left zero-filled.
int siss_gn2p_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, float* ws,
                  int N, int H, int W, int C, int G, float eps, int silu, int out_compact, int ldx, void* stream) {
    Shape2 s;
    if (!make_shape2(H, W, C, G, N, s)) return -1;
    if (ldx) s.ldx = ldx;
    int* ctr = reinterpret_cast<int*>(ws);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(ws + (long)N * kCtrInts);
    static unsigned char a1[kMaxDevices], a2[kMaxDevices];
    const dim3 grid(s.bps * s.spr);
    if (silu) {
        if (siss_ensure_smem((const void*)gn2p_fwd_kernel<true>, kFwdSmem, a1) != SISS_OK) return SISS_ERR_LAUNCH;
        gn2p_fwd_kernel<true><<<grid, kT, kFwdSmem, (hipStream_t)stream>>>((const bf16_t*)x, gamma, beta, s, eps, out_compact, (bf16_t*)y, mean, rstd, N, ctr, acc);
    } else {
        if (siss_ensure_smem((const void*)gn2p_fwd_kernel<false>, kFwdSmem, a2) != SISS_OK) return SISS_ERR_LAUNCH;
        gn2p_fwd_kernel<false><<<grid, kT, kFwdSmem, (hipStream_t)stream>>>((const bf16_t*)x, gamma, beta, s, eps, out_compact, (bf16_t*)y, mean, rstd, N, ctr, acc);
    }
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

// The slab kernels (small sites): SISS_OK / error, or -1 when the site is not covered.
int siss_gn_slab_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int N, int H,
                     int W, int C, int G, float eps, int silu, int out_compact, int ldx, void* stream) {
    SlabShape s;
    if (!slab_slice(H, W, C, G, N, s)) return -1;
    if (ldx) s.ldx = ldx;
    static unsigned char a1[kMaxDevices], a2[kMaxDevices];
    const dim3 grid(C / s.Cs, N);
    constexpr int smem = 2 * kT * 8 * 4;
    siss_count_dispatch(SISS_K_GN_SLAB);
    if (silu) {
        if (siss_ensure_smem((const void*)gn_slab_fwd_kernel<true>, smem, a1) != SISS_OK) return SISS_ERR_LAUNCH;
        gn_slab_fwd_kernel<true><<<grid, kT, smem, (hipStream_t)stream>>>((const bf16_t*)x, gamma, beta, s, eps, out_compact, (bf16_t*)y, mean, rstd);
    } else {
        if (siss_ensure_smem((const void*)gn_slab_fwd_kernel<false>, smem, a2) != SISS_OK) return SISS_ERR_LAUNCH;
        gn_slab_fwd_kernel<false><<<grid, kT, smem, (hipStream_t)stream>>>((const bf16_t*)x, gamma, beta, s, eps, out_compact, (bf16_t*)y, mean, rstd);
    }
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

int siss_gn_slab_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, void* dx, const void* accum, const void* accum2, void* dx2, int split_c,
                     int accumulate2, float* dgamma, float* dbeta, float* colsum, long colsum_ld, int n2, int nx,
                     int set_images, long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx,
                     void* stream) {
    SlabShape s;
    // backward: only up to 16 x 16 pixels (measured per site, B = 16: 8x8 38.8 -> 24.0 us, 16x16 40.9 -> 26.9 us, but 32x32
    // 40.7 -> 45.7 us: three tensors of 1024 pixels per block leave one block per CU and a serial walk of 8 vectors per lane)
    if (H * W > 256 || !slab_slice(H, W, C, G, nx, s)) return -1;
    if (ldx) s.ldx = ldx;
    BwdArgs a;
    a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
    a.accum = (const bf16_t*)accum; a.accum2 = (const bf16_t*)accum2; a.dx = (bf16_t*)dx; a.dx2 = (bf16_t*)dx2;
    a.dgamma = dgamma; a.dbeta = dbeta; a.colsum = colsum; a.colsum_ld = colsum_ld; a.set_stride = set_stride;
    a.split_c = split_c; a.accumulate2 = accumulate2; a.nx = nx; a.dy_compact = dy_compact; a.set_images = set_images;
    const bool extra = accum || accum2 || (dx2 && accumulate2);
    const dim3 grid(C / s.Cs, nx);
    siss_count_dispatch(SISS_K_GN_SLAB);
    hipStream_t st = (hipStream_t)stream;
    static unsigned char att[8][kMaxDevices];
#define GN_SLAB_BWD(SILU, SETS, EXTRA, SLOT)                                                                               \
    do {                                                                                                                   \
        if (siss_ensure_smem((const void*)gn_slab_bwd_kernel<SILU, SETS, EXTRA>, kSlabBwdSmem, att[SLOT]) != SISS_OK) return SISS_ERR_LAUNCH; \
        gn_slab_bwd_kernel<SILU, SETS, EXTRA><<<grid, kT, kSlabBwdSmem, st>>>(a, s);                                     \
    } while (0)
    if (n2 == nx) {
        if (silu) { if (extra) GN_SLAB_BWD(true, 1, true, 0); else GN_SLAB_BWD(true, 1, false, 1); }
        else      { if (extra) GN_SLAB_BWD(false, 1, true, 2); else GN_SLAB_BWD(false, 1, false, 3); }
    } else {
        if (silu) { if (extra) GN_SLAB_BWD(true, 2, true, 4); else GN_SLAB_BWD(true, 2, false, 5); }
        else      { if (extra) GN_SLAB_BWD(false, 2, true, 6); else GN_SLAB_BWD(false, 2, false, 7); }
    }
#undef GN_SLAB_BWD
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}

int siss_gn2p_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                  const float* rstd, void* dx, const void* accum, const void* accum2, void* dx2, int split_c,
                  int accumulate2, float* dgamma, float* dbeta, float* colsum, long colsum_ld, float* ws, int n2, int nx,
                  int set_images, long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx,
                  void* stream) {
    Shape2 s;
    if (!make_shape2(H, W, C, G, nx, s)) return -1;
    if (n2 / set_images > 2) return -1;                              // the block-level dgamma / dbeta accumulators hold two sets
    if (ldx) s.ldx = ldx;
    BwdArgs a;
    a.dy = (const bf16_t*)dy; a.x = (const bf16_t*)x; a.gamma = gamma; a.beta = beta; a.mean = mean; a.rstd = rstd;
    a.accum = (const bf16_t*)accum; a.accum2 = (const bf16_t*)accum2; a.dx = (bf16_t*)dx; a.dx2 = (bf16_t*)dx2;
    a.dgamma = dgamma; a.dbeta = dbeta; a.colsum = colsum; a.colsum_ld = colsum_ld; a.set_stride = set_stride;
    a.split_c = split_c; a.accumulate2 = accumulate2; a.nx = nx; a.dy_compact = dy_compact; a.set_images = set_images;
    int* ctr = reinterpret_cast<int*>(ws);
    unsigned long long* acc = reinterpret_cast<unsigned long long*>(ws + (long)nx * kCtrInts);
    const bool extra = accum || accum2 || (dx2 && accumulate2);
    const dim3 grid(s.bps * s.spr);
    hipStream_t st = (hipStream_t)stream;
    static unsigned char att[8][kMaxDevices];
#define GN2P_BWD(SILU, SETS, EXTRA, SLOT)                                                                                  \
    do {                                                                                                                   \
        if (siss_ensure_smem((const void*)gn2p_bwd_kernel<SILU, SETS, EXTRA>, kBwdSmem, att[SLOT]) != SISS_OK) return SISS_ERR_LAUNCH; \
        gn2p_bwd_kernel<SILU, SETS, EXTRA><<<grid, kT, kBwdSmem, st>>>(a, s, ctr, acc);                                  \
    } while (0)
    const int sets = n2 / nx;
    if (sets == 1) {
        if (silu) { if (extra) GN2P_BWD(true, 1, true, 0); else GN2P_BWD(true, 1, false, 1); }
        else      { if (extra) GN2P_BWD(false, 1, true, 2); else GN2P_BWD(false, 1, false, 3); }
    } else {
        if (silu) { if (extra) GN2P_BWD(true, 2, true, 4); else GN2P_BWD(true, 2, false, 5); }
        else      { if (extra) GN2P_BWD(false, 2, true, 6); else GN2P_BWD(false, 2, false, 7); }
    }
#undef GN2P_BWD
    return hipGetLastError() == hipSuccess ? SISS_OK : SISS_ERR_LAUNCH;
}
