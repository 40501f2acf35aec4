// One-launch "slab" GroupNorm (+SiLU) forward / backward for the small sites (at most 32 x 32 pixels).
//
// groupnorm.hip needs two passes over its inputs (statistics, then apply) because the statistics of a (sample, group)
// cover the whole image.  At the 8x8 / 16x16 / 32x32 levels a block can hold ALL pixels of a channel slice of whole groups
// of one sample on chip, so the statistics need no other block: one launch instead of two, no atomics for the statistics,
// every byte read once, bitwise deterministic (fixed-order LDS folds).
// (The two-phase persistent form for the LARGE sites -- per-sample barrier, fixed-point accumulators -- measured 2x slower
// than the two-pass kernels and lives in tools/probes/groupnorm2p.hip; docs/experiments.md.)
#include "common.h"
#include <type_traits>

namespace {

constexpr int kT = 512;                 // threads per block (8 waves: 256 VGPRs per lane)
constexpr int kMaxG = 32;

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

struct BwdArgs {
    const bf16_t* dy; const bf16_t* x; const float* gamma; const float* beta; const float* mean; const float* rstd;
    const bf16_t* accum; const bf16_t* accum2; bf16_t* dx; bf16_t* dx2;
    float* dgamma; float* dbeta; float* colsum;
    long colsum_ld, set_stride;
    int split_c, accumulate2, nx, dy_compact, set_images;
};


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

}  // namespace

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
