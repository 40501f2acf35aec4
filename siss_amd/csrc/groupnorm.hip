// GroupNorm (+ optional SiLU) forward / backward on the padded-NHWC layout.
//
// HBM-bound.  A block owns a run of pixels of ONE sample and ALL channels: every lane moves
// 16 B (8 consecutive bf16 channels of one pixel), so a pixel row (C*2 bytes) is read by C/8
// adjacent lanes -- fully coalesced -- and per-group statistics are formed by folding the 8
// per-lane channel accumulators into LDS by group id.  Statistics take two launches (partial
// slab, then fold in the consumer's prologue): no atomics on the forward path, deterministic.
//
//   fwd : stats(x) -> partial[n][chunk][g] = (sum, sumsq)
//         apply    -> y = act(gamma * (x - mean) * rstd + beta)        act = SiLU or identity
//   bwd : stats(dy, x) -> partial[n2][chunk][g] = (S1, S2);  dgamma/dbeta per gradient set
//         apply    -> dx = rstd * (dz*gamma - S1/cnt - xhat * S2/cnt) (+ accum) ; optional
//                     per-(sample, channel) column sums of dx (time-embedding / bias gradient)
// Backward takes n2 = sets * B cotangent samples against B saved samples (x index = n2 % Bx):
// the dual-cotangent backward of the SISS step.
// Output / cotangent rows may be "compact" ([N][H*W][C], no halo) for the attention block.
#include "common.h"

namespace {

constexpr int kStatsDepth = 2;          // pixels in flight per lane in the backward statistics pass (3: -9 %, 4: -20 %: measured)
constexpr int kThreads = 256;
constexpr int kMaxG = 32;
constexpr int kMaxC = 1024;

// Wide tensors (C > kMaxC: the SD UNet's 1280..2560-channel concats) are cut into channel SLICES of whole
// groups (blockIdx.z): C / G below are the slice's, ld / Gf the full tensor's.
struct GNShape {
    int H, W, C, G, cpg, lpp, ppi;     // lanes per pixel (C/8), pixels per iteration
    int chunk_px, nchunks;             // interior pixels per block, blocks per sample
    int ld, Gf, nslices;               // row stride (= full channel count), full group count, channel slices
    int ldx;                           // row stride of the INPUT x (> ld when x is a column view of a concat buffer)
};

__device__ __forceinline__ long compact_row(int n, int pi, int H, int W) { return (long)n * H * W + pi; }

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

// Walks the interior pixels pi = p0+slot, +ppi, ... of one image; keeps (y, x) incrementally so the
// hot loop has no integer division.  row() = padded row index within the image.
struct PixelWalk {
    int pi, p1, x, W, ppi, dx, rw, drw;
    __device__ __forceinline__ PixelWalk(const GNShape& s, int chunk, int slot) {
        const int p0 = chunk * s.chunk_px;
        p1 = p0 + s.chunk_px; p1 = p1 < s.H * s.W ? p1 : s.H * s.W;
        pi = p0 + slot; W = s.W; ppi = s.ppi;
        const int y = pi / W; x = pi - y * W;
        const int dy = ppi / W; dx = ppi - dy * W;
        rw = (y + 1) * (W + 2) + (x + 1);              // padded row of (y, x) within the image ...
        drw = dy * (W + 2) + dx;                        // ... and what a step adds to it (+ 2 halo pixels when x wraps)
    }
    __device__ __forceinline__ bool ok() const { return pi < p1; }
    __device__ __forceinline__ int row() const { return rw; }
    __device__ __forceinline__ void next() {
        pi += ppi; x += dx; rw += drw;
        if (x >= W) { x -= W; rw += 2; }
    }
};

// Addresses are a wave-uniform 64-bit base (sample, channel slice: SGPRs) + a 32-bit per-lane byte offset = row * row bytes
// (v_mul_u32_u24: full rate; rows < 2^24, row bytes < 2^24, one sample < 4 GiB) + the lane's channel offset: two vector
// instructions per access where pointer arithmetic in 64 bits cost two quarter-rate v_mul_lo_u32, a v_mad_u64_u32 and
// 64-bit adds -- a quarter of the vector work of these VALU-co-limited kernels.
__device__ __forceinline__ unsigned boff(int row, unsigned row_bytes, unsigned lane_bytes) { return __umul24(row, row_bytes) + lane_bytes; }
__device__ __forceinline__ u32x4_t ld16(const bf16_t* base, unsigned off) {
    return *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(base) + off);
}
__device__ __forceinline__ void st16(bf16_t* base, unsigned off, u32x4_t v) {
    *reinterpret_cast<u32x4_t*>(reinterpret_cast<char*>(base) + off) = v;
}

// Block-level reduction over the pixel slots WITHOUT atomics (deterministic): every active thread parks its
// 8 per-channel partial sums in LDS as red[slot][C], then thread c < C adds the slots of channel c.
// red must hold 2048 floats (ppi * C <= 256 * 8).  out[c] is valid for all threads after the call.
__device__ __forceinline__ void reduce_slots(const float (&v)[8], bool active, int slot, int cc, const GNShape& s,
                                             float* red, float* out) {
    if (active) {
        float* dst = red + slot * s.C + cc * 8;
        *reinterpret_cast<f32x4_t*>(dst) = f32x4_t{v[0], v[1], v[2], v[3]};
        *reinterpret_cast<f32x4_t*>(dst + 4) = f32x4_t{v[4], v[5], v[6], v[7]};
    }
    __syncthreads();
    for (int c = threadIdx.x; c < s.C; c += kThreads) {
        float a = 0.f;
        for (int sl = 0; sl < s.ppi; ++sl) a += red[sl * s.C + c];
        out[c] = a;
    }
    __syncthreads();
}

// ---------------------------------------------------------------- forward: partial statistics
__global__ __launch_bounds__(kThreads) void gn_stats_kernel(const bf16_t* __restrict__ x, GNShape s,
                                                            float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float red[2048];
    __shared__ float ch_a[kMaxC], ch_b[kMaxC];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int c0 = blockIdx.z * s.C;
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    const bool active = slot < s.ppi;
    float a[8] = {}, b[8] = {};
    if (active) {
        const bf16_t* base = x + (long)n * (s.H + 2) * (s.W + 2) * s.ldx + c0;
        const unsigned xb = s.ldx * 2, lb = cc * 16;
        PixelWalk w(s, chunk, slot);
        while (w.ok()) {                       // two pixels per trip: both loads are in flight together
            const u32x4_t r0 = ld16(base, boff(w.row(), xb, lb));
            w.next();
            const bool two = w.ok();
            u32x4_t r1 = u32x4_t{0u, 0u, 0u, 0u};
            if (two) { r1 = ld16(base, boff(w.row(), xb, lb)); w.next(); }
            float v[8], u[8];
            unpack8(r0, v); unpack8(r1, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) { a[e] += v[e] + u[e]; b[e] += v[e] * v[e] + u[e] * u[e]; }
        }
    }
    reduce_slots(a, active, slot, cc, s, red, ch_a);
    reduce_slots(b, active, slot, cc, s, red, ch_b);
    if (tid < 2 * s.G) {
        const int g = tid >> 1;
        const float* src = (tid & 1) ? ch_b : ch_a;
        float t = 0.f;
        for (int c = g * s.cpg; c < (g + 1) * s.cpg; ++c) t += src[c];
        partial[(((long)n * s.nslices + blockIdx.z) * s.nchunks + chunk) * 2 * s.G + tid] = t;
    }
}

// Sum the [nchunks][2G] partial slab of one sample with ALL 256 threads: thread (j = tid>>6, e = tid&63)
// adds chunks j, j+4, ... of entry e (coalesced 256-B rows, independent loads), then the four partial
// sums meet in LDS.  (A one-thread-per-group serial walk over the chunks cost ~60 us of dependent L2
// latency in every block.)  Result: red[e] for e < 2G.
__device__ __forceinline__ void fold_slab(const float* __restrict__ slab, int nchunks, int G, float (*red)[64]) {
    const int tid = threadIdx.x, j = tid >> 6, e = tid & 63;
    float a = 0.f;
    if (e < 2 * G) {
#pragma unroll 4
        for (int c = j; c < nchunks; c += 4) a += slab[(long)c * 2 * G + e];
    }
    red[j][e] = a;
    __syncthreads();
}

__device__ __forceinline__ void fold_stats(const float* __restrict__ partial, const GNShape& s, int n,
                                           float eps, float* sh_mean, float* sh_rstd,
                                           float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                           bool write) {
    __shared__ float red[4][64];
    fold_slab(partial + ((long)n * s.nslices + blockIdx.z) * s.nchunks * 2 * s.G, s.nchunks, s.G, red);
    const int tid = threadIdx.x;
    if (tid < s.G) {
        const double a = (double)red[0][2 * tid] + red[1][2 * tid] + red[2][2 * tid] + red[3][2 * tid];
        const double b = (double)red[0][2 * tid + 1] + red[1][2 * tid + 1] + red[2][2 * tid + 1] + red[3][2 * tid + 1];
        const double cnt = (double)s.H * s.W * s.cpg;
        const double m = a / cnt;
        double var = b / cnt - m * m;
        var = var > 0 ? var : 0;
        const float r = (float)(1.0 / sqrt(var + (double)eps));
        sh_mean[tid] = (float)m; sh_rstd[tid] = r;
        if (write) {
            const long go = (long)n * s.Gf + blockIdx.z * s.G + tid;
            mean_out[go] = (float)m; rstd_out[go] = r;
        }
    }
    __syncthreads();
}

template <bool SILU>
__global__ __launch_bounds__(kThreads) void gn_apply_kernel(
    const bf16_t* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ partial, GNShape s, float eps, int out_compact, bf16_t* __restrict__ y,
    float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    __shared__ float sh_mean[kMaxG], sh_rstd[kMaxG];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    if (partial) fold_stats(partial, s, n, eps, sh_mean, sh_rstd, mean_out, rstd_out, chunk == 0);
    else {
        // mean / rstd were finalised by gn_qstats_finalize_kernel (statistics from the producing convolution's epilogue)
        if (tid < s.G) {
            const long go = (long)n * s.Gf + blockIdx.z * s.G + tid;
            sh_mean[tid] = mean_out[go]; sh_rstd[tid] = rstd_out[go];
        }
        __syncthreads();
    }
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    if (slot >= s.ppi) return;
    const int c0 = blockIdx.z * s.C;
    float sc[8], sf[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = cc * 8 + e, g = c / s.cpg;
        sc[e] = sh_rstd[g] * gamma[c0 + c];
        sf[e] = beta[c0 + c] - sh_mean[g] * sc[e];
    }
    const long img = (long)n * (s.H + 2) * (s.W + 2);
    const bf16_t* base = x + img * s.ldx + c0;
    const unsigned xb = s.ldx * 2, ob = s.ld * 2, lb = cc * 16;
    bf16_t* const ybase = y + c0 + (out_compact ? (long)n * s.H * s.W : img) * s.ld;
    auto out_off = [&](const PixelWalk& w) { return boff(out_compact ? w.pi : w.row(), ob, lb); };
    PixelWalk w(s, chunk, slot);
    while (w.ok()) {
        const u32x4_t r0 = ld16(base, boff(w.row(), xb, lb));
        const unsigned o0 = out_off(w);
        w.next();
        const bool two = w.ok();
        u32x4_t r1 = u32x4_t{0u, 0u, 0u, 0u};
        unsigned o1 = 0;
        if (two) { r1 = ld16(base, boff(w.row(), xb, lb)); o1 = out_off(w); w.next(); }
        float v[8], u[8];
        unpack8(r0, v); unpack8(r1, u);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float z0 = v[e] * sc[e] + sf[e], z1 = u[e] * sc[e] + sf[e];
            v[e] = SILU ? silu_f(z0) : z0;
            u[e] = SILU ? silu_f(z1) : z1;
        }
        st16(ybase, o0, pack8(v));
        if (two) st16(ybase, o1, pack8(u));
    }
}

// Statistics handed over by the producing 3x3 convolution(s) (NTParams::qstats): block (sample n, 4 groups), ONE WAVE per
// group walks the sample's (half tile, quad) entries four independent loads at a time; sums in double from the first
// addition on.  The normalised tensor may be the channel concat of two producers' outputs: quads [0, qa) come from qsA (qa
// quads per entry), the rest from qsB (qb per entry).
__global__ __launch_bounds__(kThreads) void gn_qstats_finalize_kernel(
    const float* __restrict__ qsA, int qa, const float* __restrict__ qsB, int qb, int H, int W, int G, int cpg, float eps,
    float* __restrict__ mean, float* __restrict__ rstd) {
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int g = blockIdx.y * 4 + (tid >> 6);
    if (g >= G) return;                                    // whole waves only: no barrier below
    const long rpi = (long)(H + 2) * (W + 2);
    const int t0 = (int)(n * rpi / kQsTileRows), t1 = (int)(((n + 1) * rpi - 1) / kQsTileRows);
    const int qpg = cpg >> 2;
    const int items = (t1 - t0 + 1) * 2 * qpg;
    auto entry = [&](int it) -> float2 {
        if (it >= items) return float2{0.f, 0.f};
        const int hl = it / qpg, qi = it - hl * qpg;
        const int t = t0 + (hl >> 1), h = hl & 1;
        const int slot = (int)((long)t * kQsTileRows / rpi) == n ? 0 : 1;       // the tile starts in image n, or in n - 1
        const int q = g * qpg + qi;
        const float* src = q < qa ? qsA : qsB;
        const int stride = q < qa ? qa : qb, qq = q < qa ? q : q - qa;
        return *reinterpret_cast<const float2*>(src + (((long)(2 * t + h) * 2 + slot) * stride + qq) * 2);
    };
    double a = 0.0, b = 0.0;
    for (int it = lane; it < items; it += 256) {
        const float2 e0 = entry(it), e1 = entry(it + 64), e2 = entry(it + 128), e3 = entry(it + 192);
        a += ((double)e0.x + (double)e1.x) + ((double)e2.x + (double)e3.x);
        b += ((double)e0.y + (double)e1.y) + ((double)e2.y + (double)e3.y);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { a += __shfl_xor(a, o); b += __shfl_xor(b, o); }
    if (lane == 0) {
        const double cnt = (double)H * W * cpg;
        const double m = a / cnt;
        double var = b / cnt - m * m;
        var = var > 0 ? var : 0;
        mean[(long)n * G + g] = (float)m;
        rstd[(long)n * G + g] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

// The same (half tile, image slot, quad) entries for a tensor that did NOT come out of the persistent 3x3 kernel (conv_in, the
// stride-2 downsample, the sub-pixel upsample: the generic kernels form no statistics): one read of the tensor, block
// (tile t, half h) over its <= 128 flat padded rows, lane = (row in flight, 16-B channel chunk = two quads).  Halo rows are zero
// (every producer masks them) and add nothing.  A skip tensor is normalised twice (alone on the way down, as one half of a concat
// on the way up): the engine keeps these entries, and the second GroupNorm has no statistics pass at all.
__global__ __launch_bounds__(kThreads) void gn_quad_stats_kernel(const bf16_t* __restrict__ x, long ld, int M, int N, int rpi,
                                                                 float* __restrict__ qs) {
    __shared__ float red[8][kThreads];
    const int t = blockIdx.x >> 1, h = blockIdx.x & 1, tid = threadIdx.x;
    const int r0 = t * kQsTileRows + h * kQsHalfRows;
    int r1 = r0 + (h ? kQsTileRows - kQsHalfRows : kQsHalfRows);
    r1 = r1 < M ? r1 : M;
    const int bnd = (t * kQsTileRows / rpi + 1) * rpi;      // first row of the tile's second image (slot 1)
    const int lpr = N >> 3, par = kThreads / lpr;           // lanes per row, rows in flight
    const int rr = tid / lpr, cc = tid - rr * lpr;
    float a[8];                                             // [quad j][slot][sum, sum of squares]
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = 0.f;
    if (rr < par) {
        const bf16_t* src = x + cc * 8;
        for (int r = r0 + rr; r < r1; r += par) {
            const u32x4_t v4 = *reinterpret_cast<const u32x4_t*>(src + (long)r * ld);
            float v[8];
            unpack8(v4, v);
            const float w1 = r >= bnd ? 1.f : 0.f, w0 = 1.f - w1;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const float s1 = (v[4 * j] + v[4 * j + 1]) + (v[4 * j + 2] + v[4 * j + 3]);
                const float s2 = (v[4 * j] * v[4 * j] + v[4 * j + 1] * v[4 * j + 1]) + (v[4 * j + 2] * v[4 * j + 2] + v[4 * j + 3] * v[4 * j + 3]);
                a[4 * j] += w0 * s1; a[4 * j + 1] += w0 * s2;
                a[4 * j + 2] += w1 * s1; a[4 * j + 3] += w1 * s2;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) red[i][tid] = a[i];
    __syncthreads();
    if (tid < lpr * 4) {                                    // one thread per (chunk, quad j, slot)
        const int c = tid >> 2, j = (tid >> 1) & 1, slot = tid & 1;
        float s1 = 0.f, s2 = 0.f;
        for (int k = 0; k < par; ++k) {
            s1 += red[4 * j + 2 * slot][k * lpr + c];
            s2 += red[4 * j + 2 * slot + 1][k * lpr + c];
        }
        const long e = ((long)(2 * t + h) * 2 + slot) * (N >> 2) + (2 * c + j);
        *reinterpret_cast<float2*>(qs + 2 * e) = float2{s1, s2};
    }
}

// ---------------------------------------------------------------- backward
// grid.y = nx (saved samples); each block handles the SETS cotangent samples n2 = set*nx + n that share
// saved sample n, so x is read once for both gradient sets.
template <bool SILU, int SETS>
__global__ __launch_bounds__(kThreads) void gn_bwd_stats_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
    GNShape s, int nx, int dy_compact, int set_images, long set_stride, float* __restrict__ partial,
    float* __restrict__ dgamma, float* __restrict__ dbeta) {
    __shared__ __attribute__((aligned(16))) float red[2048];
    __shared__ float ch1[kMaxC], ch2[kMaxC];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int c0 = blockIdx.z * s.C, g0 = blockIdx.z * s.G;
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    const bool active = slot < s.ppi;
    gamma += c0; beta += c0; dy += c0;
    float a1[SETS][8], a2[SETS][8];
#pragma unroll
    for (int k = 0; k < SETS; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) { a1[k][e] = 0.f; a2[k][e] = 0.f; }
    if (active) {
        float mr[8], rs[8], ga[8], be[8], ga2[8], be2[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e, g = c / s.cpg;
            rs[e] = rstd[(long)n * s.Gf + g0 + g]; mr[e] = mean[(long)n * s.Gf + g0 + g] * rs[e];
            ga[e] = gamma[c]; be[e] = beta[c];
            ga2[e] = -1.44269504088896f * ga[e]; be2[e] = -1.44269504088896f * be[e];
        }
        const long rpi = (long)(s.H + 2) * (s.W + 2);
        const bf16_t* xbase = x + (long)n * rpi * s.ldx + c0;
        const bf16_t* dbase[SETS];                    // cotangent sample n2 = k * nx + n of each set (dy already points at c0)
#pragma unroll
        for (int k = 0; k < SETS; ++k)
            dbase[k] = dy + (dy_compact ? (long)(k * nx + n) * s.H * s.W : (long)(k * nx + n) * rpi) * s.ld;
        const unsigned xb = s.ldx * 2, db = s.ld * 2, lb = cc * 16;
        // Software pipeline, TWO pixels deep: the loads of pixels i+1 and i+2 (x + one dy per set each) are in flight while
        // pixel i is consumed.  A launch is 512 blocks = 8 waves per CU; with one pixel ahead that was ~24 KB in flight per CU,
        // below what a 6 TB/s read stream needs at this latency (tools/probes/hbm_bw.hip: 6.1-6.4 TB/s for three read streams
        // with two loads in flight per lane, 4.0 TB/s measured for this kernel before).
        PixelWalk w(s, chunk, slot);
        constexpr int DEPTH = kStatsDepth;
        u32x4_t bx[DEPTH], bd[DEPTH][SETS];
        bool have[DEPTH];
        auto issue = [&](const PixelWalk& q, u32x4_t& ox, u32x4_t (&od)[SETS]) {
            ox = ld16(xbase, boff(q.row(), xb, lb));
            const unsigned doff = boff(dy_compact ? q.pi : q.row(), db, lb);
#pragma unroll
            for (int k = 0; k < SETS; ++k) od[k] = ld16(dbase[k], doff);
        };
#pragma unroll
        for (int b2 = 0; b2 < DEPTH; ++b2) {
            bx[b2] = u32x4_t{0u, 0u, 0u, 0u};
#pragma unroll
            for (int k = 0; k < SETS; ++k) bd[b2][k] = u32x4_t{0u, 0u, 0u, 0u};
            have[b2] = w.ok();
            if (have[b2]) { issue(w, bx[b2], bd[b2]); w.next(); }
        }
        while (have[0]) {
#pragma unroll
            for (int b2 = 0; b2 < DEPTH; ++b2) {
                if (!have[b2]) break;
                const u32x4_t rx = bx[b2];
                u32x4_t rd[SETS];
#pragma unroll
                for (int k = 0; k < SETS; ++k) rd[k] = bd[b2][k];
                have[b2] = w.ok();
                if (have[b2]) { issue(w, bx[b2], bd[b2]); w.next(); }
                // Vector-instruction diet (this pass streams 4.4 TB/s where the apply pass streams 5.6: every issue slot it frees
                // goes to address arithmetic and loads): SiLU'(z) = s (1 + z e s) with e = exp(-z), s = 1 / (1 + e) -- (1 - s) is
                // e s, and exp's argument is ONE fma of xhat against pre-scaled (-gamma log2 e, -beta log2 e); both sums of a set are
                // ONE fma each against products formed once per element (q1 = SiLU', q2 = SiLU' xhat) instead of a multiply, an add
                // and an fma per set: 19 slots per element and two sets where there were 22.
                float v[8], q1[8], q2[8];
                unpack8(rx, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float xh = v[e] * rs[e] - mr[e];
                    if (SILU) {
                        const float ex = __builtin_amdgcn_exp2f(xh * ga2[e] + be2[e]);
                        const float sg = __builtin_amdgcn_rcpf(1.f + ex);
                        const float z = xh * ga[e] + be[e];
                        // (1 - s) as a subtraction, not as e s: for z < -88.7 e overflows to +inf, s = 0 and inf * 0 would poison the
                        // whole (sample, group) with NaN; 1 - 0 = 1 gives SiLU' = 0 * (1 + z) = 0 like dsilu_f of the apply pass
                        q1[e] = sg * (1.f + z * (1.f - sg));
                    } else {
                        q1[e] = 1.f;
                    }
                    q2[e] = q1[e] * xh;
                }
#pragma unroll
                for (int k = 0; k < SETS; ++k) {
                    float d[8];
                    unpack8(rd[k], d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        a1[k][e] += d[e] * q1[e]; a2[k][e] += d[e] * q2[e];
                    }
                }
            }
        }
    }
    for (int k = 0; k < SETS; ++k) {
        reduce_slots(a1[k], active, slot, cc, s, red, ch1);     // per channel: sum dz        (= dbeta partial)
        reduce_slots(a2[k], active, slot, cc, s, red, ch2);     // per channel: sum dz * xhat (= dgamma partial)
        const int n2 = k * nx + n;
        if (tid < 2 * s.G) {
            const int g = tid >> 1;
            const float* src = (tid & 1) ? ch2 : ch1;
            float t = 0.f;
            for (int c = g * s.cpg; c < (g + 1) * s.cpg; ++c) t += src[c] * gamma[c];
            partial[(((long)n2 * s.nslices + blockIdx.z) * s.nchunks + chunk) * 2 * s.G + tid] = t;
        }
        const long so = (long)(n2 / set_images) * set_stride + c0;
        for (int i = tid; i < s.C; i += kThreads) {
            atomicAdd(dgamma + so + i, ch2[i]);
            atomicAdd(dbeta + so + i, ch1[i]);
        }
        __syncthreads();
    }
}

// S2D (with a split target, dx2 != null): the FIRST part of the split (channels [0, split_c)) is written in SPACE-TO-DEPTH layout --
// pixel (y, x) goes to row (y / 2, x / 2) of a (H / 2) x (W / 2) padded tensor of 4 split_c channels, columns [plane * split_c, ...),
// plane = 2 (y & 1) + (x & 1) -- which is how a sub-pixel upsample convolution's backward wants its cotangent (unet.py
// _upsample_subpixel): the space-to-depth pass over that tensor disappears (round 4: 0.3 ms of the CelebA-HQ step).
template <bool SILU, int SETS, bool S2D>
__global__ __launch_bounds__(kThreads) void gn_bwd_apply_kernel(
    const bf16_t* __restrict__ dy, const bf16_t* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ partial, GNShape s, int nx, int dy_compact, const bf16_t* __restrict__ accum,
    const bf16_t* __restrict__ accum2, bf16_t* __restrict__ dx, bf16_t* __restrict__ dx2, int split_c,
    int accumulate2, float* __restrict__ colsum, long colsum_ld, int pchunks) {
    __shared__ float sh_s1[SETS][kMaxG], sh_s2[SETS][kMaxG];
    __shared__ __attribute__((aligned(16))) float red[2048];
    __shared__ float chs[kMaxC];
    const int n = blockIdx.y, chunk = blockIdx.x, tid = threadIdx.x;
    const int c0 = blockIdx.z * s.C, g0 = blockIdx.z * s.G;
    gamma += c0; beta += c0;
    {
        __shared__ float fr[4][64];
        const double cnt = (double)s.H * s.W * s.cpg;
        for (int k = 0; k < SETS; ++k) {
            const int n2 = k * nx + n;
            fold_slab(partial + ((long)n2 * s.nslices + blockIdx.z) * pchunks * 2 * s.G, pchunks, s.G, fr);   // pchunks: the STATISTICS launch's blocks per sample
            if (tid < s.G) {
                sh_s1[k][tid] = (float)(((double)fr[0][2 * tid] + fr[1][2 * tid] + fr[2][2 * tid] + fr[3][2 * tid]) / cnt);
                sh_s2[k][tid] = (float)(((double)fr[0][2 * tid + 1] + fr[1][2 * tid + 1] + fr[2][2 * tid + 1] + fr[3][2 * tid + 1]) / cnt);
            }
            __syncthreads();
        }
    }
    const int slot = tid / s.lpp, cc = tid - slot * s.lpp;
    const bool active = slot < s.ppi;
    float cs[SETS][8];
#pragma unroll
    for (int k = 0; k < SETS; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) cs[k][e] = 0.f;
    if (active) {
        float mr[8], rs[8], ga[8], be[8], m1[SETS][8], m2[SETS][8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = cc * 8 + e, g = c / s.cpg;
            rs[e] = rstd[(long)n * s.Gf + g0 + g]; mr[e] = mean[(long)n * s.Gf + g0 + g] * rs[e];
            ga[e] = gamma[c]; be[e] = beta[c];
#pragma unroll
            for (int k = 0; k < SETS; ++k) { m1[k][e] = sh_s1[k][g]; m2[k][e] = sh_s2[k][g]; }
        }
        const long rpi = (long)(s.H + 2) * (s.W + 2);
        const bf16_t* xbase = x + (long)n * rpi * s.ldx + c0;
        const int ch = c0 + cc * 8;                       // first channel of this lane in the full tensor
        // Output routing: one tensor of C channels, or (dx2 != null: the input was a channel concat)
        // channels [0, split_c) -> dx (row stride split_c) and [split_c, C) -> dx2 (row stride C - split_c,
        // optionally accumulated): the concat backward costs no extra pass.
        const bool second = dx2 != nullptr && ch >= split_c;
        const int ostride = dx2 ? (second ? s.ld - split_c : split_c) : s.ld;
        const bool oacc = second && accumulate2;
        const bool any_oacc = dx2 != nullptr && accumulate2;      // (uniform) some lanes of the block accumulate into dx2
        // per set: wave-uniform bases of the sample's cotangent / running cotangents, per-lane base of its output rows
        const bf16_t* dbase[SETS]; const bf16_t* abase[SETS]; const bf16_t* bbase[SETS]; bf16_t* obase[SETS];
#pragma unroll
        for (int k = 0; k < SETS; ++k) {
            const long n2 = k * nx + n;
            dbase[k] = dy + c0 + (dy_compact ? n2 * s.H * s.W : n2 * rpi) * s.ld;
            abase[k] = accum ? accum + c0 + n2 * rpi * s.ld : nullptr;
            bbase[k] = accum2 ? accum2 + c0 + n2 * rpi * s.ld : nullptr;
            obase[k] = (second ? dx2 + (ch - split_c) : dx + ch) + n2 * rpi * ostride;
            if (S2D && !second) obase[k] = dx + ch + n2 * (long)((s.H >> 1) + 2) * ((s.W >> 1) + 2) * (4 * split_c);
        }
        const unsigned xb = s.ldx * 2, db = s.ld * 2, ob = ((S2D && !second) ? 4 * ostride : ostride) * 2, lb = cc * 16;
        const int wl2 = (s.W >> 1) + 2;                     // S2D: padded width of the half-resolution target
        const bool s2d_lane = S2D && !second;
        auto out_off = [&](int row, int yy, int xx) -> unsigned {
            if (!s2d_lane) return __umul24(row, ob);
            return __umul24(((yy >> 1) + 1) * wl2 + (xx >> 1) + 1, ob) + (unsigned)((((yy & 1) << 1) | (xx & 1)) * split_c * 2);
        };
        // Software pipeline (as in the stats kernel): pixel i+1's loads are in flight while pixel i is computed and
        // stored.  Reading the next pixel's accum / output before this pixel's store is safe: different rows.
        struct In { u32x4_t x, d[SETS], a[SETS], b[SETS], c[SETS]; };
        auto issue = [&](const PixelWalk& q, In& o) {
            o.x = ld16(xbase, boff(q.row(), xb, lb));
            const unsigned roff = boff(q.row(), db, lb);
            const unsigned doff = dy_compact ? boff(q.pi, db, lb) : roff;
            const unsigned ooff = __umul24(q.row(), ob);   // (only read by the accumulating lanes of the SECOND part: never space-to-depth)
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                o.d[k] = ld16(dbase[k], doff);
                if (accum) o.a[k] = ld16(abase[k], roff);
                if (accum2) o.b[k] = ld16(bbase[k], roff);
                if (any_oacc) o.c[k] = oacc ? ld16(obase[k], ooff) : u32x4_t{0u, 0u, 0u, 0u};
            }
        };
        PixelWalk w(s, chunk, slot);
        int yy = S2D ? w.pi / s.W : 0;                      // S2D: the walk's image row, kept beside it (PixelWalk itself does not need it)
        const int dyi = S2D ? s.ppi / s.W : 0;
        In nxt;
        if (w.ok()) issue(w, nxt);
        while (w.ok()) {
            const In cur = nxt;
            const unsigned ooff = out_off(w.row(), yy, w.x);
            if (S2D) yy += dyi + (w.x + w.dx >= s.W ? 1 : 0);
            w.next();
            if (w.ok()) issue(w, nxt);
            float v[8], xh[8], dsl[8];
            unpack8(cur.x, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                xh[e] = v[e] * rs[e] - mr[e];
                dsl[e] = (SILU ? dsilu_f(xh[e] * ga[e] + be[e]) : 1.f) * ga[e];
            }
#pragma unroll
            for (int k = 0; k < SETS; ++k) {
                float d[8], o[8];
                unpack8(cur.d[k], d);
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float t = rs[e] * (d[e] * dsl[e] - m1[k][e] - xh[e] * m2[k][e]);
                    cs[k][e] += t;
                    o[e] = t;
                }
                // the cotangents x already carries: (uniform) branches -- most sites have none or one of them, and unpacking
                // and adding eight zeros per absent operand was a fifth of this loop's vector work
                if (accum) {
                    float r[8];
                    unpack8(cur.a[k], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                if (accum2) {
                    float r[8];
                    unpack8(cur.b[k], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                if (any_oacc) {
                    float r[8];
                    unpack8(cur.c[k], r);
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += r[e];
                }
                st16(obase[k], ooff, pack8(o));
            }
        }
    }
    if (colsum) {
        for (int k = 0; k < SETS; ++k) {
            reduce_slots(cs[k], active, slot, cc, s, red, chs);
            for (int i = tid; i < s.C; i += kThreads) atomicAdd(colsum + (long)(k * nx + n) * colsum_ld + c0 + i, chs[i]);
            __syncthreads();
        }
    }
}

// Target block count of a launch.  Forward kernels (34-60 VGPRs, many blocks resident per CU): ~3 per CU.  Backward kernels
// (118-190 VGPRs: two 256-thread blocks resident per CU): 512 = exactly one resident round -- 768 was a round and a half, and the
// half-empty second round cost 3-12 % of every site (profiles/r03_gn_block_count_sweep.txt: 256^2 x 128 548 -> 520 us,
// 128^2 x 128 150 -> 135 us, 64^2 x 512 164 -> 144 us; round-robin pixel groups instead of contiguous runs: 0-4 % slower).
constexpr int kBlocksFwd = 768, kBlocksBwd = 512, kBlocksBwdStats = 767;

bool make_shape(int H, int W, int C, int G, GNShape& s, int N = 16, int blocks = kBlocksFwd) {
    if (H <= 0 || W <= 0 || C <= 0 || G <= 0 || G > kMaxG || C % G || C % 8) return false;
    s.H = H; s.W = W; s.ld = C; s.ldx = C; s.Gf = G; s.cpg = C / G;
    int gs = G;                                         // groups per slice
    if (C > kMaxC) {
        // the largest whole-group slice of <= kMaxC channels (a multiple of 8) that keeps >= 90 % of the
        // block's lanes busy; else the best utilisation found
        int best = 0, best_util = -1;
        gs = 0;
        for (int d = G; d >= 1; --d) {
            if (G % d) continue;
            const int cs = d * s.cpg;
            if (cs > kMaxC || cs % 8) continue;
            const int lpp = cs / 8, util = (kThreads / lpp) * lpp;
            if (util * 10 >= kThreads * 9) { gs = d; break; }
            if (util > best_util) { best_util = util; best = d; }
        }
        if (gs == 0) gs = best;
        if (gs == 0) return false;
    }
    s.G = gs; s.C = gs * s.cpg; s.nslices = G / gs;
    s.lpp = s.C / 8;
    if (s.lpp > kThreads) return false;
    s.ppi = kThreads / s.lpp;
    const int px = H * W;
    // ~3 blocks per CU, and >= 16 pixel iterations per thread so that the per-block prologue (slab fold)
    // and epilogue (LDS + global atomics) are amortised
    int nch = (blocks + N * s.nslices - 1) / (N * s.nslices);
    const int max_by_work = (px + 16 * s.ppi - 1) / (16 * s.ppi);
    if (nch > max_by_work) nch = max_by_work;
    if (blocks == kBlocksBwd || blocks == kBlocksBwdStats) {
        // backward: 32 iterations per thread where that still leaves a block per CU (64^2 x 256: 80 -> 68 us with 256 blocks)
        const int by_work32 = (px + 32 * s.ppi - 1) / (32 * s.ppi);
        if (nch > by_work32 && (long)by_work32 * N * s.nslices >= 256) nch = by_work32;
    }
    if (nch > 256) nch = 256;
    if (nch < 1) nch = 1;
    s.chunk_px = (px + nch - 1) / nch;
    s.nchunks = (px + s.chunk_px - 1) / s.chunk_px;
    return true;
}

// slab kernels at the small sites (groupnorm_slab.hip); siss_groupnorm_set_slab(0) sends every site to the two-pass kernels (tests)
int g_use_slab = 1;

}  // namespace

int siss_gn_slab_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, int N, int H,
                     int W, int C, int G, float eps, int silu, int out_compact, int ldx, void* stream);
int siss_gn_slab_bwd(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                     const float* rstd, void* dx, const void* accum, const void* accum2, void* dx2, int split_c,
                     int accumulate2, float* dgamma, float* dbeta, float* colsum, long colsum_ld, int n2, int nx,
                     int set_images, long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx,
                     void* stream);
extern "C" {

// 1 (default): the small sites (<= 32 x 32 pixels forward, <= 16 x 16 backward) run on the one-launch slab kernels
// (groupnorm_slab.hip); 0: every site takes the two-pass kernels (the parity tests compare the two); -1: back to the default.
// Process-wide; returns the value in effect.
int siss_groupnorm_set_slab(int on) {
    g_use_slab = on != 0;
    return g_use_slab;
}

// floats needed in `partial` for n samples: scratch between a site's statistics and apply launches, 16-B aligned.
long siss_gn_partial_words(int n, int H, int W, int C, int G) {
    GNShape s;
    if (!make_shape(H, W, C, G, s, 1)) return -1;   // N = 1 gives the largest chunk count -> upper bound
    return (long)n * s.nslices * s.nchunks * 2 * s.G;
}

// GroupNorm statistics of a padded-NHWC tensor (M flat rows of N channels, row stride ld elements, zero halo rows) in the format
// the persistent 3x3 convolution leaves them (siss_gemm_nt_qstats; siss_conv_qstats_words(M, N) floats): what siss_groupnorm_fwd_qs
// takes as qsA / qsB for a producer that forms none.  rows_per_image >= 256 (a 254-row tile spans at most two images), N % 8 == 0,
// N <= 512 (one block folds a tile's quads: four partial sums per 8-channel chunk in 256 threads).  One read of the tensor.
int siss_quad_stats(const void* x, long ld, int M, int N, int rows_per_image, float* qs, void* stream) {
    SISS_CHECK_ARG(x && qs && M > 0 && N > 0 && N % 8 == 0 && N <= 8 * kThreads && ld >= N && ld % 8 == 0 && rows_per_image >= 256);
    SISS_CHECK_ARG((uintptr_t)x % 16 == 0 && (uintptr_t)qs % 8 == 0 && (N >> 3) * 4 <= kThreads);
    const int tiles = (M + kQsTileRows - 1) / kQsTileRows;
    gn_quad_stats_kernel<<<dim3(2 * tiles), kThreads, 0, (hipStream_t)stream>>>((const bf16_t*)x, ld, M, N, rows_per_image, qs);
    SISS_LAUNCH_RET();
}

// y = act(GroupNorm(x)); x padded NHWC; y padded or compact ([N][H*W][C]).  Writes mean/rstd [N][G].
// ldx: row stride of x in elements (0 = C; > C when x is a column view of a wider concat buffer).
// qsA / qsB (optional): the statistics the producing convolution(s) left (siss_gemm_nt_qstats wrote them: *written == 1) --
// channels [0, ca) of x are the ca output channels of producer A, the remaining C - ca those of producer B (qsB null and
// ca == C for a single producer).  With them the two-pass form needs no statistics pass; sites served by the one-launch
// kernels, or whose groups are not whole 4-channel quads, ignore them.
int siss_groupnorm_fwd_qs(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                          float* rstd, float* partial, const float* qsA, int ca, const float* qsB, int N, int H, int W,
                          int C, int G, float eps, int silu, int out_compact, int ldx, void* stream) {
    GNShape s;
    SISS_CHECK_ARG(!qsA || (ca > 0 && ca <= C && ca % 4 == 0 && (ca == C) == (qsB == nullptr)));
    SISS_CHECK_ARG(((uintptr_t)qsA | (uintptr_t)qsB) % 8 == 0);
    SISS_CHECK_ARG(x && gamma && beta && y && mean && rstd && partial && N > 0);
    SISS_CHECK_ARG(make_shape(H, W, C, G, s, N, kBlocksFwd));
    SISS_CHECK_ARG(ldx == 0 || (ldx >= C && ldx % 8 == 0));
    if (ldx) s.ldx = ldx;
    SISS_CHECK_ARG(((uintptr_t)x | (uintptr_t)y | (uintptr_t)partial) % 16 == 0);
    if (g_use_slab && s.nslices == 1) {
        // small sites: one launch, the block holds its (sample, channel slice) on chip (groupnorm_slab.hip)
        const int rc = siss_gn_slab_fwd(x, gamma, beta, y, mean, rstd, N, H, W, C, G, eps, silu, out_compact, ldx, stream);
        if (rc >= 0) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    dim3 grid(s.nchunks, N, s.nslices);
    if (qsA && s.cpg % 4 == 0 && (long)(H + 2) * (W + 2) >= 256) {
        siss_count_dispatch(SISS_K_GN_QSTATS);
        gn_qstats_finalize_kernel<<<dim3(N, (G + 3) / 4), kThreads, 0, st>>>(qsA, ca / 4, qsB, (C - ca) / 4, H, W, G, s.cpg, eps,
                                                                             mean, rstd);
        partial = nullptr;                              // gn_apply_kernel reads mean / rstd instead of folding a slab
    } else
        gn_stats_kernel<<<grid, kThreads, 0, st>>>((const bf16_t*)x, s, partial);
    if (silu)
        gn_apply_kernel<true><<<grid, kThreads, 0, st>>>((const bf16_t*)x, gamma, beta, partial, s, eps, out_compact, (bf16_t*)y, mean, rstd);
    else
        gn_apply_kernel<false><<<grid, kThreads, 0, st>>>((const bf16_t*)x, gamma, beta, partial, s, eps, out_compact, (bf16_t*)y, mean, rstd);
    SISS_LAUNCH_RET();
}
/* the same without statistics from the producer */
int siss_groupnorm_fwd_ld(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                          float* rstd, float* partial, int N, int H, int W, int C, int G, float eps,
                          int silu, int out_compact, int ldx, void* stream) {
    return siss_groupnorm_fwd_qs(x, gamma, beta, y, mean, rstd, partial, nullptr, C, nullptr, N, H, W, C, G, eps, silu,
                                 out_compact, ldx, stream);
}
/* the same with ldx = C */
int siss_groupnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean,
                       float* rstd, float* partial, int N, int H, int W, int C, int G, float eps,
                       int silu, int out_compact, void* stream) {
    return siss_groupnorm_fwd_ld(x, gamma, beta, y, mean, rstd, partial, N, H, W, C, G, eps, silu, out_compact, 0, stream);
}

static int gn_bwd_launch(const void* dy, const void* x, const float* gamma, const float* beta,
                          const float* mean, const float* rstd, void* dx, const void* accum, const void* accum2,
                          void* dx2, int split_c, int accumulate2, float* dgamma,
                          float* dbeta, float* colsum, long colsum_ld, float* partial, int n2, int nx, int set_images,
                          long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx, int s2d, void* stream) {
    GNShape s;
    SISS_CHECK_ARG(!s2d || (dx2 && H % 2 == 0 && W % 2 == 0));
    SISS_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta && partial);
    SISS_CHECK_ARG(n2 > 0 && nx > 0 && set_images > 0 && n2 % set_images == 0);
    SISS_CHECK_ARG(n2 == nx || n2 == 2 * nx);     // cotangent sets per saved sample: 1 or 2
    SISS_CHECK_ARG(make_shape(H, W, C, G, s, nx, kBlocksBwd));
    SISS_CHECK_ARG(((uintptr_t)dy | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)accum | (uintptr_t)accum2 | (uintptr_t)dx2) % 16 == 0);
    SISS_CHECK_ARG(!dx2 || (split_c > 0 && split_c < C && split_c % 8 == 0));
    SISS_CHECK_ARG(ldx == 0 || (ldx >= C && ldx % 8 == 0));
    SISS_CHECK_ARG((uintptr_t)partial % 16 == 0);
    if (ldx) s.ldx = ldx;
    if (!s2d && g_use_slab && s.nslices == 1 && (n2 == nx || n2 == 2 * nx) && n2 / set_images <= 2) {
        const int rc = siss_gn_slab_bwd(dy, x, gamma, beta, mean, rstd, dx, accum, accum2, dx2, split_c, accumulate2, dgamma, dbeta,
                                        colsum, colsum_ld, n2, nx, set_images, set_stride, H, W, C, G, silu, dy_compact, ldx, stream);
        if (rc >= 0) return rc;
    }
    hipStream_t st = (hipStream_t)stream;
    // the statistics kernel (142 VGPRs: three blocks resident per CU) and the apply kernel (190: two) each get ONE resident round
    GNShape ss;
    SISS_CHECK_ARG(make_shape(H, W, C, G, ss, nx, kBlocksBwdStats));
    ss.ldx = s.ldx;
    dim3 grid(s.nchunks, nx, s.nslices), grid_s(ss.nchunks, nx, ss.nslices);
    const bf16_t* dyp = (const bf16_t*)dy; const bf16_t* xp = (const bf16_t*)x;
#define GN_BWD(SILU, SETS)                                                                                          \
    gn_bwd_stats_kernel<SILU, SETS><<<grid_s, kThreads, 0, st>>>(dyp, xp, gamma, beta, mean, rstd, ss, nx, dy_compact, \
                                                               set_images, set_stride, partial, dgamma, dbeta);    \
    if (s2d)                                                                                                        \
        gn_bwd_apply_kernel<SILU, SETS, true><<<grid, kThreads, 0, st>>>(dyp, xp, gamma, beta, mean, rstd, partial, s, nx,    \
                                                               dy_compact, (const bf16_t*)accum, (const bf16_t*)accum2, (bf16_t*)dx, (bf16_t*)dx2, split_c,  \
                                                               accumulate2, colsum, colsum_ld, ss.nchunks);                \
    else                                                                                                            \
        gn_bwd_apply_kernel<SILU, SETS, false><<<grid, kThreads, 0, st>>>(dyp, xp, gamma, beta, mean, rstd, partial, s, nx,    \
                                                               dy_compact, (const bf16_t*)accum, (const bf16_t*)accum2, (bf16_t*)dx, (bf16_t*)dx2, split_c,  \
                                                               accumulate2, colsum, colsum_ld, ss.nchunks)
    if (n2 == nx) { if (silu) { GN_BWD(true, 1); } else { GN_BWD(false, 1); } }
    else          { if (silu) { GN_BWD(true, 2); } else { GN_BWD(false, 2); } }
#undef GN_BWD
    SISS_LAUNCH_RET();
}
// dx (padded, n2 samples) from dy (n2 samples, padded or compact) and the saved x (nx samples,
// x index = n2 % nx).  dgamma/dbeta: [sets][...] accumulated atomically at set = n2 / set_images
// with `set_stride` floats between sets.  accum / accum2 (optional, padded [.., C] like a C-channel dx) are added;
// dx2 (optional): the normalised input was a channel concat -- channels [0, split_c) of the result go to dx
// (row stride split_c), channels [split_c, C) to dx2 (row stride C - split_c; += when accumulate2);
// colsum (optional, f32 rows of colsum_ld floats, pre-zeroed) receives the per-sample channel sums of dx.
// ldx: row stride of the saved x in elements (0 = C).
int siss_groupnorm_bwd_ld(const void* dy, const void* x, const float* gamma, const float* beta,
                          const float* mean, const float* rstd, void* dx, const void* accum, const void* accum2,
                          void* dx2, int split_c, int accumulate2, float* dgamma,
                          float* dbeta, float* colsum, long colsum_ld, float* partial, int n2, int nx, int set_images,
                          long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx, void* stream) {
    return gn_bwd_launch(dy, x, gamma, beta, mean, rstd, dx, accum, accum2, dx2, split_c, accumulate2, dgamma, dbeta, colsum, colsum_ld,
                         partial, n2, nx, set_images, set_stride, H, W, C, G, silu, dy_compact, ldx, 0, stream);
}
// The same for a split target (dx2 != null; H, W even) whose FIRST part -- channels [0, split_c), `dx` -- is written in
// space-to-depth layout: dx is a padded (H / 2) x (W / 2) tensor of 4 split_c channels, pixel (y, x) at row (y / 2, x / 2), columns
// [plane * split_c, (plane + 1) * split_c), plane = 2 (y & 1) + (x & 1) -- the layout siss_space_to_depth produces.
int siss_groupnorm_bwd_ld_s2d(const void* dy, const void* x, const float* gamma, const float* beta,
                              const float* mean, const float* rstd, void* dx, const void* accum, const void* accum2,
                              void* dx2, int split_c, int accumulate2, float* dgamma,
                              float* dbeta, float* colsum, long colsum_ld, float* partial, int n2, int nx, int set_images,
                              long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx, void* stream) {
    return gn_bwd_launch(dy, x, gamma, beta, mean, rstd, dx, accum, accum2, dx2, split_c, accumulate2, dgamma, dbeta, colsum, colsum_ld,
                         partial, n2, nx, set_images, set_stride, H, W, C, G, silu, dy_compact, ldx, 1, stream);
}
/* the same with ldx = C */
int siss_groupnorm_bwd(const void* dy, const void* x, const float* gamma, const float* beta,
                       const float* mean, const float* rstd, void* dx, const void* accum, const void* accum2,
                       void* dx2, int split_c, int accumulate2, float* dgamma,
                       float* dbeta, float* colsum, long colsum_ld, float* partial, int n2, int nx, int set_images,
                       long set_stride, int H, int W, int C, int G, int silu, int dy_compact, void* stream) {
    return siss_groupnorm_bwd_ld(dy, x, gamma, beta, mean, rstd, dx, accum, accum2, dx2, split_c, accumulate2, dgamma, dbeta,
                                 colsum, colsum_ld, partial, n2, nx, set_images, set_stride, H, W, C, G, silu, dy_compact, 0,
                                 stream);
}

}  // extern "C"
