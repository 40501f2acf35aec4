// Flat-buffer gradient bookkeeping + AdamW.
//
// Reference semantics: delete_celeb.py:714-753 (||g_x||, ||g_a||, s = scaling_norm/||g_a||,
// g = g_x - s*g_a), :767 (clip_grad_norm_ 1.0) and :769 (torch.optim.AdamW.step).
// The reference walks 450 tensors five times (~2,250 launches, 3 clones); here parameters
// and both gradient sets live in ONE flat buffer each, so the whole block is two
// streaming passes:
//   pass 1  reads g_x, g_a                      -> ||g_x||^2, ||g_a||^2, <g_x,g_a>  (f64 slabs)
//   pass 2  reads g_x, g_a, p, m, v; writes p, m, v (+ bf16 shadow of p)
// ||g||^2 = ||g_x||^2 - 2 s <g_x,g_a> + s^2 ||g_a||^2 is formed from the f64 sums, so the clip
// coefficient needs no third pass.  All scalars (s, clip, step count) stay on the device:
// no host sync, and the launches replay correctly from a hipGraph.
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxBlocks = 2048;

struct StepScalars {   // lives in device memory; 16 floats
    float norm_x, norm_a, dot, scale;      // 0..3  (scale = s in g = g_x - s*g_a)
    float pre_clip_norm, clip_coef, step, pad0;  // 4..7
    float bc1, bc2_sqrt, pad1, pad2;       // 8..11
    float pad3[4];
};

__global__ __launch_bounds__(kThreads) void norms_kernel(const float* __restrict__ gx,
                                                         const float* __restrict__ ga, long n,
                                                         double* __restrict__ partials) {
    __shared__ double sh[3 * kThreads / 64];
    double sxx = 0, saa = 0, sxa = 0;
    const long nvec = n / 4;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long)gridDim.x * kThreads) {
        f32x4_t x = reinterpret_cast<const f32x4_t*>(gx)[i];
        f32x4_t a = reinterpret_cast<const f32x4_t*>(ga)[i];
        float pxx = 0, paa = 0, pxa = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) { pxx += x[j] * x[j]; paa += a[j] * a[j]; pxa += x[j] * a[j]; }
        sxx += pxx; saa += paa; sxa += pxa;
    }
    if (blockIdx.x == 0)
        for (long i = nvec * 4 + threadIdx.x; i < n; i += kThreads) {
            sxx += (double)gx[i] * gx[i]; saa += (double)ga[i] * ga[i]; sxa += (double)gx[i] * ga[i];
        }
    sxx = wave_sum_d(sxx); saa = wave_sum_d(saa); sxa = wave_sum_d(sxa);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[3 * w] = sxx; sh[3 * w + 1] = saa; sh[3 * w + 2] = sxa; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0, c = 0;
        for (int i = 0; i < kThreads / 64; ++i) { a += sh[3 * i]; b += sh[3 * i + 1]; c += sh[3 * i + 2]; }
        partials[3 * blockIdx.x] = a; partials[3 * blockIdx.x + 1] = b; partials[3 * blockIdx.x + 2] = c;
    }
}

// mode 0: norm fixing  s = scaling_norm / ||g_a||        (delete_celeb.py:746)
// mode 1: erasediff    s = -max(eta - <gx,ga>/||ga||^2, 0) (:740-742)
// mode 2: like 0 but s = 0 when it would be inf          (delete_tshirt.py:688-690)
__global__ void scalars_kernel(const double* __restrict__ partials, int nblk, int mode, float knob,
                               float max_norm, float beta1, float beta2, StepScalars* __restrict__ sc) {
    // one block folds the per-block partial sums in parallel (a single thread walking 2048 x 3 dependent L2 reads
    // cost 240 us: more than the streaming pass that produced them); fixed lane -> index map: deterministic
    __shared__ double shs[3 * kThreads / 64];
    double xx = 0, aa = 0, xa = 0;
    for (int i = threadIdx.x; i < nblk; i += kThreads) { xx += partials[3 * i]; aa += partials[3 * i + 1]; xa += partials[3 * i + 2]; }
    xx = wave_sum_d(xx); aa = wave_sum_d(aa); xa = wave_sum_d(xa);
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; shs[3 * w] = xx; shs[3 * w + 1] = aa; shs[3 * w + 2] = xa; }
    __syncthreads();
    if (threadIdx.x != 0) return;
    xx = aa = xa = 0;
    for (int i = 0; i < kThreads / 64; ++i) { xx += shs[3 * i]; aa += shs[3 * i + 1]; xa += shs[3 * i + 2]; }
    const double nx = sqrt(xx), na = sqrt(aa);
    double s;
    if (mode == 1) {
        s = (double)knob - xa / aa;
        s = -(s > 0 ? s : 0);
    } else {
        s = (double)knob / na;
        if (mode == 2 && isinf(s)) s = 0;
    }
    double g2 = xx - 2 * s * xa + s * s * aa;
    if (g2 < 0) g2 = 0;
    const double gn = sqrt(g2);
    double coef = (double)max_norm / (gn + 1e-6);   // torch.nn.utils.clip_grad_norm_
    if (coef > 1) coef = 1;
    const float step = sc->step + 1.f;
    sc->norm_x = (float)nx; sc->norm_a = (float)na; sc->dot = (float)xa; sc->scale = (float)s;
    sc->pre_clip_norm = (float)gn; sc->clip_coef = (float)coef; sc->step = step;
    sc->bc1 = 1.f - powf(beta1, step);
    sc->bc2_sqrt = sqrtf(1.f - powf(beta2, step));
}

// pass 2: g = clip * (g_x - s g_a); torch.optim.AdamW single-tensor update order.
__global__ __launch_bounds__(kThreads) void recombine_adamw_kernel(
    const float* __restrict__ gx, const float* __restrict__ ga, float* __restrict__ p,
    float* __restrict__ m, float* __restrict__ v, bf16_t* __restrict__ shadow,
    float* __restrict__ g_out, long n, float lr, float beta1, float beta2, float eps, float wd,
    const StepScalars* __restrict__ sc) {
    const float s = sc->scale, clip = sc->clip_coef, bc1 = sc->bc1, bc2s = sc->bc2_sqrt;
    const float step_size = lr / bc1;
    const float decay = 1.f - lr * wd;
    const long nvec = n / 4;
    auto upd = [&](float x, float a, float& pp, float& mm, float& vv) -> float {
        const float g = __fmul_rn(__fsub_rn(x, __fmul_rn(s, a)), clip);
        pp = __fmul_rn(pp, decay);
        mm = __fadd_rn(mm, __fmul_rn(__fsub_rn(g, mm), 1.f - beta1));           // lerp
        vv = __fadd_rn(__fmul_rn(vv, beta2), __fmul_rn(__fmul_rn(g, g), 1.f - beta2));
        const float den = __fadd_rn(sqrtf(vv) / bc2s, eps);
        pp = __fsub_rn(pp, __fmul_rn(step_size, mm / den));
        return g;
    };
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long)gridDim.x * kThreads) {
        f32x4_t x = reinterpret_cast<const f32x4_t*>(gx)[i], a = reinterpret_cast<const f32x4_t*>(ga)[i];
        f32x4_t pp = reinterpret_cast<f32x4_t*>(p)[i], mm = reinterpret_cast<f32x4_t*>(m)[i],
                vv = reinterpret_cast<f32x4_t*>(v)[i], gg;
#pragma unroll
        for (int j = 0; j < 4; ++j) { float P = pp[j], M = mm[j], V = vv[j]; gg[j] = upd(x[j], a[j], P, M, V); pp[j] = P; mm[j] = M; vv[j] = V; }
        reinterpret_cast<f32x4_t*>(p)[i] = pp;
        reinterpret_cast<f32x4_t*>(m)[i] = mm;
        reinterpret_cast<f32x4_t*>(v)[i] = vv;
        if (g_out) reinterpret_cast<f32x4_t*>(g_out)[i] = gg;
        if (shadow) reinterpret_cast<u32x2_t*>(shadow)[i] = u32x2_t{pack_bf2(pp[0], pp[1]), pack_bf2(pp[2], pp[3])};
    }
    if (blockIdx.x == 0)
        for (long i = nvec * 4 + threadIdx.x; i < n; i += kThreads) {
            float P = p[i], M = m[i], V = v[i];
            const float g = upd(gx[i], ga[i], P, M, V);
            p[i] = P; m[i] = M; v[i] = V;
            if (g_out) g_out[i] = g;
            if (shadow) shadow[i] = f2bf(P);
        }
}

__global__ __launch_bounds__(kThreads) void cast_bf16_kernel(const float* __restrict__ src,
                                                             bf16_t* __restrict__ dst, long n) {
    const long nvec = n / 4;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long)gridDim.x * kThreads) {
        f32x4_t x = reinterpret_cast<const f32x4_t*>(src)[i];
        reinterpret_cast<u32x2_t*>(dst)[i] = u32x2_t{pack_bf2(x[0], x[1]), pack_bf2(x[2], x[3])};
    }
    if (blockIdx.x == 0)
        for (long i = nvec * 4 + threadIdx.x; i < n; i += kThreads) dst[i] = f2bf(src[i]);
}

// [taps][co][ci] f32 master -> [taps][ci][co] bf16 with the tap order reversed (dgrad operand).
__global__ void conv_weight_dgrad_kernel(const float* __restrict__ w, bf16_t* __restrict__ wt, int taps,
                                         int co, int ci) {
    __shared__ float tile[32][33];
    const int tap = blockIdx.z;
    const float* src = w + (long)tap * co * ci;
    bf16_t* dst = wt + (long)(taps - 1 - tap) * co * ci;
    const int c0 = blockIdx.x * 32, o0 = blockIdx.y * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int o = o0 + r, c = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (o < co && c < ci) ? src[(long)o * ci + c] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r, o = o0 + threadIdx.x;
        if (c < ci && o < co) dst[(long)c * co + o] = f2bf(tile[threadIdx.x][r]);
    }
}


// All dgrad weight copies in ONE launch.  job table (device): per weight {src off (floats), dst off (bf16
// elements), taps, co, ci, first tile}; a block finds its job by binary search over `first tile`.
struct WtJob { long src, dst; int taps, co, ci, tile0; };

__global__ void conv_weight_dgrad_multi_kernel(const float* __restrict__ flat, bf16_t* __restrict__ wt_all,
                                               const WtJob* __restrict__ jobs, int njobs) {
    // 64 x 64 tiles (256-B reads, 128-B bf16 writes): with 32 x 32 tiles the per-block job lookup below (a
    // dependent walk of ~8 L2 reads) cost as much as the tile's own traffic (2.8 TB/s measured)
    constexpr int TS = 64;
    __shared__ float tile[TS][TS + 1];
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WtJob j = jobs[lo];
    int t = blockIdx.x - j.tile0;
    const int tc = (j.ci + TS - 1) / TS, to = (j.co + TS - 1) / TS;
    const int tap = t / (tc * to); t -= tap * tc * to;
    const int o0 = (t / tc) * TS, c0 = (t % tc) * TS;
    const float* src = flat + j.src + (long)tap * j.co * j.ci;
    bf16_t* dst = wt_all + j.dst + (long)(j.taps - 1 - tap) * j.co * j.ci;
    for (int r = threadIdx.y; r < TS; r += blockDim.y) {
        const int o = o0 + r, c = c0 + threadIdx.x;
        tile[r][threadIdx.x] = (o < j.co && c < j.ci) ? src[(long)o * j.ci + c] : 0.f;
    }
    __syncthreads();
    for (int r = threadIdx.y; r < TS; r += blockDim.y) {
        const int c = c0 + r, o = o0 + threadIdx.x;
        if (c < j.ci && o < j.co) dst[(long)c * j.co + o] = f2bf(tile[threadIdx.x][r]);
    }
}

// The same from the bf16 operand shadow (bitwise the rounded master: the fused AdamW kernel / siss_cast_f32_bf16 keep it so): half
// the bytes read, and 16-B accesses on both sides -- a lane reads 8 consecutive c of one o, scatters them into the transposed LDS
// tile, and writes 8 consecutive o of one c.  Weights whose co or ci is not a multiple of 8 (conv_in) take the scalar walk.
__global__ __launch_bounds__(256) void conv_weight_dgrad_multi_bf16_kernel(const bf16_t* __restrict__ shadow, bf16_t* __restrict__ wt_all,
                                                                           const WtJob* __restrict__ jobs, int njobs) {
    constexpr int TS = 64, LD = TS + 8;                   // 144-B rows: 16-B aligned vectors, rows 4 banks apart
    __shared__ __attribute__((aligned(16))) bf16_t tile[TS][LD];   // [c][o]
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const WtJob j = jobs[lo];
    int t = blockIdx.x - j.tile0;
    const int tc = (j.ci + TS - 1) / TS, to = (j.co + TS - 1) / TS;
    const int tap = t / (tc * to); t -= tap * tc * to;
    const int o0 = (t / tc) * TS, c0 = (t % tc) * TS;
    const bf16_t* src = shadow + j.src + (long)tap * j.co * j.ci;
    bf16_t* dst = wt_all + j.dst + (long)(j.taps - 1 - tap) * j.co * j.ci;
    const int tid = threadIdx.x;
    if ((j.ci | j.co) % 8 == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int id = tid + 256 * k, r = id >> 3, ch = id & 7;
            const int o = o0 + r, c = c0 + ch * 8;
            u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
            if (o < j.co && c < j.ci) v = *reinterpret_cast<const u32x4_t*>(src + (long)o * j.ci + c);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                tile[ch * 8 + 2 * e][r] = (bf16_t)(v[e] & 0xffffu);
                tile[ch * 8 + 2 * e + 1][r] = (bf16_t)(v[e] >> 16);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int id = tid + 256 * k, r = id >> 3, ch = id & 7;
            const int c = c0 + r, o = o0 + ch * 8;
            if (c < j.ci && o < j.co) *reinterpret_cast<u32x4_t*>(dst + (long)c * j.co + o) = *reinterpret_cast<const u32x4_t*>(&tile[r][ch * 8]);
        }
    } else {
        for (int id = tid; id < TS * TS; id += 256) {
            const int r = id >> 6, cc = id & 63, o = o0 + r, c = c0 + cc;
            tile[cc][r] = (o < j.co && c < j.ci) ? src[(long)o * j.ci + c] : (bf16_t)0;
        }
        __syncthreads();
        for (int id = tid; id < TS * TS; id += 256) {
            const int r = id >> 6, oo = id & 63, c = c0 + r, o = o0 + oo;
            if (c < j.ci && o < j.co) dst[(long)c * j.co + o] = tile[r][oo];
        }
    }
}

inline int grid_for(long n) {
    long b = (n / 4 + kThreads - 1) / kThreads;
    if (b < 1) b = 1;
    if (b > kMaxBlocks) b = kMaxBlocks;
    return (int)b;
}

// ---- sub-pixel form of Upsample2D (nearest 2x -> conv3x3; round 4).  Output pixel (2Y + py, 2X + px) reads the LOW-resolution rows
// {Y - 1 + py, Y + py} and columns {X - 1 + px, X + px}: four 2x2-tap phase convolutions whose weights are sums of the 3x3 taps
// (16 instead of 36 tap products per low-resolution pixel).  Phase tap a of plane row py collects filter rows: py = 0: a = 0 <- {0},
// a = 1 <- {1, 2}; py = 1: a = 0 <- {0, 1}, a = 1 <- {2}; columns alike.  phase_tap(k, p) is that map (which 2-tap index filter
// index k lands on in phase p).
__device__ __host__ __forceinline__ int phase_tap(int k, int p) { return p == 0 ? (k >= 1) : (k >= 2); }

// w [9][Co][Ci] f32 (master) -> wf [4 planes][4 taps][Co][Ci] bf16 (fprop operand) and wd [16 = plane * 4 + tap][Ci][Co] bf16 (dgrad
// operand: the transposes): sums in f32, ONE rounding.
template <typename T>
__global__ void upsample_phase_weights_kernel(const float* __restrict__ w, T* __restrict__ wf, T* __restrict__ wd, int Co, int Ci) {
    const long total = (long)16 * Co * Ci;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int ci = i % Ci; long t = i / Ci; const int co = t % Co; const int pt = t / Co;
        const int plane = pt >> 2, tap = pt & 3, py = plane >> 1, px = plane & 1, a = tap >> 1, b = tap & 1;
        float acc = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                if (phase_tap(ky, py) == a && phase_tap(kx, px) == b) acc += w[((long)(ky * 3 + kx) * Co + co) * Ci + ci];
        const T v = from_f<T>(acc);
        wf[i] = v;
        wd[((long)pt * Ci + ci) * Co + co] = v;
    }
}
// dW[set][ky * 3 + kx][co][ci] += sum over the four planes of dW4[set][plane][phase tap of (ky, kx) in that plane][co][ci]
__global__ void upsample_phase_wgrad_fold_kernel(const float* __restrict__ dW4, float* __restrict__ dW, long set_stride, int Co, int Ci) {
    const long per = (long)Co * Ci, total = 9 * per;
    const int set = blockIdx.y;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int k = i / per; const long e = i - k * per;
        const int ky = k / 3, kx = k - ky * 3;
        float acc = 0.f;
#pragma unroll
        for (int plane = 0; plane < 4; ++plane) {
            const int tap = phase_tap(ky, plane >> 1) * 2 + phase_tap(kx, plane & 1);
            acc += dW4[(((long)set * 4 + plane) * 4 + tap) * per + e];
        }
        dW[(long)set * set_stride + i] += acc;
    }
}

// Zero fill of a LIST of stretches of one buffer in one launch (the gradient buffer minus what the coming backward pass overwrites).
// Units are 16-byte granules: tab = n starts, then n + 1 prefix sums of the lengths; thread g of the compacted index space finds
// its stretch by binary search (the table is a few KB: L1 / L2 hits) and stores one float4.
__global__ __launch_bounds__(256) void zero_ranges_kernel(float* __restrict__ base, const long* __restrict__ tab, int n, long total) {
    const long* start = tab;
    const long* pre = tab + n;
    for (long g = (long)blockIdx.x * 256 + threadIdx.x; g < total; g += (long)gridDim.x * 256) {
        int lo = 0, hi = n - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (pre[mid] <= g) lo = mid; else hi = mid - 1;
        }
        reinterpret_cast<f32x4_t*>(base)[start[lo] + (g - pre[lo])] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
}

}  // namespace

extern "C" {

// base[16 * start_i .. 16 * (start_i + len_i)) bytes = 0 for the n stretches of the DEVICE table tab: n starts followed by the n + 1
// prefix sums of the lengths (longs, in 16-byte granules; total_granules = the last prefix sum).  Stretches must not overlap.
int siss_zero_ranges(float* base, const long* tab, int n, long total_granules, void* stream) {
    SISS_CHECK_ARG(base && tab && n > 0 && total_granules > 0 && (uintptr_t)base % 16 == 0);
    long blocks = cdiv(total_granules, 256L);
    if (blocks > 8192) blocks = 8192;
    zero_ranges_kernel<<<dim3((unsigned)blocks), 256, 0, (hipStream_t)stream>>>(base, tab, n, total_granules);
    SISS_LAUNCH_RET();
}

long siss_opt_partials_words(void) { return 3L * kMaxBlocks; }
long siss_opt_scalars_words(void) { return sizeof(StepScalars) / sizeof(float); }

// pass 1 + on-device scalars.  `scalars` (16 floats, zero-initialised once; holds the step count).
int siss_grad_norms_scale(const float* gx, const float* ga, long n, int mode, float knob, float max_norm,
                          float beta1, float beta2, double* partials, float* scalars, void* stream) {
    SISS_CHECK_ARG(gx && ga && partials && scalars && n > 0 && mode >= 0 && mode <= 2);
    SISS_CHECK_ARG(((uintptr_t)gx | (uintptr_t)ga) % 16 == 0);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = grid_for(n);
    norms_kernel<<<nblk, kThreads, 0, s>>>(gx, ga, n, partials);
    scalars_kernel<<<1, kThreads, 0, s>>>(partials, nblk, mode, knob, max_norm, beta1, beta2,
                                    reinterpret_cast<StepScalars*>(scalars));
    SISS_LAUNCH_RET();
}

// The two halves of siss_grad_norms_scale on their own, for the SHARDED data-parallel update (reduce-scatter
// -> shard-local recombine / clip / AdamW -> all-gather): every rank sums its parameter shard, the three sums
// (|g_x|^2, |g_a|^2, <g_x, g_a>) are all-reduced by the host, and the step scalars are formed from the global sums.
// siss_grad_norm_partials writes 3 doubles per block to `partials` and returns the block count in *nblk_out (host int).
int siss_grad_norm_partials(const float* gx, const float* ga, long n, double* partials, int* nblk_out, void* stream) {
    SISS_CHECK_ARG(gx && ga && partials && nblk_out && n > 0);
    SISS_CHECK_ARG(((uintptr_t)gx | (uintptr_t)ga) % 16 == 0);
    const int nblk = grid_for(n);
    norms_kernel<<<nblk, kThreads, 0, (hipStream_t)stream>>>(gx, ga, n, partials);
    *nblk_out = nblk;
    SISS_LAUNCH_RET();
}
// sums: `nrows` rows of 3 doubles (device) whose column sums are the GLOBAL |g_x|^2, |g_a|^2, <g_x, g_a>.
int siss_grad_scalars(const double* sums, int nrows, int mode, float knob, float max_norm, float beta1, float beta2,
                      float* scalars, void* stream) {
    SISS_CHECK_ARG(sums && scalars && nrows > 0 && mode >= 0 && mode <= 2);
    scalars_kernel<<<1, kThreads, 0, (hipStream_t)stream>>>(sums, nrows, mode, knob, max_norm, beta1, beta2,
                                                            reinterpret_cast<StepScalars*>(scalars));
    SISS_LAUNCH_RET();
}

// pass 2.  shadow (bf16 copy of the updated parameters) and g_out (final clipped gradient) are optional.
int siss_recombine_clip_adamw(const float* gx, const float* ga, float* p, float* m, float* v, void* shadow,
                              float* g_out, long n, float lr, float beta1, float beta2, float eps, float wd,
                              const float* scalars, void* stream) {
    SISS_CHECK_ARG(gx && ga && p && m && v && scalars && n > 0);
    SISS_CHECK_ARG(((uintptr_t)gx | (uintptr_t)ga | (uintptr_t)p | (uintptr_t)m | (uintptr_t)v) % 16 == 0);
    SISS_CHECK_ARG(!shadow || (uintptr_t)shadow % 8 == 0);
    recombine_adamw_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>(
        gx, ga, p, m, v, reinterpret_cast<bf16_t*>(shadow), g_out, n, lr, beta1, beta2, eps, wd,
        reinterpret_cast<const StepScalars*>(scalars));
    SISS_LAUNCH_RET();
}

int siss_cast_f32_bf16(const float* src, void* dst, long n, void* stream) {
    SISS_CHECK_ARG(src && dst && n > 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 8 == 0);
    cast_bf16_kernel<<<grid_for(n), kThreads, 0, (hipStream_t)stream>>>(src, reinterpret_cast<bf16_t*>(dst), n);
    SISS_LAUNCH_RET();
}

int siss_conv_weight_dgrad_layout(const float* w, void* wt, int taps, int co, int ci, void* stream) {
    SISS_CHECK_ARG(w && wt && taps > 0 && co > 0 && ci > 0);
    dim3 grid(cdiv(ci, 32), cdiv(co, 32), taps), block(32, 8);
    conv_weight_dgrad_kernel<<<grid, block, 0, (hipStream_t)stream>>>(w, reinterpret_cast<bf16_t*>(wt), taps, co, ci);
    SISS_LAUNCH_RET();
}

// jobs: device array of njobs records {long src_off, long dst_off, int taps, int co, int ci, int tile0}
// (tile0 = running sum of taps*ceil(co/64)*ceil(ci/64)); total_tiles = the grid size.
int siss_conv_weight_dgrad_multi(const float* flat, void* wt_all, const void* jobs, int njobs, int total_tiles,
                                 void* stream) {
    SISS_CHECK_ARG(flat && wt_all && jobs && njobs > 0 && total_tiles > 0);
    conv_weight_dgrad_multi_kernel<<<total_tiles, dim3(64, 4), 0, (hipStream_t)stream>>>(
        flat, reinterpret_cast<bf16_t*>(wt_all), reinterpret_cast<const WtJob*>(jobs), njobs);
    SISS_LAUNCH_RET();
}

// The same from the bf16 operand shadow of the flat buffer (same element offsets; it must hold the rounded master: the fused
// AdamW launch and siss_cast_f32_bf16 leave it so): half the bytes read, 16-B accesses.
int siss_conv_weight_dgrad_multi_bf16(const void* shadow, void* wt_all, const void* jobs, int njobs, int total_tiles,
                                      void* stream) {
    SISS_CHECK_ARG(shadow && wt_all && jobs && njobs > 0 && total_tiles > 0);
    SISS_CHECK_ARG(((uintptr_t)shadow | (uintptr_t)wt_all) % 16 == 0);
    conv_weight_dgrad_multi_bf16_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>(
        reinterpret_cast<const bf16_t*>(shadow), reinterpret_cast<bf16_t*>(wt_all), reinterpret_cast<const WtJob*>(jobs), njobs);
    SISS_LAUNCH_RET();
}


// Phase weights of a sub-pixel upsample convolution from its f32 master weights w [9][Co][Ci]: wf [4][4][Co][Ci] bf16 (plane-major:
// the fprop operand of plane p is wf + p * 4 * Co * Ci, four panels) and wd [16][Ci][Co] bf16 (the dgrad operand, panel plane * 4 + tap).
int siss_upsample_phase_weights(const float* w, void* wf, void* wd, int Co, int Ci, void* stream) {
    SISS_CHECK_ARG(w && wf && wd && Co > 0 && Ci > 0);
    long nb = ((long)16 * Co * Ci + 255) / 256;
    if (nb > 4096) nb = 4096;
    upsample_phase_weights_kernel<bf16_t><<<(int)nb, 256, 0, (hipStream_t)stream>>>(w, (bf16_t*)wf, (bf16_t*)wd, Co, Ci);
    SISS_LAUNCH_RET();
}
// The same with f32 phase weights (the f32 parity mode: no rounding at all)
int siss_upsample_phase_weights_f32(const float* w, void* wf, void* wd, int Co, int Ci, void* stream) {
    SISS_CHECK_ARG(w && wf && wd && Co > 0 && Ci > 0);
    long nb = ((long)16 * Co * Ci + 255) / 256;
    if (nb > 4096) nb = 4096;
    upsample_phase_weights_kernel<float><<<(int)nb, 256, 0, (hipStream_t)stream>>>(w, (float*)wf, (float*)wd, Co, Ci);
    SISS_LAUNCH_RET();
}
// The adjoint for the weight gradient: the 16 phase-tap gradients dW4 [nsets][4][4][Co][Ci] f32 folded onto the nine taps,
// dW[set * set_stride + (tap * Co + co) * Ci + ci] += ...
int siss_upsample_phase_wgrad_fold(const float* dW4, float* dW, long set_stride, int nsets, int Co, int Ci, void* stream) {
    SISS_CHECK_ARG(dW4 && dW && nsets > 0 && nsets <= 65535 && Co > 0 && Ci > 0);
    long nb = ((long)9 * Co * Ci + 255) / 256;
    if (nb > 4096) nb = 4096;
    upsample_phase_wgrad_fold_kernel<<<dim3((int)nb, nsets), 256, 0, (hipStream_t)stream>>>(dW4, dW, set_stride, Co, Ci);
    SISS_LAUNCH_RET();
}

}  // extern "C"
