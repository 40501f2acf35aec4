// Fused SISS pre/post kernels (reference math:
// losses/ddpm_deletion_loss.py:12-53 and delete_celeb.py:602-603, :686-687).
//
//   pre : (x0, a0, noise, t, u) -> x_mix = keep ? q_sample(x0) : q_sample(a0),
//         dist_x, dist_a, iw_x, iw_a                      (one pass over 3 inputs + 1 output)
//   post: (pred, x_mix, x0, a0, iw) -> cotangents c_x, c_a seeding the dual backward,
//         per-sample sums of loss_x / loss_a               (one pass over 4 inputs + 2 outputs)
//
// Both are HBM-bound streaming kernels: NCHW rows are contiguous per sample, every lane
// moves 16 B (bf16x8) or 16 B (f32x4) per access, per-sample reductions go
// wave -> block -> a [B][nblk] partial slab in f64 (no atomics; bit-reproducible).
#include "common.h"

namespace {

constexpr int kThreads = 256;
constexpr int kVec = 8;  // elements per lane per iteration

template <bool BF16>
struct Ld {
    static __device__ __forceinline__ void load8(const void* p, long i, float (&v)[8]) {
        if constexpr (BF16) {
            u32x4_t r = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const bf16_t*>(p) + i);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[2 * j] = __builtin_bit_cast(float, r[j] << 16);
                v[2 * j + 1] = __builtin_bit_cast(float, r[j] & 0xffff0000u);
            }
        } else {
            const f32x4_t* q = reinterpret_cast<const f32x4_t*>(reinterpret_cast<const float*>(p) + i);
            f32x4_t a = q[0], b = q[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
        }
    }
    static __device__ __forceinline__ float load1(const void* p, long i) {
        if constexpr (BF16) return bf2f(reinterpret_cast<const bf16_t*>(p)[i]);
        else return reinterpret_cast<const float*>(p)[i];
    }
    static __device__ __forceinline__ void store8(void* p, long i, const float (&v)[8]) {
        if constexpr (BF16) {
            u32x4_t r;
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = pack_bf2(v[2 * j], v[2 * j + 1]);
            *reinterpret_cast<u32x4_t*>(reinterpret_cast<bf16_t*>(p) + i) = r;
        } else {
            f32x4_t* q = reinterpret_cast<f32x4_t*>(reinterpret_cast<float*>(p) + i);
            q[0] = f32x4_t{v[0], v[1], v[2], v[3]};
            q[1] = f32x4_t{v[4], v[5], v[6], v[7]};
        }
    }
    static __device__ __forceinline__ void store1(void* p, long i, float v) {
        if constexpr (BF16) reinterpret_cast<bf16_t*>(p)[i] = f2bf(v);
        else reinterpret_cast<float*>(p)[i] = v;
    }
};

// DDPMScheduler.add_noise coefficient rule: alphas_cumprod is cast to the sample dtype FIRST.
template <bool BF16>
__device__ __forceinline__ void noise_coeffs(float ac, float& a, float& b) {
    if constexpr (BF16) {
        float acb = bfround(ac);
        a = bfround(sqrtf(acb));
        b = bfround(sqrtf(bfround(1.f - acb)));
    } else {
        a = sqrtf(ac);
        b = sqrtf(1.f - ac);
    }
}
template <bool BF16>
__device__ __forceinline__ float q_sample(float a, float b, float x, float n) {
    // two products and one sum, each rounded on its own (no fma contraction): matches torch
    if constexpr (BF16) return bfround(bfround(a * x) + bfround(b * n));
    else return __fadd_rn(__fmul_rn(a, x), __fmul_rn(b, n));
}

__device__ __forceinline__ void block_reduce2(double& s0, double& s1, double* sh) {
    s0 = wave_sum_d(s0);
    s1 = wave_sum_d(s1);
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) { sh[2 * w] = s0; sh[2 * w + 1] = s1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double a = 0, b = 0;
        for (int i = 0; i < kThreads / 64; ++i) { a += sh[2 * i]; b += sh[2 * i + 1]; }
        s0 = a; s1 = b;
    }
}

template <bool BF16>
__global__ __launch_bounds__(kThreads) void mixture_main_kernel(
    const void* __restrict__ x0, const void* __restrict__ a0, const void* __restrict__ noise,
    const int64_t* __restrict__ t, const float* __restrict__ u, const float* __restrict__ ac_tab,
    const float* __restrict__ gamma_tab, float lambd, long chw, void* __restrict__ x_mix,
    double* __restrict__ partials) {
    __shared__ double sh[2 * kThreads / 64];
    const int n = blockIdx.y;
    const long tn = t[n];
    float ca, cb;
    noise_coeffs<BF16>(ac_tab[tn], ca, cb);
    const float gamma = gamma_tab[tn];
    const bool keep = u[n] > lambd;  // ddpm_deletion_loss.py:18
    const long base = (long)n * chw;
    double sx = 0, sa = 0;
    const long nvec = chw / kVec;
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long)gridDim.x * kThreads) {
        float vx[8], va[8], vn[8], vs[8], vm[8];
        const long e = base + i * kVec;
        Ld<BF16>::load8(x0, e, vx);
        Ld<BF16>::load8(a0, e, va);
        Ld<BF16>::load8(noise, e, vn);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            vs[j] = keep ? vx[j] : va[j];
            vm[j] = q_sample<BF16>(ca, cb, vs[j], vn[j]);
            float rx = __fsub_rn(vm[j], __fmul_rn(gamma, vx[j]));
            float ra = __fsub_rn(vm[j], __fmul_rn(gamma, va[j]));
            sx += (double)__fmul_rn(rx, rx);
            sa += (double)__fmul_rn(ra, ra);
        }
        Ld<BF16>::store8(x_mix, e, vm);
    }
    // ragged tail (chw not a multiple of 8): block 0 only
    if (blockIdx.x == 0) {
        for (long k = nvec * kVec + threadIdx.x; k < chw; k += kThreads) {
            const long e = base + k;
            float x = Ld<BF16>::load1(x0, e), a = Ld<BF16>::load1(a0, e), nn = Ld<BF16>::load1(noise, e);
            float m = q_sample<BF16>(ca, cb, keep ? x : a, nn);
            float rx = __fsub_rn(m, __fmul_rn(gamma, x)), ra = __fsub_rn(m, __fmul_rn(gamma, a));
            sx += (double)__fmul_rn(rx, rx);
            sa += (double)__fmul_rn(ra, ra);
            Ld<BF16>::store1(x_mix, e, m);
        }
    }
    block_reduce2(sx, sa, sh);
    if (threadIdx.x == 0) {
        double* p = partials + ((long)n * gridDim.x + blockIdx.x) * 2;
        p[0] = sx; p[1] = sa;
    }
}

// Class-surface variant: the caller already holds both noisy batches (the reference's
// all_samples_dict / deletion_samples_dict 'noisy_latents'); only the row select + distances remain.
template <bool BF16>
__global__ __launch_bounds__(kThreads) void mixture_select_kernel(
    const void* __restrict__ nk, const void* __restrict__ nf, const void* __restrict__ x0,
    const void* __restrict__ a0, const int64_t* __restrict__ t, const float* __restrict__ u,
    const float* __restrict__ gamma_tab, float lambd, long chw, void* __restrict__ x_mix,
    double* __restrict__ partials) {
    __shared__ double sh[2 * kThreads / 64];
    const int n = blockIdx.y;
    const float gamma = gamma_tab[t[n]];
    const bool keep = u[n] > lambd;
    const long base = (long)n * chw;
    double sx = 0, sa = 0;
    for (long k = (long)blockIdx.x * kThreads + threadIdx.x; k < chw; k += (long)gridDim.x * kThreads) {
        const long e = base + k;
        const float m = keep ? Ld<BF16>::load1(nk, e) : Ld<BF16>::load1(nf, e);
        const float x = Ld<BF16>::load1(x0, e), a = Ld<BF16>::load1(a0, e);
        const float rx = __fsub_rn(m, __fmul_rn(gamma, x)), ra = __fsub_rn(m, __fmul_rn(gamma, a));
        sx += (double)__fmul_rn(rx, rx);
        sa += (double)__fmul_rn(ra, ra);
        Ld<BF16>::store1(x_mix, e, m);
    }
    block_reduce2(sx, sa, sh);
    if (threadIdx.x == 0) {
        double* p = partials + ((long)n * gridDim.x + blockIdx.x) * 2;
        p[0] = sx; p[1] = sa;
    }
}

// One thread per sample: fold the partial slab, then ddpm_deletion_loss.py:33-45 literally.
__global__ void mixture_finalize_kernel(const double* __restrict__ partials, int nblk,
                                        const int64_t* __restrict__ t, const float* __restrict__ gamma_tab,
                                        const float* __restrict__ sigma_tab, float lambd, int B,
                                        float* __restrict__ gamma_t, float* __restrict__ sigma_t,
                                        float* __restrict__ dist_x, float* __restrict__ dist_a,
                                        float* __restrict__ iw_x, float* __restrict__ iw_a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B) return;
    double sx = 0, sa = 0;
    for (int i = 0; i < nblk; ++i) { sx += partials[((long)n * nblk + i) * 2]; sa += partials[((long)n * nblk + i) * 2 + 1]; }
    const long tn = t[n];
    const float g = gamma_tab[tn], s = sigma_tab[tn];
    const float den = __fmul_rn(2.f, __fmul_rn(s, s));
    const float dx = (float)sx / den, da = (float)sa / den;
    // exp may overflow to +inf: 1/inf = 0 gives the saturated weights {0, 1/(1-lambd)} -- kept on purpose
    const float r_ax = expf(__fsub_rn(dx, da)), r_xa = expf(__fsub_rn(da, dx));
    gamma_t[n] = g; sigma_t[n] = s; dist_x[n] = dx; dist_a[n] = da;
    iw_x[n] = 1.f / __fadd_rn(1.f - lambd, __fmul_rn(lambd, r_ax));
    iw_a[n] = 1.f / __fadd_rn(__fmul_rn(1.f - lambd, r_xa), lambd);
}

// post: cotangents + per-sample loss sums.  MODE 0 = SISS (two targets from x_mix), 1 = plain MSE
// against `target` (SISS-No-IS / NegGrad / naive: ddpm_deletion_loss.py:62,65,84,93).
template <bool BF16, int MODE>
__global__ __launch_bounds__(kThreads) void loss_seed_kernel(
    const float* __restrict__ pred, const void* __restrict__ x_mix, const void* __restrict__ x0,
    const void* __restrict__ a0, const float* __restrict__ gamma_t, const float* __restrict__ sigma_t,
    const float* __restrict__ iw_x, const float* __restrict__ iw_a, float scale, long chw,
    float* __restrict__ c_x, float* __restrict__ c_a, float* __restrict__ loss_x,
    float* __restrict__ loss_a, double* __restrict__ partials) {
    __shared__ double sh[2 * kThreads / 64];
    const int n = blockIdx.y;
    const long base = (long)n * chw;
    float g = 0, s = 1, wx = 1, wa = 1;
    if (MODE == 0) { g = gamma_t[n]; s = sigma_t[n]; wx = iw_x[n]; wa = iw_a[n]; }
    const float kx = 2.f * wx * scale, ka = 2.f * wa * scale;
    double sx = 0, sa = 0;
    const long nvec = chw / kVec;
    auto one = [&](float p, float m, float x, float a, float& ox, float& oa, float& lx, float& la) {
        float ex, ea;
        if (MODE == 0) {
            ex = __fsub_rn(m, __fmul_rn(g, x)) / s;   // ddpm_deletion_loss.py:26
            ea = __fsub_rn(m, __fmul_rn(g, a)) / s;   // :27
        } else {
            ex = ea = m;  // x_mix carries the regression target in MODE 1
        }
        const float dx = __fsub_rn(p, ex), da = __fsub_rn(p, ea);
        lx = __fmul_rn(dx, dx); la = __fmul_rn(da, da);   // :29-30
        ox = kx * dx; oa = ka * da;
        sx += (double)lx; sa += (double)la;
    };
    auto st8 = [](float* dst, long e, const float (&v)[8]) {
        if (!dst) return;
        reinterpret_cast<f32x4_t*>(dst + e)[0] = f32x4_t{v[0], v[1], v[2], v[3]};
        reinterpret_cast<f32x4_t*>(dst + e)[1] = f32x4_t{v[4], v[5], v[6], v[7]};
    };
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < nvec; i += (long)gridDim.x * kThreads) {
        const long e = base + i * kVec;
        float vp[8], vm[8], vx[8] = {}, va[8] = {}, ox[8], oa[8], lx[8], la[8];
        const f32x4_t* q = reinterpret_cast<const f32x4_t*>(pred + e);
        f32x4_t p0 = q[0], p1 = q[1];
#pragma unroll
        for (int j = 0; j < 4; ++j) { vp[j] = p0[j]; vp[4 + j] = p1[j]; }
        Ld<BF16>::load8(x_mix, e, vm);
        if (MODE == 0) { Ld<BF16>::load8(x0, e, vx); Ld<BF16>::load8(a0, e, va); }
#pragma unroll
        for (int j = 0; j < 8; ++j) one(vp[j], vm[j], vx[j], va[j], ox[j], oa[j], lx[j], la[j]);
        st8(c_x, e, ox); st8(c_a, e, oa); st8(loss_x, e, lx); st8(loss_a, e, la);
    }
    if (blockIdx.x == 0) {  // ragged tail (chw not a multiple of 8)
        for (long k = nvec * kVec + threadIdx.x; k < chw; k += kThreads) {
            const long e = base + k;
            float ox, oa, lx, la;
            one(pred[e], Ld<BF16>::load1(x_mix, e), MODE == 0 ? Ld<BF16>::load1(x0, e) : 0.f,
                MODE == 0 ? Ld<BF16>::load1(a0, e) : 0.f, ox, oa, lx, la);
            if (c_x) c_x[e] = ox;
            if (c_a) c_a[e] = oa;
            if (loss_x) loss_x[e] = lx;
            if (loss_a) loss_a[e] = la;
        }
    }
    block_reduce2(sx, sa, sh);
    if (threadIdx.x == 0) {
        double* p = partials + ((long)n * gridDim.x + blockIdx.x) * 2;
        p[0] = sx; p[1] = sa;
    }
}

__global__ void fold_partials_kernel(const double* __restrict__ partials, int nblk, int B,
                                     float* __restrict__ sum_x, float* __restrict__ sum_a) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B) return;
    double sx = 0, sa = 0;
    for (int i = 0; i < nblk; ++i) { sx += partials[((long)n * nblk + i) * 2]; sa += partials[((long)n * nblk + i) * 2 + 1]; }
    sum_x[n] = (float)sx;
    if (sum_a) sum_a[n] = (float)sa;
}


// One DDPM reverse step x_t -> x_{t-1} (epsilon prediction, clip_sample, fixed_small variance), fused:
//   x0 = clamp((x - sqrt_b * eps) / sqrt_a, -1, 1);  x_prev = c_x0 * x0 + c_xt * x + sigma * noise
// (diffusers DDPMScheduler.step as used by evaluate.py:37-79).  f32 NCHW in / out, HBM-bound.
__global__ __launch_bounds__(kThreads) void ddpm_step_kernel(const float* __restrict__ x, const float* __restrict__ eps,
                                                             const float* __restrict__ noise, float* __restrict__ out,
                                                             long n, float sqrt_a, float sqrt_b, float c_x0, float c_xt,
                                                             float sigma, int clip) {
    for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
        float x0 = (x[i] - sqrt_b * eps[i]) / sqrt_a;
        if (clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
        float o = c_x0 * x0 + c_xt * x[i];
        if (noise) o += sigma * noise[i];
        out[i] = o;
    }
}

inline int blocks_for(long chw) {
    long v = chw / kVec / kThreads;
    if (v < 1) v = 1;
    if (v > 64) v = 64;
    return (int)v;
}

}  // namespace

extern "C" {

// Number of f64 words the caller must provide in `partials` for a given (B, chw).
long siss_loss_partials_words(int B, long chw) { return (long)B * blocks_for(chw) * 2; }

int siss_mixture_fwd(const void* x0, const void* a0, const void* noise, int in_bf16, const int64_t* t,
                     const float* u, const float* alphas_cumprod, const float* gamma_tab,
                     const float* sigma_tab, float lambd, int B, long chw, void* x_mix, float* gamma_t,
                     float* sigma_t, float* dist_x, float* dist_a, float* iw_x, float* iw_a,
                     double* partials, void* stream) {
    SISS_CHECK_ARG(x0 && a0 && noise && t && u && alphas_cumprod && gamma_tab && sigma_tab && x_mix);
    SISS_CHECK_ARG(gamma_t && sigma_t && dist_x && dist_a && iw_x && iw_a && partials);
    SISS_CHECK_ARG(B > 0 && chw > 0);
    SISS_CHECK_ARG(((uintptr_t)x0 | (uintptr_t)a0 | (uintptr_t)noise | (uintptr_t)x_mix) % 16 == 0);
    SISS_CHECK_ARG(chw % (in_bf16 ? 8 : 4) == 0 || B == 1);  // per-sample rows stay 16-B aligned
    hipStream_t s = (hipStream_t)stream;
    const int nblk = blocks_for(chw);
    dim3 grid(nblk, B);
    if (in_bf16)
        mixture_main_kernel<true><<<grid, kThreads, 0, s>>>(x0, a0, noise, t, u, alphas_cumprod, gamma_tab, lambd, chw, x_mix, partials);
    else
        mixture_main_kernel<false><<<grid, kThreads, 0, s>>>(x0, a0, noise, t, u, alphas_cumprod, gamma_tab, lambd, chw, x_mix, partials);
    mixture_finalize_kernel<<<cdiv(B, 64), 64, 0, s>>>(partials, nblk, t, gamma_tab, sigma_tab, lambd, B,
                                                       gamma_t, sigma_t, dist_x, dist_a, iw_x, iw_a);
    SISS_LAUNCH_RET();
}

// Same outputs as siss_mixture_fwd, but from caller-provided noisy batches (DDPMDeletionLoss surface).
int siss_mixture_select(const void* noisy_keep, const void* noisy_forget, const void* x0, const void* a0,
                        int in_bf16, const int64_t* t, const float* u, const float* gamma_tab,
                        const float* sigma_tab, float lambd, int B, long chw, void* x_mix, float* gamma_t,
                        float* sigma_t, float* dist_x, float* dist_a, float* iw_x, float* iw_a,
                        double* partials, void* stream) {
    SISS_CHECK_ARG(noisy_keep && noisy_forget && x0 && a0 && t && u && gamma_tab && sigma_tab && x_mix);
    SISS_CHECK_ARG(gamma_t && sigma_t && dist_x && dist_a && iw_x && iw_a && partials && B > 0 && chw > 0);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = blocks_for(chw);
    dim3 grid(nblk, B);
    if (in_bf16)
        mixture_select_kernel<true><<<grid, kThreads, 0, s>>>(noisy_keep, noisy_forget, x0, a0, t, u, gamma_tab, lambd, chw, x_mix, partials);
    else
        mixture_select_kernel<false><<<grid, kThreads, 0, s>>>(noisy_keep, noisy_forget, x0, a0, t, u, gamma_tab, lambd, chw, x_mix, partials);
    mixture_finalize_kernel<<<cdiv(B, 64), 64, 0, s>>>(partials, nblk, t, gamma_tab, sigma_tab, lambd, B,
                                                       gamma_t, sigma_t, dist_x, dist_a, iw_x, iw_a);
    SISS_LAUNCH_RET();
}

int siss_loss_bwd_seed(const float* pred, const void* x_mix, const void* x0, const void* a0, int in_bf16,
                       const float* gamma_t, const float* sigma_t, const float* iw_x, const float* iw_a,
                       float scale, int B, long chw, float* c_x, float* c_a, float* loss_x, float* loss_a,
                       float* sum_loss_x, float* sum_loss_a, double* partials, void* stream) {
    SISS_CHECK_ARG(pred && x_mix && x0 && a0 && gamma_t && sigma_t && iw_x && iw_a && partials);
    SISS_CHECK_ARG(sum_loss_x && sum_loss_a && B > 0 && chw > 0);
    SISS_CHECK_ARG(chw % 8 == 0 || B == 1);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = blocks_for(chw);
    dim3 grid(nblk, B);
    if (in_bf16)
        loss_seed_kernel<true, 0><<<grid, kThreads, 0, s>>>(pred, x_mix, x0, a0, gamma_t, sigma_t, iw_x, iw_a, scale, chw, c_x, c_a, loss_x, loss_a, partials);
    else
        loss_seed_kernel<false, 0><<<grid, kThreads, 0, s>>>(pred, x_mix, x0, a0, gamma_t, sigma_t, iw_x, iw_a, scale, chw, c_x, c_a, loss_x, loss_a, partials);
    fold_partials_kernel<<<cdiv(B, 64), 64, 0, s>>>(partials, nblk, B, sum_loss_x, sum_loss_a);
    SISS_LAUNCH_RET();
}

// Plain squared error against `target` (No-IS / NegGrad / naive): c = 2*scale*(pred-target).
int siss_mse_bwd_seed(const float* pred, const void* target, int target_bf16, float scale, int B, long chw,
                      float* c, float* loss, float* sum_loss, double* partials, void* stream) {
    SISS_CHECK_ARG(pred && target && sum_loss && partials && B > 0 && chw > 0);
    SISS_CHECK_ARG(chw % 8 == 0 || B == 1);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = blocks_for(chw);
    dim3 grid(nblk, B);
    if (target_bf16)
        loss_seed_kernel<true, 1><<<grid, kThreads, 0, s>>>(pred, target, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, scale, chw, c, nullptr, loss, nullptr, partials);
    else
        loss_seed_kernel<false, 1><<<grid, kThreads, 0, s>>>(pred, target, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, scale, chw, c, nullptr, loss, nullptr, partials);
    fold_partials_kernel<<<cdiv(B, 64), 64, 0, s>>>(partials, nblk, B, sum_loss, nullptr);
    SISS_LAUNCH_RET();
}

// out = c_x0 * clamp((x - sqrt_b*eps)/sqrt_a) + c_xt * x + sigma * noise   (noise may be NULL at t = 0)
int siss_ddpm_step(const float* x, const float* eps, const float* noise, float* out, long n, float sqrt_a, float sqrt_b,
                   float c_x0, float c_xt, float sigma, int clip, void* stream) {
    SISS_CHECK_ARG(x && eps && out && n > 0 && sqrt_a > 0.f);
    long nb = (n + kThreads - 1) / kThreads;
    if (nb > 4096) nb = 4096;
    ddpm_step_kernel<<<(int)nb, kThreads, 0, (hipStream_t)stream>>>(x, eps, noise, out, n, sqrt_a, sqrt_b, c_x0, c_xt, sigma, clip);
    SISS_LAUNCH_RET();
}

}  // extern "C"
