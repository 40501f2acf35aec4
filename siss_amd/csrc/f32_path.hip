// The f32 PARITY MODE of the engine (round 4): `mixed_precision: null` -- the reference's shipped default,
// /root/reference config/delete_celeb.yaml:103 -- with every activation, operand and product in f32, so that the HIP network can
// be held against the fp32 oracle at 1e-4 instead of at bf16-wide bounds (SURVEY.md section 8c's fp32 row).
//
// Same layouts (padded NHWC rows, flat f32 parameters), same entry-point argument lists as the bf16 launchers -- the engine code
// (siss_amd/unet.py) is the SAME for both element types and only picks the `_f32` entry point -- but simple kernels: this is an
// instrument, not a fast path.  The products run on v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate: an fmaf chain bit for bit, at
// 1/16 of the bf16 MFMA rate), one wave per 16 x 16 output tile with operands straight from global memory; the fused forms of the
// product path (persistent 3x3 kernel, folded shortcuts, GroupNorm statistics from the conv epilogue, depth-to-space epilogue,
// grouped wgrads, slab GroupNorm, flash attention) are bf16-only and are bypassed by the engine in this mode.
// Here: siss_gemm_nt_f32, siss_gemm_tn_f32, siss_groupnorm_fwd_ld_f32, siss_groupnorm_bwd_ld_f32, siss_conv_out_fprop_f32.
// The data-movement, softmax and weight-copy kernels are templates on the element type in their own files.
#include "common.h"

namespace {

constexpr int kMaxPanelsF = 16;           // (16: the (plane, tap) panels of a sub-pixel upsample convolution's dgrad)

struct NTF {
    const float* A; const float* W; float* C;
    const float* bias; const float* rowbias; const float* R; const float* rowsub;
    long lda, ldc, ldr, ldrb, strideA, strideW, strideC;
    int M, N, Kp, npanels, rows_per_image, Hp, Wp, mul_r, d2s, alpha_cols;
    float alpha;
    int shift[kMaxPanelsF], coff[kMaxPanelsF];
};

// C[bz][r][n] = epi( sum_p sum_k A[bz][r + shift_p][coff_p + k] W[bz][p][n][k] ): the arithmetic of nt_common.h's epilogue --
// (acc - rowsub[r]) * alpha + bias[n] + rowbias[image][n]; halo rows of a pixel grid are written as zeros; then + R (or * R).
// One wave per 16 x 16 tile; a lane loads 4 consecutive k of its A row and of its W row per four MFMAs (the k-slot of MFMA j in
// lane group q is k0 + 4 q + j on BOTH operands: any bijection onto the 16 k of a step gives the same sum).
__global__ __launch_bounds__(256) void gemm_nt_f32_kernel(const NTF p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tn = blockIdx.y * 4 + wv, tm = blockIdx.x, bz = blockIdx.z;          // (M tiles on grid.x: no 65535-tile limit on the rows)
    if (tn * 16 >= p.N) return;
    const int m0 = tm * 16, n0 = tn * 16;
    const int li = lane & 15, q = lane >> 4;
    const int arow = m0 + li, wcol = n0 + li;
    const bool aok = arow < p.M, wok = wcol < p.N;
    const float* A = p.A + (long)bz * p.strideA;
    const float* W = p.W + (long)bz * p.strideW;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int pn = 0; pn < p.npanels; ++pn) {
        const float* ap = A + ((long)arow + p.shift[pn]) * p.lda + p.coff[pn] + 4 * q;
        const float* wp = W + ((long)pn * p.N + wcol) * p.Kp + 4 * q;
        for (int k0 = 0; k0 < p.Kp; k0 += 16) {
            const f32x4_t a = aok ? *reinterpret_cast<const f32x4_t*>(ap + k0) : f32x4_t{0.f, 0.f, 0.f, 0.f};
            const f32x4_t b = wok ? *reinterpret_cast<const f32x4_t*>(wp + k0) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
        }
    }
    // acc[r]: row m0 + 4 q + r, column n0 + li
    const int n = n0 + li;
    if (n >= p.N) return;
    const float al = (p.alpha_cols == 0 || n < p.alpha_cols) ? p.alpha : 1.f;
    const float bs = p.bias ? p.bias[n] : 0.f;
    float* C = p.C + (long)bz * p.strideC;
    const float* R = p.R ? p.R + (long)bz * p.strideC : nullptr;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int m = m0 + 4 * q + r;
        if (m >= p.M) continue;
        float v = acc[r];
        if (p.rowsub) v -= p.rowsub[(long)bz * p.M + m];
        v = v * al + bs;
        const int img = m / p.rows_per_image;
        if (p.rowbias) v += p.rowbias[(long)img * p.ldrb + n];
        long ro = m;
        if (p.Hp > 0) {
            const int rem = m - img * p.rows_per_image;
            const int y = rem / p.Wp, x = rem - y * p.Wp;
            const bool halo = (y == 0) | (y == p.Hp - 1) | (x == 0) | (x == p.Wp - 1);
            if (p.d2s) {
                if (halo) continue;
                const int pl = p.d2s - 1, wf = 2 * p.Wp - 2;
                ro = (long)img * (2 * p.Hp - 2) * wf + (long)(2 * y - 1 + (pl >> 1)) * wf + (2 * x - 1 + (pl & 1));
            }
            if (halo) { C[ro * p.ldc + n] = 0.f; continue; }
        }
        if (R) v = p.mul_r ? v * R[ro * p.ldr + n] : v + R[ro * p.ldr + n];
        C[ro * p.ldc + n] = v;
    }
}

struct TNF {
    const float* Y; const float* X; float* dW; float* dbias; float* dbias2;
    long ldy, ldx, set_stride, x_set_rows, bias_stride;
    int N, C, npanels, nsets, rows_per_set, row_begin, row_end, overwrite;
    int shift[kMaxPanelsF], coff[kMaxPanelsF];
};

// dW[set][p][n][c] (+)= sum_{r in [row_begin, row_end)} Y[set rows + r][n] X[set x rows + r + shift_p][coff_p + c]: one wave per
// 16 x 16 tile of one (set, panel); the reduction index is the MFMA's k (4 rows per instruction, lane group q takes row r0 + q).
__global__ __launch_bounds__(256) void gemm_tn_f32_kernel(const TNF p) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tc = blockIdx.x * 4 + wv, tn = blockIdx.y;
    const int pn = blockIdx.z % p.npanels, set = blockIdx.z / p.npanels;
    if (tc * 16 >= p.C) return;
    const int li = lane & 15, q = lane >> 4;
    const int n = tn * 16 + li, c = tc * 16 + li;
    const float* Y = p.Y + ((long)set * p.rows_per_set) * p.ldy + n;
    const float* X = p.X + ((long)set * p.x_set_rows + p.shift[pn]) * p.ldx + p.coff[pn] + c;
    const bool nok = n < p.N, cok = c < p.C;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int r0 = p.row_begin; r0 < p.row_end; r0 += 4) {
        const int r = r0 + q;
        const bool rok = r < p.row_end;
        const float a = (rok && nok) ? Y[(long)r * p.ldy] : 0.f;
        const float b = (rok && cok) ? X[(long)r * p.ldx] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    if (!cok) return;
    float* out = p.dW + (long)set * p.set_stride + (long)pn * p.N * p.C;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int nn = tn * 16 + 4 * q + r;
        if (nn >= p.N) continue;
        float* d = out + (long)nn * p.C + c;
        *d = p.overwrite ? acc[r] : *d + acc[r];
    }
}
// dbias[set][n] += column sums of Y over the set's rows (one thread per (set, n): the bias gradient that rides in the bf16 wgrad)
__global__ void tn_bias_f32_kernel(const TNF p) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, set = blockIdx.y;
    if (n >= p.N) return;
    const float* Y = p.Y + ((long)set * p.rows_per_set) * p.ldy + n;
    double a = 0.0;
    for (int r = p.row_begin; r < p.row_end; ++r) a += (double)Y[(long)r * p.ldy];
    p.dbias[(long)set * p.bias_stride + n] += (float)a;
    if (p.dbias2) p.dbias2[(long)set * p.bias_stride + n] += (float)a;
}

// ---------------------------------------------------------------- GroupNorm (+ SiLU), one block per (sample, group)
constexpr int kGT = 256;

__device__ __forceinline__ double block_sum_d(double v, double* sh) {
    v = wave_sum_d(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    for (int i = 0; i < kGT / 64; ++i) t += sh[i];
    return t;
}
__device__ __forceinline__ long gn_prow(int n, int pi, int H, int W) {
    const int y = pi / W, x = pi - y * W;
    return ((long)n * (H + 2) + y + 1) * (W + 2) + x + 1;
}

__global__ __launch_bounds__(kGT) void gn_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, float* __restrict__ y,
                                                        float* __restrict__ mean, float* __restrict__ rstd, int H, int W, int C,
                                                        int G, float eps, int silu, int out_compact, long ldx) {
    __shared__ double sh[kGT / 64];
    const int n = blockIdx.y, g = blockIdx.x, cpg = C / G, c0 = g * cpg;
    const int items = H * W * cpg;
    double a = 0.0, b = 0.0;
    for (int i = threadIdx.x; i < items; i += kGT) {
        const int pi = i / cpg, c = c0 + i - pi * cpg;
        const double v = (double)x[gn_prow(n, pi, H, W) * ldx + c];
        a += v; b += v * v;
    }
    a = block_sum_d(a, sh); b = block_sum_d(b, sh);
    const double m = a / items;
    double var = b / items - m * m;
    var = var > 0 ? var : 0;
    const float r = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)m;
    if (threadIdx.x == 0) { mean[(long)n * G + g] = mf; rstd[(long)n * G + g] = r; }
    for (int i = threadIdx.x; i < items; i += kGT) {
        const int pi = i / cpg, c = c0 + i - pi * cpg;
        const float sc = r * gamma[c], sf = beta[c] - mf * sc;
        const float z = x[gn_prow(n, pi, H, W) * ldx + c] * sc + sf;
        const long orow = out_compact ? (long)n * H * W + pi : gn_prow(n, pi, H, W);
        y[orow * C + c] = silu ? silu_f(z) : z;
    }
}

// The backward of groupnorm.hip, same arguments: cotangent samples n2 = k nx + n against saved sample n; dgamma / dbeta per set
// (set = n2 / set_images, set_stride floats apart); accum / accum2 added; channels [split_c, C) to dx2 (+= when accumulate2) when
// the normalised input was a concat; colsum += per-(cotangent sample, channel) sums of the GroupNorm term of dx.
__global__ __launch_bounds__(kGT) void gn_bwd_f32_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ mean, const float* __restrict__ rstd, float* __restrict__ dx, const float* __restrict__ accum,
    const float* __restrict__ accum2, float* __restrict__ dx2, int split_c, int accumulate2, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ colsum, long colsum_ld, int nsets_per_x, int nx, int set_images, long set_stride,
    int H, int W, int C, int G, int silu, int dy_compact, long ldx, int s2d) {
    __shared__ double sh[kGT / 64];
    const int n = blockIdx.y, g = blockIdx.x, cpg = C / G, c0 = g * cpg;
    const int items = H * W * cpg;
    const float mf = mean[(long)n * G + g], rs = rstd[(long)n * G + g];
    for (int k = 0; k < nsets_per_x; ++k) {
        const int n2 = k * nx + n;
        const long so = (long)(n2 / set_images) * set_stride;
        double s1 = 0.0, s2 = 0.0;
        for (int i = threadIdx.x; i < items; i += kGT) {
            const int pi = i / cpg, c = c0 + i - pi * cpg;
            const float xh = (x[gn_prow(n, pi, H, W) * ldx + c] - mf) * rs;
            const float d = dy[(dy_compact ? (long)n2 * H * W + pi : gn_prow(n2, pi, H, W)) * C + c];
            const float dsl = silu ? dsilu_f(xh * gamma[c] + beta[c]) : 1.f;
            const float dz = d * dsl * gamma[c];
            s1 += (double)dz; s2 += (double)dz * xh;
        }
        s1 = block_sum_d(s1, sh); s2 = block_sum_d(s2, sh);
        // per-channel dgamma / dbeta: thread t < cpg walks its channel's pixels (serial: this is the instrument, not the fast path)
        for (int cc = threadIdx.x; cc < cpg; cc += kGT) {
            const int c = c0 + cc;
            double dg = 0.0, db = 0.0;
            for (int pi = 0; pi < H * W; ++pi) {
                const float xh = (x[gn_prow(n, pi, H, W) * ldx + c] - mf) * rs;
                const float d = dy[(dy_compact ? (long)n2 * H * W + pi : gn_prow(n2, pi, H, W)) * C + c];
                const float t = d * (silu ? dsilu_f(xh * gamma[c] + beta[c]) : 1.f);
                dg += (double)t * xh; db += (double)t;
            }
            atomicAdd(dgamma + so + c, (float)dg);
            atomicAdd(dbeta + so + c, (float)db);
        }
        const float m1 = (float)(s1 / items), m2 = (float)(s2 / items);
        for (int i = threadIdx.x; i < items; i += kGT) {
            const int pi = i / cpg, c = c0 + i - pi * cpg;
            const float xh = (x[gn_prow(n, pi, H, W) * ldx + c] - mf) * rs;
            const float d = dy[(dy_compact ? (long)n2 * H * W + pi : gn_prow(n2, pi, H, W)) * C + c];
            const float dsl = silu ? dsilu_f(xh * gamma[c] + beta[c]) : 1.f;
            const float t = rs * (d * dsl * gamma[c] - m1 - xh * m2);
            const long prow = gn_prow(n2, pi, H, W);
            float o = t;
            if (accum) o += accum[prow * C + c];
            if (accum2) o += accum2[prow * C + c];
            if (dx2 && c >= split_c) {
                float* dst = dx2 + prow * (C - split_c) + (c - split_c);
                *dst = accumulate2 ? *dst + o : o;
            } else if (s2d) {
                // the first part of a split target in space-to-depth layout (groupnorm.hip S2D): pixel (y, x) -> pixel (y / 2, x / 2) of a
                // half-resolution padded tensor of 4 split_c channels, column plane * split_c + c, plane = 2 (y & 1) + (x & 1)
                const int y = pi / W, xx = pi - y * W;
                const long hrow = ((long)n2 * (H / 2 + 2) + y / 2 + 1) * (W / 2 + 2) + xx / 2 + 1;
                dx[hrow * (4 * split_c) + (2 * (y & 1) + (xx & 1)) * split_c + c] = o;
            } else {
                dx[prow * (dx2 ? split_c : C) + c] = o;
            }
            if (colsum) atomicAdd(colsum + (long)n2 * colsum_ld + c, t);
        }
        __syncthreads();
    }
}

// conv_out (Cout = image channels): pred[n][co][y][x] = bias[co] + sum_{tap, c} x[n, y + ky - 1, x + kx - 1, c] w[tap][co][c]
__global__ void conv_out_fprop_f32_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                          float* __restrict__ pred, int B, int H, int W, int C, int CO) {
    const long total = (long)B * CO * H * W;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = i % W; long t = i / W;
        const int yy = t % H; t /= H;
        const int co = t % CO; const int n = t / CO;
        float acc = bias[co];
        for (int tap = 0; tap < 9; ++tap) {
            const long row = ((long)n * (H + 2) + yy + tap / 3) * (W + 2) + xx + tap % 3;
            const float* xr = x + row * C;
            const float* wr = w + ((long)tap * CO + co) * C;
            for (int c = 0; c < C; ++c) acc = fmaf(xr[c], wr[c], acc);
        }
        pred[i] = acc;
    }
}

// ---------------------------------------------------------------- data movement on padded NHWC, one thread per element
// (the arithmetic of elementwise.hip's 16-B-chunk kernels; same argument lists)
#define FOR_ELEMS(total) \
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long)gridDim.x * blockDim.x)
__device__ __forceinline__ long prow(int n, int y, int x, int H, int W) { return ((long)n * (H + 2) + (y + 1)) * (W + 2) + (x + 1); }
struct Px { int n, y, x, c; };
__device__ __forceinline__ Px decode(long i, int H, int W, int C) {
    Px p; p.c = i % C; long r = i / C; p.x = r % W; r /= W; p.y = r % H; p.n = r / H; return p;
}
enum EwOp { EW_UP, EW_UP_BWD, EW_CAT, EW_CAT_TAIL, EW_CAT_BWD, EW_ADD, EW_S2D, EW_D2S, EW_P2C, EW_C2P };

template <int OP>
__global__ void ew_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, float* __restrict__ o2,
                              int flag, int N, int H, int W, int Ca, int Cb, long ld) {
    if (OP == EW_UP) {                       // o[n, y, x, :] = a[n, y / 2, x / 2, :]  (o is 2H x 2W)
        FOR_ELEMS((long)N * 4 * H * W * Ca) { const Px p = decode(i, 2 * H, 2 * W, Ca);
            o[prow(p.n, p.y, p.x, 2 * H, 2 * W) * Ca + p.c] = a[prow(p.n, p.y >> 1, p.x >> 1, H, W) * Ca + p.c]; }
    } else if (OP == EW_UP_BWD) {            // o[n, y, x, :] = sum of the 2 x 2 block of a
        FOR_ELEMS((long)N * H * W * Ca) { const Px p = decode(i, H, W, Ca);
            float t = 0.f;
            for (int d = 0; d < 4; ++d) t += a[prow(p.n, 2 * p.y + (d >> 1), 2 * p.x + (d & 1), 2 * H, 2 * W) * Ca + p.c];
            o[prow(p.n, p.y, p.x, H, W) * Ca + p.c] = t; }
    } else if (OP == EW_CAT) {               // o[..., :Ca] = a, o[..., Ca:] = b
        FOR_ELEMS((long)N * H * W * (Ca + Cb)) { const Px p = decode(i, H, W, Ca + Cb); const long r = prow(p.n, p.y, p.x, H, W);
            o[r * (Ca + Cb) + p.c] = p.c < Ca ? a[r * Ca + p.c] : b[r * Cb + p.c - Ca]; }
    } else if (OP == EW_CAT_TAIL) {          // o[..., Ca:] = b
        FOR_ELEMS((long)N * H * W * Cb) { const Px p = decode(i, H, W, Cb); const long r = prow(p.n, p.y, p.x, H, W);
            o[r * (Ca + Cb) + Ca + p.c] = b[r * Cb + p.c]; }
    } else if (OP == EW_CAT_BWD) {           // o = a[..., :Ca] ; o2 (+)= a[..., Ca:]
        FOR_ELEMS((long)N * H * W * (Ca + Cb)) { const Px p = decode(i, H, W, Ca + Cb); const long r = prow(p.n, p.y, p.x, H, W);
            const float v = a[r * (Ca + Cb) + p.c];
            if (p.c < Ca) o[r * Ca + p.c] = v;
            else { float* d = o2 + r * Cb + p.c - Ca; *d = flag ? *d + v : v; } }
    } else if (OP == EW_ADD) {               // o += b over the interior
        FOR_ELEMS((long)N * H * W * Ca) { const Px p = decode(i, H, W, Ca); const long e = prow(p.n, p.y, p.x, H, W) * Ca + p.c;
            o[e] += b[e]; }
    } else if (OP == EW_S2D) {               // o[n, y / 2, x / 2, plane * C + c] = a[n, y, x, c]  (a's row stride ld)
        FOR_ELEMS((long)N * H * W * Ca) { const Px p = decode(i, H, W, Ca); const int plane = (p.y & 1) * 2 + (p.x & 1);
            o[prow(p.n, p.y >> 1, p.x >> 1, H / 2, W / 2) * (4 * Ca) + plane * Ca + p.c] = a[prow(p.n, p.y, p.x, H, W) * ld + p.c]; }
    } else if (OP == EW_D2S) {               // the inverse, optionally accumulating
        FOR_ELEMS((long)N * H * W * Ca) { const Px p = decode(i, H, W, Ca); const int plane = (p.y & 1) * 2 + (p.x & 1);
            const float v = a[prow(p.n, p.y >> 1, p.x >> 1, H / 2, W / 2) * (4 * Ca) + plane * Ca + p.c];
            float* d = o + prow(p.n, p.y, p.x, H, W) * Ca + p.c; *d = flag ? *d + v : v; }
    } else if (OP == EW_P2C) {               // padded -> compact [N][H W][C]
        FOR_ELEMS((long)N * H * W * Ca) { const Px p = decode(i, H, W, Ca); o[i] = a[prow(p.n, p.y, p.x, H, W) * Ca + p.c]; }
    } else if (OP == EW_C2P) {               // padded = compact (+ padded residual b)
        FOR_ELEMS((long)N * H * W * Ca) { const Px p = decode(i, H, W, Ca); const long e = prow(p.n, p.y, p.x, H, W) * Ca + p.c;
            o[e] = a[i] + (b ? b[e] : 0.f); }
    }
}
inline int ew_grid(long total) { long b = (total + 255) / 256; return (int)(b < 1 ? 1 : (b > 8192 ? 8192 : b)); }

// in [B][R][C] -> out [B][C][R]
__global__ void transpose_f32_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int C) {
    const long bo = (long)blockIdx.y * R * C;
    FOR_ELEMS((long)R * C) { const int r = i / C, c = i - (long)r * C; out[bo + (long)c * R + r] = in[bo + i]; }
}
// elementwise.hip's im2col3x3 with f32 rows: NCHW f32 image -> [N][H+2][W+2][K] f32, k = tap * Cin + ci (zero beyond 9 Cin, zero halo
// rows); flip mirrors the tap offsets (the im2col of a cotangent image)
__global__ void im2col3x3_f32_kernel(const float* __restrict__ img, float* __restrict__ out, int N, int Cin, int H, int W, int K,
                                     int flip) {
    FOR_ELEMS((long)N * (H + 2) * (W + 2) * K) {
        const int k = i % K; long r = i / K;
        const int xp = r % (W + 2); long t = r / (W + 2);
        const int yp = t % (H + 2); const int n = t / (H + 2);
        const int tap = k / Cin, ci = k - tap * Cin;
        float val = 0.f;
        if (!(xp == 0 || yp == 0 || xp == W + 1 || yp == H + 1) && tap < 9) {
            const int dy = tap / 3 - 1, dx = tap % 3 - 1;
            const int y = yp - 1 + (flip ? -dy : dy), x = xp - 1 + (flip ? -dx : dx);
            if (y >= 0 && y < H && x >= 0 && x < W) val = img[(((long)n * Cin + ci) * H + y) * W + x];
        }
        out[i] = val;
    }
}
// row softmax forward / backward (attention.hip's, f32 rows of any length): one wave per row
__global__ __launch_bounds__(256) void softmax_fwd_f32_kernel(const float* __restrict__ s, float* __restrict__ p, long rows, int S) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    float mx = -INFINITY;
    for (int k = lane; k < S; k += 64) mx = fmaxf(mx, s[r * S + k]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int k = lane; k < S; k += 64) sum += expf(s[r * S + k] - mx);
    sum = wave_sum(sum);
    for (int k = lane; k < S; k += 64) p[r * S + k] = expf(s[r * S + k] - mx) / sum;
}
__global__ __launch_bounds__(256) void softmax_bwd_f32_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                              float* __restrict__ ds, long rows, long p_rows, int S, float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* pr = p + (r % p_rows) * S;
    float dot = 0.f;
    for (int k = lane; k < S; k += 64) dot += pr[k] * dp[r * S + k];
    dot = wave_sum(dot);
    for (int k = lane; k < S; k += 64) ds[r * S + k] = scale * pr[k] * (dp[r * S + k] - dot);
}
__global__ void copy_f32_kernel(const float* __restrict__ a, float* __restrict__ o, long n) { FOR_ELEMS(n) o[i] = a[i]; }
// [taps][co][ci] f32 master slices -> [taps][ci][co] f32 with the tap order reversed (the dgrad operand copies; job table as
// optimizer.hip's conv_weight_dgrad_multi: one block per 64 x 64 tile)
struct WtJobF { long src, dst; int taps, co, ci, tile0; };
__global__ void weight_dgrad_f32_kernel(const float* __restrict__ flat, float* __restrict__ wt_all, const WtJobF* __restrict__ jobs,
                                        int njobs) {
    int lo = 0, hi = njobs - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (jobs[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
    const WtJobF j = jobs[lo];
    int t = blockIdx.x - j.tile0;
    const int tc = (j.ci + 63) / 64, to = (j.co + 63) / 64;
    const int tap = t / (tc * to); t -= tap * tc * to;
    const int o0 = (t / tc) * 64, c0 = (t % tc) * 64;
    const float* src = flat + j.src + (long)tap * j.co * j.ci;
    float* dst = wt_all + j.dst + (long)(j.taps - 1 - tap) * j.co * j.ci;
    for (int e = threadIdx.x; e < 64 * 64; e += blockDim.x) {
        const int o = o0 + e / 64, c = c0 + e % 64;
        if (o < j.co && c < j.ci) dst[(long)c * j.co + o] = src[(long)o * j.ci + c];
    }
}

// ---------------------------------------------------------------- token space (transformer.hip's kernels with f32 rows [rows][C])
// LayerNorm over the last dim: one wave per row; mean / rstd saved
__global__ __launch_bounds__(256) void layernorm_fwd_f32_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float* __restrict__ y,
                                                                float* __restrict__ mean, float* __restrict__ rstd, long rows, int C,
                                                                float eps) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    double a = 0.0, b = 0.0;
    for (int c = lane; c < C; c += 64) { const double v = x[r * C + c]; a += v; b += v * v; }
    a = wave_sum_d(a); b = wave_sum_d(b);
    const double m = a / C;
    double var = b / C - m * m; var = var > 0 ? var : 0;
    const float rs = (float)(1.0 / sqrt(var + (double)eps)), mf = (float)m;
    if (lane == 0) { mean[r] = mf; rstd[r] = rs; }
    for (int c = lane; c < C; c += 64) y[r * C + c] = (x[r * C + c] - mf) * rs * gamma[c] + beta[c];
}
// dx[r] = (accum[r] +) rstd (dy gamma - mean_c(dy gamma) - xhat mean_c(dy gamma xhat)); saved row = r % rows_x
__global__ __launch_bounds__(256) void layernorm_bwd_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                                const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ accum,
                                                                float* __restrict__ dx, long rows2, long rows_x, int C) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows2) return;
    const long rx = r % rows_x;
    const float mf = mean[rx], rs = rstd[rx];
    double s1 = 0.0, s2 = 0.0;
    for (int c = lane; c < C; c += 64) {
        const float dg = dy[r * C + c] * gamma[c], xh = (x[rx * C + c] - mf) * rs;
        s1 += (double)dg; s2 += (double)dg * xh;
    }
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    const float m1 = (float)(s1 / C), m2 = (float)(s2 / C);
    for (int c = lane; c < C; c += 64) {
        const float dg = dy[r * C + c] * gamma[c], xh = (x[rx * C + c] - mf) * rs;
        dx[r * C + c] = (accum ? accum[r * C + c] : 0.f) + rs * (dg - m1 - xh * m2);
    }
}
// dgamma / dbeta[set][c] += column sums over the set's rows (one thread per (set, column))
__global__ void layernorm_dgb_f32_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
                                         const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                         long rows_x, long set_rows, long set_stride, int C) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x, set = blockIdx.y;
    if (c >= C) return;
    double dg = 0.0, db = 0.0;
    for (long i = 0; i < set_rows; ++i) {
        const long r = (long)set * set_rows + i, rx = r % rows_x;
        const float d = dy[r * C + c];
        dg += (double)d * ((x[rx * C + c] - mean[rx]) * rstd[rx]); db += (double)d;
    }
    dgamma[(long)set * set_stride + c] += (float)dg;
    dbeta[(long)set * set_stride + c] += (float)db;
}
__device__ __forceinline__ float gelu_phi(float g) { return 0.5f * (1.f + erff(g * 0.70710678118654752f)); }
// out[r][f] = h[r][f] gelu(h[r][F + f]);  dh[r][:F] = dout gelu(g), dh[r][F:] = dout a gelu'(g) (saved row = r % rows_x)
__global__ void geglu_fwd_f32_kernel(const float* __restrict__ h, float* __restrict__ out, long rows, int F) {
    FOR_ELEMS(rows * F) { const long r = i / F; const int f = i - r * F; const float g = h[r * 2 * F + F + f];
        out[i] = h[r * 2 * F + f] * g * gelu_phi(g); }
}
__global__ void geglu_bwd_f32_kernel(const float* __restrict__ dout, const float* __restrict__ h, float* __restrict__ dh, long rows2,
                                     long rows_x, int F) {
    FOR_ELEMS(rows2 * F) { const long r = i / F; const int f = i - r * F; const long rx = r % rows_x;
        const float a = h[rx * 2 * F + f], g = h[rx * 2 * F + F + f], d = dout[i];
        const float Phi = gelu_phi(g), phi = 0.3989422804014327f * expf(-0.5f * g * g);
        dh[r * 2 * F + f] = d * g * Phi;
        dh[r * 2 * F + F + f] = d * a * (Phi + g * phi); }
}
// dst[(b H + h)][s][d] = src[b][s][h D + d] for s < S, d < D, zero in the padding; and back
__global__ void head_split_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int S, int H, int D, int Sp, int Dp) {
    FOR_ELEMS((long)B * H * Sp * Dp) { const int d = i % Dp; long t = i / Dp; const int s_ = t % Sp; t /= Sp; const int h = t % H;
        const int b = t / H; dst[i] = (s_ < S && d < D) ? src[((long)b * S + s_) * H * D + h * D + d] : 0.f; }
}
__global__ void head_merge_f32_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int S, int H, int D, int Sp, int Dp) {
    FOR_ELEMS((long)B * S * H * D) { const int d = i % D; long t = i / D; const int h = t % H; t /= H; const int s_ = t % S;
        const int b = t / S; dst[i] = src[(((long)b * H + h) * Sp + s_) * Dp + d]; }
}
// p[r][k] = softmax over the visible k < valid of s[r][k] (causal_period > 0: also k <= r % causal_period), 0 up to ld; and its backward
__global__ __launch_bounds__(256) void softmax_rows_fwd_f32_kernel(const float* __restrict__ s, float* __restrict__ p, long rows,
                                                                   int valid, int ld, int causal_period) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    int lim = valid;
    if (causal_period > 0) { const int q = (int)(r % causal_period) + 1; lim = q < lim ? q : lim; }
    float mx = -INFINITY;
    for (int k = lane; k < lim; k += 64) mx = fmaxf(mx, s[r * ld + k]);
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    float sum = 0.f;
    for (int k = lane; k < lim; k += 64) sum += expf(s[r * ld + k] - mx);
    sum = wave_sum(sum);
    for (int k = lane; k < ld; k += 64) p[r * ld + k] = k < lim ? expf(s[r * ld + k] - mx) / sum : 0.f;
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_f32_kernel(const float* __restrict__ p, const float* __restrict__ dp,
                                                                   float* __restrict__ ds, long rows, long p_rows, int valid, int ld,
                                                                   float scale) {
    const int lane = threadIdx.x & 63;
    const long r = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* pr = p + (r % p_rows) * ld;
    float dot = 0.f;
    for (int k = lane; k < valid; k += 64) dot += pr[k] * dp[r * ld + k];
    dot = wave_sum(dot);
    for (int k = lane; k < ld; k += 64) ds[r * ld + k] = k < valid ? scale * pr[k] * (dp[r * ld + k] - dot) : 0.f;
}
// out[r] = <a[r], b[r % rows_b]> over D contiguous floats (one thread per row)
__global__ void rowdot_f32_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out, long rows,
                                  long rows_b, int D) {
    FOR_ELEMS(rows) { const float* x = a + i * D; const float* y = b + (i % rows_b) * D; float t = 0.f;
        for (int d = 0; d < D; ++d) t = fmaf(x[d], y[d], t);
        out[i] = t; }
}

}  // namespace

extern "C" {

// siss_gemm_nt with f32 tensors (A, W, C, R f32; lda / ldc / ldr / strides in elements), same argument list and epilogue.
int siss_gemm_nt_f32(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const float* rowbias, long ldrb,
                     const void* R, long ldr, int M, int N, int Kp, int npanels, const int* shifts, const int* coffs,
                     int rows_per_image, int Hp, int Wp, float alpha, int batch, long strideA, long strideW, long strideC,
                     void* stream) {
    SISS_CHECK_ARG(A && W && C && shifts && coffs && M > 0 && N > 0 && Kp > 0 && Kp % 16 == 0);
    SISS_CHECK_ARG(npanels >= 1 && npanels <= kMaxPanelsF && batch >= 1 && batch <= 65535 && rows_per_image > 0);
    SISS_CHECK_ARG(lda % 4 == 0 && ((uintptr_t)A | (uintptr_t)W) % 16 == 0 && (Hp == 0 || (long)Hp * Wp == rows_per_image));
    NTF p;
    p.A = (const float*)A; p.W = (const float*)W; p.C = (float*)C; p.bias = bias; p.rowbias = rowbias; p.R = (const float*)R;
    p.rowsub = nullptr; p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb; p.strideA = strideA; p.strideW = strideW;
    p.strideC = strideC; p.M = M; p.N = N; p.Kp = Kp; p.npanels = npanels; p.rows_per_image = rows_per_image; p.Hp = Hp; p.Wp = Wp;
    p.mul_r = 0; p.d2s = 0; p.alpha_cols = 0; p.alpha = alpha;
    for (int i = 0; i < kMaxPanelsF; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    for (int i = 0; i < npanels; ++i) SISS_CHECK_ARG(p.coff[i] % 4 == 0);
    SISS_CHECK_ARG(cdiv(N, 64) <= 65535);
    gemm_nt_f32_kernel<<<dim3(cdiv(M, 16), cdiv(N, 64), batch), 256, 0, (hipStream_t)stream>>>(p);
    SISS_LAUNCH_RET();
}

// siss_gemm_tn with f32 operands (Y, X f32; dW f32 as ever).  nsplits is accepted for the signature's sake: -1 = overwrite, anything
// else = accumulate into dW (one wave owns a tile: plain read-add-write, deterministic).  zero_page is not used.
int siss_gemm_tn_f32(const void* Y, long ldy, const void* X, long ldx, float* dW, long set_stride, int N, int C, int npanels,
                     const int* shifts, const int* coffs, int nsets, int rows_per_set, long x_set_rows, int row_begin, int row_end,
                     int nsplits, const void* zero_page, float* dbias, float* dbias2, void* stream) {
    (void)zero_page;
    SISS_CHECK_ARG(Y && X && dW && shifts && coffs && N > 0 && C > 0 && npanels >= 1 && npanels <= kMaxPanelsF && nsets >= 1);
    SISS_CHECK_ARG(row_begin >= 0 && row_end > row_begin && row_end <= rows_per_set && (long)npanels * nsets <= 65535);
    TNF p;
    p.Y = (const float*)Y; p.X = (const float*)X; p.dW = dW; p.dbias = dbias; p.dbias2 = dbias ? dbias2 : nullptr;
    p.ldy = ldy; p.ldx = ldx; p.set_stride = set_stride; p.bias_stride = set_stride; p.x_set_rows = x_set_rows; p.N = N; p.C = C; p.npanels = npanels;
    p.nsets = nsets; p.rows_per_set = rows_per_set; p.row_begin = row_begin; p.row_end = row_end; p.overwrite = nsplits == -1;
    for (int i = 0; i < kMaxPanelsF; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    hipStream_t st = (hipStream_t)stream;
    gemm_tn_f32_kernel<<<dim3(cdiv(C, 64), cdiv(N, 16), npanels * nsets), 256, 0, st>>>(p);
    if (dbias) tn_bias_f32_kernel<<<dim3(cdiv(N, 64), nsets), 64, 0, st>>>(p);
    SISS_LAUNCH_RET();
}

// siss_groupnorm_fwd_ld with f32 tensors (`partial` is accepted and not used: a block owns its (sample, group)).
int siss_groupnorm_fwd_ld_f32(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                              float* partial, int N, int H, int W, int C, int G, float eps, int silu, int out_compact, int ldx,
                              void* stream) {
    (void)partial;
    SISS_CHECK_ARG(x && gamma && beta && y && mean && rstd && N > 0 && H > 0 && W > 0 && C > 0 && G > 0 && C % G == 0 && N <= 65535);
    SISS_CHECK_ARG(ldx == 0 || ldx >= C);
    gn_fwd_f32_kernel<<<dim3(G, N), kGT, 0, (hipStream_t)stream>>>((const float*)x, gamma, beta, (float*)y, mean, rstd, H, W, C, G, eps,
                                                                silu, out_compact, ldx ? ldx : C);
    SISS_LAUNCH_RET();
}

// siss_groupnorm_bwd_ld with f32 tensors.
int siss_groupnorm_bwd_ld_f32(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                              const float* rstd, void* dx, const void* accum, const void* accum2, void* dx2, int split_c,
                              int accumulate2, float* dgamma, float* dbeta, float* colsum, long colsum_ld, float* partial, int n2,
                              int nx, int set_images, long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx,
                              void* stream) {
    (void)partial;
    SISS_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dgamma && dbeta);
    SISS_CHECK_ARG(n2 > 0 && nx > 0 && set_images > 0 && n2 % set_images == 0 && n2 % nx == 0 && C % G == 0 && nx <= 65535);
    SISS_CHECK_ARG(!dx2 || (split_c > 0 && split_c < C));
    SISS_CHECK_ARG(ldx == 0 || ldx >= C);
    gn_bwd_f32_kernel<<<dim3(G, nx), kGT, 0, (hipStream_t)stream>>>(
        (const float*)dy, (const float*)x, gamma, beta, mean, rstd, (float*)dx, (const float*)accum, (const float*)accum2, (float*)dx2,
        split_c, accumulate2, dgamma, dbeta, colsum, colsum_ld, n2 / nx, nx, set_images, set_stride, H, W, C, G, silu, dy_compact,
        ldx ? ldx : C, 0);
    SISS_LAUNCH_RET();
}

// siss_conv_out_fprop with an f32 activation.
int siss_conv_out_fprop_f32(const void* x, const float* w, const float* bias, float* pred, int B, int H, int W, int C, int CO,
                            void* stream) {
    SISS_CHECK_ARG(x && w && bias && pred && B > 0 && H > 0 && W > 0 && C > 0 && CO > 0);
    const long total = (long)B * CO * H * W;
    long nb = (total + 255) / 256;
    if (nb > 8192) nb = 8192;
    conv_out_fprop_f32_kernel<<<(int)nb, 256, 0, (hipStream_t)stream>>>((const float*)x, w, bias, pred, B, H, W, C, CO);
    SISS_LAUNCH_RET();
}


// ---- the data-movement launchers of elementwise.hip / attention.hip / optimizer.hip with f32 tensors (same argument lists) ----
#define EW(OP, total, a, b, o, o2, flag, N, H, W, Ca, Cb, ld)                                                                  \
    ew_f32_kernel<OP><<<ew_grid(total), 256, 0, (hipStream_t)stream>>>((const float*)(a), (const float*)(b), (float*)(o), (float*)(o2), \
                                                                      flag, N, H, W, Ca, Cb, ld);                               \
    SISS_LAUNCH_RET()
#define EW_OK(N, H, W, C) ((N) > 0 && (H) > 0 && (W) > 0 && (C) > 0)
int siss_upsample2x_f32(const void* in, void* out, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(in && out && EW_OK(N, H, W, C)); EW(EW_UP, (long)N * 4 * H * W * C, in, nullptr, out, nullptr, 0, N, H, W, C, 0, C);
}
int siss_upsample2x_bwd_f32(const void* dout, void* din, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(dout && din && EW_OK(N, H, W, C)); EW(EW_UP_BWD, (long)N * H * W * C, dout, nullptr, din, nullptr, 0, N, H, W, C, 0, C);
}
int siss_concat_f32(const void* a, const void* b, void* out, int N, int H, int W, int Ca, int Cb, void* stream) {
    SISS_CHECK_ARG(a && b && out && EW_OK(N, H, W, Ca) && Cb > 0); EW(EW_CAT, (long)N * H * W * (Ca + Cb), a, b, out, nullptr, 0, N, H, W, Ca, Cb, 0);
}
int siss_concat_tail_f32(const void* b, void* out, int N, int H, int W, int Ca, int Cb, void* stream) {
    SISS_CHECK_ARG(b && out && EW_OK(N, H, W, Ca) && Cb > 0); EW(EW_CAT_TAIL, (long)N * H * W * Cb, nullptr, b, out, nullptr, 0, N, H, W, Ca, Cb, 0);
}
int siss_concat_bwd_f32(const void* dcat, void* da, void* db, int accumulate_b, int N, int H, int W, int Ca, int Cb, void* stream) {
    SISS_CHECK_ARG(dcat && da && db && EW_OK(N, H, W, Ca) && Cb > 0);
    EW(EW_CAT_BWD, (long)N * H * W * (Ca + Cb), dcat, nullptr, da, db, accumulate_b, N, H, W, Ca, Cb, 0);
}
int siss_add_inplace_f32(void* a, const void* b, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(a && b && EW_OK(N, H, W, C)); EW(EW_ADD, (long)N * H * W * C, nullptr, b, a, nullptr, 0, N, H, W, C, 0, C);
}
int siss_space_to_depth_ld_f32(const void* in, void* z, int N, int H, int W, int C, int ld_in, void* stream) {
    SISS_CHECK_ARG(in && z && EW_OK(N, H, W, C) && H % 2 == 0 && W % 2 == 0 && (ld_in == 0 || ld_in >= C));
    EW(EW_S2D, (long)N * H * W * C, in, nullptr, z, nullptr, 0, N, H, W, C, 0, ld_in ? ld_in : C);
}
int siss_depth_to_space_f32(const void* dz, void* din, int accumulate, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(dz && din && EW_OK(N, H, W, C) && H % 2 == 0 && W % 2 == 0);
    EW(EW_D2S, (long)N * H * W * C, dz, nullptr, din, nullptr, accumulate, N, H, W, C, 0, C);
}
int siss_pad_to_compact_f32(const void* in, void* out, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(in && out && EW_OK(N, H, W, C)); EW(EW_P2C, (long)N * H * W * C, in, nullptr, out, nullptr, 0, N, H, W, C, 0, C);
}
int siss_compact_add_to_pad_f32(const void* comp, const void* res, void* out, int N, int H, int W, int C, void* stream) {
    SISS_CHECK_ARG(comp && out && EW_OK(N, H, W, C)); EW(EW_C2P, (long)N * H * W * C, comp, res, out, nullptr, 0, N, H, W, C, 0, C);
}
int siss_transpose_f32(const void* in, void* out, int batch, int R, int C, void* stream) {
    SISS_CHECK_ARG(in && out && batch > 0 && R > 0 && C > 0 && batch <= 65535);
    transpose_f32_kernel<<<dim3(ew_grid((long)R * C), batch), 256, 0, (hipStream_t)stream>>>((const float*)in, (float*)out, R, C);
    SISS_LAUNCH_RET();
}
// (img_bf16 must be 0: the f32 mode's images are f32)
int siss_im2col3x3_f32(const void* img, int img_bf16, void* out, int N, int Cin, int H, int W, int K, int flip, void* stream) {
    SISS_CHECK_ARG(img && out && !img_bf16 && N > 0 && Cin > 0 && H > 0 && W > 0 && K >= 9 * Cin);
    im2col3x3_f32_kernel<<<ew_grid((long)N * (H + 2) * (W + 2) * K), 256, 0, (hipStream_t)stream>>>((const float*)img, (float*)out, N, Cin, H, W, K, flip);
    SISS_LAUNCH_RET();
}
int siss_softmax_fwd_f32(const void* s, void* p, long rows, int S, void* stream) {
    SISS_CHECK_ARG(s && p && rows > 0 && S > 0);
    softmax_fwd_f32_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>((const float*)s, (float*)p, rows, S);
    SISS_LAUNCH_RET();
}
int siss_softmax_bwd_f32(const void* p, const void* dp, void* ds, long rows, long p_rows, int S, float scale, void* stream) {
    SISS_CHECK_ARG(p && dp && ds && rows > 0 && p_rows > 0 && S > 0);
    softmax_bwd_f32_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>((const float*)p, (const float*)dp, (float*)ds, rows, p_rows, S, scale);
    SISS_LAUNCH_RET();
}
// siss_cast_f32_bf16's place in the f32 mode: a copy (the attention block's f32 dK / dV partials -> their f32 cotangent slots)
int siss_copy_f32(const float* src, void* dst, long n, void* stream) {
    SISS_CHECK_ARG(src && dst && n > 0);
    copy_f32_kernel<<<ew_grid(n), 256, 0, (hipStream_t)stream>>>(src, (float*)dst, n);
    SISS_LAUNCH_RET();
}
// siss_conv_weight_dgrad_multi with f32 copies (jobs: the same device table; dst offsets in ELEMENTS of wt_all)
int siss_conv_weight_dgrad_multi_f32(const float* flat, void* wt_all, const void* jobs, int njobs, int total_tiles, void* stream) {
    SISS_CHECK_ARG(flat && wt_all && jobs && njobs > 0 && total_tiles > 0);
    weight_dgrad_f32_kernel<<<total_tiles, 256, 0, (hipStream_t)stream>>>(flat, (float*)wt_all, (const WtJobF*)jobs, njobs);
    SISS_LAUNCH_RET();
}


// ---- transformer.hip's launchers with f32 rows (the SD UNet's token space in the f32 parity mode: materialised attention path) ----
int siss_layernorm_fwd_f32(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd, long rows, int C,
                           float eps, void* stream) {
    SISS_CHECK_ARG(x && gamma && beta && y && mean && rstd && rows > 0 && C > 0);
    layernorm_fwd_f32_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>((const float*)x, gamma, beta, (float*)y, mean, rstd, rows, C, eps);
    SISS_LAUNCH_RET();
}
int siss_layernorm_bwd_f32(const void* dy, const void* x, const float* gamma, const float* mean, const float* rstd, const void* accum,
                           void* dx, float* dgamma, float* dbeta, long rows2, long rows_x, long set_rows, long set_stride, int C,
                           void* stream) {
    SISS_CHECK_ARG(dy && x && gamma && mean && rstd && dx && rows2 > 0 && rows_x > 0 && set_rows > 0 && C > 0);
    SISS_CHECK_ARG(rows2 % set_rows == 0 && (!dgamma || dbeta) && rows2 / set_rows <= 65535);
    hipStream_t st = (hipStream_t)stream;
    // (column sums first: dx may alias accum, never dy or x)
    if (dgamma)
        layernorm_dgb_f32_kernel<<<dim3(cdiv(C, 64), (unsigned)(rows2 / set_rows)), 64, 0, st>>>((const float*)dy, (const float*)x, mean, rstd,
                                                                                                dgamma, dbeta, rows_x, set_rows, set_stride, C);
    layernorm_bwd_f32_kernel<<<cdiv(rows2, 4), 256, 0, st>>>((const float*)dy, (const float*)x, gamma, mean, rstd, (const float*)accum,
                                                             (float*)dx, rows2, rows_x, C);
    SISS_LAUNCH_RET();
}
int siss_geglu_fwd_f32(const void* h, void* out, long rows, int F, void* stream) {
    SISS_CHECK_ARG(h && out && rows > 0 && F > 0);
    geglu_fwd_f32_kernel<<<ew_grid(rows * F), 256, 0, (hipStream_t)stream>>>((const float*)h, (float*)out, rows, F);
    SISS_LAUNCH_RET();
}
int siss_geglu_bwd_f32(const void* dout, const void* h, void* dh, long rows2, long rows_x, int F, void* stream) {
    SISS_CHECK_ARG(dout && h && dh && rows2 > 0 && rows_x > 0 && F > 0);
    geglu_bwd_f32_kernel<<<ew_grid(rows2 * F), 256, 0, (hipStream_t)stream>>>((const float*)dout, (const float*)h, (float*)dh, rows2, rows_x, F);
    SISS_LAUNCH_RET();
}
int siss_head_split_f32(const void* src, void* dst, int B, int S, int H, int D, int Sp, int Dp, void* stream) {
    SISS_CHECK_ARG(src && dst && B > 0 && S > 0 && H > 0 && D > 0 && Sp >= S && Dp >= D);
    head_split_f32_kernel<<<ew_grid((long)B * H * Sp * Dp), 256, 0, (hipStream_t)stream>>>((const float*)src, (float*)dst, B, S, H, D, Sp, Dp);
    SISS_LAUNCH_RET();
}
int siss_head_merge_f32(const void* src, void* dst, int B, int S, int H, int D, int Sp, int Dp, void* stream) {
    SISS_CHECK_ARG(src && dst && B > 0 && S > 0 && H > 0 && D > 0 && Sp >= S && Dp >= D);
    head_merge_f32_kernel<<<ew_grid((long)B * S * H * D), 256, 0, (hipStream_t)stream>>>((const float*)src, (float*)dst, B, S, H, D, Sp, Dp);
    SISS_LAUNCH_RET();
}
int siss_softmax_rows_fwd_f32(const void* s, void* p, long rows, int valid, int ld, int causal_period, void* stream) {
    SISS_CHECK_ARG(s && p && rows > 0 && valid > 0 && valid <= ld && causal_period >= 0);
    softmax_rows_fwd_f32_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>((const float*)s, (float*)p, rows, valid, ld, causal_period);
    SISS_LAUNCH_RET();
}
int siss_softmax_rows_bwd_f32(const void* p, const void* dp, void* ds, long rows, long p_rows, int valid, int ld, float scale,
                              void* stream) {
    SISS_CHECK_ARG(p && dp && ds && rows > 0 && p_rows > 0 && valid > 0 && valid <= ld);
    softmax_rows_bwd_f32_kernel<<<cdiv(rows, 4), 256, 0, (hipStream_t)stream>>>((const float*)p, (const float*)dp, (float*)ds, rows, p_rows, valid, ld, scale);
    SISS_LAUNCH_RET();
}
int siss_rowdot_f32(const void* a, const void* b, float* out, long rows, long rows_b, int D, void* stream) {
    SISS_CHECK_ARG(a && b && out && rows > 0 && rows_b > 0 && D > 0);
    rowdot_f32_kernel<<<ew_grid(rows), 256, 0, (hipStream_t)stream>>>((const float*)a, (const float*)b, out, rows, rows_b, D);
    SISS_LAUNCH_RET();
}
// siss_gemm_nt_mulsub with f32 tensors: C = R o (alpha (acc - rowsub[row])) (the attention backward's dS from one product)
int siss_gemm_nt_mulsub_f32(const void* A, long lda, const void* W, void* C, long ldc, const void* R, long ldr, const float* rowsub,
                            int M, int N, int Kp, float alpha, int batch, long strideA, long strideW, long strideC, void* stream) {
    SISS_CHECK_ARG(A && W && C && R && rowsub && M > 0 && N > 0 && Kp > 0 && Kp % 16 == 0 && batch >= 1 && batch <= 65535);
    SISS_CHECK_ARG(lda % 4 == 0 && ((uintptr_t)A | (uintptr_t)W) % 16 == 0 && cdiv(N, 64) <= 65535);
    NTF p;
    p.A = (const float*)A; p.W = (const float*)W; p.C = (float*)C; p.bias = nullptr; p.rowbias = nullptr; p.R = (const float*)R;
    p.rowsub = rowsub; p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldrb = 0; p.strideA = strideA; p.strideW = strideW; p.strideC = strideC;
    p.M = M; p.N = N; p.Kp = Kp; p.npanels = 1; p.rows_per_image = 1; p.Hp = 0; p.Wp = 0; p.mul_r = 1; p.d2s = 0; p.alpha_cols = 0;
    p.alpha = alpha;
    for (int i = 0; i < kMaxPanelsF; ++i) { p.shift[i] = 0; p.coff[i] = 0; }
    gemm_nt_f32_kernel<<<dim3(cdiv(M, 16), cdiv(N, 64), batch), 256, 0, (hipStream_t)stream>>>(p);
    SISS_LAUNCH_RET();
}


// ---------------------------------------------------------------------------------------------------------------------------------
// f32 forms of the SCHEDULE SWITCHES of the bf16 engine (round 5): folded 1x1 shortcut (forward and dgrad), depth-to-space epilogue,
// sub-pixel upsample pieces, grouped weight gradients.  Same argument lists as the bf16 launchers; each is the same arithmetic on
// the simple f32 kernels above (a fold = one accumulation of two products = the second launch adds onto the first's result), so that
// UNetEngine(dtype=float32, f32_fused=True) runs the schedule bench.py runs and is held against the fp32 oracle at 1e-4.
static int nt_f32(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const float* rowbias, long ldrb,
                  const void* R, long ldr, int M, int N, int Kp, int npanels, const int* shifts, const int* coffs, int rows_per_image,
                  int Hp, int Wp, int d2s, void* stream) {
    SISS_CHECK_ARG(A && W && C && shifts && coffs && M > 0 && N > 0 && Kp > 0 && Kp % 16 == 0);
    SISS_CHECK_ARG(npanels >= 1 && npanels <= kMaxPanelsF && rows_per_image > 0);
    SISS_CHECK_ARG(lda % 4 == 0 && ((uintptr_t)A | (uintptr_t)W) % 16 == 0 && (Hp == 0 || (long)Hp * Wp == rows_per_image));
    NTF p;
    p.A = (const float*)A; p.W = (const float*)W; p.C = (float*)C; p.bias = bias; p.rowbias = rowbias; p.R = (const float*)R;
    p.rowsub = nullptr; p.lda = lda; p.ldc = ldc; p.ldr = ldr; p.ldrb = ldrb; p.strideA = 0; p.strideW = 0; p.strideC = 0;
    p.M = M; p.N = N; p.Kp = Kp; p.npanels = npanels; p.rows_per_image = rows_per_image; p.Hp = Hp; p.Wp = Wp;
    p.mul_r = 0; p.d2s = d2s; p.alpha_cols = 0; p.alpha = 1.f;
    for (int i = 0; i < kMaxPanelsF; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    for (int i = 0; i < npanels; ++i) SISS_CHECK_ARG(p.coff[i] % 4 == 0);
    SISS_CHECK_ARG(cdiv(N, 64) <= 65535);
    gemm_nt_f32_kernel<<<dim3(cdiv(M, 16), cdiv(N, 64), 1), 256, 0, (hipStream_t)stream>>>(p);
    SISS_LAUNCH_RET();
}

// siss_gemm_nt_d2s / siss_gemm_nt_d2s_bias with f32 tensors
int siss_gemm_nt_d2s_f32(const void* A, long lda, const void* W, void* C, long ldc, const void* R, long ldr, int M, int N, int Kp,
                         int npanels, const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp, int plane, void* stream) {
    SISS_CHECK_ARG(plane >= 0 && plane < 4 && Hp > 2 && Wp > 2);
    return nt_f32(A, lda, W, C, ldc, nullptr, nullptr, 0, R, ldr, M, N, Kp, npanels, shifts, coffs, rows_per_image, Hp, Wp, 1 + plane, stream);
}
int siss_gemm_nt_d2s_bias_f32(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, int M, int N, int Kp,
                              int npanels, const int* shifts, const int* coffs, int rows_per_image, int Hp, int Wp, int plane,
                              void* stream) {
    SISS_CHECK_ARG(plane >= 0 && plane < 4 && Hp > 2 && Wp > 2);
    return nt_f32(A, lda, W, C, ldc, bias, nullptr, 0, nullptr, 0, M, N, Kp, npanels, shifts, coffs, rows_per_image, Hp, Wp, 1 + plane, stream);
}

// siss_gemm_nt_d2s_phases with f32 tensors (round 6: the one-launch form of the four planes / phases is part of the schedule the f32
// instrument covers): the four single-plane products one after the other -- the same arithmetic per element, W's planes back to back.
int siss_gemm_nt_d2s_phases_f32(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const void* R, long ldr,
                                int M, int N, int Kp, const int* phase_p0, const int* shifts, const int* coffs, int rows_per_image,
                                int Hp, int Wp, void* stream) {
    SISS_CHECK_ARG(phase_p0 && phase_p0[0] == 0 && shifts && coffs && Hp > 2 && Wp > 2 && W);
    for (int z = 0; z < 4; ++z) {
        const int np = phase_p0[z + 1] - phase_p0[z];
        SISS_CHECK_ARG(np >= 1);
        const int rc = nt_f32(A, lda, (const float*)W + (long)phase_p0[z] * N * Kp, C, ldc, bias, nullptr, 0, R, ldr, M, N, Kp, np,
                              shifts + phase_p0[z], coffs + phase_p0[z], rows_per_image, Hp, Wp, 1 + z, stream);
        if (rc != SISS_OK) return rc;
    }
    return SISS_OK;
}

// siss_conv3x3_sc with f32 tensors: C = conv1x1(A2; W2) + bias2 first, then C = conv3x3(A; W) + bias + rowbias + C.  No statistics
// (`written` reports 0: the consuming GroupNorm makes its own pass).
int siss_conv3x3_sc_f32(const void* A, long lda, const void* W, void* C, long ldc, const float* bias, const float* rowbias, long ldrb,
                        const void* A2, long lda2, const void* W2, int K2, const float* bias2, int M, int N, int Kp, const int* shifts,
                        const int* coffs, int rows_per_image, int Hp, int Wp, float* qstats, int* written, void* stream) {
    (void)qstats;
    if (written) *written = 0;
    SISS_CHECK_ARG(A2 && W2 && K2 > 0);
    const int z = 0;
    int rc = nt_f32(A2, lda2, W2, C, ldc, bias2, nullptr, 0, nullptr, 0, M, N, K2, 1, &z, &z, rows_per_image, Hp, Wp, 0, stream);
    if (rc != SISS_OK) return rc;
    return nt_f32(A, lda, W, C, ldc, bias, rowbias, ldrb, C, ldc, M, N, Kp, 9, shifts, coffs, rows_per_image, Hp, Wp, 0, stream);
}
// siss_conv3x3_dgrad_sc with f32 tensors: C = conv3x3^T(A; W) (+ R) and Cx = conv1x1^T(A; Wx), two launches over the same cotangent
int siss_conv3x3_dgrad_sc_f32(const void* A, long lda, const void* W, void* C, long ldc, const void* R, long ldr, const void* Wx,
                              void* Cx, long ldcx, int Nx, int M, int N, int Kp, const int* shifts, const int* coffs, int rows_per_image,
                              int Hp, int Wp, void* stream) {
    SISS_CHECK_ARG(Wx && Cx && Nx > 0);
    int rc = nt_f32(A, lda, W, C, ldc, nullptr, nullptr, 0, R, ldr, M, N, Kp, 9, shifts, coffs, rows_per_image, Hp, Wp, 0, stream);
    if (rc != SISS_OK) return rc;
    const int z = 0;
    return nt_f32(A, lda, Wx, Cx, ldcx, nullptr, nullptr, 0, nullptr, 0, M, Nx, Kp, 1, &z, &z, rows_per_image, Hp, Wp, 0, stream);
}

// siss_gemm_tn_bs with f32 operands (bias gradients with a set stride of their own)
int siss_gemm_tn_bs_f32(const void* Y, long ldy, const void* X, long ldx, float* dW, long set_stride, int N, int C, int npanels,
                        const int* shifts, const int* coffs, int nsets, int rows_per_set, long x_set_rows, int row_begin, int row_end,
                        int nsplits, const void* zero_page, float* dbias, float* dbias2, long bias_set_stride, void* stream) {
    (void)zero_page;
    SISS_CHECK_ARG(Y && X && dW && shifts && coffs && N > 0 && C > 0 && npanels >= 1 && npanels <= kMaxPanelsF && nsets >= 1);
    SISS_CHECK_ARG(row_begin >= 0 && row_end > row_begin && row_end <= rows_per_set && (long)npanels * nsets <= 65535);
    TNF p;
    p.Y = (const float*)Y; p.X = (const float*)X; p.dW = dW; p.dbias = dbias; p.dbias2 = dbias ? dbias2 : nullptr;
    p.ldy = ldy; p.ldx = ldx; p.set_stride = set_stride; p.bias_stride = bias_set_stride; p.x_set_rows = x_set_rows; p.N = N; p.C = C;
    p.npanels = npanels; p.nsets = nsets; p.rows_per_set = rows_per_set; p.row_begin = row_begin; p.row_end = row_end;
    p.overwrite = nsplits == -1;
    for (int i = 0; i < kMaxPanelsF; ++i) { p.shift[i] = i < npanels ? shifts[i] : 0; p.coff[i] = i < npanels ? coffs[i] : 0; }
    hipStream_t st = (hipStream_t)stream;
    gemm_tn_f32_kernel<<<dim3(cdiv(C, 64), cdiv(N, 16), npanels * nsets), 256, 0, st>>>(p);
    if (dbias) tn_bias_f32_kernel<<<dim3(cdiv(N, 64), nsets), 64, 0, st>>>(p);
    SISS_LAUNCH_RET();
}

// siss_gemm_tn_grouped with f32 operands: the job table walked on the host, one siss_gemm_tn_f32 per job (what the engine's wgrad
// QUEUE -- deferred launches, held operands -- needs in the f32 mode; there is nothing to group for an instrument)
struct siss_tn_job_f32 {
    const void* Y; long ldy; const void* X; long ldx; float* dW; long set_stride;
    int N, C, npanels, nsets, rows_per_set, row_begin, row_end, nsplits;
    long x_set_rows;
    const void* zero_page; float* dbias; float* dbias2;
    int shifts[9]; int coffs[9];
    long bias_set_stride;
};
int siss_gemm_tn_grouped_f32(const void* jobs, int njobs, void* stream) {
    SISS_CHECK_ARG(jobs && njobs > 0 && njobs <= 256);
    const siss_tn_job_f32* js = (const siss_tn_job_f32*)jobs;
    for (int i = 0; i < njobs; ++i) {
        const siss_tn_job_f32& j = js[i];
        const int rc = siss_gemm_tn_bs_f32(j.Y, j.ldy, j.X, j.ldx, j.dW, j.set_stride, j.N, j.C, j.npanels, j.shifts, j.coffs, j.nsets,
                                           j.rows_per_set, j.x_set_rows, j.row_begin, j.row_end, j.nsplits, j.zero_page, j.dbias, j.dbias2,
                                           j.bias_set_stride ? j.bias_set_stride : j.set_stride, stream);
        if (rc != SISS_OK) return rc;
    }
    return SISS_OK;
}

// siss_groupnorm_bwd_ld_s2d with f32 tensors
int siss_groupnorm_bwd_ld_s2d_f32(const void* dy, const void* x, const float* gamma, const float* beta, const float* mean,
                                  const float* rstd, void* dx, const void* accum, const void* accum2, void* dx2, int split_c,
                                  int accumulate2, float* dgamma, float* dbeta, float* colsum, long colsum_ld, float* partial, int n2,
                                  int nx, int set_images, long set_stride, int H, int W, int C, int G, int silu, int dy_compact, int ldx,
                                  void* stream) {
    (void)partial;
    SISS_CHECK_ARG(dy && x && gamma && beta && mean && rstd && dx && dx2 && dgamma && dbeta);
    SISS_CHECK_ARG(n2 > 0 && nx > 0 && set_images > 0 && n2 % set_images == 0 && n2 % nx == 0 && C % G == 0 && nx <= 65535);
    SISS_CHECK_ARG(split_c > 0 && split_c < C && H % 2 == 0 && W % 2 == 0 && (ldx == 0 || ldx >= C));
    gn_bwd_f32_kernel<<<dim3(G, nx), kGT, 0, (hipStream_t)stream>>>(
        (const float*)dy, (const float*)x, gamma, beta, mean, rstd, (float*)dx, (const float*)accum, (const float*)accum2, (float*)dx2,
        split_c, accumulate2, dgamma, dbeta, colsum, colsum_ld, n2 / nx, nx, set_images, set_stride, H, W, C, G, silu, dy_compact,
        ldx ? ldx : C, 1);
    SISS_LAUNCH_RET();
}

}  // extern "C"
