// PROBE (not part of the product build): Winograd F(2x2, 3x3) for ONE site -- 128 -> 128 channels at 256 x 256,
// forward, B = 16 -- in its UNFUSED form: input transform (this file) -> 16 batched products on the product library's NT GEMM
// (v_mfma_f32_16x16x32_bf16) -> output transform (this file).  2.25x fewer MACs than the direct convolution; what it costs in bytes
// is what the probe measures (tools/probes/winograd_probe.py).  bf16 storage, f32 transform arithmetic, one rounding per stored value.
//   V = B^T d B,  U = G g G^T (host, f32, rounded once),  M_f = V_f U_f^T,  Y = A^T M A
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   A^T = [1 1 1 0; 0 1 -1 -1]
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

__device__ __forceinline__ void unpack8(const u32x4_t& v, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) { f[2 * e] = __builtin_bit_cast(float, v[e] << 16); f[2 * e + 1] = __builtin_bit_cast(float, v[e] & 0xffff0000u); }
}
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f2{lo, hi}, b2));
}
__device__ __forceinline__ u32x4_t pack8(const float (&f)[8]) {
    return u32x4_t{pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7])};
}

// x: padded NHWC [N][H + 2][W + 2][C] (zero halo); V: [16][T][C], T = N (H / 2) (W / 2) tiles.  One thread per (tile, 8 channels).
__global__ void wino_input_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ V, int N, int H, int W, int C) {
    const int c8 = C / 8;
    const long T = (long)N * (H / 2) * (W / 2);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * c8) return;
    const int cc = i % c8; const long t = i / c8;
    const int tx = t % (W / 2); const long r = t / (W / 2); const int ty = r % (H / 2); const int n = r / (H / 2);
    const bf16_t* base = x + (((long)n * (H + 2) + 2 * ty) * (W + 2) + 2 * tx) * C + cc * 8;
    float d[4][4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) unpack8(*reinterpret_cast<const u32x4_t*>(base + ((long)a * (W + 2) + b) * C), d[a][b]);
    float v[4][4][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float tm[4][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {           // B^T d  (columns)
            tm[0][b] = d[0][b][e] - d[2][b][e]; tm[1][b] = d[1][b][e] + d[2][b][e];
            tm[2][b] = d[2][b][e] - d[1][b][e]; tm[3][b] = d[1][b][e] - d[3][b][e];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {           // ... B  (rows)
            v[a][0][e] = tm[a][0] - tm[a][2]; v[a][1][e] = tm[a][1] + tm[a][2];
            v[a][2][e] = tm[a][2] - tm[a][1]; v[a][3][e] = tm[a][1] - tm[a][3];
        }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b)
            *reinterpret_cast<u32x4_t*>(V + ((long)(a * 4 + b) * T + t) * C + cc * 8) = pack8(v[a][b]);
}

// M: [16][T][Co] bf16; y: padded NHWC [N][H + 2][W + 2][Co] (interior written); bias f32 [Co].  One thread per (tile, 8 channels).
__global__ void wino_output_kernel(const bf16_t* __restrict__ M, const float* __restrict__ bias, bf16_t* __restrict__ y, int N, int H,
                                   int W, int Co) {
    const int c8 = Co / 8;
    const long T = (long)N * (H / 2) * (W / 2);
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T * c8) return;
    const int cc = i % c8; const long t = i / c8;
    const int tx = t % (W / 2); const long r = t / (W / 2); const int ty = r % (H / 2); const int n = r / (H / 2);
    float m[4][4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) unpack8(*reinterpret_cast<const u32x4_t*>(M + ((long)(a * 4 + b) * T + t) * Co + cc * 8), m[a][b]);
    float o[2][2][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float tm[2][4];
#pragma unroll
        for (int b = 0; b < 4; ++b) { tm[0][b] = m[0][b][e] + m[1][b][e] + m[2][b][e]; tm[1][b] = m[1][b][e] - m[2][b][e] - m[3][b][e]; }
        const float bs = bias ? bias[cc * 8 + e] : 0.f;
#pragma unroll
        for (int a = 0; a < 2; ++a) { o[a][0][e] = tm[a][0] + tm[a][1] + tm[a][2] + bs; o[a][1][e] = tm[a][1] - tm[a][2] - tm[a][3] + bs; }
    }
    bf16_t* base = y + (((long)n * (H + 2) + 2 * ty + 1) * (W + 2) + 2 * tx + 1) * Co + cc * 8;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) *reinterpret_cast<u32x4_t*>(base + ((long)a * (W + 2) + b) * Co) = pack8(o[a][b]);
}

extern "C" {
int wino_input(const void* x, void* V, int N, int H, int W, int C, void* stream) {
    const long n = (long)N * (H / 2) * (W / 2) * (C / 8);
    wino_input_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)x, (bf16_t*)V, N, H, W, C);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
int wino_output(const void* M, const float* bias, void* y, int N, int H, int W, int Co, void* stream) {
    const long n = (long)N * (H / 2) * (W / 2) * (Co / 8);
    wino_output_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const bf16_t*)M, bias, (bf16_t*)y, N, H, W, Co);
    return hipGetLastError() == hipSuccess ? 0 : 2;
}
}
