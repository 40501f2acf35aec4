// Probe: sustained v_mfma_f32_32x32x16_bf16 rate with register-resident operands (the companion of mfma_peak.hip,
// which measures the 16x16x32 form): does the 32-cycle instruction amortise the issue gap a lone wave per SIMD
// pays (18.9 cycles per 16-cycle MFMA)?   hipcc --offload-arch=gfx950 -O3 mfma_peak_32.hip -o mfma_peak_32
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__global__ __launch_bounds__(256, 1) void k(float* out, long long* clk, int iters, int zero) {
    bf16x8_t a[4], b[2];
    uint32_t s = (blockIdx.x * 977 + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u;
            a[i][e] = zero ? 0 : (short)(0x3f00 | ((s >> 9) & 0xff) | ((s >> 3) & 0x8000));
            s = s * 1664525u + 1013904223u;
            if (i < 2) b[i][e] = zero ? 0 : (short)(0x3f00 | ((s >> 9) & 0xff) | ((s >> 3) & 0x8000));
        }
    f32x16_t acc[4][2];                                   // the 128 x 64 wave tile of the persistent conv kernel
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16_t{0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (!zero && (it & 63) == 63) {
#pragma unroll
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 2; ++j) acc[i][j] *= 1e-3f;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) t += acc[i][j][0] + acc[i][j][15];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 16 * 256 * 4 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero)
        for (int wps = 1; wps <= 2; wps *= 2) {
            const int threads = 256, blocks = 256 * wps;
            const int iters = 40000 / wps;
            k<<<blocks, threads>>>(out, clk, 1000, zero);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<blocks, threads>>>(out, clk, iters, zero);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double flops = (double)blocks * 4 * iters * 8 * 32768.0;
            printf("32x32x16 %s data, %d waves/SIMD: %.1f ms  %.0f TFLOP/s   shader clock %.0f MHz   %.1f cycles / MFMA / wave\n",
                   zero ? "zero  " : "random", wps, ms, flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0,
                   (double)h[0] / ((double)iters * 8));
        }
    return 0;
}
