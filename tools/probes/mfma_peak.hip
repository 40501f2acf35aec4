// Probe: sustained v_mfma_f32_16x16x32_bf16 rate and shader clock with register-resident operands
// (no LDS, no memory): the ceiling a GEMM can reach on this chip with random vs zero data, and with 1 / 2 / 4
// waves per SIMD.   hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak && ./mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__global__ void k(float* out, long long* clk, int iters, int zero) {
    const int lane = threadIdx.x & 63;
    bf16x8_t a[4], b[4];
    uint32_t s = (blockIdx.x * 977 + threadIdx.x) * 2654435761u + 12345u;
    for (int i = 0; i < 4; ++i)
        for (int e = 0; e < 8; ++e) {
            s = s * 1664525u + 1013904223u; short va = zero ? 0 : (short)(0x3c00 | ((s >> 9) & 0x3ff) | ((s >> 3) & 0x8000));   // ~+-[0.0078,0.0156)... bf16 bits
            a[i][e] = zero ? 0 : (short)(0x3f00 | ((s >> 9) & 0xff) | ((s >> 3) & 0x8000));
            s = s * 1664525u + 1013904223u;
            b[i][e] = zero ? 0 : (short)(0x3f00 | ((s >> 9) & 0xff) | ((s >> 3) & 0x8000));
            (void)va;
        }
    f32x4_t acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        if (!zero && (it & 63) == 63) {   // keep magnitudes bounded
#pragma unroll
            for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) acc[i][j] *= 1e-3f;
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    float t = 0.f;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) t += acc[i][j][0] + acc[i][j][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = t;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
    (void)lane;
}

int main() {
    float* out; long long* clk;
    hipMalloc(&out, 256 * 16 * 256 * 4 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int zero = 0; zero < 2; ++zero)
        for (int wps = 1; wps <= 4; wps *= 2) {           // waves per SIMD
            const int threads = 256, blocks = 256 * wps;   // 4 waves per block -> one per SIMD per block
            const int iters = 40000 / wps;
            k<<<blocks, threads>>>(out, clk, 1000, zero);  // warm
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<blocks, threads>>>(out, clk, iters, zero);
            hipEventRecord(e1); hipDeviceSynchronize();
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
            const double flops = (double)blocks * 4 * iters * 16 * 16384.0;
            printf("%s data, %d waves/SIMD: %.1f ms  %.0f TFLOP/s   shader clock %.0f MHz (clock64 %lld / wall %lld @100MHz)\n",
                   zero ? "zero  " : "random", wps, ms, flops / ms / 1e9, (double)h[0] / (double)h[1] * 100.0, h[0], h[1]);
        }
    return 0;
}
