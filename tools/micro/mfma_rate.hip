// Micro-benchmark: sustained rate of v_mfma_f32_16x16x4_f32 and the shader clock under that load (MI355X).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/mfma_rate.hip -o /tmp/mfma_rate && /tmp/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0;
    const float x = threadIdx.x * 1e-3f, y = blockIdx.x * 1e-3f;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a3, 0, 0, 0);
        a4 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a4, 0, 0, 0);
        a5 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a5, 0, 0, 0);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + a4[0] + a5[1];
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
    float* out; long long* clk; hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&clk, 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpc : {1, 2, 3}) for (int iters : {2000, 20000}) {
        const int grid = 256 * wpc;      // wpc workgroups of 4 wavefronts per CU
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, clk, iters);
        hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, out, clk, iters); hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)grid * 4 * iters * 6 * 2048.0;
        printf("wg/CU %d iters %6d: %8.1f us  %6.1f TFLOP/s   clock64/wall = %.0f cycles/us (%lld cycles, %.1f us); cycles per MFMA per SIMD %.1f\n",
               wpc, iters, ms * 1e3, flops / ms / 1e9, (double)h[0] / (h[1] * 0.01), h[0], h[1] * 0.01, (double)h[0] / (iters * 6.0 * wpc));
    }
    return 0;
}
