// Micro-benchmark: how fast ONE workgroup per compute unit streams a matrix that every workgroup reads (L2-resident,
// the situation of the VPoser matrix-vector products and the needed-rows adjoint of the per-frame kernel), by number of
// wavefronts per workgroup and 16-byte loads in flight per lane.  MI355X.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_cu.hip -o /tmp/stream_cu && /tmp/stream_cu
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT, int UNR>
__global__ __launch_bounds__(NT) void k(const float4* __restrict__ W, float* out, int n4, int reps) {
    extern __shared__ float pad[];              // forces one workgroup per CU
    float4 acc = {0, 0, 0, 0};
    for (int r = 0; r < reps; ++r) {
        for (int i0 = threadIdx.x; i0 < n4; i0 += NT * UNR) {
            float4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = W[(i0 + u * NT) % n4];
#pragma unroll
            for (int u = 0; u < UNR; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (threadIdx.x == 0) pad[0] = acc.x;
    out[blockIdx.x * NT + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + pad[0] * 0.f;
}
template <int NT, int UNR> void run(const float4* W, float* out, int n4, int grid) {
    const int reps = 8;
    hipFuncSetAttribute((const void*)k<NT, UNR>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NT, UNR>), dim3(grid), dim3(NT), 120 * 1024, 0, W, out, n4, reps);
    hipEventRecord(e0); hipLaunchKernelGGL((k<NT, UNR>), dim3(grid), dim3(NT), 120 * 1024, 0, W, out, n4, reps); hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)n4 * 16 * reps;
    printf("grid %3d  %4d threads, %2d loads in flight per lane: %7.1f us per MB streamed, %6.1f GB/s per CU (%.1f B/clk at 2.4 GHz), %5.1f TB/s aggregate\n",
           grid, NT, UNR, ms * 1e3 / (bytes / 1048576.0), bytes / ms / 1e6, bytes / ms / 1e6 / 2.4, bytes * grid / ms / 1e9);
}
int main() {
    const int n4 = 1 << 16;                     // 1 MiB matrix
    float4* W; float* out; hipMalloc(&W, n4 * 16); hipMalloc(&out, 256 * 1024 * 4); hipMemset(W, 0, n4 * 16);
    for (int grid : {1, 32, 256}) {
        run<256, 8>(W, out, n4, grid); run<256, 16>(W, out, n4, grid);
        run<512, 8>(W, out, n4, grid); run<512, 16>(W, out, n4, grid);
        run<1024, 4>(W, out, n4, grid); run<1024, 8>(W, out, n4, grid);
    }
    return 0;
}
