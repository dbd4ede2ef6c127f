// Micro-benchmark: one 512-thread workgroup per compute unit streams 1 MiB that is NOT in its L2 when the kernel starts (the
// situation of the per-frame kernel: the GEMM between two of its launches walks 86 MB through every L2) -- the same MiB for
// all workgroups (VPoser weights, static blend-shape rows) or one of its own, by loads in flight per lane; burst-and-drain
// or a rolling window.   hipcc --offload-arch=gfx950 -O3 tools/micro/stream_cold.hip -o /tmp/stream_cold && /tmp/stream_cold
#include <hip/hip_runtime.h>
#include <cstdio>
template <int UNR, bool ROLL>
__global__ __launch_bounds__(512) void k(const float4* __restrict__ W, float* out, int n4, size_t stride4) {
    extern __shared__ float pad[];              // forces one workgroup per CU
    const float4* w = W + (size_t)blockIdx.x * stride4 + threadIdx.x;
    float4 acc = {0, 0, 0, 0};
    const int rows = n4 / 512;                  // 16-byte loads per thread
    float4 v[UNR];
    if (ROLL) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
        for (int u = 0; u < UNR; ++u) v[u] = w[(size_t)u * 512];
        int i = 0;
        for (; i + UNR < rows; i += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w;
                asm volatile("" : "+v"(acc.x), "+v"(acc.y), "+v"(acc.z), "+v"(acc.w) :: "memory");
                v[u] = w[(size_t)(i + u + UNR) * 512];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
    } else {
        for (int i = 0; i < rows; i += UNR) {
#pragma unroll
            for (int u = 0; u < UNR; ++u) v[u] = w[(size_t)(i + u) * 512];
#pragma unroll
            for (int u = 0; u < UNR; ++u) { acc.x += v[u].x; acc.y += v[u].y; acc.z += v[u].z; acc.w += v[u].w; }
        }
    }
    if (threadIdx.x == 0) pad[0] = acc.x;
    out[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w + pad[0] * 0.f;
}
__global__ void k_flush(const float4* __restrict__ X, float* out, size_t n4) {
    float a = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a += X[i].x;
    if (a == 1234.5f) out[0] = a;
}
template <int UNR, bool ROLL> void run(const float4* W, const float4* X, size_t flush4, float* out, int grid, bool shared, const char* what) {
    const int n4 = 1 << 16;                     // 1 MiB per workgroup
    hipFuncSetAttribute((const void*)k<UNR, ROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f, sum = 0.f; const int reps = 5;
    for (int r = 0; r < reps; ++r) {
        hipLaunchKernelGGL(k_flush, dim3(2048), dim3(256), 0, 0, X, out, flush4);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<UNR, ROLL>), dim3(grid), dim3(512), 120 * 1024, 0, W, out, n4, shared ? (size_t)0 : (size_t)n4);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
    }
    printf("%-28s grid %3d %s window %2d %s: %6.1f us (best %6.1f) per launch = %6.1f GB/s per CU\n", what, grid, shared ? "shared MiB " : "private MiB", UNR,
           ROLL ? "rolling" : "burst  ", sum / reps * 1e3, best * 1e3, 1048576.0 / (sum / reps) / 1e6);
}
int main() {
    const size_t wbytes = (size_t)256 << 20, xbytes = (size_t)1 << 30;
    float4 *W, *X; float* out; hipMalloc(&W, wbytes); hipMalloc(&X, xbytes); hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(W, 0, wbytes); hipMemset(X, 0, xbytes);
    for (int mode = 0; mode < 2; ++mode) {
        const size_t flush4 = mode == 0 ? ((size_t)96 << 20) / 16 : xbytes / 16;       // 96 MB: every L2, not the 256-MB MALL; 1 GB: both
        const char* what = mode == 0 ? "after 96 MB (L2 cold)" : "after 1 GB (L2 + MALL cold)";
        for (int grid : {32, 142, 256})
            for (int sh = 1; sh >= 0; --sh) {
                run<8, false>(W, X, flush4, out, grid, sh, what); run<16, false>(W, X, flush4, out, grid, sh, what);
                run<8, true>(W, X, flush4, out, grid, sh, what); run<16, true>(W, X, flush4, out, grid, sh, what); run<32, true>(W, X, flush4, out, grid, sh, what);
            }
    }
    return 0;
}
