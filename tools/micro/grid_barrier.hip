// Micro-benchmark: what does a device-wide barrier INSIDE one resident kernel cost on MI355X?  G co-resident workgroups
// (one or two per compute unit), T threads each; a barrier = every workgroup's thread 0 releases (agent scope) an increment
// of one global counter, spins (s_sleep between polls) on an acquire load until the counter reaches G x (epoch + 1), then the
// workgroup's own __syncthreads.  Between two barriers every thread does a token amount of global work (a read-modify-write
// of its own word, so the barrier has something to order).  Prints microseconds per barrier, against the ~5 us (launch floor,
// rocprofv3 p10 of an empty-ish kernel in a captured graph) and 8-12 us (back-to-back launches on a stream) a kernel boundary
// costs.   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ void grid_barrier(unsigned* counter, unsigned target, int sleep) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (sleep == 1) __builtin_amdgcn_s_sleep(1); else if (sleep == 2) __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
}

// two-level form: workgroups arrive on one of `groups` counters (128 bytes apart: different channels), the last arrival of a
// group on the root; everybody polls the root with plain acquire loads.  Serialised atomics per barrier: G / groups + groups
// instead of G.
__device__ __forceinline__ void grid_barrier_tree(unsigned* counters /* [1 + groups] x 32 words */, int groups, unsigned epoch, int sleep) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const int G = gridDim.x, g = blockIdx.x % groups;
        const unsigned in_group = (unsigned)((G - g + groups - 1) / groups);
        const unsigned old = __hip_atomic_fetch_add(counters + 32 * (1 + g), 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == in_group * (epoch + 1)) __hip_atomic_fetch_add(counters, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counters, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)groups * (epoch + 1)) {
            if (sleep == 1) __builtin_amdgcn_s_sleep(1); else if (sleep == 2) __builtin_amdgcn_s_sleep(8);
        }
    }
    __syncthreads();
}

__global__ void k_bar_tree(unsigned* counters, int groups, float* data, int n_bar, int sleep, long long* clocks) {
    const int G = gridDim.x;
    float* mine = data + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long t0 = 0;
    for (int e = 0; e < n_bar; ++e) {
        if (e == 1 && blockIdx.x == 0 && threadIdx.x == 0) t0 = wall_clock64();
        *mine += 1.f;
        grid_barrier_tree(counters, groups, (unsigned)e, sleep);
        const float other = data[(size_t)((blockIdx.x + 1) % G) * blockDim.x + threadIdx.x];
        if (other < (float)(e + 1) && clocks) atomicAdd((unsigned long long*)&clocks[2], 1ull);      // (the neighbour may be an epoch ahead, never behind)
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = wall_clock64() - t0; clocks[1] = n_bar - 1; }
}

__global__ void k_bar(unsigned* counter, float* data, int n_bar, int sleep, long long* clocks) {
    const int G = gridDim.x;
    float* mine = data + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    long long t0 = 0;
    for (int e = 0; e < n_bar; ++e) {
        if (e == 1 && blockIdx.x == 0 && threadIdx.x == 0) t0 = wall_clock64();     // (the first barrier absorbs the launch skew)
        *mine += 1.f;
        grid_barrier(counter, (unsigned)G * (unsigned)(e + 1), sleep);
        // read a neighbour workgroup's word: the barrier must have made it visible
        const float other = data[(size_t)((blockIdx.x + 1) % G) * blockDim.x + threadIdx.x];
        if (other < (float)(e + 1) && clocks) atomicAdd((unsigned long long*)&clocks[2], 1ull);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { clocks[0] = wall_clock64() - t0; clocks[1] = n_bar - 1; }
}

__global__ void k_empty(float* data) { if (threadIdx.x == 9999) data[0] = 1.f; }

int main() {
    unsigned* counter; float* data; long long* clocks;
    hipMalloc(&counter, 4); hipMalloc(&data, 4096 * 1024 * 4); hipMalloc(&clocks, 64);
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("%s: %d CUs, wall clock 100 MHz\n", p.gcnArchName, p.multiProcessorCount);
    const int n_bar = 201;
    for (int T : {256, 512, 1024})
        for (int G : {64, 128, 256, 512})
            for (int sleep : {0, 2}) {
                if ((size_t)G * T > 4096 * 1024) continue;
                int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bar, T, 0);
                if (G > occ * p.multiProcessorCount) continue;          // not co-resident: the barrier would deadlock
                float best = 1e30f; long long h[3] = {0, 0, 0};
                for (int r = 0; r < 3; ++r) {
                    hipMemset(counter, 0, 4); hipMemset(data, 0, (size_t)G * T * 4); hipMemset(clocks, 0, 64);
                    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                    hipEventRecord(e0);
                    hipLaunchKernelGGL(k_bar, dim3(G), dim3(T), 0, 0, counter, data, n_bar, sleep, clocks);
                    hipEventRecord(e1); hipDeviceSynchronize();
                    float ms; hipEventElapsedTime(&ms, e0, e1);
                    hipMemcpy(h, clocks, 24, hipMemcpyDeviceToHost);
                    const float us = (float)h[0] / 100.f / (float)h[1];
                    best = us < best ? us : best;
                }
                printf("G %4d workgroups x %4d threads, poll %s: %6.2f us per barrier (visibility errors %lld)\n", G, T,
                       sleep == 0 ? "busy     " : sleep == 1 ? "s_sleep 1" : "s_sleep 8", best, h[2]);
            }
    {
        unsigned* ctr; hipMalloc(&ctr, 65 * 32 * 4);
        for (int T : {256, 1024})
            for (int G : {64, 128, 256, 512})
                for (int groups : {8, 16, 32}) {
                    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, k_bar_tree, T, 0);
                    if (G > occ * p.multiProcessorCount) continue;
                    float best = 1e30f; long long h[3] = {0, 0, 0};
                    for (int r = 0; r < 3; ++r) {
                        hipMemset(ctr, 0, 65 * 32 * 4); hipMemset(data, 0, (size_t)G * T * 4); hipMemset(clocks, 0, 64);
                        hipLaunchKernelGGL(k_bar_tree, dim3(G), dim3(T), 0, 0, ctr, groups, data, n_bar, 0, clocks);
                        hipDeviceSynchronize();
                        hipMemcpy(h, clocks, 24, hipMemcpyDeviceToHost);
                        const float us = (float)h[0] / 100.f / (float)h[1];
                        best = us < best ? us : best;
                    }
                    printf("two-level: G %4d workgroups x %4d threads, %2d groups: %6.2f us per barrier (visibility errors %lld)\n", G, T, groups, best, h[2]);
                }
    }
    // for scale: back-to-back empty launches on one stream
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0);
        for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, 0, data);
        hipEventRecord(e1); hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("200 empty launches back to back: %.2f us per launch\n", ms * 1e3 / 200);
    }
    // and the same 200 launches as one captured graph
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, data);
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 2; ++r) {
        hipEventRecord(e0, s); hipGraphLaunch(ge, s); hipEventRecord(e1, s); hipStreamSynchronize(s);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("200 empty kernels in one captured graph: %.2f us per node\n", ms * 1e3 / 200);
    }
    return 0;
}
