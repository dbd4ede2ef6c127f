// lbfgs.hip -- stand-alone launch of the optimiser tick (see lbfgs_body.h).
#include "lbfgs_body.h"

size_t sfx_optstate_size() { return sizeof(OptState); }

__global__ __launch_bounds__(64)
void k_lbfgs_tick(DevModel M, BatchDev D, const VarList* __restrict__ vls, int first_stage, int last_stage,
                  int init, int step_mode) {
    __shared__ float s_al[SFX_HIST_MAX + 2 * LB_BS];
    __shared__ OptScal s_state;
    __shared__ __align__(16) float s_work[2048];
    const int b = blockIdx.x;
    lbfgs_tick_body<3>(M, D, vls, first_stage, last_stage, init, step_mode, b, threadIdx.x, s_al, s_state, s_work,
                    D.f + b, D.g + (size_t)b * SFX_NVAR_MAX);
}

void launch_lbfgs_tick(const DevModel& M, const BatchDev& D, const VarList* vl_dev, int first_stage,
                       int last_stage, int init, int step_mode, hipStream_t s) {
    hipLaunchKernelGGL(k_lbfgs_tick, dim3(D.cfg.B), dim3(64), 0, s, M, D, vl_dev, first_stage, last_stage, init,
                       step_mode);
}

// ---- debug: the blocked recursion on a history given by the caller (tests/test_gpu_optimizer_steps.py)
__global__ __launch_bounds__(64)
void k_debug_two_loop(OptState* gst, float* hist, const float* S_in, const float* Y_in, int cnt, int hist_cap, const float* g_in, float* d_out) {
    __shared__ float s_al[SFX_HIST_MAX + 2 * LB_BS];
    const int lane = threadIdx.x;
    const int R = hist_cap > SFX_HIST ? hist_cap : SFX_HIST, hrows = R + 8;      // (the ring api.hip gives a batch of that history_size)
    float* hY = hist; float* hS = hY + (size_t)hrows * SFX_NVAR_MAX;
    for (int i = lane; i < hrows * LB_BROW; i += 64) { gst->syb[i] = 0.f; gst->syt[i] = 0.f; }
    LB_SYNC();
    int hn = 0, hh = 0; float hd = 1.f;
    for (int i = 0; i < cnt; ++i) {
        const Lane3 y = ld3(Y_in + (size_t)i * SFX_NVAR_MAX, lane, SFX_NVAR_MAX), sv = ld3(S_in + (size_t)i * SFX_NVAR_MAX, lane, SFX_NVAR_MAX);
        const float ys = dot3(y, sv);
        lb_push_pair(hY, hS, gst, hn, hh, y, sv, ys, lane, hist_cap, R);
        hd = ys / dot3(y, y);
        LB_SYNC();
    }
    const Lane3 g = ld3(g_in, lane, SFX_NVAR_MAX);
    Lane3 q; for (int e = 0; e < NE3; ++e) q.v[e] = -g.v[e];
    const Lane3 r = lb_two_loop<3>(hS, hY, gst, s_al, hn, hh, hd, q, lane, R);
    st3_full(d_out, r, lane);
}

int debug_two_loop(const float* S, const float* Y, int cnt, int hist_cap, const float* g, float* d_out) {
    OptState* gst = nullptr; float *hist = nullptr, *dS = nullptr, *dY = nullptr, *dg = nullptr, *dd = nullptr;
    const size_t row = SFX_NVAR_MAX * sizeof(float), hb = (size_t)2 * ((hist_cap > SFX_HIST ? hist_cap : SFX_HIST) + 8) * row;
    int rc = 0;
    auto ok = [&rc](hipError_t e) { if (e != hipSuccess && !rc) rc = (int)e; return e == hipSuccess; };
    if (ok(hipMalloc(&gst, sizeof(OptState))) && ok(hipMalloc(&hist, hb)) && ok(hipMalloc(&dS, row * cnt)) && ok(hipMalloc(&dY, row * cnt)) &&
        ok(hipMalloc(&dg, row)) && ok(hipMalloc(&dd, row)) && ok(hipMemset(gst, 0, sizeof(OptState))) && ok(hipMemset(hist, 0, hb)) &&
        ok(hipMemcpy(dS, S, row * cnt, hipMemcpyHostToDevice)) && ok(hipMemcpy(dY, Y, row * cnt, hipMemcpyHostToDevice)) &&
        ok(hipMemcpy(dg, g, row, hipMemcpyHostToDevice))) {
        hipLaunchKernelGGL(k_debug_two_loop, dim3(1), dim3(64), 0, 0, gst, hist, dS, dY, cnt, hist_cap, dg, dd);
        ok(hipDeviceSynchronize()); ok(hipMemcpy(d_out, dd, row, hipMemcpyDeviceToHost));
    }
    hipFree(gst); hipFree(hist); hipFree(dS); hipFree(dY); hipFree(dg); hipFree(dd);
    return rc;
}
