// closure.hip -- stand-alone launch of the per-frame closure (see closure_body.h).
#include "closure_body.h"

template <class LDS>
__global__ __launch_bounds__(LDS::kThreads)
void k_closure(DevModel M, BatchDev D, const VarList* __restrict__ vls, const StageW* __restrict__ sws,
               ClosureArgs args) {
    __shared__ LDS S;
    closure_body(S, M, D, vls, sws, args, blockIdx.x, nullptr, nullptr);
}

void launch_closure(const DevModel& M, const BatchDev& D, const VarList* vl_dev, const StageW* sw_dev,
                    const ClosureArgs& a, hipStream_t s) {
    if (sfx_small_closure(M, D)) hipLaunchKernelGGL(k_closure<FrameLDSSmall>, dim3(D.cfg.B), dim3(FrameLDSSmall::kThreads), 0, s, M, D, vl_dev, sw_dev, a);
    else hipLaunchKernelGGL(k_closure<FrameLDS>, dim3(D.cfg.B), dim3(FrameLDS::kThreads), 0, s, M, D, vl_dev, sw_dev, a);
}
