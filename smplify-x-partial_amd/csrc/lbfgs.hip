// lbfgs.hip -- stand-alone launch of the optimiser tick (see lbfgs_body.h).
#include "lbfgs_body.h"

size_t sfx_optstate_size() { return sizeof(OptState); }

__global__ __launch_bounds__(64)
void k_lbfgs_tick(DevModel M, BatchDev D, const VarList* __restrict__ vls, int first_stage, int last_stage,
                  int init, int step_mode) {
    __shared__ float s_al[SFX_HIST + 2 * LB_BS];
    __shared__ OptScal s_state;
    const int b = blockIdx.x;
    __shared__ LbCoop cp;
    lbfgs_tick_body<1>(M, D, vls, first_stage, last_stage, init, step_mode, b, threadIdx.x, s_al, s_state, cp,
                    D.f + b, D.g + (size_t)b * SFX_NVAR_MAX);
}

void launch_lbfgs_tick(const DevModel& M, const BatchDev& D, const VarList* vl_dev, int first_stage,
                       int last_stage, int init, int step_mode, hipStream_t s) {
    hipLaunchKernelGGL(k_lbfgs_tick, dim3(D.cfg.B), dim3(64), 0, s, M, D, vl_dev, first_stage, last_stage, init,
                       step_mode);
}
