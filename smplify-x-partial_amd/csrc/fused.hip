// fused.hip -- the fitting loop with as few launches as the data dependencies allow.
//
//  k_fit_rows   (needed-rows LBS): ONE workgroup per frame runs that frame's whole schedule
//               -- closure (256 threads) -> optimiser tick (wavefront 0) -> closure -> ... --
//               without returning to the host: no launch latency, no lock-step between frames;
//               workgroups of finished frames retire and queued frames take their CU.
//  k_tick_dense (dense LBS): between two launches of the batched MFMA GEMM (lbs_dense.hip) one
//               launch does [loss + adjoint of evaluation i] -> [optimiser tick] -> [pose
//               assembly / kinematic chain of evaluation i+1 and its export to the GEMM operands].
#include "closure_body.h"
#include "lbfgs_body.h"

#ifndef SFX_SETS_SMALL8
#define SFX_SETS_SMALL8 3      // register sets of the two-loop recursion in the body-only tick kernels with a workgroup per CU
#endif
#ifndef SFX_TICK_OCC
#define SFX_TICK_OCC 2
#endif

template <class LDS>
__global__ __launch_bounds__(LDS::kThreads, LDS::kBlocksPerCU)
void k_fit_rows(DevModel M, BatchDev D, const VarList* __restrict__ vls, const StageW* __restrict__ sws,
                int first_stage, int last_stage, int max_ticks) {
    __shared__ LDS S;
    __shared__ float gflat[SFX_NVAR_MAX];
    __shared__ float fval;
    __shared__ float s_al[SFX_HIST_MAX + 2 * LB_BS];
    __shared__ OptScal st;
    const int b = blockIdx.x;
    ClosureArgs a{};
    a.stage_override = -2;
    for (int it = 0; it < max_ticks; ++it) {
        if (D.stage[b] > last_stage) break;
        closure_body(S, M, D, vls, sws, a, b, gflat, &fval);
        a.keep_tables = 1;
        __syncthreads();
        // (one wavefront: sharing the dot products of the two-loop recursion between four was measured slower --
        //  a workgroup barrier per block of 8 history pairs costs more than the reductions it removes)
        if (threadIdx.x < 64)
            lbfgs_tick_body<2>(M, D, vls, first_stage, last_stage, 0, 0, b, threadIdx.x, s_al, st, S.T, &fval, gflat);      // (S.T: >= 2048 floats of closure scratch, dead between evaluations)
        __syncthreads();
    }
}

// (small variant, OCC = 2: two workgroups per CU -- batches of more than 256 frames then run two latency-bound frames per
//  CU; OCC = 1 is the same code with the whole register file, launched when every frame has a CU of its own: no spills,
//  73.8 instead of 76.2 us per launch)
template <class LDS, int OCC>
__global__ __launch_bounds__(LDS::kThreads, OCC)
void k_tick_dense(DevModel M, BatchDev D, const VarList* __restrict__ vls, const StageW* __restrict__ sws,
                  int first_stage, int last_stage, int has_eval) {
    __shared__ LDS S;
    __shared__ float gflat[SFX_NVAR_MAX];
    __shared__ float fval;
    __shared__ float s_al[SFX_HIST_MAX + 2 * LB_BS];
    __shared__ OptScal st;
    // PF (a workgroup per CU: LDS to spare): the optimiser tick's working set -- its 8 vectors, X, Xt, the scalar state, both
    // variable lists -- is requested with the loss / adjoint pass's entry batch (LDS-DMA into memory of its own, waited for
    // by that entry's barrier) instead of at the tick's start, where it was a memory round trip of the frame's serial chain;
    // the pass that follows the tick takes the trial point and the stage from LDS instead of reading back what the tick
    // has just stored (two more round trips).  Same values either way.
    constexpr bool PF = (OCC == 1);
    constexpr int CTK = LDS::kThreads;
    __shared__ __align__(16) float s_pf[PF ? 2048 : 4];
    __shared__ __align__(16) VarList s_vl[PF ? 2 : 1];
    __shared__ int s_stage;
    static_assert(sizeof(VarList) % 4 == 0 && sizeof(OptScal) % 4 == 0 && (NVEC * SFX_NVAR_MAX) % 4 == 0, "dword / 16-byte LDS-DMA");
    const int b = D.act ? D.act[blockIdx.x] : blockIdx.x;      // (frames that finished or still wait in the queue are not launched)
    if (D.stage[b] > last_stage) return;
    const long long wc0 = D.dbg ? wall_clock64() : 0;      // debug: per-workgroup duration statistics (100 MHz clock)
    // debug clocks: stamps freeze after launch number dbg[61] of this kernel, so a mid-fit launch is what is read back
    if (D.dbg && blockIdx.x == 0 && threadIdx.x == 0) { D.dbg[62] += 1; if (D.dbg[62] <= D.dbg[61]) D.dbg[24] = clock64(); }
    if (has_eval) {
        ClosureArgs a{};
        a.stage_override = -2; a.use_dense_verts = 1; a.reuse_fwd = 1;
        if constexpr (PF) {
            lds_fill_async16<CTK>(s_pf, D.vec + (size_t)b * NVEC * SFX_NVAR_MAX, NVEC * SFX_NVAR_MAX / 4);
            lds_fill_async16<CTK>(s_pf + NVEC * SFX_NVAR_MAX, D.X + (size_t)b * SFX_NPAR_MAX, SFX_NPAR_MAX / 4);
            lds_fill_async16<CTK>(s_pf + NVEC * SFX_NVAR_MAX + SFX_NPAR_MAX, D.Xt + (size_t)b * SFX_NPAR_MAX, SFX_NPAR_MAX / 4);
            lds_fill_async<CTK>(&st, &(reinterpret_cast<const OptState*>(D.opt) + b)->s, (int)(sizeof(OptScal) / 4));
            lds_fill_async<CTK>(s_vl, vls, (int)(2 * sizeof(VarList) / 4));
        }
        closure_body(S, M, D, vls, sws, a, b, gflat, &fval);
        __syncthreads();
        const long long wc1 = D.dbg ? wall_clock64() : 0;
        // (one wavefront: sharing the dot products of the two-loop recursion between four was measured slower --
        //  a workgroup barrier per block of 8 history pairs costs more than the reductions it removes)
        if (threadIdx.x < 64) {
            if constexpr (PF)
                lbfgs_tick_body<((LDS::kMaxItems <= SFX_SMALL_ITEMS) ? SFX_SETS_SMALL8 : 2), true>(M, D, s_vl, first_stage, last_stage, 0, 0, b, threadIdx.x, s_al, st, s_pf, &fval, gflat, &s_stage);
            else
                lbfgs_tick_body<2>(M, D, vls, first_stage, last_stage, 0, 0, b, threadIdx.x, s_al, st, S.T, &fval, gflat);      // (S.T: >= 2048 floats of closure scratch, dead between evaluations)
        }
        __syncthreads();
        if (D.dbg && threadIdx.x == 0) {       // debug: mean duration of the two segments over all workgroups (100 MHz ticks)
            const long long wc2 = wall_clock64();
            atomicAdd((unsigned long long*)&D.dbg[29], (unsigned long long)(wc1 - wc0));
            atomicAdd((unsigned long long*)&D.dbg[30], (unsigned long long)(wc2 - wc1));
        }
        if (D.dbg && b == 0 && threadIdx.x == 0 && D.dbg[62] <= D.dbg[61]) { D.dbg[25] = clock64(); for (int i = 0; i < 17; ++i) D.dbg[40 + i] = D.dbg[i]; }
        if ((PF ? s_stage : D.stage[b]) > last_stage) {          // finished in this launch: its column carries no collision weight any more
            if (D.pen_want && threadIdx.x == 0) D.pen_want[D.slot[b]] = 0;
            return;
        }
    }
    const int stage_now = (PF && has_eval) ? s_stage : D.stage[b];
    // interpenetration: does the evaluation exported below carry a collision weight (fitting.py:437)?  Kept here, per column, by
    // the workgroup that knows the frame's stage (a memset and a launch of their own per round until round 4)
    if (D.pen_want && threadIdx.x == 0) {
        const int st_ = stage_now;
        D.pen_want[D.slot[b]] = (st_ >= 0 && st_ < D.cfg.n_stages) ? (sws[st_].coll > 0.f ? 1 : 0) : 0;
    }
    ClosureArgs e{};
    e.stage_override = -2; e.export_dense = 1; e.forward_only = 2; e.keep_tables = has_eval;
    if (PF && has_eval) { e.stage_override = stage_now; e.x_lds = s_pf + NVEC * SFX_NVAR_MAX + SFX_NPAR_MAX; }
    closure_body(S, M, D, vls, sws, e, b, nullptr, nullptr);
    if (D.dbg && b == 0 && threadIdx.x == 0 && D.dbg[62] <= D.dbg[61]) D.dbg[26] = clock64();
    if (D.dbg && threadIdx.x == 0) {
        const long long dt = wall_clock64() - wc0;
        atomicMax((unsigned long long*)&D.dbg[58], (unsigned long long)dt);
        atomicAdd((unsigned long long*)&D.dbg[59], (unsigned long long)dt);
        atomicAdd((unsigned long long*)&D.dbg[60], 1ull);
    }
}

void launch_fit_rows(const DevModel& M, const BatchDev& D, const VarList* vl_dev, const StageW* sw_dev,
                     int first_stage, int last_stage, int max_ticks, hipStream_t s) {
    if (sfx_small_closure(M, D))
        hipLaunchKernelGGL(k_fit_rows<FrameLDSSmall>, dim3(D.cfg.B), dim3(FrameLDSSmall::kThreads), 0, s, M, D, vl_dev, sw_dev, first_stage, last_stage, max_ticks);
    else
        hipLaunchKernelGGL(k_fit_rows<FrameLDS>, dim3(D.cfg.B), dim3(FrameLDS::kThreads), 0, s, M, D, vl_dev, sw_dev, first_stage, last_stage, max_ticks);
}
void launch_tick_dense(const DevModel& M, const BatchDev& D, const VarList* vl_dev, const StageW* sw_dev,
                       int first_stage, int last_stage, int has_eval, hipStream_t s) {
    const int grid = D.act ? D.nrun : D.cfg.B;
    if (grid <= 0) return;
    static const int n_cu = [] { hipDeviceProp_t p; int dev = 0; (void)hipGetDevice(&dev); return hipGetDeviceProperties(&p, dev) == hipSuccess ? p.multiProcessorCount : 256; }();
    if (sfx_small_closure(M, D)) {
        // a workgroup per CU: eight wavefronts (FrameLDSSmall8: same bits as four, closure_body.h); lab build, SFX_TICK_THREADS=256:
        // the four-wavefront kernel of round 3 (measurement switch)
#ifdef SFX_LAB
        static const bool t256 = [] { const char* e = getenv("SFX_TICK_THREADS"); return e && atoi(e) == 256; }();
#else
        constexpr bool t256 = false;
#endif
        if (grid <= n_cu && !t256)
            hipLaunchKernelGGL((k_tick_dense<FrameLDSSmall8, 1>), dim3(grid), dim3(FrameLDSSmall8::kThreads), 0, s, M, D, vl_dev, sw_dev, first_stage, last_stage, has_eval);
        else if (grid <= n_cu)
            hipLaunchKernelGGL((k_tick_dense<FrameLDSSmall, 1>), dim3(grid), dim3(FrameLDSSmall::kThreads), 0, s, M, D, vl_dev, sw_dev, first_stage, last_stage, has_eval);
        else
            hipLaunchKernelGGL((k_tick_dense<FrameLDSSmall, SFX_TICK_OCC>), dim3(grid), dim3(FrameLDSSmall::kThreads), 0, s, M, D, vl_dev, sw_dev, first_stage, last_stage, has_eval);
    } else
        hipLaunchKernelGGL((k_tick_dense<FrameLDS, 1>), dim3(grid), dim3(FrameLDS::kThreads), 0, s, M, D, vl_dev, sw_dev, first_stage, last_stage, has_eval);
}
