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
#include <cstdlib>
#include <cstddef>

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


// ---------------------------------------------------------------------------------------------------------------------
// EXPERIMENT (round 6, measured and reverted by the next commit: LAB_NOTES R6.5): the SPLIT form of the tick for batches with the
// VPoser decoder in the loop and 200 or more running frames.  The decoder's products as kernels of their own over all running
// frames, VP_FB frames per workgroup and pass over the weights, between the three pieces of the tick:
//     k_tick_a (loss + adjoint, d body_pose -> D.vp_dbody) -> k_vp_backward (d latent added to D.g) -> k_tick_b (optimiser tick)
//     -> k_vp_forward (latent of the trial point -> activations + body pose in D.fwd) -> k_tick_c (pose, chain, GEMM operands)
// The products keep vp_gemv_partial<512>'s association -- a thread's k range, the group order of the partial sums -- so the two
// forms are interchangeable bit for bit (tested).
#define VP_FB 8                 // frames per workgroup of the decoder kernels
#define VP_NT SFX_BIG_THREADS   // their threads: the fused kernel's (the association of its partial sums is defined on them)
static_assert(VP_NT == 512, "vp_gemv_partial<512>'s thread-to-(row quad, k group) mapping");

__device__ __forceinline__ int vp_gemv_partial_batched(const float* __restrict__ Wt, const int K, const int ld, const int nout,
                                                       const float* xs /* [VP_FB][512] */, float* part /* [VP_FB][2048] */) {
    const int t = threadIdx.x;
    const int tpr = nout >> 2;
    const int groups = VP_NT / tpr;
    const int c4 = t % tpr, g = t / tpr;
    if (g < groups) {
        const int k0 = (K * g) / groups, k1 = (K * (g + 1)) / groups;
        float4 acc[VP_FB];
#pragma unroll
        for (int f = 0; f < VP_FB; ++f) acc[f] = float4{0.f, 0.f, 0.f, 0.f};
        const float4* w = reinterpret_cast<const float4*>(Wt + (size_t)k0 * ld) + c4;
        const int ld4 = ld >> 2;
#pragma unroll 4
        for (int k = k0; k < k1; ++k, w += ld4) {
            const float4 wv = *w;
#pragma unroll
            for (int f = 0; f < VP_FB; ++f) {
                const float xv = xs[f * VP_H + k];
                acc[f].x += wv.x * xv; acc[f].y += wv.y * xv; acc[f].z += wv.z * xv; acc[f].w += wv.w * xv;
            }
        }
#pragma unroll
        for (int f = 0; f < VP_FB; ++f) reinterpret_cast<float4*>(part + (size_t)f * 2048 + (size_t)g * nout)[c4] = acc[f];
    }
    return groups;
}

__device__ __forceinline__ int vp_frame(const BatchDev& D, const int slot, const int last_stage) {
    const int n = D.act ? D.nrun : D.cfg.B;
    if (slot >= n) return -1;
    const int b = D.act ? D.act[slot] : slot;
    return D.stage[b] > last_stage ? -1 : b;
}

__global__ __launch_bounds__(VP_NT)
void k_vp_forward(DevModel M, BatchDev D, int last_stage, int fwd_prefix) {
    extern __shared__ float vp_lds[];
    float* xa = vp_lds;                         // [VP_FB][512] input vector of the current product
    float* xb = xa + VP_FB * VP_H;              // [VP_FB][512] its output
    float* part = xb + VP_FB * VP_H;            // [VP_FB][2048]
    __shared__ int s_b[VP_FB];
    const int t = threadIdx.x, L = M.vp_latent;
    if (t < VP_FB) s_b[t] = vp_frame(D, blockIdx.x * VP_FB + t, last_stage);
    __syncthreads();
    bool any = false;
    for (int f = 0; f < VP_FB; ++f) any = any || s_b[f] >= 0;
    if (!any) return;
    for (int i = t; i < VP_FB * L; i += VP_NT) {
        const int f = i / L, k = i % L, b = s_b[f];
        xa[f * VP_H + k] = b >= 0 ? D.Xt[(size_t)b * SFX_NPAR_MAX + D.L.emb + k] : 0.f;
    }
    __syncthreads();
    int G = vp_gemv_partial_batched(M.vp_w1T, L, VP_H, VP_H, xa, part);
    __syncthreads();
    for (int i = t; i < VP_FB * VP_H; i += VP_NT) {
        const int f = i / VP_H, o = i % VP_H;
        float acc = M.vp_b1[o];
        for (int g = 0; g < G; ++g) acc += part[(size_t)f * 2048 + g * VP_H + o];
        xb[i] = leaky(acc);                     // h1
    }
    __syncthreads();
    G = vp_gemv_partial_batched(M.vp_w2T, VP_H, VP_H, VP_H, xb, part);
    __syncthreads();
    for (int i = t; i < VP_FB * VP_H; i += VP_NT) {
        const int f = i / VP_H, o = i % VP_H;
        float acc = M.vp_b2[o];
        for (int g = 0; g < G; ++g) acc += part[(size_t)f * 2048 + g * VP_H + o];
        xa[i] = leaky(acc);                     // h2
    }
    __syncthreads();
    G = vp_gemv_partial_batched(M.vp_w3T, VP_H, 128, 128, xa, part);
    __syncthreads();
    for (int i = t; i < VP_FB * VP_H; i += VP_NT) {
        const int f = i / VP_H, o = i % VP_H, b = s_b[f];
        if (b >= 0) { float* vx = D.fwd + (size_t)b * SFX_FWD_N + fwd_prefix + 96; vx[o] = xb[i]; vx[VP_H + o] = xa[i]; }
    }
    float ov[(VP_FB * 128 + VP_NT - 1) / VP_NT];
#pragma unroll
    for (int q = 0; q < (VP_FB * 128 + VP_NT - 1) / VP_NT; ++q) {
        const int i = t + q * VP_NT, f = i / 128, o = i % 128;
        float acc = 0.f;
        if (i < VP_FB * 128 && o < VP_O) { acc = M.vp_b3[o]; for (int g = 0; g < G; ++g) acc += part[(size_t)f * 2048 + g * 128 + o]; }
        ov[q] = acc;
    }
    __syncthreads();
    float* os = xb;                             // [VP_FB][128] (h1 is stored)
#pragma unroll
    for (int q = 0; q < (VP_FB * 128 + VP_NT - 1) / VP_NT; ++q) { const int i = t + q * VP_NT; if (i < VP_FB * 128) os[i] = ov[q]; }
    __syncthreads();
    float* bs = xb + VP_FB * 128;               // [VP_FB][64]
    if (t < VP_FB * 21) { const int f = t / 21, j = t % 21; vposer_joint(&os[f * 128 + 6 * j], &bs[f * 64 + 3 * j], nullptr, nullptr); }
    if (t >= VP_NT - VP_FB) bs[(t - (VP_NT - VP_FB)) * 64 + 63] = 0.f;
    __syncthreads();
    for (int i = t; i < VP_FB * 128; i += VP_NT) {
        const int f = i / 128, o = i % 128, b = s_b[f];
        if (b >= 0) D.fwd[(size_t)b * SFX_FWD_N + fwd_prefix + 96 + 2 * VP_H + o] = os[i];
    }
    for (int i = t; i < VP_FB * 64; i += VP_NT) {
        const int f = i / 64, o = i % 64, b = s_b[f];
        if (b >= 0) { D.fwd[(size_t)b * SFX_FWD_N + fwd_prefix + 96 + 2 * VP_H + 128 + o] = bs[i]; if (o < 63) D.bodypose[(size_t)b * 63 + o] = bs[i]; }
    }
}

__global__ __launch_bounds__(VP_NT)
void k_vp_backward(DevModel M, BatchDev D, const VarList* __restrict__ vls, int last_stage, int fwd_prefix) {
    extern __shared__ float vp_lds[];
    float* xa = vp_lds;
    float* xb = xa + VP_FB * VP_H;
    float* part = xb + VP_FB * VP_H;
    __shared__ int s_b[VP_FB];
    const int t = threadIdx.x, L = M.vp_latent;
    if (t < VP_FB) { const int b = vp_frame(D, blockIdx.x * VP_FB + t, last_stage); s_b[t] = (b >= 0 && D.stage[b] >= 0) ? b : -1; }      // (the camera stage has no latent variable)
    __syncthreads();
    bool any = false;
    for (int f = 0; f < VP_FB; ++f) any = any || s_b[f] >= 0;
    if (!any) return;
    if (t < VP_FB * 21) {
        const int f = t / 21, j = t % 21, b = s_b[f];
        float o6[6], db[3], aa[3], d6[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (b >= 0) {
            const float* vx = D.fwd + (size_t)b * SFX_FWD_N + fwd_prefix + 96;
            for (int e = 0; e < 6; ++e) o6[e] = vx[2 * VP_H + 6 * j + e];
            for (int e = 0; e < 3; ++e) db[e] = D.vp_dbody[(size_t)b * 64 + 3 * j + e];
            vposer_joint(o6, aa, db, d6);
        }
        for (int e = 0; e < 6; ++e) xa[f * VP_H + 6 * j + e] = d6[e];
    }
    if (t >= 256 && t < 256 + 2 * VP_FB) xa[((t - 256) >> 1) * VP_H + VP_O + ((t - 256) & 1)] = 0.f;
    __syncthreads();
    int G = vp_gemv_partial_batched(M.vp_w3, VP_O, VP_H, VP_H, xa, part);      // d h2 = W3^T d o, through leaky'
    __syncthreads();
    for (int i = t; i < VP_FB * VP_H; i += VP_NT) {
        const int f = i / VP_H, o = i % VP_H, b = s_b[f];
        float acc = 0.f;
        for (int g = 0; g < G; ++g) acc += part[(size_t)f * 2048 + g * VP_H + o];
        const float h2 = b >= 0 ? D.fwd[(size_t)b * SFX_FWD_N + fwd_prefix + 96 + VP_H + o] : 0.f;
        xb[i] = acc * (h2 > 0.f ? 1.f : 0.2f);
    }
    __syncthreads();
    G = vp_gemv_partial_batched(M.vp_w2, VP_H, VP_H, VP_H, xb, part);           // d h1 = W2^T d pre2, through leaky'
    __syncthreads();
    for (int i = t; i < VP_FB * VP_H; i += VP_NT) {
        const int f = i / VP_H, o = i % VP_H, b = s_b[f];
        float acc = 0.f;
        for (int g = 0; g < G; ++g) acc += part[(size_t)f * 2048 + g * VP_H + o];
        const float h1 = b >= 0 ? D.fwd[(size_t)b * SFX_FWD_N + fwd_prefix + 96 + o] : 0.f;
        xa[i] = acc * (h1 > 0.f ? 1.f : 0.2f);
    }
    __syncthreads();
    G = vp_gemv_partial_batched(M.vp_w1, VP_H, L, L, xa, part);                 // d z = W1^T d pre1
    __syncthreads();
    const VarList& vl = vls[1];
    for (int q = t; q < VP_FB * vl.n; q += VP_NT) {
        const int f = q / vl.n, i = q % vl.n, b = s_b[f];
        const int c = vl.idx[i] - D.L.emb;
        if (b < 0 || c < 0 || c >= L) continue;
        float acc = 0.f;
        for (int g = 0; g < G; ++g) acc += part[(size_t)f * 2048 + g * L + c];
        float* gp = D.g + (size_t)b * SFX_NVAR_MAX + i;
        *gp = *gp + acc;
    }
}

__global__ __launch_bounds__(FrameLDS::kThreads, 1)
void k_tick_a(DevModel M, BatchDev D, const VarList* __restrict__ vls, const StageW* __restrict__ sws, int last_stage) {
    __shared__ FrameLDS S;
    const int b = D.act ? D.act[blockIdx.x] : blockIdx.x;
    if (D.stage[b] > last_stage) return;
    ClosureArgs a{};
    a.stage_override = -2; a.use_dense_verts = 1; a.reuse_fwd = 1; a.vp_split = 1;
    closure_body(S, M, D, vls, sws, a, b, nullptr, nullptr);
}
__global__ __launch_bounds__(64)
void k_tick_b(DevModel M, BatchDev D, const VarList* __restrict__ vls, int first_stage, int last_stage) {
    __shared__ float s_al[SFX_HIST_MAX + 2 * LB_BS];
    __shared__ OptScal s_state;
    __shared__ __align__(16) float s_work[2048];
    const int b = D.act ? D.act[blockIdx.x] : blockIdx.x;
    if (D.stage[b] > last_stage) return;
    lbfgs_tick_body<2>(M, D, vls, first_stage, last_stage, 0, 0, b, threadIdx.x, s_al, s_state, s_work,
                       D.f + b, D.g + (size_t)b * SFX_NVAR_MAX);
}
__global__ __launch_bounds__(FrameLDS::kThreads, 1)
void k_tick_c(DevModel M, BatchDev D, const VarList* __restrict__ vls, const StageW* __restrict__ sws, int last_stage) {
    __shared__ FrameLDS S;
    const int b = D.act ? D.act[blockIdx.x] : blockIdx.x;
    const int stage_now = D.stage[b];
    if (stage_now > last_stage) {
        if (D.pen_want && threadIdx.x == 0) D.pen_want[D.slot[b]] = 0;
        return;
    }
    if (D.pen_want && threadIdx.x == 0)
        D.pen_want[D.slot[b]] = (stage_now >= 0 && stage_now < D.cfg.n_stages) ? (sws[stage_now].coll > 0.f ? 1 : 0) : 0;
    ClosureArgs e{};
    e.stage_override = -2; e.export_dense = 1; e.forward_only = 2; e.vp_split = 1;
    closure_body(S, M, D, vls, sws, e, b, nullptr, nullptr);
}

#ifndef SFX_TICK_SPLIT_MIN
#define SFX_TICK_SPLIT_MIN 200      // running frames from which the split form is launched (VPoser batches only)
#endif
static int tick_split_min() {
#ifdef SFX_LAB       // SFX_TICK_SPLIT=n (read per call): the split form from n running frames on (1: always, 1000000: never) -- A/B, same bits
    if (const char* e = getenv("SFX_TICK_SPLIT")) return atoi(e);
#endif
    return SFX_TICK_SPLIT_MIN;
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
    } else if (has_eval && D.cfg.use_vposer && D.fwd && D.vp_dbody && grid >= tick_split_min()) {
        static bool attr = false;
        const size_t lds = (size_t)(2 * VP_FB * VP_H + VP_FB * 2048) * sizeof(float);
        if (!attr) {
            (void)hipFuncSetAttribute((const void*)k_vp_forward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            (void)hipFuncSetAttribute((const void*)k_vp_backward, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr = true;
        }
        constexpr int fwd_prefix = (int)(offsetof(FrameLDS, vp) / sizeof(float));
        const int nvp = (grid + VP_FB - 1) / VP_FB;
        hipLaunchKernelGGL(k_tick_a, dim3(grid), dim3(FrameLDS::kThreads), 0, s, M, D, vl_dev, sw_dev, last_stage);
        hipLaunchKernelGGL(k_vp_backward, dim3(nvp), dim3(VP_NT), lds, s, M, D, vl_dev, last_stage, fwd_prefix);
        hipLaunchKernelGGL(k_tick_b, dim3(grid), dim3(64), 0, s, M, D, vl_dev, first_stage, last_stage);
        hipLaunchKernelGGL(k_vp_forward, dim3(nvp), dim3(VP_NT), lds, s, M, D, last_stage, fwd_prefix);
        hipLaunchKernelGGL(k_tick_c, dim3(grid), dim3(FrameLDS::kThreads), 0, s, M, D, vl_dev, sw_dev, last_stage);
    } else
        hipLaunchKernelGGL((k_tick_dense<FrameLDS, 1>), dim3(grid), dim3(FrameLDS::kThreads), 0, s, M, D, vl_dev, sw_dev, first_stage, last_stage, has_eval);
}
