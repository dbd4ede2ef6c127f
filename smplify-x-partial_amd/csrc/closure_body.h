// closure.hip -- one workgroup per frame: SMPL-X forward on the needed rows, perspective
// reprojection, GMoF / prior losses, and the hand-derived adjoint, all in LDS.
//
// Replaces one call of the reference's fitting closure (smplifyx/fitting.py:232-273):
//   body_model(...)            external smplx.lbs.lbs         
//   camera(joints)             smplifyx/camera.py:93-117
//   SMPLifyLoss.forward        smplifyx/fitting.py:375-461
//   SMPLifyCameraInitLoss      smplifyx/fitting.py:499-520
//   total_loss.backward()      autograd -> explicit reverse sweep below
//
// Work decomposition (256 threads = 4 wavefronts of 64):
//   pose assembly / Rodrigues / joint regression : one lane per output
//   kinematic chain                              : one lane per joint, level by level
//   needed vertices (<=225 "items")              : one wavefront per 506-long blend-shape
//                                                  dot product, xor-shuffle reduction
//   loss                                         : one lane per keypoint, fixed-order reduce
//   reverse sweep                                : gathers only (no atomics) -> deterministic
#pragma once
#include "sfx_internal.h"
#include "wave_ops.h"
// contraction only inside one source expression (decided by the front end): the stand-alone
// closure kernel and the fused fit kernels then round identically, so optimizer.step() driven from
// the host and sfx_batch_fit walk the same trajectory bit for bit
#pragma clang fp contract(on)
#include "vposer.h"

// Precision of the keypoint forward (rotations, rest joints, kinematic chain, keypoint skinning: fwd_t) and of the
// projection up to the pixel residual (proj_t); see "forward precision" in closure_body.  Measured on the 64 + 64
// reference fits of the benchmark configuration (LAB_NOTES.md §3.1; evaluations per frame / signed mean final-loss
// difference to the reference's fp32 run; the reference: 2 362 in fp32, 4 098 and -2.1 % in fp64):
//   fwd fp64, proj fp64 : 3 221 / -1.7 %   gradient noise 0.13 x torch fp32's: keeps working where the reference stalls
//   fwd fp64, proj fp32 : 2 482 / -0.4 %   <- built: never noisier than the reference, closest to it in loss and in work
//   fwd fp32, proj fp32 : 2 233 / +0.6 %   noise ~ torch's (pointer-jumping chain 1.3 x): stops a little early
#ifdef SFX_FWD_FP32
typedef float fwd_t;
#else
typedef double fwd_t;
#endif
#ifdef SFX_PROJ_FP64
typedef double proj_t;
#else
typedef float proj_t;
#endif

// camera.py:93-117: p_cam = R p + t, pixel = f p_cam.xy / p_cam.z + c; residual gt - pixel, all in T.
template <class T>
__device__ __forceinline__ void project_residual(const fwd_t* pj, const float* Rc, const float* ct, float fx, float fy, float cx,
                                                 float cy, float gtx, float gty, float& pcx, float& pcy, float& pcz, float& rx,
                                                 float& ry) {
    const T p[3] = {(T)pj[0], (T)pj[1], (T)pj[2]};
    const T pxd = (T)Rc[0] * p[0] + (T)Rc[1] * p[1] + (T)Rc[2] * p[2] + (T)ct[0];
    const T pyd = (T)Rc[3] * p[0] + (T)Rc[4] * p[1] + (T)Rc[5] * p[2] + (T)ct[1];
    const T pzd = (T)Rc[6] * p[0] + (T)Rc[7] * p[1] + (T)Rc[8] * p[2] + (T)ct[2];
    pcx = (float)pxd; pcy = (float)pyd; pcz = (float)pzd;
    rx = (float)((T)gtx - ((T)fx * (pxd / pzd) + (T)cx));
    ry = (float)((T)gty - ((T)fy * (pyd / pzd) + (T)cy));
}
__device__ __forceinline__ void sincos_t(double a, double* s, double* c) { sincos(a, s, c); }
__device__ __forceinline__ void sincos_t(float a, float* s, float* c) { *s = sinf(a); *c = cosf(a); }

// threads per closure workgroup: a property of the LDS variant (FrameLDSx::kThreads) -- 256 for the body-only working set
// (two or three workgroups share a CU), 512 for the full one (142 KB: the workgroup owns its CU, and what it does there is
// stream VPoser weights and adjoint rows through that CU's L2 port: 8 wavefronts keep 108 GB/s in flight, 4 keep 93)
#ifndef SFX_BIG_THREADS
#define SFX_BIG_THREADS 512
#endif
#define SFX_MAX_THREADS 512
// RIF = 2-KiB blend-shape rows in flight per wavefront (forward dots and dfeat adjoint): 4 for the
// body-only variant (33 rows: 16 + 16 + 1), SFX_RIF_BIG for the full model (675 rows)
// debug timing: block 0 / thread 0 stores the shader clock at phase boundaries when D.dbg != NULL
#define MARK(i) do { if (D.dbg && blockIdx.x == 0 && threadIdx.x == 0 && D.dbg[62] <= D.dbg[61]) { D.dbg[i] = clock64(); \
        if ((i) == 0) D.dbg[17] = wall_clock64(); if ((i) == 16) D.dbg[18] = wall_clock64(); } } while (0)

// thread t handles elements t, t + CT, ... of an N-element pass; the constant trip count lets the
// compiler unroll and overlap the (dependent) LDS reads of the iterations
#define FOR_CT(w, N) _Pragma("unroll") for (int w##_i = 0; w##_i < ((N) + CT - 1) / CT; ++w##_i)             \
                         if (const int w = t + w##_i * CT; w < (N))

struct EmptyLDS {};

// Asynchronous global -> LDS copies by the whole workgroup (LDS-DMA, `global_load_lds_dword[x4]`): no registers, and no wait
// until the next __syncthreads (which carries the vmcnt(0)).  Every table a closure workgroup needs at entry -- model
// tables, this frame's record, the forward state of the previous launch -- is requested back to back this way: ONE memory
// round trip instead of one per copy loop (the entry of k_tick_dense was 7.7 us body-only / 11 us with hands + face).
// The destination of one wave-instruction is a contiguous run of 64 dwords (or 64 x 16 bytes) starting at a wave-uniform
// LDS address; sources are per-lane.
typedef __attribute__((address_space(3))) void* sfx_lds_vp;
typedef const __attribute__((address_space(1))) void* sfx_glb_vp;
template <int CT>
__device__ __forceinline__ void lds_fill_async(void* lds_dst, const void* gsrc, const int n_dwords) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const char* g = reinterpret_cast<const char*>(gsrc);
    char* l = reinterpret_cast<char*>(lds_dst);
    for (int base = wv * 64; base < n_dwords; base += CT) {
        const int i = base + lane;
        if (i < n_dwords) __builtin_amdgcn_global_load_lds((sfx_glb_vp)(g + 4 * (size_t)i), (sfx_lds_vp)(l + 4 * base), 4, 0, 0);
    }
}
// LDS-DMA completes in vmcnt order: the barrier that publishes DMA-filled LDS to the other wavefronts must be preceded by a
// wait for this wavefront's own copies.  The compiler emits that s_waitcnt in front of every such s_barrier today; the memory
// model does not oblige it to (a workgroup-scope release needs lgkmcnt only), so it is spelled out (CK: block_sync_lds_direct_load).
__device__ __forceinline__ void lds_dma_wait() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0); expcnt / lgkmcnt untouched
// the same in 16-byte units (both sides 16-byte aligned)
template <int CT>
__device__ __forceinline__ void lds_fill_async16(void* lds_dst, const void* gsrc, const int n16) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const char* g = reinterpret_cast<const char*>(gsrc);
    char* l = reinterpret_cast<char*>(lds_dst);
    for (int base = wv * 64; base < n16; base += CT) {
        const int i = base + lane;
        if (i < n16) __builtin_amdgcn_global_load_lds((sfx_glb_vp)(g + 16 * (size_t)i), (sfx_lds_vp)(l + 16 * base), 16, 0, 0);
    }
}

// Per-frame working set of the closure workgroup.  MAXI = item capacity, VP = VPoser activations
// present.  Two variants are instantiated: <240, true> (any model / configuration, 86 KB) and
// <32, false> (body-only keypoints without VPoser, 47 KB -> three workgroups per CU, or one next
// to two 48-KB GEMM workgroups).
// TH (round 4): threads of the workgroup when not the variant's default -- the body-only set on 512 threads where a workgroup owns
// its CU (k_tick_dense at <= 256 frames): two wavefronts per SIMD cover each other's LDS round trips in the barrier-separated
// passes.  The same bits as on 256 threads: every sum across wavefronts either has its terms in the first four wavefronts on
// both (threads >= 256 hold no keypoint, parameter or item of this variant: they add zeros) or is dealt to kRowWaves = 4
// wavefronts whatever the thread count (the adjoint's row streams), and the layout -- the forward-state blob two launches
// hand each other -- does not depend on TH.
template <int MAXI, bool VP, int TH = 0>
struct __align__(16) FrameLDSx {
    static constexpr int kMaxItems = MAXI;
    static constexpr int kBlocksPerCU = (MAXI <= SFX_SMALL_ITEMS && !VP) ? SFX_SMALL_OCC : 1;    // register budget of the fused kernels
    static constexpr int kThreads = TH ? TH : ((MAXI <= SFX_SMALL_ITEMS && !VP) ? 256 : SFX_BIG_THREADS);
    static constexpr int kRowWaves = (MAXI <= SFX_SMALL_ITEMS && !VP) ? 4 : kThreads / 64;        // wavefronts that stream adjoint rows
    // scratch T: the item transforms, and (reverse sweep) one 512-float partial per row-streaming wavefront
    static constexpr int kScratch = (MAXI * 12 > kRowWaves * 512) ? MAXI * 12 : kRowWaves * 512;
    float feat[SFX_KD_PAD];        // first: read as float4
    float x[SFX_NPAR_MAX];
    float full_pose[168];
    float R[SFX_J * 9];
    float Jr[SFX_J * 3];
    float G[SFX_J * 12];
    float A[SFX_J * 12];
    // the part of the forward whose ROUNDING NOISE decides when the line search stalls is carried in fp64 (see
    // "forward precision" below): skinning transforms and posed kinematic joints; saved with the prefix above
    fwd_t Ad[SFX_J * 12];
    fwd_t Gt[SFX_J * 3 + 3];      // (+3: keeps the saved prefix a multiple of 16 bytes in either precision)
    float vp[MAXI * 3];            // v_posed of the items (template + blend offsets), for the reverse sweep
    float vpo[MAXI * 3];           // the blend offsets alone (small numbers: fp32 sums of them carry ~1e-10 m)
    float vt[MAXI * 3];            // v_template rows of the items
    alignas(16) float T[kScratch]; // item transforms [MAXI][12]; reused as scratch (>= 2048 floats) by the reverse sweep and, between
                                   // evaluations, as the working set of the optimiser tick (float4 accesses)
    fwd_t cd[2 * SFX_J * 12];     // kinematic chain in fp64: the two buffers of the pointer-jumping rounds
    fwd_t Jd[SFX_J * 3];          // rest joints, fp64
    fwd_t jd[SFX_MAX_K * 3];      // mapped joints, fp64 (what the projection reads)
    float dvert[MAXI * 3];
    float dvp[MAXI * 3];
    int   rl[MAXI * 3];            // compacted list of the blend-shape rows the adjoint streams (row = vertex * 3 + coordinate) ...
    float rc[MAXI * 3];            // ... and their coefficients (d v_posed), nonzero entries only
    int   ivid[MAXI];
    int   uslot[MAXI];             // static items: index of the item's vertex in the dense GEMM's export (BatchDev.uvp)
    float iw[MAXI];
    int   wj[MAXI * SFX_NW];       // sparse skinning weights of the items
    float ww[MAXI * SFX_NW];
    int   sjs[SFX_J + 1];          // per-joint lists of the static items (adjoint of the skinning: dA), copied from the
    int   sji[MAXI * SFX_NW];      //   model's CSR when it fits (M.n_sj <= MAXI * SFX_NW): two dependent global round trips
    float sjw[MAXI * SFX_NW];      //   per evaluation otherwise
    static constexpr int kMaxDyn = (MAXI > SFX_SMALL_ITEMS) ? SFX_MAX_DYN : 1;
    int   djs[SFX_J + 1];          // the same for the dynamic-contour items of this frame's LUT row (DevModel.dynp_*)
    int   dji[kMaxDyn * SFX_NW];
    float djw[kMaxDyn * SFX_NW];
    float joints[SFX_MAX_K * 3];
    float dj[SFX_MAX_K * 3];
    float dA[SFX_J * 12];
    float dG[SFX_J * 12];
    float drel[SFX_J * 3];
    float dJ[SFX_J * 3];
    float dR[SFX_J * 9];
    float dfeat[SFX_KD_PAD];
    float dpose[168];
    float gc[SFX_NPAR_MAX];
    float red[SFX_MAX_THREADS];
    float lh45[SFX_NHAND], rh45[SFX_NHAND];    // } contiguous: saved / reloaded as one run of 2 * SFX_NHAND + 1 dwords
    int   lut_row;                             // }
    typename std::conditional<VP, VposerLDS, EmptyLDS>::type V;   // VPoser activations (use_vposer only)
    alignas(16) float fd[FD_N]; // this frame's keypoints / weights / camera / regression pose (image of BatchDev.fd)
    alignas(16) int meta[SFX_META_N];     // tree / joint-map tables (one coalesced load instead of
                                // dependent global loads inside every level of the chain)
};
using FrameLDS = FrameLDSx<SFX_MAX_ITEMS, true>;
using FrameLDSSmall = FrameLDSx<SFX_SMALL_ITEMS, false>;
using FrameLDSSmall8 = FrameLDSx<SFX_SMALL_ITEMS, false, 512>;      // the same set on eight wavefronts (k_tick_dense, a workgroup per CU)

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }
template <int CTRL>
__device__ __forceinline__ float lb_quad(float x) {       // DPP quad permutation of x (all lanes of the quad active)
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}

// fixed-order block reduction: DPP sum per wavefront, then the CT/64 partials in order
// (2 barriers instead of a 9-barrier LDS tree); result in all threads
template <int CT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    const float w = wave_sum_dpp(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = w;
    __syncthreads();
    float r = red[0];
#pragma unroll
    for (int i = 1; i < CT / 64; ++i) r += red[i];
    __syncthreads();
    return r;
}

// smplx.lbs.batch_rodrigues: angle = ||theta + 1e-8||, R = I + sin K + (1-cos) K K
__device__ __forceinline__ void rodrigues_fwd(const float* th, float* R) {
    const float ex = th[0] + 1e-8f, ey = th[1] + 1e-8f, ez = th[2] + 1e-8f;
    const float a = sqrtf(ex * ex + ey * ey + ez * ez);
    const float dx = th[0] / a, dy = th[1] / a, dz = th[2] / a;
    const float s = sinf(a), c = cosf(a);
    const float K[9] = {0.f, -dz, dy, dz, 0.f, -dx, -dy, dx, 0.f};
    const float omc = 1.f - c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float kk = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
            R[i * 3 + j] = ((i == j) ? 1.f : 0.f) + s * K[i * 3 + j] + omc * kk;
        }
}

// the same in fp64 (forward precision, see closure_body): R row-major
__device__ __forceinline__ void rodrigues_fwd_d(const float* th, fwd_t* R) {
    const fwd_t t0 = th[0], t1 = th[1], t2 = th[2];
    const fwd_t ex = t0 + (fwd_t)1e-8, ey = t1 + (fwd_t)1e-8, ez = t2 + (fwd_t)1e-8;
    const fwd_t a = sqrt(ex * ex + ey * ey + ez * ez);
    const fwd_t dx = t0 / a, dy = t1 / a, dz = t2 / a;
    fwd_t s, c;
    sincos_t(a, &s, &c);
    const fwd_t K[9] = {0, -dz, dy, dz, 0, -dx, -dy, dx, 0};
    const fwd_t omc = (fwd_t)1 - c;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const fwd_t kk = K[i * 3 + 0] * K[0 * 3 + j] + K[i * 3 + 1] * K[1 * 3 + j] + K[i * 3 + 2] * K[2 * 3 + j];
            R[i * 3 + j] = ((i == j) ? (fwd_t)1 : (fwd_t)0) + s * K[i * 3 + j] + omc * kk;
        }
}

// reverse of rodrigues_fwd: dth += J^T dR
__device__ __forceinline__ void rodrigues_bwd(const float* th, const float* dR, float* dth) {
    const float ex = th[0] + 1e-8f, ey = th[1] + 1e-8f, ez = th[2] + 1e-8f;
    const float a = sqrtf(ex * ex + ey * ey + ez * ez);
    const float inv = 1.f / a;
    const float d[3] = {th[0] * inv, th[1] * inv, th[2] * inv};
    const float s = sinf(a), c = cosf(a), omc = 1.f - c;
    const float K[9] = {0.f, -d[2], d[1], d[2], 0.f, -d[0], -d[1], d[0], 0.f};
    float KK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            KK[i * 3 + j] = K[i * 3 + 0] * K[j] + K[i * 3 + 1] * K[3 + j] + K[i * 3 + 2] * K[6 + j];
    float dRK = 0.f, dRKK = 0.f;
#pragma unroll
    for (int e = 0; e < 9; ++e) { dRK += dR[e] * K[e]; dRKK += dR[e] * KK[e]; }
    float da = c * dRK + s * dRKK;
    // dK = s dR + (1-c) (dR K^T + K^T dR)
    float dK[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                t1 += dR[i * 3 + k] * K[j * 3 + k];     // dR K^T
                t2 += K[k * 3 + i] * dR[k * 3 + j];     // K^T dR
            }
            dK[i * 3 + j] = s * dR[i * 3 + j] + omc * (t1 + t2);
        }
    const float dd[3] = {dK[7] - dK[5], dK[2] - dK[6], dK[3] - dK[1]};
    const float ddth = dd[0] * th[0] + dd[1] * th[1] + dd[2] * th[2];
    da -= ddth * inv * inv;
    dth[0] += dd[0] * inv + da * ex * inv;
    dth[1] += dd[1] * inv + da * ey * inv;
    dth[2] += dd[2] * inv + da * ez * inv;
}

__device__ __forceinline__ float gmof_grad(float r, float rho2) {
    // d/dr [ rho^2 r^2 / (r^2 + rho^2) ] = 2 r rho^4 / (r^2 + rho^2)^2
    const float den = r * r + rho2;
    return 2.f * r * (rho2 / den) * (rho2 / den);
}

// One closure evaluation of frame b by the whole workgroup (CT threads).  When `gflat` is not
// NULL the flat gradient / loss are ALSO left in LDS (gflat[NVAR_MAX], *fout) for a consumer in
// the same workgroup (fused kernels).
template <class LDS>
__device__ __forceinline__ void closure_body(LDS& S, const DevModel& M, const BatchDev& D,
                                             const VarList* __restrict__ vls, const StageW* __restrict__ sws,
                                             const ClosureArgs& args, const int b, float* gflat, float* fout) {
    constexpr int CT = LDS::kThreads;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const ParLayout& L = D.L;
    const BatchCfgDev& C = D.cfg;

    const int stage = __builtin_amdgcn_readfirstlane((args.stage_override != -2) ? args.stage_override : D.stage[b]);
    if (stage >= C.n_stages && !args.forward_only) return;      // frame finished
    const bool cam_stage = (stage < 0);
    const StageW sw = (cam_stage || args.forward_only) ? StageW{} : sws[stage];     // (11 scalars, requested here: the loss section is 20 k cycles away)

    // Live keypoints of this evaluation.  Keypoints are ordered body | hands | face (+ contour) and the vertex items follow
    // that order; a stage whose hand / face joint weight is zero (fit_single_frame.py:569-572: the first three of the five
    // stages of fit_smplx_smplifyx.yaml) multiplies everything those keypoints produce by zero -- loss terms and gradients
    // alike -- so their vertex rows, skinning, projection and adjoint lists are not walked at all (exact: the skipped terms
    // are zeros).  Forward-only passes and the camera stage keep every keypoint.
    const int cls = (cam_stage || args.forward_only) ? 2 : ((sw.face_jw != 0.f) ? 2 : ((sw.hand_jw != 0.f) ? 1 : 0));
    const int KL = __builtin_amdgcn_readfirstlane(cls == 2 ? M.K : C.kl[cls]);             // keypoints < KL are live
    const int NIL = __builtin_amdgcn_readfirstlane(cls == 2 ? M.n_items : C.nil[cls]);     // items < NIL belong to them

    MARK(0);
    // ------------------------------------------------------------------ load parameters
    // reuse_fwd: the export pass of the previous launch left this frame's forward state (feat, x,
    // full_pose, R, Jr, G, A -- the leading members of FrameLDS -- plus hand poses, LUT row and
    // the VPoser activations) in D.fwd; reload it instead of recomputing pose assembly,
    // Rodrigues, joint regression and the kinematic chain
    constexpr bool HAS_VP = !std::is_same<decltype(S.V), EmptyLDS>::value;
    constexpr int RIF = (LDS::kMaxItems <= SFX_SMALL_ITEMS) ? SFX_RIF_SMALL : SFX_RIF_BIG;
    constexpr int FWD_PREFIX = (int)(offsetof(LDS, vp) / sizeof(float));
    static_assert(FWD_PREFIX % 4 == 0 && FWD_PREFIX + 96 + 2 * VP_H + 128 + 64 <= SFX_FWD_N, "forward-state blob layout");
    static_assert(offsetof(LDS, Ad) % sizeof(fwd_t) == 0 && offsetof(LDS, cd) % sizeof(fwd_t) == 0, "forward-precision members");
    const bool reuse = args.reuse_fwd != 0;
    float* fwd = D.fwd ? D.fwd + (size_t)b * SFX_FWD_N : nullptr;
    const float* xsrc = (args.from_X ? D.X : D.Xt) + (size_t)b * SFX_NPAR_MAX;
    // every global source of this section goes to LDS asynchronously (lds_fill_async): one round trip for all of it
    if (!reuse) {
        if (args.x_lds) { for (int i = t; i < L.npar; i += CT) S.x[i] = args.x_lds[i]; }      // (published by the barrier below)
        else lds_fill_async<CT>(S.x, xsrc, L.npar);
    }
    if (!args.keep_tables) {   // (a persistent workgroup keeps the tables and its frame's data in LDS between evaluations)
    static_assert(SFX_META_N % 4 == 0 && FD_N % 4 == 0 && offsetof(LDS, meta) % 16 == 0 && offsetof(LDS, fd) % 16 == 0, "16-byte copies");
    lds_fill_async16<CT>(S.meta, M.meta, SFX_META_N / 4);
    lds_fill_async16<CT>(S.fd, D.fd + (size_t)b * FD_N, FD_N / 4);     // per-frame record (packed by k_pack_fd)
    // static items (vertex joints, static landmarks): vertex ids, weights, template rows, skinning weights, per-joint
    // adjoint lists -- none of it depends on the pose, so it is fetched here, next to the other tables (a persistent
    // workgroup keeps it for the whole fit), not item by item inside the evaluation
    const int ns = M.n_static_items;
    lds_fill_async<CT>(S.ivid, M.item_vid, ns); lds_fill_async<CT>(S.iw, M.item_w, ns); lds_fill_async<CT>(S.uslot, M.item_uslot, ns);
    lds_fill_async<CT>(S.vt, M.item_vt, ns * 3);
    lds_fill_async<CT>(S.wj, M.item_wj, ns * SFX_NW); lds_fill_async<CT>(S.ww, M.item_ww, ns * SFX_NW);
    if (M.n_sj <= LDS::kMaxItems * SFX_NW) {
        lds_fill_async<CT>(S.sjs, M.sj_start, SFX_J + 1);
        lds_fill_async<CT>(S.sji, M.sj_item, M.n_sj); lds_fill_async<CT>(S.sjw, M.sj_w, M.n_sj);
    }
    }
    if (reuse) {
        static_assert(offsetof(LDS, lut_row) == offsetof(LDS, lh45) + 2 * SFX_NHAND * sizeof(float), "hand poses + LUT row are one run");
        lds_fill_async16<CT>(&S, fwd, FWD_PREFIX / 4);
        const float* ex = fwd + FWD_PREFIX;
        lds_fill_async<CT>(S.lh45, ex, 2 * SFX_NHAND + 1);
        if constexpr (HAS_VP) if (C.use_vposer) {
            const float* vx = ex + 96;
            static_assert(offsetof(VposerLDS, h2) == VP_H * sizeof(float) && offsetof(VposerLDS, o) == 2 * VP_H * sizeof(float), "h1 | h2 | o are one run");
            lds_fill_async16<CT>(S.V.h1, vx, (2 * VP_H + 128) / 4);
            lds_fill_async16<CT>(S.V.body, vx + 2 * VP_H + 128, 64 / 4);
        }
    }
    for (int i = t; i < SFX_KD_PAD; i += CT) { if (!reuse) S.feat[i] = 0.f; S.dfeat[i] = 0.f; }
    for (int i = t; i < SFX_NPAR_MAX; i += CT) S.gc[i] = 0.f;
    for (int i = t; i < 168; i += CT) S.dpose[i] = 0.f;
    lds_dma_wait();
    __syncthreads();
    const float* bodypose = S.x + L.emb;
    if constexpr (HAS_VP) {
        if (C.use_vposer && !reuse) {          // body_pose = vposer.decode(pose_embedding) (fitting.py:236-238)
            vposer_forward<CT>(S.V, M, S.x + L.emb, S.T);
            if (t < 63) D.bodypose[(size_t)b * 63 + t] = S.V.body[t];
        }
        if (C.use_vposer) bodypose = S.V.body;
    }
  if (!reuse) {

    MARK(1);
    // ------------------------------------------------------------------ pose assembly
    if (t < SFX_POSE) {
        float v;
        if (t < 3) v = S.x[L.go + t];
        else if (t < 66) v = bodypose[t - 3];
        else if (t < 69) v = S.x[L.jaw + t - 66];
        else if (t < 72) v = S.x[L.leye + t - 69];
        else if (t < 75) v = S.x[L.reye + t - 72];
        else {
            const bool left = t < 120;
            const int c = left ? t - 75 : t - 120;
            const float* comp = left ? M.comp_l : M.comp_r;
            const float* pc = S.x + (left ? L.lh : L.rh);
            v = 0.f;
            for (int i = 0; i < L.NPCA; ++i) v += pc[i] * comp[i * SFX_NHAND + c];
            if (left) S.lh45[c] = v; else S.rh45[c] = v;
        }
        S.full_pose[t] = v + M.pose_mean[t];
    }
    if (t < M.S) S.feat[t] = (t < L.NB) ? S.x[L.betas + t] : S.x[L.expr + t - L.NB];
    __syncthreads();

    MARK(2);
    // ------------------------------------------------------------------ Rodrigues, rest joints
    // Forward precision.  Near the optimum the gradient is the small difference of keypoint forces of
    // ~5e3 per metre each, and d(force)/d(joint position) ~ 1e6 per metre: a joint position off by 1e-7 m
    // (fp32 rounding of a carelessly ordered chain / skinning) moves the gradient by 0.1 while |g| ~ 4 -- the line search
    // stalls on that noise and run_fitting's ftol test (fitting.py:185-189) ends the stage (measured:
    // tests/probe_drift.py).  Rotations, rest joints, the kinematic chain and the skinning of the keypoint vertices are
    // therefore evaluated in fwd_t = fp64 (a few thousand flops per evaluation): whatever this implementation's
    // summation orders are, the joints carry less rounding noise than torch fp32's.  The joints are then rounded to
    // fp32 and projected in fp32 like the reference's camera (proj_t): that rounding is the noise floor the reference
    // itself has, and with it the fits stop where the reference's fp32 fits stop.  Parameters, blend-shape sums, losses
    // and the whole reverse sweep are fp32, as in the reference.
    if (t < SFX_J) {
        fwd_t R[9];
        rodrigues_fwd_d(&S.full_pose[3 * t], R);
        fwd_t* src0 = (M.n_rounds & 1) ? (S.cd + SFX_J * 12) : S.cd;
#pragma unroll
        for (int e = 0; e < 9; ++e) { S.R[t * 9 + e] = (float)R[e]; src0[t * 12 + (e / 3) * 4 + e % 3] = R[e]; }
        if (t > 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                S.feat[M.S + 9 * (t - 1) + e] = (float)(R[e] - ((e == 0 || e == 4 || e == 8) ? (fwd_t)1 : (fwd_t)0));
        }
    } else if (t >= 64 && t < 64 + SFX_J * 3) {
        const int i = t - 64;
        fwd_t v = M.J_template[i];
        const float* jd = M.J_dirs + (size_t)i * M.S;
        for (int l = 0; l < M.S; ++l) v += (fwd_t)jd[l] * (fwd_t)S.feat[l];
        S.Jd[i] = v;
        S.Jr[i] = (float)v;
    }
    __syncthreads();

    MARK(3);
    // ------------------------------------------------------------------ kinematic chain
    // pointer jumping instead of a level-by-level walk: start from the local transforms
    // T_j = [R_j | J_j - J_parent]; in round k every joint composes with the transform of its
    // 2^k-th ancestor, T_j <- T_anc o T_j, so after ceil(log2(depth)) rounds T_j = G_j.  Two
    // buffers alternate; the result lands in S.cd[0 ..).  4 barriers instead of 11.
    {
        fwd_t* src = (M.n_rounds & 1) ? (S.cd + SFX_J * 12) : S.cd;
        fwd_t* dst = (M.n_rounds & 1) ? S.cd : (S.cd + SFX_J * 12);
        if (t < SFX_J * 3) {
            const int j = t / 3, r = t % 3;
            const int p = S.meta[MO_PAR + j];
            src[j * 12 + r * 4 + 3] = S.Jd[t] - (p < 0 ? (fwd_t)0 : S.Jd[p * 3 + r]);
        }
        __syncthreads();
        MARK(27);
        for (int k = 0; k < M.n_rounds; ++k) {
            FOR_CT(w, SFX_J * 12) {
                const int j = w / 12, e = w % 12, r = e >> 2, c = e & 3;
                const int a = S.meta[MO_ANC + k * 56 + j];
                fwd_t v = src[w];
                if (a >= 0) {
                    const fwd_t* Ta = &src[a * 12 + r * 4];
                    const fwd_t* Tb = &src[j * 12 + c];
                    v = Ta[0] * Tb[0] + Ta[1] * Tb[4] + Ta[2] * Tb[8];
                    if (c == 3) v += Ta[3];
                }
                dst[w] = v;
            }
            __syncthreads();
            fwd_t* tmp = src; src = dst; dst = tmp;
        }
    }
    MARK(28);
    if (t < SFX_J) {
        const fwd_t* Gj = &S.cd[t * 12];
        const fwd_t* Jj = &S.Jd[t * 3];
        fwd_t* Adj = &S.Ad[t * 12];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const fwd_t at = Gj[r * 4 + 3] - (Gj[r * 4 + 0] * Jj[0] + Gj[r * 4 + 1] * Jj[1] + Gj[r * 4 + 2] * Jj[2]);
            Adj[r * 4 + 0] = Gj[r * 4 + 0]; Adj[r * 4 + 1] = Gj[r * 4 + 1]; Adj[r * 4 + 2] = Gj[r * 4 + 2]; Adj[r * 4 + 3] = at;
            S.Gt[t * 3 + r] = Gj[r * 4 + 3];
#pragma unroll
            for (int c = 0; c < 4; ++c) S.G[t * 12 + r * 4 + c] = (float)Gj[r * 4 + c];
            S.A[t * 12 + r * 4 + 0] = (float)Gj[r * 4 + 0]; S.A[t * 12 + r * 4 + 1] = (float)Gj[r * 4 + 1];
            S.A[t * 12 + r * 4 + 2] = (float)Gj[r * 4 + 2]; S.A[t * 12 + r * 4 + 3] = (float)at;
        }
    }
    // dynamic-contour LUT row (smplx find_dynamic_lmk_idx_and_bcoords; no gradient)
    if (t == CT - 1) {
        int row = 0;
        if (M.n_dyn > 0) {
            float rel[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
            const int chain[5] = {12, 9, 6, 3, 0};
            for (int q = 0; q < 5; ++q) {
                const float* Rk = &S.R[chain[q] * 9];
                float o[9];
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j)
                        o[i * 3 + j] = Rk[i * 3] * rel[j] + Rk[i * 3 + 1] * rel[3 + j] + Rk[i * 3 + 2] * rel[6 + j];
                for (int e = 0; e < 9; ++e) rel[e] = o[e];
            }
            const float sy = sqrtf(rel[0] * rel[0] + rel[3] * rel[3]);
            const float ang = atan2f(-rel[6], sy);
            float deg = (-ang) * 180.0f / 3.14159265358979323846f;
            deg = fminf(deg, 39.f);
            const int y = (int)rintf(deg);
            row = (y < -39) ? 78 : ((y < 0) ? (39 - y) : y);
        }
        S.lut_row = row;
    }
    __syncthreads();

  }   // !reuse
    MARK(4);
    // ------------------------------------------------------------------ dense export
    if (args.export_dense) {
        const int slot = D.slot[b];      // column of this frame in the GEMM operands (compacted)
        // coefficients of this frame: one contiguous 2-KiB row (entries >= KD are zero).  (As a COLUMN of a [K][frames]
        // matrix -- what the GEMM's A operand looks like in LDS -- these were 506 scattered 4-byte writes per frame into
        // lines shared by 32 frames: 10 k cycles of the launch; the GEMM transposes while staging instead.)
        if (t < SFX_KD_PAD / 4) reinterpret_cast<float4*>(D.featR + (size_t)slot * SFX_KD_PAD)[t] = reinterpret_cast<const float4*>(S.feat)[t];
        MARK(17);
        for (int i = t; i < SFX_J * 12; i += CT) {
            const int j = i / 12, e = i % 12;
            D.AT[((size_t)e * SFX_JPAD + j) * D.Bpad + slot] = S.A[i];
        }
        MARK(18);
        if (args.forward_only == 2) {           // export pass only: keep the forward state for the adjoint pass
            if (fwd) {
                const float* sp = reinterpret_cast<const float*>(&S);
                for (int i = t; i < FWD_PREFIX / 4; i += CT)
                    reinterpret_cast<float4*>(fwd)[i] = reinterpret_cast<const float4*>(sp)[i];
                float* ex = fwd + FWD_PREFIX;
                if (t < SFX_NHAND) { ex[t] = S.lh45[t]; ex[SFX_NHAND + t] = S.rh45[t]; }
                if (t == 64) ex[2 * SFX_NHAND] = __int_as_float(S.lut_row);
                if constexpr (HAS_VP) if (C.use_vposer) {
                    float* vx = ex + 96;
                    for (int i = t; i < VP_H; i += CT) { vx[i] = S.V.h1[i]; vx[VP_H + i] = S.V.h2[i]; }
                    if (t < 128) vx[2 * VP_H + t] = S.V.o[t];
                    if (t < 64) vx[2 * VP_H + 128 + t] = S.V.body[t];
                }
            }
            MARK(19);
            return;
        }
    }

    MARK(5);
    // ------------------------------------------------------------------ needed vertices
    const int NI = NIL;
    // dynamic contour items: their vertices follow the head pose -- everything that depends on the LUT row (vertex ids,
    // barycentric weights, template rows, sparse skinning weights, per-joint adjoint lists) is one block of DevModel.dynp_*
    const bool dyn_live = NI > M.n_static_items;
    const bool dyn_lds = dyn_live && M.dynp_js != nullptr && M.n_dyn_items <= LDS::kMaxDyn;
    if (dyn_live) {
        const int ns = M.n_static_items, nd = M.n_dyn_items;
        const size_t ro = (size_t)S.lut_row * nd;
        lds_fill_async<CT>(S.ivid + ns, M.dynp_vid + ro, nd); lds_fill_async<CT>(S.iw + ns, M.dynp_w + ro, nd);
        lds_fill_async<CT>(S.vt + ns * 3, M.dynp_vt + ro * 3, nd * 3);
        if (M.dynp_us) lds_fill_async<CT>(S.uslot + ns, M.dynp_us + ro, nd);
        lds_fill_async<CT>(S.wj + ns * SFX_NW, M.dynp_wj + ro * SFX_NW, nd * SFX_NW);
        lds_fill_async<CT>(S.ww + ns * SFX_NW, M.dynp_ww + ro * SFX_NW, nd * SFX_NW);
        if (dyn_lds) {
            lds_fill_async<CT>(S.djs, M.dynp_js + (size_t)S.lut_row * (SFX_J + 1), SFX_J + 1);
            lds_fill_async<CT>(S.dji, M.dynp_ji + ro * SFX_NW, nd * SFX_NW); lds_fill_async<CT>(S.djw, M.dynp_jw + ro * SFX_NW, nd * SFX_NW);
        }
        lds_dma_wait();
        __syncthreads();
    }
    // v_posed rows and skinning transforms of `ni` vertices listed in S.ivid (the model's items, or a
    // chunk of vertices that carry a penetration gradient)
    auto items_forward = [&](const int ib, const int ni) {       // items ib .. ib + ni - 1
    // v_posed rows: one wavefront per (item, coord) dot product of length KD_PAD
    {
        const float4* f4 = reinterpret_cast<const float4*>(S.feat);
        const float4 fa = f4[lane], fb = f4[64 + lane];
        // RIF rows per wavefront per pass: 2 RIF independent 1-KiB loads in flight before the reductions.
        // All frames need the SAME rows (static items), and workgroups launched together walk them
        // in step: every CU of an XCD then asks one L2 channel for one 2-KiB row at the same moment.
        // Each workgroup therefore starts at its own rotation of the row list; the rows are
        // independent dot products, so the result is bit-identical.
        // Row addresses are wave-uniform: lane u looks up row u of the pass (one LDS read for the whole pass), the
        // row index travels through v_readlane into scalar registers and the loads use scalar base + lane offset --
        // looked up one by one in front of each load, the passes were issue-bound (2 dependent LDS round trips per row).
        const int nrow = ni * 3, rb = ib * 3;
        const int rot = (int)((blockIdx.x * 40u) % (unsigned)nrow);
        for (int w0 = wv * RIF; w0 < nrow; w0 += (CT / 64) * RIF) {
            float4 da[RIF], db[RIF];
            int myw, myrow;
            {
                const int u = lane < RIF ? lane : 0;
                int w = ((w0 + u < nrow) ? w0 + u : w0) + rot;
                w = w >= nrow ? w - nrow : w;
                w += rb;
                myw = w;
                myrow = S.ivid[w / 3] * 3 + w % 3;
            }
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const int r = __builtin_amdgcn_readlane(myrow, u);
                const float4* row = reinterpret_cast<const float4*>(M.dirsT + (size_t)r * SFX_KD_PAD);
                da[u] = row[lane]; db[u] = row[64 + lane];
            }
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const int w = __builtin_amdgcn_readlane(myw, u);
                float acc = fa.x * da[u].x + fa.y * da[u].y + fa.z * da[u].z + fa.w * da[u].w +
                            fb.x * db[u].x + fb.y * db[u].y + fb.z * db[u].z + fb.w * db[u].w;
                acc = wave_sum(acc);
                if (lane == 0 && w0 + u < nrow) { S.vpo[w] = acc; S.vp[w] = S.vt[w] + acc; }
            }
        }
    }
    };
    // skinning transforms of the items (fp32 copy for the reverse sweep)
    // (sparse rows of lbs_weights: <= SFX_NW nonzeros per vertex, ascending joint order, so the
    //  sum visits the same nonzero terms in the same order as the dense product)
    auto items_transforms = [&](const int ib, const int ni) {
    // (sparse weights: the static items' came with the tables, the dynamic items' with their LUT-row block)
    for (int w = t + ib * 12; w < (ib + ni) * 12; w += CT) {
        const int i = w / 12, e = w % 12;
        float acc = 0.f;
        if (S.wj[i * SFX_NW] >= 0) {
#pragma unroll
            for (int q2 = 0; q2 < SFX_NW; ++q2) {
                const float wq = S.ww[i * SFX_NW + q2];
                if (wq != 0.f) acc += wq * S.A[S.wj[i * SFX_NW + q2] * 12 + e];
            }
        } else {                    // more than SFX_NW nonzero weights: full row of lbs_weights
            const float* Wv = M.W + (size_t)S.ivid[i] * SFX_J;
            for (int j = 0; j < SFX_J; ++j) { const float wq = Wv[j]; if (wq != 0.f) acc += wq * S.A[j * 12 + e]; }
        }
        S.T[w] = acc;
    }
    __syncthreads();
    };
    if (args.use_dense_verts && !args.from_scratch_items) {
        // dense path: the GEMM that produced the vertices also left the blend offsets (v_posed - v_template) of every item
        // vertex (lbs_dense.hip epilogue) -- of the static items and of every vertex a dynamic-contour item can land on,
        // whichever LUT row this frame's head pose selects: no blend-shape row is streamed forward here
        const size_t ub = (size_t)D.slot[b] * M.n_uniq;
        const int nst = (M.dynp_us || NI < M.n_static_items) ? NI : M.n_static_items;
        for (int w = t; w < nst * 3; w += CT) {
            const float o = D.uvp[(ub + S.uslot[w / 3]) * 3 + w % 3];
            S.vpo[w] = o; S.vp[w] = S.vt[w] + o;
        }
        if (NI > nst) items_forward(nst, NI - nst);
    } else items_forward(0, NI);
    items_transforms(0, NI);
    MARK(6);

    MARK(7);
    // ------------------------------------------------------------------ mapped joints
    // keypoint vertices: sum_j w_j (A_j [v_posed; 1]) in fp64 on the fp64 transforms (the dense kernel's fp32
    // vertex of the same index serves the interpenetration term and the output mesh, not the keypoints)
    const int K = M.K;
    auto item_vertex = [&](const int i, const int r) -> fwd_t {
        const fwd_t vx = (fwd_t)S.vt[i * 3] + (fwd_t)S.vpo[i * 3], vy = (fwd_t)S.vt[i * 3 + 1] + (fwd_t)S.vpo[i * 3 + 1],
                     vz = (fwd_t)S.vt[i * 3 + 2] + (fwd_t)S.vpo[i * 3 + 2];
        fwd_t acc = 0;
        if (S.wj[i * SFX_NW] >= 0) {
#pragma unroll
            for (int q2 = 0; q2 < SFX_NW; ++q2) {
                const float wq = S.ww[i * SFX_NW + q2];
                if (wq != 0.f) { const fwd_t* Aq = &S.Ad[S.wj[i * SFX_NW + q2] * 12 + r * 4];
                                 acc += (fwd_t)wq * (Aq[0] * vx + Aq[1] * vy + Aq[2] * vz + Aq[3]); }
            }
        } else {
            const float* Wv = M.W + (size_t)S.ivid[i] * SFX_J;
            for (int j = 0; j < SFX_J; ++j) { const float wq = Wv[j];
                if (wq != 0.f) { const fwd_t* Aq = &S.Ad[j * 12 + r * 4]; acc += (fwd_t)wq * (Aq[0] * vx + Aq[1] * vy + Aq[2] * vz + Aq[3]); } }
        }
        return acc;
    };
    for (int w = t; w < K * 3; w += CT) {
        const int k = w / 3, r = w % 3;
        fwd_t v;
        if (k >= KL) v = 0;                      // (not live in this stage: never read)
        else if (S.meta[MO_JT + k] == 0) v = S.Gt[S.meta[MO_JS + k] * 3 + r];
        else {
            v = 0;
            const int i0 = S.meta[MO_JI0 + k], n = S.meta[MO_JN + k];
            if (n == 1 && S.iw[i0] == 1.f) v = item_vertex(i0, r);
            else for (int i = 0; i < n; ++i) v += item_vertex(i0 + i, r) * (fwd_t)S.iw[i0 + i];
        }
        S.jd[w] = v;
        S.joints[w] = (float)v;
    }
    __syncthreads();
    if (args.forward_only) {
        if (D.joints) for (int w = t; w < K * 3; w += CT) D.joints[(size_t)b * K * 3 + w] = S.joints[w];
        if (D.fullpose) for (int w = t; w < SFX_POSE; w += CT) D.fullpose[(size_t)b * SFX_POSE + w] = S.full_pose[w];
        return;
    }

    MARK(8);
    // ------------------------------------------------------------------ losses
    // per-frame data was prefetched into LDS (S.fd) at kernel entry; all sums of this section go
    // through ONE fixed-order reduction (DPP per wavefront, then 4 partials): 2 barriers in total
    const float* fd = S.fd;
    const float fx = fd[FD_CAM + 0], fy = fd[FD_CAM + 1], cx = fd[FD_CAM + 2], cy = fd[FD_CAM + 3];
    const float dwt = fd[FD_CAM + 4], est_tz = fd[FD_CAM + 5];
    const float* Rc = fd + FD_CAMR;
    const float* ct = S.x + L.cam_t;
    const float dw2 = dwt * dwt;
    const float rho2 = C.rho * C.rho;

    MARK(20);
    float csum = 1.f;
    if (cam_stage && C.use_conf_cam) {
        float p = 0.f;
        if (t < K) { const float cm = fd[FD_CMASK + t]; const float cf = fd[FD_CONF + t]; p = (cm != 0.f) ? cf * cf : 0.f; }
        csum = block_sum<CT>(p, S.red);
    }
    enum { Q_L = 0, Q_D0, Q_D1, Q_D2, Q_PP, Q_SH, Q_ANG, Q_LH, Q_RH, Q_EX, Q_JW, NQ };
    float q[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) q[i] = 0.f;
    if (t >= KL && t < K) { S.dj[t * 3 + 0] = 0.f; S.dj[t * 3 + 1] = 0.f; S.dj[t * 3 + 2] = 0.f; }
    if (t < KL) {
        // camera.py:93-117 in proj_t (fp32, as the reference; see "forward precision" above) -- except in the camera
        // stage, always fp64: there the residuals are tens of pixels, so fp32 pixel rounding (3e-5 px) is +-3e-3 of loss
        // noise, enough to make a strong-Wolfe zoom on a badly scaled direction collapse to t = 0 twice and end the stage
        // on the ftol test far from the minimum (1 of 96 reference frames, tests/golden/e2e_vposer_set.npz frame 14 --
        // the reference's fp32 run is exposed to the same coin toss).  Every later stage inherits this stage's camera;
        // its ~45 evaluations of K = 4 joints cost nothing.
        float pcx, pcy, pcz, rx, ry;
        if (cam_stage) project_residual<double>(&S.jd[t * 3], Rc, ct, fx, fy, cx, cy, fd[FD_GT + 2 * t], fd[FD_GT + 2 * t + 1], pcx, pcy, pcz, rx, ry);
        else if (C.proj64) project_residual<double>(&S.jd[t * 3], Rc, ct, fx, fy, cx, cy, fd[FD_GT + 2 * t], fd[FD_GT + 2 * t + 1], pcx, pcy, pcz, rx, ry);      // (cfg float_dtype: float64)
        else project_residual<proj_t>(&S.jd[t * 3], Rc, ct, fx, fy, cx, cy, fd[FD_GT + 2 * t], fd[FD_GT + 2 * t + 1], pcx, pcy, pcz, rx, ry);
        float du, dv;       // dL/du, dL/dv
        if (cam_stage) {
            if (fd[FD_CMASK + t] != 0.f) {
                q[Q_L] = rx * rx + ry * ry;
                du = -2.f * rx * csum * dw2; dv = -2.f * ry * csum * dw2;
            } else { du = 0.f; dv = 0.f; }
        } else {
            float w = fd[FD_JW + t];
            if (t >= C.nbj) w = (t < C.nbj + 42) ? ((w != 0.f) ? sw.hand_jw : 0.f) : ((w != 0.f) ? sw.face_jw : 0.f);
            if (C.use_conf) w *= fd[FD_CONF + t];
            const float w2 = w * w;
            if (w2 != 0.f) {
                const float sx = rx * rx, sy = ry * ry;
                const float gmx = rho2 * (sx / (sx + rho2)), gmy = rho2 * (sy / (sy + rho2));
                q[Q_L] = w2 * gmx + w2 * gmy;
                du = -(w2 * dw2) * gmof_grad(rx, rho2);
                dv = -(w2 * dw2) * gmof_grad(ry, rho2);
            } else { du = 0.f; dv = 0.f; }
        }
        const float dix = du * fx, diy = dv * fy;
        const float d0 = dix / pcz, d1 = diy / pcz, d2 = -(dix * pcx + diy * pcy) / (pcz * pcz);
        q[Q_D0] = d0; q[Q_D1] = d1; q[Q_D2] = d2;
        S.dj[t * 3 + 0] = Rc[0] * d0 + Rc[3] * d1 + Rc[6] * d2;
        S.dj[t * 3 + 1] = Rc[1] * d0 + Rc[4] * d1 + Rc[7] * d2;
        S.dj[t * 3 + 2] = Rc[2] * d0 + Rc[5] * d1 + Rc[8] * d2;
    }
    MARK(21);
    const float bpw2 = sw.bpw * sw.bpw, sw2 = sw.sw * sw.sw, h2 = sw.hpw * sw.hpw, e2 = sw.epw * sw.epw;
    if (!cam_stage) {
        const bool latent_reg = C.use_vposer ? (stage + 1 == C.n_stages && C.has_reg) : (C.has_reg != 0);
        const bool gmm = D.gmm_M > 0 && !C.use_vposer && !C.has_reg;
        if (gmm) {
            // MaxMixturePrior.merged_log_likelihood (prior.py:174-187) on body_pose = the embedding:
            //   min_m [ 0.5 (x - mu_m)^T P_m (x - mu_m) - log nll_weights_m ],  gradient through the minimum's component
            // (P symmetrised on the host: autograd's 0.5 (P + P^T) d).  Wavefront wv takes components wv, wv + 4;
            // lane i forms (P d)_i from column i (coalesced), the quadratic form is a DPP sum.
            float gy[2] = {0.f, 0.f};
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const int m = wv + mi * (CT / 64);
                float contrib = 0.f;
                if (m < D.gmm_M && lane < L.NEMB) {
                    const float* mu = D.gmm_mean + m * 64;
                    const float* Pm = D.gmm_prec + (size_t)m * 64 * 64;
                    float y = 0.f;
                    for (int j = 0; j < L.NEMB; ++j) y += Pm[j * 64 + lane] * (S.x[L.emb + j] - mu[j]);
                    gy[mi] = y;
                    contrib = y * (S.x[L.emb + lane] - mu[lane]);
                }
                const float quad = wave_sum_dpp(contrib);
                if (lane == 0 && m < D.gmm_M) S.red[m] = D.gmm_scale * quad + D.gmm_csel[m];      // (merged: 0.5 q - log nll_w; per component: q + const)
            }
            __syncthreads();
            int best = 0; float bl = S.red[0];
            for (int m = 1; m < D.gmm_M; ++m) { const float v = S.red[m]; if (v < bl) { bl = v; best = m; } }
            __syncthreads();
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
                if (wv + mi * (CT / 64) == best && lane < L.NEMB) S.gc[L.emb + lane] = (2.f * D.gmm_scale) * gy[mi] * bpw2;
            if (t == 0) q[Q_PP] = bl + D.gmm_cadd[best];
        } else if (t < L.NEMB) {       // pose prior on the embedding (fitting.py:390-401)
            const float e = S.x[L.emb + t];
            const float dlt = latent_reg ? (e - fd[FD_REG + t]) : e;
            q[Q_PP] = dlt * dlt;
            S.gc[L.emb + t] = 2.f * dlt * bpw2;
        }
        if (t < L.NB) { const float bt = S.x[L.betas + t]; q[Q_SH] = bt * bt; S.gc[L.betas + t] = 2.f * bt * sw2; }
        if (t < 4) {            // angle prior: exp(pose[idx]*sign)^2 * bending weight (prior.py:73-89)
            const int idx = (t == 0) ? 52 : (t == 1) ? 55 : (t == 2) ? 9 : 12;
            const float sg = (t == 0) ? 1.f : -1.f;
            const float e = expf(S.full_pose[3 + idx] * sg);
            q[Q_ANG] = e * e;
            S.dpose[3 + idx] = 2.f * (e * e) * sg * sw.bend;
        }
        if (C.use_hands && t < SFX_NHAND) {
            q[Q_LH] = S.lh45[t] * S.lh45[t]; q[Q_RH] = S.rh45[t] * S.rh45[t];
            S.dpose[75 + t] = 2.f * S.lh45[t] * h2; S.dpose[120 + t] = 2.f * S.rh45[t] * h2;
        }
        if (C.use_face) {
            if (t < L.NE) { const float ev = S.x[L.expr + t]; q[Q_EX] = ev * ev; S.gc[L.expr + t] = 2.f * ev * e2; }
            if (t < 3) {    // (select, not sw.jaw[t]: a dynamically indexed struct becomes a per-thread LDS copy)
                const float jwt = (t == 0) ? sw.jaw[0] : (t == 1) ? sw.jaw[1] : sw.jaw[2];
                const float jv = S.x[L.jaw + t] * jwt; q[Q_JW] = jv * jv; S.gc[L.jaw + t] = 2.f * jv * jwt;
            }
        }
    }
    MARK(22);
    {
        // (every term of these sums sits in a thread < 256 -- keypoints, parameters --: wavefronts beyond the fourth hold zeros
        //  and are left out, so eight wavefronts sum what four do)
        constexpr int QW = (CT / 64 < 4) ? CT / 64 : 4;
        if (wv < QW) {
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const float w = wave_sum_dpp(q[i]);
                if (lane == 0) S.red[wv * NQ + i] = w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            float r = S.red[i];
#pragma unroll
            for (int w = 1; w < QW; ++w) r += S.red[w * NQ + i];
            q[i] = r;
        }
    }
    MARK(23);
    float total;
    if (cam_stage) {
        float joint = q[Q_L];
        if (C.use_conf_cam) joint *= csum;
        joint *= dw2;
        const float dz = ct[2] - est_tz;
        float depth = 0.f;
        if (C.depth_w > 0.f) depth = (C.depth_w * C.depth_w) * (dz * dz);
        total = joint + depth;
        if (t < 3) {
            float g = (t == 0) ? q[Q_D0] : (t == 1) ? q[Q_D1] : q[Q_D2];
            if (t == 2 && C.depth_w > 0.f) g += (C.depth_w * C.depth_w) * 2.f * dz;
            S.gc[L.cam_t + t] = g;
        }
    } else {
        total = q[Q_L] * dw2 + q[Q_PP] * bpw2 + q[Q_SH] * sw2 + q[Q_ANG] * sw.bend;
        if (C.use_face) total = total + q[Q_JW] + q[Q_EX] * e2;
        if (C.use_hands) total = total + q[Q_LH] * h2 + q[Q_RH] * h2;
        // interpenetration (fitting.py:437-455): evaluated on all vertices by csrc/collide.hip right
        // after the dense LBS; its vertex gradient enters the reverse sweep below
        if (C.pen && args.use_dense_verts && sw.coll > 0.f) {
            total = total + sw.coll * D.pen_loss[D.slot[b]];
            if (t == 0 && D.pen_over && D.pen_over[D.slot[b]]) D.pen_flag[b] = 1;      // (diagnostic: see BatchDev.pen_flag)
        }
        if (t < 3) S.gc[L.cam_t + t] = (t == 0) ? q[Q_D0] : (t == 1) ? q[Q_D1] : q[Q_D2];
    }
    __syncthreads();

    MARK(9);
    // ------------------------------------------------------------------ reverse sweep
    // d joints -> items / kinematic joints
    for (int w = t; w < NI * 3; w += CT) {
        const int i = w / 3, r = w % 3;
        S.dvert[w] = S.dj[S.meta[MO_IK + i] * 3 + r] * S.iw[i];
    }
    __syncthreads();
    auto items_dvp = [&](const int ni) {          // d v_posed = T^T d vertex
        for (int w = t; w < ni * 3; w += CT) {
            const int i = w / 3, c = w % 3;
            S.dvp[w] = S.T[i * 12 + 0 + c] * S.dvert[i * 3] + S.T[i * 12 + 4 + c] * S.dvert[i * 3 + 1] +
                       S.T[i * 12 + 8 + c] * S.dvert[i * 3 + 2];
        }
        __syncthreads();
    };
    items_dvp(NI);
    MARK(10);
    // dA[j][e] = sum_items W[v][j] * dT[e]: per-joint item lists (CSR built at model creation,
    // one per dynamic-contour LUT row), visited in ascending item order -> deterministic.  Four lanes share one (j, e):
    // lane g takes entries g, g + 4, ... of the joint's lists and the four partial sums are added as (p0 + p1) + (p2 + p3)
    // inside the quad -- the face landmarks hang on two or three joints whose lists hold ~150 entries each, walked by one
    // lane they were 10 us of a full-model evaluation.
    // (only when the face keypoints are live: with the body's or the hands' few items one lane per (j, e) is faster)
    const int gsh = (cls == 2 && NI > 64) ? 2 : 0, gst = 1 << gsh;
    for (int u = t; u < (SFX_J * 12) << gsh; u += CT) {
        const int w = u >> gsh, g = u & (gst - 1);
        const int j = w / 12, e = w % 12, r = e >> 2, c = e & 3;
        float acc = 0.f;
        const bool sj_lds = M.n_sj <= LDS::kMaxItems * SFX_NW;
        for (int pass = 0; pass < 2; ++pass) {
            const int* st = pass ? (dyn_lds ? S.djs : M.dj_start + (size_t)S.lut_row * (SFX_J + 1)) : (sj_lds ? S.sjs : M.sj_start);
            const int* it = pass ? (dyn_lds ? S.dji : M.dj_item) : (sj_lds ? S.sji : M.sj_item);
            const float* wt = pass ? (dyn_lds ? S.djw : M.dj_w) : (sj_lds ? S.sjw : M.sj_w);
            if (pass && (M.n_dyn_items == 0 || NI <= M.n_static_items)) break;
            for (int q2 = st[j] + g; q2 < st[j + 1]; q2 += gst) {
                const int i = it[q2];
                if (i >= NI) break;              // (lists ascend by item: the rest belongs to keypoints that are not live)
                const float dv = S.dvert[i * 3 + r];
                if (dv != 0.f) acc += wt[q2] * (dv * (c < 3 ? S.vp[i * 3 + c] : 1.f));
            }
        }
        if (gsh) {
            acc = acc + lb_quad<0xB1>(acc);      // quad_perm [1,0,3,2]: p0 + p1 | p2 + p3
            acc = acc + lb_quad<0x4E>(acc);      // quad_perm [2,3,0,1]: (p0 + p1) + (p2 + p3)
        }
        if (g == 0) S.dA[w] = acc;
    }
    MARK(11);
    auto items_dfeat = [&](const int ni, const bool accumulate) {
    // dfeat[k] = sum_items sum_c dirsT[v][c][k] * dvp[c]: each wavefront streams whole 2-KiB rows
    // (RIF in flight), lane l keeps k = 4l..4l+3 and 256+4l..+3; the 4 per-wave partials are added
    // in wave order (fixed association -> deterministic)
    {
        // 1. the rows that carry a nonzero coefficient, compacted in ascending (item, coordinate) order: thread t looks at
        //    item t's three coordinates; positions from wave ballots + the 4 wave totals (a keypoint with zero weight or
        //    confidence contributes nothing and costs nothing)
        int nrows;
        {
            float c3[3] = {0.f, 0.f, 0.f};
            if (t < ni) { c3[0] = S.dvp[t * 3]; c3[1] = S.dvp[t * 3 + 1]; c3[2] = S.dvp[t * 3 + 2]; }
            const unsigned long long b0 = __ballot(c3[0] != 0.f), b1 = __ballot(c3[1] != 0.f), b2 = __ballot(c3[2] != 0.f);
            const unsigned long long lt = (1ull << lane) - 1ull;
            int pos = __popcll(b0 & lt) + __popcll(b1 & lt) + __popcll(b2 & lt);
            if (lane == 0) S.red[wv] = __int_as_float(__popcll(b0) + __popcll(b1) + __popcll(b2));
            __syncthreads();
            int tot = 0;
#pragma unroll
            for (int q2 = 0; q2 < CT / 64; ++q2) { const int n = __float_as_int(S.red[q2]); if (q2 < wv) pos += n; tot += n; }
            nrows = tot;
            if (t < ni) {
                const int v3 = S.ivid[t] * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) if (c3[c] != 0.f) { S.rl[pos] = v3 + c; S.rc[pos] = c3[c]; ++pos; }
            }
            __syncthreads();
        }
        // 2. stream them: lane u of a wavefront looks up entry u of the pass, v_readlane moves row index and coefficient
        //    into scalar registers, the loads use scalar base + lane offset (all RIF rows of a pass are requested back to back)
        constexpr int RW = LDS::kRowWaves;      // (<= CT / 64: wavefronts beyond it hold no rows and no partial)
        float4 pa = {0.f, 0.f, 0.f, 0.f}, pb = pa;
        if (wv < RW)
        for (int w0 = wv * RIF; w0 < nrows; w0 += RW * RIF) {
            float4 da[RIF], db[RIF];
            const bool mine = lane < RIF && w0 + lane < nrows;
            const int myrow = S.rl[mine ? w0 + lane : w0];
            const float myc = mine ? S.rc[w0 + lane] : 0.f;      // (entries past the end: the first row again, coefficient 0)
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const int r = __builtin_amdgcn_readlane(myrow, u);
                const float4* row = reinterpret_cast<const float4*>(M.dirsT + (size_t)r * SFX_KD_PAD);
                da[u] = row[lane]; db[u] = row[64 + lane];
            }
#pragma unroll
            for (int u = 0; u < RIF; ++u) {
                const float dv = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(myc), u));
                pa.x += da[u].x * dv; pa.y += da[u].y * dv; pa.z += da[u].z * dv; pa.w += da[u].w * dv;
                pb.x += db[u].x * dv; pb.y += db[u].y * dv; pb.z += db[u].z * dv; pb.w += db[u].w * dv;
            }
        }
        float4* part = reinterpret_cast<float4*>(S.T);          // S.T is dead here: one 512-float partial per wavefront
        if (wv < RW) { part[wv * 128 + lane] = pa; part[wv * 128 + 64 + lane] = pb; }
        __syncthreads();
        for (int k = t; k < SFX_KD_PAD; k += CT) {
            const int l4 = (k & 255) >> 2, hi = k >> 8, c = k & 3;
            const float* pf = S.T + (hi * 64 + l4) * 4 + c;
            float sumw = pf[0];
#pragma unroll
            for (int w = 1; w < RW; ++w) sumw += pf[w * 512];
            S.dfeat[k] = accumulate ? S.dfeat[k] + sumw : sumw;
        }
    }
    __syncthreads();
    };
    items_dfeat(NI, false);
    // ---- interpenetration: its gradient lives on every vertex; lbs_adjoint.hip has already taken it
    // back to feat and to the skinning transforms (adjoint GEMM), the chain adjoint below does the rest
    if (C.pen && args.use_dense_verts && !cam_stage && sw.coll > 0.f) {
        const int slot = D.slot[b];
        const float* pf = D.pen_dfeat + (size_t)slot * SFX_KD_PAD;
        const float* pa = D.pen_dA + (size_t)slot * SFX_J * 12;
        for (int k = t; k < SFX_KD_PAD; k += CT) S.dfeat[k] += sw.coll * pf[k];
        for (int w = t; w < SFX_J * 12; w += CT) S.dA[w] += sw.coll * pa[w];
        __syncthreads();
    }
    MARK(12);
    // adjoint of the kinematic chain without walking it level by level:
    //   dG_j = sum over the subtree of j of  loc_d . Gh_d^T . Gh_j^-T        (Gh = 4x4 homogeneous G)
    // because G_d = G_j . Rel(j->d).  Pass A forms M_d = loc_d . Gh_d^T (stored at the DFS
    // pre-order position of d), pass B sums contiguous pre-order ranges in fixed order (no
    // atomics -> deterministic), pass C applies Gh_j^-T.  Then dR, d(rel) (pass D) and dJ (pass E).
    // (the 14 joint-regressor entries this thread needs for d(coefficients) below are requested here: they come from
    //  global memory and have the whole chain adjoint to arrive)
    float jdv[14];
    {
        const int l = t % 20, ch = t / 20;
#pragma unroll
        for (int i = 0; i < 14; ++i) {
            const int row = ch * 14 + i;
            jdv[i] = (ch < 12 && l < M.S && row < SFX_J * 3) ? M.J_dirs[(size_t)row * M.S + l] : 0.f;
        }
    }
    float* Mpre = S.T;          // scratch: the item transforms are dead here
    float* Ssub = S.T + 704;    // the subtree sums, before Gh_j^-T is applied (a second 660-float region of the same scratch)
    FOR_CT(w, SFX_J * 12) {
        const int d = w / 12, e = w % 12, r = e >> 2, k = e & 3;
        const float* dAd = &S.dA[d * 12 + r * 4];
        // gradient on the joint's position from the keypoints mapped to it (a list of one or two entries: summed here by
        // each of the four threads of the row rather than in a pass of its own)
        float dpj = 0.f;
        for (int q2 = S.meta[MO_SK0 + d]; q2 < S.meta[MO_SK0 + d + 1]; ++q2) dpj += S.dj[S.meta[MO_SKL + q2] * 3 + r];
        const float l3 = dAd[3] + dpj;
        float v = l3;
        if (k < 3) {
            const float* Gk = &S.G[d * 12 + k * 4];
            const float* Jd = &S.Jr[d * 3];
            v = (dAd[0] - dAd[3] * Jd[0]) * Gk[0] + (dAd[1] - dAd[3] * Jd[1]) * Gk[1] + (dAd[2] - dAd[3] * Jd[2]) * Gk[2] + l3 * Gk[3];
        }
        Mpre[S.meta[MO_PRE + d] * 12 + e] = v;
    }
    __syncthreads();
    FOR_CT(w, SFX_J * 12) {
        const int j = w / 12, e = w % 12;
        const int q0 = S.meta[MO_PRE + j], n = S.meta[MO_SUB + j];
        float acc = 0.f;
        for (int q2 = 0; q2 < n; ++q2) acc += Mpre[(q0 + q2) * 12 + e];
        Ssub[w] = acc;          // subtree sum, still in the basis of the world frame
    }
    __syncthreads();
    FOR_CT(w, SFX_J * 12) {
        const int j = w / 12, e = w % 12, r = e >> 2, c = e & 3;
        const float* Sj = &Ssub[j * 12 + r * 4];
        float v = Sj[3];
        if (c < 3) {
            const float* Gj = &S.G[j * 12];
            v = Gj[c] * (Sj[0] - Sj[3] * Gj[3]) + Gj[4 + c] * (Sj[1] - Sj[3] * Gj[7]) + Gj[8 + c] * (Sj[2] - Sj[3] * Gj[11]);
        }
        S.dG[w] = v;
    }
    __syncthreads();
    FOR_CT(w, SFX_J * 12) {
        const int j = w / 12, e = w % 12;
        const int p = S.meta[MO_PAR + j];
        const float* dGj = &S.dG[j * 12];
        if (e < 9) {
            const int r = e / 3, c = e % 3;
            float v;
            if (p < 0) v = dGj[r * 4 + c];
            else { const float* Gp = &S.G[p * 12]; v = Gp[0 + r] * dGj[0 + c] + Gp[4 + r] * dGj[4 + c] + Gp[8 + r] * dGj[8 + c]; }
            if (j > 0) v += S.dfeat[M.S + 9 * (j - 1) + e];
            S.dR[j * 9 + e] = v;
        } else {
            const int r = e - 9;
            float v;
            if (p < 0) v = dGj[r * 4 + 3];
            else { const float* Gp = &S.G[p * 12]; v = Gp[0 + r] * dGj[3] + Gp[4 + r] * dGj[7] + Gp[8 + r] * dGj[11]; }
            S.drel[j * 3 + r] = v;
        }
    }
    __syncthreads();
    for (int w = t; w < SFX_J * 3; w += CT) {
        const int j = w / 3, c = w % 3;
        float v = -(S.G[j * 12 + 0 + c] * S.dA[j * 12 + 3] + S.G[j * 12 + 4 + c] * S.dA[j * 12 + 7] +
                    S.G[j * 12 + 8 + c] * S.dA[j * 12 + 11]);
        for (int q2 = S.meta[MO_CS + j]; q2 < S.meta[MO_CS + j + 1]; ++q2) v -= S.drel[S.meta[MO_CL + q2] * 3 + c];
        S.dJ[w] = v + S.drel[w];
    }
    __syncthreads();
    MARK(13);
    // Rodrigues adjoint -> dpose ; joint regression adjoint -> shape coefficients
    if (t < SFX_J) {
        float dth[3] = {0.f, 0.f, 0.f};
        rodrigues_bwd(&S.full_pose[3 * t], &S.dR[t * 9], dth);
        S.dpose[3 * t] += dth[0]; S.dpose[3 * t + 1] += dth[1]; S.dpose[3 * t + 2] += dth[2];
    }
    {   // d(coefficients) = dfeat[0..S) + J_dirs^T dJ : 12 partial sums of 14 rows per coefficient
        const int ch = t / 20;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 14; ++i) { const int row = ch * 14 + i; acc += jdv[i] * S.dJ[row < SFX_J * 3 ? row : 0]; }      // (rows past the end: jdv = 0)
        S.red[t] = acc;
    }
    __syncthreads();
    if (t < M.S) {
        float acc = S.dfeat[t];
        for (int ch = 0; ch < 12; ++ch) acc += S.red[ch * 20 + t];
        if (t < L.NB) S.gc[L.betas + t] += acc; else S.gc[L.expr + t - L.NB] += acc;
    }
    __syncthreads();
    MARK(14);
    // dpose -> canonical parameters
    if (t < 3) { S.gc[L.go + t] += S.dpose[t]; S.gc[L.jaw + t] += S.dpose[66 + t];
                 S.gc[L.leye + t] += S.dpose[69 + t]; S.gc[L.reye + t] += S.dpose[72 + t]; }
    if (!C.use_vposer && t >= 64 && t < 64 + 63) S.gc[L.emb + t - 64] += S.dpose[3 + t - 64];
    if (t >= 128 && t < 128 + 2 * L.NPCA) {
        const int q = t - 128; const bool left = q < L.NPCA; const int i = left ? q : q - L.NPCA;
        const float* comp = (left ? M.comp_l : M.comp_r) + i * SFX_NHAND;
        const float* dp = &S.dpose[left ? 75 : 120];
        float acc = 0.f;
        for (int c = 0; c < SFX_NHAND; ++c) acc += comp[c] * dp[c];
        S.gc[(left ? L.lh : L.rh) + i] += acc;
    }
    __syncthreads();
    MARK(15);
    if constexpr (HAS_VP) if (C.use_vposer) vposer_backward<CT>(S.V, M, &S.dpose[3], &S.gc[L.emb], S.T);   // d body_pose -> d latent
    const VarList& vl = vls[cam_stage ? 0 : 1];
    float* gout = D.g + (size_t)b * SFX_NVAR_MAX;
    for (int i = t; i < vl.n; i += CT) { const float gv = S.gc[vl.idx[i]]; gout[i] = gv; if (gflat) gflat[i] = gv; }
    if (t == 0) { D.f[b] = total; if (fout) *fout = total; }
    MARK(16);
}

#pragma clang fp contract(fast)
