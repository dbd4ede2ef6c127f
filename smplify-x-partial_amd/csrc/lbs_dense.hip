// lbs_dense.hip -- dense SMPL-X linear blend skinning for a batch of frames on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32; exact f32 = a k-ordered fmaf chain).
//
// Replaces the vertex half of smplx.lbs.lbs (external package):
//     v_posed = v_template + [betas|expr|pose_feature] . [shapedirs|posedirs]   (K = 506)
//     T       = lbs_weights . A                                                   (K = 55)
//     verts   = T[:3,:3] v_posed + T[:3,3]
// as two GEMMs sharing one output tile:  rows = frames (M), cols = vertices (N).
//   A operand  featT[k][b]      (written per frame by k_closure's export pass)
//   B operand  dirs[k][3*v + c] (the .npz posedirs layout [486, 3V]: 3 coords interleaved,
//                                row length padded to 3*Vpad so tiles are 16-B aligned)
// Workgroup = 4 wavefronts = 32 vertices x 128 frames; each wavefront owns a 32x32 tile with
// 3 accumulators (x,y,z) for v_posed, then per output row 4 accumulators for that row of T,
// fused with the skinning epilogue, so v_posed and T never touch HBM.  The K=506 loop is
// staged through LDS in 22-row chunks (23 chunks), register-prefetched one chunk ahead:
// the 4 wavefronts share the dirs chunk (B operand), each reads its own 32-frame slice of
// the feat chunk (A operand).  Algorithmic traffic per launch:
//     66.0 MB of constants (dirs 61.1+2.5, W 2.3, template 0.1) + B * 125.7 KB of vertices.
#include "sfx_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DT 256           // 4 wavefronts
#define KC 22            // K rows per LDS chunk (506 = 23 * 22)
#define FB 128           // frames per workgroup
#define NB3 96           // 32 vertices * 3 coords
#define A4 (KC * FB / 4)       // float4 per feat chunk  (704)
#define B4 (KC * NB3 / 4)      // float4 per dirs chunk  (528)
#define NLD ((A4 + B4 + DT - 1) / DT)   // float4 loads per thread per chunk (5)

struct __align__(16) DenseLDS {
    float a[2][KC][FB];
    float b[2][KC][NB3];
};

__global__ __launch_bounds__(DT, 2)
void k_lbs_dense(DevModel M, BatchDev D) {
    __shared__ DenseLDS S;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int v0 = blockIdx.x * 32;
    const int fb0 = blockIdx.y * FB;
    const int b0 = fb0 + wv * 32;
    const int jl = lane & 31, kh = lane >> 5;
    const int V = M.V, B = D.nact;      // active (compacted) frames
    const int vtx = v0 + jl;
    const int v = vtx < V ? vtx : V - 1;
    const size_t Bp = (size_t)D.Bpad;
    const size_t LD = (size_t)3 * M.Vpad;
    const int nchunk = M.KD / KC;
    const bool active = b0 < B;        // wave-uniform: this wavefront's 32 frames exist

    // per-thread staging assignment: float4 slots tid + q*256, q = 0..4.  Slots 0,1 always
    // belong to the feat chunk (A4 = 704 > 512), slots 3,4 to the dirs chunk, slot 2 is mixed.
    const float4* gA = reinterpret_cast<const float4*>(D.featT + fb0);
    const float4* gB = reinterpret_cast<const float4*>(M.dirs + (size_t)v0 * 3);
    float4* lA = reinterpret_cast<float4*>(&S.a[0][0][0]);
    float4* lB = reinterpret_cast<float4*>(&S.b[0][0][0]);
    const int Bp4 = (int)(Bp / 4), LD4 = (int)(LD / 4);
    const int stepA = KC * Bp4, stepB = KC * LD4;            // float4 per chunk
    constexpr int bufA = KC * FB / 4, bufB = KC * NB3 / 4;   // float4 per LDS buffer
    static_assert(A4 > 2 * DT && A4 < 3 * DT && A4 + B4 > 4 * DT && A4 + B4 <= 5 * DT, "staging map");
    auto offA = [&](int idx) { return (idx / (FB / 4)) * Bp4 + idx % (FB / 4); };
    auto offB = [&](int i2) { return (i2 / (NB3 / 4)) * LD4 + i2 % (NB3 / 4); };
    const int i0 = tid, i1 = tid + DT, i2 = tid + 2 * DT, i3 = tid + 3 * DT - A4, i4 = tid + 4 * DT - A4;
    const bool a2 = i2 < A4;                 // slot 2: feat or dirs
    const bool ok4 = i4 < B4;                // slot 4: partially filled
    const int g0 = offA(i0), g1 = offA(i1), g2 = a2 ? offA(i2) : offB(i2 - A4), g3 = offB(i3), g4 = offB(ok4 ? i4 : 0);
    const int l2 = a2 ? i2 : i2 - A4;
    float4 s0, s1, s2, s3, s4;
#define STAGE_LOAD(c) do { s0 = gA[g0 + (c) * stepA]; s1 = gA[g1 + (c) * stepA];                      \
        s2 = a2 ? gA[g2 + (c) * stepA] : gB[g2 + (c) * stepB];                                         \
        s3 = gB[g3 + (c) * stepB]; s4 = gB[g4 + (c) * stepB]; } while (0)
#define STAGE_WRITE(buf) do { lA[i0 + (buf) * bufA] = s0; lA[i1 + (buf) * bufA] = s1;                   \
        if (a2) lA[l2 + (buf) * bufA] = s2; else lB[l2 + (buf) * bufB] = s2;                           \
        lB[i3 + (buf) * bufB] = s3; if (ok4) lB[i4 + (buf) * bufB] = s4; } while (0)

    f32x16 ax, ay, az;
    {
        const float tx = M.v_template[v * 3], ty = M.v_template[v * 3 + 1], tz = M.v_template[v * 3 + 2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { ax[r] = tx; ay[r] = ty; az[r] = tz; }
    }
    STAGE_LOAD(0);
    STAGE_WRITE(0);
    __syncthreads();

    for (int c = 0; c < nchunk; ++c) {
        const int cur = c & 1;
        if (c + 1 < nchunk) STAGE_LOAD(c + 1);
        const float* sa = &S.a[cur][kh][wv * 32 + jl];
        const float* sb = &S.b[cur][kh][jl * 3];
        if (active)
#pragma unroll
        for (int kp = 0; kp < KC / 2; ++kp) {
            const float a = sa[kp * 2 * FB];
            const float bx = sb[kp * 2 * NB3], by = sb[kp * 2 * NB3 + 1], bz = sb[kp * 2 * NB3 + 2];
            ax = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bx, ax, 0, 0, 0);
            ay = __builtin_amdgcn_mfma_f32_32x32x2f32(a, by, ay, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bz, az, 0, 0, 0);
        }
        if (c + 1 < nchunk) STAGE_WRITE(cur ^ 1);
        __syncthreads();
    }

    if (!active) return;
    const bool vok = vtx < V;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        f32x16 t0, t1, t2, t3;
#pragma unroll
        for (int r = 0; r < 16; ++r) { t0[r] = 0.f; t1[r] = 0.f; t2[r] = 0.f; t3[r] = 0.f; }
        const float* wt = M.WT + (size_t)kh * M.Vpad + v0 + jl;
        const float* at = D.AT + ((size_t)(rr * 4) * SFX_JPAD + kh) * Bp + b0 + jl;
        const size_t estep = (size_t)SFX_JPAD * Bp;
#pragma unroll 4
        for (int jp = 0; jp < SFX_JPAD / 2; ++jp) {
            const float w = wt[0];
            const float a0 = at[0], a1 = at[estep], a2 = at[2 * estep], a3 = at[3 * estep];
            t0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, w, t0, 0, 0, 0);
            t1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, w, t1, 0, 0, 0);
            t2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2, w, t2, 0, 0, 0);
            t3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3, w, t3, 0, 0, 0);
            wt += (size_t)2 * M.Vpad; at += 2 * Bp;
        }
        if (vok) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int fr = b0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (fr < B)
                    D.verts[((size_t)fr * V + vtx) * 3 + rr] = t0[r] * ax[r] + t1[r] * ay[r] + t2[r] * az[r] + t3[r];
            }
        }
    }
}

void launch_lbs_dense(const DevModel& M, const BatchDev& D, hipStream_t s) {
    dim3 grid((M.V + 31) / 32, (D.nact + FB - 1) / FB);
    if (D.nact <= 0) return;
    hipLaunchKernelGGL(k_lbs_dense, grid, dim3(DT), 0, s, M, D);
}
