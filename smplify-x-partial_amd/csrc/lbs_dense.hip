// lbs_dense.hip -- dense SMPL-X linear blend skinning for a batch of frames on the
// fp32 matrix cores (v_mfma_f32_32x32x2_f32; exact f32 = a k-ordered fmaf chain).
//
// Replaces the vertex half of smplx.lbs.lbs (external package):
//     v_posed = v_template + [betas|expr|pose_feature] . [shapedirs|posedirs]   (K = 506)
//     T       = lbs_weights . A                                                   (K = 55)
//     verts   = T[:3,:3] v_posed + T[:3,3]
// as two GEMMs sharing one output tile:  rows = frames (M), cols = vertices (N).
//   A operand  featT[k][b]   (written per frame by k_closure's export pass)
//   B operand  dirs[k][v][c] (the .npz posedirs layout [486, 3V]: 3 coords interleaved)
// Each wavefront owns a 32-frame x 32-vertex tile: 3 accumulators (x,y,z) for v_posed, then
// per output row 4 accumulators for that row of T, fused with the skinning epilogue, so
// v_posed and T never touch HBM.  Algorithmic traffic per launch:
//     66.0 MB of constants (dirs 61.1+2.5, W 2.3, template 0.1) + B * 125.7 KB of vertices.
#include "sfx_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

#define DT 256           // 4 wavefronts: 32 vertices x 128 frames per workgroup

__global__ __launch_bounds__(DT)
void k_lbs_dense(DevModel M, BatchDev D) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int v0 = blockIdx.x * 32;
    const int b0 = blockIdx.y * 128 + wv * 32;
    if (b0 >= D.Bpad) return;
    const int jl = lane & 31, kh = lane >> 5;
    const int V = M.V, B = D.cfg.B;
    const int vtx = v0 + jl;
    const int v = vtx < V ? vtx : V - 1;
    const size_t Bp = (size_t)D.Bpad;

    f32x16 ax, ay, az;
    {
        const float tx = M.v_template[v * 3], ty = M.v_template[v * 3 + 1], tz = M.v_template[v * 3 + 2];
#pragma unroll
        for (int r = 0; r < 16; ++r) { ax[r] = tx; ay[r] = ty; az[r] = tz; }
    }
    {
        const float* fT = D.featT + (size_t)kh * Bp + b0 + jl;
        const float* dr = M.dirs + ((size_t)kh * V + v) * 3;
        const size_t fstep = 2 * Bp, dstep = (size_t)2 * V * 3;
        const int nkp = M.KD >> 1;
#pragma unroll 4
        for (int kp = 0; kp < nkp; ++kp) {
            const float a = fT[0];
            const float bx = dr[0], by = dr[1], bz = dr[2];
            ax = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bx, ax, 0, 0, 0);
            ay = __builtin_amdgcn_mfma_f32_32x32x2f32(a, by, ay, 0, 0, 0);
            az = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bz, az, 0, 0, 0);
            fT += fstep; dr += dstep;
        }
    }
    f32x16 o[3];
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
#pragma unroll
        for (int r = 0; r < 16; ++r) o[rr][r] = t0[r] * ax[r] + t1[r] * ay[r] + t2[r] * az[r] + t3[r];
    }
    if (vtx < V) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int fr = b0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (fr < B) {
                float* dst = D.verts + ((size_t)fr * V + vtx) * 3;
                dst[0] = o[0][r]; dst[1] = o[1][r]; dst[2] = o[2][r];
            }
        }
    }
}

void launch_lbs_dense(const DevModel& M, const BatchDev& D, hipStream_t s) {
    dim3 grid((M.V + 31) / 32, (D.Bpad + 127) / 128);
    hipLaunchKernelGGL(k_lbs_dense, grid, dim3(DT), 0, s, M, D);
}
