// lbs_dense.hip -- dense SMPL-X linear blend skinning for a batch of frames on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32; exact f32 = a k-ordered fmaf chain).
//
// Replaces the vertex half of smplx.lbs.lbs (external package):
//     v_posed = v_template + [betas|expr|pose_feature] . [shapedirs|posedirs]   (K = 506)
//     T       = lbs_weights . A                                                   (K = 55)
//     verts   = T[:3,:3] v_posed + T[:3,3]
// as two GEMMs sharing one output tile:  rows = frames (M), cols = vertices (N).
//   A operand  featT[k][b]      (written per frame by the tick kernel's export pass; 512 rows,
//                                rows >= 506 are zero)
//   B operand  dirs[k][3*v + c] (the .npz posedirs layout [486, 3V]: 3 coords interleaved; rows
//                                padded to 3*Vpad floats and to 512 rows)
// Work unit = one wavefront = 16 vertices x 32 frames (two 16x16 MFMA tiles): 6 accumulators
// (x,y,z) x 4 registers for v_posed, then per output row 4 x 2 accumulators of T fused with the
// skinning epilogue, so v_posed and T never touch HBM.  Small units (about 17 us of MFMA time)
// keep the 1024 SIMDs balanced for any number of active frames (the fit loop compacts finished
// frames away), and 24 + 32 accumulator registers leave room for 3+ wavefronts per SIMD.
// Workgroup = 4 wavefronts = 16 vertices x 128 frames; K is staged through LDS in 32-row chunks
// (16 chunks), register-prefetched one chunk ahead; the wavefronts share the dirs chunk.
// Algorithmic traffic per launch:
//     66.0 MB of constants (dirs 61.1+2.5, W 2.3, template 0.1) + B * 125.7 KB of vertices.
#include "sfx_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DT 256           // 4 wavefronts
#ifndef KC
#define KC 32            // K rows per LDS chunk (512 = 16 * 32)
#endif
#ifndef MINW
#define MINW 3
#endif
#define FB 128           // frames per workgroup
#define VB 16            // vertices per workgroup
#define NB3 (VB * 3)     // 48 floats of dirs per K row
#define LDA (FB + 16)    // padded row stride: the K rows of one MFMA operand fetch hit distinct banks
#define LDB NB3
#define B4 (KC * NB3 / 4)      // float4 per dirs chunk (384)

struct __align__(16) DenseLDS {
    float a[2][KC][LDA];
    float b[2][KC][LDB];
};

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

__global__ __launch_bounds__(DT, MINW)
void k_lbs_dense(DevModel M, BatchDev D) {
    __shared__ DenseLDS S;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    const int v0 = blockIdx.x * VB;
    const int fb0 = blockIdx.y * FB;
    const int b0 = fb0 + wv * 32;
    const int jl = lane & 15, kq = lane >> 4;
    const int V = M.V, B = D.nact;
    const int vtx = v0 + jl;
    const int v = vtx < V ? vtx : V - 1;
    const size_t Bp = (size_t)D.Bpad;
    const size_t LD = (size_t)3 * M.Vpad;
    const bool active = b0 < B;        // wave-uniform: this wavefront's 32 frames exist

    // staging: 1024 (feat) + 384 (dirs) float4 per chunk: slots tid + q*256; q = 0..3 -> feat rows,
    // slot 4 -> dirs, slot 5 (tid < 128) -> dirs
    const float4* gA = reinterpret_cast<const float4*>(D.featT + fb0);
    const float4* gB = reinterpret_cast<const float4*>(M.dirs + (size_t)v0 * 3);
    const int Bp4 = (int)(Bp / 4), LD4 = (int)(LD / 4);
    const int stepA = KC * Bp4, stepB = KC * LD4;
    int gA_off[4], lA_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + q * DT, row = idx / (FB / 4), c4 = idx % (FB / 4);
        gA_off[q] = row * Bp4 + c4; lA_off[q] = row * LDA + c4 * 4;
    }
    const int i4 = tid, i5 = tid + DT;
    const bool ok5 = i5 < B4;
    const int gB4 = (i4 / (NB3 / 4)) * LD4 + i4 % (NB3 / 4), lB4 = (i4 / (NB3 / 4)) * LDB + (i4 % (NB3 / 4)) * 4;
    const int j5 = ok5 ? i5 : 0;
    const int gB5 = (j5 / (NB3 / 4)) * LD4 + j5 % (NB3 / 4), lB5 = (j5 / (NB3 / 4)) * LDB + (j5 % (NB3 / 4)) * 4;
    float4 s0, s1, s2, s3, s4, s5;
#define STAGE_LOAD(c) do { s0 = gA[gA_off[0] + (c) * stepA]; s1 = gA[gA_off[1] + (c) * stepA];         \
        s2 = gA[gA_off[2] + (c) * stepA]; s3 = gA[gA_off[3] + (c) * stepA];                            \
        s4 = gB[gB4 + (c) * stepB]; s5 = gB[gB5 + (c) * stepB]; } while (0)
#define STAGE_WRITE(buf) do {                                                                          \
        *reinterpret_cast<float4*>(&S.a[buf][0][0] + lA_off[0]) = s0;                                  \
        *reinterpret_cast<float4*>(&S.a[buf][0][0] + lA_off[1]) = s1;                                  \
        *reinterpret_cast<float4*>(&S.a[buf][0][0] + lA_off[2]) = s2;                                  \
        *reinterpret_cast<float4*>(&S.a[buf][0][0] + lA_off[3]) = s3;                                  \
        *reinterpret_cast<float4*>(&S.b[buf][0][0] + lB4) = s4;                                        \
        if (ok5) *reinterpret_cast<float4*>(&S.b[buf][0][0] + lB5) = s5; } while (0)

    f32x4 ax0, ay0, az0, ax1, ay1, az1;       // frames 0-15 / 16-31 of this wavefront's slice
    {
        const float tx = M.v_template[v * 3], ty = M.v_template[v * 3 + 1], tz = M.v_template[v * 3 + 2];
#pragma unroll
        for (int r = 0; r < 4; ++r) { ax0[r] = tx; ay0[r] = ty; az0[r] = tz; ax1[r] = tx; ay1[r] = ty; az1[r] = tz; }
    }
    STAGE_LOAD(0);
    STAGE_WRITE(0);
    __syncthreads();

    constexpr int NCHUNK = SFX_KD_PAD / KC;
    for (int c = 0; c < NCHUNK; ++c) {
        const int cur = c & 1;
        if (c + 1 < NCHUNK) STAGE_LOAD(c + 1);
        if (active) {
            const float* sa = &S.a[cur][kq][wv * 32 + jl];
            const float* sb = &S.b[cur][kq][jl * 3];
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                const float a0 = sa[ks * 4 * LDA], a1 = sa[ks * 4 * LDA + 16];
                const float bx = sb[ks * 4 * LDB], by = sb[ks * 4 * LDB + 1], bz = sb[ks * 4 * LDB + 2];
                ax0 = MFMA(a0, bx, ax0); ay0 = MFMA(a0, by, ay0); az0 = MFMA(a0, bz, az0);
                ax1 = MFMA(a1, bx, ax1); ay1 = MFMA(a1, by, ay1); az1 = MFMA(a1, bz, az1);
            }
        }
        if (c + 1 < NCHUNK) STAGE_WRITE(cur ^ 1);
        __syncthreads();
    }
    if (!active) return;
#ifdef NO_T
    if (vtx < V && b0 < B) D.verts[((size_t)b0 * V + vtx) * 3] = ax0[0] + ay0[1] + az0[2] + ax1[3] + ay1[0] + az1[1];
    return;
#endif
    const bool vok = vtx < V;
    // skinning GEMM T = W . A restricted to the joints that carry weight in this 16-vertex tile
    // (exact: the skipped products are structural zeros of lbs_weights; ascending joint order kept)
    const int tile = blockIdx.x;
    const int njs = M.tj_n[tile] >> 2;
    const int* jl4 = M.tj_list + (size_t)tile * SFX_JPAD + kq;
    const float* wl = M.tj_w + ((size_t)tile * SFX_JPAD + kq) * 16 + jl;
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        f32x4 t00 = {0, 0, 0, 0}, t01 = t00, t02 = t00, t03 = t00, t10 = t00, t11 = t00, t12 = t00, t13 = t00;
        const float* at0 = D.AT + ((size_t)(rr * 4) * SFX_JPAD) * Bp + b0 + jl;
        const size_t estep = (size_t)SFX_JPAD * Bp;
        for (int js = 0; js < njs; ++js) {
            const float w = wl[js * 64];
            const float* at = at0 + (size_t)jl4[js * 4] * Bp;
            const float p0 = at[0], p1 = at[estep], p2 = at[2 * estep], p3 = at[3 * estep];
            const float q0 = at[16], q1 = at[estep + 16], q2 = at[2 * estep + 16], q3 = at[3 * estep + 16];
            t00 = MFMA(p0, w, t00); t01 = MFMA(p1, w, t01); t02 = MFMA(p2, w, t02); t03 = MFMA(p3, w, t03);
            t10 = MFMA(q0, w, t10); t11 = MFMA(q1, w, t11); t12 = MFMA(q2, w, t12); t13 = MFMA(q3, w, t13);
        }
        if (vok) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f0 = b0 + kq * 4 + r, f1 = f0 + 16;
                if (f0 < B) D.verts[((size_t)f0 * V + vtx) * 3 + rr] = t00[r] * ax0[r] + t01[r] * ay0[r] + t02[r] * az0[r] + t03[r];
                if (f1 < B) D.verts[((size_t)f1 * V + vtx) * 3 + rr] = t10[r] * ax1[r] + t11[r] * ay1[r] + t12[r] * az1[r] + t13[r];
            }
        }
    }
}

void launch_lbs_dense(const DevModel& M, const BatchDev& D, hipStream_t s) {
    if (D.nact <= 0) return;
    dim3 grid((M.V + VB - 1) / VB, (D.nact + FB - 1) / FB);
    hipLaunchKernelGGL(k_lbs_dense, grid, dim3(DT), 0, s, M, D);
}
