// lbs_adjoint.hip -- adjoint of the dense SMPL-X skinning for a gradient that lives on EVERY vertex
// (the interpenetration term, smplifyx/fitting.py:437-455: autograd walks back through the whole of
// smplx.lbs.lbs; here the vertex gradient g = d pen_loss / d verts comes from csrc/collide.hip).
//
//     verts = T(v) [v_posed; 1],  T(v) = sum_j W[v][j] A_j,  v_posed = v_template + dirs^T feat
//  =>  d v_posed(v) = T(v)[:3,:3]^T g(v)                                   k_adj_prep
//      d feat[k]    = sum_{v,c} dirs[k][3v+c] d v_posed(v)[c]               k_lbs_dense_adj (fp32 MFMA) + k_adj_reduce
//      d A_j        = sum_v W[v][j] g(v) (x) [v_posed(v); 1]                k_adj_dA
// The tick kernel's adjoint pass adds d feat and d A (times coll_loss_weight) to the keypoint term's
// before it walks the kinematic chain back (closure_body.h).
//
// k_lbs_dense_adj is the transpose of k_lbs_dense: C[k][b] = sum_r dirs[k][r] G[b][r] with the
// reduction r = 3v + c over 3 * Vpad = 31 440 and a small output (512 x frames), so the work is
// split along r.  Both operands are contiguous along r: lane (m, q) of a wavefront loads the float4
// dirs[k0 + m][r + 4q .. 4q + 3] and G[b0 + m][r + 4q .. 4q + 3]; element s of those float4 is the
// operand of MFMA step s -- any assignment of reduction indices to MFMA k-slots is valid as long as
// A and B agree, so no LDS transpose is needed and every global load is a 16-byte vector load.
// Wavefront tile: 64 k x 64 frames (4 x 4 MFMA tiles, 64 accumulators); a workgroup's 4 wavefronts
// take 4 consecutive r ranges of the same tile and add their accumulators through LDS in wave
// order; k_adj_reduce adds the workgroups' partials in slice order: no atomics, fixed association.
// Algorithmic traffic per launch: dirs 64.4 MB + G (frames x 125.8 KB); flops 2 x 512 x 31 440 per frame.
#include "sfx_internal.h"
#include "wave_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define ADJ_T 256
#define ADJ_RW_MIN 256           // r range of one wavefront: 256 (few frame tiles: more workgroups) or 512

// d v_posed = T^T g for every vertex of every column that wants the term (zeros where g = 0)
__global__ __launch_bounds__(256)
void k_adj_prep(DevModel M, BatchDev D) {
    const int b = blockIdx.y;
    if (!D.pen_want[b]) return;
    const int v = blockIdx.x * 256 + threadIdx.x;
    if (v >= M.V) return;
    const float* g = D.pen_dverts + ((size_t)b * M.V + v) * 3;
    const float g0 = g[0], g1 = g[1], g2 = g[2];
    float o0 = 0.f, o1 = 0.f, o2 = 0.f;
    const bool nz = g0 != 0.f || g1 != 0.f || g2 != 0.f;
    {   // diagnostics: vertices that carry a gradient (one integer atomic per wavefront)
        const unsigned long long m = __ballot(nz);
        if (m && (threadIdx.x & 63) == (unsigned)(__ffsll((long long)m) - 1)) atomicAdd(&D.ext_n[b], __popcll(m));
    }
    if (nz) {
        const size_t Bp = (size_t)D.Bpad;
        float T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        auto add = [&](const int j, const float w) {
#pragma unroll
            for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                for (int c = 0; c < 3; ++c) T[rr * 3 + c] += w * D.AT[((size_t)(rr * 4 + c) * SFX_JPAD + j) * Bp + b];
        };
        const int* wj = M.Wsp_j + (size_t)v * SFX_NW;
        if (wj[0] >= 0) {
            const float* ww = M.Wsp_w + (size_t)v * SFX_NW;
            for (int q = 0; q < SFX_NW; ++q) if (ww[q] != 0.f) add(wj[q], ww[q]);
        } else {
            for (int j = 0; j < SFX_J; ++j) { const float w = M.W[(size_t)v * SFX_J + j]; if (w != 0.f) add(j, w); }
        }
        o0 = T[0] * g0 + T[3] * g1 + T[6] * g2;
        o1 = T[1] * g0 + T[4] * g1 + T[7] * g2;
        o2 = T[2] * g0 + T[5] * g1 + T[8] * g2;
    }
    float* o = D.adj_G + (size_t)b * 3 * M.Vpad + (size_t)v * 3;
    o[0] = o0; o[1] = o1; o[2] = o2;
}

struct __align__(16) AdjLDS { float acc[3][64][64]; };      // wavefronts 1..3 hand their tile to wavefront 0's lanes

__global__ __launch_bounds__(ADJ_T, 2)
void k_lbs_dense_adj(DevModel M, BatchDev D, int rw) {
    __shared__ AdjLDS S;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    // grid: x = r slice (fastest: the slices of one (k tile, frame tile) stream disjoint parts of dirs),
    // y = k tile (8), z = frame tile
    const int slice = blockIdx.x, k0 = blockIdx.y * 64, b0 = blockIdx.z * 64;
    const int LD = 3 * M.Vpad;
    const int r_lo = min(LD, (slice * 4 + wv) * rw);
    const int r_hi = min(LD, r_lo + rw);
    const float* pa = M.dirs + (size_t)(k0 + m) * LD + 4 * q;
    const float* pb = D.adj_G + (size_t)(b0 + m) * LD + 4 * q;
    const size_t sa = (size_t)16 * LD;          // next MFMA tile: 16 rows further
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 a_c[4], b_c[4], a_n[4], b_n[4];
    auto load = [&](float4 (&a)[4], float4 (&b)[4], const int r) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            a[i] = *reinterpret_cast<const float4*>(pa + i * sa + r);
            b[i] = *reinterpret_cast<const float4*>(pb + i * sa + r);
        }
    };
    if (r_lo < r_hi) load(a_c, b_c, r_lo);
    for (int r = r_lo; r < r_hi; r += 16) {
        const bool more = r + 16 < r_hi;
        if (more) load(a_n, b_n, r + 16);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] = MFMA(a_c[i].x, b_c[j].x, acc[i][j]);
                acc[i][j] = MFMA(a_c[i].y, b_c[j].y, acc[i][j]);
                acc[i][j] = MFMA(a_c[i].z, b_c[j].z, acc[i][j]);
                acc[i][j] = MFMA(a_c[i].w, b_c[j].w, acc[i][j]);
            }
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { a_c[i] = a_n[i]; b_c[i] = b_n[i]; }
        }
    }
    // accumulator (i, j), register e of lane (m, q) = C[k0 + 16 i + 4 q + e][b0 + 16 j + m]
    if (wv > 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) S.acc[wv - 1][16 * i + 4 * q + e][16 * j + m] = acc[i][j][e];
    }
    __syncthreads();
    if (wv == 0) {
        float* out = D.adj_part + ((size_t)slice * SFX_KD_PAD + k0) * D.Bpad + b0;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kr = 16 * i + 4 * q + e, bc = 16 * j + m;
                    const float s = ((acc[i][j][e] + S.acc[0][kr][bc]) + S.acc[1][kr][bc]) + S.acc[2][kr][bc];
                    out[(size_t)kr * D.Bpad + bc] = s;
                }
    }
}

// d feat[b][k] = sum over the slices, in slice order
__global__ __launch_bounds__(256)
void k_adj_reduce(BatchDev D, int n_slices) {
    const int b = blockIdx.x * 64 + (threadIdx.x & 63);
    const int k = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (b >= D.nact || !D.pen_want[b]) return;
    float s = 0.f;
    for (int i = 0; i < n_slices; ++i) s += D.adj_part[((size_t)i * SFX_KD_PAD + k) * D.Bpad + b];
    D.pen_dfeat[(size_t)b * SFX_KD_PAD + k] = s;
}

// d A_j[b] = sum over the vertices skinned by joint j (ascending) of W[v][j] g(v) (x) [v_posed(v); 1]:
// one wavefront per (joint, column); lanes stride the joint's vertex list, DPP reduction (fixed order)
__global__ __launch_bounds__(64)
void k_adj_dA(DevModel M, BatchDev D) {
    const int j = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    if (!D.pen_want[b]) return;
    const float* g = D.pen_dverts + (size_t)b * M.V * 3;
    const float* vp = D.vposed + (size_t)b * M.V * 3;
    float acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = M.jv_start[j] + lane; i < M.jv_start[j + 1]; i += 64) {
        const int v = M.jv_vid[i];
        const float g0 = g[v * 3], g1 = g[v * 3 + 1], g2 = g[v * 3 + 2];
        if (g0 == 0.f && g1 == 0.f && g2 == 0.f) continue;
        const float w = M.jv_w[i];
        const float p[4] = {vp[v * 3], vp[v * 3 + 1], vp[v * 3 + 2], 1.f};
        const float wg[3] = {w * g0, w * g1, w * g2};
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r * 4 + c] += wg[r] * p[c];
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = wave_sum_dpp(acc[e]);
    if (lane == 0) {
        float* o = D.pen_dA + ((size_t)b * SFX_J + j) * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) o[e] = acc[e];
    }
}

int sfx_adj_slices(const DevModel& M) { return (3 * M.Vpad + 4 * ADJ_RW_MIN - 1) / (4 * ADJ_RW_MIN); }      // capacity of adj_part

// gradient of the penetration term with respect to feat (betas / expression / pose feature) and the
// skinning transforms, for every active column whose pen_want flag is set
void launch_pen_adjoint(const DevModel& M, const BatchDev& D, hipStream_t s) {
    if (D.nact <= 0) return;
    const int ftiles = (D.nact + 63) / 64;
    const int rw = ftiles >= 3 ? 512 : ADJ_RW_MIN;       // 8 k tiles x ftiles x slices workgroups: keep >= 256 of them
    const int ns = (3 * M.Vpad + 4 * rw - 1) / (4 * rw);
    hipLaunchKernelGGL(k_adj_prep, dim3((M.V + 255) / 256, D.nact), dim3(256), 0, s, M, D);
    hipLaunchKernelGGL(k_adj_dA, dim3(SFX_J, D.nact), dim3(64), 0, s, M, D);
    hipLaunchKernelGGL(k_lbs_dense_adj, dim3(ns, SFX_KD_PAD / 64, ftiles), dim3(ADJ_T), 0, s, M, D, rw);
    hipLaunchKernelGGL(k_adj_reduce, dim3((D.nact + 63) / 64, SFX_KD_PAD / 4), dim3(256), 0, s, D, ns);
}
