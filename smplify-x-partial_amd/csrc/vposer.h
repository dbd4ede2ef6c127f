// vposer.h -- VPoser-v1 decoder (latent 32 -> 21 axis-angle body joints) and its adjoint, executed
// by the closure workgroup in LDS.
//
// Replaces `vposer.decode(pose_embedding, output_type='aa')` of the external human_body_prior
// (cvpr19) package + torchgeometry 0.1.2 (reference call sites smplifyx/fitting.py:236-238,
// fit_single_frame.py:241-245,607; algorithm: SURVEY.md appendix A.3):
//   fc1 32->512, leaky_relu(0.2), fc2 512->512, leaky_relu(0.2), out 512->126,
//   view(-1,3,2) -> Gram-Schmidt -> R -> quaternion (4-branch, on R^T, eps 1e-6) -> angle-axis.
// Weights are read coalesced: forward GEMVs from the transposed copies ([in][out]), the adjoint
// GEMVs from the original layout ([out][in]).
#pragma once
#include "sfx_internal.h"

#define VP_H 512
#ifndef SFX_VP_UNROLL
#define SFX_VP_UNROLL 12     // 16-byte weight loads in flight per thread of a VPoser matrix-vector product
#endif
#define SFX_STR_(x) #x
#define SFX_STR(x) SFX_STR_(x)
#define VP_O 126

struct alignas(16) VposerLDS {
    float h1[VP_H], h2[VP_H], o[128];      // (contiguous: saved / reloaded as one run)
    float dh[VP_H], dg[VP_H];
    float body[64];
};

__device__ __forceinline__ float leaky(float x) { return x > 0.f ? x : 0.2f * x; }

// 6-D -> axis-angle for one joint; optionally the adjoint (daa -> do6)
__device__ __forceinline__ void vposer_joint(const float* a, float* aa, const float* daa, float* da) {
    const float c0[3] = {a[0], a[2], a[4]}, c1[3] = {a[1], a[3], a[5]};
    const float n0 = fmaxf(sqrtf(c0[0] * c0[0] + c0[1] * c0[1] + c0[2] * c0[2]), 1e-12f);
    const float b1[3] = {c0[0] / n0, c0[1] / n0, c0[2] / n0};
    const float dot = b1[0] * c1[0] + b1[1] * c1[1] + b1[2] * c1[2];
    const float u[3] = {c1[0] - dot * b1[0], c1[1] - dot * b1[1], c1[2] - dot * b1[2]};
    const float n1 = fmaxf(sqrtf(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]), 1e-12f);
    const float b2[3] = {u[0] / n1, u[1] / n1, u[2] / n1};
    const float b3[3] = {b1[1] * b2[2] - b1[2] * b2[1], b1[2] * b2[0] - b1[0] * b2[2], b1[0] * b2[1] - b1[1] * b2[0]};
    // rt = R^T : rows b1, b2, b3
    const float r00 = b1[0], r01 = b1[1], r02 = b1[2], r10 = b2[0], r11 = b2[1], r12 = b2[2],
                r20 = b3[0], r21 = b3[1], r22 = b3[2];
    const bool d2 = r22 < 1e-6f, d0d1 = r00 > r11, d0nd1 = r00 < -r11;
    const int br = d2 ? (d0d1 ? 0 : 1) : (d0nd1 ? 2 : 3);
    float qs[4], ts;
    if (br == 0)      { ts = 1.f + r00 - r11 - r22; qs[0] = r12 - r21; qs[1] = ts; qs[2] = r01 + r10; qs[3] = r20 + r02; }
    else if (br == 1) { ts = 1.f - r00 + r11 - r22; qs[0] = r20 - r02; qs[1] = r01 + r10; qs[2] = ts; qs[3] = r12 + r21; }
    else if (br == 2) { ts = 1.f - r00 - r11 + r22; qs[0] = r01 - r10; qs[1] = r20 + r02; qs[2] = r12 + r21; qs[3] = ts; }
    else              { ts = 1.f + r00 + r11 + r22; qs[0] = ts; qs[1] = r12 - r21; qs[2] = r20 - r02; qs[3] = r01 - r10; }
    const float rs = 0.5f / sqrtf(ts);
    const float w = qs[0] * rs, x = qs[1] * rs, y = qs[2] * rs, z = qs[3] * rs;
    const float s2 = x * x + y * y + z * z;
    const float s = sqrtf(s2);
    const float tt = 2.f * ((w < 0.f) ? atan2f(-s, -w) : atan2f(s, w));
    const float k = (s2 > 0.f) ? tt / s : 2.f;
    aa[0] = x * k; aa[1] = y * k; aa[2] = z * k;
    if (!daa) return;
    // ---- adjoint ----
    float dx = k * daa[0], dy = k * daa[1], dz = k * daa[2], dw = 0.f;
    if (s2 > 0.f) {
        const float dk = daa[0] * x + daa[1] * y + daa[2] * z;
        const float dtt = dk / s;
        float ds = -dk * tt / s2;
        const float den = w * w + s2;
        ds += dtt * 2.f * w / den;
        dw = dtt * 2.f * (-s) / den;
        const float ds2 = ds / (2.f * s);
        dx += 2.f * x * ds2; dy += 2.f * y * ds2; dz += 2.f * z * ds2;
    }
    const float dq[4] = {dw, dx, dy, dz};
    float dqs[4], dts = 0.f;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) { dqs[i] = rs * dq[i]; acc += dq[i] * qs[i]; }
    dts = acc * (-0.5f * rs / ts);
    float d00 = 0, d01 = 0, d02 = 0, d10 = 0, d11 = 0, d12 = 0, d20 = 0, d21 = 0, d22 = 0;
    if (br == 0) { dts += dqs[1]; d12 += dqs[0]; d21 -= dqs[0]; d01 += dqs[2]; d10 += dqs[2]; d20 += dqs[3]; d02 += dqs[3];
                   d00 += dts; d11 -= dts; d22 -= dts; }
    else if (br == 1) { dts += dqs[2]; d20 += dqs[0]; d02 -= dqs[0]; d01 += dqs[1]; d10 += dqs[1]; d12 += dqs[3]; d21 += dqs[3];
                        d00 -= dts; d11 += dts; d22 -= dts; }
    else if (br == 2) { dts += dqs[3]; d01 += dqs[0]; d10 -= dqs[0]; d20 += dqs[1]; d02 += dqs[1]; d12 += dqs[2]; d21 += dqs[2];
                        d00 -= dts; d11 -= dts; d22 += dts; }
    else { dts += dqs[0]; d12 += dqs[1]; d21 -= dqs[1]; d20 += dqs[2]; d02 -= dqs[2]; d01 += dqs[3]; d10 -= dqs[3];
           d00 += dts; d11 += dts; d22 += dts; }
    float db1[3] = {d00, d01, d02}, db2[3] = {d10, d11, d12};
    const float db3[3] = {d20, d21, d22};
    // b3 = b1 x b2 : db1 += b2 x db3 ; db2 += db3 x b1
    db1[0] += b2[1] * db3[2] - b2[2] * db3[1]; db1[1] += b2[2] * db3[0] - b2[0] * db3[2]; db1[2] += b2[0] * db3[1] - b2[1] * db3[0];
    db2[0] += db3[1] * b1[2] - db3[2] * b1[1]; db2[1] += db3[2] * b1[0] - db3[0] * b1[2]; db2[2] += db3[0] * b1[1] - db3[1] * b1[0];
    const float pb2 = b2[0] * db2[0] + b2[1] * db2[1] + b2[2] * db2[2];
    const float du[3] = {(db2[0] - b2[0] * pb2) / n1, (db2[1] - b2[1] * pb2) / n1, (db2[2] - b2[2] * pb2) / n1};
    float dc1[3] = {du[0], du[1], du[2]};
    const float ddot = -(du[0] * b1[0] + du[1] * b1[1] + du[2] * b1[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) { db1[i] += -dot * du[i] + ddot * c1[i]; dc1[i] += ddot * b1[i]; }
    const float pb1 = b1[0] * db1[0] + b1[1] * db1[1] + b1[2] * db1[2];
    const float dc0[3] = {(db1[0] - b1[0] * pb1) / n0, (db1[1] - b1[1] * pb1) / n0, (db1[2] - b1[2] * pb1) / n0};
    da[0] = dc0[0]; da[2] = dc0[1]; da[4] = dc0[2];
    da[1] = dc1[0]; da[3] = dc1[1]; da[5] = dc1[2];
}

// y[o] = sum_k Wt[k][o] * x[k] for o < nout (nout a multiple of 4), all NT threads of the
// workgroup: nout/4 threads cover one row with 16-byte loads, the NT / (nout/4) thread groups split
// the k range and leave partial sums in `part` ([groups][nout], LDS); the caller adds them in group
// order.  16 bytes per lane and 8 rows in flight per thread keep ~64 KB of weights in flight per
// CU -- a scalar 4-byte-per-lane loop is latency bound at a tenth of the L2 bandwidth.
template <int NT, int UNR = SFX_VP_UNROLL>
__device__ __forceinline__ int vp_gemv_partial(const float* __restrict__ Wt, const int K, const int ld, const int nout,
                                               const float* x, float* part) {
    const int t = threadIdx.x;
    const int tpr = nout >> 2;                  // threads per row
    const int groups = NT / tpr;
    const int c4 = t % tpr, g = t / tpr;
    if (g < groups) {
        const int k0 = (K * g) / groups, k1 = (K * (g + 1)) / groups;
        float4 acc = {0.f, 0.f, 0.f, 0.f};
        const float4* w = reinterpret_cast<const float4*>(Wt + (size_t)k0 * ld) + c4;
        const int ld4 = ld >> 2;
#pragma unroll UNR
        for (int k = k0; k < k1; ++k, w += ld4) {
            const float4 wv = *w;
            const float xv = x[k];
            acc.x += wv.x * xv; acc.y += wv.y * xv; acc.z += wv.z * xv; acc.w += wv.w * xv;
        }
        reinterpret_cast<float4*>(part + (size_t)g * nout)[c4] = acc;
    }
    return groups;
}

// z[latent] (LDS) -> V.body[63]; all threads of the workgroup
template <int NT>
__device__ __forceinline__ void vposer_forward(VposerLDS& V, const DevModel& M, const float* z, float* scratch = nullptr) {
    const int t = threadIdx.x, L = M.vp_latent;
    float* part = NT > 256 ? scratch : V.dh;                         // dh, dg (2 x 512 floats, contiguous) are free during the forward
    // (16 weight loads in flight per thread in the forward products: 43 k -> 34 k cycles per decode at 256 frames; the
    //  transposed products, which follow a longer dependent prologue, are fastest with 12 -- 8 / 12 / 16: 51.8 k / 44.8 k /
    //  51.5 k cycles per backward at 256 frames, shader clocks of tools/phase_dense.py)
    int G = vp_gemv_partial<NT, 16>(M.vp_w1T, L, VP_H, VP_H, z, part);
    __syncthreads();
    for (int o = t; o < VP_H; o += NT) {
        float acc = M.vp_b1[o];
        for (int g = 0; g < G; ++g) acc += part[g * VP_H + o];
        V.h1[o] = leaky(acc);
    }
    __syncthreads();
    G = vp_gemv_partial<NT, 16>(M.vp_w2T, VP_H, VP_H, VP_H, V.h1, part);
    __syncthreads();
    for (int o = t; o < VP_H; o += NT) {
        float acc = M.vp_b2[o];
        for (int g = 0; g < G; ++g) acc += part[g * VP_H + o];
        V.h2[o] = leaky(acc);
    }
    __syncthreads();
    G = vp_gemv_partial<NT, 16>(M.vp_w3T, VP_H, 128, 128, V.h2, part);
    __syncthreads();
    for (int o = t; o < VP_O; o += NT) {
        float acc = M.vp_b3[o];
        for (int g = 0; g < G; ++g) acc += part[g * 128 + o];
        V.o[o] = acc;
    }
    __syncthreads();
    if (t < 21) vposer_joint(&V.o[6 * t], &V.body[3 * t], nullptr, nullptr);
    __syncthreads();
}

// dbody[63] (LDS) -> dz[latent] accumulated into gz (LDS, latent entries).  `part` = 1024 floats of
// LDS scratch that is dead here (the caller passes the item-transform array).
template <int NT>
__device__ __forceinline__ void vposer_backward(VposerLDS& V, const DevModel& M, const float* dbody, float* gz, float* part) {
    const int t = threadIdx.x, L = M.vp_latent;
    if (t < 21) { float aa[3]; vposer_joint(&V.o[6 * t], aa, &dbody[3 * t], &V.dg[6 * t]); }
    if (t >= 64 && t < 66) V.dg[VP_O + t - 64] = 0.f;        // pad to 128 rows of W3 (rows 126, 127 are not read)
    __syncthreads();
    int G = vp_gemv_partial<NT>(M.vp_w3, VP_O, VP_H, VP_H, V.dg, part);      // d h2 = W3^T d o, through leaky'
    __syncthreads();
    for (int i = t; i < VP_H; i += NT) {
        float acc = 0.f;
        for (int g = 0; g < G; ++g) acc += part[g * VP_H + i];
        V.dh[i] = acc * (V.h2[i] > 0.f ? 1.f : 0.2f);
    }
    __syncthreads();
    G = vp_gemv_partial<NT>(M.vp_w2, VP_H, VP_H, VP_H, V.dh, part);          // d h1 = W2^T d pre2, through leaky'
    __syncthreads();
    for (int i = t; i < VP_H; i += NT) {
        float acc = 0.f;
        for (int g = 0; g < G; ++g) acc += part[g * VP_H + i];
        V.dg[i] = acc * (V.h1[i] > 0.f ? 1.f : 0.2f);
    }
    __syncthreads();
    G = vp_gemv_partial<NT>(M.vp_w1, VP_H, L, L, V.dg, part);                // d z = W1^T d pre1
    __syncthreads();
    for (int i = t; i < L; i += NT) {
        float acc = 0.f;
        for (int g = 0; g < G; ++g) acc += part[g * L + i];
        gz[i] += acc;
    }
    __syncthreads();
}
