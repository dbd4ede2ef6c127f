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
#include "sfx_internal.h"
#include "wave_ops.h"

#define CT 256

struct __align__(16) FrameLDS {
    float feat[SFX_KD_PAD];        // first: read as float4
    float x[SFX_NPAR_MAX];
    float full_pose[168];
    float R[SFX_J * 9];
    float Jr[SFX_J * 3];
    float G[SFX_J * 12];
    float A[SFX_J * 12];
    float vp[SFX_MAX_ITEMS * 3];
    float T[SFX_MAX_ITEMS * 12];
    float vert[SFX_MAX_ITEMS * 3];
    float dvert[SFX_MAX_ITEMS * 3];
    float dvp[SFX_MAX_ITEMS * 3];
    int   ivid[SFX_MAX_ITEMS];
    float iw[SFX_MAX_ITEMS];
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
    float red[CT];
    float lh45[SFX_NHAND], rh45[SFX_NHAND];
    float scal[16];
    int   lut_row;
    int   meta[SFX_META_N];     // tree / joint-map tables (one coalesced load instead of
                                // dependent global loads inside every level of the chain)
};

__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }

// fixed-order block reduction: DPP sum per wavefront, then the CT/64 partials in order
// (2 barriers instead of a 9-barrier LDS tree); result in all threads
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

__global__ __launch_bounds__(CT)
void k_closure(DevModel M, BatchDev D, const VarList* __restrict__ vls, const StageW* __restrict__ sws,
               ClosureArgs args) {
    __shared__ FrameLDS S;
    const int b = blockIdx.x;
    const int t = threadIdx.x;
    const int lane = t & 63, wv = t >> 6;
    const ParLayout& L = D.L;
    const BatchCfgDev& C = D.cfg;

    int stage = (args.stage_override != -2) ? args.stage_override : D.stage[b];
    if (stage >= C.n_stages && !args.forward_only) return;      // frame finished
    const bool cam_stage = (stage < 0);

    // ------------------------------------------------------------------ load parameters
    const float* xsrc = (args.from_X ? D.X : D.Xt) + (size_t)b * SFX_NPAR_MAX;
    for (int i = t; i < L.npar; i += CT) S.x[i] = xsrc[i];
    for (int i = t; i < SFX_META_N; i += CT) S.meta[i] = M.meta[i];
    for (int i = t; i < SFX_KD_PAD; i += CT) { S.feat[i] = 0.f; S.dfeat[i] = 0.f; }
    for (int i = t; i < SFX_NPAR_MAX; i += CT) S.gc[i] = 0.f;
    for (int i = t; i < 168; i += CT) S.dpose[i] = 0.f;
    __syncthreads();
    const float* bodypose = C.use_vposer ? (D.bodypose + (size_t)b * 63) : (S.x + L.emb);

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

    // ------------------------------------------------------------------ Rodrigues, rest joints
    if (t < SFX_J) {
        float R[9];
        rodrigues_fwd(&S.full_pose[3 * t], R);
#pragma unroll
        for (int e = 0; e < 9; ++e) S.R[t * 9 + e] = R[e];
        if (t > 0) {
#pragma unroll
            for (int e = 0; e < 9; ++e)
                S.feat[M.S + 9 * (t - 1) + e] = R[e] - ((e == 0 || e == 4 || e == 8) ? 1.f : 0.f);
        }
    } else if (t >= 64 && t < 64 + SFX_J * 3) {
        const int i = t - 64;
        float v = M.J_template[i];
        const float* jd = M.J_dirs + (size_t)i * M.S;
        for (int l = 0; l < M.S; ++l) v += jd[l] * S.feat[l];
        S.Jr[i] = v;
    }
    __syncthreads();

    // ------------------------------------------------------------------ kinematic chain
    for (int lev = 0; lev < M.n_levels; ++lev) {
        const int i0 = M.level_start[lev], n = M.level_start[lev + 1] - i0;
        if (t < n) {
            const int j = S.meta[MO_LJ + i0 + t];
            const int p = S.meta[MO_PAR + j];
            const float* Rj = &S.R[j * 9];
            float* Gj = &S.G[j * 12];
            if (p < 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    Gj[r * 4 + 0] = Rj[r * 3 + 0]; Gj[r * 4 + 1] = Rj[r * 3 + 1];
                    Gj[r * 4 + 2] = Rj[r * 3 + 2]; Gj[r * 4 + 3] = S.Jr[j * 3 + r];
                }
            } else {
                const float* Gp = &S.G[p * 12];
                const float rel[3] = {S.Jr[j * 3] - S.Jr[p * 3], S.Jr[j * 3 + 1] - S.Jr[p * 3 + 1],
                                      S.Jr[j * 3 + 2] - S.Jr[p * 3 + 2]};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        Gj[r * 4 + c] = Gp[r * 4 + 0] * Rj[0 * 3 + c] + Gp[r * 4 + 1] * Rj[1 * 3 + c] +
                                        Gp[r * 4 + 2] * Rj[2 * 3 + c];
                    Gj[r * 4 + 3] = Gp[r * 4 + 0] * rel[0] + Gp[r * 4 + 1] * rel[1] + Gp[r * 4 + 2] * rel[2] +
                                    Gp[r * 4 + 3];
                }
            }
        }
        __syncthreads();
    }
    if (t < SFX_J) {
        const float* Gj = &S.G[t * 12];
        float* Aj = &S.A[t * 12];
        const float* Jj = &S.Jr[t * 3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            Aj[r * 4 + 0] = Gj[r * 4 + 0]; Aj[r * 4 + 1] = Gj[r * 4 + 1]; Aj[r * 4 + 2] = Gj[r * 4 + 2];
            Aj[r * 4 + 3] = Gj[r * 4 + 3] - (Gj[r * 4 + 0] * Jj[0] + Gj[r * 4 + 1] * Jj[1] + Gj[r * 4 + 2] * Jj[2]);
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

    // ------------------------------------------------------------------ dense export
    if (args.export_dense) {
        for (int k = t; k < M.KD; k += CT) D.featT[(size_t)k * D.Bpad + b] = S.feat[k];
        for (int i = t; i < SFX_J * 12; i += CT) {
            const int j = i / 12, e = i % 12;
            D.AT[((size_t)e * SFX_JPAD + j) * D.Bpad + b] = S.A[i];
        }
        if (args.forward_only == 2) return;     // export pass only
    }

    // ------------------------------------------------------------------ needed vertices
    const int NI = M.n_items;
    for (int i = t; i < NI; i += CT) {
        const int dd = M.item_dyn[i];
        if (dd < 0) { S.ivid[i] = M.item_vid[i]; S.iw[i] = M.item_w[i]; }
        else {
            const int l = dd / 3, c = dd % 3;
            const int face = M.dyn_faces[S.lut_row * M.n_dyn + l];
            S.ivid[i] = M.faces[face * 3 + c];
            S.iw[i] = M.dyn_bary[(S.lut_row * M.n_dyn + l) * 3 + c];
        }
    }
    __syncthreads();
    // v_posed rows: one wavefront per (item, coord) dot product of length KD_PAD
    {
        const float4* f4 = reinterpret_cast<const float4*>(S.feat);
        const float4 fa = f4[lane], fb = f4[64 + lane];
        // 4 rows per wavefront per pass: 8 independent 1-KiB loads in flight before the reductions
        for (int w0 = wv * 4; w0 < NI * 3; w0 += (CT / 64) * 4) {
            float4 da[4], db[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = (w0 + u < NI * 3) ? w0 + u : w0;
                const int v = S.ivid[w / 3];
                const float4* row = reinterpret_cast<const float4*>(M.dirsT + ((size_t)v * 3 + w % 3) * SFX_KD_PAD);
                da[u] = row[lane]; db[u] = row[64 + lane];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int w = w0 + u;
                float acc = fa.x * da[u].x + fa.y * da[u].y + fa.z * da[u].z + fa.w * da[u].w +
                            fb.x * db[u].x + fb.y * db[u].y + fb.z * db[u].z + fb.w * db[u].w;
                acc = wave_sum(acc);
                if (lane == 0 && w < NI * 3) S.vp[w] = M.v_template[S.ivid[w / 3] * 3 + w % 3] + acc;
            }
        }
    }
    // skinning transforms of the items
    for (int w = t; w < NI * 12; w += CT) {
        const int i = w / 12, e = w % 12;
        const float* Wr = M.W + (size_t)S.ivid[i] * SFX_J;
        float acc = 0.f;
        for (int j = 0; j < SFX_J; ++j) {
            const float wj = Wr[j];
            if (wj != 0.f) acc += wj * S.A[j * 12 + e];
        }
        S.T[w] = acc;
    }
    __syncthreads();
    for (int w = t; w < NI * 3; w += CT) {
        const int i = w / 3, r = w % 3;
        const float* Ti = &S.T[i * 12 + r * 4];
        const float* vp = &S.vp[i * 3];
        float v = Ti[0] * vp[0] + Ti[1] * vp[1] + Ti[2] * vp[2] + Ti[3];
        if (args.use_dense_verts) v = D.verts[((size_t)b * M.V + S.ivid[i]) * 3 + r];
        S.vert[w] = v;
    }
    __syncthreads();

    // ------------------------------------------------------------------ mapped joints
    const int K = M.K;
    for (int w = t; w < K * 3; w += CT) {
        const int k = w / 3, r = w % 3;
        float v;
        if (S.meta[MO_JT + k] == 0) v = S.G[S.meta[MO_JS + k] * 12 + r * 4 + 3];
        else {
            v = 0.f;
            const int i0 = S.meta[MO_JI0 + k], n = S.meta[MO_JN + k];
            if (n == 1 && S.iw[i0] == 1.f) v = S.vert[i0 * 3 + r];
            else for (int i = 0; i < n; ++i) v += S.vert[(i0 + i) * 3 + r] * S.iw[i0 + i];
        }
        S.joints[w] = v;
    }
    __syncthreads();
    if (args.forward_only) {
        if (D.joints) for (int w = t; w < K * 3; w += CT) D.joints[(size_t)b * K * 3 + w] = S.joints[w];
        if (D.fullpose) for (int w = t; w < SFX_POSE; w += CT) D.fullpose[(size_t)b * SFX_POSE + w] = S.full_pose[w];
        return;
    }

    // ------------------------------------------------------------------ losses
    const float* cam = D.cam + (size_t)b * 8;
    const float fx = cam[0], fy = cam[1], cx = cam[2], cy = cam[3], dwt = cam[4], est_tz = cam[5];
    const float* Rc = D.camR + (size_t)b * 9;
    const float* ct = S.x + L.cam_t;
    const float dw2 = dwt * dwt;
    const float rho2 = C.rho * C.rho;
    const StageW sw = cam_stage ? StageW{} : sws[stage];

    float csum = 1.f;
    if (cam_stage && C.use_conf_cam) {
        float p = 0.f;
        if (t < K) { const float cm = D.cmask[(size_t)b * K + t]; const float cf = D.conf[(size_t)b * K + t];
                     p = (cm != 0.f) ? cf * cf : 0.f; }
        csum = block_sum(p, S.red);
    }
    float lpart = 0.f;
    float dpc[3] = {0.f, 0.f, 0.f};
    if (t < K) {
        const float* p = &S.joints[t * 3];
        const float pcx = Rc[0] * p[0] + Rc[1] * p[1] + Rc[2] * p[2] + ct[0];
        const float pcy = Rc[3] * p[0] + Rc[4] * p[1] + Rc[5] * p[2] + ct[1];
        const float pcz = Rc[6] * p[0] + Rc[7] * p[1] + Rc[8] * p[2] + ct[2];
        const float ix = pcx / pcz, iy = pcy / pcz;
        const float u = fx * ix + cx, v = fy * iy + cy;
        const float gx = D.gt[((size_t)b * K + t) * 2], gy = D.gt[((size_t)b * K + t) * 2 + 1];
        const float rx = gx - u, ry = gy - v;
        float du, dv;       // dL/du, dL/dv
        if (cam_stage) {
            const float cm = D.cmask[(size_t)b * K + t];
            if (cm != 0.f) {
                lpart = rx * rx + ry * ry;
                du = -2.f * rx * csum * dw2; dv = -2.f * ry * csum * dw2;
            } else { du = 0.f; dv = 0.f; }
        } else {
            float w = D.jw[(size_t)b * K + t];
            if (t >= C.nbj) w = (t < C.nbj + 42) ? ((w != 0.f) ? sw.hand_jw : 0.f) : ((w != 0.f) ? sw.face_jw : 0.f);
            if (C.use_conf) w *= D.conf[(size_t)b * K + t];
            const float w2 = w * w;
            if (w2 != 0.f) {
                const float sx = rx * rx, sy = ry * ry;
                const float gmx = rho2 * (sx / (sx + rho2)), gmy = rho2 * (sy / (sy + rho2));
                lpart = w2 * gmx + w2 * gmy;
                du = -(w2 * dw2) * gmof_grad(rx, rho2);
                dv = -(w2 * dw2) * gmof_grad(ry, rho2);
            } else { du = 0.f; dv = 0.f; }
        }
        const float dix = du * fx, diy = dv * fy;
        dpc[0] = dix / pcz; dpc[1] = diy / pcz;
        dpc[2] = -(dix * pcx + diy * pcy) / (pcz * pcz);
        S.dj[t * 3 + 0] = Rc[0] * dpc[0] + Rc[3] * dpc[1] + Rc[6] * dpc[2];
        S.dj[t * 3 + 1] = Rc[1] * dpc[0] + Rc[4] * dpc[1] + Rc[7] * dpc[2];
        S.dj[t * 3 + 2] = Rc[2] * dpc[0] + Rc[5] * dpc[1] + Rc[8] * dpc[2];
    }
    float lsum = block_sum(lpart, S.red);
    // camera-translation gradient = sum_k dpc
    const float gct0 = block_sum(dpc[0], S.red), gct1 = block_sum(dpc[1], S.red), gct2 = block_sum(dpc[2], S.red);
    float total;
    if (cam_stage) {
        float joint = lsum;
        if (C.use_conf_cam) joint *= csum;
        joint *= dw2;
        const float dz = ct[2] - est_tz;
        float depth = 0.f;
        if (C.depth_w > 0.f) depth = (C.depth_w * C.depth_w) * (dz * dz);
        total = joint + depth;
        if (t < 3) {
            float g = (t == 0) ? gct0 : (t == 1) ? gct1 : gct2;
            if (t == 2 && C.depth_w > 0.f) g += (C.depth_w * C.depth_w) * 2.f * dz;
            S.gc[L.cam_t + t] = g;
        }
    } else {
        const float joint = lsum * dw2;
        // ---- priors (single wavefront 0; tiny) ----
        float pp = 0.f, shp = 0.f, ang = 0.f, lhp = 0.f, rhp = 0.f, exl = 0.f, jwl = 0.f;
        const float bpw2 = sw.bpw * sw.bpw;
        const bool latent_reg = C.use_vposer ? (stage + 1 == C.n_stages && C.has_reg) : (C.has_reg != 0);
        const float* reg = D.regpose + (size_t)b * 63;
        {   // pose prior on the embedding (fitting.py:390-401)
            float p = 0.f;
            if (t < L.NEMB) {
                const float e = S.x[L.emb + t];
                const float dlt = latent_reg ? (e - reg[t]) : e;
                p = dlt * dlt;
                S.gc[L.emb + t] = 2.f * dlt * bpw2;
            }
            pp = block_sum(p, S.red) * bpw2;
        }
        {
            float p = 0.f;
            if (t < L.NB) { const float bt = S.x[L.betas + t]; p = bt * bt; S.gc[L.betas + t] = 2.f * bt * (sw.sw * sw.sw); }
            shp = block_sum(p, S.red) * (sw.sw * sw.sw);
        }
        {   // angle prior: exp(pose[idx]*sign)^2 * bending weight (prior.py:73-89, fitting.py:407-408)
            float p = 0.f;
            if (t < 4) {
                const int idx = (t == 0) ? 52 : (t == 1) ? 55 : (t == 2) ? 9 : 12;
                const float sg = (t == 0) ? 1.f : -1.f;
                const float e = expf(S.full_pose[3 + idx] * sg);
                p = e * e;
                S.dpose[3 + idx] = 2.f * p * sg * sw.bend;
            }
            ang = block_sum(p, S.red) * sw.bend;
        }
        if (C.use_hands) {
            const float h2 = sw.hpw * sw.hpw;
            float p = 0.f, q = 0.f;
            if (t < SFX_NHAND) { p = S.lh45[t] * S.lh45[t]; q = S.rh45[t] * S.rh45[t];
                                 S.dpose[75 + t] = 2.f * S.lh45[t] * h2; S.dpose[120 + t] = 2.f * S.rh45[t] * h2; }
            lhp = block_sum(p, S.red) * h2;
            rhp = block_sum(q, S.red) * h2;
        }
        if (C.use_face) {
            const float e2 = sw.epw * sw.epw;
            float p = 0.f, q = 0.f;
            if (t < L.NE) { const float ev = S.x[L.expr + t]; p = ev * ev; S.gc[L.expr + t] = 2.f * ev * e2; }
            if (t < 3) { const float jv = S.x[L.jaw + t] * sw.jaw[t]; q = jv * jv; S.gc[L.jaw + t] = 2.f * jv * sw.jaw[t]; }
            exl = block_sum(p, S.red) * e2;
            jwl = block_sum(q, S.red);
        }
        total = joint + pp + shp + ang;
        if (C.use_face) total = total + jwl + exl;
        if (C.use_hands) total = total + lhp + rhp;
        if (t < 3) S.gc[L.cam_t + t] = (t == 0) ? gct0 : (t == 1) ? gct1 : gct2;
    }
    __syncthreads();

    // ------------------------------------------------------------------ reverse sweep
    // d joints -> items / kinematic joints
    for (int w = t; w < NI * 3; w += CT) {
        const int i = w / 3, r = w % 3;
        S.dvert[w] = S.dj[S.meta[MO_IK + i] * 3 + r] * S.iw[i];
    }
    __syncthreads();
    for (int w = t; w < NI * 3; w += CT) {
        const int i = w / 3, c = w % 3;
        S.dvp[w] = S.T[i * 12 + 0 + c] * S.dvert[i * 3] + S.T[i * 12 + 4 + c] * S.dvert[i * 3 + 1] +
                   S.T[i * 12 + 8 + c] * S.dvert[i * 3 + 2];
    }
    __syncthreads();
    // dA[j][e] = sum_items W[v][j] * dT[e]
    for (int w = t; w < SFX_J * 12; w += CT) {
        const int j = w / 12, e = w % 12, r = e >> 2, c = e & 3;
        float acc = 0.f;
        for (int i = 0; i < NI; ++i) {
            const float dv = S.dvert[i * 3 + r];
            if (dv == 0.f) continue;
            const float wj = M.W[(size_t)S.ivid[i] * SFX_J + j];
            if (wj != 0.f) acc += wj * (dv * (c < 3 ? S.vp[i * 3 + c] : 1.f));
        }
        S.dA[w] = acc;
    }
    // dfeat[k] = sum_items sum_c dirsT[v][c][k] * dvp[c]
    for (int k = t; k < M.KD; k += CT) {
        float acc = 0.f;
        for (int i = 0; i < NI; ++i) {
            const float d0 = S.dvp[i * 3], d1 = S.dvp[i * 3 + 1], d2 = S.dvp[i * 3 + 2];
            if (d0 == 0.f && d1 == 0.f && d2 == 0.f) continue;
            const float* row = M.dirsT + (size_t)S.ivid[i] * 3 * SFX_KD_PAD + k;
            acc += row[0] * d0 + row[SFX_KD_PAD] * d1 + row[2 * SFX_KD_PAD] * d2;
        }
        S.dfeat[k] = acc;
    }
    __syncthreads();
    // kinematic chain, deepest level first; parents gather from their children
    for (int lev = M.n_levels - 1; lev >= 0; --lev) {
        const int i0 = M.level_start[lev], n = M.level_start[lev + 1] - i0;
        if (t < n) {
            const int j = S.meta[MO_LJ + i0 + t];
            const int p = S.meta[MO_PAR + j];
            const float* dAj = &S.dA[j * 12];
            const float* Jj = &S.Jr[j * 3];
            float dGj[12];
            // posed joint adjoint: every mapped joint reading kinematic joint j
            float dpj[3] = {0.f, 0.f, 0.f};
            for (int q = S.meta[MO_SK0 + j]; q < S.meta[MO_SK0 + j + 1]; ++q) {
                const int k = S.meta[MO_SKL + q];
                dpj[0] += S.dj[k * 3]; dpj[1] += S.dj[k * 3 + 1]; dpj[2] += S.dj[k * 3 + 2];
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const float dat = dAj[r * 4 + 3];
                dGj[r * 4 + 0] = dAj[r * 4 + 0] - dat * Jj[0];
                dGj[r * 4 + 1] = dAj[r * 4 + 1] - dat * Jj[1];
                dGj[r * 4 + 2] = dAj[r * 4 + 2] - dat * Jj[2];
                dGj[r * 4 + 3] = dat + dpj[r];
            }
            float dJj[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                dJj[c] = -(S.G[j * 12 + 0 + c] * dAj[3] + S.G[j * 12 + 4 + c] * dAj[7] + S.G[j * 12 + 8 + c] * dAj[11]);
            for (int q = S.meta[MO_CS + j]; q < S.meta[MO_CS + j + 1]; ++q) {
                const int ch = S.meta[MO_CL + q];
                const float* dGc = &S.dG[ch * 12];
                const float* Rch = &S.R[ch * 9];
                const float rel[3] = {S.Jr[ch * 3] - Jj[0], S.Jr[ch * 3 + 1] - Jj[1], S.Jr[ch * 3 + 2] - Jj[2]};
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        dGj[r * 4 + c] += dGc[r * 4 + 0] * Rch[c * 3 + 0] + dGc[r * 4 + 1] * Rch[c * 3 + 1] +
                                          dGc[r * 4 + 2] * Rch[c * 3 + 2] + dGc[r * 4 + 3] * rel[c];
                    dGj[r * 4 + 3] += dGc[r * 4 + 3];
                }
                dJj[0] -= S.drel[ch * 3]; dJj[1] -= S.drel[ch * 3 + 1]; dJj[2] -= S.drel[ch * 3 + 2];
            }
#pragma unroll
            for (int e = 0; e < 12; ++e) S.dG[j * 12 + e] = dGj[e];
            float dRj[9], drl[3];
            if (p < 0) {
#pragma unroll
                for (int r = 0; r < 3; ++r) { dRj[r * 3] = dGj[r * 4]; dRj[r * 3 + 1] = dGj[r * 4 + 1];
                                              dRj[r * 3 + 2] = dGj[r * 4 + 2]; drl[r] = dGj[r * 4 + 3]; }
            } else {
                const float* Gp = &S.G[p * 12];
#pragma unroll
                for (int r = 0; r < 3; ++r) {
#pragma unroll
                    for (int c = 0; c < 3; ++c)
                        dRj[r * 3 + c] = Gp[0 + r] * dGj[0 + c] + Gp[4 + r] * dGj[4 + c] + Gp[8 + r] * dGj[8 + c];
                    drl[r] = Gp[0 + r] * dGj[3] + Gp[4 + r] * dGj[7] + Gp[8 + r] * dGj[11];
                }
            }
            if (j > 0) {
#pragma unroll
                for (int e = 0; e < 9; ++e) dRj[e] += S.dfeat[M.S + 9 * (j - 1) + e];
            }
#pragma unroll
            for (int e = 0; e < 9; ++e) S.dR[j * 9 + e] = dRj[e];
            S.drel[j * 3] = drl[0]; S.drel[j * 3 + 1] = drl[1]; S.drel[j * 3 + 2] = drl[2];
            S.dJ[j * 3] = dJj[0] + drl[0]; S.dJ[j * 3 + 1] = dJj[1] + drl[1]; S.dJ[j * 3 + 2] = dJj[2] + drl[2];
        }
        __syncthreads();
    }
    // Rodrigues adjoint -> dpose ; joint regression adjoint -> shape coefficients
    if (t < SFX_J) {
        float dth[3] = {0.f, 0.f, 0.f};
        rodrigues_bwd(&S.full_pose[3 * t], &S.dR[t * 9], dth);
        S.dpose[3 * t] += dth[0]; S.dpose[3 * t + 1] += dth[1]; S.dpose[3 * t + 2] += dth[2];
    } else if (t >= 64 && t < 64 + M.S) {
        const int l = t - 64;
        float acc = S.dfeat[l];
        for (int i = 0; i < SFX_J * 3; ++i) acc += M.J_dirs[(size_t)i * M.S + l] * S.dJ[i];
        if (l < L.NB) S.gc[L.betas + l] += acc; else S.gc[L.expr + l - L.NB] += acc;
    }
    __syncthreads();
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
    // TODO(vposer): body-pose adjoint through the VPoser decoder is applied by k_vposer_bwd.
    const VarList& vl = vls[cam_stage ? 0 : 1];
    float* gout = D.g + (size_t)b * SFX_NVAR_MAX;
    for (int i = t; i < vl.n; i += CT) gout[i] = S.gc[vl.idx[i]];
    if (t == 0) D.f[b] = total;
}

void launch_closure(const DevModel& M, const BatchDev& D, const VarList* vl_dev, const StageW* sw_dev,
                    const ClosureArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(k_closure, dim3(D.cfg.B), dim3(CT), 0, s, M, D, vl_dev, sw_dev, a);
}
