// lbs_dense_lab.h -- LAB BUILD ONLY (-DSFX_LAB): the 32-frames-per-wavefront form of the dense LBS kernel (rounds 1-3; bit for bit
// the results of the product kernel k_lbs_dense16, within 2 % of it at 97-128 active frames and slower everywhere else) for the
// interchangeability test and A/B measurements (sfx_debug_lbs_dense_form(32), SFX_LBS_DENSE=32), and the -DMF32 diagnostic kernel.
// Included by lbs_dense.hip.  Work unit = one wavefront = 16 vertices x 32 frames (two 16x16 MFMA tiles): 6 accumulators
// (x,y,z) x 4 registers for v_posed, then per output row 4 x 2 accumulators of T fused with the skinning epilogue; workgroup =
// 4 wavefronts = 16 vertices x 128 frames.
#pragma once

struct __align__(16) DenseLDS {
    float a[2][FB][LDK];     // [frame][k]: written as float4 along k (the global layout), read one (frame, k) per lane
    float b[2][KC][LDB];
};

__global__ __launch_bounds__(DT, MINW)
void k_lbs_dense(DevModel M, BatchDev D) {
    __shared__ DenseLDS S;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    // XCD-aware mapping of the 1-D grid: workgroups are dealt round-robin to the 8 XCDs, so block
    // id L runs on XCD L % 8.  Each XCD owns a contiguous range of vertex tiles and visits
    // (tile, frame block) pairs with the frame block fastest: the frame blocks of one tile and the
    // neighbouring tiles (whose 192-byte dirs rows share 128-byte lines) meet in the same L2.
    int tile, fblk, fpb;
    {
        const int ny = (D.nact + FB - 1) / FB, ntile = (M.V + VB - 1) / VB;
        const int tpx = (ntile + 7) / 8;                        // tiles per XCD
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        tile = xcd * tpx + slot / ny; fblk = slot % ny;
        if (tile >= ntile) return;                              // padding of the last XCD's range
        // frames are dealt evenly to the ny frame blocks in 32-frame wavefront slices, so that a
        // batch of 160 active frames runs as 96 + 64 rather than 128 + 32
        fpb = 32 * (((D.nact + 31) / 32 + ny - 1) / ny);
    }
    const int v0 = tile * VB;
    const int fb0 = fblk * fpb;
    const int b0 = fb0 + wv * 32;
    const int jl = lane & 15, kq = lane >> 4;
    const int V = M.V, B = D.nact;
    const int vtx = v0 + jl;
    const int v = vtx < V ? vtx : V - 1;
    const size_t Bp = (size_t)D.Bpad;
    const size_t LD = (size_t)3 * M.Vpad;
    const bool active = wv * 32 < fpb && b0 < B;        // wave-uniform: this wavefront's 32 frames exist

    // staging: 1024 (feat) + 384 (dirs) float4 per chunk: slots tid + q*256; q = 0..3 -> feat rows,
    // slot 4 -> dirs, slot 5 (tid < 128) -> dirs
    const float4* gA = reinterpret_cast<const float4*>(D.featR + (size_t)fb0 * SFX_KD_PAD);
    // dirs tile-major ([tile of 16 vertices][k][48]: the 98 KB a workgroup streams are ONE contiguous block and a chunk is
    // 6 KB of consecutive float4 -- out of the k-major matrix it was 32 separate 192-byte pieces 240 KB apart per chunk)
    const float4* gB = reinterpret_cast<const float4*>(M.dirs_tiled + (size_t)tile * SFX_KD_PAD * NB3);
    const int stepA = KC / 4, stepB = KC * (NB3 / 4);
    int gA_off[4], lA_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {       // thread -> (frame, 4 consecutive k): 8 threads cover the 128 bytes a frame contributes to a chunk
        const int idx = tid + q * DT, f = idx / (KC / 4), k4 = idx % (KC / 4);
        gA_off[q] = f * (SFX_KD_PAD / 4) + k4; lA_off[q] = f * LDK + k4 * 4;
    }
    const int i4 = tid, i5 = tid + DT;
    const bool ok5 = i5 < B4;
    const int gB4 = i4, lB4 = (i4 / (NB3 / 4)) * LDB + (i4 % (NB3 / 4)) * 4;
    const int j5 = ok5 ? i5 : 0;
    const int gB5 = j5, lB5 = (j5 / (NB3 / 4)) * LDB + (j5 % (NB3 / 4)) * 4;
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

    // the accumulators collect the 506 blend-shape terms alone (millimetres: their fp32 chain carries ~1e-10 m);
    // v_template is added once at the end -- started from the template (~0.5 m) the chain rounds 506 times at
    // 3e-8 m and v_posed ends 2e-7 m off (measured, tests/probe_drift.py)
    f32x4 ax0 = {0, 0, 0, 0}, ay0 = ax0, az0 = ax0, ax1 = ax0, ay1 = ax0, az1 = ax0;       // frames 0-15 / 16-31 of this wavefront's slice
    const float tx = M.v_template[v * 3], ty = M.v_template[v * 3 + 1], tz = M.v_template[v * 3 + 2];
    // K order: chunks 1, 2, ..., 15, 0 -- the 486 pose-corrective terms (1e-4 m each) are summed first and the 20
    // shape / expression terms (centimetres, rows 0-19 of chunk 0) last: added to a centimetre-sized partial sum,
    // every one of the small terms would round at 2e-9 m (2.5e-8 m over the chain, measured on the keypoints)
    constexpr int NCHUNK = SFX_KD_PAD / KC;
    STAGE_LOAD(KCHUNK(0));
    STAGE_WRITE(0);
    __syncthreads();

    for (int c = 0; c < NCHUNK; ++c) {
        const int cur = c & 1;
        if (c + 1 < NCHUNK) STAGE_LOAD(KCHUNK(c + 1));
        if (active) {
            const float* sa = &S.a[cur][wv * 32 + jl][kq];
            const float* sb = &S.b[cur][kq][jl * 3];
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                const float a0 = sa[ks * 4], a1 = sa[16 * LDK + ks * 4];
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
    const int us = vok ? M.vslot[vtx] : -1;      // export index of an item vertex
    if (us >= 0) {      // keypoint vertex: its blend offsets go to the loss / adjoint pass (a fraction of a percent of the lanes)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = b0 + kq * 4 + r, f1 = f0 + 16;
            if (f0 < B) { float* o = D.uvp + ((size_t)f0 * M.n_uniq + us) * 3; o[0] = ax0[r]; o[1] = ay0[r]; o[2] = az0[r]; }
            if (f1 < B) { float* o = D.uvp + ((size_t)f1 * M.n_uniq + us) * 3; o[0] = ax1[r]; o[1] = ay1[r]; o[2] = az1[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { ax0[r] += tx; ay0[r] += ty; az0[r] += tz; ax1[r] += tx; ay1[r] += ty; az1[r] += tz; }
    // skinning GEMM T = W . A restricted to the joints that carry weight in this 16-vertex tile
    // (exact: the skipped products are structural zeros of lbs_weights; ascending joint order kept).
    // The A-operand gathers of step (rr, js+1) are in flight while step (rr, js) multiplies.
    const int njs = M.tj_n[tile] >> 2;
    const int* jl4 = M.tj_list + (size_t)tile * SFX_JPAD + kq;
    const float* wl = M.tj_w + ((size_t)tile * SFX_JPAD + kq) * 16 + jl;
    const float* atb = D.AT + b0 + jl;
    const size_t estep = (size_t)SFX_JPAD * Bp;
    float P0, P1, P2, P3, Q0, Q1, Q2, Q3, N0, N1, N2, N3, R0, R1, R2, R3, wc, wn;
#define AT_LOAD(p0, p1, p2, p3, q0, q1, q2, q3, ww, rr, js) do {                                          \
        const float* at_ = atb + ((size_t)((rr) * 4) * SFX_JPAD + jl4[(js) * 4]) * Bp;                    \
        p0 = at_[0]; p1 = at_[estep]; p2 = at_[2 * estep]; p3 = at_[3 * estep];                           \
        q0 = at_[16]; q1 = at_[estep + 16]; q2 = at_[2 * estep + 16]; q3 = at_[3 * estep + 16];           \
        ww = wl[(js) * 64]; } while (0)
    float o0[4][3], o1[4][3];
    AT_LOAD(P0, P1, P2, P3, Q0, Q1, Q2, Q3, wc, 0, 0);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        f32x4 t00 = {0, 0, 0, 0}, t01 = t00, t02 = t00, t03 = t00, t10 = t00, t11 = t00, t12 = t00, t13 = t00;
        for (int js = 0; js < njs; ++js) {
            if (js + 1 < njs) AT_LOAD(N0, N1, N2, N3, R0, R1, R2, R3, wn, rr, js + 1);
            else if (rr < 2) AT_LOAD(N0, N1, N2, N3, R0, R1, R2, R3, wn, rr + 1, 0);
            t00 = MFMA(P0, wc, t00); t01 = MFMA(P1, wc, t01); t02 = MFMA(P2, wc, t02); t03 = MFMA(P3, wc, t03);
            t10 = MFMA(Q0, wc, t10); t11 = MFMA(Q1, wc, t11); t12 = MFMA(Q2, wc, t12); t13 = MFMA(Q3, wc, t13);
            P0 = N0; P1 = N1; P2 = N2; P3 = N3; Q0 = R0; Q1 = R1; Q2 = R2; Q3 = R3; wc = wn;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o0[r][rr] = t00[r] * ax0[r] + t01[r] * ay0[r] + t02[r] * az0[r] + t03[r];
            o1[r][rr] = t10[r] * ax1[r] + t11[r] * ay1[r] + t12[r] * az1[r] + t13[r];
        }
    }
#undef AT_LOAD
    if (vok && D.vposed) {      // interpenetration on: the adjoint of a gradient on every vertex needs every v_posed
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = b0 + kq * 4 + r, f1 = f0 + 16;
            if (f0 < B) { float* o = D.vposed + ((size_t)f0 * V + vtx) * 3; o[0] = ax0[r]; o[1] = ay0[r]; o[2] = az0[r]; }
            if (f1 < B) { float* o = D.vposed + ((size_t)f1 * V + vtx) * 3; o[0] = ax1[r]; o[1] = ay1[r]; o[2] = az1[r]; }
        }
    }
    if (vok) {      // one 12-byte store per (frame, vertex): 16 lanes cover 192 contiguous bytes
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = b0 + kq * 4 + r, f1 = f0 + 16;
            if (f0 < B) { float* o = D.verts + ((size_t)f0 * V + vtx) * 3; o[0] = o0[r][0]; o[1] = o0[r][1]; o[2] = o0[r][2]; }
            if (f1 < B) { float* o = D.verts + ((size_t)f1 * V + vtx) * 3; o[0] = o1[r][0]; o[1] = o1[r][1]; o[2] = o1[r][2]; }
        }
    }
}


#ifdef MF32
// DIAGNOSTIC build only (tools/build_variant.sh mf32 lbs_dense -DMF32 -DNO_T): the K loop of the blend-shape GEMM on
// v_mfma_f32_32x32x2_f32 -- wavefront tile 32 vertices x 32 frames, 3 x 16 accumulator registers, per K step of two rows
// 1 A read + 3 B reads for 3 MFMAs -- against the same loop of k_lbs_dense built with -DNO_T (no skinning epilogue, a
// checksum store).  Measured and rejected: LAB_NOTES.md §4.5.  A operand staged k-major ([k][frame], +1 pad) so that the 32
// frames of an operand fetch hit 32 banks.
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define VB2 32
#define LDA2 (FB + 1)
struct __align__(16) DenseLDS32 {
    float a[2][KC][LDA2];
    float b[2][KC][VB2 * 3];
};
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
__global__ __launch_bounds__(DT, MINW)
void k_lbs_dense32(DevModel M, BatchDev D) {
    __shared__ DenseLDS32 S;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int tile, fblk, fpb;
    {
        const int ny = (D.nact + FB - 1) / FB, ntile = (M.V + VB2 - 1) / VB2;
        const int tpx = (ntile + 7) / 8;
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        tile = xcd * tpx + slot / ny; fblk = slot % ny;
        if (tile >= ntile) return;
        fpb = 32 * (((D.nact + 31) / 32 + ny - 1) / ny);
    }
    const int v0 = tile * VB2, fb0 = fblk * fpb, b0 = fb0 + wv * 32;
    const int il = lane & 31, kh = lane >> 5;
    const int V = M.V, B = D.nact;
    const size_t LD = (size_t)3 * M.Vpad;
    const bool active = wv * 32 < fpb && b0 < B;
    const float4* gA = reinterpret_cast<const float4*>(D.featR + (size_t)fb0 * SFX_KD_PAD);
    const float4* gB = reinterpret_cast<const float4*>(M.dirs + (size_t)v0 * 3);
    const int LD4 = (int)(LD / 4);
    const int stepA = KC / 4, stepB = KC * LD4;
    int gA_off[4], fA[4], kA[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + q * DT, f = idx / (KC / 4), k4 = idx % (KC / 4);
        gA_off[q] = f * (SFX_KD_PAD / 4) + k4; fA[q] = f; kA[q] = k4 * 4;
    }
    // dirs chunk: KC rows x 96 floats = 24 float4 per row -> 768 float4: 3 per thread
    int gBo[3], lBo[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { const int idx = tid + q * DT; gBo[q] = (idx / 24) * LD4 + idx % 24; lBo[q] = (idx / 24) * (VB2 * 3) + (idx % 24) * 4; }
    float4 s0, s1, s2, s3, t0, t1, t2;
#define SL32(c) do { s0 = gA[gA_off[0] + (c) * stepA]; s1 = gA[gA_off[1] + (c) * stepA]; s2 = gA[gA_off[2] + (c) * stepA];   \
        s3 = gA[gA_off[3] + (c) * stepA]; t0 = gB[gBo[0] + (c) * stepB]; t1 = gB[gBo[1] + (c) * stepB]; t2 = gB[gBo[2] + (c) * stepB]; } while (0)
#define SWA(buf, q, sv) do { float* p_ = &S.a[buf][kA[q]][fA[q]]; p_[0] = sv.x; p_[LDA2] = sv.y; p_[2 * LDA2] = sv.z; p_[3 * LDA2] = sv.w; } while (0)
#define SW32(buf) do { SWA(buf, 0, s0); SWA(buf, 1, s1); SWA(buf, 2, s2); SWA(buf, 3, s3);                                    \
        *reinterpret_cast<float4*>(&S.b[buf][0][0] + lBo[0]) = t0; *reinterpret_cast<float4*>(&S.b[buf][0][0] + lBo[1]) = t1; \
        *reinterpret_cast<float4*>(&S.b[buf][0][0] + lBo[2]) = t2; } while (0)
    f32x16 ax = {0}, ay = {0}, az = {0};
    constexpr int NCHUNK = SFX_KD_PAD / KC;
    SL32(KCHUNK(0)); SW32(0);
    __syncthreads();
    for (int c = 0; c < NCHUNK; ++c) {
        const int cur = c & 1;
        if (c + 1 < NCHUNK) SL32(KCHUNK(c + 1));
        if (active) {
            const float* sa = &S.a[cur][kh][wv * 32 + il];
            const float* sb = &S.b[cur][kh][il * 3];
#pragma unroll
            for (int ks = 0; ks < KC / 2; ++ks) {
                const float a0 = sa[ks * 2 * LDA2];
                const float bx = sb[ks * 2 * VB2 * 3], by = sb[ks * 2 * VB2 * 3 + 1], bz = sb[ks * 2 * VB2 * 3 + 2];
                ax = MFMA32(a0, bx, ax); ay = MFMA32(a0, by, ay); az = MFMA32(a0, bz, az);
            }
        }
        if (c + 1 < NCHUNK) SW32(cur ^ 1);
        __syncthreads();
    }
    if (!active) return;
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum += ax[r] + ay[r] + az[r];
    const int vtx = v0 + il;
    if (vtx < V && b0 < B) D.verts[((size_t)b0 * V + vtx) * 3 + kh] = sum;
}
#endif
