// lbs_dense.hip -- dense SMPL-X linear blend skinning for a batch of frames on the
// fp32 matrix cores (v_mfma_f32_16x16x4_f32; exact f32 = a k-ordered fmaf chain).
//
// Replaces the vertex half of smplx.lbs.lbs (external package):
//     v_posed = v_template + [betas|expr|pose_feature] . [shapedirs|posedirs]   (K = 506)
//     T       = lbs_weights . A                                                   (K = 55)
//     verts   = T[:3,:3] v_posed + T[:3,3]
// as two GEMMs sharing one output tile:  rows = frames (M), cols = vertices (N).
//   A operand  featR[b][k]      (one 2-KiB row per frame, written by the tick kernel's export pass; entries
//                                >= 506 are zero); transposed to [k-chunk][frame] order while staged into LDS
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
#include <cstdlib>
#include <algorithm>

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
#define LDK (KC + 4)     // padded K stride of a frame's row in LDS: the 16 frames x 4 K rows of one MFMA operand fetch hit 64 distinct banks
#define LDB NB3
#define B4 (KC * NB3 / 4)      // float4 per dirs chunk (384)

#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// K order of every kernel below: chunks 1, 2, ..., 15, 0 -- the 486 pose-corrective terms (1e-4 m each) are summed first and the 20
// shape / expression terms (centimetres, rows 0-19 of chunk 0) last: added to a centimetre-sized partial sum, every one of the
// small terms would round at 2e-9 m (2.5e-8 m over the chain, measured on the keypoints)
#define KCHUNK(c) (((c) + 1) % NCHUNK)

#ifdef SFX_LAB
#include "lbs_dense_lab.h"
#endif

// ---------------------------------------------------------------------------------------------------------------------
// k_lbs_dense16: the same kernel with 16 instead of 32 frames per wavefront (one 16 x 16 MFMA tile per coordinate): work
// units of 434 instead of 870 MFMAs, 64 frames per workgroup, 90 registers.  Every (vertex, frame) goes through exactly the
// chain of fp32 operations it goes through in k_lbs_dense (rows of an MFMA tile are independent; same K order, same
// skinning), so the two kernels are interchangeable bit for bit and the launcher picks by the number of active frames:
// this one where the batch is small or fills k_lbs_dense's 128-frame blocks badly (<= 64 frames: the launch floor, set by one
// wavefront's MFMA chain, drops from 29 to ~20 us; 129-192 frames: 74 instead of 91 us), k_lbs_dense where its fewer LDS
// operand reads per MFMA pay (it reads 5 operands per 6 MFMAs, this kernel 4 per 3).
// (Measured and dropped on the way: a K-split pair of wavefronts per 16 x 32 tile -- halves of the K range summed as
//  (half 0) + (half 1) after an exchange through LDS: faster below 64 and at 129-192 frames, 4-8 % slower at 256-1024, and NOT
//  interchangeable with k_lbs_dense bit for bit, so it could not be used for part of the batch sizes only.)
// Round 4: the number of wavefronts per workgroup is a template parameter (16 W frames per workgroup, 64 W threads; W = 3, 4, 5
// are instantiated, the launcher picks: launch_lbs_dense).  A wavefront's arithmetic does not depend on W: interchangeable
// bit for bit.
#define FB3 64            // frames per workgroup at W = 4 (the shape of round 3)
template <int W>
struct __align__(16) DenseLDS16 {
    float a[2][16 * W][LDK];
    float b[2][KC][LDB];
};
#ifndef MINW16
#define MINW16 4
#endif
template <int W>
__global__ __launch_bounds__(64 * W, MINW16)
void k_lbs_dense16(DevModel M, BatchDev D) {
    constexpr int DTW = 64 * W, FBW = 16 * W;
    static_assert(W >= 3 && W <= 8, "staging: two rounds of DTW threads cover the 384 float4 of a dirs chunk");
    __shared__ DenseLDS16<W> S;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6;
    int tile, fblk, fpb;
    {
        const int ny = (D.nact + FBW - 1) / FBW, ntile = (M.V + VB - 1) / VB;
        const int tpx = (ntile + 7) / 8;                        // tiles per XCD
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        tile = xcd * tpx + slot / ny; fblk = slot % ny;
        if (tile >= ntile) return;
        fpb = 16 * (((D.nact + 15) / 16 + ny - 1) / ny);        // frames dealt evenly to the frame blocks in 16-frame slices
    }
    const int v0 = tile * VB;
    const int fb0 = fblk * fpb;
    const int b0 = fb0 + wv * 16;
    const int jl = lane & 15, kq = lane >> 4;
    const int V = M.V, B = D.nact;
    const int vtx = v0 + jl;
    const int v = vtx < V ? vtx : V - 1;
    const size_t Bp = (size_t)D.Bpad;
    const size_t LD = (size_t)3 * M.Vpad;
    const bool active = wv * 16 < fpb && b0 < B;
    // staging: 128 W (feat: two per thread) + 384 (dirs) float4 per chunk
    const float4* gA = reinterpret_cast<const float4*>(D.featR + (size_t)fb0 * SFX_KD_PAD);
    // dirs tile-major ([tile of 16 vertices][k][48]: the 98 KB a workgroup streams are ONE contiguous block and a chunk is
    // 6 KB of consecutive float4 -- out of the k-major matrix it was 32 separate 192-byte pieces 240 KB apart per chunk)
    const float4* gB = reinterpret_cast<const float4*>(M.dirs_tiled + (size_t)tile * SFX_KD_PAD * NB3);
    const int stepA = KC / 4, stepB = KC * (NB3 / 4);
    int gA_off[2], lA_off[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int idx = tid + q * DTW, f = idx / (KC / 4), k4 = idx % (KC / 4);
        // (the last frame block may reach past the padded batch when 16 W does not divide it: those rows feed lanes whose
        //  results are not stored -- read the last row instead)
        const int fr = min(fb0 + f, D.Bpad - 1) - fb0;
        gA_off[q] = fr * (SFX_KD_PAD / 4) + k4; lA_off[q] = f * LDK + k4 * 4;
    }
    const bool ok4 = tid < B4;                              // (W = 7, 8: more threads than float4 in a dirs chunk)
    const int i4 = ok4 ? tid : 0, i5 = tid + DTW;
    const bool ok5 = i5 < B4;
    const int gB4 = i4, lB4 = (i4 / (NB3 / 4)) * LDB + (i4 % (NB3 / 4)) * 4;
    const int j5 = ok5 ? i5 : 0;
    const int gB5 = j5, lB5 = (j5 / (NB3 / 4)) * LDB + (j5 % (NB3 / 4)) * 4;
    float4 s0, s1, s4, s5;
#define ST3_LOAD(c) do { s0 = gA[gA_off[0] + (c) * stepA]; s1 = gA[gA_off[1] + (c) * stepA];            \
        s4 = gB[gB4 + (c) * stepB]; s5 = gB[gB5 + (c) * stepB]; } while (0)
#define ST3_WRITE(buf) do {                                                                            \
        *reinterpret_cast<float4*>(&S.a[buf][0][0] + lA_off[0]) = s0;                                  \
        *reinterpret_cast<float4*>(&S.a[buf][0][0] + lA_off[1]) = s1;                                  \
        if (W <= 6 || ok4) *reinterpret_cast<float4*>(&S.b[buf][0][0] + lB4) = s4;                     \
        if (ok5) *reinterpret_cast<float4*>(&S.b[buf][0][0] + lB5) = s5; } while (0)
    f32x4 ax0 = {0, 0, 0, 0}, ay0 = ax0, az0 = ax0;
    const float tx = M.v_template[v * 3], ty = M.v_template[v * 3 + 1], tz = M.v_template[v * 3 + 2];
    constexpr int NCHUNK = SFX_KD_PAD / KC;
    ST3_LOAD(KCHUNK(0));
    ST3_WRITE(0);
    __syncthreads();
    for (int c = 0; c < NCHUNK; ++c) {
        const int cur = c & 1;
        if (c + 1 < NCHUNK) ST3_LOAD(KCHUNK(c + 1));
        if (active) {
            const float* sa = &S.a[cur][wv * 16 + jl][kq];
            const float* sb = &S.b[cur][kq][jl * 3];
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) {
                const float a0 = sa[ks * 4];
                const float bx = sb[ks * 4 * LDB], by = sb[ks * 4 * LDB + 1], bz = sb[ks * 4 * LDB + 2];
                ax0 = MFMA(a0, bx, ax0); ay0 = MFMA(a0, by, ay0); az0 = MFMA(a0, bz, az0);
            }
        }
        if (c + 1 < NCHUNK) ST3_WRITE(cur ^ 1);
        __syncthreads();
    }
#undef ST3_LOAD
#undef ST3_WRITE
    if (!active) return;
    const bool vok = vtx < V;
    const int us = vok ? M.vslot[vtx] : -1;      // export index of an item vertex
    if (us >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = b0 + kq * 4 + r;
            if (f0 < B) { float* o = D.uvp + ((size_t)f0 * M.n_uniq + us) * 3; o[0] = ax0[r]; o[1] = ay0[r]; o[2] = az0[r]; }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { ax0[r] += tx; ay0[r] += ty; az0[r] += tz; }
    const int njs = M.tj_n[tile] >> 2;
    const int* jl4 = M.tj_list + (size_t)tile * SFX_JPAD + kq;
    const float* wl = M.tj_w + ((size_t)tile * SFX_JPAD + kq) * 16 + jl;
    const float* atb = D.AT + b0 + jl;
    const size_t estep = (size_t)SFX_JPAD * Bp;
    float P0, P1, P2, P3, N0, N1, N2, N3, wc, wn;
#define AT3_LOAD(p0, p1, p2, p3, ww, rr, js) do {                                                            \
        const float* at_ = atb + ((size_t)((rr) * 4) * SFX_JPAD + jl4[(js) * 4]) * Bp;                        \
        p0 = at_[0]; p1 = at_[estep]; p2 = at_[2 * estep]; p3 = at_[3 * estep];                               \
        ww = wl[(js) * 64]; } while (0)
    float o0[4][3];
    AT3_LOAD(P0, P1, P2, P3, wc, 0, 0);
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        f32x4 t00 = {0, 0, 0, 0}, t01 = t00, t02 = t00, t03 = t00;
        for (int js = 0; js < njs; ++js) {
            if (js + 1 < njs) AT3_LOAD(N0, N1, N2, N3, wn, rr, js + 1);
            else if (rr < 2) AT3_LOAD(N0, N1, N2, N3, wn, rr + 1, 0);
            t00 = MFMA(P0, wc, t00); t01 = MFMA(P1, wc, t01); t02 = MFMA(P2, wc, t02); t03 = MFMA(P3, wc, t03);
            P0 = N0; P1 = N1; P2 = N2; P3 = N3; wc = wn;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) o0[r][rr] = t00[r] * ax0[r] + t01[r] * ay0[r] + t02[r] * az0[r] + t03[r];
    }
#undef AT3_LOAD
    if (vok && D.vposed) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = b0 + kq * 4 + r;
            if (f0 < B) { float* o = D.vposed + ((size_t)f0 * V + vtx) * 3; o[0] = ax0[r]; o[1] = ay0[r]; o[2] = az0[r]; }
        }
    }
    if (vok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = b0 + kq * 4 + r;
            if (f0 < B) { float* o = D.verts + ((size_t)f0 * V + vtx) * 3; o[0] = o0[r][0]; o[1] = o0[r][1]; o[2] = o0[r][2]; }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// k_lbs_dense16c (round 5): the launch FLOOR.  At <= 32 active frames -- the tail of every fit: the frames still running when most
// have finished -- k_lbs_dense16 has one or two wavefronts per workgroup doing all three coordinates of their 16 x 16 tile: 24
// dependent-issue MFMAs per chunk on one wavefront while the workgroup's others only stage (20-21 us per launch whatever the
// number of frames).  Here the three coordinates of a slice go to three wavefronts (6 wavefronts = 2 slices x 3 coordinates per
// workgroup): every accumulator is its own MFMA chain over the same operands in the same K order, the skinning rows likewise
// (wavefront c forms output row c from the three blend-shape accumulators, exchanged through LDS), so a (vertex, frame) goes
// through exactly the fp32 operations of k_lbs_dense16 -- interchangeable bit for bit
// (test_the_two_dense_kernels_are_interchangeable_bit_for_bit).
struct __align__(16) DenseLDS16c {
    float a[2][32][LDK];
    float b[2][KC][LDB];
    float x[2][3][64][4];        // [slice][coordinate][lane][register]: v_posed accumulators on their way to the skinning rows
};
__global__ __launch_bounds__(384, 2)
void k_lbs_dense16c(DevModel M, BatchDev D) {
    __shared__ DenseLDS16c S;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wv = tid >> 6, sl = wv / 3, co = wv % 3;
    int tile;
    {
        const int ntile = (M.V + VB - 1) / VB, tpx = (ntile + 7) / 8;
        const int L = blockIdx.x, xcd = L & 7, slot = L >> 3;
        tile = xcd * tpx + slot;
        if (tile >= ntile || slot >= tpx) return;
    }
    const int v0 = tile * VB, b0 = sl * 16;
    const int jl = lane & 15, kq = lane >> 4;
    const int V = M.V, B = D.nact;
    const int vtx = v0 + jl;
    const int v = vtx < V ? vtx : V - 1;
    const size_t Bp = (size_t)D.Bpad;
    const bool active = b0 < B;
    const float4* gA = reinterpret_cast<const float4*>(D.featR);
    const float4* gB = reinterpret_cast<const float4*>(M.dirs_tiled + (size_t)tile * SFX_KD_PAD * NB3);
    const int stepA = KC / 4, stepB = KC * (NB3 / 4);
    // staging: 384 float4 of dirs (one per thread) + 256 of feat (threads < 256) per chunk
    const bool okA = tid < 32 * (KC / 4);
    const int ia = okA ? tid : 0, fa = ia / (KC / 4), ka = ia % (KC / 4);
    const int gAo = (min(fa, D.Bpad - 1)) * (SFX_KD_PAD / 4) + ka, lAo = fa * LDK + ka * 4;
    const int gBo = tid, lBo = (tid / (NB3 / 4)) * LDB + (tid % (NB3 / 4)) * 4;
    float4 sA, sB;
#define STC_LOAD(c) do { sA = gA[gAo + (c) * stepA]; sB = gB[gBo + (c) * stepB]; } while (0)
#define STC_WRITE(buf) do { if (okA) *reinterpret_cast<float4*>(&S.a[buf][0][0] + lAo) = sA;           \
        *reinterpret_cast<float4*>(&S.b[buf][0][0] + lBo) = sB; } while (0)
    f32x4 acc = {0, 0, 0, 0};
    const float tc = M.v_template[v * 3 + co];
    constexpr int NCHUNK = SFX_KD_PAD / KC;
    STC_LOAD(KCHUNK(0));
    STC_WRITE(0);
    __syncthreads();
    for (int c = 0; c < NCHUNK; ++c) {
        const int cur = c & 1;
        if (c + 1 < NCHUNK) STC_LOAD(KCHUNK(c + 1));
        if (active) {
            const float* sa = &S.a[cur][sl * 16 + jl][kq];
            const float* sb = &S.b[cur][kq][jl * 3 + co];
#pragma unroll
            for (int ks = 0; ks < KC / 4; ++ks) acc = MFMA(sa[ks * 4], sb[ks * 4 * LDB], acc);
        }
        if (c + 1 < NCHUNK) STC_WRITE(cur ^ 1);
        __syncthreads();
    }
#undef STC_LOAD
#undef STC_WRITE
    const bool vok = vtx < V;
    if (active) {
        const int us = vok ? M.vslot[vtx] : -1;      // export index of an item vertex
        if (us >= 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int f0 = b0 + kq * 4 + r;
                if (f0 < B) D.uvp[((size_t)f0 * M.n_uniq + us) * 3 + co] = acc[r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += tc;
        *reinterpret_cast<f32x4*>(&S.x[sl][co][lane][0]) = acc;
    }
    __syncthreads();
    if (!active) return;
    const f32x4 ax0 = *reinterpret_cast<const f32x4*>(&S.x[sl][0][lane][0]);
    const f32x4 ay0 = *reinterpret_cast<const f32x4*>(&S.x[sl][1][lane][0]);
    const f32x4 az0 = *reinterpret_cast<const f32x4*>(&S.x[sl][2][lane][0]);
    const int njs = M.tj_n[tile] >> 2;
    const int* jl4 = M.tj_list + (size_t)tile * SFX_JPAD + kq;
    const float* wl = M.tj_w + ((size_t)tile * SFX_JPAD + kq) * 16 + jl;
    const float* atb = D.AT + b0 + jl;
    const size_t estep = (size_t)SFX_JPAD * Bp;
    float P0, P1, P2, P3, N0, N1, N2, N3, wc, wn;
#define ATC_LOAD(p0, p1, p2, p3, ww, js) do {                                                                \
        const float* at_ = atb + ((size_t)(co * 4) * SFX_JPAD + jl4[(js) * 4]) * Bp;                          \
        p0 = at_[0]; p1 = at_[estep]; p2 = at_[2 * estep]; p3 = at_[3 * estep];                               \
        ww = wl[(js) * 64]; } while (0)
    ATC_LOAD(P0, P1, P2, P3, wc, 0);
    f32x4 t00 = {0, 0, 0, 0}, t01 = t00, t02 = t00, t03 = t00;
    for (int js = 0; js < njs; ++js) {
        if (js + 1 < njs) ATC_LOAD(N0, N1, N2, N3, wn, js + 1);
        t00 = MFMA(P0, wc, t00); t01 = MFMA(P1, wc, t01); t02 = MFMA(P2, wc, t02); t03 = MFMA(P3, wc, t03);
        P0 = N0; P1 = N1; P2 = N2; P3 = N3; wc = wn;
    }
#undef ATC_LOAD
    float o0[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o0[r] = t00[r] * ax0[r] + t01[r] * ay0[r] + t02[r] * az0[r] + t03[r];
    if (vok) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int f0 = b0 + kq * 4 + r;
            if (f0 < B) {
                if (D.vposed) D.vposed[((size_t)f0 * V + vtx) * 3 + co] = co == 0 ? ax0[r] : co == 1 ? ay0[r] : az0[r];
                D.verts[((size_t)f0 * V + vtx) * 3 + co] = o0[r];
            }
        }
    }
}


#ifdef SFX_LAB       // A/B switches of the lab build (include/sfx_lab.h): which kernel the rounds launch, one W for every launch
int g_lbs_dense_form = [] { const char* e = getenv("SFX_LBS_DENSE"); return e ? atoi(e) : 16; }();
static int g_lbs_dense_w = [] { const char* e = getenv("SFX_LBS_W"); const int v = e ? atoi(e) : 0; return (v >= 3 && v <= 5) ? v : 0; }();
#else
constexpr int g_lbs_dense_form = 16, g_lbs_dense_w = 0;
#endif

void launch_lbs_dense(const DevModel& M, const BatchDev& D, hipStream_t s) {
    if (D.nact <= 0) return;
#ifdef MF32
    {
        const int ny = (D.nact + FB - 1) / FB, ntile = (M.V + VB2 - 1) / VB2;
        dim3 grid(8 * ((ntile + 7) / 8) * ny);
        hipLaunchKernelGGL(k_lbs_dense32, grid, dim3(DT), 0, s, M, D);
        return;
    }
#endif
    // k_lbs_dense16 / k_lbs_dense16c are the product kernels; the lab build also holds k_lbs_dense (32 frames per wavefront, bit for
    // bit the same results: lbs_dense_lab.h)
#ifdef SFX_LAB
    if (g_lbs_dense_form == 32) {
        const int ny = (D.nact + FB - 1) / FB, ntile = (M.V + VB - 1) / VB;
        hipLaunchKernelGGL(k_lbs_dense, dim3(8 * ((ntile + 7) / 8) * ny), dim3(DT), 0, s, M, D);
        return;
    }
#endif
    {
        // frame blocks of at most 4 slices as in round 3, and the workgroup as wide as its busiest block: 6 and 9 slices are
        // blocks of 3 (W = 3: no idle wavefront, six workgroups per CU), 5 slices one block of 5; tools/bench_dense.py with
        // SFX_LBS_W, us per launch at 5 / 6 / 9 slices: W = 3: 44.2 / 45.1 / 60.1, W = 4: 47.6 / 48.5 / 65.5, W = 5: 41.2 / 55.3 / 74.1;
        // every other count is fastest (or as fast) at W = 4, and W = 6 .. 8 lose everywhere (measured, not instantiated)
        const int ns = (D.nact + 15) / 16, ny4 = (ns + 3) / 4;
        if (ns <= 2 && g_lbs_dense_form != 17 && !g_lbs_dense_w) {       // the tail of a fit: three coordinates on three wavefronts (same bits)
            const int ntile_ = (M.V + VB - 1) / VB;
            hipLaunchKernelGGL(k_lbs_dense16c, dim3(8 * ((ntile_ + 7) / 8)), dim3(384), 0, s, M, D);
            return;
        }
        int w = ns <= 4 ? 4 : (ns == 5 ? 5 : std::max(3, (ns + ny4 - 1) / ny4));
        if (g_lbs_dense_w) w = g_lbs_dense_w;
        const int ny = (ns + w - 1) / w, ntile = (M.V + VB - 1) / VB;
        dim3 grid(8 * ((ntile + 7) / 8) * ny);
        switch (w) {
        case 3: hipLaunchKernelGGL(k_lbs_dense16<3>, grid, dim3(192), 0, s, M, D); break;
        case 5: hipLaunchKernelGGL(k_lbs_dense16<5>, grid, dim3(320), 0, s, M, D); break;
        default: hipLaunchKernelGGL(k_lbs_dense16<4>, grid, dim3(256), 0, s, M, D); break;
        }
    }
}
