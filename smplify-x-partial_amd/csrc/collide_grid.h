// collide_grid.h -- broad phase, part 1: triangle boxes, part culling, the uniform grid of a mesh (k_pen_g1 / g2 / g3)
// Part of csrc/collide.hip (included there, in this order: collide_field.h, collide_grid.h, collide_pairs.h, collide_eval.h);
// one translation unit, compiled with -ffp-contract=off.
#pragma once

// ---------------------------------------------------------------------------------------------
// The uniform grid of a frame in three launches (one 1024-lane workgroup per frame walked its 21 triangles per lane through
// four dependent passes: 230-290 us on ONE compute unit per frame; its first pass alone -- 188 k scattered 4-byte vertex
// gathers through one compute unit's address unit -- took 100 us):
//   k_pen_g1  (PEN_GW workgroups per frame) triangle boxes, per-workgroup partial frame box / extent sum, part boxes (LDS
//             atomics per workgroup, merged with a few hundred global atomicMin / Max)
//   k_pen_g2  (PEN_GW workgroups per frame) frame box + cell size from the partials (every workgroup, same fixed order),
//             part culling, packed cell range of every surviving triangle
//   k_pen_g3  (one workgroup per frame) bucket part masks, histogram, scan, scatter on LDS atomics
// Cross-workgroup results are order-independent (min / max) or combined in index order (extent sum).
#define PEN_GW 8                // workgroups per frame in k_pen_g1 / g2
#define PEN_GU 3                // triangles per lane of those kernels: ceil(F / (PEN_GW * PEN_T)) for F <= 24576; more loop
__device__ __forceinline__ int pen_ford(float x) { int i = __float_as_int(x); return i ^ ((i >> 31) & 0x7fffffff); }      // order-preserving
// The PEN_GW workgroups of a column on ONE XCD (round 5).  Workgroup i of a launch, x fastest, runs on XCD i % 8 (observed; a matter
// of speed only): with (w, b) = blockIdx the eight workgroups of a column sat on eight XCDs, each of whose L2s fetched the column's
// vertices for its share of the triangles (k_pen_g1 read 5x the vertices' bytes from HBM).  Here a group of 64 consecutive
// workgroups serves 8 columns, column = group * 8 + (i % 8); the launch has a multiple of 8 rows, nb: the real column count.
__device__ __forceinline__ bool pen_gw_map(const int nb, int& b, int& w) {
    static_assert(PEN_GW == 8, "one workgroup of a column per slot of an XCD group");
    const int lin = blockIdx.y * PEN_GW + blockIdx.x;
    b = (lin >> 6) * 8 + (lin & 7); w = (lin >> 3) & 7;
    return b < nb;
}
__device__ __forceinline__ int pen_bucket(int x, int y, int z) {
    return (int)(((unsigned)x * 73856093u ^ (unsigned)y * 19349663u ^ (unsigned)z * 83492791u) & (PEN_CELLS - 1)); }

// zero_dverts / zero_G (lab forms 1 / 2, k_pen_narrow): the per-column kernel writes the gradient of the vertices that HAVE one (a few
// hundred of 10 475 on a body); the other rows of d loss / d vertices and of the adjoint GEMM's operand are zeroed here, by the
// launch that has eight workgroups per column and nothing else to write but the boxes.
#ifndef PEN_G1_OCC
#define PEN_G1_OCC 1
#endif
__global__ __launch_bounds__(PEN_T, PEN_G1_OCC)
void k_pen_g1(PenDev P, const float* __restrict__ verts, const int* __restrict__ want, float* __restrict__ zero_dverts,
              float* __restrict__ zero_G, int Vpad, int nb, float* __restrict__ loss_out) {
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { P.callno[0] += 1; P.nheavy[0] = 0; P.ovm[0] = 0; }      // (one writer per launch; launches of a handle are ordered)
    // (round 5) the columns that carry the term, as a list for the launches behind the grid build (P.wl / P.nw): their rows loop
    // over it, where a grid row per ACTIVE column sent two workgroups in three through a load and out again (76 of 119 columns
    // want nothing in an average round: ~10 k workgroups per launch of the pair tests)
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64 && want) {
        const int lane = threadIdx.x;
        int cnt = 0;
        for (int base = 0; base < nb; base += 64) {
            const int c = base + lane;
            const bool w_ = c < nb && want[c] != 0;
            const unsigned long long m = __ballot(w_);
            if (w_) P.wl[cnt + __popcll(m & ((1ull << lane) - 1ull))] = c;
            cnt += __popcll(m);
        }
        if (lane == 0) P.nw[0] = cnt;
    }
    __shared__ float red[PEN_T / 64];
    __shared__ int pbox[64 * 6];               // this workgroup's part boxes (LDS atomics), merged into the frame's afterwards: atomics
                                               // straight to the frame's 12 cache lines serialise in L2 (measured: 0.4-1.5 ms)
    const int t = threadIdx.x;
    int b, w;
    if (!pen_gw_map(nb, b, w)) return;
    if (want && !want[b]) { if (w == 0 && t == 0 && loss_out) loss_out[b] = 0.f; return; }      // (what the gather's row of such a column wrote)
    if (zero_dverts) {
        float* d = zero_dverts + (size_t)b * P.V * 3;
        for (int i = w * PEN_T + t; i < P.V * 3; i += PEN_GW * PEN_T) d[i] = 0.f;
        if (zero_G) { float* g = zero_G + (size_t)b * 3 * Vpad; for (int i = w * PEN_T + t; i < P.V * 3; i += PEN_GW * PEN_T) g[i] = 0.f; }
    }
    const float* vb = verts + (size_t)b * P.V * 3;
    float* aabb = P.aabb + (size_t)b * P.F * 6;
    const int F = P.F;
    if (t < 64 * 6) pbox[t] = (t % 6) < 3 ? 0x7fffffff : (int)0x80000000;
    __syncthreads();
    float lo[3] = {3e38f, 3e38f, 3e38f}, hi[3] = {-3e38f, -3e38f, -3e38f}, ext_sum = 0.f;
    for (int fw = w * PEN_T + (t & ~63); fw < F; fw += PEN_GW * PEN_T * PEN_GU) {        // (wave-uniform trip count: wave reductions inside)
        const int f0 = fw + (t & 63);
        int vid[PEN_GU][3], seg[PEN_GU];
#pragma unroll
        for (int u = 0; u < PEN_GU; ++u) {
            const int f = f0 + u * PEN_GW * PEN_T, ff = f < F ? f : 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) vid[u][k] = P.faces[ff * 3 + k];
            seg[u] = P.segm[ff];
            if (f < F) P.pcount[(size_t)b * F + f] = 0;
        }
        float px[PEN_GU][9];
#pragma unroll
        for (int u = 0; u < PEN_GU; ++u)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const float* p = vb + (size_t)vid[u][k] * 3;
                px[u][k * 3] = p[0]; px[u][k * 3 + 1] = p[1]; px[u][k * 3 + 2] = p[2];
            }
#pragma unroll
        for (int u = 0; u < PEN_GU; ++u) {
            const int f = f0 + u * PEN_GW * PEN_T;
            const bool valid = f < F;
            if (!__ballot(valid)) continue;            // (wave-uniform: the reductions below need every lane)
            float a[3], c[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                a[e] = valid ? fminf(fminf(px[u][e], px[u][3 + e]), px[u][6 + e]) : 3e38f;
                c[e] = valid ? fmaxf(fmaxf(px[u][e], px[u][3 + e]), px[u][6 + e]) : -3e38f;
                if (valid) { aabb[f * 6 + e] = a[e]; aabb[f * 6 + 3 + e] = c[e]; }
                lo[e] = fminf(lo[e], a[e]); hi[e] = fmaxf(hi[e], c[e]);
            }
            if (valid) ext_sum += fmaxf(fmaxf(c[0] - a[0], c[1] - a[1]), c[2] - a[2]);
            // part boxes: consecutive triangles mostly belong to one part -- a complete wavefront of one part reduces its 64
            // boxes on DPP and one lane updates the part's box.  The wavefront's own box -- a cluster of 64 consecutive triangles --
            // was kept as well in round 5 (k_pen_frame culled whole clusters against the part boxes; P.wbox is NULL since round 6).
            const int s0 = __builtin_amdgcn_readfirstlane(seg[u]);
            const bool one_part = __ballot(!valid || seg[u] == s0) == ~0ull;
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const float wl = -wave_max_dpp(-a[e]), wh = wave_max_dpp(c[e]);
                if ((t & 63) == 0) {
                    if (P.wbox) { float* wb = P.wbox + ((size_t)b * P.n_clus + (f >> 6)) * 6; wb[e] = wl; wb[3 + e] = wh; }
                    if (one_part) { atomicMin(&pbox[s0 * 6 + e], pen_ford(wl)); atomicMax(&pbox[s0 * 6 + 3 + e], pen_ford(wh)); }
                }
                if (!one_part && valid) { atomicMin(&pbox[seg[u] * 6 + e], pen_ford(a[e])); atomicMax(&pbox[seg[u] * 6 + 3 + e], pen_ford(c[e])); }
            }
        }
    }
    __syncthreads();
    if (t < 64 * 6) {
        const int v = pbox[t];
        int* g = P.pbox + (size_t)b * 64 * 6 + t;
        if ((t % 6) < 3) { if (v != 0x7fffffff) atomicMin(g, v); } else if (v != (int)0x80000000) atomicMax(g, v);
    }
    float r[7];
    for (int e = 0; e < 3; ++e) { r[e] = block_min(lo[e], red); r[3 + e] = block_max(hi[e], red); }
    r[6] = block_sum_fixed(ext_sum, red);
    if (t < 7) P.gpart[((size_t)b * PEN_GW + w) * 8 + t] = r[t];
}

// frame box, cell size, skip / near masks: what every workgroup of g2 / g3 / g5 needs (recomputed per workgroup, fixed order)
struct PenGridCtx { float glo[3], ih; };
__device__ __forceinline__ PenGridCtx pen_grid_ctx(const PenDev& P, const int b) {
    PenGridCtx c;
    float lo[3] = {3e38f, 3e38f, 3e38f}, ext = 0.f;
    for (int w = 0; w < PEN_GW; ++w) {
        const float* g = P.gpart + ((size_t)b * PEN_GW + w) * 8;
        for (int e = 0; e < 3; ++e) lo[e] = fminf(lo[e], g[e]);
        ext += g[6];
    }
    const float h = fmaxf(2.f * (ext / (float)P.F), 1e-6f);
    for (int e = 0; e < 3; ++e) c.glo[e] = lo[e];
    c.ih = 1.f / h;
    return c;
}
__device__ __forceinline__ int pen_cell_of(const PenGridCtx& c, float x, int e) { return min(1 << 20, max(0, (int)fminf((x - c.glo[e]) * c.ih, 1048576.f))); }

// fn(bucket, packed cell key) for every cell of a packed range
template <class FN>
__device__ __forceinline__ void pen_for_cells(const int2 pk, FN&& fn) {
    const int x0 = pk.x & 1023, y0 = (pk.x >> 10) & 1023, z0 = (pk.x >> 20) & 1023;
    const int sx = pk.y & 7, sy = (pk.y >> 3) & 7, sz = (pk.y >> 6) & 7;
    // (the key's two spare bits, and bit 0 of the third argument, say on which axes -- x, y, z -- this cell is the one that
    //  holds the LOW corner of the triangle's box: the pair tests decide ownership of a pair on these bits)
    for (int dz = 0; dz <= sz; ++dz) for (int dy = 0; dy <= sy; ++dy) for (int dx = 0; dx <= sx; ++dx) {
        const int x = (x0 + dx) & 1023, y = (y0 + dy) & 1023, z = (z0 + dz) & 1023;
        fn(pen_bucket(x, y, z), x | (y << 10) | (z << 20) | ((dx == 0) << 30) | ((dy == 0) << 31), dz == 0);
    }
}
__global__ __launch_bounds__(PEN_T)
void k_pen_g2(PenDev P, const int* __restrict__ want, int nb) {
    __shared__ unsigned long long s_mask[64], s_near[64];
    __shared__ int s_pbox[64][6];
    __shared__ int s_cnt, s_base, s_ccnt, s_cbase;
    const int t = threadIdx.x, lane = t & 63;
    int b, w;
    if (!pen_gw_map(nb, b, w)) return;
    // (round 5: the inputs of the culling -- part boxes, the static part table, the frame-box partials -- come in ONE round trip,
    //  fetched by different lanes, and the 64 x 64 "do these parts' boxes meet" tests are dealt over the lanes, 16 per part: the
    //  prologue was a string of dependent loads and a 55-trip loop on 64 lanes)
    __shared__ float s_gpart[PEN_GW * 8];
    __shared__ unsigned s_near32[128];
    const int wanted = want ? want[b] : 1;
    {
        int pb_v = 0; unsigned long long sk_v = 0ull; float gp_v = 0.f;
        if (t < 64 * 6) pb_v = P.pbox[(size_t)b * 64 * 6 + t];
        else if (t < 64 * 6 + 64) sk_v = P.skipmask[t - 64 * 6];
        else if (t < 64 * 6 + 64 + PEN_GW * 8) gp_v = P.gpart[(size_t)b * PEN_GW * 8 + (t - 64 * 6 - 64)];
        if (!wanted) return;
        if (t < 64 * 6) (&s_pbox[0][0])[t] = pb_v;
        else if (t < 64 * 6 + 64) s_mask[t - 64 * 6] = sk_v;
        else if (t < 64 * 6 + 64 + PEN_GW * 8) s_gpart[t - 64 * 6 - 64] = gp_v;
        if (t < 128) s_near32[t] = 0u;
    }
    __syncthreads();
    const int F = P.F;
    const float* aabb = P.aabb + (size_t)b * F * 6;
    int2* cand = P.cand + (size_t)b * P.ent_cap;
    PenGridCtx C;                               // (pen_grid_ctx on the staged partials: same operations, same order)
    {
        float lo3[3] = {3e38f, 3e38f, 3e38f}, ext = 0.f;
        for (int w_ = 0; w_ < PEN_GW; ++w_) {
            const float* g = s_gpart + w_ * 8;
            for (int e = 0; e < 3; ++e) lo3[e] = fminf(lo3[e], g[e]);
            ext += g[6];
        }
        const float h = fmaxf(2.f * (ext / (float)P.F), 1e-6f);
        for (int e = 0; e < 3; ++e) C.glo[e] = lo3[e];
        C.ih = 1.f / h;
    }
    if (w == 0 && t == 0) { float* gp = P.gridp + b * 4; gp[0] = C.glo[0]; gp[1] = C.glo[1]; gp[2] = C.glo[2]; gp[3] = C.ih; }
    {
        const int p_ = t >> 4, q0 = t & 15;
        unsigned lo_m = 0u, hi_m = 0u;
        if (p_ < P.n_parts && s_pbox[p_][0] <= s_pbox[p_][3]) {
            const unsigned long long sk = s_mask[p_];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int q = q0 + 16 * k;
                const bool meet = (q < P.n_parts) & (s_pbox[p_][0] <= s_pbox[q][3]) & (s_pbox[q][0] <= s_pbox[p_][3]) & (s_pbox[p_][1] <= s_pbox[q][4]) &
                                  (s_pbox[q][1] <= s_pbox[p_][4]) & (s_pbox[p_][2] <= s_pbox[q][5]) & (s_pbox[q][2] <= s_pbox[p_][5]);
                if (meet && !((sk >> q) & 1ull)) { if (q < 32) lo_m |= 1u << q; else hi_m |= 1u << (q - 32); }
            }
        }
        if (lo_m) atomicOr(&s_near32[2 * p_], lo_m);
        if (hi_m) atomicOr(&s_near32[2 * p_ + 1], hi_m);
    }
    __syncthreads();
    if (t < 64) s_near[t] = (unsigned long long)s_near32[2 * t] | ((unsigned long long)s_near32[2 * t + 1] << 32);
    __syncthreads();
    // part culling (a triangle whose box meets the box of no part it may collide with cannot have a partner and never
    // enters the grid), packed cell range of the survivors, part masks of the buckets (folded to 32 bits)
    // The survivors are COMPACTED into the frame's list (k_pen_g3 makes three passes over them and a wavefront's pass lasts as
    // long as its widest triangle: dead lanes between live ones cost as much as live ones): a wavefront reserves its share of
    // the workgroup's range with one LDS atomic per batch, the workgroup its range of the frame's list with one global atomic.
    // The order of the list is immaterial (the buckets are filled through atomics anyway; the pair set does not depend on it).
    for (int fb = 0; fb < F; fb += PEN_GW * PEN_T * PEN_GU) {        // (uniform trip count: barriers inside)
        const int f0 = fb + w * PEN_T + t;
        float bx[PEN_GU][6]; int seg[PEN_GU];
#pragma unroll
        for (int u = 0; u < PEN_GU; ++u) {
            const int f = f0 + u * PEN_GW * PEN_T, ff = f < F ? f : 0;
            seg[u] = P.segm[ff];
#pragma unroll
            for (int e = 0; e < 6; ++e) bx[u][e] = aabb[ff * 6 + e];
        }
        if (t == 0) { s_cnt = 0; s_ccnt = 0; }
        __syncthreads();
        int2 pk[PEN_GU]; int coff[PEN_GU];
#pragma unroll
        for (int u = 0; u < PEN_GU; ++u) {
            const int f = f0 + u * PEN_GW * PEN_T;
            bool any = false;
            unsigned long long nm = f < F ? s_near[seg[u]] : 0ull;
            if (nm) {
                int a6[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) a6[e] = pen_ford(bx[u][e]);
                while (nm && !any) {      // (two parts per trip: independent LDS reads; all comparisons combined with `&`: `&&` compiles to a branch per condition)
                    const int q0_ = __ffsll((long long)nm) - 1; nm &= nm - 1;
                    const int q1_ = nm ? __ffsll((long long)nm) - 1 : q0_; nm &= nm - 1;
                    const int* pa = s_pbox[q0_]; const int* pb = s_pbox[q1_];
                    any = ((a6[0] <= pa[3]) & (pa[0] <= a6[3]) & (a6[1] <= pa[4]) & (pa[1] <= a6[4]) & (a6[2] <= pa[5]) & (pa[2] <= a6[5])) |
                          ((a6[0] <= pb[3]) & (pb[0] <= a6[3]) & (a6[1] <= pb[4]) & (pb[1] <= a6[4]) & (a6[2] <= pb[5]) & (pb[2] <= a6[5]));
                }
            }
            pk[u] = make_int2(0, 0);
            if (any) {
                int c0[3], sp[3];
#pragma unroll
                for (int e = 0; e < 3; ++e) { c0[e] = pen_cell_of(C, bx[u][e], e); sp[e] = min(pen_cell_of(C, bx[u][3 + e], e), c0[e] + PEN_SPAN - 1) - c0[e]; }
                pk[u].x = (c0[0] & 1023) | ((c0[1] & 1023) << 10) | ((c0[2] & 1023) << 20) | (int)0x80000000;
                pk[u].y = sp[0] | (sp[1] << 3) | (sp[2] << 6) | (seg[u] << 9);
            }
            // Round 4: the survivor's CELLS are listed here, one entry record per cell its box touches, in one flat list of the
            // frame (a wavefront reserves its share with one DPP scan and one LDS atomic, the workgroup its range with one global
            // atomic; the order of the list is immaterial).  k_pen_g3 used to walk the cells of 4-7 triangles per lane in each of its
            // three passes -- a wavefront's pass lasted as long as its widest lane (a triangle of 18 cells next to lanes with 2) --
            // and now makes three balanced passes over this list.
            const unsigned long long m = __ballot(any);
            if (lane == 0 && m) atomicAdd(&s_cnt, __popcll(m));
            const int nc = any ? ((pk[u].y & 7) + 1) * (((pk[u].y >> 3) & 7) + 1) * (((pk[u].y >> 6) & 7) + 1) : 0;
            const int inc = wave_incl_scan_dpp(nc);
            const int wtot = __builtin_amdgcn_readlane(inc, 63);
            int wo = 0;
            if (lane == 0 && wtot) wo = atomicAdd(&s_ccnt, wtot);
            coff[u] = __builtin_amdgcn_readfirstlane(wo) + inc - nc;
        }
        __syncthreads();
        if (t == 0) { if (s_cnt) atomicAdd(&P.tcount[b * 16], s_cnt);            // survivors (statistics)
                      s_cbase = s_ccnt ? atomicAdd(&P.tcount[b * 16 + 1], s_ccnt) : 0; }
        __syncthreads();
        const int cbase = s_cbase;
#pragma unroll
        for (int u = 0; u < PEN_GU; ++u)
            if (pk[u].x < 0) {
                const int f = f0 + u * PEN_GW * PEN_T, pf = (pk[u].y >> 9) & 63;
                int pos = cbase + coff[u];
                pen_for_cells(pk[u], [&](int, int key, int lowz) {
                    if (pos < P.ent_cap) cand[pos] = make_int2(f | (pf << 24) | (lowz << 30), key);      // (triangle | part << 24 | low-corner bit z << 30, cell | low-corner bits x, y << 30)
                    ++pos;
                });
            }
        __syncthreads();        // (the counters are reset by the next batch)
    }
}

// parts a triangle of part p may collide with, folded to 32 bits (a triangle only enters a cell that also holds such a
// part: the crowded interior of a limb, and joints where only parent and child meet, never reach the pair tests)
// (round 5: both halves of the 64-bit word -- [t] parts 0..31, [64 + t] parts 32..63.  Folded to one 32-bit word per bucket, as
//  until round 4, part p and part p + 32 were one bit: on the SMPL-X part table every finger of the right hand (40..54) looked
//  like a collar, the head or an arm (8..22) to the cell filter, and the grid held 3-4 x the entries an exact filter leaves.)
__device__ __forceinline__ void pen_coll32(const PenDev& P, unsigned* s_coll32 /* [128] */) {
    const int t = threadIdx.x;
    if (t < 64) {
        unsigned long long c = 0ull;
        if (t < P.n_parts) c = ~P.skipmask[t] & (P.n_parts >= 64 ? ~0ull : (1ull << P.n_parts) - 1ull);
        s_coll32[t] = (unsigned)c; s_coll32[64 + t] = (unsigned)(c >> 32);
    }
    __syncthreads();
}
// The cell filter of the grid build: a (triangle, cell) record becomes a grid entry only if its cell also holds a triangle of a
// part the record's part may collide with.  Which parts a bucket holds is one 32-bit word of LDS per bucket, so the 64 possible
// parts take two rounds over the records: parts 0..31 first -- the verdict is parked in bit 31 of the record --, then parts
// 32..63 in the same words.  Leaves pmask holding the second round's words: `pen_cell_keep` is the test the histogram and the
// scatter pass apply.  Contains barriers: the whole workgroup calls it.
template <int U2>
__device__ __forceinline__ void pen_cell_filter(const PenDev& P, int2* __restrict__ cand, const int NC, unsigned* pmask, const unsigned* s_coll32) {
    const int t = threadIdx.x;
    auto cell_bucket = [](const int key) { return pen_bucket(key & 1023, (key >> 10) & 1023, (key >> 20) & 1023); };
    for (int i0 = t; i0 < NC; i0 += PEN_T * U2) {
        int2 r[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int i = i0 + u * PEN_T; r[u] = cand[i < NC ? i : 0]; }
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int pf = (r[u].x >> 24) & 63; if (i0 + u * PEN_T < NC && pf < 32) atomicOr(&pmask[cell_bucket(r[u].y)], 1u << pf); }
    }
    __syncthreads();
    if (P.n_parts <= 32) return;
    for (int i0 = t; i0 < NC; i0 += PEN_T * U2) {
        int2 r[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int i = i0 + u * PEN_T; r[u] = cand[i < NC ? i : 0]; }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int i = i0 + u * PEN_T;
            if (i < NC && (pmask[cell_bucket(r[u].y)] & s_coll32[(r[u].x >> 24) & 63])) cand[i].x = r[u].x | (int)0x80000000;
        }
    }
    __threadfence_block();
    __syncthreads();
    for (int c = t; c < PEN_CELLS; c += PEN_T) pmask[c] = 0u;
    __syncthreads();
    for (int i0 = t; i0 < NC; i0 += PEN_T * U2) {
        int2 r[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int i = i0 + u * PEN_T; r[u] = cand[i < NC ? i : 0]; }
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int pf = (r[u].x >> 24) & 63; if (i0 + u * PEN_T < NC && pf >= 32) atomicOr(&pmask[cell_bucket(r[u].y)], 1u << (pf - 32)); }
    }
    __syncthreads();
}
__device__ __forceinline__ bool pen_cell_keep(const PenDev& P, const int2 r, const unsigned* pmask, const unsigned* s_coll32, const int bk) {
    const int pf = (r.x >> 24) & 63;
    return P.n_parts <= 32 ? (pmask[bk] & s_coll32[pf]) != 0u : ((r.x < 0) | ((pmask[bk] & s_coll32[64 + pf]) != 0u));
}

// One workgroup per frame: bucket part masks, histogram, scan and scatter of the (cell, triangle) entries, all on LDS atomics
// (the same passes on global atomics -- eight workgroups per frame -- measured slower: 60-90 us each).  Every pass reads one
// coalesced 8-byte word per triangle (k_pen_g2's packed cell range), 7 of them in flight per lane.
__global__ __launch_bounds__(PEN_T)
void k_pen_g3(PenDev P, const int* __restrict__ want) {
    extern __shared__ int cell_cnt[];           // [PEN_CELLS + 1] histogram, then start offsets, then cursors | [PEN_CELLS] part masks
    __shared__ int slice[PEN_T];
    __shared__ int s_total;
    __shared__ unsigned s_coll32[128];
    const int b = blockIdx.x, t = threadIdx.x;
    int* st = P.stats + b * PEN_STATS;
    int* cells = P.cells + (size_t)b * (PEN_CELLS + 1);
    if (t == 0) { P.wqn[b] = 0; P.pcnt[b] = 0; }      // (the pair tests' chunk queue and pair list of this mesh start empty)
    if (want && !want[b]) {                     // the frame's stage carries no collision weight: nothing to do
        if (t == 0) { P.ptotal[b] = 0; cells[PEN_CELLS] = 0; st[0] = st[1] = st[2] = st[3] = 0; st[13] = 0; st[15] = 0; }
        return;
    }
    const int F = P.F;
    int2* cand = P.cand + (size_t)b * P.ent_cap;
    const int NT = min(P.tcount[b * 16], F);              // triangles that survived the part culling (statistics)
    const int NC_raw = P.tcount[b * 16 + 1];              // (triangle, cell) records k_pen_g2 listed
    const int NC = min(NC_raw, P.ent_cap);
    unsigned* pmask = reinterpret_cast<unsigned*>(cell_cnt + PEN_GRID_INTS);
#ifdef PEN_COUNT    // diagnostic build: shader clocks at the phase boundaries -> stats[24..30] (cycles per phase, thread 0)
    long long g3c[8]; int g3n = 0;
#define G3MARK() do { g3c[g3n++] = clock64(); } while (0)
#else
#define G3MARK() do { } while (0)
#endif
    G3MARK();
    for (int c = t; c <= PEN_CELLS; c += PEN_T) cell_cnt[c] = 0;
    for (int c = t; c < PEN_CELLS; c += PEN_T) pmask[c] = 0u;
    pen_coll32(P, s_coll32);                    // (ends with a barrier)
    // this kernel is the last reader of the frame's counts, and k_pen_g2 was the last reader of its part boxes: leave
    // them empty for the NEXT evaluation of this column (a launch of its own until round 4)
    if (t < 64 * 6) P.pbox[(size_t)b * 64 * 6 + t] = (t % 6) < 3 ? 0x7fffffff : (int)0x80000000;
    if (t == 0) { P.tcount[b * 16] = 0; P.tcount[b * 16 + 1] = 0; }
    G3MARK();
    // Three passes over the flat candidate list (coalesced 8-byte records, 8 in flight per lane): every lane does the same
    // amount of work whatever the shapes of the triangles.
    constexpr int U2 = 8;
    auto cell_bucket = [](const int key) { return pen_bucket(key & 1023, (key >> 10) & 1023, (key >> 20) & 1023); };
    // which parts are present in each bucket: the cell filter (two rounds of 32 parts each; pen_cell_filter)
    pen_cell_filter<U2>(P, cand, NC, pmask, s_coll32);
    G3MARK();
    // histogram (a triangle only enters a cell that also holds a part it may collide with)
    for (int i0 = t; i0 < NC; i0 += PEN_T * U2) {
        int2 r[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int i = i0 + u * PEN_T; r[u] = cand[i < NC ? i : 0]; }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int bk = cell_bucket(r[u].y);
            if (i0 + u * PEN_T < NC && pen_cell_keep(P, r[u], pmask, s_coll32, bk)) atomicAdd(&cell_cnt[bk], 1);
        }
    }
    __syncthreads();
    G3MARK();
    {   // exclusive scan over the buckets: each lane owns a contiguous slice
        // (wavefront w owns buckets [w * 1024, (w + 1) * 1024) in 16 rows of 64: lane l reads bucket row * 64 + l -- conflict-free;
        //  a lane that owned 16 CONSECUTIVE buckets read them at a stride of 16 words, a 16-way bank conflict on every access:
        //  26 k of this kernel's 180 k cycles)
        constexpr int per = PEN_CELLS / PEN_T;
        static_assert(PEN_CELLS == PEN_T * per && PEN_T / 64 * 64 * per == PEN_CELLS, "scan layout");
        const int lane = t & 63, wv = t >> 6;
        int* row0 = cell_cnt + wv * (64 * per) + lane;
        int ex[per], carry = 0;
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int v = row0[i * 64];
            const int inc = wave_incl_scan_dpp(v);          // (six DPP adds; the __shfl_up ladder was 6 LDS-crossbar round trips, x 16 rows: 13.6 k of this kernel's 58 k cycles)
            ex[i] = carry + inc - v;
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
        if (lane == 0) slice[wv] = carry;
        __syncthreads();
        int base = 0, tot = 0;
        for (int i = 0; i < PEN_T / 64; ++i) { const int x = slice[i]; if (i < wv) base += x; tot += x; }
#pragma unroll
        for (int i = 0; i < per; ++i) row0[i * 64] = base + ex[i];
        if (t == 0) { cell_cnt[PEN_CELLS] = tot; s_total = tot; }
        __syncthreads();
    }
    G3MARK();
    int2* ent = P.entries + (size_t)b * P.ent_cap;
    const bool ent_ok = s_total <= P.ent_cap - 4 && NC_raw <= P.ent_cap;
    if (t == 0) { st[2] = ent_ok ? 0 : max(s_total, NC_raw); st[3] = PEN_CELLS; st[13] = 0; st[14] = s_total; st[15] = 0; for (int q = 4; q < 13; ++q) st[q] = 0;
                  for (int q = 16; q < PEN_STATS; ++q) st[q] = 0;
                  if (P.work) { atomicAdd(&P.work[0], (unsigned long long)s_total); atomicAdd(&P.work[2], 1ull); atomicAdd(&P.work[3], (unsigned long long)NT); } }
    if (!ent_ok) {       // grid too crowded for the entry buffer: report, produce no pairs
        if (t == 0) { st[0] = 0; st[1] = 0; P.ptotal[b] = 0; cells[PEN_CELLS] = 0; }
        return;
    }
    // scatter: the start offsets double as cursors, so bucket c ends up holding its END offset
    // (= the start of bucket c + 1); a bucket's entries are [c ? cell_cnt[c - 1] : 0, cell_cnt[c])
    for (int i0 = t; i0 < NC; i0 += PEN_T * U2) {
        int2 r[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int i = i0 + u * PEN_T; r[u] = cand[i < NC ? i : 0]; }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int bk = cell_bucket(r[u].y);
            if (i0 + u * PEN_T < NC && pen_cell_keep(P, r[u], pmask, s_coll32, bk)) ent[atomicAdd(&cell_cnt[bk], 1)] = make_int2(r[u].x & 0x7fffffff, r[u].y);      // one 8-byte store
        }
    }
    __threadfence_block();
    __syncthreads();
    G3MARK();
    for (int c = t; c <= PEN_CELLS; c += PEN_T) cells[c] = cell_cnt[c];
    G3MARK();
#ifdef PEN_COUNT
    if (t == 0) for (int q = 1; q < g3n; ++q) st[23 + q] = (int)(g3c[q] - g3c[q - 1]);      // [24] init, [25] part masks, [26] histogram, [27] scan, [28] scatter, [29] copy
#endif
#undef G3MARK
}


