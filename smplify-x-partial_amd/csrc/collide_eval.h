// collide_eval.h -- the distance-field penalty on the pair list and its gradient (k_pen_eval / facesum / gather; k_pen_narrow: one workgroup per mesh)
// Part of csrc/collide.hip (included there, in this order: collide_field.h, collide_grid.h, collide_pairs.h, collide_eval.h);
// one translation unit, compiled with -ffp-contract=off.
#pragma once

// one lane per ORDERED pair (f receives g, and f's vertices intrude into g): the lane differentiates
// with respect to f's 9 coordinates only, so every number has one owner.
// Loss of the frame = sum over the kept ordered pairs (f, g) of sum_{v in g} Psi_f(v)^2; the kept set is symmetric
// (see below), so the gradient is exact also when max_collisions cuts a list
//
// Work distribution (round 4): ONE flat list of 64-pair chunks over all meshes of the call.  With a grid per mesh (128
// workgroups each) a launch lasted as long as its most crowded mesh -- a frame whose limbs a trial step has pushed through each
// other carries ten times the pairs of the others (p50 22 us, p90 178 us, mean 60) -- while the lanes of every other mesh idled.
// Every workgroup forms the exclusive prefix of the meshes' chunk counts (ptotal, a few hundred integers) in LDS; a wavefront
// takes chunks c = w, w + W, ...; the mesh of a chunk is found by bisection.  A chunk is 64 consecutive pairs of ONE mesh's
// list, aligned to 64 in that list -- what k_pen_facesum's run sums rely on -- so the numbers are what they were.
// P2P (DistanceFieldPenetrationLoss(point2plane=True), oracle/penetration.py assumption A6): the repulsion -Psi n of a vertex
// is measured along the other triangle's normal -- every Psi^2 of the pair is weighted by c = (n_f . n_g)^2, and the gradient
// gains the path through both unit normals.  A lane (f, g) owns d / d (vertices of f): its own cone's terms (1) and the terms
// of g's cone at its vertices (2) both depend on n_f through c.
// One ordered pair (f receives g): the loss this lane owns and its gradient with respect to f's nine coordinates -> v[0..8], v[9].
// Shared by k_pen_eval and pen_narrow; this file is compiled with -ffp-contract=off, so the two instances perform the same
// fp32 operations in the same order whatever surrounds them (a fused multiply-add chosen in one context and not in the other
// would make the two forms of the term differ in the last bit).
template <bool P2P>
__device__ __forceinline__ void pen_pair_eval(const PenDev& P, const float* __restrict__ vb, const int f, const int g, const bool sym,
                                              const float sigma, const int penalize_outside, float (&v)[10]) {
    float p[9], qv[9];
    for (int k = 0; k < 3; ++k) for (int e = 0; e < 3; ++e) {
        p[k * 3 + e] = vb[(size_t)P.faces[f * 3 + k] * 3 + e];
        qv[k * 3 + e] = vb[(size_t)P.faces[g * 3 + k] * 3 + e];
    }
    float loss = 0.f, g9[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (sym) {
        const V3 P0 = {p[0], p[1], p[2]}, P1 = {p[3], p[4], p[5]}, P2 = {p[6], p[7], p[8]};
        const V3 Q[3] = {{qv[0], qv[1], qv[2]}, {qv[3], qv[4], qv[5]}, {qv[6], qv[7], qv[8]}};
        if constexpr (!P2P) {
        {   // (1) this triangle receives the partner's vertices: the loss, and its gradient through the own cone's geometry
            const ConeGeo gg_ = cone_geometry(P0, P1, P2);
            V3 go = {0.f, 0.f, 0.f}, gn = {0.f, 0.f, 0.f}; float gr = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                V3 gd, gnk; float grk;
                loss += cone_penalty(gg_.o, gg_.r, gg_.n, Q[k], sigma, penalize_outside, gd, gnk, grk);
                go = go - gd; gn = gn + gnk; gr += grk;            // d = v - o
            }
            V3 g0, g1, g2;
            cone_geometry_adj(gg_, go, gr, gn, g0, g1, g2);
            g9[0] += g0.x; g9[1] += g0.y; g9[2] += g0.z; g9[3] += g1.x; g9[4] += g1.y; g9[5] += g1.z; g9[6] += g2.x; g9[7] += g2.y; g9[8] += g2.z;
        }
        {   // (2) this triangle's vertices intrude into the partner's cone (partner geometry constant): d / d v = d / d d
            const ConeGeo gg_ = cone_geometry(Q[0], Q[1], Q[2]);
            const V3 Pk[3] = {P0, P1, P2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                V3 gd, gnk; float grk;
                (void)cone_penalty(gg_.o, gg_.r, gg_.n, Pk[k], sigma, penalize_outside, gd, gnk, grk);
                g9[k * 3] += gd.x; g9[k * 3 + 1] += gd.y; g9[k * 3 + 2] += gd.z;
            }
        }
        } else {
            const ConeGeo gf = cone_geometry(P0, P1, P2), gg = cone_geometry(Q[0], Q[1], Q[2]);
            const float dt = vdot(gf.n, gg.n), c = dt * dt;
            // (1) own cone at the partner's vertices: value S1, adjoint with respect to the own (o, r, n)
            V3 go = {0.f, 0.f, 0.f}, gn = {0.f, 0.f, 0.f}; float gr = 0.f, S1 = 0.f, S2 = 0.f;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                V3 gd, gnk; float grk;
                S1 += cone_penalty(gf.o, gf.r, gf.n, Q[k], sigma, penalize_outside, gd, gnk, grk);
                go = go - gd; gn = gn + gnk; gr += grk;
            }
            // (2) the partner's cone at the own vertices: value S2 (owned as a LOSS by the lane (g, f)), d / d v = d / d d
            const V3 Pk[3] = {P0, P1, P2};
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                V3 gd, gnk; float grk;
                S2 += cone_penalty(gg.o, gg.r, gg.n, Pk[k], sigma, penalize_outside, gd, gnk, grk);
                g9[k * 3] += c * gd.x; g9[k * 3 + 1] += c * gd.y; g9[k * 3 + 2] += c * gd.z;
            }
            loss += c * S1;
            // c = (n_f . n_g)^2 multiplies both sums: d c / d n_f = 2 (n_f . n_g) n_g
            gn = gn * c + gg.n * ((S1 + S2) * 2.f * dt);
            V3 g0, g1, g2;
            cone_geometry_adj(gf, go * c, gr * c, gn, g0, g1, g2);
            g9[0] += g0.x; g9[1] += g0.y; g9[2] += g0.z; g9[3] += g1.x; g9[4] += g1.y; g9[5] += g1.z; g9[6] += g2.x; g9[7] += g2.y; g9[8] += g2.z;
        }
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) v[j] = g9[j];
    v[9] = loss;
}
// sum over the pairs of one triangle that sit in this wavefront (adjacent lanes): segmented inclusive scan, then the last lane of
// every run stores the run's sum at its own list position i (po: [10][pair_cap])
__device__ __forceinline__ void pen_run_sums(float (&v)[10], const int fkey, const bool valid, const int lane, float* __restrict__ po,
                                             const int pair_cap, const int i) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int fu = __shfl_up(fkey, d);
        const bool take = lane >= d && fu == fkey;
#pragma unroll
        for (int j = 0; j < 10; ++j) { const float vu = __shfl_up(v[j], d); if (take) v[j] += vu; }
    }
    const int fnext = __shfl_down(fkey, 1);
    if (valid && (lane == 63 || fnext != fkey)) {
#pragma unroll
        for (int j = 0; j < 10; ++j) po[(size_t)j * pair_cap + i] = v[j];
    }
}

template <bool P2P>
__global__ __launch_bounds__(256)
void k_pen_eval(PenDev P, const float* __restrict__ verts, float sigma, int penalize_outside, int B, int flat, PenSel sel) {
    extern __shared__ int s_pref[];             // [B + 1] (flat distribution)
    __shared__ int s_scan[256];
    const int lane = threadIdx.x & 63;
    if (sel.hlist && *sel.nheavy == 0) return;
    int n_chunks = 0;
    if (flat) n_chunks = pen_prefix(B, s_pref, s_scan, [&](int b_) { return pen_sel_on(sel, b_) ? (P.ptotal[b_] + 63) >> 6 : 0; });
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = gridDim.x * 4;
    // flat: chunk ids wave, wave + n_waves, ...; per mesh (flat = 0): blockIdx.y is the mesh, chunks of its own list
    for (int c = flat ? wave : (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6); ; c += n_waves) {
        int b, i0;
        if (flat) { if (c >= n_chunks) break; b = pen_chunk_mesh(s_pref, B, c); i0 = (c - s_pref[b]) * 64; }
        else { b = blockIdx.y; i0 = c * 64; if (i0 >= P.ptotal[b]) break; }
        const int total = P.ptotal[b];
        const float* vb = verts + (size_t)b * P.V * 3;
        const int* pown = P.pown + (size_t)b * P.pair_cap;
        const int* plist = P.plist + (size_t)b * P.pair_cap;
        float* po = P.pout + (size_t)b * 10 * P.pair_cap;
        const int i = i0 + lane;
        const bool valid = i < total;
        const int is_ = valid ? i : 0;
        const int f_ld = pown[is_], g_ld = plist[is_];
        const int f = valid ? f_ld : 0, g = valid ? g_ld : 0;
        // BVH(max_collisions): a triangle with more than max_collisions partners keeps its lowest ids (k_pen_list), and a
        // pair counts only if BOTH triangles kept each other -- the kept set is symmetric, so the two lanes (f, g) and
        // (g, f) exist together and every gradient term has its owner.  Lists that were not cut hold every partner; a cut
        // list is searched for f.
        bool sym = valid;
        if (valid) {
            const int cg = P.pcount[(size_t)b * P.F + g];
            if (P.pavail[(size_t)b * P.F + g] > cg) {
                const int og = P.poff[(size_t)b * P.F + g];
                int lo = 0, hi = max(0, min(cg, P.pair_cap - og));
                const int top = hi;
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (plist[og + mid] < f) lo = mid + 1; else hi = mid; }
                sym = lo < top && plist[og + lo] == f;
            }
        }
        {   // pairs kept by one side only: counted as dropped (stats), contribute nothing
            const unsigned long long dead = __ballot(valid && !sym);
            if (dead && lane == 0) atomicAdd(&P.stats[b * PEN_STATS + 15], __popcll(dead));
        }
        float v[10];
        pen_pair_eval<P2P>(P, vb, f, g, sym, sigma, penalize_outside, v);
        pen_run_sums(v, valid ? f : -1, valid, lane, po, P.pair_cap, i);
    }
}

// per triangle: sum over its pair range of the 9 gradient components and the loss.  k_pen_eval has summed
// the pairs of a triangle inside each 64-pair chunk of the list; the lane that sits on the first pair of
// a range adds the (1 + range / 64) chunk sums in ascending order.
// (Round 4 tried these sums inside k_pen_gather, per incident corner: one launch fewer, but every corner then walks two
//  dependent loads and its chunk loop on the lane's own chain -- 75 us against 36 + 11 for the two kernels.  Kept apart.)
// the sums of one triangle's pair range [i, i + n) from the run sums k_pen_eval / pen_narrow left per 64-pair chunk of the list
__device__ __forceinline__ void pen_face_sum(const float* __restrict__ po, const int pair_cap, const int i, const int n, float* __restrict__ tg /* [9] */,
                                             float* __restrict__ tl) {
    float acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int end = i + n - 1;
    for (int c = i >> 6; c <= end >> 6; ++c) {           // one run sum per 64-pair chunk of the range
        const int q = min(end, c * 64 + 63);
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[j] += po[(size_t)j * pair_cap + q];
    }
#pragma unroll
    for (int j = 0; j < 9; ++j) tg[j] = acc[j];
    *tl = acc[9];
}
__global__ __launch_bounds__(256)
void k_pen_facesum(PenDev P, PenSel sel) {
    const int b_first = pen_sel_first(sel, blockIdx.y);
    const int nsel = pen_sel_n(sel, sel.hlist ? 0x7fffffff : (int)gridDim.y);
    for (int si = blockIdx.y; si < nsel; si += gridDim.y) {
    const int b = si == (int)blockIdx.y ? b_first : pen_sel_col(sel, si);
    if (!pen_sel_on(sel, b)) continue;
    const int total = P.ptotal[b];
    const int* pown = P.pown + (size_t)b * P.pair_cap;
    const int* pc = P.pcount + (size_t)b * P.F;
    const float* po = P.pout + (size_t)b * 10 * P.pair_cap;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int f = pown[i];
        if (i > 0 && pown[i - 1] == f) continue;
        pen_face_sum(po, P.pair_cap, i, min(pc[f], total - i), P.tgrad + ((size_t)b * P.F + f) * 9, P.tloss + (size_t)b * P.F + f);
    }
    }
}

// vertex gradient = fixed-order sum over the incident triangle corners (CSR); frame loss = sum over the
// triangles in index order (independent of where a pair sits in the list).  When the caller is a fitting batch the lane
// that has formed g(v) goes on to d v_posed = T^T g, the operand of the adjoint GEMM (a launch of its own, k_adj_prep,
// until round 4).
// g(v) of one vertex and what follows from it (shared by k_pen_gather, every vertex, and pen_narrow, the vertices of triangles
// that have pairs -- the others' rows are zeroed by k_pen_g1).  s_hasp: the frame's "triangle has pairs" bits in LDS.
__device__ __forceinline__ void pen_vertex_out(const PenDev& P, const int b, const int v, const int total, const unsigned* s_hasp,
                                               float* __restrict__ dverts, const PenAdjPrep& ap) {
    auto has = [&](int face) { return (s_hasp[face >> 5] >> (face & 31)) & 1u; };
    float g[3] = {0.f, 0.f, 0.f};
    // (round 5: the vertex' skinning row is fetched with the first loads of the chain, not behind the gradient it multiplies --
    //  one dependent round trip less on every vertex that carries a gradient)
    int wj_[SFX_NW]; float ww_[SFX_NW];
    if (ap.adj_G) {
#pragma unroll
        for (int q = 0; q < SFX_NW; ++q) { wj_[q] = ap.Wsp_j[(size_t)v * SFX_NW + q]; ww_[q] = ap.Wsp_w[(size_t)v * SFX_NW + q]; }
    }
    if (total > 0) {
        const float* tg = P.tgrad + (size_t)b * P.F * 9;
        // (the incident corners in batches of 8 -- a vertex of a closed mesh has ~6 -- so that the three dependent
        //  loads per corner overlap across the corners instead of forming one chain per corner; same summation order)
        const int q0 = P.vf_start[v], q1 = P.vf_start[v + 1];
        for (int qb = q0; qb < q1; qb += 8) {
            int fc[8]; bool use[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int l_ = P.vf_list[min(qb + u, q1 - 1)]; fc[u] = qb + u < q1 ? l_ : -1; }      // (unconditional loads)
#pragma unroll
            for (int u = 0; u < 8; ++u) use[u] = fc[u] >= 0 && has(fc[u] / 3);      // (2.6 KB of bits in LDS instead of two gathers per corner)
            float tv[8][3];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int e = 0; e < 3; ++e) tv[u][e] = use[u] ? tg[(size_t)fc[u] * 3 + e] : 0.f;      // fc = face * 3 + corner -> [face][corner][3]
#pragma unroll
            for (int u = 0; u < 8; ++u) if (use[u]) { g[0] += tv[u][0]; g[1] += tv[u][1]; g[2] += tv[u][2]; }
        }
    }
    for (int e = 0; e < 3; ++e) dverts[((size_t)b * P.V + v) * 3 + e] = g[e];
    if (ap.adj_G) {         // d v_posed(v) = T(v)[:3,:3]^T g(v),  T(v) = sum_j W[v][j] A_j  (zeros where g = 0)
        float o0 = 0.f, o1 = 0.f, o2 = 0.f;
        if (g[0] != 0.f || g[1] != 0.f || g[2] != 0.f) {
            const size_t Bp = (size_t)ap.Bpad;
            float T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            auto add = [&](const int j, const float w) {
#pragma unroll
                for (int rr = 0; rr < 3; ++rr)
#pragma unroll
                    for (int c = 0; c < 3; ++c) T[rr * 3 + c] += w * ap.AT[((size_t)(rr * 4 + c) * SFX_JPAD + j) * Bp + b];
            };
            if (wj_[0] >= 0) {
#pragma unroll
                for (int q = 0; q < SFX_NW; ++q) if (ww_[q] != 0.f) add(wj_[q], ww_[q]);
            } else {
                for (int j = 0; j < SFX_J; ++j) { const float w = ap.W[(size_t)v * SFX_J + j]; if (w != 0.f) add(j, w); }
            }
            o0 = T[0] * g[0] + T[3] * g[1] + T[6] * g[2];
            o1 = T[1] * g[0] + T[4] * g[1] + T[7] * g[2];
            o2 = T[2] * g[0] + T[5] * g[1] + T[8] * g[2];
        }
        float* o = ap.adj_G + (size_t)b * 3 * ap.Vpad + (size_t)v * 3;
        o[0] = o0; o[1] = o1; o[2] = o2;
    }
}
// the frame's loss: triangles with pairs in index order, dealt to 256 lanes, lanes and wavefronts combined in a fixed order
// (called by the first 256 threads of a workgroup; red: 4 floats of LDS; contains a barrier: every thread of the FIRST FOUR
// wavefronts must arrive -- the callers make the call wave-uniform)
__device__ __forceinline__ float pen_frame_loss_partial(const PenDev& P, const int b, const int total, const unsigned* s_hasp, const int t256) {
    float s = 0.f;
    // (round 5: eight unconditional loads per trip, the bit decides what is added -- a load under `if (bit)` in a loop of 82 trips was a
    //  dependent round trip for every triangle with pairs a lane met; same order of the sum)
    if (total > 0) for (int f0 = t256; f0 < P.F; f0 += 256 * 8) {
        float v[8]; bool on[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {               // (a lane without a pair reads the column's first word: one line for all of them)
            const int f = f0 + u * 256;
            on[u] = f < P.F && ((s_hasp[min(f, P.F - 1) >> 5] >> (f & 31)) & 1u);
            v[u] = P.tloss[(size_t)b * P.F + (on[u] ? f : 0)];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (on[u]) s += v[u];
    }
    return wave_sum_dpp(s);
}

__global__ __launch_bounds__(256)
void k_pen_gather(PenDev P, float* __restrict__ dverts, float* __restrict__ loss_out, PenSel sel, PenAdjPrep ap) {
    __shared__ float red[4];
    extern __shared__ unsigned s_hasp[];        // [hasp_words] triangles of this frame that have pairs
    const int b_first = pen_sel_first(sel, blockIdx.y);
    const int nsel = pen_sel_n(sel, sel.hlist ? 0x7fffffff : (int)gridDim.y);
    for (int si = blockIdx.y; si < nsel; si += gridDim.y) {
    const int b = si == (int)blockIdx.y ? b_first : pen_sel_col(sel, si);
    if (!pen_sel_on(sel, b)) { if (blockIdx.x == 0 && threadIdx.x == 0) loss_out[b] = 0.f; continue; }
    const int v = blockIdx.x * 256 + threadIdx.x;
    const int total = P.ptotal[b];
    __syncthreads();                            // (the previous column's readers of the bits are done)
    for (int w = threadIdx.x; w < P.hasp_words; w += 256) s_hasp[w] = P.hasp[(size_t)b * P.hasp_words + w];
    __syncthreads();
    if (v < P.V) pen_vertex_out(P, b, v, total, s_hasp, dverts, ap);
    if (blockIdx.x == gridDim.x - 1) {          // (the row's last workgroup: it has the fewest vertices)
        const float s = pen_frame_loss_partial(P, b, total, s_hasp, threadIdx.x);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
        __syncthreads();
        if (threadIdx.x == 0) loss_out[b] = ((red[0] + red[1]) + red[2]) + red[3];
    }
    }
}

// =============================================================================================
// One workgroup per mesh from the accepted pairs to the outputs (pen_narrow / k_pen_narrow) -- the stand-alone pair evaluation
// (sfx_pen_eval_pairs: DistanceFieldPenetrationLoss on pairs the caller supplies) and, in the lab build, forms 1 / 2 of the fitting
// loop's step.  On the mesh the reference evaluates, the part boxes turn away 88 % of the triangles before the grid, ~2 500 survivors
// make ~2 000 grid entries and ~1 700 ordered pairs per evaluation, which one workgroup holds in LDS:
//   D  both orders of every pair as 32-bit keys f * F + g, bitonic sort in LDS, rank within a triangle's run: the max_collisions
//      LOWEST partners are kept -> the mesh's pair list, triangles ascending, partners ascending   (k_pen_list + k_pen_rank)
//   E  pair evaluation per 64-aligned chunk of that list (pen_pair_eval, pen_run_sums: the general kernels' functions)
//   F  per-triangle sums (pen_face_sum)
//   G  gradient of the vertices of triangles that have pairs (pen_vertex_out), d v_posed = T^T g, the mesh's loss
// Every number is formed by the same fp32 operations in the same order as in the general kernels: the pair list is canonical, the
// sums are defined on it (tests/test_gpu_topology.py, tests/test_gpu_penetration.py compare bit for bit on the lab build).  A mesh
// with more than PEN_FP pairs does not fit (P.heavy / P.hlist: the lab forms hand it to the general kernels, the stand-alone
// evaluation reports it).  (Round 5's k_pen_frame -- the grid build and the pair tests in the same workgroup, phases A-C -- lost to
// the flat kernels on whole fits, 145 against 332 frames/s, and was deleted in round 6: LAB_NOTES.md.)
#define PEN_FP 8192             // unordered pairs on the fast path: 2 x PEN_FP sort keys = 64 KB of LDS
#define PEN_FW 16               // wavefronts of the workgroup

__device__ __forceinline__ int pen_block_excl_scan_max(const int v, int* wmax /* [PEN_T / 64] */) {      // exclusive prefix MAX over the block's lanes (values >= -1)
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (lane >= d) inc = max(inc, o); }
    __syncthreads();
    if (lane == 63) wmax[wv] = inc;
    __syncthreads();
    int base = -1;
    for (int i = 0; i < wv; ++i) base = max(base, wmax[i]);
    const int prev = __shfl_up(inc, 1);
    return max(base, lane > 0 ? prev : -1);
}

// Phases D-G of the per-column work: from the column's accepted pairs (P.pbuf, any order) to the pair list, the pair evaluation,
// the per-triangle sums, the gradient of the vertices that have one, d v_posed and the frame's loss -- one workgroup of PEN_T lanes,
// everything between the pair buffer and the outputs in LDS.  Used by k_pen_narrow.
struct PenNarrowLds { unsigned* keys /* [2 PEN_FP] */; unsigned* bits /* [2 hasp_words + (V + 31) / 32 + V] */; int* slice /* [PEN_T] */; float* red /* [PEN_T / 64] */; int* dead; };
template <bool P2P, class MARK>
__device__ __forceinline__ void pen_narrow(const PenDev& P, const int b, const int npairs, const PenNarrowLds L, const float* __restrict__ verts,
                                           const float sigma, const int penalize_outside, float* __restrict__ dverts, float* __restrict__ loss_out,
                                           const PenAdjPrep& ap, MARK&& mark) {
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6, F = P.F;
    int* st = P.stats + b * PEN_STATS;
    int* slice = L.slice; float* red = L.red;
    // ---------------------------------------------------------------- D: the pair list (k_pen_list + k_pen_rank)
    unsigned* keys = L.keys;                                        // [np] both orders of every pair, then the kept list in place
    unsigned* bits = L.bits;                                        // [hw] cut lists | [hw] has pairs | [vw] touched vertices | vertex list
    const int hw = P.hasp_words, vw = (P.V + 31) >> 5;
    unsigned* cutb = bits; unsigned* hasb = bits + hw; unsigned* vtxb = bits + 2 * hw; int* vlist = reinterpret_cast<int*>(bits + 2 * hw + vw);
    const int n2 = 2 * npairs;
    int np = 64;
    while (np < n2) np <<= 1;
    {
        const int2* pbuf = P.pbuf + (size_t)b * P.pf_cap;
        for (int i = t; i < np / 2; i += PEN_T) {
            unsigned k0 = 0xffffffffu, k1 = 0xffffffffu;
            if (i < npairs) { const int2 pr = pbuf[i]; k0 = (unsigned)pr.x * (unsigned)F + (unsigned)pr.y; k1 = (unsigned)pr.y * (unsigned)F + (unsigned)pr.x; }
            keys[2 * i] = k0; keys[2 * i + 1] = k1;
        }
        for (int w = t; w < 2 * hw + vw; w += PEN_T) bits[w] = 0u;
    }
    for (int k = 2; k <= np; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __syncthreads();
            for (int q = t; q < np / 2; q += PEN_T) {           // compare-exchange q of this step: i = q with a 0 inserted at bit j
                const int i = 2 * q - (q & (j - 1)), ixj = i + j;
                const unsigned va = keys[i], vb = keys[ixj];
                if ((va > vb) == ((i & k) == 0)) { keys[i] = vb; keys[ixj] = va; }
            }
        }
    __syncthreads();
    // rank within the triangle's run; the max_collisions lowest partners stay, positions by a prefix sum (every lane a contiguous range)
    int T_ = 0;
    {
        const int per = (np + PEN_T - 1) / PEN_T;          // <= 16
        const int j0 = min(n2, t * per), j1 = min(n2, j0 + per);
        unsigned kk[16]; int fj[16];
        int last_start = -1;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = j0 + u;
            if (u < per && j < j1) {
                kk[u] = keys[j]; fj[u] = (int)(kk[u] / (unsigned)F);
                const bool start = j == 0 || (int)(keys[j - 1] / (unsigned)F) != fj[u];
                if (start) last_start = j;
            }
        }
        const int before = pen_block_excl_scan_max(last_start, slice);        // start of the run that is open when this lane's range begins
        int cur = before, nkeep = 0, ncut = 0;
        bool kp[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int j = j0 + u;
            kp[u] = false;
            if (u < per && j < j1) {
                const bool start = j == 0 || (u > 0 ? fj[u - 1] != fj[u] : cur < 0 || (int)(keys[j - 1] / (unsigned)F) != fj[u]);
                if (start) cur = j;
                kp[u] = j - cur < P.cap;
                if (kp[u]) ++nkeep; else { ++ncut; atomicOr(&cutb[fj[u] >> 5], 1u << (fj[u] & 31)); }
            }
        }
        int ptot;
        int pos = block_excl_scan(nkeep, slice, &ptot);
        const float cut_all = block_sum_fixed((float)ncut, red);           // (ends with a barrier: every read of the sorted keys is done)
        T_ = ptot;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (u < per && kp[u]) keys[pos++] = kk[u];
        if (t == 0) { P.ptotal[b] = T_; st[0] = T_; st[1] = (int)cut_all; if (P.over) P.over[b] = st[13] > 0 ? 1 : 0;      // (a cut bucket walk: pairs missing, order dependent)
                      if (P.work) { atomicAdd(&P.work[1], (unsigned long long)T_); if (st[13] > 0) atomicAdd(&P.work[5], (unsigned long long)st[13]); } }
    }
    __syncthreads();
    mark();                                     // [7] D: pair list
    const int T = T_;
    {   // the list as the diagnostics read it (sfx_pen_pairs)
        int* pown = P.pown + (size_t)b * P.pair_cap; int* plist = P.plist + (size_t)b * P.pair_cap;
        for (int i = t; i < T; i += PEN_T) { const unsigned k = keys[i]; const int f = (int)(k / (unsigned)F); pown[i] = f; plist[i] = (int)(k - (unsigned)f * (unsigned)F); }
    }
    // ---------------------------------------------------------------- E: pair evaluation, a 64-aligned chunk of the list per wavefront
    const float* vb = verts + (size_t)b * P.V * 3;
    float* po = P.pout + (size_t)b * 10 * P.pair_cap;
    for (int c = wv; c * 64 < T; c += PEN_FW) {
        const int i = c * 64 + lane;
        const bool valid = i < T;
        const unsigned k = keys[valid ? i : 0];
        const int f_ = (int)(k / (unsigned)F), g_ = (int)(k - (unsigned)f_ * (unsigned)F);
        const int f = valid ? f_ : 0, g = valid ? g_ : 0;
        bool sym = valid;
        if (valid && ((cutb[g >> 5] >> (g & 31)) & 1u)) {          // the partner's list was cut: did it keep this triangle?
            const unsigned want_k = (unsigned)g * (unsigned)F + (unsigned)f;
            int lo = 0, hi = T;
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < want_k) lo = mid + 1; else hi = mid; }
            sym = lo < T && keys[lo] == want_k;
        }
        {
            const unsigned long long dead = __ballot(valid && !sym);
            if (dead && lane == 0) atomicAdd(L.dead, __popcll(dead));
        }
        float v[10];
        pen_pair_eval<P2P>(P, vb, f, g, sym, sigma, penalize_outside, v);
        pen_run_sums(v, valid ? f : -1, valid, lane, po, P.pair_cap, i);
    }
    __threadfence_block();
    __syncthreads();
    if (t == 0) st[15] = (*L.dead);
    mark();                                     // [8] E: pair evaluation
    // ---------------------------------------------------------------- F: per-triangle sums; which triangles / vertices carry a gradient
    for (int i = t; i < T; i += PEN_T) {
        const unsigned k = keys[i];
        const int f = (int)(k / (unsigned)F);
        if (i > 0 && (int)(keys[i - 1] / (unsigned)F) == f) continue;
        const unsigned nextf = (unsigned)(f + 1) * (unsigned)F;        // (F^2 < 2^32: no wrap for f + 1 <= F)
        int lo = i + 1, hi = T;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (keys[mid] < nextf) lo = mid + 1; else hi = mid; }
        pen_face_sum(po, P.pair_cap, i, lo - i, P.tgrad + ((size_t)b * F + f) * 9, P.tloss + (size_t)b * F + f);
        atomicOr(&hasb[f >> 5], 1u << (f & 31));
        const int4 vf = P.faces4[f];
        atomicOr(&vtxb[vf.x >> 5], 1u << (vf.x & 31)); atomicOr(&vtxb[vf.y >> 5], 1u << (vf.y & 31)); atomicOr(&vtxb[vf.z >> 5], 1u << (vf.z & 31));
    }
    __threadfence_block();
    __syncthreads();
    mark();                                     // [9] F: per-triangle sums
    // ---------------------------------------------------------------- G: vertex gradients, d v_posed, the frame's loss
    {
        int nv = 0;
        for (int w0 = 0; w0 < vw; w0 += PEN_T) {             // (vw <= PEN_T for meshes of up to 32 k vertices: one trip)
            const int w = w0 + t;
            const unsigned word = w < vw ? vtxb[w] : 0u;
            int tot;
            int pos = nv + block_excl_scan(__popc(word), slice, &tot);
            unsigned m = word;
            while (m) { const int bit = __ffs((int)m) - 1; m &= m - 1; vlist[pos++] = w * 32 + bit; }
            nv += tot;
        }
        __syncthreads();
        for (int q = t; q < nv; q += PEN_T) pen_vertex_out(P, b, vlist[q], T, hasb, dverts, ap);
        if (t < 256) {
            const float s = pen_frame_loss_partial(P, b, T, hasb, t);
            if (lane == 0) red[wv] = s;
        }
        __syncthreads();
        if (t == 0) loss_out[b] = ((red[0] + red[1]) + red[2]) + red[3];
    }
    mark();                                     // [10] G: vertices, loss
}

// Round 5's default form: the grid build and the pair tests stay spread over the chip (k_pen_g1 / g2 / g3, k_pen_walk / walk2 --
// the tests are matrix-free ALU work, ~50 instructions per candidate and 10^4-10^5 candidates per column: one compute unit
// needs 70-150 us for a column's, measured in round 5's k_pen_frame), the accepted pairs land in one list per column, and ONE workgroup per
// column does everything behind them (pen_narrow) -- what k_pen_list, k_pen_rank, k_pen_eval, k_pen_facesum and k_pen_gather did
// with a pass over all F triangles or V vertices and a launch each.  A column with more pairs than the LDS sort holds (2 x PEN_FP
// keys) is handed to those kernels (P.heavy / P.hlist), which redo its pair tests into the partner lists.
template <bool P2P>
__global__ __launch_bounds__(PEN_T)
void k_pen_narrow(PenDev P, const float* __restrict__ verts, const float sigma, const int penalize_outside, float* __restrict__ dverts,
                  float* __restrict__ loss_out, const int* __restrict__ want, PenAdjPrep ap, const int force_heavy) {
    extern __shared__ int lds[];                // [2 PEN_FP] sort keys | bit sets and vertex list
    __shared__ int slice[PEN_T];
    __shared__ float red[PEN_T / 64];
    __shared__ int s_dead;
    const int b = blockIdx.x, t = threadIdx.x;
    int* st = P.stats + b * PEN_STATS;
    const long long t_start = wall_clock64();
    int n_mark = 3;                             // (stats[7..10]: the stamps of phases D-G)
    auto mark = [&]() { if (t == 0) st[4 + n_mark] = (int)(wall_clock64() - t_start); ++n_mark; };
    const int wanted = want ? want[b] : 1, npairs = P.pcnt[b], overflow = st[2], cut = st[13];
    if (t == 0) { P.heavy[b] = 0; P.wqn[b] = 0; s_dead = 0; }      // (the chunk queue is consumed: the general kernels start from an empty one)
    if (!wanted || overflow != 0) {             // no collision weight in this column's stage / grid overflow (reported): no pairs
        if (t == 0) { loss_out[b] = 0.f; if (P.over) P.over[b] = 0; }
        return;
    }
    if (force_heavy || !P.fast_ok || npairs > P.pf_cap) {
        if (t == 0) { P.heavy[b] = 1; P.hlist[atomicAdd(P.nheavy, 1)] = b; st[13] = 0; }      // (the general kernels count the cut walks of their own pass)
        return;
    }
    (void)cut;
    __syncthreads();
    const int t_entry = (int)(wall_clock64() - t_start);
    pen_narrow<P2P>(P, b, npairs, PenNarrowLds{reinterpret_cast<unsigned*>(lds), reinterpret_cast<unsigned*>(lds + 2 * PEN_FP), slice, red, &s_dead},
                    verts, sigma, penalize_outside, dverts, loss_out, ap, mark);
    if (t == 0 && P.work) {                     // (debug: sfx_debug_pen_phase_ticks)
        const int s7 = st[7], s8 = st[8], s9 = st[9], s10 = st[10];
        atomicAdd(&P.work[8], (unsigned long long)t_entry); atomicAdd(&P.work[9], (unsigned long long)(s7 - t_entry));
        atomicAdd(&P.work[10], (unsigned long long)(s8 - s7)); atomicAdd(&P.work[11], (unsigned long long)(s9 - s8));
        atomicAdd(&P.work[12], (unsigned long long)(s10 - s9)); atomicAdd(&P.work[13], 1ull); atomicAdd(&P.work[14], (unsigned long long)st[0]);
    }
}

