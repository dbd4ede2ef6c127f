// collide_pairs.h -- broad phase, part 2: pair tests over the grid, partner lists, the canonical pair list (k_pen_walk / walk2 / list / rank)
// Part of csrc/collide.hip (included there, in this order: collide_field.h, collide_grid.h, collide_pairs.h, collide_eval.h);
// one translation unit, compiled with -ffp-contract=off.
#pragma once

// ---- one flat work list over all meshes of a call (k_pen_walk2, k_pen_eval): every workgroup forms the exclusive prefix of the
// meshes' item counts in LDS, a wavefront takes items w, w + W, ... and finds an item's mesh by bisection
#define PEN_FLAT_MAXB 4096      // meshes per call the flat distribution handles (beyond: one grid row per mesh, as before)
#ifndef PEN_FLAT_BLOCKS
#define PEN_FLAT_BLOCKS 2048
#endif
template <class CNT>
__device__ __forceinline__ int pen_prefix(const int B, int* s_pref /* [B + 1] */, int* s_scan /* [256] */, CNT&& count) {
    const int t = threadIdx.x;
    const int per = (B + 255) / 256;
    const int b0 = min(B, t * per), b1 = min(B, b0 + per);
    int sum = 0;
    for (int b = b0; b < b1; ++b) sum += count(b);
    s_scan[t] = sum;
    __syncthreads();
    // 256-entry scan by one wavefront (four per lane), fixed order
    if (t < 64) {
        int v[4], run = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) { v[u] = s_scan[t * 4 + u]; run += v[u]; }
        int inc = run;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(inc, d); if (t >= d) inc += o; }
        int ex = inc - run;
#pragma unroll
        for (int u = 0; u < 4; ++u) { s_scan[t * 4 + u] = ex; ex += v[u]; }
    }
    __syncthreads();
    int acc = s_scan[t];
    for (int b = b0; b < b1; ++b) { s_pref[b] = acc; acc += count(b); }
    if (b1 == B && b0 < B) s_pref[B] = acc;
    if (B == 0 && t == 0) s_pref[0] = 0;
    __syncthreads();
    return s_pref[B];
}
__device__ __forceinline__ int pen_chunk_prefix(const PenDev& P, const int B, int* s_pref, int* s_scan) {
    return pen_prefix(B, s_pref, s_scan, [&](int b_) { return (P.ptotal[b_] + 63) >> 6; });
}
__device__ __forceinline__ int pen_chunk_mesh(const int* s_pref, const int B, const int c) {      // last b with s_pref[b] <= c
    int lo = 0, hi = B - 1;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (s_pref[mid] <= c) lo = mid; else hi = mid - 1; }
    return lo;
}


// ---- pair tests over the bucket-sorted entries of the grid.
// A wavefront takes a BLOCK of 64 consecutive entries of the bucket-sorted list: its lanes hold one entry each (header = entry
// record + AABB: 32 bytes) and lane i tests itself against the entries after it in its bucket -- entry i + d, the same d for
// all lanes, so the headers it needs form a window sliding over the list, held in wavefront-private LDS (two int4 arrays: lane
// i reading entry i + d is conflict-free).  All memory traffic is one gather per ENTRY; the pair tests run on registers and
// LDS.  A pair is accepted in the cell that holds the low corner of the AABB intersection (both triangles are entered there),
// and appended to both triangles' partner lists -- unless the triangles share a vertex, which is looked at when the queue of
// accepted pairs is flushed (neighbours are almost always of one part or of parent and child, which the part mask has already
// turned away: the vertex ids are not worth 16 bytes of every header).
//
// Round 4: the walk of a block is cut into CHUNKS of 64 steps (d = 64 k + 1 .. 64 k + 64; window = entries 64 k .. 64 k + 127
// behind the block's first).  k_pen_walk does chunk 0 of every block -- all a block needs unless a bucket runs past its end --
// and queues the chunks k >= 1; k_pen_walk2 runs the queued chunks of ALL meshes of the call as one flat list, a chunk per
// wavefront.  Until then a block walked its bucket to the end on its own: the launch lasted as long as the block that
// sits at the head of the fullest cell (418 entries on the synthetic surface: 209 dependent iterations and six window refills
// on one wavefront, p50 97 us) while the other 500 wavefronts of the mesh had long finished.  Same candidates, same tests, same
// accepted pairs (their order of arrival differs; the lists are ranked afterwards); the cut after PEN_MAX_WALK steps is the
// chunk limit.
#ifndef PEN_WIN
#define PEN_WIN 128
#endif
#ifndef PEN_NC
#define PEN_NC 2               // candidates per lane and iteration: independent instruction streams cover the LDS / compare latencies
#endif
#define PEN_MAX_CHUNK ((PEN_MAX_WALK + 63) / 64)      // chunks per block (k < PEN_MAX_CHUNK: d <= PEN_MAX_WALK)
static_assert(PEN_WIN == 128 && 64 % PEN_NC == 0, "a chunk's window is two 64-entry halves");

struct PenWalkCtx {            // per wavefront
    int4* tA; int4* tB;        // [PEN_WIN] window: entry | cell | lo.x | lo.y  and  lo.z | hi.x | hi.y | hi.z
    int* queue; int qn;        // accepted pairs waiting to be appended (128 pairs)
    const unsigned long long* s_mask;
};

__device__ __forceinline__ void pen_load_hdr(const int2* ent, const float* aabb, int q, bool ok, int (&hd)[8]) {
    // (every load unconditional, from a clamped index: written as `ok ? p[i] : 0` each of the loads became its own
    //  exec-masked branch with a full s_waitcnt behind it -- serial round trips per header)
    const int qs = ok ? q : 0;
    const int2 e01 = ent[qs];
    const int e0 = e01.x, e1 = e01.y;
    // (the mask goes through inline assembly: from `e0 & 0xffffff` the compiler forms a 24-bit multiply -- which masks
    //  implicitly -- and, once the wide loads below make the row address 64-bit, turns it into v_mad_u64_u32 on the UNMASKED
    //  word: rows 2^24 x part id beyond the array, a memory fault with ROCm 7.2's compiler)
    int f;
    asm("v_and_b32 %0, 0xffffff, %1" : "=v"(f) : "v"(e0));
    const int2* bp = reinterpret_cast<const int2*>(aabb) + (size_t)f * 3;      // the box as three 8-byte loads (rows of 24 bytes)
    const int2 b0 = bp[0], b1 = bp[1], b2 = bp[2];
    hd[0] = ok ? e0 : 0; hd[1] = ok ? e1 : 0x3fffffff;
    hd[2] = ok ? b0.x : 0; hd[3] = ok ? b0.y : 0; hd[4] = ok ? b1.x : 0; hd[5] = ok ? b1.y : 0; hd[6] = ok ? b2.x : 0; hd[7] = ok ? b2.y : 0;
}

__device__ __forceinline__ void pen_flush_queue(const PenDev& P, const int b, PenWalkCtx& W, const int lane) {
    const int n = W.qn;
    if (!n) return;
    int* pc = P.pcount + (size_t)b * P.F;
    int* part = P.partners + (size_t)b * P.F * P.pcap;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int q = lane; q < n; q += 64) {
        const int fa = W.queue[2 * q], fb = W.queue[2 * q + 1];
        const int4 va = P.faces4[fa], vb = P.faces4[fb];      // triangles that share a vertex do not collide
        const bool shared = va.x == vb.x || va.x == vb.y || va.x == vb.z || va.y == vb.x || va.y == vb.y || va.y == vb.z ||
                            va.z == vb.x || va.z == vb.y || va.z == vb.z;
        if (shared) continue;
        const int pa = atomicAdd(&pc[fa], 1), pb = atomicAdd(&pc[fb], 1);
        if (pa < P.pcap) part[(size_t)fa * P.pcap + pa] = fb;
        if (pb < P.pcap) part[(size_t)fb * P.pcap + pb] = fa;
    }
    __builtin_amdgcn_wave_barrier();
    W.qn = 0;
}

// round 5: the accepted pairs go to ONE list of the frame, any order (k_pen_narrow sorts them in LDS) -- one returning atomic per
// flush instead of two per pair; the partner lists are used by the columns k_pen_narrow hands back
__device__ __forceinline__ void pen_flush_pairs(const PenDev& P, const int b, PenWalkCtx& W, const int lane) {
    const int n = W.qn;
    if (!n) return;
    int2* pbuf = P.pbuf + (size_t)b * P.pf_cap;
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
    for (int q0 = 0; q0 < n; q0 += 64) {
        const int q = q0 + lane;
        bool keep = false; int fa = 0, fb = 0;
        if (q < n) {
            fa = W.queue[2 * q]; fb = W.queue[2 * q + 1];
            const int4 va = P.faces4[fa], vb = P.faces4[fb];      // triangles that share a vertex do not collide
            keep = !(va.x == vb.x || va.x == vb.y || va.x == vb.z || va.y == vb.x || va.y == vb.y || va.y == vb.z ||
                     va.z == vb.x || va.z == vb.y || va.z == vb.z);
        }
        const unsigned long long m = __ballot(keep);
        if (!m) continue;
        int base = 0;
        if (lane == 0) base = atomicAdd(&P.pcnt[b], __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < P.pf_cap) pbuf[pos] = make_int2(fa, fb);
    }
    __builtin_amdgcn_wave_barrier();
    W.qn = 0;
}

// The block's own side of the tests: what a lane knows about ITS entry
struct PenOwn { int qi, fi, ck, bend; unsigned need; unsigned long long skip_i; float ai[6]; };

__device__ __forceinline__ int pen_bucket_of(int ck) {
    return (int)(((unsigned)(ck & 1023) * 73856093u ^ (unsigned)((ck >> 10) & 1023) * 19349663u ^ (unsigned)((ck >> 20) & 1023) * 83492791u) & (PEN_CELLS - 1));
}

// headers of block i0 -> the lane's own record (and its header words, for the window of chunk 0)
// (cells: the frame's bucket END offsets -- global memory for the general kernels, the per-frame kernel's LDS copy on the fast path)
__device__ __forceinline__ PenOwn pen_own(const PenDev& P, const int b, const int i0, const int s_total, const PenWalkCtx& W,
                                          const int lane, int (&hi_)[8], const int* cells) {
    const float* aabb = P.aabb + (size_t)b * P.F * 6;
    const int2* ent = P.entries + (size_t)b * P.ent_cap;
    PenOwn O;
    O.qi = i0 + lane;
    const bool vi = O.qi < s_total;
    pen_load_hdr(ent, aabb, O.qi, vi, hi_);
    O.fi = hi_[0] & 0xffffff;
    O.skip_i = vi ? W.s_mask[(hi_[0] >> 24) & 63] : ~0ull;
#pragma unroll
    for (int e = 0; e < 6; ++e) O.ai[e] = __int_as_float(hi_[2 + e]);
    O.ck = hi_[1] & 0x3fffffff;
    // partners of an entry: the entries after it up to the end of ITS bucket
    O.bend = vi ? cells[pen_bucket_of(O.ck)] : 0;
    // Ownership: a pair is accepted in the cell that holds the low corner of the boxes' intersection.  Both triangles are
    // entered in THIS cell, so on every axis the cells of both low corners are <= this cell's coordinate, and (the cell
    // function is monotone) cell(max(a, k)) == c  <=>  cell(a) == c or cell(k) == c.  Whether an entry's cell holds its
    // box's low corner on an axis is a bit of the entry record (k_pen_g3: bits 30, 31 of the key, bit 30 of the triangle
    // word): `need` has the axes where this lane's own corner is elsewhere -- there the partner's must be here.
    const unsigned lowb = ((unsigned)hi_[1] >> 30) | (((unsigned)hi_[0] >> 28) & 4u);      // x | y << 1 | z << 2
    O.need = ~lowb & 7u;
    return O;
}

// chunk k of the block at i0: steps d = 64 k + 1 .. 64 k + 64.  own_hdr: the block's own header words (chunk 0: they are the
// first half of the window and are not loaded again).
template <class FLUSH>
__device__ __forceinline__ void pen_walk_chunk(const PenDev& P, const int b, const int i0, const int k, const int bend_max,
                                               const PenOwn& O, const int (&own_hdr)[8], PenWalkCtx& W, const int lane, FLUSH&& flush) {
    const float* aabb = P.aabb + (size_t)b * P.F * 6;
    const int2* ent = P.entries + (size_t)b * P.ent_cap;
    const int w0 = i0 + 64 * k;                    // entry in window slot 0
    __builtin_amdgcn_wave_barrier();               // (the previous chunk's reads of the window are done)
    {
        int h0[8], h1[8];
        if (k == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) h0[e] = own_hdr[e];
        } else pen_load_hdr(ent, aabb, w0 + lane, w0 + lane < bend_max, h0);
        pen_load_hdr(ent, aabb, w0 + 64 + lane, w0 + 64 + lane < bend_max, h1);
        W.tA[lane] = make_int4(h0[0], h0[1], h0[2], h0[3]); W.tB[lane] = make_int4(h0[4], h0[5], h0[6], h0[7]);
        W.tA[64 + lane] = make_int4(h1[0], h1[1], h1[2], h1[3]); W.tB[64 + lane] = make_int4(h1[4], h1[5], h1[6], h1[7]);
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
    // one candidate of this lane: window entry (lane + dd)
    auto test = [&](const bool act, const int4 h0, const int4 h1) {
        // (every test is evaluated, the results are combined with `&`: written with `&&` the compiler nests one exec-masked
        //  branch per condition -- five s_and_saveexec / s_cbranch_execz pairs per candidate, the second LDS read inside them)
        const float kl0 = __int_as_float(h0.z), kl1 = __int_as_float(h0.w), kl2 = __int_as_float(h1.x);
        const float kh0 = __int_as_float(h1.y), kh1 = __int_as_float(h1.z), kh2 = __int_as_float(h1.w);
        const bool same = ((h0.y ^ O.ck) & 0x3fffffff) == 0;
        const bool coll = ((unsigned)(O.skip_i >> ((h0.x >> 24) & 63)) & 1u) == 0u;
        const bool box = (O.ai[0] <= kh0) & (kl0 <= O.ai[3]) & (O.ai[1] <= kh1) & (kl1 <= O.ai[4]) & (O.ai[2] <= kh2) & (kl2 <= O.ai[5]);
        const unsigned klow = ((unsigned)h0.y >> 30) | (((unsigned)h0.x >> 28) & 4u);
        const bool own = (O.need & ~klow) == 0u;
#ifdef PEN_COUNT    // diagnostic build: where do the candidates die?  stats[16..19] = walked, same cell, part mask passed, boxes overlap
        {
            int* st = P.stats + b * PEN_STATS;
            const unsigned long long m0 = __ballot(act), m1 = __ballot(act & same), m2 = __ballot(act & same & coll), m3 = __ballot(act & same & coll & box);
            if (lane == 0) { atomicAdd(&st[16], __popcll(m0)); atomicAdd(&st[17], __popcll(m1)); atomicAdd(&st[18], __popcll(m2)); atomicAdd(&st[19], __popcll(m3));
                             atomicAdd(&st[20], 1); }      // [20] wavefront steps
        }
#endif
        return act & same & coll & box & own;
    };
    // accepted pairs go to a wavefront-private queue and are appended to the partner lists
    // 64 at a time: the list cursors are returning atomics, one memory round trip each
    auto push = [&](const bool pass, const int other) {
        const unsigned long long m = __ballot(pass);
        if (m) {
            const int pos = W.qn + __popcll(m & ((1ull << lane) - 1ull));
            if (pass) { W.queue[2 * pos] = O.fi; W.queue[2 * pos + 1] = other & 0xffffff; }
            W.qn += __popcll(m);
            if (W.qn >= 64) flush(W);
        }
    };
    for (int dd = 1; dd <= 64; dd += PEN_NC) {      // dd = d - 64 k
        const int d = 64 * k + dd;
        if (!__ballot(O.qi + d < O.bend)) break;
        int4 hA[PEN_NC], hB[PEN_NC];
#pragma unroll
        for (int c = 0; c < PEN_NC; ++c) { const int kk = lane + dd + c; hA[c] = W.tA[kk & (PEN_WIN - 1)]; hB[c] = W.tB[kk & (PEN_WIN - 1)]; }
        bool ps[PEN_NC];
#pragma unroll
        for (int c = 0; c < PEN_NC; ++c) ps[c] = test(O.qi + d + c < O.bend, hA[c], hB[c]);
#pragma unroll
        for (int c = 0; c < PEN_NC; ++c) push(ps[c], hA[c].x);
    }
}

#define PEN_WALK_LDS                                                                                          \
    __shared__ __align__(16) int s_tile[4 * PEN_WIN * 8];    /* per wavefront: a window of PEN_WIN entry headers (32 bytes each) */ \
    __shared__ int s_queue[4 * 256];                                                                          \
    __shared__ unsigned long long s_mask[64];

// chunk 0 of every block; PEN_WALK_BLOCKS workgroups per mesh.  Queues the chunks k >= 1 (P.wq / P.wqn); when the queue is full
// the block walks them itself, as it did before round 4.
__global__ __launch_bounds__(256)
void k_pen_walk(PenDev P, PenSel sel, int to_pbuf, int flatB) {
    PEN_WALK_LDS
    extern __shared__ int s_wpref[];            // (flat) [flatB + 1] exclusive prefix of the columns' blocks of 64 entries
    __shared__ int s_wscan[256];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int b_first = pen_sel_first(sel, blockIdx.y);
    const int nsel = pen_sel_n(sel, sel.hlist ? 0x7fffffff : (int)gridDim.y);
    if (nsel == 0 && !flatB) return;
    if (t < 64) s_mask[t] = P.skipmask[t];
    __syncthreads();
    PenWalkCtx W;
    W.tA = reinterpret_cast<int4*>(s_tile + wv * PEN_WIN * 8); W.tB = W.tA + PEN_WIN;
    W.queue = s_queue + wv * 256; W.qn = 0; W.s_mask = s_mask;
    // chunk 0 of the block of 64 entries that starts at i0 of column b; its later chunks are queued for k_pen_walk2
    auto walk_block = [&](const int b, const int i0, const int s_total, const int* cells) {
        auto flush = [&](PenWalkCtx& W_) { if (to_pbuf) pen_flush_pairs(P, b, W_, lane); else pen_flush_queue(P, b, W_, lane); };
        int hdr[8];
        const PenOwn O = pen_own(P, b, i0, s_total, W, lane, hdr, cells);
        const int bend_max = __builtin_amdgcn_readfirstlane((int)wave_max_dpp((float)O.bend));      // entries < 2^24: exact
        pen_walk_chunk(P, b, i0, 0, bend_max, O, hdr, W, lane, flush);
        // steps this block needs: the longest walk of its lanes, bend - 1 - qi
        const int dmax = __builtin_amdgcn_readfirstlane((int)wave_max_dpp((float)max(O.bend - 1 - O.qi, 0)));
        if (dmax > 64) {
            int kmax = (dmax - 1) >> 6;                    // last chunk with a live step
            if (kmax >= PEN_MAX_CHUNK) {                   // a bucket of thousands of entries: a mesh that has collapsed into a few cells
                kmax = PEN_MAX_CHUNK - 1;
                if (lane == 0) atomicAdd(&P.stats[b * PEN_STATS + 13], 1);      // (reported: sfx_pen_stats, "walks cut short")
            }
            int pos = 0;
            if (lane == 0) pos = atomicAdd(&P.wqn[b], kmax);
            pos = __builtin_amdgcn_readfirstlane(pos);
            if (pos + kmax <= P.wq_cap) {
                if (lane >= 1 && lane <= kmax) P.wq[(size_t)b * P.wq_cap + pos + lane - 1] = make_int2(i0, lane);
            } else {
                // queue full: walk on here.  The reservation is NOT rolled back (round 5; an atomicSub could interleave with a
                // third wavefront's reservation and leave its records beyond the count, stale ones inside it): the count only
                // grows, readers clamp it to the capacity, and the slots of this reservation that lie inside the capacity are
                // filled with records k_pen_walk2 skips (chunk 0 is never queued).
                if (lane >= 1 && lane <= kmax && pos + lane - 1 < P.wq_cap) P.wq[(size_t)b * P.wq_cap + pos + lane - 1] = make_int2(i0, 0);
                for (int k = 1; k <= kmax; ++k) pen_walk_chunk(P, b, i0, k, bend_max, O, hdr, W, lane, flush);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    if (flatB > 0) {
        // (round 5) ONE flat list of the blocks over all columns of the call, a block per wavefront: a body's grid has ~30 blocks of 64
        // entries, and 128 workgroups per column -- sized for a mesh that has collapsed into itself -- sent 120 of them through two
        // loads and out again, each holding a wavefront slot (LAB_NOTES §4.6)
        const int n_items = pen_prefix(flatB, s_wpref, s_wscan, [&](int b_) {
            return pen_sel_on(sel, b_) ? (P.cells[(size_t)b_ * (PEN_CELLS + 1) + PEN_CELLS] + 63) >> 6 : 0; });
        int b_prev = -1;
        for (int c = blockIdx.x * 4 + wv; c < n_items; c += gridDim.x * 4) {
            const int b = pen_chunk_mesh(s_wpref, flatB, c);
            if (b != b_prev) { if (b_prev >= 0) { if (to_pbuf) pen_flush_pairs(P, b_prev, W, lane); else pen_flush_queue(P, b_prev, W, lane); } b_prev = b; }      // (the pair queue belongs to one mesh)
            const int* cells = P.cells + (size_t)b * (PEN_CELLS + 1);
            walk_block(b, (c - s_wpref[b]) * 64, cells[PEN_CELLS], cells);
        }
        if (b_prev >= 0) { if (to_pbuf) pen_flush_pairs(P, b_prev, W, lane); else pen_flush_queue(P, b_prev, W, lane); }
        return;
    }
    for (int si = blockIdx.y; si < nsel; si += gridDim.y) {
    const int b = si == (int)blockIdx.y ? b_first : pen_sel_col(sel, si);
    const int* cells = P.cells + (size_t)b * (PEN_CELLS + 1);
    const int s_total = cells[PEN_CELLS];
    if (!pen_sel_on(sel, b) || blockIdx.x * 256 >= s_total) continue;
    // (blocks of 64 entries, NOT whole buckets: a crowded bucket is shared by many wavefronts; the
    // cell key comparison keeps different cells of one bucket apart)
    for (int i0 = (blockIdx.x * 4 + wv) * 64; i0 < s_total; i0 += gridDim.x * 256) walk_block(b, i0, s_total, cells);
    if (to_pbuf) pen_flush_pairs(P, b, W, lane); else pen_flush_queue(P, b, W, lane);
    }
}

// the queued chunks of all meshes of the call, one flat list (the distribution of k_pen_eval): a chunk per wavefront
__global__ __launch_bounds__(256)
void k_pen_walk2(PenDev P, int B, PenSel sel, int to_pbuf) {
    PEN_WALK_LDS
    extern __shared__ int s_pref[];             // [B + 1] exclusive prefix of the meshes' queued chunks
    __shared__ int s_scan[256];
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    if (sel.hlist && *sel.nheavy == 0) return;          // (only heavy columns queue chunks: k_pen_narrow leaves the others' queues empty)
    if (t < 64) s_mask[t] = P.skipmask[t];
    const int n_items = pen_prefix(B, s_pref, s_scan, [&](int b_) { return min(P.wqn[b_], P.wq_cap); });
    PenWalkCtx W;
    W.tA = reinterpret_cast<int4*>(s_tile + wv * PEN_WIN * 8); W.tB = W.tA + PEN_WIN;
    W.queue = s_queue + wv * 256; W.qn = 0; W.s_mask = s_mask;
    int b_prev = -1;
    for (int c = blockIdx.x * 4 + wv; c < n_items; c += gridDim.x * 4) {
        const int b = pen_chunk_mesh(s_pref, B, c);
        if (b != b_prev) { if (b_prev >= 0) { if (to_pbuf) pen_flush_pairs(P, b_prev, W, lane); else pen_flush_queue(P, b_prev, W, lane); } b_prev = b; }      // (the pair queue belongs to one mesh)
        const int2 it = P.wq[(size_t)b * P.wq_cap + (c - s_pref[b])];
        if (it.y == 0) continue;                       // (a slot of a reservation that did not fit: its block walked on itself)
        const int s_total = P.cells[(size_t)b * (PEN_CELLS + 1) + PEN_CELLS];
        int hdr[8];
        const PenOwn O = pen_own(P, b, it.x, s_total, W, lane, hdr, P.cells + (size_t)b * (PEN_CELLS + 1));
        const int bend_max = __builtin_amdgcn_readfirstlane((int)wave_max_dpp((float)O.bend));
        pen_walk_chunk(P, b, it.x, it.y, bend_max, O, hdr, W, lane, [&](PenWalkCtx& W_) { if (to_pbuf) pen_flush_pairs(P, b, W_, lane); else pen_flush_queue(P, b, W_, lane); });
    }
    if (b_prev >= 0) { if (to_pbuf) pen_flush_pairs(P, b_prev, W, lane); else pen_flush_queue(P, b_prev, W, lane); }
}

// offsets of the triangles' partner ranges in the frame's pair list (k_pen_rank fills the list)
// can a triangle's partners be derived again from the grid by one wavefront (pen_rewalk: an LDS tile of `tcap` ids per wavefront
// of k_pen_rank, which must hold the cap kept ones and a wavefront's worth of new ones)?  True for max_collisions <= 1024.
__device__ __host__ __forceinline__ int pen_rank_tile(const int pcap) { int c = 64; while (c < pcap) c <<= 1; return c > 2048 ? 0 : (c < 128 ? 128 : c); }
__device__ __forceinline__ bool pen_can_rewalk(const PenDev& P) { const int t = pen_rank_tile(P.pcap); return t > 0 && P.cap + 64 <= t; }

#ifndef PEN_SHORT
#define PEN_SHORT 16
#endif
__global__ __launch_bounds__(PEN_T)
void k_pen_list(PenDev P, PenSel sel) {
    extern __shared__ int s_cnt[];             // [F] partner counts of the frame, then [hasp_words] bitmask
    __shared__ float red[PEN_T / 64];
    __shared__ int slice[PEN_T];
    __shared__ int s_nl;
    const int t = threadIdx.x;
    const int nsel = pen_sel_n(sel, sel.hlist ? 0x7fffffff : (int)gridDim.x);
    for (int si = blockIdx.x; si < nsel; si += gridDim.x) {
    const int b = pen_sel_col(sel, si);
    __syncthreads();                                 // (the previous column's reads of the staging arrays are done)
    int* st = P.stats + b * PEN_STATS;
    unsigned* hasp = P.hasp + (size_t)b * P.hasp_words;
    if (!pen_sel_on(sel, b) || st[2] != 0) {         // skipped frame / grid overflow: the grid build has zeroed the totals
        for (int w = t; w < P.hasp_words; w += PEN_T) hasp[w] = 0u;
        if (P.over && t == 0) P.over[b] = 0;
        if (t == 0) { P.nrb[b] = 0; P.nlq[b] = 0; }
        continue;
    }
    const int F = P.F;
    int* pc = P.pcount + (size_t)b * F;

    // ---- the frame's pair list: triangles ascending, partners ascending within a triangle (the
    // partner lists were appended in scheduling order; ranking them here fixes every later summation
    // order).  A triangle with more than max_collisions partners keeps the max_collisions LOWEST triangle
    // ids (the package the reference calls keeps the ones its BVH traversal meets first: implementation
    // defined there; a rule on ids does not depend on scheduling or on the other frames of the batch).  The lists
    // hold up to pcap = 2 x max_collisions partners while they are collected; only beyond that is the choice
    // left to arrival order.  Cut partners, and pairs beyond pair_cap, are counted.
    int* poff = P.poff + (size_t)b * F;
    int* pav = P.pavail + (size_t)b * F;
    unsigned* s_has = reinterpret_cast<unsigned*>(s_cnt + F);
    // (every lane owns a contiguous run of triangles for the scan; read straight from global memory those runs are 84-byte
    //  strides across the lanes and two chains of 21 dependent loads -- the counts are staged through LDS coalesced instead)
    // (round 4: every global access of this kernel is coalesced -- the clamped counts are written in the pass that reads the raw
    //  ones, the offsets go to LDS in place and leave in a pass of their own; the per-lane runs of 21 triangles used to write
    //  three arrays at a stride of 84 bytes across the lanes: 63 store instructions of 64 cache lines each, most of the kernel)
    if (t == 0) { P.ovn[b * 2] = 0; P.ovn[b * 2 + 1] = 0; s_nl = 0; }
    __syncthreads();
    // (round 5: the counts eight at a time, from clamped indices -- a trip of this loop was load, then stores the compiler cannot
    //  move the next load across: 21 dependent round trips for a body's 20 908 triangles, most of this kernel's 19 us)
    constexpr int LU = 8;
    for (int f0 = t; f0 < F; f0 += PEN_T * LU) {
        int raws[LU];
#pragma unroll
        for (int u = 0; u < LU; ++u) raws[u] = pc[min(f0 + u * PEN_T, F - 1)];
#pragma unroll
        for (int u = 0; u < LU; ++u) {
            const int f = f0 + u * PEN_T, raw = raws[u];
            const bool in = f < F;
            if (in) {
                s_cnt[f] = raw;
                pav[f] = raw;                        // (uncapped: > pcap tells k_pen_rank that the held list is incomplete)
                pc[f] = min(raw, P.cap);
                if (raw > P.pcap) { P.ovq[(size_t)b * F + atomicAdd(&P.ovn[b * 2], 1)] = f; P.callno[1] = P.callno[0]; }      // (rare; the order of the queue is immaterial)
                if (min(raw, P.cap) > PEN_SHORT || raw > P.cap) P.lq[(size_t)b * F + atomicAdd(&s_nl, 1)] = f;      // (what k_pen_rank calls a long list: a work item of its own there)
            }
            // (round 5) does this block of 64 consecutive triangles -- the wavefront's lanes of this trip -- have partners at all?
            // -> k_pen_rank's flat work list (a body: ~30 blocks of 327)
            const unsigned long long any = __ballot(in && raw > 0);
            if ((t & 63) == 0 && in) slice[f >> 6] = any ? 1 : 0;
        }
    }
    for (int w = t; w < P.hasp_words; w += PEN_T) s_has[w] = 0u;
    __syncthreads();
    if (t < 64) {
        int cnt = 0;
        for (int base = 0; base < P.n_clus; base += 64) {
            const int j = base + t;
            const bool w_ = j < P.n_clus && slice[j] != 0;
            const unsigned long long m = __ballot(w_);
            if (w_) P.rb[(size_t)b * P.n_clus + cnt + __popcll(m & ((1ull << t) - 1ull))] = j;
            cnt += __popcll(m);
        }
        if (t == 0) { P.nrb[b] = cnt; P.nlq[b] = s_nl; }
    }
    __syncthreads();
    {
        const int per = (F + PEN_T - 1) / PEN_T;
        const int f0 = min(F, t * per), f1 = min(F, f0 + per);
        int sum = 0, n_over = 0, n_arr = 0;
        for (int f = f0; f < f1; ++f) { const int cnt = s_cnt[f]; n_over += max(cnt - P.cap, 0); sum += min(cnt, P.cap); n_arr += cnt > P.pcap ? 1 : 0; }
        int ptot;
        int acc = block_excl_scan(sum, slice, &ptot);
        for (int f = f0; f < f1; ++f) {
            const int raw = s_cnt[f];
            const int c = min(raw, P.cap);
            const int keep = max(0, min(c, P.pair_cap - acc));
            n_over += c - keep;
            s_cnt[f] = acc;                      // the triangle's offset (readers cut at pair_cap: kept = clamp(pair_cap - poff, 0, pcount))
            if (c > 0 && acc < P.pair_cap) atomicOr(&s_has[f >> 5], 1u << (f & 31));
            acc += c;
        }
        const float to = block_sum_fixed((float)n_over, red);
        const float ta = block_sum_fixed((float)n_arr, red);
        if (t == 0) { const int tot = min(ptot, P.pair_cap); P.ptotal[b] = tot; st[0] = tot; st[1] = (int)to;
                      // a mesh with thousands of such triangles has collapsed onto itself (a diverged fit on its way to NaN): looking at
                      // each of them again would cost milliseconds per evaluation for a term that means nothing there -- such a mesh
                      // keeps its first arrivals, as every overflowing list did until round 4, and is reported as order dependent
                      const bool rewalk = pen_can_rewalk(P) && ta <= (float)PEN_REWALK_MAX;
                      if (!rewalk) P.ovn[b * 2] = 0;
                      else if (ta > 0.f) P.ovm[1 + atomicAdd(&P.ovm[0], 1)] = b;      // (k_pen_rank drains the queues of the meshes listed here)
                      if (P.over) P.over[b] = ((ta > 0.f && !rewalk) || st[13] > 0) ? 1 : 0;      // (lists beyond pcap are re-derived by k_pen_rank: pen_rewalk)
                      if (P.work) { atomicAdd(&P.work[1], (unsigned long long)tot);
                                    if (ta > 0.f) atomicAdd(&P.work[4], (unsigned long long)ta);
                                    if (st[13] > 0) atomicAdd(&P.work[5], (unsigned long long)st[13]); } }
    }
    __syncthreads();
    for (int f = t; f < F; f += PEN_T) poff[f] = s_cnt[f];
    __syncthreads();
    for (int w = t; w < P.hasp_words; w += PEN_T) hasp[w] = s_has[w];
    }
}

// A triangle that met more partners than its list holds (pcap = 2 x max_collisions; only a mesh pushed through itself has such
// triangles) kept the first pcap ARRIVALS -- which ones depends on scheduling.  Its kept partners are therefore derived again,
// by one wavefront, from the grid itself: every entry of every cell the triangle's box touches goes through the tests of the
// pair walk (same cell, part mask, boxes, ownership of the pair by this cell, no shared vertex), the accepted ids are collected
// in the wavefront's LDS tile and cut to the `cap` LOWEST whenever the tile fills up.  Result: tile[0 .. n) ascending, n =
// min(partners, cap) -- the rule of the lists that fit (k_pen_list), now without exception: the pair set no longer depends on
// arrival order anywhere (a cut bucket walk, reported separately, remains the only approximation).
__device__ __forceinline__ void pen_tile_sort(int* tile, const int np, const int lane) {      // ascending bitonic sort of tile[0 .. np), np a power of two >= 64
    for (int k = 2; k <= np; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < np; i += 64) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const int va = tile[i], vb = tile[ixj];
                    if ((va > vb) == ((i & k) == 0)) { tile[i] = vb; tile[ixj] = va; }
                }
            }
        }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
}
#ifndef PEN_RW
#define PEN_RW 8
#endif
__device__ __forceinline__ int pen_rewalk(const PenDev& P, const int b, const int f, int* tile, const int tcap /* power of two >= cap + 64 */, const int lane) {
    const float* aabb = P.aabb + (size_t)b * P.F * 6;
    const int2* ent = P.entries + (size_t)b * P.ent_cap;
    const int* cells = P.cells + (size_t)b * (PEN_CELLS + 1);
    PenGridCtx C;
    { const float* gp = P.gridp + b * 4; C.glo[0] = gp[0]; C.glo[1] = gp[1]; C.glo[2] = gp[2]; C.ih = gp[3]; }      // (k_pen_g2 left the frame's grid here)
    float bx[6];
#pragma unroll
    for (int e = 0; e < 6; ++e) bx[e] = aabb[(size_t)f * 6 + e];
    const int seg = P.segm[f];
    const unsigned long long skip_f = P.skipmask[seg];
    const int4 vf = P.faces4[f];
    int2 pk;
    {
        int c0[3], sp[3];
#pragma unroll
        for (int e = 0; e < 3; ++e) { c0[e] = pen_cell_of(C, bx[e], e); sp[e] = min(pen_cell_of(C, bx[3 + e], e), c0[e] + PEN_SPAN - 1) - c0[e]; }
        pk.x = (c0[0] & 1023) | ((c0[1] & 1023) << 10) | ((c0[2] & 1023) << 20);
        pk.y = sp[0] | (sp[1] << 3) | (sp[2] << 6);
    }
    int n = 0;                                  // ids in the tile (wave-uniform)
    // The scan is flat over (cell, 64-entry chunk of its bucket) items, PEN_RW of them in flight at a time: a lane looks up one
    // cell's bucket range (64 cells per step), a prefix scan numbers the chunks, and item t is found by a ballot.  (Cell after
    // cell it was three dependent memory round trips per cell -- bucket range, entry records, their boxes -- 27 to 512 times.)
    const int x0 = pk.x & 1023, y0 = (pk.x >> 10) & 1023, z0 = (pk.x >> 20) & 1023;
    const int ncx = (pk.y & 7) + 1, ncy = ((pk.y >> 3) & 7) + 1, ncz = ((pk.y >> 6) & 7) + 1, ncell = ncx * ncy * ncz;
    for (int cb = 0; cb < ncell; cb += 64) {
        const int ci = cb + lane;
        const bool cv = ci < ncell;
        const int dx = ci % ncx, dy = (ci / ncx) % ncy, dz = ci / (ncx * ncy);
        const int cx = (x0 + dx) & 1023, cy = (y0 + dy) & 1023, cz = (z0 + dz) & 1023;
        const int key_l = cx | (cy << 10) | (cz << 20);
        const int lowf_l = (int)(dx == 0) | ((int)(dy == 0) << 1) | ((int)(dz == 0) << 2);
        const int bucket = cv ? pen_bucket(cx, cy, cz) : 0;
        const int eb_ld = cells[bucket > 0 ? bucket - 1 : 0], ee_ld = cells[bucket];
        const int eb_l = cv ? (bucket > 0 ? eb_ld : 0) : 0, ee_l = cv ? ee_ld : 0;
        const int nch_l = (ee_l - eb_l + 63) >> 6;
        const int incl = wave_incl_scan_dpp(nch_l);
        const int T = __builtin_amdgcn_readlane(incl, 63);
        for (int t0 = 0; t0 < T; t0 += PEN_RW) {
            int keyu[PEN_RW], lowu[PEN_RW]; bool okc[PEN_RW]; int2 rec[PEN_RW];
#pragma unroll
            for (int u = 0; u < PEN_RW; ++u) {
                const int t = min(t0 + u, T - 1);
                const int l = __ffsll((long long)__ballot(incl > t)) - 1;              // the cell that holds chunk t
                const int k = t - (__builtin_amdgcn_readlane(incl, l) - __builtin_amdgcn_readlane(nch_l, l));
                const int eb = __builtin_amdgcn_readlane(eb_l, l) + 64 * k, ee = __builtin_amdgcn_readlane(ee_l, l);
                keyu[u] = __builtin_amdgcn_readlane(key_l, l); lowu[u] = __builtin_amdgcn_readlane(lowf_l, l);
                okc[u] = (t0 + u < T) & (eb + lane < ee);
                rec[u] = ent[eb + lane < ee ? eb + lane : eb];
            }
            int hd[PEN_RW][8];
#pragma unroll
            for (int c = 0; c < PEN_RW; ++c) {
                int g_;
                asm("v_and_b32 %0, 0xffffff, %1" : "=v"(g_) : "v"(rec[c].x));      // (see pen_load_hdr: the mask must not be folded into the address arithmetic)
                const int2* bp = reinterpret_cast<const int2*>(aabb) + (size_t)g_ * 3;
                const int2 b0 = bp[0], b1 = bp[1], b2 = bp[2];
                hd[c][0] = rec[c].x; hd[c][1] = rec[c].y;
                hd[c][2] = b0.x; hd[c][3] = b0.y; hd[c][4] = b1.x; hd[c][5] = b1.y; hd[c][6] = b2.x; hd[c][7] = b2.y;
            }
            bool pass[PEN_RW]; int gid[PEN_RW];
#pragma unroll
            for (int c = 0; c < PEN_RW; ++c) {
                const int g = hd[c][0] & 0xffffff;
                const bool same = ((hd[c][1] ^ keyu[c]) & 0x3fffffff) == 0;
                const bool coll = ((unsigned)(skip_f >> ((hd[c][0] >> 24) & 63)) & 1u) == 0u;
                const float kl0 = __int_as_float(hd[c][2]), kl1 = __int_as_float(hd[c][3]), kl2 = __int_as_float(hd[c][4]);
                const float kh0 = __int_as_float(hd[c][5]), kh1 = __int_as_float(hd[c][6]), kh2 = __int_as_float(hd[c][7]);
                const bool box = (bx[0] <= kh0) & (kl0 <= bx[3]) & (bx[1] <= kh1) & (kl1 <= bx[4]) & (bx[2] <= kh2) & (kl2 <= bx[5]);
                const unsigned klow = ((unsigned)hd[c][1] >> 30) | (((unsigned)hd[c][0] >> 28) & 4u);
                const bool own = (((unsigned)lowu[c] | klow) & 7u) == 7u;      // on every axis one of the two boxes has its low corner in this cell
                pass[c] = okc[c] & same & coll & box & own & (g != f);
                gid[c] = g;
            }
#pragma unroll
            for (int c = 0; c < PEN_RW; ++c) {
                if (!__ballot(pass[c])) continue;
                const int4 vg = P.faces4[pass[c] ? gid[c] : f];
                const bool shared = vf.x == vg.x || vf.x == vg.y || vf.x == vg.z || vf.y == vg.x || vf.y == vg.y || vf.y == vg.z ||
                                    vf.z == vg.x || vf.z == vg.y || vf.z == vg.z;
                const bool keepit = pass[c] & !shared;
                const unsigned long long m = __ballot(keepit);
                if (!m) continue;
                const int add = __popcll(m);
                if (n + add > tcap) {                // cut to the cap lowest ids, then go on collecting
                    for (int q = n + lane; q < tcap; q += 64) tile[q] = 0x7fffffff;
                    pen_tile_sort(tile, tcap, lane);
                    n = min(n, P.cap);
                }
                if (keepit) tile[n + __popcll(m & ((1ull << lane) - 1ull))] = gid[c];
                n += add;
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
            }
        }
    }
    for (int q = n + lane; q < tcap; q += 64) tile[q] = 0x7fffffff;
    pen_tile_sort(tile, tcap, lane);
    return min(n, P.cap);
}

// ranks every triangle's partner list into the frame's pair list; PEN_RANK_BLOCKS workgroups per frame
#ifndef PEN_RANK_FLAT
#define PEN_RANK_FLAT 512       // workgroups of the flat form of k_pen_rank (a block of 64 triangles with pairs per wavefront)
#endif
#ifndef PEN_RANK_BLOCKS
#define PEN_RANK_BLOCKS 64
#endif
#ifndef PEN_RANK_HELPERS
#define PEN_RANK_HELPERS 8
#endif
#ifndef PEN_SHORT
#define PEN_SHORT 16            // lists up to this length are ranked element-wise, longer ones sorted by a wavefront
#endif
#ifndef PEN_RANK_OCC
#define PEN_RANK_OCC 1
#endif
__global__ __launch_bounds__(256, PEN_RANK_OCC)
void k_pen_rank(PenDev P, PenSel sel, int cap_pad, int flatB) {
    extern __shared__ int s_sort[];             // [4][max(cap_pad, 128)], then (flat) [flatB + 1]: exclusive prefix of the columns' blocks with pairs
    const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int b_first = pen_sel_first(sel, blockIdx.y);
    const int nsel = pen_sel_n(sel, sel.hlist ? 0x7fffffff : (int)gridDim.y);
    if (nsel == 0 && !flatB) return;
    const int F = P.F;
    const int tcap = min(max(cap_pad, 128), 2048);
    int* tile = s_sort + wv * tcap;
#ifdef PEN_RANKT
    const long long rt0 = wall_clock64(); long long rt_long = 0, rt_ld = 0, rt_short = 0; int n_long = 0, n_rew = 0;
#endif
    // (round 5: the few words EVERY wavefront of the launch wants -- is this column selected, has it pairs, has it a queue; below:
    //  has any mesh a queue -- are fetched by ONE lane per workgroup and handed on through LDS.  34 k wavefronts asking the same
    //  handful of cache lines at the same moment queue up behind each other at one L2 channel: measured with -DPEN_RANKT, the
    //  launch's slow wavefronts spent 50 us on such loads and 12 us on their lists)
    __shared__ int s_u[4];
    __shared__ int s_scan[256];
    const bool can_sort = cap_pad <= 2048;
    // one long list (more than PEN_SHORT partners, or cut): the wavefront ranks the held partners in its LDS tile and keeps the cc lowest
    auto rank_one = [&](const int* part, int* pown, int* plist, const int ff, const int cc, const int off, const int found, const int (&x)[4]) {
            const int av = min(found, P.pcap);                         // sort all av held partners, keep the cc lowest
            const int* mine = part + (size_t)ff * P.pcap;
            int np = 64;
            while (np < av) np <<= 1;
            if (np <= 256 && np <= tcap) {
                // up to 256 partners (2 x the cfgs' max_collisions: what a list holds while it is collected).  Round 5: ranked, not
                // sorted -- the list goes to the wavefront's LDS tile once, every lane counts how many of its values are smaller
                // than each of its own (all lanes read the same word: a broadcast, no dependence between the reads) and stores its
                // values at their ranks; partner ids are distinct.  The bitonic network it replaces was 28-45 DEPENDENT cross-lane
                // exchanges per list (~3 us), and a collapsed mesh brings blocks of 64 such lists.
                const int R = np >> 6;
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < R) tile[lane + 64 * r] = x[r];
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
                int rk[4] = {0, 0, 0, 0};
                for (int i = 0; i < av; i += 4) {
                    const int4 v4 = *reinterpret_cast<const int4*>(tile + i);        // (entries beyond av are 0x7fffffff: never smaller)
#pragma unroll
                    for (int r = 0; r < 4; ++r) rk[r] += (int)(v4.x < x[r]) + (int)(v4.y < x[r]) + (int)(v4.z < x[r]) + (int)(v4.w < x[r]);
                }
                const int keep = min(cc, P.pair_cap - off);
#pragma unroll
                for (int r = 0; r < 4; ++r) if (r < R && lane + 64 * r < av && rk[r] < keep) { plist[off + rk[r]] = x[r]; pown[off + rk[r]] = ff; }
                __builtin_amdgcn_wave_barrier();
                return;
            }

            for (int q = lane; q < np; q += 64) tile[q] = q < av ? mine[q] : 0x7fffffff;
            for (int k = 2; k <= np; k <<= 1)
                for (int j = k >> 1; j > 0; j >>= 1) {
                    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
                    for (int i = lane; i < np; i += 64) {
                        const int ixj = i ^ j;
                        if (ixj > i) {
                            const int va = tile[i], vb = tile[ixj];
                            if ((va > vb) == ((i & k) == 0)) { tile[i] = vb; tile[ixj] = va; }
                        }
                    }
                }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
            const int keep = min(cc, P.pair_cap - off);
            for (int q = lane; q < keep; q += 64) { plist[off + q] = tile[q]; pown[off + q] = ff; }
            __builtin_amdgcn_wave_barrier();
    };
    // one block of 64 consecutive triangles of column b: short lists by a lane each, long ones by the wavefront (skip_long: they
    // are work items of their own)
    auto rank_block = [&](const int b, const int fw, const bool rewalk_b, const bool skip_long) {
    const int* pc = P.pcount + (size_t)b * F;
    const int* poff = P.poff + (size_t)b * F;
    const int* part = P.partners + (size_t)b * F * P.pcap;
    const int* pav = P.pavail + (size_t)b * F;
    int* pown = P.pown + (size_t)b * P.pair_cap;
    int* plist = P.plist + (size_t)b * P.pair_cap;
    const int flim = F;
    {
        const int f = fw + lane;
        const bool inr = f < flim;
        const int fs = inr ? f : 0;                    // (unconditional loads from a clamped index: three loads in flight, not three round trips)
        const int c_ld = pc[fs], o_ld = poff[fs], a_ld = pav[fs];
        const int c_l = inr ? c_ld : 0, off_l = inr ? o_ld : 0x3fffffff;
        const int a_l = inr ? a_ld : 0;                // partners held (> c_l: the list is cut to its c_l lowest ids)
        const int base = __builtin_amdgcn_readfirstlane(off_l);
        const int lastv = min(63, flim - 1 - fw);
#ifdef PEN_RANKT
        const long long rq0 = wall_clock64();
#endif
        const int E = __builtin_amdgcn_readlane(off_l + c_l, lastv) - base;
#ifdef PEN_RANKT
        rt_ld += wall_clock64() - rq0;
#endif
        if (E == 0) return;
        __builtin_amdgcn_wave_barrier();
#ifdef PEN_RANKT
        const long long rs0 = wall_clock64();
#endif
        if (can_sort) {
            // Short lists (<= PEN_SHORT partners, not cut): ONE LANE PER TRIANGLE -- the lane fetches its whole list in one round trip,
            // ranks its values against each other in registers and stores them at their ranks.  (Until round 5 the block's ELEMENTS
            // were dealt to the lanes, 64 per trip of a loop, every trip paying its own dependent loads: a block in a hand region --
            // 64 triangles x ~10 partners -- was ten trips; measured with -DPEN_RANKT: the wavefronts beyond 40 us spent 56 us in this
            // pass and 1.6 of them on long lists.)
            const bool mine_short = inr && c_l > 0 && c_l <= PEN_SHORT && a_l <= c_l && off_l < P.pair_cap;
            if (mine_short) {
                const int* mine = part + (size_t)f * P.pcap;
                int y[PEN_SHORT];
#pragma unroll
                for (int r = 0; r < PEN_SHORT; ++r) y[r] = mine[min(r, P.pcap - 1)];
#pragma unroll
                for (int s_ = 0; s_ < PEN_SHORT; ++s_) {
                    if (s_ < c_l) {
                        int rank = 0;
#pragma unroll
                        for (int r = 0; r < PEN_SHORT; ++r) rank += (int)((r < c_l) & ((y[r] < y[s_]) | ((y[r] == y[s_]) & (r < s_))));
                        if (off_l + rank < P.pair_cap) { plist[off_l + rank] = y[s_]; pown[off_l + rank] = f; }
                    }
                }
            }
        } else {
        tile[lane] = off_l - base;
        tile[64 + lane] = c_l | (a_l > c_l ? 0x10000 : 0);       // (kept count and "the list was cut" of the 64 triangles: no second trip to memory for them)
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
        for (int e = lane; e < E; e += 64) {            // (max_collisions beyond the wavefront sort: element-wise all the way, as before)
            int l = 0;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) if (tile[l + d] <= e) l += d;      // last l with offset <= e
            const int ff = fw + l, lo = tile[l], slot = e - lo, off = base + lo;
            const int cc = tile[64 + l] & 0xffff;
            const int* mine = part + (size_t)ff * P.pcap;
            const int x = mine[slot];
            int rank = 0;
            for (int r = 0; r < cc; ++r) { const int yy = mine[r]; rank += (int)((yy < x) | ((yy == x) & (r < slot))); }
            if (off + rank < P.pair_cap) { plist[off + rank] = x; pown[off + rank] = ff; }
        }
        }
        __builtin_amdgcn_wave_barrier();
#ifdef PEN_RANKT
        rt_short += wall_clock64() - rs0;
#endif
        unsigned long long m = skip_long ? 0ull : __ballot(can_sort && (c_l > PEN_SHORT || a_l > c_l) && off_l < P.pair_cap);
        // (round 5: the first 128 partners of the NEXT long list of the block are fetched while this one is sorted -- such lists
        //  come in crowds, 64 of a block's 64 triangles in a collapsed mesh, and a load -> sort -> store chain per list made the
        //  block's wavefront the launch's long pole: ~3 us per list, 2 of them waiting for memory)
        int nx[4] = {0x7fffffff, 0x7fffffff, 0x7fffffff, 0x7fffffff};
        auto fetch = [&](const unsigned long long mm) {
            if (!mm) return;
            const int bit_ = __ffsll((long long)mm) - 1;
            const int av_ = min(__builtin_amdgcn_readlane(a_l, bit_), P.pcap);
            const int* mine_ = part + (size_t)(fw + bit_) * P.pcap;
            int l_[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) l_[r] = mine_[min(lane + 64 * r, P.pcap - 1)];
#pragma unroll
            for (int r = 0; r < 4; ++r) nx[r] = lane + 64 * r < av_ ? l_[r] : 0x7fffffff;
        };
        fetch(m);
#ifdef PEN_RANKT
        const long long rl0 = wall_clock64(); n_long += __popcll(m);
#endif
        while (m) {
            const int bit = __ffsll((long long)m) - 1;
            m &= m - 1;
            const int ff = fw + bit;
            const int cc = __builtin_amdgcn_readlane(c_l, bit), off = __builtin_amdgcn_readlane(off_l, bit);
            const int found = __builtin_amdgcn_readlane(a_l, bit);
            int x[4] = {nx[0], nx[1], nx[2], nx[3]};
            fetch(m);
            if (found > P.pcap && rewalk_b) continue;                  // incomplete list: queued by k_pen_list, taken below
            rank_one(part, pown, plist, ff, cc, off, found, x);
        }
#ifdef PEN_RANKT
        rt_long += wall_clock64() - rl0;
#endif
    }
    };
    if (flatB > 0) {
    // (round 5) ONE flat list of the blocks that HAVE pairs over all columns of the call (k_pen_list leaves them per column: P.rb /
    // P.nrb), a block per wavefront: a body's ~400 triangles with partners sit in ~30 of its 327 blocks, and a grid of 64
    // workgroups per column sent nine wavefronts in ten through three loads and out again, each holding a wavefront slot that
    // a busy one was waiting for (LAB_NOTES §4.6: the step is bound by slots x round trips)
    int* s_pref = s_sort + 4 * tcap;
    const int n_items = pen_prefix(flatB, s_pref, s_scan, [&](int b_) { return pen_sel_on(sel, b_) && P.ptotal[b_] > 0 ? P.nrb[b_] + P.nlq[b_] : 0; });
    const int n_rankers = ((int)gridDim.x - PEN_RANK_HELPERS) * 4;
    if ((int)blockIdx.x < (int)gridDim.x - PEN_RANK_HELPERS)      // (the last workgroups start on the queues of overflowed lists at once)
    for (int c = blockIdx.x * 4 + wv; c < n_items; c += n_rankers) {
        const int b = pen_chunk_mesh(s_pref, flatB, c);
        const int r = c - s_pref[b], nb_ = P.nrb[b];
        const bool rewalk_b = pen_can_rewalk(P) && P.ovn[b * 2] > 0;
        if (r < nb_) { rank_block(b, P.rb[(size_t)b * P.n_clus + r] * 64, rewalk_b, true); continue; }
        // a long list is an item of its own (k_pen_list: P.lq): a collapsed mesh brings blocks of 64 of them, ~1.5 us each, and the
        // wavefront that held such a block was the launch's long pole (rank p50 28 us with the blocks dealt flat, p90 98)
        const int ff = P.lq[(size_t)b * F + (r - nb_)];
        const int cc = P.pcount[(size_t)b * F + ff], off = P.poff[(size_t)b * F + ff], found = P.pavail[(size_t)b * F + ff];
        if (off >= P.pair_cap || (found > P.pcap && rewalk_b)) continue;      // (incomplete list: queued by k_pen_list, taken below)
        const int* part = P.partners + (size_t)b * F * P.pcap;
        const int* mine_ = part + (size_t)ff * P.pcap;
        const int av_ = min(found, P.pcap);
        int x[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int v_ = mine_[min(lane + 64 * q, P.pcap - 1)]; x[q] = lane + 64 * q < av_ ? v_ : 0x7fffffff; }
        rank_one(part, P.pown + (size_t)b * P.pair_cap, P.plist + (size_t)b * P.pair_cap, ff, cc, off, found, x);
    }
    } else
    for (int si = blockIdx.y; si < nsel; si += gridDim.y) {
    const int b = si == (int)blockIdx.y ? b_first : pen_sel_col(sel, si);
    __syncthreads();
    if (t == 0) { s_u[0] = pen_sel_on(sel, b) ? 1 : 0; s_u[1] = P.ptotal[b]; s_u[2] = P.ovn[b * 2]; }
    __syncthreads();
    if (!s_u[0] || s_u[1] == 0) continue;
    const bool rewalk_b = pen_can_rewalk(P) && s_u[2] > 0;      // (k_pen_list empties the queue of a mesh it will not have looked at again)
    // 64 consecutive triangles at a time per wavefront.  Short lists: their elements are dealt to the
    // lanes (owner found by bisection of the 64 offsets in LDS), each lane ranks its element within its
    // list.  Long lists: bitonic sort by the whole wavefront in LDS.
    // (the last PEN_RANK_HELPERS workgroups of a mesh rank nothing: they start on the queue of overflowed lists at once, next to
    //  the ranking instead of behind it -- a triangle's second look at the grid takes one wavefront ~25 us)
    const bool helper = blockIdx.x >= PEN_RANK_BLOCKS;
    const int nw = PEN_RANK_BLOCKS * 4, gw = blockIdx.x * 4 + wv;
    // (blocks of 64 triangles dealt round-robin to the frame's wavefronts: crowded triangles are neighbours in the index too,
    //  a contiguous range per wavefront gave one wavefront all the long lists)
    const int flim = F;
    for (int fw = helper ? flim : gw * 64; fw < flim; fw += nw * 64) rank_block(b, fw, rewalk_b, false);
    // Triangles whose list overflowed while it was collected: one shared queue per mesh (k_pen_list), taken one triangle at a time
    // through an atomic cursor by whichever wavefront is free -- first the mesh's own, then those of the other meshes of the call
    // (such triangles come in crowds, in one or two meshes of a call: their own 256 wavefronts would be the launch's long pole).
    // Who derives a triangle's partners has no influence on what they are.
    }
#ifdef PEN_RANKT
    const long long rt1 = wall_clock64();
    auto rank_report = [&]() {
        const long long rt2 = wall_clock64();
        if (lane == 0 && P.work && rt2 - rt0 > 4000) {      // wavefronts that took more than 40 us
            atomicAdd(&P.work[8], 1ull); atomicAdd(&P.work[9], (unsigned long long)(rt1 - rt0)); atomicAdd(&P.work[10], (unsigned long long)rt_long);
            atomicAdd(&P.work[11], (unsigned long long)(rt2 - rt1)); atomicAdd(&P.work[12], (unsigned long long)n_long); atomicAdd(&P.work[13], (unsigned long long)n_rew);
            atomicAdd(&P.work[14], (unsigned long long)rt_ld); atomicAdd(&P.work[15], (unsigned long long)rt_short);
        }
    };
    __syncthreads();
    if (t == 0) s_u[3] = pen_can_rewalk(P) ? P.ovm[0] : 0;      // (meshes with a queue in this evaluation: k_pen_g1 -> 0, k_pen_list appends)
    __syncthreads();
    if (s_u[3] == 0) { rank_report(); return; }
#else
    __syncthreads();
    if (t == 0) s_u[3] = pen_can_rewalk(P) ? P.ovm[0] : 0;      // (meshes with a queue in this evaluation: k_pen_g1 -> 0, k_pen_list appends)
    __syncthreads();
    if (s_u[3] == 0) return;                                     // (no list of this evaluation overflowed: nothing queued anywhere)
#endif
    // (round 5: WHICH meshes have a queue is a compact list k_pen_list appends to -- a handful per evaluation.  Until then every
    //  wavefront of the launch looked through the counters of ALL the call's meshes: 12 k wavefronts x 119 meshes x 3 loads on the
    //  same few cache lines whenever any mesh had overflowed -- which the collapsed meshes of a fit make the normal case: 54 us of
    //  the slow wavefronts' 110, measured with -DPEN_RANKT)
    const int nB = min(s_u[3], P.F), b = nB > 0 ? (int)((blockIdx.x + blockIdx.y) % (unsigned)nB) : 0;      // (at most one entry per mesh of the evaluation)
    for (int g0 = 0; g0 < nB; g0 += 64) {
      const int bl = g0 + lane;
      const int bq = P.ovm[1 + (bl < nB ? (b + bl < nB ? b + bl : b + bl - nB) : 0)];           // (a rotation of the list: the takers spread over the queues)
      const int nql = P.ovn[bq * 2], ptl = P.ptotal[bq], wl = pen_sel_on(sel, bq) ? 1 : 0;
      unsigned long long todo = __ballot(bl < nB && wl && nql > 0 && ptl > 0 && P.ovn[bq * 2 + 1] < nql);
      while (todo) {
        const int bit = __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const int bb = __builtin_amdgcn_readlane(bq, bit);
        const int nq = __builtin_amdgcn_readlane(nql, bit);
        const int* pcb = P.pcount + (size_t)bb * F;
        const int* poffb = P.poff + (size_t)bb * F;
        int* pownb = P.pown + (size_t)bb * P.pair_cap;
        int* plistb = P.plist + (size_t)bb * P.pair_cap;
        for (;;) {
            int idx = 0;
            if (lane == 0) idx = atomicAdd(&P.ovn[bb * 2 + 1], 1);
            idx = __builtin_amdgcn_readfirstlane(idx);
            if (idx >= nq) break;
            const int ff = P.ovq[(size_t)bb * F + idx];
            const int cc = pcb[ff], off = poffb[ff];
            if (off >= P.pair_cap) continue;
            __builtin_amdgcn_wave_barrier();
#ifdef PEN_RANKT
            ++n_rew;
#endif
            const int got = pen_rewalk(P, bb, ff, tile, tcap, lane);
            const int keep = min(min(cc, got), P.pair_cap - off);
            for (int q = lane; q < keep; q += 64) { plistb[off + q] = tile[q]; pownb[off + q] = ff; }
            __builtin_amdgcn_wave_barrier();
        }
      }
    }
#ifdef PEN_RANKT
    rank_report();
#endif
}

