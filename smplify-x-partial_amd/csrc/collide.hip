// collide.hip -- the interpenetration term of SMPLify-X on a batch of posed meshes.
//
// Replaces, for B frames at once, what the reference obtains from the external CUDA package
// mesh_intersection (smplifyx/fitting.py:437-455, set-up smplifyx/fit_single_frame.py:300-328):
//     collision_idxs = BVH(max_collisions)(triangles)            broad phase: AABB overlap
//     collision_idxs = FilterFaces(segm, parents, ign_part_pairs)(collision_idxs)
//     pen_loss       = DistanceFieldPenetrationLoss(sigma, penalize_outside)(triangles, collision_idxs)
// and its gradient with respect to the vertices.  Algorithm and formulas: oracle/penetration.py
// (the package's source is absent: parity unpinned).
//
// MI355X design (not the package's LBVH; DESIGN.md 4.6 has the table):
//   k_pen_g1 / g2 / g3   triangle AABBs (8 x 1024 lanes per frame) -> bounding box per body part: a triangle
//                 whose box meets the box of no part it may collide with is dropped, the survivors are
//                 compacted (8 x 1024) -> they enter a uniform grid (cell = twice the mean triangle extent,
//                 every cell the AABB touches) hashed into 16384 LDS buckets by a counting sort (1 x 1024).
//   k_pen_walk    (128 x 256 lanes per frame) pair tests over blocks of 64 consecutive entries of the
//                 bucket-sorted list: each lane holds one entry (AABB, vertex ids, part, cell), the
//                 headers sit in a wavefront-private LDS window, and lane i walks the entries after it
//                 in its bucket, two per iteration, so memory is touched once per ENTRY, not per pair.
//                 Tests in order: same cell, part mask (one 64-bit word), AABB overlap, ownership by the
//                 cell of the intersection's low corner (a mask test on cell keys), shared vertices.
//                 Accepted pairs are queued per wavefront and appended to both triangles' partner lists
//                 64 at a time.
//   k_pen_list /  offsets and ranks turn the partner lists (appended in scheduling order) into the
//   k_pen_rank    frame's pair list -- triangles ascending, partners ascending -- which fixes every
//                 later summation order.
//   k_pen_eval    one lane per ORDERED pair of that list: conic distance field evaluated with
//                 forward-mode dual numbers -- the lane differentiates with respect to the OWNER's 9
//                 coordinates only, once as receiver geometry and once as intruding points -- so
//                 every number has one owner: no atomics in the arithmetic, results independent of
//                 scheduling and of batch composition, and the work is balanced over the chip however
//                 unevenly the collisions are spread over the triangles.
//   k_pen_facesum / k_pen_gather   per-triangle sums over the pair ranges; vertex gradient = fixed-order
//                 sum over the incident triangle corners (CSR); frame loss = triangles in index order; for a fitting batch
//                 the same lane writes d v_posed = T^T g, the operand of the adjoint GEMM (lbs_adjoint.hip).
// Ten launches per evaluation (twelve + two memsets until round 4: the accumulators of the grid build are left empty by
// k_pen_g3 for the next evaluation instead of by a kernel of their own, and the per-column "wanted" flags are kept by the
// fitting loop's tick kernel).
#include "../../include/sfx.h"
#include "sfx_internal.h"
#include "wave_ops.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define PEN_T 1024
#define PEN_GRID_INTS ((((PEN_CELLS + 1) + 3) / 4) * 4)
#define PEN_SPAN 8              // cells per axis one triangle may be entered in (a sane triangle spans 1-3; an exploded
                                // mesh -- diverged fit, NaN / huge coordinates -- must not turn into 10^9 cell visits)
#ifdef PEN_COUNT
#define PEN_STATS 32      // diagnostic build: room for the walk counters
#else
#define PEN_STATS 16
#endif
#define PEN_MAX_WALK 2048        // entries an entry looks ahead in its bucket before it gives up (crowded cells of a sane mesh hold hundreds:
                                // 418 on the synthetic surface).  A diverged fit folds the mesh into a few cells of 10^4 entries; walking
                                // them out took 7-8 ms per evaluation and held up the whole batch (3 % of the launches of a 256-frame fit,
                                // half of this kernel's total time).  The reference's BVH bounds a query by max_collisions hits instead.
#ifndef PEN_WALK_BLOCKS
#ifndef PEN_WALK_FLAT
#define PEN_WALK_FLAT 1024      // workgroups of the flat form of k_pen_walk (a block of 64 grid entries per wavefront)
#endif
#define PEN_WALK_BLOCKS 128     // workgroups (of 4 wavefronts) per frame of the pair tests: a frame whose limbs are pushed
                                // through each other has 100x the candidates of a clean one, and must not hold up the launch
#endif
#ifndef PEN_EVAL_BLOCKS
#define PEN_EVAL_BLOCKS 128     // workgroups per frame of the pair evaluation (grid-stride over the pair list)
#endif
#ifndef PEN_REWALK_MAX
#define PEN_REWALK_MAX 4096   // overflowed lists per mesh and evaluation that are derived from the grid again (more: the mesh has collapsed)
#endif
#define PEN_CELLS 16384         // hash buckets of the grid: one 64-KB LDS array serves as histogram, start offsets and
                                // scatter cursors (+ 48 KB of wavefront tiles, 16 KB of pair queues)

struct PenDev {
    int V, F, cap, n_parts;    // cap = max_collisions: partners KEPT per triangle
    int pcap;                  // partners HELD per triangle while the list is being collected (2 x cap)
    int p2p;                   // DistanceFieldPenetrationLoss(point2plane=True): Psi^2 weighted by (n_f . n_g)^2
    const int* faces;          // [F][3]
    const int* segm;           // [F]
    const unsigned char* skip; // [n_parts][n_parts] 1 = pair of parts never collides
    const int4* faces4;        // [F] the faces padded to 16-byte rows (one load per header in the pair tests)
    const unsigned long long* skipmask;   // [64] the same table as one 64-bit word per part (bit q: never collides with part q)
    const int* vf_start;       // [V+1] CSR: incident (face * 3 + corner)
    const int* vf_list;
    // per batch (capacity Bmax)
    float* aabb;               // [B][F][6]
    int2* entries;             // [B][ent_cap] (triangle | part << 24 | lz << 30, packed cell coordinates | lx << 30 | ly << 31), sorted by bucket;
                               //              lx, ly, lz: the cell holds the low corner of the triangle's box on that axis
    int2* cand;                // [B][ent_cap] (round 4) one record per (surviving triangle, cell its box touches), any order: the entry record k_pen_g3 sorts
    int4* tlist;               // [B][F] (unused since round 4: k_pen_g2 emits the candidate records itself)
    int* tcount;               // [B][16] (one cache line each) number of survivors (k_pen_g3 leaves 0 behind, k_pen_g2 reserves ranges)
    int* pbox;                 // [B][64][6] bounding box of every part, order-preserving ints (k_pen_g1; reset per evaluation)
    float* gpart;              // [B][PEN_GW][8] per workgroup of k_pen_g1: frame box lo / hi, extent sum
    int ent_cap;
    int* partners;             // [B][F][pcap]
    int* pavail;               // [B][F] partners FOUND (the list holds the first min(found, pcap) arrivals)
    int* pcount;               // [B][F]
    int* poff;                 // [B][F] start of the triangle's partner range in the frame's pair list
    unsigned* hasp;            // [B][hasp_words] bit f: triangle f has pairs in the list (k_pen_list -> k_pen_gather)
    int hasp_words;            // (F + 31) / 32
    int* pown;                 // [B][pair_cap] pair list: receiving triangle (ascending) ...
    int* plist;                // [B][pair_cap] ... and its partner (ascending within the triangle)
    int pair_cap;
    float* pout;               // [B][10][pair_cap] per ordered pair: gradient w.r.t. the owner's 9 coordinates, loss
    float* tgrad;              // [B][F][9] per triangle: sum over its pairs (valid where pcount > 0)
    float* tloss;              // [B][F]
    int* ptotal;               // [B] ordered pairs in the list
    int* cells;                // [B][PEN_CELLS + 1] bucket END offsets into entries ([PEN_CELLS] = number of entries)
    float* gridp;              // [B][4] low corner of the frame's box, 1 / cell size
    int* stats;                // [B][PEN_STATS]: pairs (ordered), dropped partners, overflow of entries, cells, phase clocks
    int2* wq;                  // [B][wq_cap] chunks of the pair tests beyond a block's first 64 steps: (first entry of the block, chunk k)
    int* wqn;                  // [B] chunks queued (k_pen_g3 -> 0, k_pen_walk appends, k_pen_walk2 consumes)
    int wq_cap;
    int* ovq;                  // [B][F] triangles whose partner list overflowed while it was collected (k_pen_list -> k_pen_rank: pen_rewalk)
    int* ovn;                  // [B][2] their number, and the cursor the wavefronts of k_pen_rank take them with
    int no_rewalk;             // SFX_PEN_REWALK_OFF=1 (A/B measurement switch): overflowing lists keep their first arrivals, as until round 4
    int* callno;               // [2] evaluations so far (k_pen_g1), and the last one in which some list overflowed (k_pen_list): k_pen_rank looks for queues only then
    int* ovm;                  // [1 + B] meshes of this evaluation whose overflow queue k_pen_rank has to drain: count (k_pen_g1 -> 0), ids (k_pen_list)
    int* over;                 // [B] or NULL (set per call): 1 = this evaluation of the mesh kept partners by ARRIVAL order somewhere (a list beyond
                               //     2 x max_collisions, a cut walk): its numbers are not reproducible run to run
    // round 5: the per-frame kernel (k_pen_frame) and the columns it hands to the general kernels
    int* heavy;                // [B] 1 = this evaluation of the column goes through the general kernels (crowded grid / too many pairs for one workgroup's LDS)
    int* hlist;                // [B] the heavy columns of this evaluation, any order
    int* nheavy;               // [1] their number (k_pen_g1 -> 0, k_pen_frame appends)
    float* wbox;               // [B][n_clus][6] boxes of the clusters of 64 consecutive triangles (k_pen_g1: one DPP reduction per wavefront)
    const unsigned long long* cpm;   // [n_clus] parts present in a cluster, one bit each (static)
    int n_clus;                // (F + 63) / 64
    int* rb;                   // [B][n_clus] the blocks of 64 consecutive triangles of a column that have partners, ascending (k_pen_list -> k_pen_rank)
    int* nrb;                  // [B] their number
    int* lq;                   // [B][F] the triangles of a column with a long list (more than PEN_SHORT partners, or cut), any order
    int* nlq;                  // [B] their number
    int* wl;                   // [B] the columns of this evaluation that carry the term (want != 0), ascending: k_pen_g1's first workgroup
    int* nw;                   // [1] their number      (-> the rows of the later launches loop over this list: no workgroup for a column nobody wants)
    int* pcnt;                 // [B] pairs the pair tests have accepted (k_pen_g3 -> 0; beyond pf_cap they are counted, not stored)
    int2* pbuf;                // [B][pf_cap] accepted pairs of a frame on the fast path, any order (the partner-list buffer: unused there)
    int pf_cap;                // min(PEN_FP, F * pcap / 2)
    int fast_ok;               // the mesh fits the per-frame kernel's LDS layout (F^2 < 2^32: 32-bit sort keys; bit sets and vertex list in 64 KB)
    unsigned long long* work;  // [6] process-wide counts since sfx_pen_work_reset: grid entries, ordered pairs, column evaluations, surviving triangles,
                               //     triangles with more partners than the lists hold (2 x max_collisions: arrival order decides there), bucket walks cut short
};

// Which columns a kernel of the general path works on.  hlist == NULL (the ten-kernel form, SFX / sfx_debug_pen_form 0): column
// blockIdx.y of the call, masked by `want`.  hlist != NULL (round 5): the compact list of columns k_pen_frame has handed over
// ("heavy": a crowded grid or more pairs than one workgroup's LDS sorts), *nheavy of them -- the grid's rows loop over the list,
// and a launch that finds it empty (nearly every one) ends after one load.
struct PenSel { const int* want; const int* hlist; const int* nheavy; const int* heavy; };
__device__ __forceinline__ int pen_sel_n(const PenSel& s, const int rows) { return s.hlist ? min(*s.nheavy, rows) : rows; }
__device__ __forceinline__ int pen_sel_col(const PenSel& s, const int i) { return s.hlist ? s.hlist[i] : i; }
__device__ __forceinline__ bool pen_sel_on(const PenSel& s, const int b) { return s.hlist ? (!s.heavy || s.heavy[b] != 0) : (!s.want || s.want[b] != 0); }
// the column of a row's FIRST trip, requested together with the list's length (entries beyond the length are stale but in bounds:
// the list has a slot per column and a launch has at most as many rows) -- one round trip instead of two at every kernel's entry
__device__ __forceinline__ int pen_sel_first(const PenSel& s, const int row) { return s.hlist ? s.hlist[row] : row; }

// ---------------------------------------------------------------------------------------------
// The cone field and its derivatives, written out in reverse mode (round 4; rounds 1-3 pushed forward-mode dual numbers with
// nine tangents through the same formulas: ~10 x the flops of the value, and the pair evaluation was ALU-bound whenever most
// columns of a launch carried the term: p90 155 us).  Notation of oracle/penetration.py.
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 operator+(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(const V3& a, const float s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ float vdot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 vcross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }

// circumscribed circle + unit normal of a triangle (oracle/penetration.py: _cone_geometry); the intermediates the reverse
// sweep needs are kept
struct ConeGeo { V3 a, b, axb, num, oc, o, n; float n2, aa, bb, inv, il, r; };
__device__ __forceinline__ ConeGeo cone_geometry(const V3& p0, const V3& p1, const V3& p2) {
    ConeGeo g;
    g.a = p1 - p0; g.b = p2 - p0;
    g.axb = vcross(g.a, g.b);
    g.n2 = vdot(g.axb, g.axb);
    g.aa = vdot(g.a, g.a); g.bb = vdot(g.b, g.b);
    g.num = vcross(g.axb, g.a) * g.bb + vcross(g.b, g.axb) * g.aa;
    g.inv = 1.f / (g.n2 * 2.f);
    g.oc = g.num * g.inv;
    g.o = p0 + g.oc;
    g.r = sqrtf(vdot(g.oc, g.oc));
    g.il = 1.f / sqrtf(g.n2);
    g.n = g.axb * g.il;
    return g;
}
// adjoint of cone_geometry: (d L / d o, d L / d r, d L / d n) -> d L / d (p0, p1, p2)
__device__ __forceinline__ void cone_geometry_adj(const ConeGeo& g, const V3& go, const float gr, const V3& gn, V3& gp0, V3& gp1, V3& gp2) {
    // r = |oc| (sqrt at 0: zero slope, as the forward-mode version had it); o = p0 + oc
    const V3 goc = go + g.oc * (g.r > 0.f ? gr / g.r : 0.f);
    // n = axb il, il = n2^(-1/2)
    V3 gaxb = gn * g.il;
    float gn2 = vdot(g.axb, gn) * (-0.5f * g.il / g.n2);
    // oc = num inv, inv = 1 / (2 n2)
    const V3 gnum = goc * g.inv;
    gn2 += -vdot(g.num, goc) * g.inv * g.inv * 2.f;
    gaxb = gaxb + g.axb * (2.f * gn2);
    // num = (axb x a) bb + (b x axb) aa
    const V3 u1 = vcross(g.axb, g.a), u2 = vcross(g.b, g.axb);
    const V3 gu1 = gnum * g.bb, gu2 = gnum * g.aa;
    const float gbb = vdot(u1, gnum), gaa = vdot(u2, gnum);
    gaxb = gaxb + vcross(g.a, gu1) + vcross(gu2, g.b);         // u = x x y: dx = y x du, dy = du x x
    V3 ga = vcross(gu1, g.axb) + g.a * (2.f * gaa);
    V3 gb = vcross(g.axb, gu2) + g.b * (2.f * gbb);
    // axb = a x b
    ga = ga + vcross(g.b, gaxb);
    gb = gb + vcross(gaxb, g.a);
    gp1 = ga; gp2 = gb; gp0 = go - ga - gb;
}
// Psi(v)^2 of the cone field (o, r, n) at the point v (oracle/penetration.py: _psi, squared) and its derivatives with respect
// to d = v - o (= d / d v = - d / d o), n and r
__device__ __forceinline__ float cone_penalty(const V3& o, const float r, const V3& n, const V3& v, const float sigma,
                                              const int penalize_outside, V3& gd, V3& gn, float& gr) {
    gd = {0.f, 0.f, 0.f}; gn = {0.f, 0.f, 0.f}; gr = 0.f;
    const V3 d = v - o;
    const float x = vdot(d, n);
    if (!(x < sigma) || (!penalize_outside && x > 0.f)) return 0.f;
    const V3 q = d - n * x;
    const float rho = sqrtf(vdot(q, q));
    const float s = r * (1.f / sigma);
    const float den = r - s * x;
    const float phi = rho / den;
    if (!(phi < 1.f)) return 0.f;
    float ups, dups;
    if (x <= -sigma) { ups = (x * -1.f) + (1.f - sigma); dups = -1.f; }
    else {
        const float c2 = -(1.f - 2.f * sigma) / (4.f * sigma * sigma), c1 = -1.f / (2.f * sigma);
        ups = (x * x) * c2 + x * c1 + ((3.f - 2.f * sigma) / 4.f); dups = 2.f * c2 * x + c1;
    }
    const float w = (1.f - phi) * ups;
    const float psi = w * w;
    // pen = w^4
    const float gw = 4.f * w * psi;
    const float gphi = -ups * gw, gups = (1.f - phi) * gw;
    const float grho = gphi / den, gden = -gphi * phi / den;           // phi = rho / den
    gr = gden * (1.f - x * (1.f / sigma));                              // den = r - (r / sigma) x
    float gx = gups * dups - gden * s;
    const V3 gq = q * (rho > 0.f ? grho / rho : 0.f);                   // rho = |q|
    gx -= vdot(n, gq);                                                  // q = d - n x
    gd = gq + n * gx;                                                   // x = d . n
    gn = gq * (-x) + d * gx;
    return psi * psi;
}

// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float block_min(float v, float* red) {
    v = -wave_max_dpp(-v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < PEN_T / 64; ++i) r = fminf(r, red[i]);
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) { return -block_min(-v, red); }
__device__ __forceinline__ float block_sum_fixed(float v, float* red) {
    v = wave_sum_dpp(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < PEN_T / 64; ++i) r += red[i];
    __syncthreads();
    return r;
}

// exclusive prefix sum over the PEN_T lanes of the block (fixed order); *total = sum of all
__device__ __forceinline__ int block_excl_scan(const int v, int* wsum /* [PEN_T / 64] */, int* total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int inc = wave_incl_scan_dpp(v);
    __syncthreads();
    if (lane == 63) wsum[wv] = inc;
    __syncthreads();
    int base = 0, tot = 0;
    for (int i = 0; i < PEN_T / 64; ++i) { const int x = wsum[i]; if (i < wv) base += x; tot += x; }
    *total = tot;
    return base + inc - v;
}

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

// zero_dverts / zero_G (round 5, with k_pen_frame): the per-frame kernel writes the gradient of the vertices that HAVE one (a few
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
            // is kept as well (round 5): k_pen_frame culls whole clusters against the part boxes before it looks at a triangle.
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
        // loads and out again, each holding a wavefront slot
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
    if (sel.hlist && *sel.nheavy == 0) return;          // (only heavy columns queue chunks: k_pen_frame leaves the others' queues empty)
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
__device__ __forceinline__ bool pen_can_rewalk(const PenDev& P) { const int t = pen_rank_tile(P.pcap); return t > 0 && P.cap + 64 <= t && !P.no_rewalk; }

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
    // a busy one was waiting for (the step is bound by slots x round trips)
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
// Shared by k_pen_eval and k_pen_frame; this file is compiled with -ffp-contract=off, so the two instances perform the same
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
// the sums of one triangle's pair range [i, i + n) from the run sums k_pen_eval / k_pen_frame left per 64-pair chunk of the list
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
// g(v) of one vertex and what follows from it (shared by k_pen_gather, every vertex, and k_pen_frame, the vertices of triangles
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
// Round 5: the whole term of ONE column behind the triangle boxes in one workgroup.
//
// On the mesh the reference evaluates (the SMPL-X topology with smplx_parts_segm.pkl; tests/golden/smplx_topology.npz) the part
// boxes turn away 88 % of the triangles before the grid, ~2 500 survivors make ~6 200 grid entries and ~1 700 ordered pairs per
// evaluation: two orders of magnitude below what the ten general kernels above are dimensioned for, each of which paid a launch,
// a pass over all F triangles or V vertices, and its own round trips (k_pen_g2 41 us, k_pen_g3 25, k_pen_walk 29 + 20,
// k_pen_list 17, k_pen_rank 60, k_pen_eval 12, k_pen_facesum 8, k_pen_gather 28 = 240 us per round of the halpe cfg's fit).
// k_pen_frame does the same steps for a column with the column's data in LDS:
//   A  part culling + one (triangle, cell) record per cell of a survivor's box     (k_pen_g2: coalesced pass over the F boxes)
//   B  counting sort of the records into the hashed grid                           (k_pen_g3, unchanged: 2 x 64 KB of LDS)
//   C  pair tests, a block of 64 entries per wavefront, 16 wavefronts              (k_pen_walk's chunk walk; bucket ends from LDS)
//      accepted pairs -> one list of the frame (global scratch; the per-triangle partner lists are not used)
//   D  both orders of every pair as 32-bit keys f * F + g, bitonic sort in LDS, rank within a triangle's run: the max_collisions
//      LOWEST partners are kept -> the frame's pair list, triangles ascending, partners ascending   (k_pen_list + k_pen_rank)
//   E  pair evaluation per 64-aligned chunk of that list (pen_pair_eval, pen_run_sums: the general kernels' functions)
//   F  per-triangle sums (pen_face_sum)
//   G  gradient of the vertices of triangles that have pairs (pen_vertex_out) -- the other rows were zeroed by k_pen_g1 --,
//      d v_posed = T^T g, the frame's loss (pen_frame_loss_partial)
// Every number is formed by the same fp32 operations in the same order as in the ten-kernel form (sfx_debug_pen_form(0)): the
// pair list is canonical, the sums are defined on it; tests/test_gpu_topology.py, tests/test_gpu_penetration.py compare bit for bit.
// A column that does not fit -- more than PEN_FE grid entries, a bucket beyond PEN_FB entries (a limb pushed through another by a
// trial step of the line search), more than PEN_FP pairs -- is handed to the general kernels ("heavy": P.heavy / P.hlist), which
// run on the compact list of such columns and end after one load when it is empty.
#define PEN_FE 16384            // grid entries of a column on the fast path
#define PEN_FB 256              // longest bucket on the fast path (chunks 0..3 of a block's walk)
#define PEN_FP 8192             // unordered pairs on the fast path: 2 x PEN_FP sort keys = 64 KB of LDS
#define PEN_FW 16               // wavefronts of the workgroup
#ifndef PEN_AU
#define PEN_AU 1               // (4 faulted with a memory access error on the device -- not understood; 2 ran and changed nothing)
#endif
#define PEN_HEAVY_ROWS 8        // grid rows of the general kernels when they work on the handed-over columns (they loop over the list)
#define PEN_FRAME_LDS ((PEN_GRID_INTS + PEN_CELLS + PEN_FW * 256) * 4)        // cells | part masks -> windows -> sort keys | pair queues

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
// everything between the pair buffer and the outputs in LDS.  Shared by k_pen_narrow (round 5's default form) and k_pen_frame.
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
// needs 70-150 us for a column's, measured in k_pen_frame), the accepted pairs land in one list per column, and ONE workgroup per
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
    int n_mark = 3;                             // (stats[7..10]: the stamps of phases D-G, as in k_pen_frame)
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

template <bool P2P>
__global__ __launch_bounds__(PEN_T)
void k_pen_frame(PenDev P, const float* __restrict__ verts, const float sigma, const int penalize_outside, float* __restrict__ dverts,
                 float* __restrict__ loss_out, const int* __restrict__ want, PenAdjPrep ap, const int force_heavy) {
    extern __shared__ int lds[];
    int* cell_cnt = lds;                                            // [PEN_GRID_INTS] histogram -> start offsets -> cursors -> bucket ENDS
    unsigned* pmask = reinterpret_cast<unsigned*>(lds + PEN_GRID_INTS);      // [PEN_CELLS] parts present per bucket (phase B)
    int* r1 = lds + PEN_GRID_INTS;                                  // the same 64 KB: windows (C), sort keys / pair list (D-F), scratch (G)
    int* s_queue = lds + PEN_GRID_INTS + PEN_CELLS;                 // [PEN_FW][256] pair queues of the wavefronts (C)
    __shared__ unsigned long long s_mask[64], s_near[64];
    __shared__ int s_pbox[64][6];
    __shared__ unsigned s_coll32[128];
    __shared__ int slice[PEN_T];
    __shared__ float red[PEN_T / 64];
    __shared__ int s_cnt, s_ccnt, s_total, s_maxb, s_npairs, s_dead;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, wv = t >> 6;
    const int F = P.F;
    int* st = P.stats + b * PEN_STATS;
    int* cells = P.cells + (size_t)b * (PEN_CELLS + 1);
    // wall-clock stamps (100 MHz) at the ends of the phases -> stats[4..11] (sfx_pen_phase_clocks: ticks since the kernel's start)
    const long long t_start = wall_clock64();
    int n_mark = 0;
    auto mark = [&]() { if (t == 0) st[4 + n_mark] = (int)(wall_clock64() - t_start); ++n_mark; };
#ifdef PEN_ASTAMP
    int n_sub = 0;
#define ASUB() do { if (t == 0) st[5 + n_sub] = (int)(wall_clock64() - t_start); ++n_sub; } while (0)
#else
#define ASUB() do { } while (0)
#endif
    // Everything this column needs from global memory before the culling, fetched by different lanes in ONE round trip (a dependent
    // global access costs this workgroup 1-2 us -- its inputs were written by another kernel, on other XCDs' L2s -- and the first
    // version of this prologue strung ten of them together: 20 us): part boxes, the static part table, the frame box partials.
    __shared__ float s_gpart[PEN_GW * 8];
    __shared__ unsigned s_near32[128];
    const int wanted = want ? want[b] : 1;
    {
        int pb_v = 0; unsigned long long sk_v = 0ull; float gp_v = 0.f;
        if (t < 64 * 6) pb_v = P.pbox[(size_t)b * 64 * 6 + t];
        else if (t < 64 * 6 + 64) sk_v = P.skipmask[t - 64 * 6];
        else if (t < 64 * 6 + 64 + PEN_GW * 8) gp_v = P.gpart[(size_t)b * PEN_GW * 8 + (t - 64 * 6 - 64)];
        if (t == 0) { P.wqn[b] = 0; P.heavy[b] = 0; }
        if (!wanted) {                          // the frame's stage carries no collision weight: nothing to do
            if (t == 0) { P.ptotal[b] = 0; cells[PEN_CELLS] = 0; st[0] = st[1] = st[2] = st[3] = 0; st[13] = 0; st[15] = 0; loss_out[b] = 0.f;
                          if (P.over) P.over[b] = 0; }
            return;
        }
        if (t < 64 * 6) (&s_pbox[0][0])[t] = pb_v;
        else if (t < 64 * 6 + 64) {
            const int q = t - 64 * 6;
            s_mask[q] = sk_v;
            const unsigned long long c = q < P.n_parts ? ~sk_v & (P.n_parts >= 64 ? ~0ull : (1ull << P.n_parts) - 1ull) : 0ull;
            s_coll32[q] = (unsigned)c; s_coll32[64 + q] = (unsigned)(c >> 32);
        } else if (t < 64 * 6 + 64 + PEN_GW * 8) s_gpart[t - 64 * 6 - 64] = gp_v;
        if (t < 128) s_near32[t] = 0u;
        if (t == 0) { s_cnt = 0; s_ccnt = 0; s_npairs = 0; s_dead = 0; }
        for (int c = t; c <= PEN_CELLS; c += PEN_T) cell_cnt[c] = 0;
        for (int c = t; c < PEN_CELLS; c += PEN_T) pmask[c] = 0u;
    }
    __syncthreads();
    ASUB();     // [5] inputs staged
    // the part boxes are read: leave them empty for the next evaluation of this column (k_pen_g1 accumulates into them)
    if (t < 64 * 6) P.pbox[(size_t)b * 64 * 6 + t] = (t % 6) < 3 ? 0x7fffffff : (int)0x80000000;
    const float* aabb = P.aabb + (size_t)b * F * 6;
    int2* cand = P.cand + (size_t)b * P.ent_cap;
    PenGridCtx C;                               // (pen_grid_ctx on the staged partials: same operations, same order)
    {
        float lo3[3] = {3e38f, 3e38f, 3e38f}, ext = 0.f;
        for (int w = 0; w < PEN_GW; ++w) {
            const float* g = s_gpart + w * 8;
            for (int e = 0; e < 3; ++e) lo3[e] = fminf(lo3[e], g[e]);
            ext += g[6];
        }
        const float h = fmaxf(2.f * (ext / (float)P.F), 1e-6f);
        for (int e = 0; e < 3; ++e) C.glo[e] = lo3[e];
        C.ih = 1.f / h;
    }
    if (t == 0) { float* gp = P.gridp + b * 4; gp[0] = C.glo[0]; gp[1] = C.glo[1]; gp[2] = C.glo[2]; gp[3] = C.ih; }
    // ---------------------------------------------------------------- A: part culling, candidate records (k_pen_g2)
    {   // parts whose boxes meet and that may collide, one 64-bit word per part: the 64 x 64 tests dealt over the lanes (16 lanes per part)
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
    ASUB();     // [6] near masks
    // Two levels (round 5): first the CLUSTERS of 64 consecutive triangles -- their boxes are one DPP reduction per wavefront in
    // k_pen_g1, their part sets are static -- against the boxes of the parts they may collide with, one cluster per lane; then the
    // triangles of the clusters that are left (a fifth of them on a body), a cluster per wavefront.  A lane per triangle through all
    // F boxes, each walking its part's list of near parts on LDS round trips, took 58 us of this kernel's 210.
    int* clist = reinterpret_cast<int*>(pmask);             // (the part masks are filled in phase B; the list is consumed before)
    {
        int n_surv = 0;
        for (int c0 = 0; c0 < P.n_clus; c0 += PEN_T) {      // (one trip for meshes of up to 65 k triangles)
            const int c = c0 + t;
            bool any = false;
            if (c < P.n_clus) {
                unsigned long long pm = P.cpm[c], nm = 0ull;
                while (pm) { const int q = __ffsll((long long)pm) - 1; pm &= pm - 1; nm |= s_near[q]; }
                if (nm) {
                    const float* wb = P.wbox + ((size_t)b * P.n_clus + c) * 6;
                    int a6[6];
#pragma unroll
                    for (int e = 0; e < 6; ++e) a6[e] = pen_ford(wb[e]);
                    while (nm) {      // (two parts per trip, no early exit inside a trip: independent LDS reads)
                        const int q0 = __ffsll((long long)nm) - 1; nm &= nm - 1;
                        const int q1 = nm ? __ffsll((long long)nm) - 1 : q0; nm &= nm - 1;
                        const int* pa = s_pbox[q0]; const int* pb = s_pbox[q1];
                        const bool m0 = (a6[0] <= pa[3]) & (pa[0] <= a6[3]) & (a6[1] <= pa[4]) & (pa[1] <= a6[4]) & (a6[2] <= pa[5]) & (pa[2] <= a6[5]);
                        const bool m1 = (a6[0] <= pb[3]) & (pb[0] <= a6[3]) & (a6[1] <= pb[4]) & (pb[1] <= a6[4]) & (a6[2] <= pb[5]) & (pb[2] <= a6[5]);
                        if (m0 | m1) { any = true; break; }
                    }
                }
            }
            int tot;
            const int pos = n_surv + block_excl_scan(any ? 1 : 0, slice, &tot);
            if (any) clist[pos] = c;
            n_surv += tot;
        }
        __syncthreads();
        ASUB();     // [7] clusters culled
        if (t == 0) st[12] = n_surv;
        for (int cg = wv; cg < n_surv; cg += PEN_FW * PEN_AU) {      // PEN_AU clusters of the wavefront per trip: their loads go out together
          int f4[PEN_AU], seg4[PEN_AU]; float bx4[PEN_AU][6];
#pragma unroll
          for (int u = 0; u < PEN_AU; ++u) {
              const int ci = cg + u * PEN_FW;
              const int fr = clist[min(ci, n_surv - 1)] * 64 + lane;
              f4[u] = ci < n_surv ? fr : F;
              const unsigned ff = (unsigned)min(fr, F - 1);
              seg4[u] = P.segm[ff];
              const float* bp = aabb + (size_t)ff * 6;
#pragma unroll
              for (int e = 0; e < 6; ++e) bx4[u][e] = bp[e];
          }
#pragma unroll
          for (int u = 0; u < PEN_AU; ++u) {
            if (cg + u * PEN_FW >= n_surv) break;
            const int f = f4[u], seg = seg4[u];
            float bx[6];
#pragma unroll
            for (int e = 0; e < 6; ++e) bx[e] = bx4[u][e];
            bool any = false;
            unsigned long long nm = f < F ? s_near[seg] : 0ull;
            if (nm) {
                int a6[6];
#pragma unroll
                for (int e = 0; e < 6; ++e) a6[e] = pen_ford(bx[e]);
                while (nm) {
                    const int q0 = __ffsll((long long)nm) - 1; nm &= nm - 1;
                    const int q1 = nm ? __ffsll((long long)nm) - 1 : q0; nm &= nm - 1;
                    const int* pa = s_pbox[q0]; const int* pb = s_pbox[q1];
                    const bool m0 = (a6[0] <= pa[3]) & (pa[0] <= a6[3]) & (a6[1] <= pa[4]) & (pa[1] <= a6[4]) & (a6[2] <= pa[5]) & (pa[2] <= a6[5]);
                    const bool m1 = (a6[0] <= pb[3]) & (pb[0] <= a6[3]) & (a6[1] <= pb[4]) & (pb[1] <= a6[4]) & (a6[2] <= pb[5]) & (pb[2] <= a6[5]);
                    if (m0 | m1) { any = true; break; }
                }
            }
            int2 pk = make_int2(0, 0);
            if (any) {
                int c0[3], sp[3];
#pragma unroll
                for (int e = 0; e < 3; ++e) { c0[e] = pen_cell_of(C, bx[e], e); sp[e] = min(pen_cell_of(C, bx[3 + e], e), c0[e] + PEN_SPAN - 1) - c0[e]; }
                pk.x = (c0[0] & 1023) | ((c0[1] & 1023) << 10) | ((c0[2] & 1023) << 20) | (int)0x80000000;
                pk.y = sp[0] | (sp[1] << 3) | (sp[2] << 6) | (seg << 9);
            }
            const unsigned long long m = __ballot(any);
            const int nc = any ? ((pk.y & 7) + 1) * (((pk.y >> 3) & 7) + 1) * (((pk.y >> 6) & 7) + 1) : 0;
            const int inc = wave_incl_scan_dpp(nc);
            const int wtot = __builtin_amdgcn_readlane(inc, 63);
            int wo = 0;
            if (lane == 0 && m) { atomicAdd(&s_cnt, __popcll(m)); wo = atomicAdd(&s_ccnt, wtot); }
            int pos = __builtin_amdgcn_readfirstlane(wo) + inc - nc;
            if (pk.x < 0) {
                const int pf = (pk.y >> 9) & 63;
                pen_for_cells(pk, [&](int, int key, int lowz) {
                    if (pos < P.ent_cap) cand[pos] = make_int2(f | (pf << 24) | (lowz << 30), key);
                    ++pos;
                });
            }
        }
        }
        ASUB();     // [8] wavefront 0 through its clusters
        __syncthreads();
        ASUB();     // [9] all wavefronts
        for (int c = t; c < P.n_clus && c < PEN_CELLS; c += PEN_T) pmask[c] = 0u;       // (the cluster list lay in the part masks' array)
    }
    __threadfence_block();
    __syncthreads();
    const int NT = min(s_cnt, F), NC_raw = s_ccnt, NC = min(NC_raw, P.ent_cap);
    mark();                                     // [4] A: part culling
#ifdef PEN_ASTAMP
    if (t == 0) { st[10] = NC_raw; st[11] = NT; P.ptotal[b] = 0; loss_out[b] = 0.f; st[0] = st[1] = st[2] = 0; }
    return;
#endif
    // ---------------------------------------------------------------- B: counting sort into the hashed grid (k_pen_g3)
    constexpr int U2 = 8;
    auto cell_bucket = [](const int key) { return pen_bucket(key & 1023, (key >> 10) & 1023, (key >> 20) & 1023); };
    pen_cell_filter<U2>(P, cand, NC, pmask, s_coll32);
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
    {
        constexpr int per = PEN_CELLS / PEN_T;
        int* row0 = cell_cnt + wv * (64 * per) + lane;
        int ex[per], carry = 0, mx = 0;
#pragma unroll
        for (int i = 0; i < per; ++i) {
            const int v = row0[i * 64];
            mx = max(mx, v);
            const int inc = wave_incl_scan_dpp(v);
            ex[i] = carry + inc - v;
            carry += __builtin_amdgcn_readlane(inc, 63);
        }
        mx = (int)wave_max_dpp((float)mx);              // (counts < 2^24: exact)
        if (lane == 0) { slice[wv] = carry; slice[PEN_FW + wv] = mx; }
        __syncthreads();
        int base = 0, tot = 0, mb = 0;
        for (int i = 0; i < PEN_T / 64; ++i) { const int x = slice[i]; if (i < wv) base += x; tot += x; mb = max(mb, slice[PEN_FW + i]); }
#pragma unroll
        for (int i = 0; i < per; ++i) row0[i * 64] = base + ex[i];
        if (t == 0) { cell_cnt[PEN_CELLS] = tot; s_total = tot; s_maxb = mb; }
        __syncthreads();
    }
    int2* ent = P.entries + (size_t)b * P.ent_cap;
    const int n_ent = s_total;
    const bool ent_ok = n_ent <= P.ent_cap - 4 && NC_raw <= P.ent_cap;
    if (t == 0) { st[2] = ent_ok ? 0 : max(n_ent, NC_raw); st[3] = PEN_CELLS; st[13] = 0; st[14] = n_ent; st[15] = 0; for (int q = 5; q < 11; ++q) st[q] = 0;
                  for (int q = 16; q < PEN_STATS; ++q) st[q] = 0;
                  if (P.work) { atomicAdd(&P.work[0], (unsigned long long)n_ent); atomicAdd(&P.work[2], 1ull); atomicAdd(&P.work[3], (unsigned long long)NT); } }
    if (!ent_ok) {       // grid too crowded for the entry buffer: report, produce no pairs (the gradient rows are zero already)
        if (t == 0) { st[0] = 0; st[1] = 0; P.ptotal[b] = 0; cells[PEN_CELLS] = 0; loss_out[b] = 0.f; if (P.over) P.over[b] = 0; }
        return;
    }
    for (int i0 = t; i0 < NC; i0 += PEN_T * U2) {
        int2 r[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) { const int i = i0 + u * PEN_T; r[u] = cand[i < NC ? i : 0]; }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const int bk = cell_bucket(r[u].y);
            if (i0 + u * PEN_T < NC && pen_cell_keep(P, r[u], pmask, s_coll32, bk)) ent[atomicAdd(&cell_cnt[bk], 1)] = make_int2(r[u].x & 0x7fffffff, r[u].y);
        }
    }
    __threadfence_block();
    __syncthreads();
    mark();                                     // [5] B: grid
    // the general kernels take over from the sorted grid
    auto hand_over = [&]() {
        for (int c = t; c <= PEN_CELLS; c += PEN_T) cells[c] = cell_cnt[c];
        if (t == 0) { P.heavy[b] = 1; P.hlist[atomicAdd(P.nheavy, 1)] = b; }
    };
    if (force_heavy || !P.fast_ok || n_ent > PEN_FE || s_maxb > PEN_FB) { hand_over(); return; }
    // ---------------------------------------------------------------- C: pair tests
    // The bucket-sorted entries go through LDS in TILES of PEN_TW headers (entry record + box: 32 bytes), loaded by all 1024 lanes
    // at once -- two round trips per tile for the whole workgroup -- and every wavefront then walks blocks of 64 entries of the tile
    // on LDS alone: lane i of a block looks at the entries behind it up to the end of its bucket (<= PEN_FB on this path, the halo
    // of the tile), two candidates per step, the tests of k_pen_walk (same cell, part mask, boxes, ownership by the cell of the
    // intersection's low corner; shared vertices when the queue of accepted pairs is flushed).  A wavefront that fetched its own
    // block and window (k_pen_walk's scheme) spent ~8 us per block waiting for three dependent round trips: 75-150 us per column.
    {
        constexpr int PEN_TW = 2048, PEN_TOWN = PEN_TW - PEN_FB;
        int4* tA = reinterpret_cast<int4*>(r1); int4* tB = tA + PEN_TW;
        int* queue = s_queue + wv * 256; int qn = 0;
        int2* pbuf = P.pbuf + (size_t)b * P.pf_cap;
        auto flush = [&]() {
            const int n = qn;
            if (!n) return;
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier();
            for (int q0 = 0; q0 < n; q0 += 64) {
                const int q = q0 + lane;
                bool keep = false; int fa = 0, fb_ = 0;
                if (q < n) {
                    fa = queue[2 * q]; fb_ = queue[2 * q + 1];
                    const int4 va = P.faces4[fa], vb = P.faces4[fb_];      // triangles that share a vertex do not collide
                    keep = !(va.x == vb.x || va.x == vb.y || va.x == vb.z || va.y == vb.x || va.y == vb.y || va.y == vb.z ||
                             va.z == vb.x || va.z == vb.y || va.z == vb.z);
                }
                const unsigned long long m = __ballot(keep);
                if (!m) continue;
                int base = 0;
                if (lane == 0) base = atomicAdd(&s_npairs, __popcll(m));
                base = __builtin_amdgcn_readfirstlane(base);
                const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
                if (keep && pos < P.pf_cap) pbuf[pos] = make_int2(fa, fb_);
            }
            __builtin_amdgcn_wave_barrier();
            qn = 0;
        };
        for (int tile0 = 0; tile0 < n_ent; tile0 += PEN_TOWN) {
            __syncthreads();                                        // (the previous tile's walks are done)
            for (int idx = t; idx < PEN_TW; idx += PEN_T) {
                int hd[8];
                pen_load_hdr(ent, aabb, tile0 + idx, tile0 + idx < n_ent, hd);
                tA[idx] = make_int4(hd[0], hd[1], hd[2], hd[3]); tB[idx] = make_int4(hd[4], hd[5], hd[6], hd[7]);
            }
            __syncthreads();
            if (t == 0 && tile0 == 0) st[11] = (int)(wall_clock64() - t_start);   // (diagnostic: the first tile is in LDS)
            const int nown = min(PEN_TOWN, n_ent - tile0);
            for (int blk = wv; blk * 64 < nown; blk += PEN_FW) {
                const int idx0 = blk * 64 + lane, qi = tile0 + idx0;
                const bool vi = idx0 < nown;
                const int4 o0 = tA[idx0], o1 = tB[idx0];
                const int fi = o0.x & 0xffffff, ck = o0.y & 0x3fffffff;
                const unsigned long long skip_i = vi ? s_mask[(o0.x >> 24) & 63] : ~0ull;
                const float ai0 = __int_as_float(o0.z), ai1 = __int_as_float(o0.w), ai2 = __int_as_float(o1.x);
                const float ai3 = __int_as_float(o1.y), ai4 = __int_as_float(o1.z), ai5 = __int_as_float(o1.w);
                const int bend = vi ? cell_cnt[pen_bucket_of(ck)] : 0;
                const unsigned lowb = ((unsigned)o0.y >> 30) | (((unsigned)o0.x >> 28) & 4u);
                const unsigned need = ~lowb & 7u;
                const int dmax = __builtin_amdgcn_readfirstlane((int)wave_max_dpp((float)max(bend - 1 - qi, 0)));      // <= PEN_FB - 1
                auto test = [&](const bool act, const int4 h0, const int4 h1) {
                    const float kl0 = __int_as_float(h0.z), kl1 = __int_as_float(h0.w), kl2 = __int_as_float(h1.x);
                    const float kh0 = __int_as_float(h1.y), kh1 = __int_as_float(h1.z), kh2 = __int_as_float(h1.w);
                    const bool same = ((h0.y ^ ck) & 0x3fffffff) == 0;
                    const bool coll = ((unsigned)(skip_i >> ((h0.x >> 24) & 63)) & 1u) == 0u;
                    const bool box = (ai0 <= kh0) & (kl0 <= ai3) & (ai1 <= kh1) & (kl1 <= ai4) & (ai2 <= kh2) & (kl2 <= ai5);
                    const unsigned klow = ((unsigned)h0.y >> 30) | (((unsigned)h0.x >> 28) & 4u);
                    const bool own = (need & ~klow) == 0u;
                    return act & same & coll & box & own;
                };
                auto push = [&](const bool pass, const int other) {
                    const unsigned long long m = __ballot(pass);
                    if (m) {
                        const int pos = qn + __popcll(m & ((1ull << lane) - 1ull));
                        if (pass) { queue[2 * pos] = fi; queue[2 * pos + 1] = other & 0xffffff; }
                        qn += __popcll(m);
                        if (qn >= 64) flush();
                    }
                };
                for (int d = 1; d <= dmax; d += 2) {
                    const int k0 = min(idx0 + d, PEN_TW - 1), k1 = min(idx0 + d + 1, PEN_TW - 1);
                    const int4 hA0 = tA[k0], hB0 = tB[k0], hA1 = tA[k1], hB1 = tB[k1];
                    const bool p0 = test(qi + d < bend, hA0, hB0), p1 = test(qi + d + 1 < bend, hA1, hB1);
                    push(p0, hA0.x); push(p1, hA1.x);
                }
            }
        }
        if (t == 0) st[12] = (int)(wall_clock64() - t_start);   // (diagnostic: wavefront 0 is through its blocks)
        flush();
    }
    __threadfence_block();
    __syncthreads();
    const int npairs = s_npairs;
    mark();                                     // [6] C: pair tests
    if (npairs > P.pf_cap) { hand_over(); return; }
    pen_narrow<P2P>(P, b, npairs, PenNarrowLds{reinterpret_cast<unsigned*>(r1), reinterpret_cast<unsigned*>(cell_cnt), slice, red, &s_dead},
                    verts, sigma, penalize_outside, dverts, loss_out, ap, mark);
}

// ---------------------------------------------------------------------------------------------
// Work actually done by the term since the last reset, counted on the device (one atomic per column evaluation): what the
// benchmark's byte model of the step is computed from (bench.py roofline_pen) instead of "typical" constants.
static unsigned long long* g_pen_work = nullptr;      // device [4], process-wide (shared by every handle)
static unsigned long long* pen_work_buffer() {
    if (!g_pen_work) {
        if (hipMalloc((void**)&g_pen_work, 16 * sizeof(unsigned long long)) != hipSuccess) { g_pen_work = nullptr; return nullptr; }
        hipMemset(g_pen_work, 0, 16 * sizeof(unsigned long long));
    }
    return g_pen_work;
}
extern "C" int sfx_pen_work_reset(void) {
    if (!pen_work_buffer()) { sfx_set_error("out of device memory"); return -2; }
    if (hipDeviceSynchronize() != hipSuccess || hipMemset(g_pen_work, 0, 16 * sizeof(unsigned long long)) != hipSuccess) { sfx_set_error("device error"); return -4; }
    return 0;
}
// debug: wall-clock ticks (100 MHz) k_pen_narrow's workgroups spent in their phases since sfx_pen_work_reset, summed over the column
// evaluations that went through them: [0] until the pairs are read (entry), [1] D pair list, [2] E pair evaluation, [3] F triangle
// sums, [4] G vertices and loss, [5] number of such evaluations, [6] their ordered pairs, [7] unused
extern "C" int sfx_debug_pen_phase_ticks(int64_t* out /* [8] */) {
    if (!out) { sfx_set_error("null argument"); return -1; }
    for (int i = 0; i < 8; ++i) out[i] = 0;
    if (!g_pen_work) return 0;
    if (hipDeviceSynchronize() != hipSuccess) { sfx_set_error("device error"); return -4; }
    unsigned long long h[8];
    if (hipMemcpy(h, g_pen_work + 8, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { sfx_set_error("device error"); return -4; }
    for (int i = 0; i < 8; ++i) out[i] = (int64_t)h[i];
    return 0;
}
extern "C" int sfx_pen_work_get(int64_t* out /* [6] */) {
    if (!out) { sfx_set_error("null argument"); return -1; }
    for (int i = 0; i < 6; ++i) out[i] = 0;
    if (!g_pen_work) return 0;
    if (hipDeviceSynchronize() != hipSuccess) { sfx_set_error("device error"); return -4; }
    unsigned long long h[6];
    if (hipMemcpy(h, g_pen_work, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { sfx_set_error("device error"); return -4; }
    for (int i = 0; i < 6; ++i) out[i] = (int64_t)h[i];
    return 0;
}

// which form of the term new handles take (sfx_debug_pen_form): 0 = the ten general kernels on every column, every step dealt flat
// over the chip (the default: fastest on the halpe cfg's fit, 332 frames/s); 1 = grid build and pair tests over the chip, then one
// workgroup per column behind the pairs (k_pen_narrow) + the general kernels on the columns it hands over (296 frames/s: a round
// lasts as long as its most crowded column's workgroup, and the fits always carry a few collapsed meshes); 2 = form 1 with every
// column handed over (exercises the hand-over on any mesh); 3 = one workgroup per column behind the triangle boxes (k_pen_frame:
// 145 frames/s).  Same bits in every form (tests/test_gpu_topology.py); DESIGN 4.6 has the measurements.
static int g_pen_form = [] { const char* e = getenv("SFX_PEN_FORM"); return e ? atoi(e) : 0; }();
extern "C" int sfx_debug_pen_form(int32_t form) {
    const int prev = g_pen_form;
    if (form >= 0 && form <= 3) g_pen_form = form;
    return prev;
}

#ifndef PEN_SEL_ROWS
#define PEN_SEL_ROWS 64            // rows of the launches that loop over the list of wanted columns
#endif
#define PEN_MAX_BRANCHES 4
#ifndef PEN_DEFAULT_BRANCHES
#define PEN_DEFAULT_BRANCHES 1
#endif
#define PEN_BRANCH_MIN_COLS 4       // columns a branch is worth starting for
struct sfx_pen {
    PenDev P{};
    int Bmax = 0;
    int form = 1;
    // concurrent branches of an evaluation (sfx_pen_eval_masked): streams, fork / join events, the launch-wide words of each branch
    int branches = 1;
    hipStream_t br_stream[PEN_MAX_BRANCHES] = {};
    hipEvent_t br_fork = nullptr, br_join[PEN_MAX_BRANCHES] = {};
    int* br_callno[PEN_MAX_BRANCHES] = {}; int* br_ovm[PEN_MAX_BRANCHES] = {}; int* br_nheavy[PEN_MAX_BRANCHES] = {}; int* br_nw[PEN_MAX_BRANCHES] = {};
    std::vector<void*> mem;
    template <typename T> T* up(const std::vector<T>& h) {
        T* d = nullptr;
        if (hipMalloc((void**)&d, std::max<size_t>(h.size(), 1) * sizeof(T)) != hipSuccess) return nullptr;
        if (!h.empty()) hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
        mem.push_back(d); return d;
    }
    template <typename T> T* zeros(size_t n) {
        T* d = nullptr;
        if (hipMalloc((void**)&d, std::max<size_t>(n, 1) * sizeof(T)) != hipSuccess) return nullptr;
        hipMemset(d, 0, std::max<size_t>(n, 1) * sizeof(T));
        mem.push_back(d); return d;
    }
};

extern "C" int sfx_pen_create(int32_t V, int32_t F, const int32_t* faces, const int32_t* segm, const int32_t* parents,
                              const int32_t* ign_pairs, int32_t n_ign, int32_t max_collisions, int32_t max_batch,
                              sfx_pen** out) {
    if (!faces || !out || V < 3 || F < 1 || max_batch < 1 || max_collisions < 1) { sfx_set_error("bad arguments"); return -1; }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { sfx_set_error("no HIP device: libsfx has no CPU fallback"); return -3; }
    sfx_pen* h = new sfx_pen();
    PenDev& P = h->P;
    P.V = V; P.F = F; P.cap = max_collisions; h->Bmax = max_batch;
    P.work = pen_work_buffer(); P.over = nullptr;
    P.pcap = std::max(P.cap, std::min(2 * P.cap, 2048));
    std::vector<int> fv(faces, faces + (size_t)F * 3), sg(F, 0);
    int np = 1;
    if (segm) { for (int f = 0; f < F; ++f) { sg[f] = segm[f]; np = std::max(np, segm[f] + 1); } }
    if (np > 64 || F >= (1 << 24)) { sfx_set_error("at most 64 parts and 2^24 faces"); delete h; return -1; }
    for (int f = 0; f < F; ++f) if (sg[f] < 0) { sfx_set_error("negative part label"); delete h; return -1; }
    // part-level table: same part, parent / child, or listed in ign_part_pairs (fit_single_frame.py:318-328)
    std::vector<unsigned char> skip((size_t)np * np, 0);
    if (segm) {
        std::vector<int> ppar(np, -2);
        for (int f = 0; f < F; ++f) ppar[segm[f]] = parents ? parents[f] : -1;
        for (int a = 0; a < np; ++a) for (int b2 = 0; b2 < np; ++b2)
            skip[(size_t)a * np + b2] = (a == b2) || (ppar[b2] == a) || (ppar[a] == b2);
        for (int i = 0; i < n_ign; ++i) {
            const int a = ign_pairs[2 * i], b2 = ign_pairs[2 * i + 1];
            if (a >= 0 && a < np && b2 >= 0 && b2 < np) { skip[(size_t)a * np + b2] = 1; skip[(size_t)b2 * np + a] = 1; }
        }
    }
    P.n_parts = np;
    std::vector<int> vs(V + 1, 0), vl((size_t)F * 3);
    for (size_t i = 0; i < fv.size(); ++i) {
        if (fv[i] < 0 || fv[i] >= V) { sfx_set_error("face index out of range"); delete h; return -1; }
        vs[fv[i] + 1]++;
    }
    for (int v = 0; v < V; ++v) vs[v + 1] += vs[v];
    { std::vector<int> cur(vs.begin(), vs.end() - 1); for (size_t i = 0; i < fv.size(); ++i) vl[cur[fv[i]]++] = (int)i; }
    std::vector<unsigned long long> skm(64, 0ull);
    for (int a = 0; a < np; ++a) for (int b2 = 0; b2 < np; ++b2) if (skip[(size_t)a * np + b2]) skm[a] |= 1ull << b2;
    P.skipmask = h->up(skm);
    { std::vector<int4> f4(F); for (int f = 0; f < F; ++f) f4[f] = make_int4(fv[(size_t)f * 3], fv[(size_t)f * 3 + 1], fv[(size_t)f * 3 + 2], 0); P.faces4 = h->up(f4); }
    P.faces = h->up(fv); P.segm = h->up(sg); P.skip = h->up(skip); P.vf_start = h->up(vs); P.vf_list = h->up(vl);
    const size_t B = max_batch;
    P.ent_cap = F * 32;
    P.aabb = h->zeros<float>(B * F * 6); P.entries = h->zeros<int2>(B * P.ent_cap);
    P.tlist = nullptr; P.tcount = h->zeros<int>(B * 16);
    P.cand = h->zeros<int2>(B * (size_t)P.ent_cap);
    if (!P.cand) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    {   // part boxes start EMPTY (k_pen_g3 leaves them empty again after every evaluation)
        std::vector<int> pb(B * 64 * 6);
        for (size_t i = 0; i < pb.size(); ++i) pb[i] = (i % 6) < 3 ? 0x7fffffff : (int)0x80000000;
        P.pbox = h->up(pb);
    }
    P.gpart = h->zeros<float>(B * PEN_GW * 8);
    P.partners = h->zeros<int>(B * F * P.pcap); P.pcount = h->zeros<int>(B * F); P.pavail = h->zeros<int>(B * F);
    P.ovq = h->zeros<int>(B * F); P.ovn = h->zeros<int>(B * 2); P.callno = h->zeros<int>(2); P.ovm = h->zeros<int>(B + 1);
    P.pair_cap = (int)std::min<size_t>((size_t)F * P.cap, std::max<size_t>(65536, (size_t)16 * F));
    P.hasp_words = (F + 31) / 32; P.hasp = h->zeros<unsigned>(B * P.hasp_words);
    P.poff = h->zeros<int>(B * F); P.pown = h->zeros<int>(B * P.pair_cap); P.plist = h->zeros<int>(B * P.pair_cap);
    P.pout = h->zeros<float>(B * 10 * P.pair_cap); P.ptotal = h->zeros<int>(B); P.stats = h->zeros<int>(B * PEN_STATS);
    P.cells = h->zeros<int>(B * (PEN_CELLS + 1)); P.gridp = h->zeros<float>(B * 4);
    P.wq_cap = std::max(1024, P.ent_cap / 8);      // (a block of 64 entries queues at most PEN_MAX_CHUNK - 1 chunks; a full queue makes the block walk on itself)
    P.wq = h->zeros<int2>(B * (size_t)P.wq_cap); P.wqn = h->zeros<int>(B);
    if (!P.wq || !P.wqn) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    P.tgrad = h->zeros<float>(B * F * 9); P.tloss = h->zeros<float>(B * F);
    h->form = g_pen_form;
    P.n_clus = (F + 63) / 64;
    {
        std::vector<unsigned long long> cpm(P.n_clus, 0ull);
        for (int f = 0; f < F; ++f) cpm[f >> 6] |= 1ull << sg[f];
        P.cpm = h->up(cpm);
        P.wbox = h->zeros<float>(B * P.n_clus * 6);
        if (!P.cpm || !P.wbox) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    }
    P.heavy = h->zeros<int>(B); P.hlist = h->zeros<int>(B); P.nheavy = h->zeros<int>(1); P.pcnt = h->zeros<int>(B);
    P.wl = h->zeros<int>(B); P.nw = h->zeros<int>(1); P.rb = h->zeros<int>((size_t)B * P.n_clus); P.nrb = h->zeros<int>(B); P.lq = h->zeros<int>((size_t)B * F); P.nlq = h->zeros<int>(B);
    P.pbuf = reinterpret_cast<int2*>(P.partners);         // (the partner lists are unused on the fast path)
    P.pf_cap = (int)std::min<size_t>(PEN_FP, (size_t)F * P.pcap / 2);
    { const char* e = getenv("SFX_PEN_FAST_PAIRS"); if (e && atoi(e) > 0) P.pf_cap = std::min(P.pf_cap, atoi(e)); }      // (measurement switch: columns with more pairs go to the general kernels)
    P.fast_ok = ((unsigned long long)F * (unsigned long long)F < (1ull << 32)) && (2 * ((F + 31) / 32) + (V + 31) / 32 + V <= PEN_GRID_INTS) &&
                (size_t)2 * PEN_FP <= (size_t)P.pair_cap ? 1 : 0;
    if (!P.heavy || !P.hlist || !P.nheavy || !P.pcnt || !P.wl || !P.nw || !P.rb || !P.nrb || !P.lq || !P.nlq) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    if (!P.cells || !P.gridp || !P.tgrad || !P.tloss || !P.tcount || !P.pbox || !P.gpart || !P.aabb || !P.entries) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    if (!P.stats || !P.pout || !P.plist || !P.pown || !P.poff || !P.partners || !P.pavail || !P.ptotal || !P.ovq || !P.ovn || !P.callno || !P.ovm) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    {   // branches of an evaluation (sfx_pen_eval_masked); SFX_PEN_BRANCHES=1: the single chain of rounds 1-4 (A/B switch, same bits)
        const char* e = getenv("SFX_PEN_BRANCHES");
        h->branches = std::max(1, std::min(PEN_MAX_BRANCHES, e && atoi(e) > 0 ? atoi(e) : PEN_DEFAULT_BRANCHES));
        bool ok = hipEventCreateWithFlags(&h->br_fork, hipEventDisableTiming) == hipSuccess;
        h->br_callno[0] = P.callno; h->br_ovm[0] = P.ovm; h->br_nheavy[0] = P.nheavy; h->br_nw[0] = P.nw;
        for (int i = 1; i < h->branches && ok; ++i) {
            ok = hipStreamCreateWithFlags(&h->br_stream[i], hipStreamNonBlocking) == hipSuccess &&
                 hipEventCreateWithFlags(&h->br_join[i], hipEventDisableTiming) == hipSuccess;
            h->br_callno[i] = h->zeros<int>(2); h->br_ovm[i] = h->zeros<int>(B + 1); h->br_nheavy[i] = h->zeros<int>(1); h->br_nw[i] = h->zeros<int>(1);
            ok = ok && h->br_callno[i] && h->br_ovm[i] && h->br_nheavy[i] && h->br_nw[i];
        }
        if (!ok) { sfx_set_error("cannot create the streams / events of the interpenetration branches"); sfx_pen_destroy(h); return -2; }
    }
    *out = h;
    return 0;
}

int sfx_pen_capacity(const sfx_pen* h) { return h ? h->Bmax : 0; }

extern "C" int sfx_pen_set_point2plane(sfx_pen* h, int32_t on) {
    if (!h) { sfx_set_error("null handle"); return -1; }
    h->P.p2p = on ? 1 : 0;
    return 0;
}
extern "C" void sfx_pen_destroy(sfx_pen* h) {
    if (!h) return;
    for (int i = 1; i < PEN_MAX_BRANCHES; ++i) { if (h->br_stream[i]) hipStreamDestroy(h->br_stream[i]); if (h->br_join[i]) hipEventDestroy(h->br_join[i]); }
    if (h->br_fork) hipEventDestroy(h->br_fork);
    for (void* p : h->mem) hipFree(p);
    delete h;
}

// The handle's buffers as a branch of an evaluation sees them: every per-column array advanced by c0 columns (so that the branch's
// column 0 is the call's column c0 and every result lands where the single chain would have put it), the launch-wide words its own.
static PenDev pen_view(const PenDev& P, const int c0, int* callno, int* ovm, int* nheavy, int* nw) {
    PenDev Q = P;
    const size_t c = (size_t)c0, F = (size_t)P.F;
    Q.aabb += c * F * 6; Q.entries += c * P.ent_cap; Q.cand += c * P.ent_cap; if (Q.tlist) Q.tlist += c * F;
    Q.tcount += c * 16; Q.pbox += c * 64 * 6; Q.gpart += c * PEN_GW * 8;
    Q.partners += c * F * P.pcap; Q.pavail += c * F; Q.pcount += c * F; Q.poff += c * F; Q.hasp += c * P.hasp_words;
    Q.pown += c * P.pair_cap; Q.plist += c * P.pair_cap; Q.pout += c * 10 * P.pair_cap; Q.tgrad += c * F * 9; Q.tloss += c * F;
    Q.ptotal += c; Q.cells += c * (PEN_CELLS + 1); Q.gridp += c * 4; Q.stats += c * PEN_STATS;
    Q.wq += c * P.wq_cap; Q.wqn += c; Q.ovq += c * F; Q.ovn += c * 2;
    Q.heavy += c; Q.hlist += c; if (Q.wbox) Q.wbox += c * P.n_clus * 6; Q.pcnt += c; Q.wl += c; Q.rb += c * P.n_clus; Q.nrb += c; Q.lq += c * F; Q.nlq += c;
    Q.pbuf = reinterpret_cast<int2*>(Q.partners);      // (only the general kernels run in branches: unused)
    Q.callno = callno; Q.ovm = ovm; Q.nheavy = nheavy; Q.nw = nw;
    return Q;
}

// the kernels of one evaluation for the columns [0, B) of the view P0 (the handle's buffers, or a branch's share of them: pen_view)
static int pen_eval_cols(sfx_pen* h, const PenDev& P0, int32_t B, const float* verts_dev, float sigma, int32_t penalize_outside,
                         float* loss_dev, float* dverts_dev, const int* want_dev, const PenAdjPrep* prep, int* over_dev, hipStream_t s) {
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)k_pen_list, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 8192) != hipSuccess ||
            hipFuncSetAttribute((const void*)k_pen_g3, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((PEN_GRID_INTS + PEN_CELLS) * sizeof(int))) != hipSuccess) {
            sfx_set_error("cannot reserve LDS for k_pen_list / k_pen_g3"); return -2; }
        attr_set = true;
    }
    if ((size_t)(P0.F + P0.hasp_words) * sizeof(int) > 160 * 1024 - 8192) { sfx_set_error("mesh of %d faces: k_pen_list stages the counts in LDS (<= 38 k faces)", P0.F); return -1; }
    PenAdjPrep ap{};
    if (prep) ap = *prep;
    static const bool chunks_off = getenv("SFX_PEN_WALK_CHUNKS_OFF") != nullptr;      // (A/B measurement switch: same pair set either way)
    static const bool flat_off = getenv("SFX_PEN_FLAT_OFF") != nullptr;               // (A/B measurement switch: same numbers either way)
    int cap_pad = 64;
    while (cap_pad < P0.pcap) cap_pad <<= 1;
    const int rank_rows = PEN_RANK_BLOCKS + (pen_rank_tile(P0.pcap) > 0 && P0.cap + 64 <= pen_rank_tile(P0.pcap) ? PEN_RANK_HELPERS : 0);
    const size_t rank_lds = (size_t)4 * std::min(std::max(cap_pad, 128), 2048) * sizeof(int);
    const size_t list_lds = (size_t)(P0.F + P0.hasp_words) * sizeof(int);
    PenDev Pl = P0;
    Pl.over = over_dev;         // (per call: the caller's per-mesh "arrival order decided" flags, or NULL)
    const bool fused = h->form != 0 && B <= PEN_FLAT_MAXB && !chunks_off && !flat_off;
    if (fused) {
        static bool frame_attr = false;
        const size_t narrow_lds = (size_t)(2 * PEN_FP + 2 * P0.hasp_words + (P0.V + 31) / 32 + P0.V) * sizeof(int);
        if (!frame_attr) {
            if (hipFuncSetAttribute((const void*)k_pen_frame<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PEN_FRAME_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_pen_frame<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PEN_FRAME_LDS) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_pen_narrow<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)k_pen_narrow<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
                sfx_set_error("cannot reserve LDS for k_pen_narrow / k_pen_frame"); return -2; }
            frame_attr = true;
        }
        const PenSel all{want_dev, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(k_pen_g1, dim3(PEN_GW, (B + 7) & ~7), dim3(PEN_T), 0, s, P0, verts_dev, want_dev, dverts_dev, ap.adj_G, ap.Vpad, B, (float*)nullptr);
        if (h->form == 3) {
            // one workgroup per column behind the boxes (measured and not the default: the pair tests of a column are 70-150 us of ALU
            // work on ONE compute unit, and a round lasts as long as its slowest column)
            if (P0.p2p) hipLaunchKernelGGL(k_pen_frame<true>, dim3(B), dim3(PEN_T), PEN_FRAME_LDS, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, want_dev, ap, 0);
            else hipLaunchKernelGGL(k_pen_frame<false>, dim3(B), dim3(PEN_T), PEN_FRAME_LDS, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, want_dev, ap, 0);
        } else {
            // round 5's default: grid build and pair tests over the chip, the pairs into one list per column, one workgroup per column behind them
            hipLaunchKernelGGL(k_pen_g2, dim3(PEN_GW, (B + 7) & ~7), dim3(PEN_T), 0, s, P0, want_dev, B);
            hipLaunchKernelGGL(k_pen_g3, dim3(B), dim3(PEN_T), (size_t)(PEN_GRID_INTS + PEN_CELLS) * sizeof(int), s, P0, want_dev);
            hipLaunchKernelGGL(k_pen_walk, dim3(PEN_WALK_BLOCKS, B), dim3(256), 0, s, P0, all, 1, 0);
            hipLaunchKernelGGL(k_pen_walk2, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, B, all, 1);
            if (P0.p2p) hipLaunchKernelGGL(k_pen_narrow<true>, dim3(B), dim3(PEN_T), narrow_lds, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, want_dev, ap, h->form == 2 ? 1 : 0);
            else hipLaunchKernelGGL(k_pen_narrow<false>, dim3(B), dim3(PEN_T), narrow_lds, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, want_dev, ap, h->form == 2 ? 1 : 0);
        }
        // the general kernels on the columns handed over (usually none: each of these seven launches then ends after one load)
        const PenSel hv{nullptr, P0.hlist, P0.nheavy, P0.heavy};
        const int HY = std::min(B, PEN_HEAVY_ROWS);
        hipLaunchKernelGGL(k_pen_walk, dim3(PEN_WALK_BLOCKS, HY), dim3(256), 0, s, P0, hv, 0, 0);
        hipLaunchKernelGGL(k_pen_walk2, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, B, hv, 0);
        hipLaunchKernelGGL(k_pen_list, dim3(HY), dim3(PEN_T), list_lds, s, Pl, hv);
        hipLaunchKernelGGL(k_pen_rank, dim3(rank_rows, HY), dim3(256), rank_lds, s, P0, hv, cap_pad, 0);
        if (P0.p2p) hipLaunchKernelGGL(k_pen_eval<true>, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, verts_dev, sigma, penalize_outside, B, 1, hv);
        else hipLaunchKernelGGL(k_pen_eval<false>, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, verts_dev, sigma, penalize_outside, B, 1, hv);
        hipLaunchKernelGGL(k_pen_facesum, dim3(PEN_EVAL_BLOCKS, HY), dim3(256), 0, s, P0, hv);
        hipLaunchKernelGGL(k_pen_gather, dim3((std::max(P0.V, 1) + 255) / 256, HY), dim3(256), (size_t)P0.hasp_words * sizeof(unsigned), s,
                           P0, dverts_dev, loss_dev, hv, ap);
    } else {
    const PenSel all{want_dev, nullptr, nullptr, nullptr};
    // with a mask: the launches behind the grid build take their columns from the list k_pen_g1 makes of the wanted ones (rows loop
    // over it); SFX_PEN_ROWS_OFF: a grid row per column of the call, as until round 5 (A/B switch, same bits)
    const bool rows_off = getenv("SFX_PEN_ROWS_OFF") != nullptr;      // (read per call: a batch captures its graphs with the value of its time)
    const bool listed = want_dev && !rows_off;
    const PenSel cw = listed ? PenSel{nullptr, P0.wl, P0.nw, nullptr} : all;
    const int RY = listed ? std::min(B, PEN_SEL_ROWS) : B;
    // grid build (the cross-workgroup accumulators -- part boxes, survivor counts -- are left empty by k_pen_g3 of the previous evaluation)
    hipLaunchKernelGGL(k_pen_g1, dim3(PEN_GW, (B + 7) & ~7), dim3(PEN_T), 0, s, P0, verts_dev, want_dev, (float*)nullptr, (float*)nullptr, 0, B, listed ? loss_dev : (float*)nullptr);
    hipLaunchKernelGGL(k_pen_g2, dim3(PEN_GW, (B + 7) & ~7), dim3(PEN_T), 0, s, P0, want_dev, B);
    hipLaunchKernelGGL(k_pen_g3, dim3(B), dim3(PEN_T), (size_t)(PEN_GRID_INTS + PEN_CELLS) * sizeof(int), s, P0, want_dev);
    {
        PenDev Pw = P0;
        const bool queued = B <= PEN_FLAT_MAXB && !chunks_off;
        if (!queued) Pw.wq_cap = 0;                // every block walks its bucket to the end itself
        if (queued && !flat_off && !rows_off)      // one flat list of the blocks of 64 entries
            hipLaunchKernelGGL(k_pen_walk, dim3(PEN_WALK_FLAT), dim3(256), (size_t)(B + 1) * sizeof(int), s, Pw, all, 0, B);
        else hipLaunchKernelGGL(k_pen_walk, dim3(PEN_WALK_BLOCKS, RY), dim3(256), 0, s, Pw, cw, 0, 0);
        if (queued) hipLaunchKernelGGL(k_pen_walk2, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, Pw, B, all, 0);
    }
    hipLaunchKernelGGL(k_pen_list, dim3(B), dim3(PEN_T), list_lds, s, Pl, all);
    if (B <= PEN_FLAT_MAXB && !flat_off && !rows_off && cap_pad <= 2048)      // one flat list of the blocks that have pairs (k_pen_list: P.rb)
        hipLaunchKernelGGL(k_pen_rank, dim3(PEN_RANK_FLAT + PEN_RANK_HELPERS), dim3(256), rank_lds + (size_t)(B + 1) * sizeof(int), s, P0, all, cap_pad, B);
    else hipLaunchKernelGGL(k_pen_rank, dim3(rank_rows, RY), dim3(256), rank_lds, s, P0, cw, cap_pad, 0);
    if (B <= PEN_FLAT_MAXB && !flat_off)
        { if (P0.p2p) hipLaunchKernelGGL(k_pen_eval<true>, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, verts_dev, sigma, penalize_outside, B, 1, all);
          else hipLaunchKernelGGL(k_pen_eval<false>, dim3(PEN_FLAT_BLOCKS), dim3(256), (size_t)(B + 1) * sizeof(int), s, P0, verts_dev, sigma, penalize_outside, B, 1, all); }
    else
        { if (P0.p2p) hipLaunchKernelGGL(k_pen_eval<true>, dim3(PEN_EVAL_BLOCKS, B), dim3(256), 0, s, P0, verts_dev, sigma, penalize_outside, B, 0, all);
          else hipLaunchKernelGGL(k_pen_eval<false>, dim3(PEN_EVAL_BLOCKS, B), dim3(256), 0, s, P0, verts_dev, sigma, penalize_outside, B, 0, all); }
    hipLaunchKernelGGL(k_pen_facesum, dim3(PEN_EVAL_BLOCKS, RY), dim3(256), 0, s, P0, cw);
    // (the gather keeps a row per column: 41 workgroups per column leave few to be turned away, and rows that loop cost it its p90 --
    //  48 -> 64 us where most columns carry the term)
    hipLaunchKernelGGL(k_pen_gather, dim3((std::max(P0.V, 1) + 255) / 256, B), dim3(256), (size_t)P0.hasp_words * sizeof(unsigned), s,
                       P0, dverts_dev, loss_dev, all, ap);
    }
    if (hipGetLastError() != hipSuccess) { sfx_set_error("penetration kernels failed to launch"); return -4; }
    return 0;
}

int sfx_pen_eval_masked(sfx_pen* h, int32_t B, const float* verts_dev, float sigma, int32_t penalize_outside,
                        float* loss_dev, float* dverts_dev, const int* want_dev, const PenAdjPrep* prep, int* over_dev, void* stream) {
    if (!h || !verts_dev || !loss_dev || !dverts_dev) { sfx_set_error("null argument"); return -1; }
    if (B < 1 || B > h->Bmax) { sfx_set_error("batch %d exceeds the capacity %d given to sfx_pen_create", B, h->Bmax); return -1; }
    if (!(sigma > 0.f)) { sfx_set_error("df_cone_height must be positive"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    static const bool rewalk_off = [] { const char* e = getenv("SFX_PEN_REWALK_OFF"); return e && atoi(e) != 0; }();      // (A/B measurement switch)
    h->P.no_rewalk = rewalk_off ? 1 : 0;
    // Round 5: the columns of a call in `branches` contiguous shares, each share's ten kernels on a stream of its own (fork / join
    // on events: parallel branches of the captured graph).  Every kernel of the step is a few dependent memory round trips with the
    // chip mostly idle: two chains side by side hide each other's trips.  A column's numbers do not depend on which
    // other columns share its launches, so the bits are those of one chain.  The shares work on views of the handle's buffers
    // (pen_view: every per-column array advanced by the share's first column; the few launch-wide words -- evaluation counter,
    // the list of meshes with an overflow queue -- exist once per branch).
    const int nbr = (h->form == 0 && h->branches > 1) ? std::min(h->branches, std::max(1, B / PEN_BRANCH_MIN_COLS)) : 1;
    if (nbr <= 1) return pen_eval_cols(h, h->P, B, verts_dev, sigma, penalize_outside, loss_dev, dverts_dev, want_dev, prep, over_dev, s);
    if (hipEventRecord(h->br_fork, s) != hipSuccess) { sfx_set_error("event record failed"); return -4; }
    int rc = 0;
    for (int i = 0; i < nbr; ++i) {
        const int c0 = (int)((long long)B * i / nbr), c1 = (int)((long long)B * (i + 1) / nbr), n = c1 - c0;
        hipStream_t si = i == 0 ? s : h->br_stream[i];
        if (i > 0 && hipStreamWaitEvent(si, h->br_fork, 0) != hipSuccess) { sfx_set_error("stream wait failed"); return -4; }
        const PenDev Pv = pen_view(h->P, c0, h->br_callno[i], h->br_ovm[i], h->br_nheavy[i], h->br_nw[i]);
        PenAdjPrep ap{};
        if (prep) { ap = *prep; ap.AT += c0; ap.adj_G += (size_t)c0 * 3 * ap.Vpad; }
        const size_t V3 = (size_t)h->P.V * 3;
        const int r = pen_eval_cols(h, Pv, n, verts_dev + c0 * V3, sigma, penalize_outside, loss_dev + c0, dverts_dev + c0 * V3,
                                    want_dev ? want_dev + c0 : nullptr, prep ? &ap : nullptr, over_dev ? over_dev + c0 : nullptr, si);
        if (r && !rc) rc = r;
        if (i > 0 && (hipEventRecord(h->br_join[i], si) != hipSuccess || hipStreamWaitEvent(s, h->br_join[i], 0) != hipSuccess)) {
            sfx_set_error("join of the interpenetration branches failed"); return -4; }
    }
    return rc;
}

extern "C" int sfx_pen_eval(sfx_pen* h, int32_t B, const float* verts_dev, float sigma, int32_t penalize_outside,
                            float* loss_dev, float* dverts_dev, void* stream) {
    return sfx_pen_eval_masked(h, B, verts_dev, sigma, penalize_outside, loss_dev, dverts_dev, nullptr, nullptr, nullptr, stream);
}

// ---- DistanceFieldPenetrationLoss(triangles, collision_idxs) stand-alone (fitting.py:451-455): the caller supplies the pairs.
// k_pen_pairs_in stages a mesh's pairs ([n][2] triangle ids, any order within a pair, each unordered pair once, rows with a
// negative id empty -- the package's -1 padding) where the pair tests would have left them, k_pen_narrow does the rest.
__global__ __launch_bounds__(PEN_T)
void k_pen_pairs_in(PenDev P, const int* __restrict__ pairs, const int n, float* __restrict__ dverts, float* __restrict__ dtri) {
    __shared__ int s_n;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63;
    if (b == 0 && t == 0) P.nheavy[0] = 0;
    if (t == 0) s_n = 0;
    for (int i = t; i < P.V * 3; i += PEN_T) dverts[(size_t)b * P.V * 3 + i] = 0.f;
    if (dtri) for (int i = t; i < P.F * 9; i += PEN_T) dtri[(size_t)b * P.F * 9 + i] = 0.f;
    __syncthreads();
    int2* pbuf = P.pbuf + (size_t)b * P.pf_cap;
    const int2* in = reinterpret_cast<const int2*>(pairs) + (size_t)b * n;
    for (int i0 = 0; i0 < n; i0 += PEN_T) {
        const int i = i0 + t;
        const int2 pr = i < n ? in[i] : make_int2(-1, -1);
        const bool ok = pr.x >= 0 && pr.y >= 0 && pr.x < P.F && pr.y < P.F && pr.x != pr.y;
        const unsigned long long m = __ballot(ok);
        int base = 0;
        if (lane == 0 && m) base = atomicAdd(&s_n, __popcll(m));
        base = __builtin_amdgcn_readfirstlane(base);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (ok && pos < P.pf_cap) pbuf[pos] = pr;
    }
    __syncthreads();
    if (t == 0) { int* st = P.stats + b * PEN_STATS; for (int q = 0; q < PEN_STATS; ++q) st[q] = 0; P.pcnt[b] = s_n; P.wqn[b] = 0; }
}
// per-corner gradient of the triangles that have pairs (the others' rows were zeroed): the gradient with respect to the
// `triangles` tensor the caller differentiates through
__global__ void k_pen_dtri_out(PenDev P, float* __restrict__ dtri) {
    const int b = blockIdx.y;
    const int total = P.ptotal[b];
    const int* pown = P.pown + (size_t)b * P.pair_cap;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int f = pown[i];
        if (i > 0 && pown[i - 1] == f) continue;
        for (int j = 0; j < 9; ++j) dtri[((size_t)b * P.F + f) * 9 + j] = P.tgrad[((size_t)b * P.F + f) * 9 + j];
    }
}
extern "C" int sfx_pen_eval_pairs(sfx_pen* h, int32_t B, const float* verts_dev, const int32_t* pairs_dev, int32_t n_pairs, float sigma,
                                  int32_t penalize_outside, float* loss_dev, float* dverts_dev, float* dtri_dev, void* stream) {
    if (!h || !verts_dev || !loss_dev || !dverts_dev || (n_pairs > 0 && !pairs_dev)) { sfx_set_error("null argument"); return -1; }
    if (B < 1 || B > h->Bmax) { sfx_set_error("batch %d exceeds the capacity %d given to sfx_pen_create", B, h->Bmax); return -1; }
    if (!(sigma > 0.f) || n_pairs < 0) { sfx_set_error("bad arguments"); return -1; }
    if (!h->P.fast_ok) { sfx_set_error("mesh too large for the stand-alone pair evaluation (one workgroup sorts a mesh's pairs in LDS)"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const size_t narrow_lds = (size_t)(2 * PEN_FP + 2 * h->P.hasp_words + (h->P.V + 31) / 32 + h->P.V) * sizeof(int);
    if (hipFuncSetAttribute((const void*)k_pen_narrow<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)k_pen_narrow<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
        sfx_set_error("cannot reserve LDS for k_pen_narrow"); return -2; }
    PenDev Pl = h->P;
    Pl.over = nullptr;
    hipLaunchKernelGGL(k_pen_pairs_in, dim3(B), dim3(PEN_T), 0, s, h->P, pairs_dev, n_pairs, dverts_dev, dtri_dev);
    PenAdjPrep ap{};
    if (h->P.p2p) hipLaunchKernelGGL(k_pen_narrow<true>, dim3(B), dim3(PEN_T), narrow_lds, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, (const int*)nullptr, ap, 0);
    else hipLaunchKernelGGL(k_pen_narrow<false>, dim3(B), dim3(PEN_T), narrow_lds, s, Pl, verts_dev, sigma, penalize_outside, dverts_dev, loss_dev, (const int*)nullptr, ap, 0);
    if (dtri_dev) hipLaunchKernelGGL(k_pen_dtri_out, dim3(32, B), dim3(256), 0, s, h->P, dtri_dev);
    if (hipGetLastError() != hipSuccess) { sfx_set_error("penetration kernels failed to launch"); return -4; }
    int nh = 0;
    if (hipStreamSynchronize(s) != hipSuccess || hipMemcpy(&nh, h->P.nheavy, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { sfx_set_error("device error"); return -4; }
    if (nh > 0) { sfx_set_error("%d mesh(es) carry more than %d pairs: beyond what the stand-alone pair evaluation sorts in one workgroup's LDS", nh, h->P.pf_cap); return -1; }
    return 0;
}

// debug: wall-clock ticks (100 MHz) at the end of k_pen_pairs' steps for the first B frames: [B][10] = triangle boxes,
// frame box, part boxes, part culling, grid histogram, scan, scatter ([7..9] unused: those steps are kernels of their own); [10] = grid entries
extern "C" int sfx_pen_phase_clocks(sfx_pen* h, int32_t B, int32_t* out) {
    if (!h || !out || B < 1 || B > h->Bmax) return -1;
    std::vector<int> st((size_t)B * PEN_STATS);
    hipDeviceSynchronize();
    hipMemcpy(st.data(), h->P.stats, st.size() * sizeof(int), hipMemcpyDeviceToHost);
    for (int i = 0; i < B; ++i) for (int k = 0; k < 11; ++k) out[i * 11 + k] = st[(size_t)i * PEN_STATS + 4 + k];
#ifdef PEN_COUNT
    {   // per frame: grid entries, candidates by the test they die on, wavefront steps of the walk -- and how unevenly the frames carry them
        long tot[6] = {0, 0, 0, 0, 0, 0}, mx[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < B; ++i) {
            const int* r = &st[(size_t)i * PEN_STATS];
            const long v[6] = {r[14], r[16], r[17], r[18], r[19], r[20]};
            for (int q = 0; q < 6; ++q) { tot[q] += v[q]; mx[q] = std::max(mx[q], v[q]); }
        }
        long ph[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < B; ++i) for (int q = 0; q < 6; ++q) ph[q] += st[(size_t)i * PEN_STATS + 24 + q];
        fprintf(stderr, "[pen count] k_pen_g3 mean shader cycles per phase: init %ld, part masks %ld, histogram %ld, scan %ld, scatter %ld, copy %ld\n",
                ph[0] / B, ph[1] / B, ph[2] / B, ph[3] / B, ph[4] / B, ph[5] / B);
        fprintf(stderr, "[pen count] %d frames, mean / max per frame: entries %ld / %ld; walked %ld / %ld, same cell %ld / %ld, part mask passed %ld / %ld, "
                "boxes overlap %ld / %ld; wavefront steps %ld / %ld\n", B, tot[0] / B, mx[0], tot[1] / B, mx[1], tot[2] / B, mx[2], tot[3] / B, mx[3],
                tot[4] / B, mx[4], tot[5] / B, mx[5]);
    }
#endif
    return 0;
}

// stats rows (device, [n][PEN_STATS]) -> the four public figures per mesh
int sfx_pen_stats_from(const int* stats_dev, int n, int32_t* stats_host) {
    if (hipDeviceSynchronize() != hipSuccess) { sfx_set_error("device error"); return -4; }
    std::vector<int> st((size_t)n * PEN_STATS);
    if (hipMemcpy(st.data(), stats_dev, st.size() * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { sfx_set_error("device error"); return -4; }
    for (int i = 0; i < n; ++i) {
        const int* r = &st[(size_t)i * PEN_STATS];
        // r[15]: ordered pairs in the list that only one of the two triangles kept (max_collisions cut the other's list):
        // they contribute nothing and count as dropped
        stats_host[i * 4 + 0] = r[0] - r[15]; stats_host[i * 4 + 1] = r[1] + r[15];
        stats_host[i * 4 + 2] = r[2]; stats_host[i * 4 + 3] = r[13];      // [3]: walks cut short at PEN_MAX_WALK entries (0 on a sane mesh)
    }
    return 0;
}
int sfx_pen_stats_stride(void) { return PEN_STATS; }
const int* sfx_pen_stats_dev(const sfx_pen* h) { return h ? h->P.stats : nullptr; }

// The frame's pair list (k_pen_rank: receiving triangle ascending, partner ascending) -> HOST [cap][2]; *n_out = ordered pairs
// in the list (may exceed cap: then the first cap are copied).  What BVH(...)(triangles) followed by FilterFaces(...) hands to the
// loss in the reference (fitting.py:445-450) -- here both orders of every pair.
extern "C" int sfx_pen_pairs(sfx_pen* h, int32_t mesh, int32_t cap, int32_t* pairs_host, int32_t* n_out) {
    if (!h || !n_out || mesh < 0 || mesh >= h->Bmax || cap < 0 || (cap > 0 && !pairs_host)) { sfx_set_error("bad arguments"); return -1; }
    if (hipDeviceSynchronize() != hipSuccess) { sfx_set_error("device error"); return -4; }
    int tot = 0;
    if (hipMemcpy(&tot, h->P.ptotal + mesh, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { sfx_set_error("device error"); return -4; }
    *n_out = tot;
    const int n = std::min(tot, cap);
    if (n > 0) {
        std::vector<int> a(n), b(n);
        if (hipMemcpy(a.data(), h->P.pown + (size_t)mesh * h->P.pair_cap, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(b.data(), h->P.plist + (size_t)mesh * h->P.pair_cap, (size_t)n * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) { sfx_set_error("device error"); return -4; }
        for (int i = 0; i < n; ++i) { pairs_host[2 * i] = a[i]; pairs_host[2 * i + 1] = b[i]; }
    }
    return 0;
}

extern "C" int sfx_pen_stats(sfx_pen* h, int32_t B, int32_t* stats_host /* [B][4] */) {
    if (!h || !stats_host || B < 1 || B > h->Bmax) { sfx_set_error("bad arguments"); return -1; }
    return sfx_pen_stats_from(h->P.stats, B, stats_host);
}
