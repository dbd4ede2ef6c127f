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
// MI355X design (not the package's LBVH; LAB_NOTES.md §4.6 has the table):
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
    int* callno;               // [2] evaluations so far (k_pen_g1), and the last one in which some list overflowed (k_pen_list): k_pen_rank looks for queues only then
    int* ovm;                  // [1 + B] meshes of this evaluation whose overflow queue k_pen_rank has to drain: count (k_pen_g1 -> 0), ids (k_pen_list)
    int* over;                 // [B] or NULL (set per call): 1 = this evaluation of the mesh kept partners by ARRIVAL order somewhere (a list beyond
                               //     2 x max_collisions, a cut walk): its numbers are not reproducible run to run
    // the one-workgroup-per-mesh path (k_pen_narrow: stand-alone pair evaluation; lab forms 1 / 2) and the columns it hands to the general kernels
    int* heavy;                // [B] 1 = this evaluation of the column goes through the general kernels (crowded grid / too many pairs for one workgroup's LDS)
    int* hlist;                // [B] the heavy columns of this evaluation, any order
    int* nheavy;               // [1] their number (k_pen_g1 -> 0, k_pen_narrow appends)
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
// blockIdx.y of the call, masked by `want`.  hlist != NULL (round 5): the compact list of columns k_pen_narrow has handed over (lab forms 1 / 2)
// ("heavy": a crowded grid or more pairs than one workgroup's LDS sorts), *nheavy of them -- the grid's rows loop over the list,
// and a launch that finds it empty (nearly every one) ends after one load.
struct PenSel { const int* want; const int* hlist; const int* nheavy; const int* heavy; };
__device__ __forceinline__ int pen_sel_n(const PenSel& s, const int rows) { return s.hlist ? min(*s.nheavy, rows) : rows; }
__device__ __forceinline__ int pen_sel_col(const PenSel& s, const int i) { return s.hlist ? s.hlist[i] : i; }
__device__ __forceinline__ bool pen_sel_on(const PenSel& s, const int b) { return s.hlist ? (!s.heavy || s.heavy[b] != 0) : (!s.want || s.want[b] != 0); }
// the column of a row's FIRST trip, requested together with the list's length (entries beyond the length are stale but in bounds:
// the list has a slot per column and a launch has at most as many rows) -- one round trip instead of two at every kernel's entry
__device__ __forceinline__ int pen_sel_first(const PenSel& s, const int row) { return s.hlist ? s.hlist[row] : row; }

#include "collide_field.h"
#include "collide_grid.h"
#include "collide_pairs.h"
#include "collide_eval.h"

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

#ifndef PEN_SEL_ROWS
#define PEN_SEL_ROWS 64            // rows of the launches that loop over the list of wanted columns
#endif
struct sfx_pen {
    PenDev P{};
    int Bmax = 0;
    int last_B = 0;              // meshes of the most recent evaluation: what sfx_pen_pairs / sfx_pen_stats may be asked about
    int form = 0;                // (lab build: sfx_debug_pen_form; the product has one form)
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

#ifdef SFX_LAB
#include "collide_lab.h"
#endif

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
#ifdef SFX_LAB
    h->form = g_pen_form;
#endif
    P.n_clus = (F + 63) / 64;
    P.cpm = nullptr; P.wbox = nullptr;        // (cluster boxes: only round 5's k_pen_frame read them)
    P.heavy = h->zeros<int>(B); P.hlist = h->zeros<int>(B); P.nheavy = h->zeros<int>(1); P.pcnt = h->zeros<int>(B);
    P.wl = h->zeros<int>(B); P.nw = h->zeros<int>(1); P.rb = h->zeros<int>((size_t)B * P.n_clus); P.nrb = h->zeros<int>(B); P.lq = h->zeros<int>((size_t)B * F); P.nlq = h->zeros<int>(B);
    P.pbuf = reinterpret_cast<int2*>(P.partners);         // (the partner lists are unused on the fast path)
    P.pf_cap = (int)std::min<size_t>(PEN_FP, (size_t)F * P.pcap / 2);
#ifdef SFX_LAB
    { const char* e = getenv("SFX_PEN_FAST_PAIRS"); if (e && atoi(e) > 0) P.pf_cap = std::min(P.pf_cap, atoi(e)); }      // (measurement switch: columns with more pairs go to the general kernels)
#endif
    P.pf_cap = std::max(1, std::min(P.pf_cap, P.pair_cap / 2));      // (a mesh's pair list holds both orders of every pair the one-workgroup form accepts: small meshes too)
    P.fast_ok = ((unsigned long long)F * (unsigned long long)F < (1ull << 32)) && (2 * ((F + 31) / 32) + (V + 31) / 32 + V <= PEN_GRID_INTS) ? 1 : 0;
    if (!P.heavy || !P.hlist || !P.nheavy || !P.pcnt || !P.wl || !P.nw || !P.rb || !P.nrb || !P.lq || !P.nlq) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    if (!P.cells || !P.gridp || !P.tgrad || !P.tloss || !P.tcount || !P.pbox || !P.gpart || !P.aabb || !P.entries) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    if (!P.stats || !P.pout || !P.plist || !P.pown || !P.poff || !P.partners || !P.pavail || !P.ptotal || !P.ovq || !P.ovn || !P.callno || !P.ovm) { sfx_set_error("out of device memory"); for (void* p : h->mem) hipFree(p); delete h; return -2; }
    *out = h;
    return 0;
}

int sfx_pen_capacity(const sfx_pen* h) { return h ? h->Bmax : 0; }
void sfx_pen_reuse(sfx_pen* h) {      // (api.hip: a handle taken from a model's idle slot by a new batch)
    if (!h) return;
    h->last_B = 0;
#ifdef SFX_LAB
    h->form = g_pen_form;
#endif
}
void sfx_pen_note_batch(sfx_pen* h, int B) { if (h) h->last_B = B; }      // (a replayed graph of the step: api.hip eval_penetration)

extern "C" int sfx_pen_set_point2plane(sfx_pen* h, int32_t on) {
    if (!h) { sfx_set_error("null handle"); return -1; }
    h->P.p2p = on ? 1 : 0;
    return 0;
}
extern "C" void sfx_pen_destroy(sfx_pen* h) {
    if (!h) return;
    for (void* p : h->mem) hipFree(p);
    delete h;
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
#ifdef SFX_LAB       // A/B measurement switches of the lab build (same pair set, same numbers either way): include/sfx_lab.h
    static const bool chunks_off = getenv("SFX_PEN_WALK_CHUNKS_OFF") != nullptr, flat_off = getenv("SFX_PEN_FLAT_OFF") != nullptr;
    const bool rows_off = getenv("SFX_PEN_ROWS_OFF") != nullptr;      // (read per call: a batch captures its graphs with the value of its time)
#else
    constexpr bool chunks_off = false, flat_off = false, rows_off = false;
#endif
    int cap_pad = 64;
    while (cap_pad < P0.pcap) cap_pad <<= 1;
    const int rank_rows = PEN_RANK_BLOCKS + (pen_rank_tile(P0.pcap) > 0 && P0.cap + 64 <= pen_rank_tile(P0.pcap) ? PEN_RANK_HELPERS : 0);
    const size_t rank_lds = (size_t)4 * std::min(std::max(cap_pad, 128), 2048) * sizeof(int);
    const size_t list_lds = (size_t)(P0.F + P0.hasp_words) * sizeof(int);
    PenDev Pl = P0;
    Pl.over = over_dev;         // (per call: the caller's per-mesh "arrival order decided" flags, or NULL)
#ifdef SFX_LAB
    if (h->form != 0 && B <= PEN_FLAT_MAXB && !chunks_off && !flat_off)        // forms 1 / 2 (collide_lab.h): k_pen_narrow behind the pair tests
        return pen_eval_cols_narrow(h, P0, Pl, B, verts_dev, sigma, penalize_outside, loss_dev, dverts_dev, want_dev, ap, list_lds, rank_lds, rank_rows, cap_pad, s);
#endif
    {
    const PenSel all{want_dev, nullptr, nullptr, nullptr};
    // with a mask: the launches behind the grid build take their columns from the list k_pen_g1 makes of the wanted ones (rows loop
    // over it)
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
    h->last_B = B;
    return pen_eval_cols(h, h->P, B, verts_dev, sigma, penalize_outside, loss_dev, dverts_dev, want_dev, prep, over_dev, s);
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
    if (!h->P.fast_ok) { sfx_set_error("mesh too large for the stand-alone pair evaluation (one workgroup holds a mesh's bit sets and vertex list in LDS: V + F / 16 <= %d; F < 65536)", PEN_GRID_INTS); return -1; }
    h->last_B = B;
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
    if (mesh >= h->last_B) { sfx_set_error("mesh %d was not part of the most recent evaluation (%d meshes): its pair list is stale", mesh, h->last_B); return -1; }
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
