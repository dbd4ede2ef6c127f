// api.hip -- host layer of libsfx.so: the C ABI of include/sfx.h over the gfx950 kernels.
// Owns HBM residency of the model constants and of every batch's state; the hot loop
// (sfx_batch_fit) is a stream of kernel launches with no host arithmetic and one small
// device->host poll every POLL ticks.
#include "../../include/sfx.h"
#include "sfx_internal.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

static thread_local char g_err[1024] = "";
void sfx_set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
extern "C" const char* sfx_last_error(void) { return g_err; }
extern "C" const char* sfx_version(void) { return "sfx 0.1.0 (gfx950)"; }

// ---------------------------------------------------------------------------------------
// profiling: HIP-event timing of named kernels on the launch stream
struct ProfAcc { double ms = 0; int64_t n = 0, seen = 0; double units = 0; std::vector<std::pair<hipEvent_t, hipEvent_t>> pend; };
static int g_prof = 0;
static int g_prof_every = 1;  // time every N-th launch of a name (events cost a queue packet each)
static int g_unfused = 0;     // debug: run the fitting loop with the stand-alone kernels
static std::map<std::string, ProfAcc> g_acc;
static std::vector<hipEvent_t> g_ev_pool;
static std::mutex g_prof_mu;     // batches may be driven from several host threads (one stream each)

static hipEvent_t ev_get() {
    if (!g_ev_pool.empty()) { hipEvent_t e = g_ev_pool.back(); g_ev_pool.pop_back(); return e; }
    hipEvent_t e = nullptr; hipEventCreate(&e); return e;
}
static void prof_flush(ProfAcc& a) {
    for (auto& p : a.pend) {
        hipEventSynchronize(p.second);
        float ms = 0; hipEventElapsedTime(&ms, p.first, p.second);
        a.ms += ms; a.n += 1;
        g_ev_pool.push_back(p.first); g_ev_pool.push_back(p.second);
    }
    a.pend.clear();
}
struct ProfScope {
    const char* name; hipStream_t s; hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(const char* n, hipStream_t st, double units = 0) : name(n), s(st) {
        if (!g_prof) return;
        std::lock_guard<std::mutex> lk(g_prof_mu);
        auto& a = g_acc[name];
        if (a.seen++ % g_prof_every) return;
        a.units += units;
        e0 = ev_get(); e1 = ev_get();
        hipEventRecord(e0, s);
    }
    ~ProfScope() {
        if (!e0) return;
        hipEventRecord(e1, s);
        std::lock_guard<std::mutex> lk(g_prof_mu);
        auto& a = g_acc[name]; a.pend.push_back({e0, e1});
        if (a.pend.size() > 4096) prof_flush(a);
    }
};
// bit 0: time launches with HIP events; bit 1: debug, stand-alone kernels; bits 8..23: time every
// N-th launch of each name only (0/1 = every launch)
extern "C" int sfx_prof_enable(int32_t on) {
    g_prof = on & 1; g_unfused = (on >> 1) & 1;
    const int every = (on >> 8) & 0xffff;
    g_prof_every = every > 1 ? every : 1;
    return 0;
}
extern "C" void sfx_prof_reset(void) { std::lock_guard<std::mutex> lk(g_prof_mu); for (auto& kv : g_acc) { prof_flush(kv.second); } g_acc.clear(); }
// Host side of the polled dense loops since the last reset (HOST [4]): seconds spent enqueueing, seconds spent waiting for a
// batch's stage flags (the GPU was ahead of the host exactly when this is ~0 while the loop's wall time exceeds the kernels'),
// wall seconds of the loops, rounds enqueued.
static double g_loop_host[4] = {0, 0, 0, 0};
extern "C" int sfx_loop_host_stats(double* out, int32_t reset) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    if (out) for (int i = 0; i < 4; ++i) out[i] = g_loop_host[i];
    if (reset) for (int i = 0; i < 4; ++i) g_loop_host[i] = 0.0;
    return 0;
}
extern "C" int sfx_prof_get(const char* name, double* total_ms, int64_t* launches, double* units) {
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto it = g_acc.find(name);
    if (it == g_acc.end()) { if (total_ms) *total_ms = 0; if (launches) *launches = 0; if (units) *units = 0; return 0; }
    prof_flush(it->second);
    if (total_ms) *total_ms = it->second.ms;
    if (launches) *launches = it->second.n;
    if (units) *units = it->second.units;
    return (int)std::min<int64_t>(it->second.n, 1 << 30);
}

// ---------------------------------------------------------------------------------------
struct DevAlloc {
    std::vector<void*> ptrs;
    bool failed = false;        // sticky: any hipMalloc / hipMemcpy / hipMemset of this owner failed (checked once by the creator)
    template <typename T> T* up(const std::vector<T>& h) {
        T* d = nullptr;
        size_t n = std::max<size_t>(h.size(), 1) * sizeof(T);
        if (hipMalloc((void**)&d, n) != hipSuccess) { failed = true; return nullptr; }
        ptrs.push_back(d);
        if (!h.empty() && hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) failed = true;
        return d;
    }
    template <typename T> T* zeros(size_t count) {
        T* d = nullptr;
        size_t n = std::max<size_t>(count, 1) * sizeof(T);
        if (hipMalloc((void**)&d, n) != hipSuccess) { failed = true; return nullptr; }
        ptrs.push_back(d);
        if (hipMemset(d, 0, n) != hipSuccess) failed = true;
        return d;
    }
    void free_all() { for (void* p : ptrs) hipFree(p); ptrs.clear(); }
};

struct sfx_pen;
extern "C" int sfx_pen_create(int32_t V, int32_t F, const int32_t* faces, const int32_t* segm, const int32_t* parents,
                              const int32_t* ign_pairs, int32_t n_ign, int32_t max_collisions, int32_t max_batch,
                              sfx_pen** out);
extern "C" void sfx_pen_destroy(sfx_pen* h);
extern "C" int sfx_pen_eval(sfx_pen* h, int32_t B, const float* verts_dev, float sigma, int32_t penalize_outside,
                            float* loss_dev, float* dverts_dev, void* stream);
extern "C" int sfx_pen_stats(sfx_pen* h, int32_t B, int32_t* stats_host);
extern "C" int sfx_pen_pairs(sfx_pen* h, int32_t mesh, int32_t cap, int32_t* pairs_host, int32_t* n_out);
int sfx_pen_eval_masked(sfx_pen* h, int32_t B, const float* verts_dev, float sigma, int32_t penalize_outside,
                        float* loss_dev, float* dverts_dev, const int* want_dev, const PenAdjPrep* prep, int* over_dev, void* stream);
int sfx_pen_capacity(const sfx_pen* h);          // meshes per call the handle's buffers hold (collide.hip)
void sfx_pen_note_batch(sfx_pen* h, int B);      // meshes of the evaluation a replayed graph performs
void sfx_pen_reuse(sfx_pen* h);                  // a handle taken from a model's idle slot: per-process settings of the lab build re-read
int sfx_pen_stats_from(const int* stats_dev, int n, int32_t* stats_host);
int sfx_pen_stats_stride(void);
const int* sfx_pen_stats_dev(const sfx_pen* h);

struct sfx_model {
    DevModel M{};
    DevAlloc mem;
    std::vector<int> faces_host, segm_host, parents_host, ign_host;   // interpenetration set-up
    int NB = 0, NE = 0, NPCA = 0;
    std::vector<int> meta_host = std::vector<int>(SFX_META_N, 0);
    sfx_batch* fwd = nullptr;     // lazily created batch behind sfx_lbs_forward
    int fwd_B = 0;
    // The collision buffers of the interpenetration term (~45 MB per GEMM column: 11.5 GB for 256 columns) outlive the batch that
    // used them: a job is a sequence of batches on one model (driver.fit_frames per (gender, H, W) group, bench.py per step), and
    // returning 11 GB to the driver and asking for them again cost 10 ms on some boxes and 450 ms per batch on others (round 6:
    // `outside_loop_ms_per_step` of the --workload pen line, 402 against 239 frames/s for the same kernels).  ONE idle handle is
    // kept per model; it is reused when it holds at least the columns asked for, at the same max_collisions and part table.
    sfx_pen* pen_idle = nullptr;
    int pen_idle_cols = 0, pen_idle_cap = 0, pen_idle_gen = -1;
    int parts_gen = 0;            // bumped by sfx_model_set_parts: handles made for another part table are not reused
    std::mutex pen_mu;
};

// The fused dense loop keeps `ahead` batches of 8 rounds queued beyond the one whose stage flags the host is waiting for: the
// GPU only idles when the host thread stays away for longer than that much queued work.  Measured in round 4 (LAB_NOTES §4.6):
// the host's own enqueueing is 4-7 % of the loop (7 us per round of 2 launches, 43 us per round of 13), so a captured graph
// would save nothing on a quiet host; on a host whose 16 cores carry 32 spinning processes EVERY workload halves (body 509 ->
// 239, interpenetration 349 -> 182 frames/s: the polling thread waits milliseconds for a time slice), and a deeper queue buys
// back part of it where a round is two launches (body: 249 / 275 frames/s with 3 / 6 batches ahead) but HURTS where a round
// is thirteen (interpenetration, quiet host: 349 / 322 / 221 frames/s with 1 / 3 / 6 ahead -- hundreds of queued packets).
// Default: 3 batches ahead for the two-launch loops, 1 with the interpenetration term; SFX_POLL_AHEAD overrides.  The price of
// depth: decisions (retirement, admission, compaction, the end) lag by ahead x 8 rounds; finished frames only ever stay
// finished, and the surplus rounds at the end run on empty launches (0.1 % of a 256-frame fit at depth 3).
#define SFX_POLL_BUFS 8
struct sfx_batch {
    sfx_model* m = nullptr;
    BatchDev D{};
    DevAlloc mem;
    VarList* vl_dev = nullptr;    // [2]: camera, body
    StageW* sw_dev = nullptr;     // [n_stages]
    VarList vl_host[2];
    int* stage_host = nullptr;    // pinned, [SFX_POLL_BUFS][B]
    int* map_host = nullptr;      // pinned, [SFX_POLL_BUFS][3B]: slot[], running-frame list, newly admitted frames (one upload per batch in flight)
    int* act_dev = nullptr;       // [B] running frames of the fused dense loop
    int* act_new_dev = nullptr;   // [B] frames admitted at the latest poll (export-only launch)
    int slots = 0;                // GEMM columns of the fused dense loop (0: one per frame)
    hipEvent_t poll_ev[SFX_POLL_BUFS] = {};
    std::vector<int> slot_host;
    int K = 0;
    sfx_pen* pen = nullptr;       // interpenetration operator (cfg.interpenetration)
    float pen_sigma = 0.f; int pen_outside = 1;
    int pen_cols = 0, pen_cap = 0, pen_gen = -1;       // what b->pen was made for (returned to the model's idle slot on destroy)
    // the interpenetration step of a round of the fused loop as ONE captured graph per column count (twelve kernels; the count
    // only changes when the columns are compacted -- a handful of times per fit): one API call per round instead of twelve
    std::map<int, hipGraphExec_t> pen_graphs;
    int pen_graph_nodes = 0;       // kernel nodes of the most recently captured interpenetration step (sfx_batch_pen_launches)
    hipStream_t cap_stream = nullptr;   // capture happens on a stream of its own (the caller's may be the legacy NULL stream, which cannot capture)
    int* pen_stats_all = nullptr; // [B][stride] diagnostics of a CHUNKED evaluation (stand-alone call on a pooled batch), else unused
    bool pen_chunked = false;     // the most recent evaluation was chunked: sfx_batch_pen_stats reads pen_stats_all
};

extern "C" int sfx_model_create(const sfx_model_desc* d, sfx_model** out) {
    if (!d || !out) { sfx_set_error("null argument"); return -1; }
    if (d->J != SFX_J) { sfx_set_error("only J=55 (SMPL-X) is supported, got %d", d->J); return -1; }
    const int V = d->V, S = d->num_betas + d->num_expr, P = 9 * (d->J - 1), KD = S + P;
    if (KD > SFX_KD_PAD) { sfx_set_error("blend-shape depth %d unsupported", KD); return -1; }
    if (d->K > SFX_MAX_K) { sfx_set_error("K=%d > %d", d->K, SFX_MAX_K); return -1; }
    int dev_count = 0;
    if (hipGetDeviceCount(&dev_count) != hipSuccess || dev_count == 0) {
        sfx_set_error("no HIP device: libsfx has no CPU fallback"); return -3;
    }
    sfx_model* m = new sfx_model();
    DevModel& M = m->M;
    m->NB = d->num_betas; m->NE = d->num_expr; m->NPCA = d->num_pca;
    M.V = V; M.F = d->F; M.S = S; M.P = P; M.KD = KD; M.K = d->K;
    M.n_extra = d->n_extra; M.n_lmk = d->n_lmk; M.n_dyn_rows = d->n_dyn_rows; M.n_dyn = d->n_dyn;
    M.Vpad = ((V + 31) / 32) * 32;

    std::vector<float> vt(d->v_template, d->v_template + (size_t)V * 3);
    M.v_template = m->mem.up(vt);
    // blend-shape matrix, k-major [KD_PAD][3*Vpad] (zero rows beyond KD) and vertex-major [V][3][KD_PAD]
    {
        const size_t LD = (size_t)3 * M.Vpad;
        std::vector<float> dirs((size_t)SFX_KD_PAD * LD, 0.f), dirsT((size_t)V * 3 * SFX_KD_PAD, 0.f);
        for (int v = 0; v < V; ++v)
            for (int c = 0; c < 3; ++c) {
                const float* sd = d->shapedirs + ((size_t)v * 3 + c) * S;
                const float* pd = d->posedirs + ((size_t)v * 3 + c) * P;
                float* row = &dirsT[((size_t)v * 3 + c) * SFX_KD_PAD];
                for (int k = 0; k < S; ++k) { row[k] = sd[k]; dirs[(size_t)k * LD + (size_t)v * 3 + c] = sd[k]; }
                for (int k = 0; k < P; ++k) { row[S + k] = pd[k]; dirs[(size_t)(S + k) * LD + (size_t)v * 3 + c] = pd[k]; }
            }
        M.dirs = m->mem.up(dirs);
        M.dirsT = m->mem.up(dirsT);
        {   // tile-major copy for the forward GEMM: [tile][k][16 vertices x 3 coordinates]
            const int ntile = M.Vpad / 16;
            std::vector<float> dt((size_t)ntile * SFX_KD_PAD * 48, 0.f);
            for (int tl = 0; tl < ntile; ++tl)
                for (int k = 0; k < SFX_KD_PAD; ++k)
                    std::memcpy(&dt[((size_t)tl * SFX_KD_PAD + k) * 48], &dirs[(size_t)k * LD + (size_t)tl * 48], 48 * sizeof(float));
            M.dirs_tiled = m->mem.up(dt);
        }
    }
    {
        std::vector<float> W(d->lbs_weights, d->lbs_weights + (size_t)V * SFX_J);
        std::vector<float> WT((size_t)SFX_JPAD * M.Vpad, 0.f);
        for (int v = 0; v < V; ++v)
            for (int j = 0; j < SFX_J; ++j) WT[(size_t)j * M.Vpad + v] = W[(size_t)v * SFX_J + j];
        std::vector<int> wj((size_t)V * SFX_NW, 0); std::vector<float> ww((size_t)V * SFX_NW, 0.f);
        for (int v = 0; v < V; ++v) {
            int n = 0;
            for (int j = 0; j < SFX_J; ++j) {
                const float w = W[(size_t)v * SFX_J + j];
                if (w == 0.f) continue;
                if (n < SFX_NW) { wj[(size_t)v * SFX_NW + n] = j; ww[(size_t)v * SFX_NW + n] = w; }
                ++n;
            }
            // a row with more nonzeros than the packed form holds is flagged: the needed-rows path
            // then reads the full row of lbs_weights for this vertex (same ascending joint order)
            if (n > SFX_NW) wj[(size_t)v * SFX_NW] = -1;
        }
        M.Wsp_j = m->mem.up(wj); M.Wsp_w = m->mem.up(ww);
        {   // transposed CSR: per joint the vertices it skins, ascending
            std::vector<int> js(SFX_J + 1, 0), jv; std::vector<float> jw;
            for (int j = 0; j < SFX_J; ++j) {
                for (int v = 0; v < V; ++v) { const float w = W[(size_t)v * SFX_J + j]; if (w != 0.f) { jv.push_back(v); jw.push_back(w); } }
                js[j + 1] = (int)jv.size();
            }
            M.jv_start = m->mem.up(js); M.jv_vid = m->mem.up(jv); M.jv_w = m->mem.up(jw);
        }
        M.W = m->mem.up(W);
        M.WT = m->mem.up(WT);
        // per 16-vertex tile: the joints with any nonzero weight (ascending), weights in MFMA B layout
        const int nt = M.Vpad / 16;
        std::vector<int> tn(nt, 0), tl((size_t)nt * SFX_JPAD, 0);
        std::vector<float> tw((size_t)nt * SFX_JPAD * 16, 0.f);
        for (int t = 0; t < nt; ++t) {
            int n = 0;
            for (int j = 0; j < SFX_J; ++j) {
                bool used = false;
                for (int q = 0; q < 16 && !used; ++q) { const int v = t * 16 + q; used = v < V && W[(size_t)v * SFX_J + j] != 0.f; }
                if (!used) continue;
                tl[(size_t)t * SFX_JPAD + n] = j;
                for (int q = 0; q < 16; ++q) { const int v = t * 16 + q; tw[((size_t)t * SFX_JPAD + n) * 16 + q] = v < V ? W[(size_t)v * SFX_J + j] : 0.f; }
                ++n;
            }
            tn[t] = ((n + 3) / 4) * 4;
        }
        M.tj_n = m->mem.up(tn); M.tj_list = m->mem.up(tl); M.tj_w = m->mem.up(tw);
    }
    // folded joint regressor: J = J_template + J_dirs . coeff   (J_regressor . v_shaped)
    {
        std::vector<float> Jt((size_t)SFX_J * 3), Jd((size_t)SFX_J * 3 * S);
        for (int j = 0; j < SFX_J; ++j) {
            std::vector<double> acc(3 + 3 * S, 0.0);
            const float* jr = d->J_regressor + (size_t)j * V;
            for (int v = 0; v < V; ++v) {
                const double w = jr[v];
                if (w == 0.0) continue;
                for (int c = 0; c < 3; ++c) {
                    acc[c] += w * d->v_template[(size_t)v * 3 + c];
                    const float* sd = d->shapedirs + ((size_t)v * 3 + c) * S;
                    for (int l = 0; l < S; ++l) acc[3 + c * S + l] += w * sd[l];
                }
            }
            for (int c = 0; c < 3; ++c) {
                Jt[j * 3 + c] = (float)acc[c];
                for (int l = 0; l < S; ++l) Jd[((size_t)j * 3 + c) * S + l] = (float)acc[3 + c * S + l];
            }
        }
        M.J_template = m->mem.up(Jt);
        M.J_dirs = m->mem.up(Jd);
    }
    // kinematic tree: depth levels and child lists
    {
        std::vector<int> par(d->parents, d->parents + SFX_J), depth(SFX_J, 0);
        par[0] = -1;
        int maxd = 0;
        for (int j = 1; j < SFX_J; ++j) {
            if (par[j] < 0 || par[j] >= j) { sfx_set_error("parents must be topologically ordered"); delete m; return -1; }
            depth[j] = depth[par[j]] + 1; maxd = std::max(maxd, depth[j]);
        }
        if (maxd + 1 > SFX_MAX_LEVELS) { sfx_set_error("tree too deep"); delete m; return -1; }
        M.n_levels = maxd + 1;
        std::vector<int> lj;
        for (int l = 0; l <= maxd; ++l) {
            M.level_start[l] = (int)lj.size();
            for (int j = 0; j < SFX_J; ++j) if (depth[j] == l) lj.push_back(j);
        }
        M.level_start[maxd + 1] = (int)lj.size();
        std::vector<int> cs(SFX_J + 1, 0), cl;
        for (int j = 0; j < SFX_J; ++j) {
            cs[j] = (int)cl.size();
            for (int c = j + 1; c < SFX_J; ++c) if (par[c] == j) cl.push_back(c);
        }
        cs[SFX_J] = (int)cl.size();
        for (int j = 0; j < SFX_J; ++j) { m->meta_host[MO_PAR + j] = par[j]; m->meta_host[MO_LJ + j] = lj[j]; }
        for (int j = 0; j <= SFX_J; ++j) m->meta_host[MO_CS + j] = cs[j];
        for (size_t q = 0; q < cl.size(); ++q) m->meta_host[MO_CL + q] = cl[q];
        // DFS pre-order (children in ascending joint order) and subtree sizes: the adjoint of the
        // chain sums over subtrees, which are contiguous pre-order ranges
        {
            std::vector<int> pre(SFX_J, 0), sub(SFX_J, 1), stack{0};
            int pos = 0;
            while (!stack.empty()) {
                const int j = stack.back(); stack.pop_back();
                pre[j] = pos++;
                for (int q = cs[j + 1] - 1; q >= cs[j]; --q) stack.push_back(cl[q]);
            }
            for (int j = SFX_J - 1; j > 0; --j) sub[par[j]] += sub[j];
            for (int j = 0; j < SFX_J; ++j) { m->meta_host[MO_PRE + j] = pre[j]; m->meta_host[MO_SUB + j] = sub[j]; }
        }
        // 2^k-th ancestors for the pointer-jumping evaluation of the chain
        {
            int rounds = 0;
            while ((1 << rounds) < M.n_levels) ++rounds;
            if (rounds > SFX_MAX_ROUNDS) { sfx_set_error("tree too deep"); delete m; return -1; }
            M.n_rounds = rounds;
            std::vector<int> anc(par);
            for (int k = 0; k < rounds; ++k) {
                for (int j = 0; j < SFX_J; ++j) m->meta_host[MO_ANC + k * 56 + j] = anc[j];
                std::vector<int> nxt(SFX_J);
                for (int j = 0; j < SFX_J; ++j) nxt[j] = anc[j] < 0 ? -1 : anc[anc[j]];
                anc = nxt;
            }
        }
        M.parents = m->mem.up(par); M.level_joints = m->mem.up(lj);
        M.child_start = m->mem.up(cs); M.child_list = m->mem.up(cl);
    }
    {
        std::vector<float> cl(d->hands_comp_l, d->hands_comp_l + (size_t)d->num_pca * SFX_NHAND);
        std::vector<float> cr(d->hands_comp_r, d->hands_comp_r + (size_t)d->num_pca * SFX_NHAND);
        std::vector<float> pm(d->pose_mean, d->pose_mean + SFX_POSE);
        M.comp_l = m->mem.up(cl); M.comp_r = m->mem.up(cr); M.pose_mean = m->mem.up(pm);
    }
    {
        std::vector<int> faces(d->faces, d->faces + (size_t)d->F * 3);
        M.faces = m->mem.up(faces);
        m->faces_host = faces;
        std::vector<int> df; std::vector<float> db;
        if (d->n_dyn > 0) {
            df.assign(d->dyn_lmk_faces_idx, d->dyn_lmk_faces_idx + (size_t)d->n_dyn_rows * d->n_dyn);
            db.assign(d->dyn_lmk_bary, d->dyn_lmk_bary + (size_t)d->n_dyn_rows * d->n_dyn * 3);
        }
        M.dyn_faces = m->mem.up(df); M.dyn_bary = m->mem.up(db);
    }
    // mapped joints -> kinematic joints / vertex items
    {
        const int K = d->K;
        std::vector<int> jt(K), js(K, 0), ji0(K, 0), jn(K, 0), ivid, idyn, ik;
        std::vector<float> iw;
        std::vector<std::vector<int>> readers(SFX_J);
        const int e0 = SFX_J, l0 = e0 + d->n_extra, d0 = l0 + d->n_lmk, end = d0 + d->n_dyn;
        for (int k = 0; k < K; ++k) {
            const int s = d->joint_map[k];
            if (s < 0 || s >= end) { sfx_set_error("joint_map[%d]=%d out of range [0,%d)", k, s, end); delete m; return -1; }
            if (s < e0) { jt[k] = 0; js[k] = s; readers[s].push_back(k); continue; }
            jt[k] = 1; ji0[k] = (int)ivid.size();
            if (s < l0) { ivid.push_back(d->extra_vertex_ids[s - e0]); iw.push_back(1.f); idyn.push_back(-1); ik.push_back(k); jn[k] = 1; }
            else if (s < d0) {
                const int l = s - l0, f = d->lmk_faces_idx[l];
                for (int c = 0; c < 3; ++c) { ivid.push_back(d->faces[(size_t)f * 3 + c]); iw.push_back(d->lmk_bary[l * 3 + c]);
                                              idyn.push_back(-1); ik.push_back(k); }
                jn[k] = 3;
            } else {
                const int l = s - d0;
                for (int c = 0; c < 3; ++c) { ivid.push_back(-1); iw.push_back(0.f); idyn.push_back(l * 3 + c); ik.push_back(k); }
                jn[k] = 3;
            }
        }
        if ((int)ivid.size() > SFX_MAX_ITEMS) { sfx_set_error("too many vertex items"); delete m; return -1; }
        M.n_items = (int)ivid.size();
        std::vector<int> sk0(SFX_J + 1, 0), skl;
        for (int s = 0; s < SFX_J; ++s) { sk0[s] = (int)skl.size(); for (int k : readers[s]) skl.push_back(k); }
        sk0[SFX_J] = (int)skl.size();
        for (int s2 = 0; s2 <= SFX_J; ++s2) m->meta_host[MO_SK0 + s2] = sk0[s2];
        for (size_t q = 0; q < skl.size(); ++q) m->meta_host[MO_SKL + q] = skl[q];
        for (int k = 0; k < K; ++k) { m->meta_host[MO_JT + k] = jt[k]; m->meta_host[MO_JS + k] = js[k];
                                      m->meta_host[MO_JI0 + k] = ji0[k]; m->meta_host[MO_JN + k] = jn[k]; }
        for (size_t q = 0; q < ik.size(); ++q) m->meta_host[MO_IK + q] = ik[q];
        std::vector<int> dyn_pv;        // [rows][nd] vertices of the dynamic-contour items (filled below; the export table needs them)
        // by-joint adjoint lists
        {
            auto build = [&](const std::vector<int>& vids, const std::vector<int>& items, std::vector<int>& start,
                             std::vector<int>& it, std::vector<float>& wv, int base) {
                for (int j = 0; j < SFX_J; ++j) {
                    start.push_back(base + (int)it.size());
                    for (size_t q = 0; q < items.size(); ++q) {
                        const float w = d->lbs_weights[(size_t)vids[q] * SFX_J + j];
                        if (w != 0.f) { it.push_back(items[q]); wv.push_back(w); }
                    }
                }
                start.push_back(base + (int)it.size());
            };
            std::vector<int> svid, sitem, dynitems;
            for (int i = 0; i < (int)ivid.size(); ++i) { if (idyn[i] < 0) { svid.push_back(ivid[i]); sitem.push_back(i); } else dynitems.push_back(i); }
            std::vector<int> ss, si; std::vector<float> sw2;
            build(svid, sitem, ss, si, sw2, 0);
            M.sj_start = m->mem.up(ss); M.sj_item = m->mem.up(si); M.sj_w = m->mem.up(sw2); M.n_sj = (int)si.size();
            std::vector<int> ds, di; std::vector<float> dw;
            M.n_dyn_items = (int)dynitems.size();
            for (int row = 0; row < d->n_dyn_rows && !dynitems.empty(); ++row) {
                std::vector<int> vids;
                for (int i : dynitems) {
                    const int l = idyn[i] / 3, c = idyn[i] % 3;
                    const int f = d->dyn_lmk_faces_idx[(size_t)row * d->n_dyn + l];
                    vids.push_back(d->faces[(size_t)f * 3 + c]);
                }
                std::vector<int> st;
                build(vids, dynitems, st, di, dw, 0);
                // offsets are absolute into di/dw: rebuild with the running base
                ds.insert(ds.end(), st.begin(), st.end());
            }
            // 'build' used base 0 relative to the current size of di at call time -> already absolute
            M.dj_start = m->mem.up(ds); M.dj_item = m->mem.up(di); M.dj_w = m->mem.up(dw);
            // the same per LUT row in fixed-size blocks, together with the row's vertices, barycentric weights, template
            // rows and sparse skinning weights (closure_body fetches one block asynchronously)
            const int nd = (int)dynitems.size(), rows = nd ? d->n_dyn_rows : 0;
            if (nd > SFX_MAX_DYN) { sfx_set_error("too many dynamic-contour items (%d > %d)", nd, SFX_MAX_DYN); delete m; return -1; }
            std::vector<int> pv((size_t)rows * nd), pwj((size_t)rows * nd * SFX_NW, 0), pjs((size_t)rows * (SFX_J + 1), 0), pji((size_t)rows * nd * SFX_NW, 0);
            std::vector<float> pw((size_t)rows * nd), pvt((size_t)rows * nd * 3), pww((size_t)rows * nd * SFX_NW, 0.f), pjw((size_t)rows * nd * SFX_NW, 0.f);
            for (int row = 0; row < rows; ++row) {
                for (int q = 0; q < nd; ++q) {
                    const int i = dynitems[q], l = idyn[i] / 3, c = idyn[i] % 3;
                    const int f = d->dyn_lmk_faces_idx[(size_t)row * d->n_dyn + l];
                    const int v = d->faces[(size_t)f * 3 + c];
                    const size_t o = (size_t)row * nd + q;
                    pv[o] = v; pw[o] = d->dyn_lmk_bary[((size_t)row * d->n_dyn + l) * 3 + c];
                    for (int e = 0; e < 3; ++e) pvt[o * 3 + e] = d->v_template[(size_t)v * 3 + e];
                    int n = 0;
                    for (int j = 0; j < SFX_J; ++j) {
                        const float w = d->lbs_weights[(size_t)v * SFX_J + j];
                        if (w == 0.f) continue;
                        if (n < SFX_NW) { pwj[o * SFX_NW + n] = j; pww[o * SFX_NW + n] = w; }
                        ++n;
                    }
                    if (n > SFX_NW) pwj[o * SFX_NW] = -1;
                }
                const int* st = &ds[(size_t)row * (SFX_J + 1)];
                const int n_row = st[SFX_J] - st[0];
                if (n_row > nd * SFX_NW) {      // (> SFX_NW weights per vertex on average: the closure falls back to dj_*)
                    pjs.clear(); break; }
                for (int j = 0; j <= SFX_J; ++j) pjs[(size_t)row * (SFX_J + 1) + j] = st[j] - st[0];
                for (int q = 0; q < n_row; ++q) { pji[(size_t)row * nd * SFX_NW + q] = di[st[0] + q]; pjw[(size_t)row * nd * SFX_NW + q] = dw[st[0] + q]; }
            }
            dyn_pv = pv;
            M.dynp_vid = m->mem.up(pv); M.dynp_w = m->mem.up(pw); M.dynp_vt = m->mem.up(pvt);
            M.dynp_wj = m->mem.up(pwj); M.dynp_ww = m->mem.up(pww);
            M.dynp_js = pjs.empty() ? nullptr : m->mem.up(pjs);
            M.dynp_ji = m->mem.up(pji); M.dynp_jw = m->mem.up(pjw);
        }
        M.jk_type = m->mem.up(jt); M.jk_src = m->mem.up(js); M.jk_item0 = m->mem.up(ji0); M.jk_nitem = m->mem.up(jn);
        M.item_vid = m->mem.up(ivid); M.item_w = m->mem.up(iw); M.item_dyn = m->mem.up(idyn); M.item_k = m->mem.up(ik);
        {   // template rows and packed skinning weights of the items, gathered per item: the per-frame kernels fetch them at
            // entry in ONE round trip (through item_vid it took two, in every launch of the tick kernel)
            const size_t ni = ivid.size();
            std::vector<float> svt(ni * 3, 0.f), sww(ni * SFX_NW, 0.f); std::vector<int> swj(ni * SFX_NW, 0);
            for (size_t i = 0; i < ni; ++i) {
                const int v = ivid[i];
                if (v < 0) continue;
                for (int c = 0; c < 3; ++c) svt[i * 3 + c] = d->v_template[(size_t)v * 3 + c];
                int n = 0;
                for (int j = 0; j < SFX_J; ++j) {
                    const float w = d->lbs_weights[(size_t)v * SFX_J + j];
                    if (w == 0.f) continue;
                    if (n < SFX_NW) { swj[i * SFX_NW + n] = j; sww[i * SFX_NW + n] = w; }
                    ++n;
                }
                if (n > SFX_NW) swj[i * SFX_NW] = -1;      // (same flag as Wsp_j: the full row of lbs_weights is read instead)
            }
            M.item_vt = m->mem.up(svt); M.item_wj = m->mem.up(swj); M.item_ww = m->mem.up(sww);
        }
        {   // distinct vertices of the static items -- and of every vertex a dynamic-contour item can land on (all LUT rows):
            // the dense GEMM hands their blend offsets to the per-frame kernel, which then streams no blend-shape row forward
            std::vector<int> vslot(M.Vpad, -1), uslot(ivid.size(), -1);
            int nu = 0, nstat = 0;
            for (size_t i = 0; i < ivid.size(); ++i) {
                if (idyn[i] >= 0) continue;
                ++nstat;
                if (vslot[ivid[i]] < 0) vslot[ivid[i]] = nu++;
                uslot[i] = vslot[ivid[i]];
            }
            for (size_t i = 0; i + 1 < ivid.size(); ++i)
                if (idyn[i] >= 0 && idyn[i + 1] < 0) { sfx_set_error("internal: dynamic items must trail the static ones"); delete m; return -1; }
            std::vector<int> pus(dyn_pv.size(), -1);
            for (size_t o = 0; o < dyn_pv.size(); ++o) {
                const int v = dyn_pv[o];
                if (vslot[v] < 0) vslot[v] = nu++;
                pus[o] = vslot[v];
            }
            M.dynp_us = pus.empty() ? nullptr : m->mem.up(pus);
            M.n_uniq = nu; M.n_static_items = nstat;
            M.vslot = m->mem.up(vslot); M.item_uslot = m->mem.up(uslot);
        }
        M.src_k0 = m->mem.up(sk0); M.src_klist = m->mem.up(skl);
    }
    if (m->meta_host.size() != SFX_META_N) { sfx_set_error("internal: meta table"); delete m; return -1; }
    M.meta = m->mem.up(m->meta_host);
    if (m->mem.failed) { (void)hipGetLastError(); sfx_set_error("out of device memory (model constants)"); m->mem.free_all(); delete m; return -2; }
    if (hipDeviceSynchronize() != hipSuccess) { sfx_set_error("model upload failed"); m->mem.free_all(); delete m; return -2; }
    *out = m;
    return 0;
}

extern "C" void sfx_batch_destroy(sfx_batch* b);
extern "C" void sfx_model_destroy(sfx_model* m) {
    if (!m) return;
    if (m->fwd) sfx_batch_destroy(m->fwd);
    if (m->pen_idle) sfx_pen_destroy(m->pen_idle);
    m->mem.free_all();
    delete m;
}

extern "C" int sfx_model_set_parts(sfx_model* m, const int32_t* segm, const int32_t* parents, const int32_t* ign_pairs,
                                   int32_t n_ign) {
    if (!m) { sfx_set_error("null argument"); return -1; }
    { std::lock_guard<std::mutex> lk(m->pen_mu); ++m->parts_gen; if (m->pen_idle) { sfx_pen_destroy(m->pen_idle); m->pen_idle = nullptr; } }
    if (!segm && !parents) {        // no FilterFaces module (fit_single_frame.py:316 without part_segm_fn): no pair is filtered by part
        m->segm_host.clear(); m->parents_host.clear(); m->ign_host.clear();
        return 0;
    }
    if (!segm || !parents) { sfx_set_error("null argument"); return -1; }
    const size_t F = m->faces_host.size() / 3;
    m->segm_host.assign(segm, segm + F);
    m->parents_host.assign(parents, parents + F);
    m->ign_host.clear();
    if (ign_pairs && n_ign > 0) m->ign_host.assign(ign_pairs, ign_pairs + (size_t)2 * n_ign);
    return 0;
}

extern "C" int sfx_model_set_vposer(sfx_model* m, int32_t latent, int32_t hidden, const float* w1, const float* b1,
                                    const float* w2, const float* b2, const float* w3, const float* b3) {
    if (!m) { sfx_set_error("null model"); return -1; }
    if (hidden != 512 || latent < 4 || latent > 60 || latent % 4) {
        sfx_set_error("VPoser v1 decoder expected (hidden 512, latent a multiple of 4 <= 60), got %d/%d", latent, hidden); return -1; }
    auto v = [](const float* p, size_t n) { return std::vector<float>(p, p + n); };
    const int H = hidden, L = latent;
    std::vector<float> w1T((size_t)L * H), w2T((size_t)H * H), w3T((size_t)H * 128, 0.f);
    for (int o = 0; o < H; ++o) for (int i = 0; i < L; ++i) w1T[(size_t)i * H + o] = w1[(size_t)o * L + i];
    for (int o = 0; o < H; ++o) for (int i = 0; i < H; ++i) w2T[(size_t)i * H + o] = w2[(size_t)o * H + i];
    for (int o = 0; o < 126; ++o) for (int i = 0; i < H; ++i) w3T[(size_t)i * 128 + o] = w3[(size_t)o * H + i];
    m->M.vp_latent = latent; m->M.vp_hidden = hidden;
    m->M.vp_w1 = m->mem.up(v(w1, (size_t)H * L)); m->M.vp_b1 = m->mem.up(v(b1, H));
    m->M.vp_w2 = m->mem.up(v(w2, (size_t)H * H)); m->M.vp_b2 = m->mem.up(v(b2, H));
    m->M.vp_w3 = m->mem.up(v(w3, (size_t)126 * H)); m->M.vp_b3 = m->mem.up(v(b3, 126));
    m->M.vp_w1T = m->mem.up(w1T); m->M.vp_w2T = m->mem.up(w2T); m->M.vp_w3T = m->mem.up(w3T);
    if (m->mem.failed) { (void)hipGetLastError(); m->M.vp_latent = 0; sfx_set_error("out of device memory (VPoser weights)"); return -2; }
    if (m->fwd) { sfx_batch_destroy(m->fwd); m->fwd = nullptr; }
    return 0;
}

// ---------------------------------------------------------------------------------------
static void build_layout(ParLayout& L, int NB, int NE, int NPCA, int use_vposer, int latent) {
    L.NB = NB; L.NE = NE; L.NPCA = NPCA;
    L.cam_t = 0; L.go = 3; L.betas = 6; L.lh = L.betas + NB; L.rh = L.lh + NPCA; L.expr = L.rh + NPCA;
    L.jaw = L.expr + NE; L.leye = L.jaw + 3; L.reye = L.leye + 3; L.bodyp = L.reye + 3;
    L.has_bodyp = use_vposer ? 0 : 1;
    L.emb = L.bodyp + (L.has_bodyp ? 63 : 0);
    L.NEMB = use_vposer ? latent : 63;
    L.npar = L.emb + L.NEMB;
}

static void add_group(VarList& v, int off, int len, int has) {
    const int g = v.ngroups++;
    v.g_off[g] = (short)v.n; v.g_len[g] = (short)len; v.g_has[g] = (short)has;
    for (int i = 0; i < len; ++i, ++v.n) if (v.n < SFX_NVAR_MAX) v.idx[v.n] = (short)(off + i);     // (n > NVAR_MAX: rejected by the caller)
}

extern "C" int sfx_batch_create(sfx_model* m, const sfx_batch_cfg* c, const sfx_stage_weights* st, sfx_batch** out) {
    if (!m || !c || !out) { sfx_set_error("null argument"); return -1; }
    if (c->n_stages < 0 || c->n_stages > SFX_MAX_STAGES) { sfx_set_error("n_stages=%d unsupported", c->n_stages); return -1; }
    if (c->use_vposer && m->M.vp_latent == 0) { sfx_set_error("use_vposer without sfx_model_set_vposer"); return -1; }
    sfx_batch* b = new sfx_batch();
    b->m = m; b->K = m->M.K;
    BatchDev& D = b->D;
    const int B = c->B, K = m->M.K;
    D.cfg.B = B; D.cfg.n_stages = c->n_stages; D.cfg.use_vposer = c->use_vposer; D.cfg.use_hands = c->use_hands;
    D.cfg.use_face = c->use_face; D.cfg.use_conf = c->use_joints_conf; D.cfg.has_reg = c->has_regression_pose;
    D.cfg.use_conf_cam = c->use_conf_cam_init; D.cfg.nbj = c->num_body_joints; D.cfg.maxiters = c->maxiters;
    D.cfg.lbfgs_max_iter = c->lbfgs_max_iter > 0 ? c->lbfgs_max_iter : c->maxiters;
    D.cfg.max_eval = D.cfg.lbfgs_max_iter * 5 / 4; D.cfg.ftol = c->ftol; D.cfg.gtol = c->gtol;
    D.cfg.lr = c->lr; D.cfg.rho = c->rho; D.cfg.depth_w = c->depth_loss_weight; D.cfg.lbs_mode = c->lbs_mode;
    D.cfg.reuse = c->reuse_entry_eval;
    D.cfg.side_thsh = c->side_view_thsh; D.cfg.lsh = c->left_shoulder_idx; D.cfg.rsh = c->right_shoulder_idx;
    D.cfg.pen = c->interpenetration ? 1 : 0;
    D.cfg.proj64 = c->high_precision ? 1 : 0;
    // negative = the reference's default; 0 is a legal value of lbfgs_ls.LBFGS (it disables the test) and reaches the device as 0
    D.cfg.tol_grad = c->lbfgs_tolerance_grad >= 0 ? c->lbfgs_tolerance_grad : 1e-5;
    D.cfg.tol_change = c->lbfgs_tolerance_change >= 0 ? c->lbfgs_tolerance_change : 1e-9;
    if (c->lbfgs_max_eval > 0) D.cfg.max_eval = c->lbfgs_max_eval;
    // (round 5: a history_size beyond the default's 100 gets a ring of that many slots; the bound is the LDS array of the alphas)
    if (c->lbfgs_history_size > SFX_HIST_MAX) { sfx_set_error("history_size %d > %d", c->lbfgs_history_size, SFX_HIST_MAX); delete b; return -1; }
    D.cfg.hist_cap = c->lbfgs_history_size > 0 ? c->lbfgs_history_size : SFX_HIST;
    D.cfg.hist_ring = std::max(D.cfg.hist_cap, SFX_HIST);
    if (D.cfg.pen && c->lbs_mode != 1) {
        sfx_set_error("interpenetration needs lbs_mode = 1 (the term reads every vertex)"); delete b; return -1; }
    if (D.cfg.pen && !(c->df_cone_height > 0.f)) { sfx_set_error("df_cone_height must be positive"); delete b; return -1; }
    if (c->side_view_thsh > 0.f && (c->left_shoulder_idx < 0 || c->left_shoulder_idx >= K || c->right_shoulder_idx < 0 || c->right_shoulder_idx >= K)) {
        sfx_set_error("shoulder indices out of range"); delete b; return -1; }
    {   // keypoints (and their vertex items: ascending in keypoint order) that are live while the hand / face joint weights are zero
        const int kl[3] = {std::min(K, c->num_body_joints), std::min(K, c->num_body_joints + 42), K};
        for (int q = 0; q < 3; ++q) {
            int n = 0;
            for (int i = 0; i < m->M.n_items; ++i) if (m->meta_host[MO_IK + i] < kl[q]) ++n;
            D.cfg.kl[q] = kl[q]; D.cfg.nil[q] = n;
        }
    }
    build_layout(D.L, m->NB, m->NE, m->NPCA, c->use_vposer, m->M.vp_latent);
    if (D.L.npar > SFX_NPAR_MAX) { sfx_set_error("parameter block too large"); delete b; return -1; }
    const ParLayout& L = D.L;
    VarList cam{}, body{};
    add_group(cam, L.cam_t, 3, 1); add_group(cam, L.go, 3, 1);
    // order of smplx.SMPLX.parameters() then pose_embedding (fit_single_frame.py:554-559)
    add_group(body, L.betas, L.NB, 1); add_group(body, L.go, 3, 1);
    // the dead body_pose parameter (zero gradient, never moves: SURVEY 7 quirk) takes 63 slots of the optimiser's vectors; with
    // all 45 + 45 hand variables (use_pca=False) the list would not fit, so it is left out there -- entries that are identically
    // zero in x, g, d, s and y contribute nothing to any dot product or norm of LBFGS.step / _strong_Wolfe
    const int n_live = L.NB + 3 + 2 * L.NPCA + 9 + L.NE + L.NEMB;
    if (L.has_bodyp && n_live + 63 <= SFX_NVAR_MAX) add_group(body, L.bodyp, 63, 0);
    add_group(body, L.lh, L.NPCA, 1); add_group(body, L.rh, L.NPCA, 1);
    add_group(body, L.jaw, 3, 1); add_group(body, L.leye, 3, 1); add_group(body, L.reye, 3, 1);
    add_group(body, L.expr, L.NE, 1); add_group(body, L.emb, L.NEMB, 1);
    if (body.n > SFX_NVAR_MAX) {
        // e.g. all 45 hand components: more optimisation variables than the optimiser's vectors hold.  Such a batch can
        // still evaluate the forward (sfx_lbs_forward / sfx_batch_forward); n_stages = 0 declares that intent
        if (c->n_stages > 0) { sfx_set_error("%d optimisation variables (limit %d): reduce num_pca_comps", body.n, SFX_NVAR_MAX); delete b; return -1; }
        body.n = SFX_NVAR_MAX;
    }
    b->vl_host[0] = cam; b->vl_host[1] = body;
    std::vector<VarList> vls = {cam, body};
    b->vl_dev = b->mem.up(vls);
    std::vector<StageW> sws(std::max(1, c->n_stages));
    for (int i = 0; i < c->n_stages; ++i) {
        StageW& w = sws[i];
        w.bpw = st[i].body_pose_weight; w.sw = st[i].shape_weight;
        // fit_single_frame.py:567-568: bending = 3.17 * body_pose_weight (fp32 product) unless given
        w.bend = (st[i].bending_prior_weight >= 0.f) ? st[i].bending_prior_weight : 3.17f * st[i].body_pose_weight;
        w.hpw = st[i].hand_prior_weight; w.epw = st[i].expr_prior_weight;
        for (int q = 0; q < 3; ++q) w.jaw[q] = st[i].jaw_prior_weight[q];
        w.hand_jw = st[i].hand_joint_weight; w.face_jw = st[i].face_joint_weight;
        w.coll = D.cfg.pen ? st[i].coll_loss_weight : 0.f;
    }
    b->sw_dev = b->mem.up(sws);
    D.Bpad = ((B + 127) / 128) * 128;
    D.X = b->mem.zeros<float>((size_t)B * SFX_NPAR_MAX);
    D.Xt = b->mem.zeros<float>((size_t)B * SFX_NPAR_MAX);
    D.gt = b->mem.zeros<float>((size_t)B * K * 2);
    D.conf = b->mem.zeros<float>((size_t)B * K);
    D.jw = b->mem.zeros<float>((size_t)B * K);
    D.cmask = b->mem.zeros<float>((size_t)B * K);
    D.cam = b->mem.zeros<float>((size_t)B * 8);
    D.camR = b->mem.zeros<float>((size_t)B * 9);
    D.regpose = b->mem.zeros<float>((size_t)B * 63);
    D.fd = b->mem.zeros<float>((size_t)B * FD_N);
    D.f = b->mem.zeros<float>(B);
    D.g = b->mem.zeros<float>((size_t)B * SFX_NVAR_MAX);
    D.bodypose = b->mem.zeros<float>((size_t)B * 63);
    D.featR = b->mem.zeros<float>((size_t)SFX_KD_PAD * D.Bpad);
    D.AT = b->mem.zeros<float>((size_t)12 * SFX_JPAD * D.Bpad);
    D.verts = b->mem.zeros<float>((size_t)B * m->M.V * 3);
    D.fwd = b->mem.zeros<float>((size_t)B * SFX_FWD_N);
    D.uvp = b->mem.zeros<float>((size_t)B * std::max(1, m->M.n_uniq) * 3);
    if (D.cfg.pen) {
        const int F = (int)(m->faces_host.size() / 3);
        const bool parts = !m->segm_host.empty();
        // collision buffers (partner lists, grid entries, pair list: ~45 MB per mesh at max_collisions 128) are indexed by
        // GEMM column, not by frame: a job that runs B frames through a pool of `slots` columns holds `slots` of them
        const int pen_cols = (c->slots > 0 && c->slots < B && c->lbs_mode == 1) ? std::min(B, ((c->slots + 31) / 32) * 32) : B;
        const int pen_cap = std::max(1, c->max_collisions);
        {
            std::lock_guard<std::mutex> lk(m->pen_mu);
            if (m->pen_idle && m->pen_idle_cols >= pen_cols && m->pen_idle_cap == pen_cap && m->pen_idle_gen == m->parts_gen) {
                b->pen = m->pen_idle; b->pen_cols = m->pen_idle_cols;
                m->pen_idle = nullptr;
                sfx_pen_reuse(b->pen);
            } else if (m->pen_idle) { sfx_pen_destroy(m->pen_idle); m->pen_idle = nullptr; }      // too small / another table: make room first
        }
        if (!b->pen) {
            int rc = sfx_pen_create(m->M.V, F, m->faces_host.data(), parts ? m->segm_host.data() : nullptr,
                                    parts ? m->parents_host.data() : nullptr, m->ign_host.empty() ? nullptr : m->ign_host.data(),
                                    (int)(m->ign_host.size() / 2), pen_cap, pen_cols, &b->pen);
            if (rc) { b->mem.free_all(); delete b; return rc; }
            b->pen_cols = pen_cols;
        }
        b->pen_cap = pen_cap; b->pen_gen = m->parts_gen;
        b->pen_sigma = c->df_cone_height; b->pen_outside = c->penalize_outside ? 1 : 0;
        (void)sfx_pen_set_point2plane(b->pen, c->point2plane);
        D.pen_loss = b->mem.zeros<float>(B);
        D.pen_dverts = b->mem.zeros<float>((size_t)B * m->M.V * 3);
        D.ext_n = b->mem.zeros<int>(B);
        D.pen_want = b->mem.zeros<int>(B);
        D.pen_over = b->mem.zeros<int>(B);
        D.pen_flag = b->mem.zeros<int>(B);
        D.vposed = b->mem.zeros<float>((size_t)B * m->M.V * 3);
        D.adj_G = b->mem.zeros<float>((size_t)D.Bpad * 3 * m->M.Vpad);
        D.adj_part = b->mem.zeros<float>((size_t)sfx_adj_slices(m->M) * SFX_KD_PAD * D.Bpad);
        D.pen_dfeat = b->mem.zeros<float>((size_t)B * SFX_KD_PAD);
        D.pen_dA = b->mem.zeros<float>((size_t)B * SFX_J * 12);
        if (!D.adj_G || !D.adj_part || !D.vposed) { sfx_set_error("out of device memory"); b->mem.free_all(); delete b; return -2; }
    }
    D.joints = b->mem.zeros<float>((size_t)B * K * 3);
    D.fullpose = b->mem.zeros<float>((size_t)B * SFX_POSE);
    D.stage = b->mem.zeros<int>(B);
    { std::vector<int> id(B); for (int i = 0; i < B; ++i) id[i] = i; D.slot = b->mem.up(id); D.nact = B; }
    b->act_dev = b->mem.zeros<int>(B); b->act_new_dev = b->mem.zeros<int>(B);
    b->slots = (c->slots > 0 && c->slots < B && c->lbs_mode == 1) ? ((c->slots + 31) / 32) * 32 : 0;
    if (b->slots >= B) b->slots = 0;
    D.opt = b->mem.zeros<char>((size_t)B * sfx_optstate_size());
    D.vec = b->mem.zeros<float>((size_t)B * NVEC * SFX_NVAR_MAX);
    D.hist = b->mem.zeros<float>((size_t)B * 2 * (D.cfg.hist_ring + 8) * SFX_NVAR_MAX);
    D.n_active = b->mem.zeros<int>(4);
    D.stage_loss = b->mem.zeros<float>((size_t)B * (1 + SFX_MAX_STAGES));
    D.stage_evals = b->mem.zeros<int>((size_t)B * (1 + SFX_MAX_STAGES));
    D.stage_ref_evals = b->mem.zeros<int>((size_t)B * (1 + SFX_MAX_STAGES));
    D.X0 = b->mem.zeros<float>((size_t)B * SFX_NPAR_MAX);
    D.gocam = b->mem.zeros<float>((size_t)B * 4);
    D.stage_loss2 = b->mem.zeros<float>((size_t)B * (1 + SFX_MAX_STAGES));
    D.try_both = b->mem.zeros<int>(B);
    D.orient_pass = b->mem.zeros<int>(B);
    if (b->mem.failed || !D.hist || !D.verts) {
        (void)hipGetLastError();
        sfx_set_error("out of device memory (batch of %d frames)", B);
        if (b->pen) sfx_pen_destroy(b->pen);
        b->mem.free_all(); delete b; return -2; }
    if (hipHostMalloc((void**)&b->stage_host, (size_t)SFX_POLL_BUFS * B * sizeof(int)) != hipSuccess) b->stage_host = nullptr;      // poll buffers
    if (hipHostMalloc((void**)&b->map_host, (size_t)SFX_POLL_BUFS * 3 * B * sizeof(int)) != hipSuccess) b->map_host = nullptr;
    if (c->lbs_mode == 1 && (!b->stage_host || !b->map_host)) {       // the fused dense loop polls through pinned memory
        sfx_set_error("out of pinned host memory"); sfx_batch_destroy(b); return -2; }
    for (int i = 0; i < SFX_POLL_BUFS; ++i)
        if (hipEventCreateWithFlags(&b->poll_ev[i], hipEventDisableTiming) != hipSuccess) { sfx_set_error("event creation failed"); sfx_batch_destroy(b); return -2; }
    *out = b;
    return 0;
}

extern "C" void sfx_batch_destroy(sfx_batch* b) {
    if (!b) return;
    for (auto& kv : b->pen_graphs) hipGraphExecDestroy(kv.second);
    b->pen_graphs.clear();
    if (b->cap_stream) hipStreamDestroy(b->cap_stream);
    if (b->pen) {          // the collision buffers go back to the model's idle slot (see sfx_model)
        sfx_model* m = b->m;
        (void)hipDeviceSynchronize();
        std::lock_guard<std::mutex> lk(m->pen_mu);
        if (b->pen_gen == m->parts_gen && (!m->pen_idle || m->pen_idle_cols <= b->pen_cols)) {
            if (m->pen_idle) sfx_pen_destroy(m->pen_idle);
            m->pen_idle = b->pen; m->pen_idle_cols = b->pen_cols; m->pen_idle_cap = b->pen_cap; m->pen_idle_gen = b->pen_gen;
        } else sfx_pen_destroy(b->pen);
        b->pen = nullptr;
    }
    if (b->D.trace) hipFree(b->D.trace);
    if (b->D.trace_n) hipFree(b->D.trace_n);
    b->mem.free_all();
    if (b->stage_host) hipHostFree(b->stage_host);
    if (b->map_host) hipHostFree(b->map_host);
    for (int i = 0; i < SFX_POLL_BUFS; ++i) if (b->poll_ev[i]) hipEventDestroy(b->poll_ev[i]);
    delete b;
}

// gt | conf | jw | cmask | cam | camR | regpose of frame b -> its packed record D.fd[b] (what the closure workgroup
// loads in one 16-byte copy); launched after every host write to one of the seven arrays
__global__ void k_pack_fd(BatchDev D, int K) {
    const int b = blockIdx.x, t = threadIdx.x;
    float* fd = D.fd + (size_t)b * FD_N;
    for (int i = t; i < 2 * K; i += blockDim.x) fd[FD_GT + i] = D.gt[(size_t)b * K * 2 + i];
    for (int i = t; i < K; i += blockDim.x) {
        fd[FD_CONF + i] = D.conf[(size_t)b * K + i]; fd[FD_JW + i] = D.jw[(size_t)b * K + i]; fd[FD_CMASK + i] = D.cmask[(size_t)b * K + i]; }
    if (t < 8) fd[FD_CAM + t] = D.cam[(size_t)b * 8 + t];
    if (t < 9) fd[FD_CAMR + t] = D.camR[(size_t)b * 9 + t];
    if (t < 63) fd[FD_REG + t] = D.regpose[(size_t)b * 63 + t];
}
static void pack_fd(sfx_batch* b, hipStream_t s) {
    hipLaunchKernelGGL(k_pack_fd, dim3(b->D.cfg.B), dim3(128), 0, s, b->D, b->K);
}

extern "C" int sfx_batch_set_frames(sfx_batch* b, const float* kp, const float* jw, const float* cmask,
                                    const float* cam, const float* camR) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    const int B = b->D.cfg.B, K = b->K;
    if (kp) {
        std::vector<float> gt((size_t)B * K * 2), cf((size_t)B * K);
        for (size_t i = 0; i < (size_t)B * K; ++i) { gt[2 * i] = kp[3 * i]; gt[2 * i + 1] = kp[3 * i + 1]; cf[i] = kp[3 * i + 2]; }
        SFX_CHECK(hipMemcpy(b->D.gt, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));
        SFX_CHECK(hipMemcpy(b->D.conf, cf.data(), cf.size() * 4, hipMemcpyHostToDevice));
    }
    if (jw) SFX_CHECK(hipMemcpy(b->D.jw, jw, (size_t)B * K * 4, hipMemcpyHostToDevice));
    if (cmask) SFX_CHECK(hipMemcpy(b->D.cmask, cmask, (size_t)B * K * 4, hipMemcpyHostToDevice));
    if (cam) {
        std::vector<float> c8((size_t)B * 8, 0.f);
        for (int i = 0; i < B; ++i) for (int q = 0; q < 6; ++q) c8[(size_t)i * 8 + q] = cam[(size_t)i * 6 + q];
        SFX_CHECK(hipMemcpy(b->D.cam, c8.data(), c8.size() * 4, hipMemcpyHostToDevice));
    }
    if (camR) SFX_CHECK(hipMemcpy(b->D.camR, camR, (size_t)B * 9 * 4, hipMemcpyHostToDevice));
    pack_fd(b, 0);
    SFX_CHECK(hipDeviceSynchronize());
    return 0;
}

static void put(std::vector<float>& X, int B, int off, int n, const float* src) {
    if (!src) return;
    for (int i = 0; i < B; ++i) for (int q = 0; q < n; ++q) X[(size_t)i * SFX_NPAR_MAX + off + q] = src[(size_t)i * n + q];
}
static void take(const std::vector<float>& X, int B, int off, int n, float* dst) {
    if (!dst) return;
    for (int i = 0; i < B; ++i) for (int q = 0; q < n; ++q) dst[(size_t)i * n + q] = X[(size_t)i * SFX_NPAR_MAX + off + q];
}

extern "C" int sfx_batch_set_params(sfx_batch* b, const float* cam_t, const float* go, const float* betas,
                                    const float* lh, const float* rh, const float* expr, const float* jaw,
                                    const float* leye, const float* reye, const float* emb, const float* reg) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    const int B = b->D.cfg.B; const ParLayout& L = b->D.L;
    std::vector<float> X((size_t)B * SFX_NPAR_MAX);
    SFX_CHECK(hipMemcpy(X.data(), b->D.X, X.size() * 4, hipMemcpyDeviceToHost));
    put(X, B, L.cam_t, 3, cam_t); put(X, B, L.go, 3, go); put(X, B, L.betas, L.NB, betas);
    put(X, B, L.lh, L.NPCA, lh); put(X, B, L.rh, L.NPCA, rh); put(X, B, L.expr, L.NE, expr);
    put(X, B, L.jaw, 3, jaw); put(X, B, L.leye, 3, leye); put(X, B, L.reye, 3, reye);
    put(X, B, L.emb, L.NEMB, emb);
    // body_model.reset_params(body_pose=pose_embedding) also fills the (dead) body_pose parameter
    if (L.has_bodyp && emb) put(X, B, L.bodyp, 63, emb);
    SFX_CHECK(hipMemcpy(b->D.X, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    SFX_CHECK(hipMemcpy(b->D.Xt, X.data(), X.size() * 4, hipMemcpyHostToDevice));
    if (reg) {
        std::vector<float> r((size_t)B * 63, 0.f);
        for (int i = 0; i < B; ++i) for (int q = 0; q < L.NEMB; ++q) r[(size_t)i * 63 + q] = reg[(size_t)i * L.NEMB + q];
        SFX_CHECK(hipMemcpy(b->D.regpose, r.data(), r.size() * 4, hipMemcpyHostToDevice));
        pack_fd(b, 0);
        SFX_CHECK(hipDeviceSynchronize());
    }
    return 0;
}

extern "C" int sfx_batch_get_params(sfx_batch* b, float* cam_t, float* go, float* betas, float* lh, float* rh,
                                    float* expr, float* jaw, float* leye, float* reye, float* emb, float* body_pose) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    const int B = b->D.cfg.B; const ParLayout& L = b->D.L;
    std::vector<float> X((size_t)B * SFX_NPAR_MAX);
    SFX_CHECK(hipMemcpy(X.data(), b->D.X, X.size() * 4, hipMemcpyDeviceToHost));
    take(X, B, L.cam_t, 3, cam_t); take(X, B, L.go, 3, go); take(X, B, L.betas, L.NB, betas);
    take(X, B, L.lh, L.NPCA, lh); take(X, B, L.rh, L.NPCA, rh); take(X, B, L.expr, L.NE, expr);
    take(X, B, L.jaw, 3, jaw); take(X, B, L.leye, 3, leye); take(X, B, L.reye, 3, reye);
    take(X, B, L.emb, L.NEMB, emb);
    if (body_pose) {
        if (b->D.cfg.use_vposer) {      // decode the ACCEPTED latent (fit_single_frame.py:653-657)
            ClosureArgs a{}; a.stage_override = 0; a.forward_only = 1; a.from_X = 1;
            launch_closure(b->m->M, b->D, b->vl_dev, b->sw_dev, a, 0);
            SFX_CHECK(hipDeviceSynchronize());
            SFX_CHECK(hipMemcpy(body_pose, b->D.bodypose, (size_t)B * 63 * 4, hipMemcpyDeviceToHost));
        }
        else take(X, B, L.emb, 63, body_pose);
    }
    return 0;
}

#ifdef SFX_LAB       // include/sfx_lab.h
extern "C" int sfx_debug_phase_clocks(sfx_batch* b, int32_t stage, int64_t* out /* [32] */) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    long long* d = nullptr;
    SFX_CHECK(hipMalloc((void**)&d, 64 * sizeof(long long)));
    SFX_CHECK(hipMemset(d, 0, 64 * sizeof(long long)));
    { const long long freeze = 1 << 30; SFX_CHECK(hipMemcpy(d + 61, &freeze, sizeof(freeze), hipMemcpyHostToDevice)); }
    b->D.dbg = d;
    ClosureArgs a{}; a.stage_override = stage; a.from_X = 1;
    launch_closure(b->m->M, b->D, b->vl_dev, b->sw_dev, a, 0);
    launch_closure(b->m->M, b->D, b->vl_dev, b->sw_dev, a, 0);      // second launch: warm caches
    SFX_CHECK(hipDeviceSynchronize());
    b->D.dbg = nullptr;
    long long h[64];
    SFX_CHECK(hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(d);
    for (int i = 0; i < 32; ++i) out[i] = h[i];
    return 0;
}
#endif

// debug / tests: copy one of the dense path's device buffers of the most recent evaluation to the host.
// name -> floats per frame (rows are GEMM columns = frames while no compaction is in force):
//   "verts" V*3, "vposed" V*3, "pen_dverts" V*3, "pen_dfeat" 512, "pen_dA" 55*12, "pen_loss" 1,
//   "A" 12*55 (skinning transforms, [row*4+col][joint] gathered from the GEMM operand), "feat" 512
extern "C" int sfx_batch_debug_read(sfx_batch* b, const char* name, float* out, int64_t n_out) {
    if (!b || !name || !out) { sfx_set_error("null argument"); return -1; }
    const BatchDev& D = b->D; const int B = D.cfg.B, V = b->m->M.V;
    const std::string k(name);
    SFX_CHECK(hipDeviceSynchronize());
    auto plain = [&](const float* src, size_t per) -> int {
        if (!src) { sfx_set_error("buffer '%s' is not allocated in this batch", name); return -1; }
        if ((int64_t)(per * B) != n_out) { sfx_set_error("'%s' holds %zu floats, caller asked for %lld", name, per * B, (long long)n_out); return -1; }
        SFX_CHECK(hipMemcpy(out, src, per * B * sizeof(float), hipMemcpyDeviceToHost));
        return 0;
    };
    if (k == "verts") return plain(D.verts, (size_t)V * 3);
    if (k == "vposed") return plain(D.vposed, (size_t)V * 3);
    if (k == "pen_dverts") return plain(D.pen_dverts, (size_t)V * 3);
    if (k == "pen_dfeat") return plain(D.pen_dfeat, SFX_KD_PAD);
    if (k == "pen_dA") return plain(D.pen_dA, (size_t)SFX_J * 12);
    if (k == "pen_loss") return plain(D.pen_loss, 1);
    if (k == "A" || k == "feat") {
        const size_t rows = k == "A" ? (size_t)12 * SFX_JPAD : SFX_KD_PAD, per = k == "A" ? (size_t)12 * SFX_J : SFX_KD_PAD;
        if ((int64_t)(per * B) != n_out) { sfx_set_error("'%s' holds %zu floats, caller asked for %lld", name, per * B, (long long)n_out); return -1; }
        std::vector<float> h(rows * D.Bpad);
        SFX_CHECK(hipMemcpy(h.data(), k == "A" ? D.AT : D.featR, h.size() * sizeof(float), hipMemcpyDeviceToHost));
        for (int i = 0; i < B; ++i) {
            if (k == "A") { for (int e = 0; e < 12; ++e) for (int j = 0; j < SFX_J; ++j) out[((size_t)i * 12 + e) * SFX_J + j] = h[((size_t)e * SFX_JPAD + j) * D.Bpad + i]; }
            else for (int q = 0; q < SFX_KD_PAD; ++q) out[(size_t)i * SFX_KD_PAD + q] = h[(size_t)i * SFX_KD_PAD + q];
        }
        return 0;
    }
    sfx_set_error("unknown buffer '%s'", name);
    return -1;
}

// Stand-alone operator: the direction the device's blocked two-loop recursion (lbfgs_body.h lb_two_loop) computes from a history of `count`
// curvature pairs pushed in order (rows of SFX_NVAR_MAX = 192 floats, zero padded; the window keeps the last history_size,
// <= 0: 100) and a gradient g: d = -H g with H_diag = y.s / y.y of the last pair (lbfgs_ls.py:312-341).  Host pointers.
extern "C" int sfx_lbfgs_two_loop(const float* S, const float* Y, int32_t count, int32_t history_size, const float* g, float* d_out) {
    if (!S || !Y || !g || !d_out || count < 1 || history_size > SFX_HIST_MAX) { sfx_set_error("sfx_lbfgs_two_loop: bad arguments"); return -1; }
    const int rc = debug_two_loop(S, Y, count, history_size > 0 ? history_size : SFX_HIST, g, d_out);
    if (rc) { sfx_set_error("sfx_lbfgs_two_loop: HIP error"); return -1; }
    return 0;
}

#ifdef SFX_LAB       // include/sfx_lab.h
extern "C" int sfx_debug_lbs_dense_form(int32_t form) {
    const int prev = g_lbs_dense_form;
    if (form == 16 || form == 17 || form == 32) g_lbs_dense_form = form;
    return prev;
}

// debug: attach (enable=1) / read out and detach (enable=0) the 64-slot clock buffer; while attached,
// closure launches stamp dbg[0..18] and optimiser ticks of frame 0 accumulate dbg[32..63]
extern "C" int sfx_debug_clocks(sfx_batch* b, int32_t enable, int64_t* out /* [64] or NULL */) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    if (enable) {
        if (!b->D.dbg) SFX_CHECK(hipMalloc((void**)&b->D.dbg, 64 * sizeof(long long)));
        SFX_CHECK(hipMemset(b->D.dbg, 0, 64 * sizeof(long long)));
        const long long freeze = enable > 1 ? enable : 40;      // k_tick_dense: stamps freeze after this launch
        SFX_CHECK(hipMemcpy(b->D.dbg + 61, &freeze, sizeof(freeze), hipMemcpyHostToDevice));
        return 0;
    }
    if (!b->D.dbg) { sfx_set_error("clock buffer not attached"); return -1; }
    SFX_CHECK(hipDeviceSynchronize());
    long long h[64];
    SFX_CHECK(hipMemcpy(h, b->D.dbg, sizeof(h), hipMemcpyDeviceToHost));
    hipFree(b->D.dbg); b->D.dbg = nullptr;
    if (out) for (int i = 0; i < 64; ++i) out[i] = h[i];
    return 0;
}
#endif

extern "C" int sfx_batch_num_vars(sfx_batch* b, int32_t stage) {
    if (!b) return -1;
    return b->vl_host[stage < 0 ? 0 : 1].n;
}

// one closure evaluation of every (active) frame; stage_override = -2 -> per-frame stage[]
// penetration term of the pending evaluation of every active frame (after the dense LBS wrote the
// vertices, before the loss / adjoint pass reads pen_loss, pen_dverts and the vertex lists)
// which GEMM columns hold a frame whose pending evaluation carries a collision weight (fitting.py:437:
// the term is only evaluated while coll_loss_weight > 0)
__global__ void k_pen_want(BatchDev D, const StageW* __restrict__ sws, int stage_override) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= D.cfg.B) return;
    const int st = stage_override != -2 ? stage_override : D.stage[b];
    if (st >= 0 && st < D.cfg.n_stages && D.slot[b] >= 0) D.pen_want[D.slot[b]] = sws[st].coll > 0.f;
}

// Upper bound on the rounds a fit can take (a guard against a loop that never ends, not a budget): an LBFGS.step costs at most
// max_eval evaluations in its iterations plus one line search that may run over -- the bracket phase up to 25 evaluations, the zoom
// phase up to LBFGS(max_iter) (lbfgs_ls.py:108) -- plus the entry evaluation; a side view is fitted from two orientations
// (fit_single_frame.py:527-551), accounted for as a factor of its own (the old bound let the `* 2` do both).
static long sfx_fit_tick_bound(const BatchCfgDev& c, int first_stage, int last_stage) {
    const long per_step = (long)c.max_eval + std::max(25, (int)c.lbfgs_max_iter) + 1;
    return (long)(last_stage - first_stage + 1) * c.maxiters * std::max(80L, per_step + 7) * 2 /* orientations */ * 2 /* slack */ + 64;
}

// want_ready: the fused tick kernel maintains pen_want[] itself (it knows every running frame's next stage when it exports
// the next trial point, and clears the flag of a frame that finishes): no launch for it inside the fitting loop
static int eval_penetration(sfx_batch* b, int stage_override, hipStream_t s, bool want_ready = false) {
    const BatchDev& D = b->D;
    if (!b->pen || D.nact <= 0) return 0;
    ProfScope p("penetration", s, D.nact);
    if (!want_ready) {
        hipMemsetAsync(D.pen_want, 0, (size_t)D.cfg.B * sizeof(int), s);
        hipLaunchKernelGGL(k_pen_want, dim3((D.cfg.B + 63) / 64), dim3(64), 0, s, D, b->sw_dev, stage_override);
    }
    // the collision buffers hold one mesh per column of the POOL (cfg.slots); inside the fused loop nact never exceeds it.
    // A stand-alone call on a pooled batch (sfx_batch_closure / sfx_batch_step: one column per frame, nact = B) walks the
    // columns in chunks of the pool's size -- the buffers are scratch between launches, every result lands in per-frame arrays
    const int cap = sfx_pen_capacity(b->pen);
    const DevModel& M = b->m->M;
    const size_t V3 = (size_t)M.V * 3;
    b->pen_chunked = D.nact > cap;
    const int stride = sfx_pen_stats_stride();
    if (b->pen_chunked && !b->pen_stats_all) {
        b->pen_stats_all = b->mem.zeros<int>((size_t)D.cfg.B * stride);
        if (!b->pen_stats_all) { sfx_set_error("out of device memory"); return -2; }
    }
    // Fused loop: the step is a captured graph (SFX_PEN_GRAPH=0 switches it off).  The first evaluation of a process runs directly
    // (one-time attribute calls inside), every new column count is captured once on the loop's own stream.
#ifdef SFX_LAB
    static const bool graph_on = [] { const char* e = getenv("SFX_PEN_GRAPH"); return !e || atoi(e) != 0; }();
#else
    constexpr bool graph_on = true;
#endif
    static bool warmed = false;
    if (want_ready && graph_on && warmed && !b->pen_chunked) {
        auto it = b->pen_graphs.find(D.nact);
        if (it == b->pen_graphs.end()) {
            hipGraph_t graph = nullptr; hipGraphExec_t exec = nullptr;
            if (!b->cap_stream) SFX_CHECK(hipStreamCreateWithFlags(&b->cap_stream, hipStreamNonBlocking));
            hipStream_t cs = b->cap_stream;
            SFX_CHECK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
            PenAdjPrep ap{D.AT, M.Wsp_j, M.Wsp_w, M.W, D.adj_G, D.Bpad, M.Vpad};
            const int rc = sfx_pen_eval_masked(b->pen, D.nact, D.verts, b->pen_sigma, b->pen_outside, D.pen_loss, D.pen_dverts, D.pen_want, &ap, D.pen_over, cs);
            launch_pen_adjoint(M, D, cs);
            const hipError_t ec = hipStreamEndCapture(cs, &graph);
            if (rc || ec != hipSuccess || !graph) { if (graph) hipGraphDestroy(graph); (void)hipGetLastError(); sfx_set_error("capture of the interpenetration step failed"); return rc ? rc : -2; }
            {   // what one round launches for the term: counted on the captured graph, not written down
                size_t nn = 0;
                if (hipGraphGetNodes(graph, nullptr, &nn) == hipSuccess) {
                    std::vector<hipGraphNode_t> nodes(nn);
                    int nk = 0;
                    if (nn && hipGraphGetNodes(graph, nodes.data(), &nn) == hipSuccess)
                        for (size_t i = 0; i < nn; ++i) { hipGraphNodeType ty; if (hipGraphNodeGetType(nodes[i], &ty) == hipSuccess && ty == hipGraphNodeTypeKernel) ++nk; }
                    b->pen_graph_nodes = nk;
                }
            }
            const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            if (ei != hipSuccess) { sfx_set_error("hipGraphInstantiate failed"); return -2; }
            it = b->pen_graphs.emplace(D.nact, exec).first;
        }
        sfx_pen_note_batch(b->pen, D.nact);
        SFX_CHECK(hipGraphLaunch(it->second, s));
        return 0;
    }
    warmed = true;
    for (int c0 = 0; c0 < D.nact; c0 += cap) {
        const int n = std::min(cap, D.nact - c0);
        // the lane that forms a vertex' gradient also writes d v_posed = T^T g, the adjoint GEMM's operand (column c0 + local index)
        PenAdjPrep ap{D.AT + c0, M.Wsp_j, M.Wsp_w, M.W, D.adj_G + (size_t)c0 * 3 * M.Vpad, D.Bpad, M.Vpad};
        int rc = sfx_pen_eval_masked(b->pen, n, D.verts + c0 * V3, b->pen_sigma, b->pen_outside, D.pen_loss + c0, D.pen_dverts + c0 * V3,
                                     D.pen_want + c0, &ap, D.pen_over + c0, s);
        if (rc) return rc;
        if (b->pen_chunked)     // keep this chunk's diagnostics: the next chunk reuses the rows
            SFX_CHECK(hipMemcpyAsync(b->pen_stats_all + (size_t)c0 * stride, sfx_pen_stats_dev(b->pen), (size_t)n * stride * sizeof(int), hipMemcpyDeviceToDevice, s));
    }
    launch_pen_adjoint(M, D, s);
    return 0;
}

static int eval_closure(sfx_batch* b, int stage_override, int from_X, hipStream_t s) {
    const DevModel& M = b->m->M; const BatchDev& D = b->D;
    ClosureArgs a{};
    a.stage_override = stage_override; a.from_X = from_X;
    // the camera stage asks for return_verts=False (fit_single_frame.py:485): the reference
    // still evaluates every vertex there, so dense mode does too.
    if (D.cfg.lbs_mode == 1) {
        ClosureArgs e = a; e.export_dense = 1; e.forward_only = 2;
        { ProfScope p("export", s); launch_closure(M, D, b->vl_dev, b->sw_dev, e, s); }
        { ProfScope p("lbs_dense", s, D.nact); launch_lbs_dense(M, D, s); }
        if (int rc = eval_penetration(b, stage_override, s)) return rc;
        a.use_dense_verts = 1;
    }
    ProfScope p("closure", s);
    launch_closure(M, D, b->vl_dev, b->sw_dev, a, s);
    return 0;
}

extern "C" int sfx_batch_closure(sfx_batch* b, int32_t stage, float* loss_out, float* grad_out, void* stream) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    if (stage >= b->D.cfg.n_stages) { sfx_set_error("stage %d out of range", stage); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (int rc = eval_closure(b, stage < 0 ? -1 : stage, 1, s)) return rc;
    SFX_CHECK(hipStreamSynchronize(s));
    SFX_CHECK(hipGetLastError());
    const int B = b->D.cfg.B, N = b->vl_host[stage < 0 ? 0 : 1].n;
    if (loss_out) SFX_CHECK(hipMemcpy(loss_out, b->D.f, (size_t)B * 4, hipMemcpyDeviceToHost));
    if (grad_out) {
        std::vector<float> g((size_t)B * SFX_NVAR_MAX);
        SFX_CHECK(hipMemcpy(g.data(), b->D.g, g.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < B; ++i) memcpy(grad_out + (size_t)i * N, &g[(size_t)i * SFX_NVAR_MAX], (size_t)N * 4);
    }
    return 0;
}

__global__ void k_guess_init(BatchDev D, int K, const int* pairs, int n_pairs) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= D.cfg.B) return;
    const float* j3 = D.joints + (size_t)b * K * 3;
    const float* j2 = D.gt + (size_t)b * K * 2;
    float s3 = 0.f, s2 = 0.f;
    for (int p = 0; p < n_pairs; ++p) {
        const int a = pairs[2 * p], c = pairs[2 * p + 1];
        const float dx = j3[a * 3] - j3[c * 3], dy = j3[a * 3 + 1] - j3[c * 3 + 1], dz = j3[a * 3 + 2] - j3[c * 3 + 2];
        const float ex = j2[a * 2] - j2[c * 2], ey = j2[a * 2 + 1] - j2[c * 2 + 1];
        s3 += sqrtf(dx * dx + dy * dy + dz * dz);
        s2 += sqrtf(ex * ex + ey * ey);
    }
    const float est = D.cam[(size_t)b * 8 + 0] * ((s3 / n_pairs) / (s2 / n_pairs));
    float* x = D.X + (size_t)b * SFX_NPAR_MAX + D.L.cam_t;
    x[0] = 0.f; x[1] = 0.f; x[2] = est;
    float* xt = D.Xt + (size_t)b * SFX_NPAR_MAX + D.L.cam_t;
    xt[0] = 0.f; xt[1] = 0.f; xt[2] = est;
    D.cam[(size_t)b * 8 + 5] = est;
}

extern "C" int sfx_batch_guess_init(sfx_batch* b, const int32_t* pairs, int32_t n_pairs, void* stream) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    std::vector<int> pv(pairs, pairs + 2 * n_pairs);
    int* pd = nullptr;
    SFX_CHECK(hipMalloc((void**)&pd, pv.size() * sizeof(int)));
    SFX_CHECK(hipMemcpyAsync(pd, pv.data(), pv.size() * sizeof(int), hipMemcpyHostToDevice, s));
    ClosureArgs a{}; a.stage_override = -1; a.forward_only = 1; a.from_X = 1;
    launch_closure(b->m->M, b->D, b->vl_dev, b->sw_dev, a, s);
    hipLaunchKernelGGL(k_guess_init, dim3((b->D.cfg.B + 63) / 64), dim3(64), 0, s, b->D, b->K, pd, n_pairs);
    pack_fd(b, s);          // (est_tz changed)
    SFX_CHECK(hipStreamSynchronize(s));
    hipFree(pd);
    SFX_CHECK(hipGetLastError());
    return 0;
}

// tick loop shared by sfx_batch_fit (whole run_fitting) and sfx_batch_step (one LBFGS.step)
static int run_ticks(sfx_batch* b, int first_stage, int last_stage, int init, int step_mode, hipStream_t s) {
    BatchDev& D = b->D; const DevModel& M = b->m->M;
    const int B = D.cfg.B;
    const bool dense = D.cfg.lbs_mode == 1;
    const bool fused = !step_mode && !g_unfused;
    D.act = nullptr; D.nrun = 0;          // (a previous fit that ended on an error may have left its running list attached)
    if (init == 1 && D.pen_flag) SFX_CHECK(hipMemsetAsync(D.pen_flag, 0, (size_t)B * sizeof(int), s));
    launch_lbfgs_tick(M, D, b->vl_dev, first_stage, last_stage, init, step_mode, s);
    // bound on the rounds of the polled loops: one resident batch needs at most stages x maxiters LBFGS.step calls of
    // <= ~160 evaluations each; a job of B frames through a pool of `slots` columns needs that once per wave of the queue
    long max_ticks = sfx_fit_tick_bound(D.cfg, first_stage, last_stage);
    if (dense && b->slots > 0 && b->slots < B) max_ticks *= (B + b->slots - 1) / b->slots;
    std::vector<int> hs(B);
    int* hp = b->stage_host ? b->stage_host : hs.data();
    long tick = 0;
    bool done = false;
    if (fused && dense && !(b->stage_host && b->map_host)) { ProfScope p("tick", s, B); launch_tick_dense(M, D, b->vl_dev, b->sw_dev, first_stage, last_stage, 0, s); }
    if (fused && dense && b->stage_host && b->map_host) {
        // dense fused loop, polled one batch of rounds AHEAD: while the host waits for the stage flags
        // copied after rounds 8i .. 8i+7, rounds 8i+8 .. 8i+15 are already queued, so the GPU never idles
        // on the host round trip.  A decision (all done / admission / compaction) therefore lags by 8 rounds:
        // finished frames only ever stay finished, so that is safe; the surplus rounds at the end find nothing to do.
        //
        // Column pool (continuous batching, cfg.slots): `pool` GEMM columns; frames beyond the first `pool` wait in
        // a queue (slot = -1, not in the running list) and take over the column of a frame that has finished -- their
        // first pose / chain export is one extra launch over the admitted frames only.  Once the queue is dry the
        // columns are compacted whenever a 32-frame MFMA slice has emptied, as before.  Frames are independent and a
        // column's arithmetic does not depend on its index: results equal those of a batch with one column per frame.
#ifdef SFX_LAB       // measurement switches of the lab build (include/sfx_lab.h): diagnostics to stderr, polling cadence
        static const bool dbg_nact = getenv("SFX_DEBUG_NACT") != nullptr, dbg_host = getenv("SFX_DEBUG_HOST") != nullptr;
        static const int rpb_env = [] { const char* e = getenv("SFX_POLL_ROUNDS"); return e ? atoi(e) : 0; }();
        static const int ahead_env = [] { const char* e = getenv("SFX_POLL_AHEAD"); return e ? atoi(e) : 0; }();
#else
        constexpr bool dbg_nact = false, dbg_host = false;
        constexpr int rpb_env = 0, ahead_env = 0;
#endif
        static long nact_hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};      // rounds by active GEMM columns: <=32, <=64, ..., <=256, more
        const int pool = b->slots > 0 ? std::min(b->slots, B) : B;
        std::vector<int>& col = b->slot_host;                          // frame -> column, -1 = queued or retired
        col.assign(B, -1);
        std::vector<int> run;                                          // running frames, ascending admission order
        run.reserve(B);
        for (int i = 0; i < pool; ++i) { col[i] = i; run.push_back(i); }
        int next_q = pool, upl = 0;
        auto upload = [&](const std::vector<int>* fresh) -> int {      // slot[], running list (+ admitted list) through pinned memory
            int* h = b->map_host + (size_t)upl * 3 * B; upl = (upl + 1) % SFX_POLL_BUFS;      // (at most one upload per processed batch, at most `ahead` + 1 batches in flight)
            memcpy(h, col.data(), (size_t)B * sizeof(int));
            memcpy(h + B, run.data(), run.size() * sizeof(int));
            SFX_CHECK(hipMemcpyAsync(D.slot, h, (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
            if (!run.empty()) SFX_CHECK(hipMemcpyAsync(b->act_dev, h + B, run.size() * sizeof(int), hipMemcpyHostToDevice, s));
            if (fresh && !fresh->empty()) {
                memcpy(h + 2 * B, fresh->data(), fresh->size() * sizeof(int));
                SFX_CHECK(hipMemcpyAsync(b->act_new_dev, h + 2 * B, fresh->size() * sizeof(int), hipMemcpyHostToDevice, s));
            }
            D.act = b->act_dev; D.nrun = (int)run.size();
            return 0;
        };
        D.nact = pool;
        if (int rc = upload(nullptr)) return rc;
        { ProfScope p("tick", s, D.nrun); launch_tick_dense(M, D, b->vl_dev, b->sw_dev, first_stage, last_stage, 0, s); }
        // SFX_DEBUG_HOST=1: how much of the loop's wall time the HOST spends enqueueing (its headroom against a busy box)
        double host_enq_s = 0.0, host_wait_s = 0.0; long host_batches = 0;
        const double wall0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
        const int rpb = std::max(1, std::min(64, rpb_env > 0 ? rpb_env : 8));          // rounds per polled batch
        auto rounds = [&](int buf) -> int {
            if (dbg_nact) nact_hist[std::min(8, (D.nact - 1) / 32)] += rpb;
            const auto h0 = std::chrono::steady_clock::now();
            struct HostClock { const std::chrono::steady_clock::time_point t0; double& acc; long& n; bool on;
                               ~HostClock() { if (on) { acc += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); ++n; } } } hc{h0, host_enq_s, host_batches, true};
            for (int q = 0; q < rpb; ++q, ++tick) {
                { ProfScope p("lbs_dense", s, D.nact); launch_lbs_dense(M, D, s); }
                if (int rc = eval_penetration(b, -2, s, true)) return rc;
                ProfScope p("tick", s, D.nrun);
                launch_tick_dense(M, D, b->vl_dev, b->sw_dev, first_stage, last_stage, 1, s);
            }
            SFX_CHECK(hipMemcpyAsync(b->stage_host + (size_t)buf * B, D.stage, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
            SFX_CHECK(hipEventRecord(b->poll_ev[buf], s));
            return 0;
        };
        const int ahead = std::max(1, std::min(SFX_POLL_BUFS - 1, ahead_env > 0 ? ahead_env : (b->pen ? 1 : 3)));
        long q_head = 0, q_next = 0;       // batches processed / queued
        for (; q_next < ahead; ++q_next) if (int rc = rounds((int)(q_next % SFX_POLL_BUFS))) return rc;
        while (!done && tick < max_ticks + (long)rpb * (ahead + 1)) {
            if (int rc = rounds((int)(q_next % SFX_POLL_BUFS))) return rc;
            ++q_next;
            const int cur = (int)(q_head % SFX_POLL_BUFS); ++q_head;
            {
                const auto w0 = std::chrono::steady_clock::now();
                SFX_CHECK(hipEventSynchronize(b->poll_ev[cur]));
                host_wait_s += std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count();
            }
            const int* hq = b->stage_host + (size_t)cur * B;
            // frames of the running list that have finished (frames admitted after this snapshot show their start stage)
            std::vector<int> fresh;
            size_t w = 0;
            bool changed = false;
            for (size_t r = 0; r < run.size(); ++r) {
                const int f = run[r];
                if (hq[f] <= last_stage) { run[w++] = f; continue; }
                changed = true;
                const int c = col[f];
                col[f] = -1;
                if (next_q < B) { const int nf = next_q++; col[nf] = c; fresh.push_back(nf); }
            }
            run.resize(w);
            for (int nf : fresh) run.push_back(nf);
            done = run.empty();
            if (!done && changed) {
                if (!fresh.empty()) {
                    // admission: the new frames inherit the freed columns; only they need an export pass
                    if (int rc = upload(&fresh)) return rc;
                    BatchDev Dn = D; Dn.act = b->act_new_dev; Dn.nrun = (int)fresh.size();
                    ProfScope p("tick_admit", s, Dn.nrun);
                    launch_tick_dense(M, Dn, b->vl_dev, b->sw_dev, first_stage, last_stage, 0, s);
                } else if ((D.nact + 31) / 32 != ((int)run.size() + 31) / 32 || (D.nact > 16 && run.size() <= 16)) {      // (round 5: also down to ONE 16-frame slice -- the few-frames GEMM's floor is 17.5 us there, 23 at two slices)
                    // queue dry and a 32-frame MFMA slice has emptied: compact.  The pending evaluation of every running
                    // frame lives in column slot[f] of featR / AT, so re-export after remapping (queued behind the rounds
                    // already in flight, which still use the old mapping consistently).
                    int q = 0;
                    for (int f : run) col[f] = q++;
                    D.nact = q;
                    if (int rc = upload(nullptr)) return rc;
                    ProfScope p("tick", s, D.nrun);
                    launch_tick_dense(M, D, b->vl_dev, b->sw_dev, first_stage, last_stage, 0, s);   // re-export only
                } else {
                    if (int rc = upload(nullptr)) return rc;      // shorter running list: fewer workgroups per tick launch
                }
            }
        }
        SFX_CHECK(hipStreamSynchronize(s));
        D.act = nullptr; D.nrun = 0;
        const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() - wall0;
        {   // the host's side of this loop, accumulated for sfx_loop_host_stats (bench.py reports it next to the kernels' times)
            std::lock_guard<std::mutex> lk(g_prof_mu);
            g_loop_host[0] += host_enq_s; g_loop_host[1] += host_wait_s; g_loop_host[2] += wall; g_loop_host[3] += (double)host_batches * rpb;
        }
        if (dbg_host) {
            fprintf(stderr, "[sfx] host enqueue: %ld batches of rounds, %.1f us of host time per round, %.1f %% of the loop's %.1f ms wall time\n",
                    host_batches, 1e6 * host_enq_s / std::max(1L, host_batches * rpb), 100.0 * host_enq_s / std::max(wall, 1e-9), 1e3 * wall);
        }
        if (dbg_nact) {
            fprintf(stderr, "[sfx] rounds by active columns (<=32, <=64, ..., <=256, more), cumulative:");
            for (int i = 0; i < 9; ++i) fprintf(stderr, " %ld", nact_hist[i]);
            fprintf(stderr, "\n");
        }
    } else
    while (!done && tick < max_ticks) {
        if (fused && !dense) {
            // persistent per-frame workgroups; the host only re-launches frames that need more ticks
            const int chunk = 512;
            { ProfScope p("fit_rows", s); launch_fit_rows(M, D, b->vl_dev, b->sw_dev, first_stage, last_stage, chunk, s); }
            tick += chunk;
        } else if (fused) {
            for (int q = 0; q < 8; ++q, ++tick) {
                { ProfScope p("lbs_dense", s, D.nact); launch_lbs_dense(M, D, s); }
                if (int rc = eval_penetration(b, -2, s, true)) return rc;
                ProfScope p("tick", s);
                launch_tick_dense(M, D, b->vl_dev, b->sw_dev, first_stage, last_stage, 1, s);
            }
        } else {
            const int POLL = dense ? 8 : 32;
            for (int q = 0; q < POLL; ++q, ++tick) {
                if (int rc = eval_closure(b, -2, 0, s)) return rc;
                ProfScope p("lbfgs", s);
                launch_lbfgs_tick(M, D, b->vl_dev, first_stage, last_stage, 0, step_mode, s);
            }
        }
        SFX_CHECK(hipMemcpyAsync(hp, D.stage, (size_t)B * sizeof(int), hipMemcpyDeviceToHost, s));
        SFX_CHECK(hipStreamSynchronize(s));
        done = true;
        for (int i = 0; i < B; ++i) if (hp[i] <= last_stage) { done = false; break; }
        if (fused && dense && !done) {
            // compaction: finished frames give up their GEMM columns.  The pending evaluation of
            // every active frame lives in column slot[b] of featR / AT, so re-export after remapping.
            int n = 0;
            for (int i = 0; i < B; ++i) if (hp[i] <= last_stage) ++n;
            if ((b->D.nact + 31) / 32 != (n + 31) / 32) {      // a 32-frame MFMA tile became free
                std::vector<int>& sl = b->slot_host;
                sl.assign(B, 0);
                int q = 0;
                for (int i = 0; i < B; ++i) sl[i] = (hp[i] <= last_stage) ? q++ : 0;
                SFX_CHECK(hipMemcpyAsync(D.slot, sl.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
                b->D.nact = n;
                ProfScope p("tick", s);
                launch_tick_dense(M, b->D, b->vl_dev, b->sw_dev, first_stage, last_stage, 0, s);   // re-export only
            }
        }
    }
    SFX_CHECK(hipGetLastError());
    if (fused && dense) {       // restore the identity mapping for stand-alone calls
        std::vector<int>& sl = b->slot_host;
        sl.resize(B);
        for (int i = 0; i < B; ++i) sl[i] = i;
        SFX_CHECK(hipMemcpyAsync(D.slot, sl.data(), (size_t)B * sizeof(int), hipMemcpyHostToDevice, s));
        SFX_CHECK(hipStreamSynchronize(s));
        b->D.nact = B;
    }
    if (!done) { sfx_set_error("fit did not finish within %ld ticks", max_ticks); return -4; }
    return 0;
}

extern "C" int sfx_batch_fit(sfx_batch* b, int32_t first_stage, int32_t last_stage, void* stream) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    if (first_stage < -1 || last_stage >= b->D.cfg.n_stages || last_stage < first_stage) {
        sfx_set_error("bad stage range [%d,%d]", first_stage, last_stage); return -1;
    }
    return run_ticks(b, first_stage, last_stage, 1, 0, (hipStream_t)stream);
}

extern "C" int sfx_batch_step(sfx_batch* b, int32_t stage, int32_t resume, float* loss_out, void* stream) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    if (stage < -1 || stage >= b->D.cfg.n_stages) { sfx_set_error("stage %d out of range", stage); return -1; }
    int rc = run_ticks(b, stage, stage, resume ? 2 : 1, 1, (hipStream_t)stream);
    if (rc) return rc;
    if (loss_out) {
        const int B = b->D.cfg.B;
        std::vector<float> l((size_t)B * (1 + SFX_MAX_STAGES));
        SFX_CHECK(hipMemcpy(l.data(), b->D.stage_loss, l.size() * 4, hipMemcpyDeviceToHost));
        for (int i = 0; i < B; ++i) loss_out[i] = l[(size_t)i * (1 + SFX_MAX_STAGES) + stage + 1];
    }
    return 0;
}

// body_pose_prior = MaxMixturePrior (prior.py:100-231) for use_vposer=False fits without a regression
// prior (fitting.py:399-401).  means [M][D], precisions [M][D][D], nll_weights [M] (as the module's buffers)
extern "C" int sfx_batch_set_gmm_form(sfx_batch* b, int32_t M, int32_t Dm, const float* means, const float* precisions,
                                      const float* nll_weights, const float* comp_const);
extern "C" int sfx_batch_set_gmm(sfx_batch* b, int32_t M, int32_t Dm, const float* means, const float* precisions,
                                 const float* nll_weights) {
    return sfx_batch_set_gmm_form(b, M, Dm, means, precisions, nll_weights, nullptr);
}
extern "C" int sfx_batch_set_gmm_form(sfx_batch* b, int32_t M, int32_t Dm, const float* means, const float* precisions,
                                      const float* nll_weights, const float* comp_const) {
    if (!b || !means || !precisions || !nll_weights) { sfx_set_error("null argument"); return -1; }
    if (M < 1 || M > 2 * (256 / 64)) { sfx_set_error("1..8 mixture components supported, got %d", M); return -1; }
    if (b->D.cfg.use_vposer) { sfx_set_error("the mixture prior acts on body_pose: use_vposer must be off"); return -1; }
    if (Dm != b->D.L.NEMB) { sfx_set_error("mixture dimension %d != body pose dimension %d", Dm, b->D.L.NEMB); return -1; }
    std::vector<float> mu((size_t)M * 64, 0.f), P((size_t)M * 64 * 64, 0.f), lw(M);
    for (int m = 0; m < M; ++m) {
        if (!(nll_weights[m] > 0.f)) { sfx_set_error("nll_weights must be positive"); return -1; }
        lw[m] = logf(nll_weights[m]);
        for (int i = 0; i < Dm; ++i) {
            mu[(size_t)m * 64 + i] = means[(size_t)m * Dm + i];
            for (int j = 0; j < Dm; ++j)
                P[((size_t)m * 64 + j) * 64 + i] = 0.5f * (precisions[((size_t)m * Dm + i) * Dm + j] + precisions[((size_t)m * Dm + j) * Dm + i]);
        }
    }
    // merged form (prior.py:186-201): min_m [0.5 q_m - log w_m]; per-component form (:203-225): argmin_m [q_m + c_m], value + (-log w_m*)
    std::vector<float> csel(M), cadd(M);
    for (int m = 0; m < M; ++m) { csel[m] = comp_const ? comp_const[m] : -lw[m]; cadd[m] = comp_const ? -lw[m] : 0.f; }
    b->D.gmm_mean = b->mem.up(mu); b->D.gmm_prec = b->mem.up(P); b->D.gmm_lognw = b->mem.up(lw);
    b->D.gmm_csel = b->mem.up(csel); b->D.gmm_cadd = b->mem.up(cadd); b->D.gmm_scale = comp_const ? 1.f : 0.5f;
    if (!b->D.gmm_mean || !b->D.gmm_prec || !b->D.gmm_lognw || !b->D.gmm_csel || !b->D.gmm_cadd) { sfx_set_error("out of device memory"); return -2; }
    b->D.gmm_M = M;
    return 0;
}

__global__ void k_count_grad_vertices(BatchDev D, int V) {
    __shared__ int cnt;
    const int b = blockIdx.x;
    if (threadIdx.x == 0) cnt = 0;
    __syncthreads();
    int n = 0;
    if (D.pen_want[b]) {
        const float* g = D.pen_dverts + (size_t)b * V * 3;
        for (int v = threadIdx.x; v < V; v += blockDim.x) n += (g[v * 3] != 0.f || g[v * 3 + 1] != 0.f || g[v * 3 + 2] != 0.f) ? 1 : 0;
    }
    atomicAdd(&cnt, n);
    __syncthreads();
    if (threadIdx.x == 0) D.ext_n[b] = cnt;
}

extern "C" int sfx_batch_pen_stats(sfx_batch* b, int32_t* stats_host /* [B][4] */, int32_t* ext_n_host /* [B] or NULL */) {
    if (!b || !stats_host) { sfx_set_error("null argument"); return -1; }
    if (!b->pen) { sfx_set_error("batch was created without interpenetration"); return -1; }
    const int n = std::max(1, b->D.nact);
    int rc = b->pen_chunked ? sfx_pen_stats_from(b->pen_stats_all, n, stats_host)
                            : sfx_pen_stats(b->pen, std::min(n, sfx_pen_capacity(b->pen)), stats_host);
    if (rc) return rc;
    if (ext_n_host) {       // vertices that carry a gradient: counted on demand (it was an atomic per wavefront in every evaluation)
        hipLaunchKernelGGL(k_count_grad_vertices, dim3(n), dim3(256), 0, 0, b->D, b->m->M.V);
        SFX_CHECK(hipDeviceSynchronize());
        SFX_CHECK(hipMemcpy(ext_n_host, b->D.ext_n, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
    }
    return 0;
}

extern "C" int sfx_batch_pen_pairs(sfx_batch* b, int32_t column, int32_t cap, int32_t* pairs_host, int32_t* n_out) {
    if (!b || !n_out) { sfx_set_error("null argument"); return -1; }
    if (!b->pen) { sfx_set_error("batch was created without interpenetration"); return -1; }
    if (b->pen_chunked) { sfx_set_error("the pair lists of a pooled batch's stand-alone evaluation are overwritten chunk by chunk"); return -1; }
    return sfx_pen_pairs(b->pen, column, cap, pairs_host, n_out);
}

extern "C" int sfx_batch_pen_launches(sfx_batch* b) {
    if (!b) { sfx_set_error("null argument"); return -1; }
    return b->pen_graph_nodes;
}

extern "C" int sfx_batch_pen_flags(sfx_batch* b, int32_t* flags_host /* [B] */) {
    if (!b || !flags_host) { sfx_set_error("null argument"); return -1; }
    if (!b->pen) { sfx_set_error("batch was created without interpenetration"); return -1; }
    SFX_CHECK(hipDeviceSynchronize());
    SFX_CHECK(hipMemcpy(flags_host, b->D.pen_flag, (size_t)b->D.cfg.B * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int sfx_batch_get_grad(sfx_batch* b, int32_t stage, float* grad_out) {
    if (!b || !grad_out) { sfx_set_error("null argument"); return -1; }
    if (stage < -1 || stage >= b->D.cfg.n_stages) { sfx_set_error("stage %d out of range", stage); return -1; }
    const int B = b->D.cfg.B, n = b->vl_host[stage < 0 ? 0 : 1].n;
    std::vector<float> g((size_t)B * SFX_NVAR_MAX);
    SFX_CHECK(hipMemcpy(g.data(), b->D.g, g.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < B; ++i) memcpy(grad_out + (size_t)i * n, g.data() + (size_t)i * SFX_NVAR_MAX, (size_t)n * 4);
    return 0;
}

extern "C" int sfx_batch_trace(sfx_batch* b, int32_t capacity) {
    if (!b) { sfx_set_error("bad argument"); return -1; }
    BatchDev& D = b->D;
    D.trace_evals = capacity < 0 ? 1 : 0;          // negative capacity: also one record per closure evaluation (debug)
    if (capacity < 0) capacity = -capacity;
    SFX_CHECK(hipDeviceSynchronize());
    if (D.trace) { hipFree(D.trace); D.trace = nullptr; }
    if (D.trace_n) { hipFree(D.trace_n); D.trace_n = nullptr; }
    D.trace_cap = 0;
    if (capacity == 0) return 0;
    // both buffers or neither: the kernels test D.trace alone, so a half-attached trace must never be left behind
    float4* tr = nullptr; int* tn = nullptr;
    hipError_t e = hipMalloc((void**)&tr, (size_t)D.cfg.B * capacity * sizeof(float4));
    if (e == hipSuccess) e = hipMalloc((void**)&tn, (size_t)D.cfg.B * sizeof(int));
    if (e == hipSuccess) e = hipMemset(tn, 0, (size_t)D.cfg.B * sizeof(int));
    if (e != hipSuccess) {
        if (tr) (void)hipFree(tr);
        if (tn) (void)hipFree(tn);
        sfx_set_error("sfx_batch_trace: %s", hipGetErrorString(e));
        return -2;
    }
    D.trace = tr; D.trace_n = tn; D.trace_cap = capacity;
    return 0;
}

extern "C" int sfx_batch_get_trace(sfx_batch* b, float* records, int32_t* counts) {
    if (!b || !b->D.trace) { sfx_set_error("no trace buffer attached (sfx_batch_trace)"); return -1; }
    const BatchDev& D = b->D;
    SFX_CHECK(hipDeviceSynchronize());
    if (records) SFX_CHECK(hipMemcpy(records, D.trace, (size_t)D.cfg.B * D.trace_cap * sizeof(float4), hipMemcpyDeviceToHost));
    if (counts) SFX_CHECK(hipMemcpy(counts, D.trace_n, (size_t)D.cfg.B * sizeof(int), hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int sfx_batch_get_stats(sfx_batch* b, float* stage_loss, int32_t* stage_evals, int32_t* stage_ref_evals) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    const int B = b->D.cfg.B, NS = b->D.cfg.n_stages + 1;
    std::vector<float> l((size_t)B * (1 + SFX_MAX_STAGES));
    std::vector<int> e(l.size()), r(l.size());
    SFX_CHECK(hipMemcpy(l.data(), b->D.stage_loss, l.size() * 4, hipMemcpyDeviceToHost));
    SFX_CHECK(hipMemcpy(e.data(), b->D.stage_evals, e.size() * 4, hipMemcpyDeviceToHost));
    SFX_CHECK(hipMemcpy(r.data(), b->D.stage_ref_evals, r.size() * 4, hipMemcpyDeviceToHost));
    for (int i = 0; i < B; ++i)
        for (int q = 0; q < NS; ++q) {
            if (stage_loss) stage_loss[(size_t)i * NS + q] = l[(size_t)i * (1 + SFX_MAX_STAGES) + q];
            if (stage_evals) stage_evals[(size_t)i * NS + q] = e[(size_t)i * (1 + SFX_MAX_STAGES) + q];
            if (stage_ref_evals) stage_ref_evals[(size_t)i * NS + q] = r[(size_t)i * (1 + SFX_MAX_STAGES) + q];
        }
    return 0;
}

extern "C" int sfx_batch_forward(sfx_batch* b, float* verts_dev, float* joints_dev, void* stream) {
    if (!b) { sfx_set_error("null batch"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    const DevModel& M = b->m->M; const BatchDev& D = b->D;
    if (D.cfg.lbs_mode == 0 && !verts_dev) {       // needed-rows batch, joints only: the forward the rows closure runs
        ClosureArgs a{}; a.stage_override = 0; a.forward_only = 1; a.from_X = 1;
        launch_closure(M, D, b->vl_dev, b->sw_dev, a, s);
        if (joints_dev) SFX_CHECK(hipMemcpyAsync(joints_dev, D.joints, (size_t)D.cfg.B * M.K * 3 * 4, hipMemcpyDeviceToDevice, s));
        SFX_CHECK(hipStreamSynchronize(s));
        SFX_CHECK(hipGetLastError());
        return 0;
    }
    ClosureArgs e{}; e.stage_override = 0; e.export_dense = 1; e.forward_only = 2; e.from_X = 1;
    launch_closure(M, D, b->vl_dev, b->sw_dev, e, s);
    { ProfScope p("lbs_dense", s, D.nact); launch_lbs_dense(M, D, s); }
    ClosureArgs a{}; a.stage_override = 0; a.forward_only = 1; a.from_X = 1; a.use_dense_verts = 1;
    launch_closure(M, D, b->vl_dev, b->sw_dev, a, s);
    const size_t nv = (size_t)D.cfg.B * M.V * 3 * 4, nj = (size_t)D.cfg.B * M.K * 3 * 4;
    if (verts_dev) SFX_CHECK(hipMemcpyAsync(verts_dev, D.verts, nv, hipMemcpyDeviceToDevice, s));
    if (joints_dev) SFX_CHECK(hipMemcpyAsync(joints_dev, D.joints, nj, hipMemcpyDeviceToDevice, s));
    SFX_CHECK(hipStreamSynchronize(s));
    SFX_CHECK(hipGetLastError());
    return 0;
}

__global__ void k_pack_params(BatchDev D, const float* go, const float* bp, const float* betas, const float* expr,
                              const float* jaw, const float* leye, const float* reye, const float* lh, const float* rh) {
    const int b = blockIdx.x, t = threadIdx.x;
    const ParLayout& L = D.L;
    float* x = D.X + (size_t)b * SFX_NPAR_MAX;
    auto cp = [&](int off, int n, const float* src) { if (t < n) x[off + t] = src ? src[(size_t)b * n + t] : 0.f; };
    cp(L.cam_t, 3, nullptr); cp(L.go, 3, go); cp(L.betas, L.NB, betas); cp(L.lh, L.NPCA, lh); cp(L.rh, L.NPCA, rh);
    cp(L.expr, L.NE, expr); cp(L.jaw, 3, jaw); cp(L.leye, 3, leye); cp(L.reye, 3, reye); cp(L.emb, 63, bp);
}

extern "C" int sfx_lbs_forward(sfx_model* m, int32_t B, const float* go, const float* bp, const float* betas,
                               const float* expr, const float* jaw, const float* leye, const float* reye,
                               const float* lh, const float* rh, float* verts_out, float* joints_out,
                               float* full_pose_out, void* stream) {
    if (!m) { sfx_set_error("null model"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    if (!m->fwd || m->fwd_B != B) {
        if (m->fwd) sfx_batch_destroy(m->fwd);
        sfx_batch_cfg c{}; c.B = B; c.n_stages = 0; c.maxiters = 1; c.num_body_joints = m->M.K; c.lbs_mode = 1;   // forward only
        sfx_stage_weights w{};
        int rc = sfx_batch_create(m, &c, &w, &m->fwd);
        if (rc) return rc;
        m->fwd_B = B;
    }
    sfx_batch* b = m->fwd;
    hipLaunchKernelGGL(k_pack_params, dim3(B), dim3(64), 0, s, b->D, go, bp, betas, expr, jaw, leye, reye, lh, rh);
    int rc = sfx_batch_forward(b, verts_out, joints_out, s);
    if (rc) return rc;
    if (full_pose_out) SFX_CHECK(hipMemcpyAsync(full_pose_out, b->D.fullpose, (size_t)B * SFX_POSE * 4, hipMemcpyDeviceToDevice, s));
    SFX_CHECK(hipStreamSynchronize(s));
    return 0;
}
