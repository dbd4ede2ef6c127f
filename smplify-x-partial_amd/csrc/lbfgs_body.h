// lbfgs.hip -- batched on-device optimiser: one wavefront per frame advances that frame's
// run_fitting / L-BFGS / strong-Wolfe state machine by ONE closure evaluation per launch.
//
// Replaces (per frame, with all Python-side control flow moved on device):
//   FittingMonitor.run_fitting        smplifyx/fitting.py:147-217
//   LBFGS.step                        smplifyx/optimizers/lbfgs_ls.py:256-445
//   _strong_Wolfe / _cubic_interpolate smplifyx/optimizers/lbfgs_ls.py:11-167
//   per-stage optimiser re-creation   smplifyx/fit_single_frame.py:553-564
// The specification is oracle/lbfgs_machine.py (same transitions, same rounding rule).
//
// The reference mixes Python floats (double) and 0-d float32 tensors; PyTorch computes in
// float32 whenever a tensor takes part.  `Sc` carries that distinction so that every
// branch is decided on identically rounded numbers.  Compile with -ffp-contract=off: the
// only fused multiply-adds are the explicit fmaf() of the axpy updates (ATen's
// add_(alpha) vector path) and of the lane partials of the two-loop recursion's dot products
// (a 64-lane summation tree that no ATen kernel shares anyway).
#pragma once
#include "sfx_internal.h"
#include "wave_ops.h"
// every multiply-add in this file is written out: no implicit contraction (see header comment)
#pragma clang fp contract(off)
#define LB_BS 8       // history pairs per block of the blocked two-loop recursion (one group of 8 lanes per member)
// single-wavefront synchronisation: order LDS/global accesses of the 64 lanes
#define LB_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); __builtin_amdgcn_wave_barrier(); } while (0)

enum { PH_ENTRY = 0, PH_BRACKET = 1, PH_ZOOM = 2 };
enum { A_NONE = 0, A_ENTRY, A_ITER_HEAD, A_ZOOM_NEXT, A_FINISH_LS, A_END_STEP, A_FINISH_STAGE };

struct Sc { double v; int t; };
__device__ __forceinline__ Sc P(double v) { Sc s; s.v = v; s.t = 0; return s; }
__device__ __forceinline__ Sc T(float v) { Sc s; s.v = (double)v; s.t = 1; return s; }
__device__ __forceinline__ Sc sc_add(Sc a, Sc b) { return (a.t | b.t) ? T((float)a.v + (float)b.v) : P(a.v + b.v); }
__device__ __forceinline__ Sc sc_sub(Sc a, Sc b) { return (a.t | b.t) ? T((float)a.v - (float)b.v) : P(a.v - b.v); }
__device__ __forceinline__ Sc sc_mul(Sc a, Sc b) { return (a.t | b.t) ? T((float)a.v * (float)b.v) : P(a.v * b.v); }
__device__ __forceinline__ Sc sc_div(Sc a, Sc b) { return (a.t | b.t) ? T((float)a.v / (float)b.v) : P(a.v / b.v); }
__device__ __forceinline__ bool sc_lt(Sc a, Sc b) { return (a.t | b.t) ? ((float)a.v < (float)b.v) : (a.v < b.v); }
__device__ __forceinline__ bool sc_le(Sc a, Sc b) { return (a.t | b.t) ? ((float)a.v <= (float)b.v) : (a.v <= b.v); }
__device__ __forceinline__ bool sc_gt(Sc a, Sc b) { return (a.t | b.t) ? ((float)a.v > (float)b.v) : (a.v > b.v); }
__device__ __forceinline__ bool sc_ge(Sc a, Sc b) { return (a.t | b.t) ? ((float)a.v >= (float)b.v) : (a.v >= b.v); }
__device__ __forceinline__ Sc sc_pmax(Sc a, Sc b) { return sc_gt(b, a) ? b : a; }   // Python max(a, b)
__device__ __forceinline__ Sc sc_pmin(Sc a, Sc b) { return sc_lt(b, a) ? b : a; }   // Python min(a, b)
__device__ __forceinline__ Sc sc_abs(Sc a) { a.v = fabs(a.v); return a; }
__device__ __forceinline__ Sc sc_neg(Sc a) { a.v = -a.v; return a; }
__device__ __forceinline__ Sc sc_sqrt(Sc a) { return a.t ? T(sqrtf((float)a.v)) : P(sqrt(a.v)); }

// lbfgs_ls.py:11-36
__device__ __forceinline__ Sc cubic_interpolate(Sc x1, Sc f1, Sc g1, Sc x2, Sc f2, Sc g2, bool has_bounds, Sc lo, Sc hi) {
    if (!has_bounds) {
        if (sc_le(x1, x2)) { lo = x1; hi = x2; } else { lo = x2; hi = x1; }
    }
    const Sc d1 = sc_sub(sc_add(g1, g2), sc_div(sc_mul(P(3.0), sc_sub(f1, f2)), sc_sub(x1, x2)));
    const Sc d2sq = sc_sub(sc_mul(d1, d1), sc_mul(g1, g2));
    if (sc_ge(d2sq, P(0.0))) {
        const Sc d2 = sc_sqrt(d2sq);
        Sc mp;
        if (sc_le(x1, x2))
            mp = sc_sub(x2, sc_mul(sc_sub(x2, x1), sc_div(sc_sub(sc_add(g2, d2), d1),
                                                          sc_add(sc_sub(g2, g1), sc_mul(P(2.0), d2)))));
        else
            mp = sc_sub(x1, sc_mul(sc_sub(x1, x2), sc_div(sc_sub(sc_add(g1, d2), d1),
                                                          sc_add(sc_sub(g1, g2), sc_mul(P(2.0), d2)))));
        return sc_pmin(sc_pmax(mp, lo), hi);
    }
    return sc_div(sc_add(lo, hi), P(2.0));
}

// cv2.Rodrigues semantics (fit_single_frame.py:528-535): rotvec -> R, R . R([0,pi,0]), -> rotvec
__device__ void flipped_orientation(const float* go, float* out) {
    const double r0 = go[0], r1 = go[1], r2 = go[2];
    const double a = sqrt(r0 * r0 + r1 * r1 + r2 * r2);
    double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    if (a >= 1e-12) {
        const double k0 = r0 / a, k1 = r1 / a, k2 = r2 / a, s = sin(a), c = 1.0 - cos(a);
        const double K[9] = {0, -k2, k1, k2, 0, -k0, -k1, k0, 0};
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
            double kk = 0; for (int q = 0; q < 3; ++q) kk += K[i * 3 + q] * K[q * 3 + j];
            R[i * 3 + j] = (i == j ? 1.0 : 0.0) + s * K[i * 3 + j] + c * kk;
        }
    }
    // R . Ry(pi): Ry(pi) = diag(-1, 1, -1) up to rounding of sin(pi); use the exact Rodrigues value
    const double sp = sin(3.14159265358979323846), cp = 1.0 - cos(3.14159265358979323846);
    const double Y[9] = {1 - cp, 0, sp, 0, 1, 0, -sp, 0, 1 - cp};
    double M_[9];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) {
        double v = 0; for (int q = 0; q < 3; ++q) v += R[i * 3 + q] * Y[q * 3 + j];
        M_[i * 3 + j] = v;
    }
    double cth = (M_[0] + M_[4] + M_[8] - 1.0) / 2.0;
    cth = cth < -1.0 ? -1.0 : (cth > 1.0 ? 1.0 : cth);
    const double ang = acos(cth);
    const double v0 = M_[7] - M_[5], v1 = M_[2] - M_[6], v2 = M_[3] - M_[1];
    const double sn = sqrt(v0 * v0 + v1 * v1 + v2 * v2) / 2.0;
    if (sn < 1e-10) {
        if (cth > 0) { out[0] = out[1] = out[2] = 0.f; return; }
        double d0 = sqrt(fmax((M_[0] + 1) / 2, 0.0)), d1 = sqrt(fmax((M_[4] + 1) / 2, 0.0)), d2 = sqrt(fmax((M_[8] + 1) / 2, 0.0));
        if (M_[1] < 0) d1 = -d1;
        if (M_[2] < 0) d2 = -d2;
        const double n = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        out[0] = (float)(d0 / n * ang); out[1] = (float)(d1 / n * ang); out[2] = (float)(d2 / n * ang);
        return;
    }
    out[0] = (float)(v0 / (2 * sn) * ang); out[1] = (float)(v1 / (2 * sn) * ang); out[2] = (float)(v2 / (2 * sn) * ang);
}

struct OptScal {
    int phase, outer, n_iter, n_iter_total, cur_evals, func_evals;
    int ls_evals, ls_iter, ls_done, insuf, low, high;
    int hist_n, hist_head, cache_valid, has_prev_outer;
    int evals, ref_evals, pad0, pad1;
    Sc t, t_prev, H_diag;
    Sc loss, prev_loss, orig_loss, f_prev, ls_f0;
    Sc gtd_prev, ls_gtd0, d_norm;
    Sc br0, br1, bf0, bf1, bgtd0, bgtd1;
    double prev_loss_outer;
};
// Band of the Gram matrix S^T Y used by the blocked two-loop recursion, one 16-float row per history slot:
// row[8 + k], k = 1..7, holds  syb: s_p . y_(k-th successor of p)   (loop 2 walks old -> new)
//                              syt: s_(k-th predecessor of p) . y_p  (loop 1 walks new -> old);
// row[0..8] stay ZERO, so a lane that reads row[8 + (c - m)] for a member c <= m of the block gets the "+0" of the
// in-block coupling without a mask.
#define LB_BROW 16
struct OptState {
    OptScal s;
    float ro[SFX_HROWS_MAX];             // (rows R..R+7 of all three mirror slots 0..7, as the history does; R: the ring's slots)
    float syb[SFX_HROWS_MAX * LB_BROW];
    float syt[SFX_HROWS_MAX * LB_BROW];
};


#define NE3 3     // elements per lane (NVAR_MAX / 64)

struct Lane3 { float v[NE3]; };

__device__ __forceinline__ float wsum(float v) { return wave_sum_dpp(v); }
__device__ __forceinline__ float wmax(float v) { return wave_max_dpp(v); }
// lane l owns elements 3l, 3l+1, 3l+2 of a (<=192)-vector: one 12-byte load per lane, 768
// contiguous bytes per wavefront instruction.  Rows are allocated NVAR_MAX long, so the wide
// load is always in bounds; elements >= N are masked to zero.
__device__ __forceinline__ Lane3 ld3(const float* p, int lane, int N) {
    Lane3 r;
    const float3 v = *reinterpret_cast<const float3*>(p + 3 * lane);
    r.v[0] = (3 * lane + 0 < N) ? v.x : 0.f;
    r.v[1] = (3 * lane + 1 < N) ? v.y : 0.f;
    r.v[2] = (3 * lane + 2 < N) ? v.z : 0.f;
    return r;
}
// history rows are written full width (elements >= N are zero in every vector derived from masked
// loads), so they can be read back without masks
__device__ __forceinline__ Lane3 ld3_raw(const float* base, unsigned row_off, int lane) {
    Lane3 r;      // 32-bit element offset: uniform base pointer + per-lane offset (saddr addressing)
    const float3 v = *reinterpret_cast<const float3*>(reinterpret_cast<const char*>(base) + (row_off * 4u + 12u * (unsigned)lane));
    r.v[0] = v.x; r.v[1] = v.y; r.v[2] = v.z;
    return r;
}
__device__ __forceinline__ void st3_full(float* p, const Lane3& a, int lane) {
    *reinterpret_cast<float3*>(p + 3 * lane) = make_float3(a.v[0], a.v[1], a.v[2]);
}
__device__ __forceinline__ void st3(float* p, const Lane3& a, int lane, int N) {
#pragma unroll
    for (int e = 0; e < NE3; ++e) { const int i = 3 * lane + e; if (i < N) p[i] = a.v[e]; }
}
__device__ __forceinline__ float dot3(const Lane3& a, const Lane3& b) {
    float p = a.v[0] * b.v[0];
    p = p + a.v[1] * b.v[1];
    p = p + a.v[2] * b.v[2];
    return wsum(p);
}
__device__ __forceinline__ float absmax3(const Lane3& a, int lane, int N) {
    float m = 0.f;
#pragma unroll
    for (int e = 0; e < NE3; ++e) if (3 * lane + e < N) m = fmaxf(m, fabsf(a.v[e]));
    return wmax(m);
}
__device__ __forceinline__ Lane3 axpy3(const Lane3& x, float a, const Lane3& d) {
    Lane3 r;
#pragma unroll
    for (int e = 0; e < NE3; ++e) r.v[e] = fmaf(a, d.v[e], x.v[e]);
    return r;
}

__device__ __forceinline__ OptScal fresh_state() {
    OptScal s;
    s.phase = PH_ENTRY; s.outer = 0; s.n_iter = 0; s.n_iter_total = 0; s.cur_evals = 0; s.func_evals = 0;
    s.ls_evals = 0; s.ls_iter = 0; s.ls_done = 0; s.insuf = 0; s.low = 0; s.high = 1;
    s.hist_n = 0; s.hist_head = 0; s.cache_valid = 0; s.has_prev_outer = 0; s.evals = 0; s.ref_evals = 0;
    s.pad0 = 0; s.pad1 = 0;
    s.t = P(0.0); s.t_prev = P(0.0); s.H_diag = P(1.0);
    s.loss = P(0.0); s.prev_loss = P(0.0); s.orig_loss = P(0.0); s.f_prev = P(0.0); s.ls_f0 = P(0.0);
    s.gtd_prev = P(0.0); s.ls_gtd0 = P(0.0); s.d_norm = P(0.0);
    s.br0 = s.br1 = s.bf0 = s.bf1 = s.bgtd0 = s.bgtd1 = P(0.0);
    s.prev_loss_outer = 0.0;
    return s;
}

// ---------------------------------------------------------------------------------------------
// Two-loop recursion (lbfgs_ls.py:322-341) on ONE wavefront, in blocks of 8 history pairs.  A single wavefront issues
// one instruction every 4 cycles whatever its kind, so the recursion is priced in instructions: per block
//   * 16 row loads (8 s rows, 8 y rows; lane l owns elements 3l..3l+2) from CONTIGUOUS history slots -- one uniform
//     base and immediate offsets; only the block that wraps around the ring or sticks out of the window computes
//     clamped slots row by row;
//   * the 8 dot products against the running vector, 3 FMAs each;
//   * ONE recursive-halving reduction of all 8 (wave_sum8_groups: 17 instructions) that leaves the sum of member c in
//     the 8 lanes of group c;
//   * the in-block coupling  s_c . (q - sum_{m<c} al_m y_m)  resolved lane-parallel on those groups with the stored
//     band of S^T Y (one 64-byte row per step, zeros where c <= m): multiply, v_readlane (the broadcast the axpy
//     needs anyway), FMA per step;
//   * the 8 axpys.
// Same arithmetic in exact terms as the sequential recursion (the order of the updates is the reference's; only the
// summation tree of each dot product differs).  Two blocks of look-ahead in registers cover the history's
// L2 / MALL latency.
template <int CTRL>
__device__ __forceinline__ float lb_dpp(float x) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xf, 0xf, true));
}
// in: v[c] = this lane's partial of sum c.  out: every lane of group g (lanes 8g..8g+7) holds the wave-wide sum g.
__device__ __forceinline__ float wave_sum8_groups(const float (&v)[8], const int lane) {
    // halves (stride 32): lanes 0-31 keep a, lanes 32-63 keep b
    auto halves = [](float a_, float b_) {
        const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a_), __float_as_uint(b_), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
    // rows of 16 (stride 16): rows 0, 2 keep a, rows 1, 3 keep b
    auto rows = [](float a_, float b_) {
        const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a_), __float_as_uint(b_), false, false);
        return __uint_as_float(r[0]) + __uint_as_float(r[1]); };
    // operand order chosen so that sum g ends in group g: rows of AB = sums 0, 2, 4, 6 and of CD = 1, 3, 5, 7
    const float A = halves(v[0], v[4]), B = halves(v[2], v[6]), C = halves(v[1], v[5]), D = halves(v[3], v[7]);
    const float AB = rows(A, B), CD = rows(C, D);
    // stride 8: lanes 0-7 of a row keep AB, lanes 8-15 keep CD
    const float tA = AB + lb_dpp<0x128>(AB), tC = CD + lb_dpp<0x128>(CD);          // row_ror:8
    float t = (lane & 8) ? tC : tA;
    t = t + lb_dpp<0x141>(t);          // row_half_mirror: l <-> 7 - l
    t = t + lb_dpp<0xB1>(t);           // quad_perm [1,0,3,2]
    t = t + lb_dpp<0x4E>(t);           // quad_perm [2,3,0,1]
    return t;
}

#define LB_ROWB (SFX_NVAR_MAX * 4)      // bytes per history row
// a lane's 3 elements of a history row as (pair, single): the 12-byte load lands in an even-aligned register triple, so
// the pair feeds v_pk_fma_f32 as it is (left to itself the compiler pairs elements 1, 2 and copies every row)
typedef float lb_v2 __attribute__((ext_vector_type(2)));
struct LbRow { lb_v2 a; float b; };
struct LbSet { LbRow mine[8]; LbRow all[8]; float bnd[7]; float ro, al; bool valid; };      // ro / al as loaded: the consumer masks them with `valid`
typedef float lb_v3 __attribute__((ext_vector_type(3)));
__device__ __forceinline__ LbRow lb_ldrow(const char* p) {
#ifdef SFX_HIST_NT
    // history rows are read once per direction and never shared: non-temporal, so that they do not push the tables every
    // frame of the XCD reads (blend-shape rows, VPoser weights) out of the 4-MB L2
    const lb_v3 u = __builtin_nontemporal_load(reinterpret_cast<const lb_v3*>(p));
#else
    const float3 u = *reinterpret_cast<const float3*>(p);
#endif
    LbRow r; r.a.x = u.x; r.a.y = u.y; r.b = u.z; return r;
}
__device__ __forceinline__ float lb_dotpart(const LbRow& x, const LbRow& y) { return fmaf(x.b, y.b, fmaf(x.a.y, y.a.y, x.a.x * y.a.x)); }
__device__ __forceinline__ void lb_axpy(LbRow& y, const float a, const LbRow& x) {
    const lb_v2 aa = {a, a};
    y.a = __builtin_elementwise_fma(aa, x.a, y.a); y.b = fmaf(a, x.b, y.b);
}

// rows of the block whose member 0 is logical index `base`, members base + DIR * c: always 8 CONSECUTIVE rows (the ring
// is followed by a mirror of its first 8 slots).  Members outside the window [0, n) get ro = al = 0: their rows are
// other slots of the ring (stale pairs or the zeros of the allocation -- finite), their steps are exact no-ops.
// mine / all: the two history arrays in the order the loop uses them; tab: the band table of this direction.  The rows
// the block touches first (mine) are requested last: one s_waitcnt at the dot products covers the whole set.
template <int DIR>
__device__ __forceinline__ void lb_load(LbSet& X, const int base, const int n, const int head, const float* hMine,
                                        const float* hAll, const float* tab, const float* ro, const float* s_alp,
                                        const bool want_al, const int lane, const int R) {
    const int grp = lane >> 3;
    int p = head + base; p = p >= R ? p - R : p;
    if (DIR < 0 && p < 7) p += R;
    const float* tb = tab + p * LB_BROW + 8 + grp;
#pragma unroll
    for (int m = 0; m < 7; ++m) X.bnd[m] = tb[DIR * m * LB_BROW - m];
    const int ig = base + DIR * grp;
    const bool valid = ig >= 0 && ig < n;
    // ro / al are handed over as loaded and masked where they are used (lb_two_loop): a select right behind its load is a
    // wait for that load, and in front of the 16 row requests it cost the prologue of loop 2 two serial memory round trips
    X.valid = valid;
    X.ro = ro[p + DIR * grp];
    X.al = 0.f;
    if (want_al) X.al = s_alp[ig];
    const char* ra = reinterpret_cast<const char*>(hAll) + (size_t)p * LB_ROWB + 12 * lane;
    const char* rm = reinterpret_cast<const char*>(hMine) + (size_t)p * LB_ROWB + 12 * lane;
#pragma unroll
    for (int c = 0; c < 8; ++c) X.all[c] = lb_ldrow(ra + DIR * c * LB_ROWB);
#pragma unroll
    for (int c = 0; c < 8; ++c) X.mine[c] = lb_ldrow(rm + DIR * c * LB_ROWB);
}

// A new curvature pair (lbfgs_ls.py:312-320: the oldest one leaves when the window is full) with what the blocked
// recursion keeps per pair: ro, the band entries s_(k-th predecessor) . y_new -- the syt row of the new pair and entry k
// of the k-th predecessor's syb row -- and the mirror of slots 0..7 behind the ring.
__device__ __forceinline__ void lb_push_pair(float* hY, float* hS, OptState* gst, int& hist_n, int& hist_head, const Lane3& y,
                                             const Lane3& sv, const float ys, const int lane, const int hist_cap = SFX_HIST,
                                             const int R = SFX_HIST) {
    if (hist_n >= hist_cap) { hist_head = hist_head + 1 >= R ? 0 : hist_head + 1; hist_n -= 1; }      // (history_size <= R, the ring's slots: 100 unless history_size asks for more)
    int ph = hist_head + hist_n; ph = ph >= R ? ph - R : ph;
    st3_full(hY + (size_t)ph * SFX_NVAR_MAX, y, lane);
    st3_full(hS + (size_t)ph * SFX_NVAR_MAX, sv, lane);
    const float ro = 1.0f / ys;
    if (lane == 0) gst->ro[ph] = ro;
    if (ph < 8) {
        st3_full(hY + (size_t)(ph + R) * SFX_NVAR_MAX, y, lane);
        st3_full(hS + (size_t)(ph + R) * SFX_NVAR_MAX, sv, lane);
        if (lane == 0) gst->ro[ph + R] = ro;
    }
    hist_n += 1;
    const int n1 = __builtin_amdgcn_readfirstlane(hist_n), hd1 = __builtin_amdgcn_readfirstlane(hist_head);
    Lane3 SP[7];
#pragma unroll
    for (int k = 1; k < 8; ++k) {
        const int i_ = n1 - 1 - k;
        int t_ = hd1 + (i_ >= 0 ? i_ : 0); t_ = t_ >= R ? t_ - R : t_;
        SP[k - 1] = ld3_raw(hS, (unsigned)t_ * SFX_NVAR_MAX, lane);
    }
    float eb[8];
    eb[0] = 0.f;
#pragma unroll
    for (int k = 1; k < 8; ++k) eb[k] = fmaf(SP[k - 1].v[2], y.v[2], fmaf(SP[k - 1].v[1], y.v[1], SP[k - 1].v[0] * y.v[0]));
    const float e = wave_sum8_groups(eb, lane);      // group k: s_(k-th predecessor) . y_new
    const int k = lane >> 3, i_ = n1 - 1 - k;
    int t_ = hd1 + (i_ >= 0 ? i_ : 0); t_ = t_ >= R ? t_ - R : t_;
    const float ev = (k >= 1 && i_ >= 0) ? e : 0.f;   // (no such predecessor: a finite 0)
    if ((lane & 7) == 0) {
        gst->syt[ph * LB_BROW + 8 + k] = ev;
        gst->syb[ph * LB_BROW + 8 + k] = 0.f;         // successors of the new pair do not exist yet
        if (ph < 8) { gst->syt[(ph + R) * LB_BROW + 8 + k] = ev; gst->syb[(ph + R) * LB_BROW + 8 + k] = 0.f; }
        if (k >= 1 && i_ >= 0) {
            gst->syb[t_ * LB_BROW + 8 + k] = ev;
            if (t_ < 8) gst->syb[(t_ + R) * LB_BROW + 8 + k] = ev;
        }
    }
}

template <int SETS>
__device__ __forceinline__ Lane3 lb_two_loop(const float* hS, const float* hY, const OptState* gst, float* s_al,
                                             const int n_, const int head_, const float hd, const Lane3 q_in, const int lane,
                                             const int R = SFX_HIST) {
    const int n = __builtin_amdgcn_readfirstlane(n_), head = __builtin_amdgcn_readfirstlane(head_);
    float* s_alp = s_al + 8;        // members below index 0 of the last block land in the padding
    const int grp = lane >> 3;
    const int nb = max(1, (n + 7) >> 3);        // block steps per loop (the old loops ran one step for an empty window too)
#define LB_RL(v, l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (l)))
    // SETS register sets of history rows: the block in use + (SETS - 1) blocks of look-ahead (3 where the registers are
    // there -- the tick kernels; 2 inside the persistent per-frame kernel, which would spill)
    static_assert(SETS == 2 || SETS == 3, "register sets");
    LbSet A, B, C;
    LbRow q; q.a.x = q_in.v[0]; q.a.y = q_in.v[1]; q.b = q_in.v[2];
    // ---- loop 1: newest -> oldest;  al_i = ro_i s_i . q;  q -= al_i y_i
    auto down = [&](const LbSet& X, const int base) {
        float part[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) part[c] = lb_dotpart(X.mine[c], q);
        float acc = wave_sum8_groups(part, lane);
        const float ro = X.valid ? X.ro : 0.f;
        float al[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            al[m] = LB_RL(acc * ro, 8 * m);
            if (m < 7) acc = fmaf(-al[m], X.bnd[m], acc);
        }
        s_alp[base - grp] = acc * ro;          // (members c <= m see zeros of the band: acc[c] is final after step c)
#pragma unroll
        for (int c = 0; c < 8; ++c) lb_axpy(q, -al[c], X.all[c]);
    };
#define LB_LD1(X, b_) lb_load<-1>(X, (b_), n, head, hS, hY, gst->syt, gst->ro, s_alp, false, lane, R)
    // The look-ahead loads are UNCONDITIONAL (a block past the end of the window re-reads block 0; it is never used): with
    // `if (i0 >= 16) load` the number of loads in flight at the next dot products depended on a branch, and the compiler then
    // waits for the shorter path's count -- vmcnt(0): every block paid its own memory round trip and the look-ahead bought
    // nothing (round 4, read off the ISA; the same holds for stores left pending by the pair push: gfx9 counts them in vmcnt).
    // (and the loads must stay where they are written: left alone, the compiler sinks the look-ahead loads of two steps into the
    //  third one, next to their use -- LB_PIN, an empty asm that memory operations do not cross)
#define LB_PIN() asm volatile("" ::: "memory")
    __builtin_amdgcn_s_waitcnt(0x0F70);      // (kept: without it the tick launch is 0.2 us slower, A/B in one session)
    {
        int i0 = n - 1, k = 0;
        // ONE exit per loop: with `if (i0 < 0) break` after every step the unified loop exit gave the header an edge from each
        // step, and the wait counts at the header and in the third step were those of the shortest such path -- vmcnt(8) and
        // vmcnt(23) where 48 loads may stay in flight: two steps out of three waited for the look-ahead of the step before
        // (round 4, read off the ISA).  Whole rounds of SETS steps in the loop, the 1 .. SETS - 1 left over behind it.
        if constexpr (SETS == 3) {
        LB_LD1(A, i0); LB_LD1(B, max(i0 - 8, 0));
        for (; k + 3 <= nb; k += 3) {
            LB_LD1(C, max(i0 - 16, 0)); LB_PIN(); down(A, i0); LB_PIN(); i0 -= 8;
            LB_LD1(A, max(i0 - 16, 0)); LB_PIN(); down(B, i0); LB_PIN(); i0 -= 8;
            LB_LD1(B, max(i0 - 16, 0)); LB_PIN(); down(C, i0); LB_PIN(); i0 -= 8;
        }
        if (k < nb) { down(A, i0); LB_PIN(); i0 -= 8; if (k + 1 < nb) { down(B, i0); LB_PIN(); } }
        } else {
        LB_LD1(A, i0);
        for (; k + 2 <= nb; k += 2) {
            LB_LD1(B, max(i0 - 8, 0)); LB_PIN(); down(A, i0); LB_PIN(); i0 -= 8;
            LB_LD1(A, max(i0 - 8, 0)); LB_PIN(); down(B, i0); LB_PIN(); i0 -= 8;
        }
        if (k < nb) { down(A, i0); LB_PIN(); }
        }
    }
#undef LB_LD1
    LB_SYNC();      // alphas (LDS) are read back below
    LbRow r; r.a = q.a * hd; r.b = q.b * hd;
    // ---- loop 2: oldest -> newest;  be_i = ro_i y_i . r;  r += (al_i - be_i) s_i
    auto up = [&](const LbSet& X) {
        float part[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) part[c] = lb_dotpart(X.mine[c], r);
        float acc = wave_sum8_groups(part, lane);
        const float ro = X.valid ? X.ro : 0.f, alv = X.valid ? X.al : 0.f;
        float cc[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            cc[m] = LB_RL(alv - acc * ro, 8 * m);
            if (m < 7) acc = fmaf(cc[m], X.bnd[m], acc);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) lb_axpy(r, cc[c], X.all[c]);
    };
#define LB_LD2(X, b_) lb_load<1>(X, (b_), n, head, hY, hS, gst->syb, gst->ro, s_alp, true, lane, R)
    {
        int i0 = 0, k = 0;
        const int last = max(n - 1, 0);
        if constexpr (SETS == 3) {
        LB_LD2(A, i0); LB_LD2(B, min(i0 + 8, last));
        for (; k + 3 <= nb; k += 3) {
            LB_LD2(C, min(i0 + 16, last)); LB_PIN(); up(A); LB_PIN(); i0 += 8;
            LB_LD2(A, min(i0 + 16, last)); LB_PIN(); up(B); LB_PIN(); i0 += 8;
            LB_LD2(B, min(i0 + 16, last)); LB_PIN(); up(C); LB_PIN(); i0 += 8;
        }
        if (k < nb) { up(A); LB_PIN(); if (k + 1 < nb) { up(B); LB_PIN(); } }
        } else {
        LB_LD2(A, i0);
        for (; k + 2 <= nb; k += 2) {
            LB_LD2(B, min(i0 + 8, last)); LB_PIN(); up(A); LB_PIN(); i0 += 8;
            LB_LD2(A, min(i0 + 8, last)); LB_PIN(); up(B); LB_PIN(); i0 += 8;
        }
        if (k < nb) { up(A); LB_PIN(); }
        }
    }
#undef LB_LD2
#undef LB_PIN
#undef LB_RL
    Lane3 out; out.v[0] = r.a.x; out.v[1] = r.a.y; out.v[2] = r.b;
    return out;
}

// One tick of frame b's optimiser, executed by ONE wavefront (lanes 0..63).  f_in / g_in: the
// closure result (global D.f/D.g, or LDS copies in fused kernels).  s_al[SFX_HIST_MAX + 2 * LB_BS], s_state and
// s_work[2048] (the tick's working copies; may alias any LDS that is dead during the tick)
// are LDS scratch owned by the caller.
// PF: the caller has the working set (vectors, X, Xt in s_work, the scalar state in s_state) and both variable lists (vls
// points to LDS copies) in place already -- k_tick_dense requests them with its entry's LDS-DMA batch, a whole loss /
// adjoint pass before the tick wants them -- and takes the frame's stage after the tick from *stage_out.
template <int SETS, bool PF = false>
__device__ __forceinline__ void lbfgs_tick_wave0(const DevModel& M, const BatchDev& D, const VarList* __restrict__ vls,
                                                 int first_stage, int last_stage, int init, int step_mode,
                                                 const int b, const int lane, float* s_al, OptScal& s_state, float* s_work,
                                                 const float* f_src, const float* g_src, int* stage_out = nullptr) {
    const BatchCfgDev& C = D.cfg;
    OptState* gst = reinterpret_cast<OptState*>(D.opt) + b;
    int stage = D.stage[b];
    float* Xg = D.X + (size_t)b * SFX_NPAR_MAX;          // global homes of the parameters, the trial point and the vectors;
    float* Xtg = D.Xt + (size_t)b * SFX_NPAR_MAX;        // a tick works on LDS copies (below)
    float* vecg = D.vec + (size_t)b * NVEC * SFX_NVAR_MAX;
#ifdef SFX_RING_CONST                                   // (measurement: the ring's size as the compile-time constant it was until round 5)
    const int R = SFX_HIST, hrows = R + 8;
#else
    const int R = C.hist_ring, hrows = R + 8;          // slots of the history ring (SFX_HIST unless history_size is larger)
#endif
    float* hY = D.hist + (size_t)b * 2 * hrows * SFX_NVAR_MAX;
    float* hS = hY + (size_t)hrows * SFX_NVAR_MAX;
#define VEC(k) (vec + (k) * SFX_NVAR_MAX)

    // ---------------------------------------------------------------- (re)initialisation
    if (init == 1) {
        const OptScal s = fresh_state();
        stage = first_stage;
        if (lane == 0) {
            D.stage[b] = stage;
            gst->s = s;            // ro[] needs no initialisation (guarded by hist_n)
            D.orient_pass[b] = 0;
            int both = 0;
            if (C.side_thsh > 0.f) {     // torch.dist of the 2-D shoulders (fit_single_frame.py:461-463)
                const float* g2 = D.gt + (size_t)b * M.K * 2;
                const float dx = g2[2 * C.lsh] - g2[2 * C.rsh], dy = g2[2 * C.lsh + 1] - g2[2 * C.rsh + 1];
                both = sqrtf(dx * dx + dy * dy) < C.side_thsh;
            }
            D.try_both[b] = both;
        }
        for (int i = lane; i < hrows * LB_BROW; i += 64) { gst->syb[i] = 0.f; gst->syt[i] = 0.f; }     // (columns 0..8 stay zero)
        for (int q = lane; q < 1 + SFX_MAX_STAGES; q += 64) {
            if (q >= first_stage + 1 && q <= last_stage + 1) {
                D.stage_evals[(size_t)b * (1 + SFX_MAX_STAGES) + q] = 0;
                D.stage_ref_evals[(size_t)b * (1 + SFX_MAX_STAGES) + q] = 0;
            }
        }
        for (int i = lane; i < SFX_NPAR_MAX; i += 64) Xtg[i] = Xg[i];
        return;
    }
    if (init == 2) {           // resume after a pause (optimizer.step granularity): keep the state
        if (stage >= 900 && lane == 0) D.stage[b] = stage - 1000;   // camera stage -1 pauses as 999
        for (int i = lane; i < SFX_NPAR_MAX; i += 64) Xtg[i] = Xg[i];
        return;
    }
    if (stage > last_stage) { if (PF && lane == 0) *stage_out = stage; return; }
    // debug: frame 0 accumulates shader-clock deltas between TMARKs into dbg[32+i], tick count in dbg[63]
    long long tm_prev = 0;
#define TMARK(i) do { if (D.dbg && b == 0 && lane == 0) { const long long c_ = clock64();                 \
        if (i) D.dbg[32 + (i)] += c_ - tm_prev; else D.dbg[63] += 1; tm_prev = c_; } } while (0)
    TMARK(0);

    // Working set in LDS for the duration of the tick, fetched in ONE round trip and written back at the end: the scalar
    // state (every lane reads -- broadcast -- and writes -- identical values -- the same words, so the single wavefront
    // stays uniform), the NVEC optimiser vectors, the parameters X and the trial point Xt.  The state machine below
    // stores a vector and reads it back a few lines later many times per tick: through global memory each of those is
    // an L2 round trip plus a fence (~10 of them in the common path), in LDS ~100 cycles.
    static_assert(NVEC * SFX_NVAR_MAX + 2 * SFX_NPAR_MAX <= 2048 && SFX_NPAR_MAX == 4 * 64, "s_work: 2048 floats");
    float* vec = s_work;
    float* X = s_work + NVEC * SFX_NVAR_MAX;
    float* Xt = X + SFX_NPAR_MAX;
    const VarList& vl = vls[stage < 0 ? 0 : 1];
    int N = vl.n;
    int idx3[NE3];           // this lane's parameter slots (vl.idx: 6 bytes per lane, once per tick)
    if constexpr (PF) {
#pragma unroll
        for (int e = 0; e < NE3; ++e) idx3[e] = vl.idx[3 * lane + e];
    } else {
        Lane3 wv_[NVEC];
#pragma unroll
        for (int k = 0; k < NVEC; ++k) wv_[k] = ld3_raw(vecg, (unsigned)k * SFX_NVAR_MAX, lane);
        const float4 x4 = reinterpret_cast<const float4*>(Xg)[lane], xt4 = reinterpret_cast<const float4*>(Xtg)[lane];
#pragma unroll
        for (int e = 0; e < NE3; ++e) idx3[e] = vl.idx[3 * lane + e];
        if (lane == 0) s_state = gst->s;
#pragma unroll
        for (int k = 0; k < NVEC; ++k) st3_full(vec + k * SFX_NVAR_MAX, wv_[k], lane);
        reinterpret_cast<float4*>(X)[lane] = x4; reinterpret_cast<float4*>(Xt)[lane] = xt4;
    }
    LB_SYNC();
    OptScal& s = s_state;
    const double tol_change = C.tol_change, tol_grad = C.tol_grad;      // (LBFGS defaults 1e-9 / 1e-5: optim_factory.py:27-65 never changes them)
    const int max_iter = C.lbfgs_max_iter, max_eval = C.max_eval, max_ls = 25;

    // incoming evaluation
    const Sc f_in = P((double)*f_src);
    Lane3 g_in = ld3(g_src, lane, N);
    int glast_cached = 0;
    s.evals += 1; s.ref_evals += 1;

    auto gather_x = [X, lane, N, &idx3]() { Lane3 r;
        for (int e = 0; e < NE3; ++e) { const int i = 3 * lane + e; r.v[e] = (i < N) ? X[idx3[e]] : 0.f; } return r; };
    auto write_trial = [X, Xt, vec, lane, N, &idx3](Sc t) {
        const Lane3 xi = ld3(VEC(VEC_XINIT), lane, N), d = ld3(VEC(VEC_D), lane, N);
        const Lane3 xt = axpy3(xi, (float)t.v, d);
        for (int i = lane; i < SFX_NPAR_MAX; i += 64) Xt[i] = X[i];
        LB_SYNC();
        for (int e = 0; e < NE3; ++e) { const int i = 3 * lane + e; if (i < N) Xt[idx3[e]] = xt.v[e]; }
    };
#define armijo_fail(f_new, t) sc_gt((f_new), sc_add(s.ls_f0, sc_mul(sc_mul(P(1e-4), (t)), s.ls_gtd0)))
#define curv_ok(gtd_new) sc_le(sc_abs(gtd_new), sc_mul(P(-0.9), s.ls_gtd0))
    // bracket set-up: gradient slot 0 = G0 (a Lane3 value), slot 1 = the incoming gradient
#define start_zoom(b0, b1, f0, f1, G0, gd0, gd1, done) do {                                              \
        s.br0 = (b0); s.br1 = (b1); s.bf0 = (f0); s.bf1 = (f1); s.bgtd0 = (gd0); s.bgtd1 = (gd1);        \
        st3(VEC(VEC_BG0), (G0), lane, N); st3(VEC(VEC_BG1), g_in, lane, N);                               \
        s.ls_done = (done) ? 1 : 0; s.insuf = 0;                                                         \
        if (sc_le(s.bf0, s.bf1)) { s.low = 0; s.high = 1; } else { s.low = 1; s.high = 0; } } while (0)

    // optimiser trace (sfx_batch_trace): one record per finished line search / LBFGS.step / stage
#define TRACE(ty, a_, b_, c_) do { if (D.trace && D.trace_n && lane == 0) { const int n_ = D.trace_n[b];                         \
        if (n_ < D.trace_cap) D.trace[(size_t)b * D.trace_cap + n_] = make_float4((float)(ty), (float)(a_), (float)(b_), (float)(c_)); \
        D.trace_n[b] = n_ + 1; } } while (0)
    if (D.trace && D.trace_evals) {      // (debug) every evaluation: trial step, loss, |g|inf  (the reduction is wave-wide: outside TRACE)
        const float ginf_ = absmax3(g_in, lane, N);
        TRACE(3, s.t.v, f_in.v, ginf_);
    }
    int act = A_NONE;
    TMARK(1);
    // ---------------------------------------------------------------- consume the evaluation
    if (s.phase == PH_ENTRY) {
        st3(VEC(VEC_G), g_in, lane, N);
        s.loss = f_in;
        act = A_ENTRY;
    } else if (s.phase == PH_BRACKET) {
        s.ls_evals += 1;
        const Sc t = s.t;
        const Lane3 d = ld3(VEC(VEC_D), lane, N);
        const Sc gtd_new = T(dot3(g_in, d));
        if (s.ls_iter == max_ls) {
            start_zoom(P(0.0), t, s.ls_f0, f_in, ld3(VEC(VEC_LSG0), lane, N), s.ls_gtd0, gtd_new, false);
            act = A_ZOOM_NEXT;
        } else if (armijo_fail(f_in, t) || (s.ls_iter > 1 && sc_ge(f_in, s.f_prev))) {
            start_zoom(s.t_prev, t, s.f_prev, f_in, ld3(VEC(VEC_GPREV), lane, N), s.gtd_prev, gtd_new, false);
            act = A_ZOOM_NEXT;
        } else if (curv_ok(gtd_new)) {
            // single-point bracket: the point itself is accepted
            start_zoom(t, t, f_in, f_in, g_in, gtd_new, gtd_new, true);
            s.low = 0; s.high = 1;
            act = A_ZOOM_NEXT;
        } else if (sc_ge(gtd_new, P(0.0))) {
            start_zoom(s.t_prev, t, s.f_prev, f_in, ld3(VEC(VEC_GPREV), lane, N), s.gtd_prev, gtd_new, false);
            act = A_ZOOM_NEXT;
        } else {
            const Sc lo = sc_add(t, sc_mul(P(0.01), sc_sub(t, s.t_prev)));
            const Sc hi = sc_mul(t, P(10.0));
            const Sc t_next = cubic_interpolate(s.t_prev, s.f_prev, s.gtd_prev, t, f_in, gtd_new, true, lo, hi);
            s.t_prev = t; s.f_prev = f_in; s.gtd_prev = gtd_new;
            st3(VEC(VEC_GPREV), g_in, lane, N);
            s.t = t_next;
            s.ls_iter += 1;
            write_trial(t_next);
            act = A_NONE;
        }
    } else {   // PH_ZOOM
        s.ls_evals += 1; s.ls_iter += 1;
        const Sc t = s.t;
        const Lane3 d = ld3(VEC(VEC_D), lane, N);
        const Sc gtd_new = T(dot3(g_in, d));
        const int lo = s.low, hi = s.high;
        if (armijo_fail(f_in, t) || sc_ge(f_in, (lo ? s.bf1 : s.bf0))) {
            { const Sc v_ = t; if (hi) s.br1 = v_; else s.br0 = v_; } { const Sc v_ = f_in; if (hi) s.bf1 = v_; else s.bf0 = v_; } { const Sc v_ = gtd_new; if (hi) s.bgtd1 = v_; else s.bgtd0 = v_; }
            st3(VEC(hi ? VEC_BG1 : VEC_BG0), g_in, lane, N);
            if (sc_le(s.bf0, s.bf1)) { s.low = 0; s.high = 1; } else { s.low = 1; s.high = 0; }
        } else {
            if (curv_ok(gtd_new)) s.ls_done = 1;
            else if (sc_ge(sc_mul(gtd_new, sc_sub((hi ? s.br1 : s.br0), (lo ? s.br1 : s.br0))), P(0.0))) {
                { const Sc v_ = (lo ? s.br1 : s.br0); if (hi) s.br1 = v_; else s.br0 = v_; } { const Sc v_ = (lo ? s.bf1 : s.bf0); if (hi) s.bf1 = v_; else s.bf0 = v_; } { const Sc v_ = (lo ? s.bgtd1 : s.bgtd0); if (hi) s.bgtd1 = v_; else s.bgtd0 = v_; }
                const Lane3 tmp = ld3(VEC(lo ? VEC_BG1 : VEC_BG0), lane, N);
                st3(VEC(hi ? VEC_BG1 : VEC_BG0), tmp, lane, N);
            }
            { const Sc v_ = t; if (lo) s.br1 = v_; else s.br0 = v_; } { const Sc v_ = f_in; if (lo) s.bf1 = v_; else s.bf0 = v_; } { const Sc v_ = gtd_new; if (lo) s.bgtd1 = v_; else s.bgtd0 = v_; }
            st3(VEC(lo ? VEC_BG1 : VEC_BG0), g_in, lane, N);
        }
        if (sc_lt(sc_mul(sc_abs(sc_sub(s.br1, s.br0)), s.d_norm), P(tol_change))) act = A_FINISH_LS;
        else act = A_ZOOM_NEXT;
    }

    TMARK(2);
    // ---------------------------------------------------------------- run until an evaluation is needed
    int guard = 0;
    while (act != A_NONE && guard++ < 4096) {
        switch (act) {
        case A_ENTRY: {          // LBFGS.step prologue (lbfgs_ls.py:279-290); (loss, G) hold f, g at x
            s.cache_valid = 1;
            s.func_evals += 1;
            s.orig_loss = s.loss;
            s.cur_evals = 1; s.n_iter = 0;
            const Lane3 g = ld3(VEC(VEC_G), lane, N);
            if (sc_le(T(absmax3(g, lane, N)), P(tol_grad))) act = A_END_STEP;
            else act = A_ITER_HEAD;
            break;
        }
        case A_ITER_HEAD: {      // direction + first trial point (lbfgs_ls.py:304-397)
            if (!(s.n_iter < max_iter)) { act = A_END_STEP; break; }
            s.n_iter += 1; s.n_iter_total += 1;
            const Lane3 g = ld3(VEC(VEC_G), lane, N);
            Lane3 d;
            if (s.n_iter_total == 1) {
                for (int e = 0; e < NE3; ++e) d.v[e] = -g.v[e];
                s.hist_n = 0; s.hist_head = 0; s.H_diag = P(1.0);
            } else {
                const Lane3 pg = ld3(VEC(VEC_PREVG), lane, N), dold = ld3(VEC(VEC_D), lane, N);
                Lane3 y, sv;
                const float tf = (float)s.t.v;
                for (int e = 0; e < NE3; ++e) { y.v[e] = g.v[e] - pg.v[e]; sv.v[e] = dold.v[e] * tf; }
                const float ys = dot3(y, sv);
                if (ys > 1e-10f) {
                    int hn = s.hist_n, hh = s.hist_head;
                    lb_push_pair(hY, hS, gst, hn, hh, y, sv, ys, lane, C.hist_cap, R);
                    s.hist_n = hn; s.hist_head = hh;
                    s.H_diag = T(ys / dot3(y, y));
                }
                LB_SYNC();
                TMARK(3);
                Lane3 r;
                {
                    const int n = __builtin_amdgcn_readfirstlane(s.hist_n);
                    const float hd = (float)s.H_diag.v;
                    Lane3 q;
                    for (int e = 0; e < NE3; ++e) q.v[e] = -g.v[e];
                    if (n == 0) {
                        for (int e = 0; e < NE3; ++e) r.v[e] = q.v[e] * hd;
                    } else {
                        r = lb_two_loop<SETS>(hS, hY, gst, s_al, n, s.hist_head, hd, q, lane, R);
                    }
                }
                TMARK(5);
                d = r;
            }
            st3(VEC(VEC_D), d, lane, N);
            st3(VEC(VEC_PREVG), g, lane, N);
            s.prev_loss = s.loss;
            Sc t;
            if (s.n_iter_total == 1) {
                float asum = 0.f;
                for (int e = 0; e < NE3; ++e) asum = asum + fabsf(g.v[e]);
                asum = wsum(asum);
                t = sc_mul(sc_pmin(P(1.0), sc_div(P(1.0), T(asum))), P((double)C.lr));
            } else t = P((double)C.lr);
            s.t = t;
            const Sc gtd = T(dot3(g, d));
            if (sc_gt(gtd, P(-tol_change))) { act = A_END_STEP; break; }
            // strong-Wolfe set-up (lbfgs_ls.py:39-52)
            const Lane3 xi = gather_x();
            st3(VEC(VEC_XINIT), xi, lane, N);
            st3(VEC(VEC_LSG0), g, lane, N);
            st3(VEC(VEC_GPREV), g, lane, N);
            s.ls_f0 = s.loss; s.ls_gtd0 = gtd;
            s.d_norm = T(absmax3(d, lane, N));
            s.ls_evals = 0; s.ls_iter = 0;
            s.t_prev = P(0.0); s.f_prev = s.loss; s.gtd_prev = gtd;
            s.phase = PH_BRACKET;
            LB_SYNC();
            write_trial(t);
            act = A_NONE;
            break;
        }
        case A_ZOOM_NEXT: {      // next zoom trial (lbfgs_ls.py:108-131)
            if (s.ls_done || !(s.ls_iter < max_iter)) { act = A_FINISH_LS; break; }
            Sc t = cubic_interpolate(s.br0, s.bf0, s.bgtd0, s.br1, s.bf1, s.bgtd1, false, P(0), P(0));
            const Sc bmax = sc_gt(s.br1, s.br0) ? s.br1 : s.br0;
            const Sc bmin = sc_lt(s.br1, s.br0) ? s.br1 : s.br0;
            const Sc eps = sc_mul(P(0.1), sc_sub(bmax, bmin));
            if (sc_lt(sc_pmin(sc_sub(bmax, t), sc_sub(t, bmin)), eps)) {
                if (s.insuf || sc_ge(t, bmax) || sc_le(t, bmin)) {
                    if (sc_lt(sc_abs(sc_sub(t, bmax)), sc_abs(sc_sub(t, bmin)))) t = sc_sub(bmax, eps);
                    else t = sc_add(bmin, eps);
                    s.insuf = 0;
                } else s.insuf = 1;
            } else s.insuf = 0;
            s.t = t;
            s.phase = PH_ZOOM;
            write_trial(t);
            act = A_NONE;
            break;
        }
        case A_FINISH_LS: {      // accept the lowest bracket end (lbfgs_ls.py:163-167,398-434)
            const int lo = s.low;
            const Sc t = (lo ? s.br1 : s.br0);
            s.loss = (lo ? s.bf1 : s.bf0); s.t = t;
            const Lane3 g = ld3(VEC(lo ? VEC_BG1 : VEC_BG0), lane, N);
            st3(VEC(VEC_G), g, lane, N);
            const Lane3 xi = ld3(VEC(VEC_XINIT), lane, N), d = ld3(VEC(VEC_D), lane, N);
            const Lane3 xn = axpy3(xi, (float)t.v, d);
            for (int e = 0; e < NE3; ++e) { const int i = 3 * lane + e; if (i < N) X[idx3[e]] = xn.v[e]; }
            s.cache_valid = 1;
            const bool opt_cond = sc_le(T(absmax3(g, lane, N)), P(tol_grad));
            s.cur_evals += s.ls_evals; s.func_evals += s.ls_evals;
            TRACE(0, t.v, s.loss.v, s.ls_evals);
            if (s.n_iter == max_iter) { act = A_END_STEP; break; }
            if (s.cur_evals >= max_eval) { act = A_END_STEP; break; }
            if (opt_cond) { act = A_END_STEP; break; }
            Lane3 stp;
            const float tf = (float)t.v;
            for (int e = 0; e < NE3; ++e) stp.v[e] = d.v[e] * tf;
            if (sc_le(T(absmax3(stp, lane, N)), P(tol_change))) { act = A_END_STEP; break; }
            if (sc_lt(sc_abs(sc_sub(s.loss, s.prev_loss)), P(tol_change))) { act = A_END_STEP; break; }
            LB_SYNC();
            act = A_ITER_HEAD;
            break;
        }
        case A_END_STEP: {       // run_fitting bookkeeping (fitting.py:175-217)
            const double loss = s.orig_loss.v;
            TRACE(1, loss, s.func_evals, s.n_iter_total);
            if (step_mode) {         // one LBFGS.step per call: hand control back, keep the state
                if (lane == 0) {
                    D.stage_loss[(size_t)b * (1 + SFX_MAX_STAGES) + stage + 1] = (float)loss;
                    D.stage[b] = stage + 1000;
                }
                s.phase = PH_ENTRY;
                for (int i = lane; i < SFX_NPAR_MAX; i += 64) Xt[i] = X[i];
                act = A_NONE;
                break;
            }
            bool stop = false;
            if (isnan(loss) || isinf(loss)) stop = true;
            if (!stop && s.outer > 0 && s.has_prev_outer && C.ftol > 0.0) {
                const double pv = s.prev_loss_outer;
                const double den = fmax(fmax(fabs(pv), fabs(loss)), 1.0);
                if ((pv - loss) / den <= C.ftol) stop = true;
            }
            if (!stop) {
                const Lane3 gl = glast_cached ? ld3(VEC(VEC_G), lane, N) : g_in;
                bool all_small = true;
                for (int gi = 0; gi < vl.ngroups; ++gi) {
                    if (!vl.g_has[gi]) continue;
                    float m = -INFINITY;
                    for (int e = 0; e < NE3; ++e) {
                        const int i = 3 * lane + e;
                        if (i >= vl.g_off[gi] && i < vl.g_off[gi] + vl.g_len[gi]) m = fmaxf(m, gl.v[e]);
                    }
                    m = wmax(m);
                    if (!((double)fabsf(m) < C.gtol)) all_small = false;
                }
                if (all_small) stop = true;
            }
            if (!stop) {
                s.prev_loss_outer = loss; s.has_prev_outer = 1;
                s.outer += 1;
                if (s.outer >= C.maxiters) stop = true;
            }
            if (stop) { act = A_FINISH_STAGE; break; }
            s.phase = PH_ENTRY;
            if (C.reuse && s.cache_valid) {
                glast_cached = 1;
                s.ref_evals += 1;
                act = A_ENTRY;          // (loss, G) already hold f, g at the unchanged x
            } else {
                for (int i = lane; i < SFX_NPAR_MAX; i += 64) Xt[i] = X[i];
                act = A_NONE;
            }
            break;
        }
        case A_FINISH_STAGE: {
            const int slot = stage + 1;     // camera stage -> 0
            const size_t so = (size_t)b * (1 + SFX_MAX_STAGES);
            const int pass = D.orient_pass[b];
            const float res = s.has_prev_outer ? (float)s.prev_loss_outer : __int_as_float(0x7fc00000);
            TRACE(2, res, s.evals, stage);
            if (lane == 0) {
                (pass ? D.stage_loss2 : D.stage_loss)[so + slot] = res;
                D.stage_evals[so + slot] += s.evals;
                D.stage_ref_evals[so + slot] += s.ref_evals;
            }
            // camera stage done: remember the orientation the flipped candidate is derived from
            if (stage < 0 && lane < 3) D.gocam[(size_t)b * 4 + lane] = X[D.L.go + lane];
            stage += 1;
            // side view (fit_single_frame.py:527-551): second fit from the orientation rotated by pi
            // about y; pose embedding and camera translation continue from the first fit, every
            // other body parameter is zeroed; the lower final loss wins (:662-667)
            if (stage == C.n_stages && last_stage == C.n_stages - 1 && D.try_both[b]) {
                float* X0 = D.X0 + (size_t)b * SFX_NPAR_MAX;
                LB_SYNC();
                if (pass == 0) {
                    for (int i = lane; i < SFX_NPAR_MAX; i += 64) X0[i] = X[i];
                    LB_SYNC();
                    const ParLayout& L = D.L;
                    for (int i = lane; i < L.npar; i += 64) {
                        const bool keep = (i >= L.cam_t && i < L.cam_t + 3) || (i >= L.emb && i < L.emb + L.NEMB);
                        if (!keep) X[i] = 0.f;
                    }
                    LB_SYNC();
                    if (lane == 0) {
                        float fl[3];
                        flipped_orientation(D.gocam + (size_t)b * 4, fl);
                        X[L.go] = fl[0]; X[L.go + 1] = fl[1]; X[L.go + 2] = fl[2];
                        D.orient_pass[b] = 1;
                    }
                    if (L.has_bodyp && lane < 63) X[L.bodyp + lane] = X[L.emb + lane];
                    LB_SYNC();
                    stage = 0;
                } else {
                    const float l0 = D.stage_loss[so + C.n_stages], l1 = res;
                    if (l0 < l1) {                       // first orientation wins: restore it
                        for (int i = lane; i < SFX_NPAR_MAX; i += 64) X[i] = X0[i];
                    } else if (lane == 0) {
                        for (int q = 1; q <= C.n_stages; ++q) D.stage_loss[so + q] = D.stage_loss2[so + q];
                    }
                    if (lane == 0) D.orient_pass[b] = 2;
                    LB_SYNC();
                }
            }
            if (lane == 0) D.stage[b] = stage;
            s = fresh_state();
            for (int i = lane; i < SFX_NPAR_MAX; i += 64) Xt[i] = X[i];
            act = A_NONE;
            break;
        }
        default: act = A_NONE; break;
        }
    }
    TMARK(6);
    LB_SYNC();
    // write the working set back
#pragma unroll
    for (int k = 0; k < NVEC; ++k) st3_full(vecg + k * SFX_NVAR_MAX, ld3_raw(vec, (unsigned)k * SFX_NVAR_MAX, lane), lane);
    reinterpret_cast<float4*>(Xg)[lane] = reinterpret_cast<const float4*>(X)[lane];
    reinterpret_cast<float4*>(Xtg)[lane] = reinterpret_cast<const float4*>(Xt)[lane];
    if (lane == 0) gst->s = s_state;     // ro[] and the band tables are written in place
    if (PF && lane == 0) *stage_out = stage;
    TMARK(7);
#undef TMARK
#undef TRACE
#undef VEC
}

// Entry for a workgroup: wavefront 0 runs the state machine, the others pass through.  SETS: see lb_two_loop.
template <int SETS, bool PF = false>
__device__ __forceinline__ void lbfgs_tick_body(const DevModel& M, const BatchDev& D, const VarList* __restrict__ vls,
                                                int first_stage, int last_stage, int init, int step_mode,
                                                const int b, const int tid, float* s_al, OptScal& s_state, float* s_work,
                                                const float* f_src, const float* g_src, int* stage_out = nullptr) {
    if (tid >= 64) return;
    lbfgs_tick_wave0<SETS, PF>(M, D, vls, first_stage, last_stage, init, step_mode, b, tid, s_al, s_state, s_work, f_src, g_src, stage_out);
}


#pragma clang fp contract(fast)
