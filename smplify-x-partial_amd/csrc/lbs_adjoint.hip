// lbs_adjoint.hip -- adjoint of the dense SMPL-X skinning for a gradient that lives on EVERY vertex
// (the interpenetration term, smplifyx/fitting.py:437-455: autograd walks back through the whole of
// smplx.lbs.lbs; here the vertex gradient g = d pen_loss / d verts comes from csrc/collide.hip).
//
//     verts = T(v) [v_posed; 1],  T(v) = sum_j W[v][j] A_j,  v_posed = v_template + dirs^T feat
//  =>  d v_posed(v) = T(v)[:3,:3]^T g(v)                                   (k_pen_gather, collide.hip: the lane that forms g(v))
//      d feat[k]    = sum_{v,c} dirs[k][3v+c] d v_posed(v)[c]               k_lbs_dense_adj (fp32 MFMA) + k_adj_finish
//      d A_j        = sum_v W[v][j] g(v) (x) [v_posed(v); 1]                k_adj_finish
// The tick kernel's adjoint pass adds d feat and d A (times coll_loss_weight) to the keypoint term's
// before it walks the kinematic chain back (closure_body.h).
//
// k_lbs_dense_adj is the transpose of k_lbs_dense: C[k][b] = sum_r dirs[k][r] G[b][r] with the
// reduction r = 3v + c over 3 * Vpad = 31 440 and a small output (512 x frames), so the work is
// split along r.  Both operands are contiguous along r: lane (m, q) of a wavefront loads the float4
// dirs[k0 + m][r + 4q .. 4q + 3] and G[b0 + m][r + 4q .. 4q + 3]; element s of those float4 is the
// operand of MFMA step s -- any assignment of reduction indices to MFMA k-slots is valid as long as
// A and B agree, so no LDS transpose is needed and every global load is a 16-byte vector load.
// Wavefront tile: 64 k x 64 frames (4 x 4 MFMA tiles, 64 accumulators); a workgroup's 4 wavefronts
// take 4 consecutive r ranges of the same tile and add their accumulators through LDS in wave
// order; k_adj_finish adds the workgroups' partials in group order: no atomics, fixed association.
// Algorithmic traffic per launch: dirs 64.4 MB + G (frames x 125.8 KB); flops 2 x 512 x 31 440 per frame.
#include "sfx_internal.h"
#include "wave_ops.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

#define ADJ_T 256
#define ADJ_RW_MIN 256           // r range of one wavefront: 256 (few frame tiles: more workgroups) or 512

// The columns of a launch that carry a collision weight (pen_want), in ascending order: frame tile `tile` of the adjoint GEMM
// works on the 64 of them with ranks 64 tile .. 64 tile + 63 (round 4: it used to work on ALL active columns -- 40 of 127 want
// the term in an average round of the benchmark, so half of its frame tiles multiplied stale operands for nobody).  Formed by
// wavefront 0 of every workgroup from the flags (a few ballots); a column's arithmetic does not depend on where it sits in a
// tile, so the numbers are what they were.  Returns the number of wanted columns.
__device__ __forceinline__ int adj_tile_columns(const int* __restrict__ want, const int nact, const int tile, int* s_cols /* [64] */, int* s_nw) {
    const int t = threadIdx.x, lane = t & 63;
    if (t < 64) {
        s_cols[lane] = 0;
        int cnt = 0;
        for (int base = 0; base < nact; base += 64) {
            const int b = base + lane;
            const bool w = b < nact && want[b] != 0;
            const unsigned long long m = __ballot(w);
            const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull)) - tile * 64;
            if (w && pos >= 0 && pos < 64) s_cols[pos] = b;
            cnt += __popcll(m);
        }
        if (lane == 0) *s_nw = cnt;
    }
    __syncthreads();
    return *s_nw;
}

template <int KI> struct __align__(16) AdjLDS { float acc[4][16 * KI][64]; };      // one 64 x 64 tile per wavefront: first its own first-chunk sums (rw = 512), then the tile it hands on

// ONE association of the sum over r for every launch shape (a column's result must not depend on how many other columns
// are active -- the rule every kernel of the fit obeys; until round 4 the r range per wavefront, 512 or 256 by the number of
// frame tiles, changed where the partial sums met).  Canonical form: chunks c_i of 256 consecutive r, each accumulated on
// its own from zero; pairs p_j = c_2j + c_2j+1; groups t_g = ((p_4g + p_4g+1) + p_4g+2) + p_4g+3; total = t_0 + t_1 + ...
//   rw = 512: wavefront w of workgroup g forms p_(4g+w) (two accumulations of 256, then one add), the workgroup writes t_g;
//   rw = 256: wavefront w of workgroup s forms c_(4s+w), the workgroup writes p_2s and p_2s+1, k_adj_finish forms the t_g.
// KI: 16-row MFMA tiles of the matrix per wavefront (4: a 64 k x 64 column tile; 2, round 5: 32 k x 64 -- twice the workgroups on the
// same r ranges, i.e. the same association of every sum, for a launch that is one workgroup per CU and waits for its own loads)
template <int KI>
__global__ __launch_bounds__(ADJ_T, 2)
void k_lbs_dense_adj(DevModel M, BatchDev D, int rw) {
    __shared__ AdjLDS<KI> S;
    __shared__ int s_cols[64], s_nw;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = lane & 15, q = lane >> 4;
    // grid: x = r slice (fastest: the slices of one (k tile, frame tile) stream disjoint parts of dirs),
    // y = k tile (8), z = frame tile: the 64 WANTED columns of ranks 64 z .. (results land at those ranks in adj_part)
    const int slice = blockIdx.x, k0 = blockIdx.y * 16 * KI, b0 = blockIdx.z * 64;
    const int n_wanted = adj_tile_columns(D.pen_want, D.nact, blockIdx.z, s_cols, &s_nw);
    if (n_wanted <= b0) return;      // no wanted column in this tile
    // (round 5) 16-column MFMA tiles of this frame tile that hold a wanted column at all: the others are neither loaded nor
    // multiplied -- in the rounds where a handful of columns carry a collision weight (the start of the stages with the term, the
    // tail of a fit) three quarters of the launch's MFMA work was for columns nobody reads (27.5 us floor).  A column's
    // arithmetic does not depend on its neighbours: same bits.
    const int nj = min(4, (n_wanted - b0 + 15) >> 4);
    const int LD = 3 * M.Vpad;
    const int r_lo = min(LD, (slice * 4 + wv) * rw);
    const int r_hi = min(LD, r_lo + rw);
    const int r_mid = min(r_hi, r_lo + ADJ_RW_MIN);      // end of the first 256-chunk of this wavefront's range
    // (round 5) the matrix is read from the TILE-major copy the forward GEMM streams ([tile of 16 vertices][k][48], lbs_dense.hip):
    // element [k][r] sits at (r / 48) * KD_PAD * 48 + k * 48 + r % 48, a 16-r step never leaves a tile (48 = 3 x 16), and the three
    // steps of a tile use every byte of its [16 k][48] block -- out of the k-major matrix a load instruction touched 16 half-used
    // lines 126 KB apart.  One 64-MB matrix per round instead of two: the forward's and the adjoint's.  Same values, same order.
    const float* pa = M.dirs_tiled + (size_t)(k0 + m) * 48 + 4 * q;
    const float* pbc[4];                        // the lane's column of each of the four 16-column MFMA tiles
#pragma unroll
    for (int i = 0; i < 4; ++i) pbc[i] = D.adj_G + (size_t)s_cols[16 * i + m] * LD + 4 * q;
    const size_t sa = (size_t)16 * 48;          // next MFMA tile of the matrix: 16 rows further
    f32x4 acc[KI][4];
#pragma unroll
    for (int i = 0; i < KI; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    bool split = false;
    float4 a_c[KI], b_c[4], a_n[KI], b_n[4];
    auto load = [&](float4 (&a)[KI], float4 (&b)[4], const int r) {
        const int tl = r / 48;
        const float* pt = pa + (size_t)tl * (SFX_KD_PAD * 48) + (r - tl * 48);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < KI) a[i] = *reinterpret_cast<const float4*>(pt + i * sa);
            if (i < nj) b[i] = *reinterpret_cast<const float4*>(pbc[i] + r);
        }
    };
    if (r_lo < r_hi) load(a_c, b_c, r_lo);
    for (int r = r_lo; r < r_hi; r += 16) {
        const bool more = r + 16 < r_hi;
        if (more) load(a_n, b_n, r + 16);
        if (r == r_mid) {        // (rw = 512) the second chunk starts from zero; the first waits in this wavefront's LDS tile (same lane writes and reads it)
            split = true;
#pragma unroll
            for (int i = 0; i < KI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) S.acc[wv][16 * i + 4 * q + e][16 * j + m] = acc[i][j][e];
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (j < nj) {
#pragma unroll
                for (int i = 0; i < KI; ++i) {
                    acc[i][j] = MFMA(a_c[i].x, b_c[j].x, acc[i][j]);
                    acc[i][j] = MFMA(a_c[i].y, b_c[j].y, acc[i][j]);
                    acc[i][j] = MFMA(a_c[i].z, b_c[j].z, acc[i][j]);
                    acc[i][j] = MFMA(a_c[i].w, b_c[j].w, acc[i][j]);
                }
            }
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { if (i < KI) a_c[i] = a_n[i]; b_c[i] = b_n[i]; }
        }
    }
    const bool wide = rw > ADJ_RW_MIN;
    if (split) {                 // p = c_lo + c_hi  (a range that ends inside its first chunk has no second one: p = that chunk, as c + 0 = c)
#pragma unroll
        for (int i = 0; i < KI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[i][j][e] = S.acc[wv][16 * i + 4 * q + e][16 * j + m] + acc[i][j][e];
    }
    // accumulator (i, j), register e of lane (m, q) = C[k0 + 16 i + 4 q + e][b0 + 16 j + m]
    if (wide ? wv > 0 : (wv & 1)) {
#pragma unroll
        for (int i = 0; i < KI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) S.acc[wv][16 * i + 4 * q + e][16 * j + m] = acc[i][j][e];
    }
    __syncthreads();
    if (wide) {
        if (wv == 0) {           // t_g = ((p_0 + p_1) + p_2) + p_3
            float* out = D.adj_part + ((size_t)slice * SFX_KD_PAD + k0) * D.Bpad + b0;
#pragma unroll
            for (int i = 0; i < KI; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int kr = 16 * i + 4 * q + e, bc = 16 * j + m;
                        const float s = ((acc[i][j][e] + S.acc[1][kr][bc]) + S.acc[2][kr][bc]) + S.acc[3][kr][bc];
                        out[(size_t)kr * D.Bpad + bc] = s;
                    }
        }
    } else if (!(wv & 1)) {      // wavefronts 0 and 2: p_2s = c_0 + c_1, p_2s+1 = c_2 + c_3
        float* out = D.adj_part + ((size_t)(slice * 2 + (wv >> 1)) * SFX_KD_PAD + k0) * D.Bpad + b0;
#pragma unroll
        for (int i = 0; i < KI; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int kr = 16 * i + 4 * q + e, bc = 16 * j + m;
                    out[(size_t)kr * D.Bpad + bc] = acc[i][j][e] + S.acc[wv + 1][kr][bc];
                }
    }
}

// The two small passes around the adjoint GEMM in ONE launch (two until round 4): blocks [0, n_red) add the GEMM's partials,
// blocks [n_red, ..) form d A.
//   d feat[b][k] = t_0 + t_1 + ... in group order; n_part partials of kind `pairs` (1: the p_j of the rw = 256 launch, summed
//                  four at a time into the t_g first; 0: the t_g themselves); 64 columns x 4 k per block
//   d A_j[b]     = sum over the vertices skinned by joint j (ascending) of W[v][j] g(v) (x) [v_posed(v); 1]: one wavefront per
//                  (joint, column), four joints per block; lanes stride the joint's vertex list, DPP reduction (fixed order)
__global__ __launch_bounds__(256)
void k_adj_finish(DevModel M, BatchDev D, int n_part, int pairs, int n_red) {
    __shared__ int s_cols[64], s_nw;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if ((int)blockIdx.x < n_red) {
        const int ftile = blockIdx.x / (SFX_KD_PAD / 4), kq = blockIdx.x % (SFX_KD_PAD / 4);
        const int nw = adj_tile_columns(D.pen_want, D.nact, ftile, s_cols, &s_nw);
        const int c = ftile * 64 + lane;            // rank among the wanted columns = position in adj_part
        const int k = kq * 4 + wv;
        if (c >= nw) return;
        const int b = s_cols[lane];
        const size_t st = (size_t)SFX_KD_PAD * D.Bpad;
        const float* p = D.adj_part + (size_t)k * D.Bpad + c;
        float s = 0.f;
        // (round 5: sixteen partials in flight per trip, from clamped indices with a select behind the loads -- the loop used to wait
        //  for four loads, add, and ask for the next four: 16 dependent round trips for the 62 partials of the rw = 256 launch.  The
        //  sums keep their association.)
        for (int i0 = 0; i0 < n_part; i0 += 16) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) { const float x = p[(size_t)min(i0 + j, n_part - 1) * st]; v[j] = i0 + j < n_part ? x : 0.f; }
            if (pairs) {
#pragma unroll
                for (int j = 0; j < 16; j += 4) if (i0 + j < n_part) s += ((v[j] + v[j + 1]) + v[j + 2]) + v[j + 3];
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) if (i0 + j < n_part) s += v[j];
            }
        }
        D.pen_dfeat[(size_t)b * SFX_KD_PAD + k] = s;
        return;
    }
    constexpr int JB = (SFX_J + 3) / 4;
    const int q = blockIdx.x - n_red;
    const int b = q / JB, j = (q % JB) * 4 + wv;
    if (j >= SFX_J || !D.pen_want[b]) return;
    const float* g = D.pen_dverts + (size_t)b * M.V * 3;
    const float* vp = D.vposed + (size_t)b * M.V * 3;
    float acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // four list entries per lane and trip, every load unconditional from a clamped index (the lane's chain was vertex id ->
    // gradient -> weight / v_posed, one round trip each, once per entry: 31 us for ~12 entries per lane); the entries are still
    // added in ascending order, and an entry without gradient is still skipped (0 x inf must not reach the sum)
    const int i1 = M.jv_start[j + 1];
    // (round 5: the ids and weights of the NEXT trip are requested before this trip's gathers are consumed -- a trip was two
    //  dependent round trips, list entry -> gradient / position, and a torso joint's list is a dozen trips.  Same order of the sums.)
    int vn[4]; float wn[4];
    auto heads = [&](const int i0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int i = max(min(i0 + u * 64, i1 - 1), 0); vn[u] = M.jv_vid[i]; wn[u] = M.jv_w[i]; }
    };
    int i0 = M.jv_start[j] + lane;
    if (i0 < i1) heads(i0);
    for (; i0 < i1; i0 += 4 * 64) {
        int vv[4]; float ww[4], gg[4][3], pp[4][3];
#pragma unroll
        for (int u = 0; u < 4; ++u) { vv[u] = vn[u]; ww[u] = wn[u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 3; ++e) { gg[u][e] = g[vv[u] * 3 + e]; pp[u][e] = vp[vv[u] * 3 + e]; }
        heads(i0 + 4 * 64);                    // (unconditional, from clamped indices: a branch here would put a full wait in front of it)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (i0 + u * 64 >= i1 || (gg[u][0] == 0.f && gg[u][1] == 0.f && gg[u][2] == 0.f)) continue;
            const float p4[4] = {pp[u][0], pp[u][1], pp[u][2], 1.f};
            const float wg[3] = {ww[u] * gg[u][0], ww[u] * gg[u][1], ww[u] * gg[u][2]};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int c = 0; c < 4; ++c) acc[r * 4 + c] += wg[r] * p4[c];
        }
    }
#pragma unroll
    for (int e = 0; e < 12; ++e) acc[e] = wave_sum_dpp(acc[e]);
    if (lane == 0) {
        float* o = D.pen_dA + ((size_t)b * SFX_J + j) * 12;
#pragma unroll
        for (int e = 0; e < 12; ++e) o[e] = acc[e];
    }
}

int sfx_adj_slices(const DevModel& M) { return 2 * ((3 * M.Vpad + 4 * ADJ_RW_MIN - 1) / (4 * ADJ_RW_MIN)); }      // capacity of adj_part: two pair sums per workgroup of the rw = 256 launch

// gradient of the penetration term with respect to feat (betas / expression / pose feature) and the
// skinning transforms, for every active column whose pen_want flag is set
void launch_pen_adjoint(const DevModel& M, const BatchDev& D, hipStream_t s) {
    if (D.nact <= 0) return;
    const int ftiles = (D.nact + 63) / 64;
    const int rw = ftiles >= 3 ? 512 : ADJ_RW_MIN;       // 8 k tiles x ftiles x slices workgroups: keep >= 256 of them
    const int ns = (3 * M.Vpad + 4 * rw - 1) / (4 * rw);
    // (d v_posed = T^T g, the GEMM's operand, was written by k_pen_gather: PenAdjPrep)
#ifdef SFX_LAB       // SFX_ADJ_KI=4: the 64-k tile of rounds 3-5 (A/B switch of the lab build; same bits)
    static const int ki = [] { const char* e = getenv("SFX_ADJ_KI"); return e && atoi(e) == 4 ? 4 : 2; }();
    if (ki == 4) { hipLaunchKernelGGL(k_lbs_dense_adj<4>, dim3(ns, SFX_KD_PAD / 64, ftiles), dim3(ADJ_T), 0, s, M, D, rw); } else
#endif
    hipLaunchKernelGGL(k_lbs_dense_adj<2>, dim3(ns, SFX_KD_PAD / 32, ftiles), dim3(ADJ_T), 0, s, M, D, rw);
    const int n_red = ftiles * (SFX_KD_PAD / 4);
    hipLaunchKernelGGL(k_adj_finish, dim3(n_red + D.nact * ((SFX_J + 3) / 4)), dim3(256), 0, s, M, D,
                       rw > ADJ_RW_MIN ? ns : 2 * ns, rw > ADJ_RW_MIN ? 0 : 1, n_red);
}
