// Internal structures shared by the host layer (api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define SFX_J 55            // SMPL-X kinematic joints
#define SFX_POSE (3 * SFX_J)
#define SFX_NHAND 45
#define SFX_MAX_K 144       // mapped joints
#define SFX_MAX_ITEMS 240   // vertex items (21 + 68*3 = 225)
#define SFX_SMALL_ITEMS 32  // item capacity of the small closure variant (body-only: 11 items)
#define SFX_MAX_DYN 64      // dynamic-contour items (17 landmarks x 3 corners = 51)
#ifndef SFX_RIF_BIG
#define SFX_RIF_BIG 8        // 2-KiB blend-shape rows in flight per wavefront in the full-model closure (1 workgroup per CU:
#endif                       // the loads of 4 wavefronts are all the memory parallelism a frame gets)
#ifndef SFX_RIF_SMALL
#define SFX_RIF_SMALL 9      // ... in the body-only closure: 33 rows over 4 wavefronts in ONE pass (every pass is a ~1-us round trip to L2 / MALL: 4 -> 9 rows in flight made the needed-rows fit 14 % faster)
#endif
#ifndef SFX_SMALL_OCC
#define SFX_SMALL_OCC 1      // workgroups per CU the register budget of the persistent small kernel is sized for
#endif
#define SFX_KD_PAD 512      // padded blend-shape depth (20 + 486 = 506)
#define SFX_JPAD 56         // joints padded to an even MFMA depth
#define SFX_MAX_LEVELS 16
#define SFX_HIST 100        // L-BFGS history (optim_factory 'lbfgsls' default): slots of the ring unless history_size asks for more
#ifndef SFX_HIST_MAX
#define SFX_HIST_MAX 400    // largest history_size (the alphas of the two-loop recursion live in LDS: lbfgs_body.h s_al)
#endif
#define SFX_HROWS (SFX_HIST + 8)   // ring of R slots + rows R..R+7 mirroring slots 0..7: any 8 consecutive members of the
#define SFX_HROWS_MAX (SFX_HIST_MAX + 8)   // window are 8 consecutive rows (lbfgs_body.h lb_load); R = BatchCfgDev::hist_ring
#define SFX_NVAR_MAX 192    // optimiser vector length (182 / 88 / 6), padded to 3*64
#define SFX_FWD_N 6016      // floats per frame of saved forward state (FrameLDS prefix incl. the fp64 transforms + 96 + VPoser 1280)
#define SFX_NPAR_MAX 256    // canonical per-frame parameter block (182 with 12 hand components; 248 with all 45: forward only)
#define SFX_MAX_STAGES 8
#define SFX_MAX_GROUPS 12
#define SFX_NW 8             // nonzero skinning weights kept per vertex (needed-rows path)
// packed integer tables of the closure kernel (DevModel.meta)
#define MO_PAR 0
#define MO_LJ 56
#define MO_CS 112
#define MO_CL 168
#define MO_SK0 224
#define MO_SKL 280
#define MO_JT 424
#define MO_JS 568
#define MO_JI0 712
#define MO_JN 856
#define MO_IK 1000
#define MO_PRE 1240         // joint -> DFS pre-order position (subtree = contiguous pre-order range)
#define MO_SUB 1296         // joint -> subtree size
#define MO_ANC 1352         // [SFX_MAX_ROUNDS][56] 2^k-th ancestor of each joint (-1 past the root)
#define SFX_MAX_ROUNDS 5
#define SFX_META_N 1632

// Per-frame record (floats): keypoints, confidences, joint weights, camera-init mask, camera, camera rotation, regression
// pose -- packed by k_pack_fd (api.hip) from the arrays the host fills, read by the closure workgroup in ONE 16-byte copy
#define FD_GT 0
#define FD_CONF (2 * SFX_MAX_K)
#define FD_JW (3 * SFX_MAX_K)
#define FD_CMASK (4 * SFX_MAX_K)
#define FD_CAM (5 * SFX_MAX_K)
#define FD_CAMR (FD_CAM + 8)
#define FD_REG (FD_CAMR + 12)
#define FD_N (FD_REG + 64)

// Canonical per-frame parameter block (floats).  cam_t | global_orient | betas | lhand |
// rhand | expression | jaw | leye | reye | body_pose param (dead, iff !use_vposer) | embedding
struct ParLayout {
    int cam_t, go, betas, lh, rh, expr, jaw, leye, reye, bodyp, emb;
    int NB, NE, NPCA, NEMB, has_bodyp, npar;
};

// Variable list of one optimisation stage kind: flat optimiser index -> canonical index.
struct VarList {
    int n;
    int ngroups;
    short idx[SFX_NVAR_MAX];
    short g_off[SFX_MAX_GROUPS], g_len[SFX_MAX_GROUPS], g_has[SFX_MAX_GROUPS];
};

struct StageW {       // per body stage
    float bpw, sw, bend, hpw, epw, jaw[3], hand_jw, face_jw, coll;
};

struct DevModel {
    int V, F, S, P, KD, K;            // S = NB+NE, P = 486, KD = S+P
    int n_extra, n_lmk, n_dyn_rows, n_dyn;
    int n_levels, n_rounds;           // tree depth + 1; ceil(log2(n_levels)) pointer-jumping rounds
    int level_start[SFX_MAX_LEVELS + 1];
    // constants
    const float* v_template;   // [V][3]
    const float* dirs;         // [KD_PAD][3*Vpad]  k-major, coords interleaved (the adjoint GEMM's operand)
    const float* dirs_tiled;   // [Vpad/16][KD_PAD][48]  the same matrix, one contiguous block per 16-vertex tile (dense GEMM B operand)
    const float* dirsT;        // [V][3][KD_PAD] vertex-major (needed-rows path)
    const float* W;            // [V][J]
    const float* WT;           // [JPAD][Vpad]   (dense skinning GEMM B operand)
    // dense skinning GEMM, compressed per 16-vertex tile to the joints that carry weight there
    const int*   tj_n;         // [ntile16] joints used by the tile, padded to a multiple of 4 (<= JPAD)
    const int*   tj_list;      // [ntile16][JPAD] joint ids (padding: joint 0 with zero weights)
    const float* tj_w;         // [ntile16][JPAD][16] weights, row = list slot, col = vertex in tile
    const int*   jv_start;     // [J+1] transposed CSR of lbs_weights: the vertices (ascending) skinned by joint j ...
    const int*   jv_vid;       // [nnz]
    const float* jv_w;         // [nnz] ... and their weights (adjoint of a gradient on every vertex, lbs_adjoint.hip)
    const int*   Wsp_j;        // [V][SFX_NW] joints of the nonzero weights (ascending), pad: j=0,w=0
    const float* Wsp_w;        // [V][SFX_NW]
    const float* J_template;   // [J][3]
    const float* J_dirs;       // [J][3][S]
    const int*   parents;      // [J]
    const int*   level_joints; // [J] joints ordered by tree depth
    const int*   child_start;  // [J+1]
    const int*   child_list;   // [J-1]
    const float* comp_l;       // [NPCA][45]
    const float* comp_r;
    const float* pose_mean;    // [165]
    const int*   faces;        // [F][3]
    const int*   dyn_faces;    // [rows][n_dyn]
    const float* dyn_bary;     // [rows][n_dyn][3]
    // mapped-joint description
    const int*   jk_type;      // [K] 0 = kinematic joint, 1 = vertex items
    const int*   jk_src;       // [K] joint id (type 0)
    const int*   jk_item0;     // [K] first item (type 1)
    const int*   jk_nitem;     // [K]
    int n_items, n_static_items;
    int n_uniq;                // distinct vertices of the static items: the dense GEMM exports their v_posed and T
    const int* vslot;          // [Vpad] vertex -> index into the export arrays, or -1
    const int* item_uslot;     // [n_items] export index of a static item's vertex (-1: dynamic item, whose index comes with its LUT-row block: dynp_us)
    const int*   item_vid;     // [n_items] static vertex id (dynamic items: -1)
    const float* item_w;       // [n_items] static bary weight
    const float* item_vt;      // [n_items][3] v_template row of the item's vertex     } gathered copies for the static items:
    const int*   item_wj;      // [n_items][SFX_NW] = Wsp_j[item_vid]                  } one round trip at kernel entry
    const float* item_ww;      // [n_items][SFX_NW] = Wsp_w[item_vid]                  }
    const int*   item_dyn;     // [n_items] -1, or (landmark*3 + corner) of the dynamic LUT
    const int*   item_k;       // [n_items] owning mapped joint
    const int*   src_k0;       // [J+1] CSR: mapped joints that read kinematic joint s
    const int*   src_klist;    // [..]
    int Vpad;
    // per-joint lists of (item, skinning weight): static items, and dynamic items per LUT row
    const int *sj_start, *sj_item; const float* sj_w;      // [J+1], [..]
    const int *dj_start, *dj_item; const float* dj_w;      // [rows][J+1] (absolute offsets), [..]
    int n_dyn_items;
    // everything about the dynamic-contour items that depends on the LUT row, packed per row (one asynchronous copy into
    // the closure workgroup's LDS instead of four dependent global round trips): nd = n_dyn_items
    const int*   dynp_vid;     // [rows][nd]      vertex of item (n_static_items + q)
    const float* dynp_w;       // [rows][nd]      its barycentric weight
    const float* dynp_vt;      // [rows][nd][3]   its v_template row
    const int*   dynp_wj;      // [rows][nd][SFX_NW]  sparse skinning weights (Wsp_j / Wsp_w of the vertex)
    const float* dynp_ww;      // [rows][nd][SFX_NW]
    const int*   dynp_js;      // [rows][J+1]     per-joint adjoint lists of the row, offsets RELATIVE to the row's block ...
    const int*   dynp_ji;      // [rows][nd*SFX_NW]   ... items (absolute item ids), padded to nd * SFX_NW entries per row
    const float* dynp_jw;      // [rows][nd*SFX_NW]
    const int*   dynp_us;      // [rows][nd]      export index of the item's vertex in the dense GEMM's uvp (nullptr: no dynamic items)
    int n_sj;                  // entries of sj_item / sj_w
    const int*   meta;         // [SFX_META_N] packed copy of the tables above
    // VPoser decoder
    int vp_latent, vp_hidden;
    const float *vp_w1, *vp_b1, *vp_w2, *vp_b2, *vp_w3, *vp_b3;      // [512][L], [512][512], [126][512]
    const float *vp_w1T, *vp_w2T, *vp_w3T;                          // [L][512], [512][512], [512][128]
};

struct BatchCfgDev {
    int B, n_stages;
    int use_vposer, use_hands, use_face, use_conf, has_reg, use_conf_cam;
    int nbj;
    int maxiters, max_eval, lbfgs_max_iter;     // (run_fitting's step count | LBFGS max_eval | LBFGS max_iter)
    double ftol, gtol;
    float lr, rho, depth_w;
    int lbs_mode, reuse;
    float side_thsh; int lsh, rsh;     // side-view test: 2-D shoulder distance threshold, indices
    int pen;                           // interpenetration term enabled (dense mode)
    int kl[3], nil[3];                 // live keypoints / vertex items by stage class: body only, + hands, all (closure_body)
    double tol_grad, tol_change;       // LBFGS tolerance_grad / tolerance_change (lbfgs_ls.py defaults 1e-5 / 1e-9)
    int hist_cap;                      // LBFGS history_size (<= SFX_HIST_MAX)
    int hist_ring;                     // slots of the history ring: SFX_HIST, or history_size when that is larger (round 5)
    int proj64;                        // projection in fp64 in every stage (cfg float_dtype float64; the camera stage always is)
};

// Per-frame data pointers (all device).
struct BatchDev {
    BatchCfgDev cfg;
    ParLayout L;
    float* X;          // [B][NPAR_MAX] accepted parameters
    float* Xt;         // [B][NPAR_MAX] trial parameters (what the closure evaluates)
    float* gt;         // [B][K][2]
    float* conf;       // [B][K]
    float* jw;         // [B][K]
    float* cmask;      // [B][K]
    float* cam;        // [B][8] fx fy cx cy data_weight est_tz - -
    float* camR;       // [B][9]
    float* regpose;    // [B][63]  (or latent)
    float* fd;         // [B][FD_N] the seven arrays above packed per frame (k_pack_fd): what the closure workgroup loads
    float* f;          // [B] loss at Xt
    float* g;          // [B][NVAR_MAX] flat gradient at Xt
    float* bodypose;   // [B][63] decoded body pose (VPoser) scratch
    // dense path
    float* featR;      // [Bpad][KD_PAD]: one 2-KiB row of blend-shape coefficients per GEMM column (frame)
    float* AT;         // [12][JPAD][Bpad]
    float* verts;      // [B][V][3]
    float* joints;     // [B][K][3] (export)
    float* fullpose;   // [B][165]  (export)
    int Bpad;
    int*   slot;       // [B] frame -> operand column / vertex-buffer row (identity or compacted; -1: waiting in the queue)
    int nact;          // columns in use (frames the dense GEMM processes)
    const int* act;    // [nrun] frames the fused dense tick kernel works on (NULL: all B frames, block = frame)
    int nrun;          // length of act[]
    // optimiser state
    int*   stage;      // [B] current stage (-1 camera, 0.. body, n_stages = done)
    void*  opt;        // [B] OptState
    float* vec;        // [B][NVEC][NVAR_MAX] optimiser vectors
    float* hist;       // [B][2][hist_ring + 8][NVAR_MAX]
    int*   n_active;   // [1] frames not done
    float* stage_loss; // [B][1+MAX_STAGES]
    int*   stage_evals;     // [B][1+MAX_STAGES]
    int*   stage_ref_evals; // [B][1+MAX_STAGES]
    float* X0;         // [B][NPAR_MAX] first-orientation result (side-view second fit)
    float* gocam;      // [B][4] global_orient at the end of the camera stage
    float* stage_loss2;// [B][1+MAX_STAGES] second-orientation stage losses
    int*   try_both;   // [B]
    int*   orient_pass;// [B] 0 first fit, 1 second fit running, 2 done
    float* uvp;             // [B][n_uniq][3]  (slot-indexed) blend offsets (v_posed - v_template) of the item vertices, written by the dense GEMM
    float* pen_loss;        // [B] (slot-indexed) unweighted penetration loss of the pending evaluation
    float* pen_dverts;      // [B][V][3] (slot-indexed) its gradient with respect to the vertices
    int gmm_M;              // Gaussian-mixture body pose prior: components (0 = off) ...
    const float* gmm_mean;  // [M][64]
    const float* gmm_prec;  // [M][64][64] symmetrised precisions, rows padded
    const float* gmm_lognw; // [M] log nll_weights
    const float* gmm_csel;  // [M] constant of a component in the selection: -log nll_weights (merged form) | comp_const (per-component form)
    const float* gmm_cadd;  // [M] added to the selected component's value: 0 | -log nll_weights
    float gmm_scale;        // factor of the quadratic form: 0.5 | 1
    float* vposed;          // [B][V][3] (slot-indexed) v_posed of every vertex, written by the dense GEMM when the term is on
    float* adj_G;           // [Bpad][3*Vpad] (slot-indexed) d v_posed = T^T d verts: operand of the adjoint GEMM
    float* adj_part;        // [slices][KD_PAD][Bpad] its per-slice partial sums
    float* pen_dfeat;       // [B][KD_PAD] (slot-indexed) d pen_loss / d feat
    float* pen_dA;          // [B][J][12] (slot-indexed) d pen_loss / d A
    int*   pen_want;        // [B] (slot-indexed) 1 = the column's pending evaluation carries a collision weight
    int*   ext_n;           // [B] (slot-indexed) vertices with a nonzero penetration gradient (diagnostics)
    int*   pen_over;        // [B] (slot-indexed) 1 = the column's latest collision evaluation kept partners by arrival order somewhere
    int*   pen_flag;        // [B] (frame-indexed, sticky over a fit) the frame consumed such an evaluation: not reproducible run to run
    float* fwd;             // [B][SFX_FWD_N] forward state handed from the export pass to the adjoint pass
    long long* dbg;         // [64] phase timestamps of block 0 (NULL = off)
    float4* trace;          // [B][trace_cap] optimiser trace records (NULL = off), see sfx_batch_trace
    int*   trace_n;         // [B] records written
    int    trace_cap;
    int    trace_evals;     // 1: also one record (3, t, loss, |g|inf) per closure evaluation
};

enum { VEC_XINIT = 0, VEC_D, VEC_G, VEC_PREVG, VEC_GPREV, VEC_BG0, VEC_BG1, VEC_LSG0, NVEC };

#define SFX_CHECK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { \
    sfx_set_error("%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e_)); return -2; } } while (0)

void sfx_set_error(const char* fmt, ...);

// kernel launchers (defined in the .hip files)
struct ClosureArgs {
    int stage_override;     // -2: use per-frame stage[]; otherwise stage for all frames
    int forward_only;       // 1: export joints / full_pose / featR / AT only
    int use_dense_verts;    // 1: item vertices come from BatchDev.verts
    int export_dense;       // 1: write featR / AT for the dense kernel
    int from_X;             // 1: evaluate at X instead of Xt
    int from_scratch_items; // 1: dense mode, but evaluate the item rows here (the GEMM export belongs to another call)
    int keep_tables;        // 1: S.meta / S.fd are still valid from the previous evaluation of this workgroup
    int reuse_fwd;          // 1: forward state of this trial point was saved by the export pass
    const float* x_lds;     // the trial point in the workgroup's LDS (k_tick_dense: left there by the optimiser tick); NULL: read X / Xt
};
// the 47-KB closure variant serves models whose keypoints need at most SFX_SMALL_ITEMS vertex rows
// when the VPoser decoder is not in the loop
static inline bool sfx_small_closure(const DevModel& M, const BatchDev& D) {
    return M.n_items <= SFX_SMALL_ITEMS && !D.cfg.use_vposer;
}
void launch_closure(const DevModel& M, const BatchDev& D, const VarList* vl_dev, const StageW* sw_dev,
                    const ClosureArgs& a, hipStream_t s);
// d v_posed = T^T g of the interpenetration gradient, written by the kernel that forms g (k_pen_gather) when the caller is a
// fitting batch: operand of the adjoint GEMM (lbs_adjoint.hip).  adj_G == nullptr: stand-alone operator, nothing to do.
struct PenAdjPrep {
    const float* AT;        // [12][JPAD][Bpad] skinning transforms of the columns (BatchDev.AT)
    const int*   Wsp_j;     // [V][SFX_NW]
    const float* Wsp_w;     // [V][SFX_NW]
    const float* W;         // [V][J] (vertices with more than SFX_NW weights)
    float*       adj_G;     // [Bpad][3 * Vpad]
    int Bpad, Vpad;
};
void launch_pen_adjoint(const DevModel& M, const BatchDev& D, hipStream_t s);
int sfx_adj_slices(const DevModel& M);
void launch_lbs_dense(const DevModel& M, const BatchDev& D, hipStream_t s);
#ifdef SFX_LAB
extern int g_lbs_dense_form;          // 16 (k_lbs_dense16, default) | 17 (k_lbs_dense16 at every size) | 32 (k_lbs_dense)
#endif
void launch_lbfgs_tick(const DevModel& M, const BatchDev& D, const VarList* vl_dev, int first_stage,
                       int last_stage, int init, int step_mode, hipStream_t s);
size_t sfx_optstate_size();
int debug_two_loop(const float* S, const float* Y, int cnt, int hist_cap, const float* g, float* d_out);   // lbfgs.hip (host pointers)
void launch_fit_rows(const DevModel& M, const BatchDev& D, const VarList* vl_dev, const StageW* sw_dev,
                     int first_stage, int last_stage, int max_ticks, hipStream_t s);
void launch_tick_dense(const DevModel& M, const BatchDev& D, const VarList* vl_dev, const StageW* sw_dev,
                       int first_stage, int last_stage, int has_eval, hipStream_t s);
