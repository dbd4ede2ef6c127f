/* sfx.h -- C ABI of libsfx.so, the MI355X (gfx950) SMPL-X fitting engine.
 *
 * The reference (xiyichen/smplify-x-partial) has no FFI seam: its hot path is a chain of
 * Python objects.  This header is the boundary the replacement defines
 * UNDER that Python surface; each entry point names the reference interface it stands
 * in for.  Plain pointers and sizes only -- no torch types.  Unless stated otherwise a
 * `const float*` / `const int32_t*` argument is a HOST pointer read during the call; a
 * `*_dev` argument is a DEVICE pointer (e.g. torch.Tensor.data_ptr()); `stream` is a
 * hipStream_t passed as void* (NULL = default stream).  No ownership is transferred.
 * One model/batch handle per GPU; handles are thread-compatible, not thread-safe.
 * Every function returns 0 on success or a negative code; sfx_last_error() describes it.
 */
#ifndef SFX_H_
#define SFX_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sfx_model sfx_model;   /* model constants resident in HBM            */
typedef struct sfx_batch sfx_batch;   /* B frames: parameters, data, optimiser state */

/* ---- model -------------------------------------------------------------------------
 * Replaces smplx.create(...) / smplx.SMPLX.__init__ (reference call site
 * smplifyx/main.py:109-127; file keys SURVEY.md appendix A.1).  Arrays use the .npz
 * layouts; the library re-lays them out for the GPU (k-major blend-shape matrix for the
 * dense MFMA GEMM, vertex-major copy for the needed-rows path, folded joint regressor). */
typedef struct sfx_model_desc {
    int32_t V, F, J;              /* 10475, 20908, 55                                  */
    int32_t num_betas, num_expr;  /* 10, 10                                            */
    int32_t num_pca;              /* hand PCA comps used (12); 0 = hands given as 45-D */
    const float*   v_template;    /* [V][3]                                            */
    const float*   shapedirs;     /* [V][3][num_betas+num_expr] (betas then expression)*/
    const float*   posedirs;      /* [V][3][9*(J-1)]                                   */
    const float*   J_regressor;   /* [J][V]                                            */
    const float*   lbs_weights;   /* [V][J]                                            */
    const int32_t* parents;       /* [J], parents[0] = -1                              */
    const float*   hands_comp_l;  /* [num_pca][45]                                     */
    const float*   hands_comp_r;  /* [num_pca][45]                                     */
    const float*   pose_mean;     /* [3*J]                                             */
    const int32_t* faces;         /* [F][3]                                            */
    int32_t        n_extra;       /* 21 vertex joints (VertexJointSelector)            */
    const int32_t* extra_vertex_ids;
    int32_t        n_lmk;         /* 51 static landmarks                               */
    const int32_t* lmk_faces_idx; /* [n_lmk]                                           */
    const float*   lmk_bary;      /* [n_lmk][3]                                        */
    int32_t        n_dyn_rows, n_dyn;  /* 79, 17 (0,0 = no face contour)               */
    const int32_t* dyn_lmk_faces_idx;  /* [n_dyn_rows][n_dyn]                          */
    const float*   dyn_lmk_bary;       /* [n_dyn_rows][n_dyn][3]                       */
    int32_t        K;             /* joints after joint_mapper                         */
    const int32_t* joint_map;     /* [K] indices into the 55+n_extra+n_lmk+n_dyn list
                                     (utils.JointMapper, smplifyx/utils.py:68-81)      */
} sfx_model_desc;

int  sfx_model_create(const sfx_model_desc* desc, sfx_model** out);
void sfx_model_destroy(sfx_model* m);

/* VPoser-v1 decoder weights (human_body_prior cvpr19; reference call sites
 * fit_single_frame.py:241-245, fitting.py:236-238): fc1 [H][L], fc2 [H][H], out [126][H]. */
int  sfx_model_set_vposer(sfx_model* m, int32_t latent, int32_t hidden,
                          const float* fc1_w, const float* fc1_b,
                          const float* fc2_w, const float* fc2_b,
                          const float* out_w, const float* out_b);

/* ---- stand-alone LBS forward ----------------------------------------------------------
 * Replaces body_model(return_verts=True, body_pose=..., return_full_pose=True)
 * (fitting.py:82,248; fit_single_frame.py:611).  All pointers are DEVICE pointers to
 * contiguous fp32; any output may be NULL.  `hands` are PCA coefficients [B][num_pca]. */
/* Per-face part labels of smplx_parts_segm.pkl and the part pairs that never collide
 * (fit_single_frame.py:316-328): used by batches created with interpenetration = 1.  segm /
 * parents [F]; ign_pairs [n_ign][2].  Without this call -- or after one with segm = parents = NULL,
 * which clears the labels -- no pair is filtered by part.                                      */
int  sfx_model_set_parts(sfx_model* m, const int32_t* segm, const int32_t* parents, const int32_t* ign_pairs,
                         int32_t n_ign);

int  sfx_lbs_forward(sfx_model* m, int32_t B,
                     const float* global_orient_dev,   /* [B][3]  */
                     const float* body_pose_dev,       /* [B][63] */
                     const float* betas_dev,           /* [B][num_betas] */
                     const float* expression_dev,      /* [B][num_expr]  */
                     const float* jaw_dev, const float* leye_dev, const float* reye_dev, /* [B][3] */
                     const float* lhand_dev, const float* rhand_dev,   /* [B][num_pca] */
                     float* vertices_out_dev,          /* [B][V][3]  or NULL */
                     float* joints_out_dev,            /* [B][K][3]  or NULL */
                     float* full_pose_out_dev,         /* [B][3J]    or NULL */
                     void* stream);

/* ---- a batch of independent frames -------------------------------------------------------
 * Replaces, for B frames at once, the objects fit_single_frame() wires together
 * (fit_single_frame.py:413-445: create_loss('camera_init'), create_loss('smplify'),
 * FittingMonitor, optim_factory.create_optimizer('lbfgsls')).                            */
typedef struct sfx_stage_weights {      /* one entry of opt_weights (fit_single_frame.py:330-353) */
    float body_pose_weight, shape_weight;
    float hand_prior_weight, expr_prior_weight;
    float jaw_prior_weight[3];
    float hand_joint_weight, face_joint_weight;   /* joint_weights slices (:569-572)    */
    float coll_loss_weight;                       /* coll_loss_weights[stage]; used when interpenetration = 1 */
    float bending_prior_weight;                   /* < 0: derive 3.17 * body_pose_weight (:567-568) */
} sfx_stage_weights;

typedef struct sfx_batch_cfg {
    int32_t B;
    int32_t n_stages;               /* body stages (3 or 5)                              */
    int32_t use_vposer;             /* pose_embedding = VPoser latent [32]                */
    int32_t use_hands, use_face;    /* SMPLifyLoss flags (fitting.py:331-339)             */
    int32_t use_joints_conf;        /* weights = joint_weights * conf (fitting.py:380)    */
    int32_t has_regression_pose;    /* pprior = |emb - regression_pose|^2 (:391-397)      */
    int32_t use_conf_cam_init;      /* use_conf quirk of camera-init loss (:509-511)      */
    int32_t num_body_joints;        /* 25 / 26 / 23: start of the hand keypoints          */
    int32_t maxiters;               /* run_fitting steps, and LBFGS max_iter unless lbfgs_max_iter says otherwise (cfg: 30) */
    double  ftol, gtol;             /* run_fitting tolerances (cfg: 1e-9, 1e-9)           */
    float   lr;                     /* cfg: 1.0                                           */
    float   rho;                    /* GMoF rho (cfg: 100)                                */
    float   depth_loss_weight;      /* camera-init depth term (default 1e2)               */
    int32_t lbs_mode;               /* 0: needed-rows LBS in the loop; 1: dense V-vertex
                                       LBS every closure (what the reference evaluates)   */
    int32_t reuse_entry_eval;       /* 1: serve LBFGS.step's entry evaluation from the
                                       value already computed at the same point           */
    float   side_view_thsh;         /* > 0: frames whose 2-D shoulder distance is below it are
                                       fitted twice (orientation flipped by pi about y) and the
                                       lower final loss kept (fit_single_frame.py:461-463,
                                       527-551,662-667); needs a fit over stages -1..n-1     */
    int32_t left_shoulder_idx, right_shoulder_idx;
    int32_t interpenetration;       /* 1: penetration term in the stages with coll_loss_weight > 0
                                       (fitting.py:437-455); needs lbs_mode 1 (all vertices)      */
    int32_t max_collisions;         /* partners kept per triangle (cfg: 8 / 128)                   */
    float   df_cone_height;         /* sigma of the cone distance field (cfg: 0.5 / 1e-4)          */
    int32_t penalize_outside;
    int32_t slots;                  /* dense mode: GEMM columns (frames resident in the fit loop at a time);
                                       0 or >= B: every frame has a column.  With slots < B the frames beyond
                                       the first `slots` wait in a queue and take over the columns of frames
                                       that finish (continuous batching): the reference's loop over frames
                                       (main.py:207) for jobs larger than one GEMM batch.  A frame's result does
                                       not depend on when it is admitted or which column it gets              */
    /* hyper-parameters of optimizers/lbfgs_ls.py's LBFGS that optim_factory.py:27-65 leaves at their defaults.  The two
       tolerances: NEGATIVE = default, 0 is passed through (legal in the reference: the test is disabled); the counts: 0 = default */
    double  lbfgs_tolerance_grad;   /* 1e-5                                                          */
    double  lbfgs_tolerance_change; /* 1e-9                                                          */
    int32_t lbfgs_max_eval;         /* maxiters * 5 / 4                                              */
    int32_t lbfgs_history_size;     /* 100; up to 400 (a larger history gets a ring of that many slots) */
    int32_t lbfgs_max_iter;         /* LBFGS(max_iter): iterations per LBFGS.step and bound of the zoom phase (lbfgs_ls.py:304,397);
                                       0 = maxiters, the one value optim_factory.py:15 hands to both; lbfgs_max_eval's default
                                       follows it (max_iter * 5 / 4, lbfgs_ls.py:203)                              */
    int32_t high_precision;         /* cfg float_dtype: float64 (main.py:99-105): besides the keypoint forward (always fp64) the
                                       projection up to the pixel residual is carried in fp64 in every stage -- gradient noise
                                       0.13 x torch fp32's, the fits behave like the reference's float64 run (LAB_NOTES.md §3.1);
                                       parameters, reverse sweep and optimiser stay fp32                          */
    int32_t point2plane;            /* DistanceFieldPenetrationLoss(point2plane=True) (cmd_parser.py:239): see
                                       sfx_pen_set_point2plane                                                     */
} sfx_batch_cfg;

int  sfx_batch_create(sfx_model* m, const sfx_batch_cfg* cfg,
                      const sfx_stage_weights* stages /* [n_stages] */, sfx_batch** out);
void sfx_batch_destroy(sfx_batch* b);

/* Per-frame inputs (HOST pointers, copied):
 *  keypoints   [B][K][3]  x, y, confidence                (fit_single_frame.py:276-284)
 *  joint_weights [B][K]   after joints_to_ign and low-confidence zeroing (:285-287,574)
 *  cam_init_mask [B][K]   1 for the trimmed init_joints_idxs (:289-294)
 *  camera      [B][6]     focal_x, focal_y, center_x, center_y, data_weight(=1000/H), est_tz
 *  cam_rot     [B][9]     camera rotation (row-major; identity in the reference)
 */
int  sfx_batch_set_frames(sfx_batch* b, const float* keypoints, const float* joint_weights,
                          const float* cam_init_mask, const float* camera, const float* cam_rot);

/* Parameters (HOST pointers).  Names = the reference's result-pkl keys
 * (fit_single_frame.py:644-657).  `pose_embedding` is [B][63] or [B][latent];
 * `regression_pose` (may be NULL) is the clone taken at fit_single_frame.py:442.          */
int  sfx_batch_set_params(sfx_batch* b, const float* cam_translation, const float* global_orient,
                          const float* betas, const float* lhand, const float* rhand,
                          const float* expression, const float* jaw, const float* leye,
                          const float* reye, const float* pose_embedding,
                          const float* regression_pose);
int  sfx_batch_get_params(sfx_batch* b, float* cam_translation, float* global_orient,
                          float* betas, float* lhand, float* rhand, float* expression,
                          float* jaw, float* leye, float* reye, float* pose_embedding,
                          float* body_pose /* [B][63] decoded body pose */);

/* One closure evaluation for every frame at the CURRENT parameters (fitting.py:232-273:
 * forward + loss + adjoint).  stage = -1: camera-init loss over [cam_t, global_orient]
 * (N=6); stage >= 0: SMPLifyLoss with that stage's weights over the body variable vector
 * (order of body_model.parameters() + pose_embedding, N = sfx_batch_num_vars).
 * loss_out [B], grad_out [B][N]: HOST pointers (either may be NULL).                      */
int  sfx_batch_num_vars(sfx_batch* b, int32_t stage);
int  sfx_batch_closure(sfx_batch* b, int32_t stage, float* loss_out, float* grad_out, void* stream);

/* Camera translation guess of fitting.guess_init (fitting.py:36-110) for every frame:
 * one LBS forward, t_z = f * mean|d3D| / mean|d2D| over `n_pairs` keypoint pairs.
 * Writes cam_translation = (0,0,t_z) and est_tz.                                           */
int  sfx_batch_guess_init(sfx_batch* b, const int32_t* pairs /* [n_pairs][2] */, int32_t n_pairs,
                          void* stream);

/* The whole per-frame schedule of fit_single_frame.py:447-612 for all frames, on device:
 * camera stage, then n_stages body stages, each = FittingMonitor.run_fitting
 * (fitting.py:147-217) driving LBFGS.step with strong-Wolfe line search
 * (optimizers/lbfgs_ls.py:39-167,256-445).  first_stage/last_stage select a sub-range
 * (-1 = camera stage).  Blocks until done.                                                */
int  sfx_batch_fit(sfx_batch* b, int32_t first_stage, int32_t last_stage, void* stream);

/* ONE optimizer.step(closure) of LBFGS('lbfgsls') for every frame (lbfgs_ls.py:256-445), for
 * callers that drive run_fitting's outer loop themselves.  resume = 0 starts a fresh optimiser
 * (new stage), 1 continues the previous one (history kept).  loss_out [B] (HOST) receives what
 * step() returns: the loss at entry.                                                        */
int  sfx_batch_step(sfx_batch* b, int32_t stage, int32_t resume, float* loss_out, void* stream);

/* Gradient of the most recent closure evaluation of every frame, grad_out [B][num_vars(stage)]
 * (HOST): what `var.grad` holds after optimizer.step() in the reference, which run_fitting's
 * gtol test reads (fitting.py:191-193).                                                       */
int  sfx_batch_get_grad(sfx_batch* b, int32_t stage, float* grad_out);

/* Interpenetration diagnostics of the most recent evaluation (batches with interpenetration = 1),
 * per active GEMM column: stats as sfx_pen_stats; ext_n = vertices that carried a gradient.    */
int  sfx_batch_pen_stats(sfx_batch* b, int32_t* stats_host /* [B][4] */, int32_t* ext_n_host /* [B] or NULL */);
/* Per FRAME, sticky since the start of the last fit / step (HOST [B]): 1 = the frame consumed a collision evaluation in which
 * a bucket walk was cut short (a mesh folded into a few grid cells) -- pairs beyond the cut are missing and which ones depends
 * on arrival order (the package's BVH is traversal-order dependent in the same situation, fitting.py:445-447), so this frame's
 * result is not reproducible run to run.  0 on any sane mesh.  (A triangle with more than 2 x max_collisions partners is no
 * such case since round 4: its kept partners -- the lowest ids -- are derived from the grid again; with max_collisions > 1024,
 * or in a mesh with more than 4096 such triangles -- one that has collapsed onto itself -- it still is.)                                                                                                              */
int  sfx_batch_pen_flags(sfx_batch* b, int32_t* flags_host);
/* Kernel launches of ONE interpenetration step of the fitting loop (broad phase ... adjoint), counted on the graph the batch
 * captured for it most recently; 0 before the first fit with the term (measurement: bench.py roofline_pen.launches_per_round). */
int  sfx_batch_pen_launches(sfx_batch* b);
/* sfx_pen_pairs for GEMM column `column` of the batch's most recent evaluation (resident batches only). */
int  sfx_batch_pen_pairs(sfx_batch* b, int32_t column, int32_t cap, int32_t* pairs_host, int32_t* n_out);

/* Per-frame results of the last sfx_batch_fit (HOST pointers, any may be NULL):
 *  stage_loss [B][1+n_stages]  value run_fitting returns per stage (camera first)
 *  stage_evals [B][1+n_stages] closure evaluations performed per stage
 *  stage_ref_evals             evaluations the reference would have performed            */
int  sfx_batch_get_stats(sfx_batch* b, float* stage_loss, int32_t* stage_evals,
                         int32_t* stage_ref_evals);

/* Optimiser trace (tests of step-level parity with smplifyx/optimizers/lbfgs_ls.py:256-445 and
 * smplifyx/fitting.py:147-217): capacity > 0 attaches a buffer of `capacity` records per frame that every
 * later sfx_batch_fit / sfx_batch_step fills, capacity = 0 detaches it.  A record is 4 floats:
 *   (0, t, loss, ls_evals)                 a line search ended: accepted step length, loss there, its evaluations
 *   (1, entry loss, func_evals, n_iter)    one LBFGS.step returned (cumulative evaluations / iterations of the stage)
 *   (2, result, closure evaluations, stage) run_fitting returned for a stage (camera stage = -1)
 *   (3, trial step, loss, |gradient|inf)   (only with a NEGATIVE capacity, |capacity| records) every closure evaluation
 * sfx_batch_get_trace: records [B][capacity][4] and the number written per frame (HOST; counts may exceed capacity). */
int  sfx_batch_trace(sfx_batch* b, int32_t capacity);
int  sfx_batch_get_trace(sfx_batch* b, float* records, int32_t* counts);

/* Gaussian-mixture body pose prior (MaxMixturePrior, smplifyx/prior.py:100-231; body_prior_type 'gmm'):
 * used by the closure when use_vposer is off and the batch has no regression pose (fitting.py:399-401).
 * means [M][D], precisions [M][D][D], nll_weights [M] = the module's buffers; D = 63, M <= 8 (HOST). */
int  sfx_batch_set_gmm(sfx_batch* b, int32_t M, int32_t D, const float* means, const float* precisions,
                       const float* nll_weights);
/* The same prior in its per-component form (MaxMixturePrior(use_merged=False), prior.py:203-225):
 *   m* = argmin_m [ d_m^T P_m d_m + comp_const_m ],  value = d^T P d + comp_const (at m*) - log nll_weights_m*
 * comp_const [M] = 0.5 (log(det cov_m + epsilon) + D log 2 pi) (HOST); comp_const NULL = the merged form above.   */
int  sfx_batch_set_gmm_form(sfx_batch* b, int32_t M, int32_t D, const float* means, const float* precisions,
                            const float* nll_weights, const float* comp_const);

/* Final meshes / joints at the current parameters (dense LBS; DEVICE pointers, may be NULL). */
int  sfx_batch_forward(sfx_batch* b, float* vertices_out_dev /* [B][V][3] */,
                       float* joints_out_dev /* [B][K][3] */, void* stream);

/* ---- interpenetration term ---------------------------------------------
 * Replaces BVH(max_collisions) + FilterFaces(segm, parents, ign_part_pairs) +
 * DistanceFieldPenetrationLoss(sigma, penalize_outside) of the external mesh_intersection
 * package as the reference uses them (fitting.py:437-455, fit_single_frame.py:300-328), for a
 * batch of posed meshes.  faces [F][3]; segm / parents [F] per-face part labels of
 * smplx_parts_segm.pkl (NULL: no part filter); ign_pairs [n_ign][2] part pairs that never
 * collide; max_collisions = partners kept per triangle; max_batch = frames per call.            */
typedef struct sfx_pen sfx_pen;
int  sfx_pen_create(int32_t V, int32_t F, const int32_t* faces, const int32_t* segm, const int32_t* parents,
                    const int32_t* ign_pairs, int32_t n_ign, int32_t max_collisions, int32_t max_batch,
                    sfx_pen** out);
void sfx_pen_destroy(sfx_pen* h);
/* vertices [B][V][3] (DEVICE) -> loss [B] (DEVICE, unweighted: the caller multiplies by
 * coll_loss_weight) and d loss / d vertices [B][V][3] (DEVICE).  sigma = df_cone_height.        */
int  sfx_pen_eval(sfx_pen* h, int32_t B, const float* verts_dev, float sigma, int32_t penalize_outside,
                  float* loss_dev, float* dverts_dev, void* stream);
/* DistanceFieldPenetrationLoss(point2plane=...) (fit_single_frame.py:311-314, cmd_parser.py:239; every shipped cfg: False).
 * on: every Psi^2 of a pair (f, g) is weighted by (n_f . n_g)^2 -- the repulsion of a vertex measured along the other
 * triangle's normal (oracle/penetration.py, assumption A6) -- in all later evaluations of the handle.               */
int  sfx_pen_set_point2plane(sfx_pen* h, int32_t on);
/* DistanceFieldPenetrationLoss(...)(triangles, collision_idxs) stand-alone (fitting.py:451-455): loss and gradients for pairs the
 * CALLER supplies -- pairs_dev DEVICE int32 [B][n_pairs][2] triangle ids, each unordered pair once, rows with a negative id empty
 * (the package's -1 padding).  Outputs as sfx_pen_eval, plus dtri_dev DEVICE [B][F][3][3] or NULL: the gradient with respect to
 * every triangle CORNER (what autograd needs for the `triangles` tensor).  At most 8192 pairs per mesh.  Synchronises the stream. */
int  sfx_pen_eval_pairs(sfx_pen* h, int32_t B, const float* verts_dev, const int32_t* pairs_dev, int32_t n_pairs, float sigma,
                        int32_t penalize_outside, float* loss_dev, float* dverts_dev, float* dtri_dev, void* stream);
/* per frame (HOST [B][4]): ordered pairs kept, partners dropped by max_collisions / the pair list's capacity, grid entries
 * when they overflowed the buffer (0 = fine; then the frame reports no pairs), bucket walks cut short (0 on a sane mesh:
 * an entry looks at most 2048 entries ahead in its bucket; a mesh folded into a few cells by a diverged fit hits that). */
int  sfx_pen_stats(sfx_pen* h, int32_t B, int32_t* stats_host);
/* The pair list of mesh `mesh` of the most recent evaluation: HOST [cap][2] ordered pairs (receiving triangle, partner), receiver
 * ascending, partner ascending within a receiver, both orders of every colliding pair; *n_out = pairs in the list (the first
 * min(n, cap) are copied).  = collision_idxs after BVH and FilterFaces in the reference (fitting.py:445-450).                  */
int  sfx_pen_pairs(sfx_pen* h, int32_t mesh, int32_t cap, int32_t* pairs_host, int32_t* n_out);
/* Work the term has done since the last reset, counted on the device over every handle of the process (HOST [6]): grid
 * entries, ordered pairs kept, column evaluations (meshes that went through the broad phase), triangles that survived the
 * part culling, triangles that met more partners than the lists hold while they are collected (2 x max_collisions: their kept
 * partners are derived from the grid a second time -- 0 on a sane mesh), bucket walks cut short.  The benchmark's byte model of the step (bench.py
 * roofline_pen) divides the first two by the third.                                                                    */
int  sfx_pen_work_reset(void);
int  sfx_pen_work_get(int64_t* work_host);
/* Host side of the fitting loops (dense mode) since the last reset, HOST [4]: seconds the host thread spent enqueueing launches,
 * seconds it spent waiting for a batch's stage flags, wall seconds of the loops, rounds enqueued.  Wall time far above the kernels'
 * time with little waiting = the queue ran dry behind a slow host (bench.py reports this next to the kernels' times).          */
int  sfx_loop_host_stats(double* stats_host, int32_t reset);

/* Timing hooks for the roofline report: total duration (ms), number of TIMED launches and
 * frames processed by them, of the named kernel since the last reset, measured with HIP events
 * on the launch stream.  name: "lbs_dense", "tick" (k_tick_dense), "fit_rows", "closure",
 * "export", "lbfgs".  sfx_prof_enable(on): bit 0 = time launches; bit 1 = debug, run the
 * fitting loop with the stand-alone kernels; bits 8..23 = N: time every N-th launch of each
 * name only (an event pair costs two queue packets; 0/1 = every launch).                   */
int  sfx_prof_enable(int32_t on);
int  sfx_prof_get(const char* name, double* total_ms, int64_t* launches, double* units /* frames processed */);
void sfx_prof_reset(void);

/* Stand-alone operator (the parity tests of LBFGS.step's direction call it): direction of the device's blocked two-loop
 * recursion for a caller-supplied history (rows of 192 floats, zero padded; `count` pairs pushed in order, the window keeps the last history_size -- <= 0: 100, at most 400) and gradient
 * g[192]; d_out[192].  Host pointers.  Specification: optimizers/lbfgs_ls.py:312-341. */
int  sfx_lbfgs_two_loop(const float* S, const float* Y, int32_t count, int32_t history_size, const float* g, float* d_out);

/* tests / diagnostics: copy a device buffer of the dense path's most recent evaluation to the host, [B][per frame]:
 * "verts", "vposed", "pen_dverts" (V*3), "pen_dfeat", "feat" (512), "pen_dA", "A" (12*55), "pen_loss" (1).
 * n_out = number of floats the caller's buffer holds (checked).                                          */
int  sfx_batch_debug_read(sfx_batch* b, const char* name, float* out, int64_t n_out);

const char* sfx_last_error(void);
const char* sfx_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SFX_H_ */
