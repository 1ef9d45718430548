/*
 * covgpu.h — C ABI of the MI355X-native global-bundle-adjustment / pose-graph back-end.
 *
 * This is the drop-in boundary for the ONE hot path of COVINS that this repository
 * replaces:
 *
 *   covins::Optimization::GlobalBundleAdjustment   reference: covins_backend/include/covins/
 *   covins::Optimization::PoseGraphOptimization               covins_backend/optimization_be.hpp:38-51
 *
 * The reference has no FFI or plugin registry for this path — callers link the two static
 * member functions directly (backend.cpp:141-156, placerec_be.cpp:327, placerec_gen_be.cpp:250).
 * The C++ facade in include/covins_gpu/optimization_gpu.hpp keeps those two signatures verbatim,
 * walks Map/Keyframe/Landmark exactly as optimization_be.cpp does, flattens them into the
 * `covgpu_problem` intermediate representation below, and calls the entry points of this header.
 * Nothing here knows about Map/Keyframe, Eigen, torch or any C++ type: plain pointers and sizes.
 *
 * All floating point is FP64 (reference: precision_t = double, typedefs_base.hpp:129).
 * Quaternions are Hamilton, stored [x,y,z,w]; a pose block is [qx,qy,qz,qw,px,py,pz] = T_w_s
 * (keyframe_base.cpp:486-499); a speed-bias block is [v_w(3), b_a(3), b_g(3)] (:513-521).
 */
#ifndef COVGPU_H_
#define COVGPU_H_

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status codes */
enum {
  COVGPU_OK = 0,
  COVGPU_ERR_INVALID_ARG = 1,   /* malformed problem (index out of range, NULL array, ...)   */
  COVGPU_ERR_NO_DEVICE = 2,     /* no HIP device / HIP runtime failure                        */
  COVGPU_ERR_OUT_OF_MEMORY = 3, /* device allocation failed                                   */
  COVGPU_ERR_NUMERIC = 4,       /* single linear solve not positive definite (covgpu_solve_reduced, covgpu_gn_step) */
  COVGPU_ERR_FATAL_MAP = 5      /* conditions on which the reference calls exit(-1)           */
};

/* trust-region strategy (reference uses DOGLEG: optimization_be.cpp:261,564,1028;
 * BASELINE.json north_star asks for Levenberg-Marquardt as well) */
enum { COVGPU_DOGLEG = 0, COVGPU_LM = 1 };

/* lens distortion of a camera (reference dispatch: optimization_be.cpp:483-521; KeyframeBase can
 * only construct RadTan / Equidistant, keyframe_base.cpp:58-82) */
enum { COVGPU_DIST_RADTAN = 0, COVGPU_DIST_EQUIDISTANT = 1 };

/* ---------------------------------------------------------------- options */
typedef struct covgpu_options {
  int32_t strategy;            /* COVGPU_DOGLEG | COVGPU_LM                                         */
  int32_t max_iterations;      /* trust-region iterations incl. rejected (Ceres max_num_iterations) */
  int32_t visual_only;         /* 1: no speed-bias blocks, no IMU factors (opt_be.cpp:332,367)      */
  int32_t device;              /* HIP device ordinal                                                */
  double  reproj_loss_a;       /* Cauchy scale on reprojection blocks, 1.0 (opt_be.cpp:302);  0=off */
  double  initial_radius;      /* 1e4  (Ceres default initial_trust_region_radius)                  */
  double  max_radius;          /* 1e16                                                              */
  double  min_relative_decrease; /* 1e-3                                                            */
  double  function_tolerance;  /* 1e-6                                                              */
  double  parameter_tolerance; /* 1e-8                                                              */
  double  gradient_tolerance;  /* 1e-10                                                             */
  /* IMU noise, already discretised (orb_slam3/src/Tracking.cc:1203-1211) and gravity magnitude */
  double  sigma_a, sigma_g, sigma_aw, sigma_gw, gravity;
  int32_t verbose;
  int32_t reserved;
} covgpu_options;

/* Fills `o` with the reference's effective settings (dogleg, 10 iterations, Cauchy(1), Ceres 1.x
 * defaults, EuRoC IMU noise at 200 Hz, g = 9.81). */
void covgpu_default_options(covgpu_options* o);

/* ---------------------------------------------------------------- flat problem IR
 * All pointers are HOST pointers owned by the caller; nothing is retained after a call returns.
 * Index arrays are int32. In/out arrays are overwritten with the optimised estimate.            */
typedef struct covgpu_problem {
  int32_t num_kf;        /* K  valid keyframes                                    */
  int32_t num_cam;       /* A  distinct cameras (one per agent in COVINS)         */
  int32_t num_lm;        /* L  landmarks that passed the >=2-observation gate     */
  int32_t num_obs;       /* O  reprojection residual blocks                       */
  int32_t num_imu;       /* I  IMU preintegration factors                         */
  int32_t num_edge;      /* E  SE3 between factors (loops; PGO odometry edges)    */
  int32_t num_imu_samples; /* S total raw IMU samples over all factors           */
  int32_t reserved;

  /* keyframes */
  double*        kf_pose;        /* [K][7]  in/out  T_w_s                          */
  double*        kf_speed_bias;  /* [K][9]  in/out  (NULL allowed if visual_only)  */
  const uint8_t* kf_fixed;       /* [K]     1 = pose block constant (opt_be.cpp:329-341, 870-881) */
  const int32_t* kf_cam;         /* [K]     camera index                           */

  /* cameras: extrinsics T_s_c, intrinsics fx fy cx cy, 4 distortion coefficients; all constant
   * in every call site (opt_be.cpp:336,349,352) */
  const double*  cam_extr;       /* [A][7] */
  const double*  cam_intr;       /* [A][4] */
  const double*  cam_dist;       /* [A][4] */
  const int32_t* cam_dist_type;  /* [A]    */

  /* landmarks + observations. Observations are grouped by landmark (the reference's iteration
   * order, opt_be.cpp:432-530): those of landmark l are [lm_obs_ptr[l], lm_obs_ptr[l+1]). */
  double*        lm_pos;         /* [L][3]  in/out  world position                 */
  const int32_t* lm_obs_ptr;     /* [L+1]                                          */
  const int32_t* obs_kf;         /* [O]                                            */
  const double*  obs_uv;         /* [O][2]  keypoint, float32 promoted to double (opt_be.cpp:477) */
  const double*  obs_sigma;      /* [O]     (octave+1)*2 px (opt_be.cpp:478)       */

  /* IMU factors between predecessor kf_i and successor kf_j (opt_be.cpp:369-416). The raw samples
   * are part of the IR because the reference re-propagates the preintegration with the current
   * bias estimate once per solve (opt_be.cpp:396). Sample = [dt, ax,ay,az, wx,wy,wz]. */
  const int32_t* imu_kf_i;       /* [I] */
  const int32_t* imu_kf_j;       /* [I] */
  const int32_t* imu_sample_ptr; /* [I+1] */
  const double*  imu_samples;    /* [S][7] */
  const double*  imu_first;      /* [I][6]  (acc_0, gyr_0): reading at the predecessor (keyframe_be.cpp:187,195) */
  /* Per-factor IMU calibration [sigma_a, sigma_g, sigma_aw, sigma_gw, gravity]: the reference builds every keyframe's
   * preintegrator from THAT keyframe's own calibration (keyframe_be.cpp:187-195: sigma_a_c, sigma_g_c, sigma_aw_c,
   * sigma_gw_c, g), so mixed agents / IMUs keep their own weights. NULL: the five values of covgpu_options apply to
   * every factor. A row with a non-positive sigma or gravity < 9 (keyframe_base.cpp:51-55) is COVGPU_ERR_INVALID_ARG.
   * The values (given or taken from the options) are captured by covgpu_upload / covgpu_gba_solve: the sigma_* / gravity
   * fields of the options passed to a later covgpu_solve_resident are not read again. */
  const double*  imu_noise;      /* [I][5] or NULL */

  /* SE3 between factors (robopt SixDofBetweenError, kImu): measurement T_s1_s2 as [q(4), t(3)],
   * row-major 6x6 sqrt-information (rotation rows first), Cauchy scale per edge (0 = no loss). */
  const int32_t* edge_i;         /* [E] */
  const int32_t* edge_j;         /* [E] */
  const double*  edge_meas;      /* [E][7] */
  const double*  edge_sqrt_info; /* [E][36] */
  const double*  edge_loss_a;    /* [E] */
} covgpu_problem;

/* per-iteration trace, for parity tests against the oracle */
#define COVGPU_MAX_TRACE 64
typedef struct covgpu_result {
  int32_t iterations;          /* trust-region iterations executed                          */
  int32_t accepted;            /* of which successful                                       */
  int32_t termination;         /* 0 max-iter, 1 function tol, 2 parameter tol, 3 gradient tol, 4 failure: the reduced system stayed
                                * non-positive-definite up to the largest damping. Like ceres::Solve (whose summary the reference
                                * ignores, opt_be.cpp:567) the call still returns COVGPU_OK and the LAST ACCEPTED estimate; callers that
                                * care check this field. COVGPU_ERR_NUMERIC is returned by covgpu_solve_reduced / covgpu_gn_step only. */
  int32_t reserved;            /* IMU factors dropped because their preintegrated covariance was not positive definite (0 samples) */
  double  initial_cost;
  double  final_cost;
  double  t_upload_s;          /* H2D incl. layout build                                    */
  double  t_solve_s;           /* device-resident solve (the timed region of bench.py)      */
  double  t_download_s;        /* D2H                                                       */
  double  t_linear_solve_s;    /* part of t_solve_s spent in the reduced-system factor+solve */
  double  cost_trace[COVGPU_MAX_TRACE];    /* cost after each iteration */
  double  radius_trace[COVGPU_MAX_TRACE];
  int32_t accepted_trace[COVGPU_MAX_TRACE];
} covgpu_result;

/* ---------------------------------------------------------------- context */
typedef struct covgpu_context covgpu_context;

/* One context per calling thread / map (the reference may run PGO for different maps on
 * different place-recognition threads, placerec_be.cpp:295). Owns one HIP stream + workspace. */
int  covgpu_create(const covgpu_options* opt, covgpu_context** out);
void covgpu_destroy(covgpu_context* ctx);
const char* covgpu_last_error(void);

/* ---------------------------------------------------------------- solve entry points
 * covgpu_gba_solve   replaces the ceres::Solve of GlobalBundleAdjustment (opt_be.cpp:560-567 and,
 *                    with max_iterations = 5, the outlier round's :257-265).
 * covgpu_pgo_solve   replaces the ceres::Solve of PoseGraphOptimization (opt_be.cpp:1024-1031):
 *                    poses + between factors only (num_lm = num_obs = num_imu = 0).
 * Both upload `p`, run the trust-region loop fully on the device, and write the optimised
 * kf_pose / kf_speed_bias / lm_pos back into the caller's arrays.                               */
int covgpu_gba_solve(covgpu_context* ctx, const covgpu_options* opt, covgpu_problem* p, covgpu_result* out);
int covgpu_pgo_solve(covgpu_context* ctx, const covgpu_options* opt, covgpu_problem* p, covgpu_result* out);

/* Split form used by bench.py so that inputs are HBM-resident before the timed region:
 * upload once, solve (restarts from the uploaded initial estimate every call), download. */
int covgpu_upload(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p);
int covgpu_upload_pgo(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p);
int covgpu_solve_resident(covgpu_context* ctx, const covgpu_options* opt, covgpu_result* out);
int covgpu_download(covgpu_context* ctx, covgpu_problem* p);

/* GBA outlier rule (opt_be.cpp:270-290): evaluates every reprojection block at the current
 * host estimate in `p` and writes the loss-corrected whitened residual norm per observation. */
int covgpu_reprojection_residual_norms(covgpu_context* ctx, const covgpu_options* opt,
                                       const covgpu_problem* p, double* norms /* [O] */);

/* Map maintenance of the outlier round on the device: the erase decisions of opt_be.cpp:276-289
 * and the bookkeeping Map::Clean / RemoveLandmarkOutliers needs (map_be.cpp:448-454, 698-743), evaluated at the estimate
 * RESIDENT on the device after covgpu_gba_solve / covgpu_solve_resident — no second upload, no O doubles over PCIe.
 *   obs_erase[o] = 1 iff the loss-corrected whitened residual norm of observation o exceeds `threshold`
 *                  (kf->EraseLandmark + lm->EraseObservation in the reference)
 *   lm_left[l]   = observations landmark l keeps; < 2 -> RemoveLandmarkOutliers drops it
 *   counts[0..1] = erased observations, landmarks left with fewer than two                                   */
int covgpu_outlier_pass(covgpu_context* ctx, double threshold, uint8_t* obs_erase /* [O] */, int32_t* lm_left /* [L] */,
                        int64_t* counts /* [2] or NULL */);

/* Both rounds of Optimization::GlobalBundleAdjustment (optimization_be.cpp:56-618) behind ONE call: the outlier round (:62-265,
 * max_num_iterations = 5, loop edges without loss), the erase decisions (:270-290), and the main round (:296-567) on the problem the
 * reference REBUILDS from the map after erasing — here derived on the device from the resident first round: the observation stream
 * minus the erased observations, minus the landmarks left with fewer than two (:428-440), the loop edges' loss switched on (:555),
 * optionally more constant poses (:338-341); restarted from the same initial estimate (the outlier round's is discarded, as in the
 * reference). One Map -> IR flatten and one upload per call instead of two.
 *   p            the FIRST round's problem (edge_loss_a as the first round wants it: 0). On return kf_pose / kf_speed_bias hold
 *                the second round's estimate, lm_pos[l] the second round's for every landmark with lm_left[l] >= 2
 *   opt          options of the second round (max_iterations = opt.gba_iteration_limit)
 *   obs_erase, lm_left, counts   as covgpu_outlier_pass, for the caller's map bookkeeping (kf->EraseLandmark, lm->EraseObservation)
 *   round1, round2 (may be NULL)  the two trust-region records; round2->t_upload_s = seconds of the device-side rebuild
 * Afterwards the context holds NO resident problem (what is resident is the second round's compacted problem, which `p` does not
 * describe): covgpu_solve_resident / covgpu_download / covgpu_outlier_pass need a new covgpu_upload first.                        */
typedef struct covgpu_two_round {
  double  outlier_threshold;         /* opt.th_gba_outlier_global (config_backend.yaml:121)                                  */
  int32_t round1_iterations;         /* 5 (optimization_be.cpp:262); <= 0: 5                                                 */
  int32_t use_loops_round2;          /* opt.gba_use_map_loop_constraints (:534)                                              */
  double  loop_loss_round2;          /* Cauchy scale of the loop edges in the second round: 1.0 (:555)                       */
  const uint8_t* kf_fixed_round2;    /* [num_kf] constant poses of the second round (opt.gba_fix_poses_loaded_maps, :338-341) or NULL: as round 1 */
} covgpu_two_round;
int covgpu_gba_two_round(covgpu_context* ctx, const covgpu_options* opt, covgpu_problem* p, const covgpu_two_round* tr,
                         uint8_t* obs_erase /* [O] */, int32_t* lm_left /* [L] */, int64_t* counts /* [2] or NULL */,
                         covgpu_result* round1, covgpu_result* round2);

/* Covisibility recount on the resident problem (Keyframe::UpdateCovisibilityConnections, keyframe_be.cpp:559-608; run over all
 * keyframes after a GBA, backend.cpp:164-167): weight(i, j) = number of landmarks both keyframes observe.
 * Pairs with weight >= threshold (sys.covis_thres) are returned as kf_i > kf_j, sorted by (kf_i, kf_j); constant keyframes
 * take part. *count = pairs found (call again with a larger capacity if it exceeds it). Counted on the device (k_pairs.hip). */
int covgpu_covisibility(covgpu_context* ctx, int32_t threshold, int64_t capacity, int32_t* kf_i, int32_t* kf_j, int32_t* weight,
                        int64_t* count);

/* Batched Optimization::OptimizeRelativePose (optimization_be.cpp:620-831): refines the relative pose
 * T_AB of MANY keyframe pairs (loop candidates, placerec_be.cpp:116-165) in one launch, one wavefront per pair. Per
 * correspondence two reprojection residuals — kNormal: pi_A(R_AB P_B + t_AB) vs kp_A, kInverse: pi_B(R_AB^T (P_A - t_AB))
 * vs kp_B — with sigma = (octave + 1) * 2 and Cauchy(1); DOGLEG 5 iterations, correspondences whose loss-corrected residual
 * norm exceeds th_outlier in either image are dropped, fewer than min_inliers (reference: 12) left -> inliers = 0 and T_ab
 * untouched, else 5 more iterations. NB with Cauchy(1) the corrected norm is < 1, so the reference's configured
 * opt.th_outlier_align = 1.3 never removes anything; that behaviour is reproduced as is.
 * Pairs b = 0..num-1 own correspondences [corr_ptr[b], corr_ptr[b+1]). cam_* rows: fx fy cx cy d0 d1 d2 d3. All HOST pointers. */
typedef struct covgpu_relpose_batch_t {
  int32_t num_pairs;
  const int32_t* corr_ptr;      /* [num_pairs + 1] */
  const double*  p_b;           /* [C][3] landmark of kf2 in camera-B frame (opt_be.cpp:655-656) */
  const double*  p_a;           /* [C][3] landmark of kf1 in camera-A frame (:653-654)           */
  const double*  kp_a;          /* [C][2] */
  const double*  kp_b;          /* [C][2] */
  const double*  sigma_a;       /* [C]    */
  const double*  sigma_b;       /* [C]    */
  const double*  cam_a;         /* [num_pairs][8] */
  const double*  cam_b;         /* [num_pairs][8] */
  const int32_t* dist_type_a;   /* [num_pairs] COVGPU_DIST_* */
  const int32_t* dist_type_b;   /* [num_pairs] */
  double*        T_ab;          /* [num_pairs][7] in/out */
  uint8_t*       outlier;       /* [C] out: 1 = correspondence removed (matches1[i] = NULL, :807) */
  int32_t*       inliers;       /* [num_pairs] out: the function's return value per pair */
} covgpu_relpose_batch_t;
int covgpu_relpose_batch(covgpu_context* ctx, const covgpu_relpose_batch_t* batch, double th_outlier, int32_t min_inliers);

/* PGO tail (opt_be.cpp:1046-1047, 1066-1081): rotate velocities and re-anchor every landmark to
 * its reference keyframe: p' = T_ws_new(ref) * T_ws_old(ref)^-1 * p.  ref_kf[l] < 0 skips l. */
int covgpu_pgo_reanchor(covgpu_context* ctx, int32_t num_kf, const double* pose_old /* [K][7] */,
                        const double* pose_new /* [K][7] */, double* velocity /* [K][3] or NULL */,
                        int32_t num_lm, const int32_t* ref_kf /* [L] */, double* lm_pos /* [L][3] */);

/* ---------------------------------------------------------------- per-kernel test entry points
 * (SURVEY.md §8b last row). Each runs exactly one device kernel family on host inputs and returns
 * raw, un-reduced outputs so tests/ can compare them with the oracle.                           */

/* R4+R5+R6: per observation r[2], J_pose[2x6], J_lm[2x3] (whitened, loss-corrected), cost rho/2. */
int covgpu_linearize_reprojection(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p,
                                  double* r /* [O][2] */, double* J_pose /* [O][12] */,
                                  double* J_lm /* [O][6] */, double* cost /* [O] */);

/* R2: per IMU factor delta [dp(3), dq(4), dv(3), dt_sum] (11), bias Jacobian J[15x15],
 * covariance P[15x15], preintegrated at the linearisation bias = speed_bias[kf_j][3:9]. */
int covgpu_preintegrate(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p,
                        double* delta /* [I][11] */, double* J /* [I][225] */, double* P /* [I][225] */);

/* R3: per IMU factor whitened residual r[15] and Jacobians w.r.t. [pose_i(6), sb_i(9), pose_j(6),
 * sb_j(9)] as one row-major 15x30 block. */
int covgpu_linearize_imu(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p,
                         double* r /* [I][15] */, double* J /* [I][450] */);

/* R7: per edge residual r[6] and row-major 6x12 Jacobian w.r.t. [pose_i(6), pose_j(6)]
 * (whitened by sqrt_info, loss-corrected), cost. */
int covgpu_linearize_between(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p,
                             double* r /* [E][6] */, double* J /* [E][72] */, double* cost /* [E] */);

/* R8 building block: damped Schur complement at the current estimate.
 * Outputs the dense reduced system S (n x n, row-major, full symmetric) and b (n),
 * n = dim_per_kf * K with dim_per_kf = 15 (VI) or 6 (visual_only / PGO), plus total cost. */
int covgpu_schur(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p, double mu,
                 double* S, double* b, double* cost);
/* same for the pose-graph problem (edges only, 6 rows per keyframe) */
int covgpu_schur_pgo(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p, double mu,
                     double* S, double* b, double* cost);

/* R8 building block: ONE damped Gauss-Newton step at the estimate in `p`, through the product solve path (speed-bias
 * multifrontal MFMA Cholesky of the reduced camera system over its elimination tree -> landmark back-substitution):
 * dx[n] (IR layout, dim_per_kf per keyframe) and dl[3L]. Tests check S dx = b against the oracle's system at full size. */
int covgpu_gn_step(covgpu_context* ctx, const covgpu_options* opt, const covgpu_problem* p, double mu,
                   double* dx /* [n] */, double* dl /* [L][3] */, double* cost);

/* R8 building block: dense FP64 Cholesky solve of S x = b on the MFMA path (S symmetric positive
 * definite, row-major n x n; only the lower triangle is read). Returns COVGPU_ERR_NUMERIC if a
 * pivot is not positive. */
int covgpu_solve_reduced(covgpu_context* ctx, int32_t n, const double* S, const double* b, double* x);

int32_t covgpu_reduced_dim(const covgpu_options* opt, const covgpu_problem* p);

/* Host-only: the block partition of round 2's block-arrow pose-graph solve (PoseGraphOptimization's linear solver,
 * optimization_be.cpp:1024-1031; since round 6 the default is the multifrontal solve on the pose graph's own elimination
 * tree — DESIGN.md 4.8 — and this scheme runs with COVGPU_PGO_ND=0). block_of_kf[k] = block index (>= 0) of
 * keyframe k, or -1 if it belongs to the border (loop-closure keyframes and separators). Returns the number of
 * blocks, or 0 if the graph is solved densely (too small, no split, or a border that is a large part of the
 * system; block_of_kf is then all -1). No edge joins two different blocks. Needs no device. */
int32_t covgpu_pgo_partition(int32_t num_kf, int32_t num_edge, const int32_t* edge_i, const int32_t* edge_j, int32_t* block_of_kf);

/* Host-only (no device needed): the nested-dissection plan of the reduced camera system — the elimination tree the
 * multifrontal MFMA Cholesky (covins_amd/csrc/k_front.hip) runs, i.e. what the reference gets from CHOLMOD's fill-reducing
 * ordering inside ceres::Solve(SPARSE_SCHUR) (optimization_be.cpp:560-565). Unknowns are, per keyframe, a 6-dim pose block
 * (variable 2k) and a 9-dim speed-bias block (variable 2k+1, visual-inertial only). Every tree node owns some variables
 * (eliminated there) and carries the ancestor variables its subtree couples to; nodes of equal height form one batch.
 * leaf_dims <= 0: default (COVGPU_ND_LEAF or 600 scalar unknowns per leaf). tests/test_nd_plan.py replays the plan in numpy. */
typedef struct covgpu_nd_plan covgpu_nd_plan;
int  covgpu_nd_plan_create(const covgpu_options* opt, const covgpu_problem* p, int32_t leaf_dims, covgpu_nd_plan** out);
/* the same for a pose-graph problem (edges only): the plan covgpu_pgo_solve runs on since round 6 — 6-dof blocks, an agent's time axis read from the
 * edge graph (connected components of the graph without its bridges, each in breadth-first order: solver.hip build_chains_pgo) */
int  covgpu_nd_plan_create_pgo(const covgpu_options* opt, const covgpu_problem* p, int32_t leaf_dims, covgpu_nd_plan** out);
void covgpu_nd_plan_destroy(covgpu_nd_plan* plan);
/* out16 = { nodes, levels, depth, own entries, front-structure entries, front elements over all batches, flops of the partial
 *           factorisations, largest own dims, largest border dims, largest root, 0... } */
void covgpu_nd_plan_info(const covgpu_nd_plan* plan, int64_t* out16);
/* parent / level [nodes]; own_ptr / st_ptr [nodes + 1]; own_var / st_var: 2 * keyframe + (0 pose | 1 speed-bias) */
void covgpu_nd_plan_arrays(const covgpu_nd_plan* plan, int32_t* parent, int32_t* level, int32_t* own_ptr, int32_t* own_var,
                           int32_t* st_ptr, int32_t* st_var);

/* ---------------------------------------------------------------- multi-GPU: ONE map sharded by sub-map
 * BASELINE.json north star: "the merged multi-agent map shards by agent/sub-map across the GPUs of one node with RCCL
 * all-reduce over xGMI on the shared-pose Hessian blocks at each LM iteration". One context per GPU (one process per GPU, or
 * several contexts in one process). The unit of the split is a SUBTREE of the elimination tree (an agent, or a stretch of an
 * agent's trajectory); the top of the tree — the separators that join the sub-maps: the "shared" poses — is replicated.
 *   1. every rank calls covgpu_shard_plan on the FULL problem (host-only, deterministic: the same answer everywhere);
 *   2. rank r keeps the landmarks / IMU factors / between factors with *_rank == r (all keyframes stay: K is unchanged),
 *      attaches a collective (covgpu_set_shard_rccl: RCCL, unique id from covgpu_rccl_unique_id on rank 0 passed to the
 *      others by the caller; covgpu_set_shard_group: host threads of one process whose contexts share a device), then
 *      covgpu_upload / covgpu_solve_resident / covgpu_download on that sub-problem;
 *   3. per trust-region iteration the library issues FOUR all-reduces, all enqueued on the context's stream (no host
 *      synchronisation): inside the linear solve ONE over [top fronts | their right-hand sides | gradient and diag(J^T J) of
 *      the top unknowns] after every rank has eliminated its own subtrees, and three of 16 + 2 x world scalars;
 *   4. after the solve an unknown is valid on the rank that owns its tree node (covgpu_nd_plan_owner; top unknowns: on
 *      every rank), a landmark on its lm_rank. */
int32_t covgpu_shard_plan(const covgpu_options* opt, const covgpu_problem* p, int32_t world, covgpu_nd_plan** plan_out,
                          int32_t* lm_rank /* [L] */, int32_t* imu_rank /* [I] */, int32_t* edge_rank /* [E] */);  /* returns the number of subtrees, 0: no split */
void covgpu_nd_plan_owner(const covgpu_nd_plan* plan, int32_t* pose_rank /* [K] */, int32_t* sb_rank /* [K] */);   /* -1: top unknown */
void covgpu_nd_plan_ranks(const covgpu_nd_plan* plan, int32_t* node_rank /* [nodes] */);                          /* -1: top node */
typedef struct covgpu_group covgpu_group;   /* in-process group of ranks (host threads), at most 16 */
int  covgpu_group_create(int32_t world, covgpu_group** out);
void covgpu_group_destroy(covgpu_group* g);
void covgpu_group_abort(covgpu_group* g);   /* a failing member releases the others from their barrier */
int  covgpu_set_shard_group(covgpu_context* ctx, const covgpu_nd_plan* plan, int32_t rank, covgpu_group* g);
int  covgpu_rccl_unique_id(uint8_t* out128);   /* ncclGetUniqueId; librccl is loaded on first use */
int  covgpu_set_shard_rccl(covgpu_context* ctx, const covgpu_nd_plan* plan, int32_t rank, int32_t world, const uint8_t* id128);
int  covgpu_set_shard_none(covgpu_context* ctx);   /* back to the single-GPU form */
/* The whole sharded solve behind one call, for a host process that drives several GPUs itself (covins_backend is one
 * process: backend.cpp:141-156): plan on the full problem, one context + one host thread per rank on devices[r], RCCL
 * between them (the in-process group if ranks share a device: virtual ranks), results merged into `p`; `out` = the common
 * trust-region record. obs_erase != NULL: also the outlier decisions of optimization_be.cpp:270-290 at the resident
 * estimate, merged over the ranks (obs_erase [O], lm_left [L], counts[2] as covgpu_outlier_pass). */
int  covgpu_gba_solve_multi(const covgpu_options* opt, covgpu_problem* p, covgpu_result* out, int32_t n_ranks, const int32_t* devices,
                            double outlier_threshold, uint8_t* obs_erase, int32_t* lm_left, int64_t* counts);
/* Both rounds of a GlobalBundleAdjustment call (optimization_be.cpp:56-618) on n_ranks devices behind one call (round 6): the plan on the full problem, ONE
 * upload per rank, the second round derived on every rank's device from its own share exactly as covgpu_gba_two_round derives it on one GPU (a
 * landmark's outlier decisions, the compaction and the pair lists are local to its rank; the elimination tree stays); obs_erase [O], lm_left [L],
 * counts[2] and the second round's estimate merged into `p` as covgpu_gba_solve_multi merges them. */
int  covgpu_gba_two_round_multi(const covgpu_options* opt, covgpu_problem* p, const covgpu_two_round* tr, int32_t n_ranks, const int32_t* devices,
                                uint8_t* obs_erase, int32_t* lm_left, int64_t* counts, covgpu_result* round1, covgpu_result* round2);
int  covgpu_allreduce_host(covgpu_context* ctx, double* host, int64_t n, int32_t op /* 0 sum, 1 max */);  /* through the context's collective */
void covgpu_shard_stats(covgpu_context* ctx, int64_t* out4);  /* collectives issued, bytes all-reduced, rank, world */

/* ---------------------------------------------------------------- measurement hooks (bench.py)
 * With profiling on, covgpu_solve_resident brackets the linearise+Schur pass, the whole factor+solve and
 * every trailing-update (SYRK) launch with HIP events on the context's own stream.
 * out[8] = { build ms, #builds, factor+solve ms, #factorisations, SYRK ms, #SYRK launches, SYRK flops,
 *           #off-diagonal 6x6 pose-pose blocks of the reduced system (covisible + loop-edge keyframe pairs) } */
void covgpu_set_profiling(covgpu_context* ctx, int on);
/* layout of the uploaded problem: out[16] = { ranks of a sharded solve (0: single GPU), this rank, scalar unknowns of the
 * replicated top nodes, top levels, KiB all-reduced per linear solve, how the context orders its streams (1 device flags, 0 HIP events
 * by COVGPU_GATES=0, -1 events and the launch-per-tile backward substitution after a flag gate or a hand-over inside the pipelined
 * backward substitution timed out), dense pose order (padded), covisible keyframe pairs,
 * edge pairs, IMU chains, device MiB allocated for the problem (from the allocator, not by hand), fronts, levels, serial
 * 256-column panels, order of the last level's fronts, MiB of fronts } */
void covgpu_get_layout(covgpu_context* ctx, int64_t* out16);
void covgpu_get_profile(covgpu_context* ctx, double* out8);
/* out16: [0..7] as covgpu_get_profile; [8] k_potrf_panel (the serial panel chain of the front factorisation) milliseconds summed over
 * its launches, [9] its launches, [10] its algorithmic flops (per front n^3/3 + n^2 on the front's real columns in the panel),
 * [11] flops of one multifrontal factorisation of the resident problem (dense count on the fronts' real sizes), [12..15] 0 */
void covgpu_get_profile2(covgpu_context* ctx, double* out16);

#ifdef __cplusplus
}
#endif
#endif /* COVGPU_H_ */
