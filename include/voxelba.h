/*
 * voxelba.h — C ABI of the MI355X-native Voxel-SLAM local-mapping hot path (libvoxelba.so).
 *
 * The reference (Wangshihu12/Voxel-SLAM, /root/reference/VoxelSLAM/src) has no FFI/plugin boundary: the
 * path is reached through C++ member calls on header-only classes.  Each entry point
 * below names the reference interface it replaces (VM = voxel_map.hpp, VS = voxelslam.cpp,
 * TL = tools.hpp, PI = preintegration.hpp).  include/voxelba_adapter.hpp wraps these calls back into
 * the reference's class/method names (LidarFactor, Lidar_BA_Optimizer, LI_BA_Optimizer, ...).
 *
 * Conventions
 *   - All numeric arrays are IEEE double, caller-owned, HOST memory unless a name ends in _dev.
 *   - Every function returns an int status (VBA_OK = 0) where the reference would printf+exit(0)
 *     (VM:401-402, VM:1490-1491) or silently return; no function throws.
 *   - A context is re-entrant per handle: one HIP stream per vba_ctx, no process-wide mutable state
 *     (the reference's globals VM:98-104, VM:500, VM:1046 are fields of vba_options / the context).
 *   - The library has NO CPU compute fallback: without a HIP device vba_create fails with
 *     VBA_ERR_NO_DEVICE.
 *
 * Flat layouts
 *   cluster : [Pxx,Pxy,Pxz,Pyy,Pyz,Pzz, vx,vy,vz, N]    (10)   PointCluster TL:304-310 (P symmetric)
 *   pose    : [R(9) row-major, p(3)]                    (12)   IMUST::R, IMUST::p  TL:139-140
 *   state   : [t, R(9), p(3), v(3), bg(3), ba(3), g(3)] (25)   IMUST TL:135-144 (cov passed separately)
 *   imu_pre : [R_delta(9) p_delta(3) v_delta(3) bg(3) ba(3) R_bg(9) p_bg(9) p_ba(9) v_bg(9) v_ba(9)
 *              dtime dbg(3) dba(3) dbg_buf(3) dba_buf(3) cov(225)]  (304)   IMU_PRE PI:15-28
 *   3x3 / NxN matrices are row-major; eigenvector matrices hold eigenvectors in COLUMNS (VM:193).
 */
#ifndef VOXELBA_H
#define VOXELBA_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VBA_CLUSTER_LEN 10
#define VBA_POSE_LEN 12
#define VBA_STATE_LEN 25
#define VBA_IMU_PRE_LEN 304
#define VBA_DIM 15 /* TL:16 */
#define VBA_MAX_WIN 16

enum vba_status {
  VBA_OK = 0,
  VBA_ERR_NO_DEVICE = 1,       /* no HIP device / kernel image: the product has no CPU path */
  VBA_ERR_BAD_ARG = 2,
  VBA_ERR_UNSUPPORTED_WINDOW = 3,
  VBA_ERR_TOO_FEW_VOXELS = 4,  /* Lidar_BA_Optimizer::only_residual "Too Less Voxel" exit(0), VM:399-403 */
  VBA_ERR_OPT_STATE = 5,       /* OctoTree::margi "Error: opt_state" exit(0), VM:1488-1492 */
  VBA_ERR_HIP = 6,
  VBA_ERR_CAPACITY = 7,
  VBA_ERR_IO = 8,              /* file missing / malformed (read_lidarstate prints "not found" and exits, VH:271-275) */
  VBA_ERR_UNSUPPORTED = 9      /* the in-library RCCL exchange step was asked for but librccl.so.1 cannot be resolved in this process */
};

typedef struct vba_ctx vba_ctx;

/* Replaces the process-wide configuration of the reference: voxel_size, min_eigen_value, max_layer,
 * max_points, plane_eigen_value_thre, min_point (VM:98-104), imu_coef (VM:500), LocalBA/win_size and
 * thread_num (VS:875-931).  plane_eigen_value_thre is passed ALREADY INVERTED as the reference stores
 * it (VS:930-931). */
typedef struct vba_options {
  int win_size;                     /* LocalBA/win_size */
  double voxel_size;                /* Odometry/voxel_size */
  int max_layer;                    /* VM:100 */
  int max_points;                   /* VM:101 */
  double min_eigen_value;           /* VM:99 */
  double plane_eigen_value_thre[4]; /* VM:104, inverted */
  double min_point[4];              /* VM:98 */
  double imu_coef;                  /* VM:500 */
  int thread_num;                   /* only for the "#voxels < thread_num" early-return quirks (VM:2044, VS:1616, VS:1693) */
  int device;                       /* HIP device ordinal, -1 = current */
  void *stream;                     /* hipStream_t to run on, NULL = the context creates its own */
  size_t max_voxels;                /* factor capacity hint (0 = grow on demand) */
  size_t max_points_per_scan;       /* map capacity hint (0 = grow on demand) */
  /* execution knobs (no reference counterpart; 0 = default).  They replace the environment switches of earlier builds. */
  int lm_spec;                      /* damping candidates per solve launch, 1..4 (default 4; 1 = the plain sequential solve) */
  int force_collective;             /* != 0: take the exchange step of the sharded LM flow with ONE rank too (rehearsals) */
  int hessian_workgroups;           /* persistent workgroups of the Hessian pass, 2..256 (default 256 = one per CU) */
  int residual_vpl_from;            /* residual pass: stores with more voxels use the voxel-per-lane kernel (default 45000) */
  int hessian_compact_tiles;        /* != 0: Hessian pass on occupancy-compact tiles (k_hessian3, W <= 10) instead of dense fixed-size ones; measured slower at the bench size */
  size_t max_map_nodes;             /* map capacity hints (0 = grow on demand): octree nodes and fixed (marginalised) points the map is */
  size_t max_fix_points;            /* sized for at the first insertion — a session that stays below them never re-allocates (no stalls) */
  int hba_workers;                  /* vba_hba_global on one rank: bottom-layer windows optimised side by side by this many worker contexts (0 = default 4, 1 = one after the other) */
} vba_options;

void vba_default_options(vba_options *opt); /* values of config/avia.yaml:26-47 */
int vba_create(const vba_options *opt, vba_ctx **out);
void vba_destroy(vba_ctx *ctx);
const char *vba_status_string(int status);
const char *vba_last_error(vba_ctx *ctx);
int vba_synchronize(vba_ctx *ctx);

/* ------------------------------------------------------------------------------------------------
 * Factor level — drop-in for class LidarFactor (VM:124-339).
 * The factor store lives in HBM as SoA [field][frame][voxel].                        */

/* LidarFactor::clear (VM:328-336) */
int vba_factor_clear(vba_ctx *ctx);
/* LidarFactor::push_voxel (VM:139-147), batched: n voxels appended.
 * clusters [n][W][10], fix [n][10], coe [n], eig_val [n][3], eig_vec [n][9], pcr_add [n][10]. */
int vba_factor_push_voxels(vba_ctx *ctx, int n, const double *clusters, const double *fix, const double *coe,
                           const double *eig_val, const double *eig_vec, const double *pcr_add);
/* plvec_voxels.size() (read by callers at VS:706) */
int vba_factor_size(vba_ctx *ctx);
/* LidarFactor::acc_evaluate2 (VM:150-282) over voxels [head,end): Hess (6W x 6W), JacT (6W), residual. */
int vba_factor_acc_evaluate2(vba_ctx *ctx, const double *poses, int head, int end, double *Hess, double *JacT,
                             double *residual);
/* LidarFactor::evaluate_only_residual (VM:285-325) over [head,end); updates the device-side
 * eig_values / eig_vectors / pcr_adds exactly as the reference does (VM:317-319). */
int vba_factor_evaluate_only_residual(vba_ctx *ctx, const double *poses, int head, int end, double *residual);
/* Public data members eig_values / eig_vectors / pcr_adds read by OctoTree::margi (VM:1495-1501) and
 * motion_init (VS:737).  Any pointer may be NULL. */
int vba_factor_read_back(vba_ctx *ctx, double *eig_val, double *eig_vec, double *pcr_add);

/* Number of occupied (voxel, frame) slots (clusters with N != 0) in the factor store — the "slots" that the
 * algorithmic-traffic figures of DESIGN.md are priced on. */
int vba_factor_occupied_slots(vba_ctx *ctx, long long *slots);
/* Occupancy mask of every stored voxel (bit i = frame i of the window sees it), in store order: `masks` holds vba_factor_size()
 * entries.  After vba_map_recut the store is in non-decreasing order of the mask's low 10 bits. */
int vba_factor_occupancy_masks(vba_ctx *ctx, unsigned int *masks);

/* ------------------------------------------------------------------------------------------------
 * Optimizers — drop-in for the three LM classes.                                                  */

/* bool Lidar_BA_Optimizer::damping_iter(x_stats, voxhess, hess, resis, max_iter, is_display) (VM:422-497).
 * poses [W][12] in/out; hess (6W)^2 out (may be NULL); resis2[2] = {first, last} appended values;
 * thd_num = Lidar_BA_Optimizer::thd_num (VM:345) — only used for the V < thd_num check (VM:399).
 * *is_converge receives the bool return value. */
int vba_lidar_ba_damping_iter(vba_ctx *ctx, double *poses, double *hess, double *resis2, int max_iter, int thd_num,
                              int *is_converge);

/* void LI_BA_Optimizer::damping_iter(x_stats, voxhess, imus_factor, hess) (VM:624-713) when gravity == 0;
 * void LI_BA_OptimizerGravity::damping_iter(x_stats, voxhess, imus_factor, resis, hess, max_iter) (VM:878-975)
 * when gravity != 0.  states [W][25] in/out, imus [W-1][304] in/out (dbg/dba/dbg_buf/dba_buf are updated,
 * PI:296-303), hess ((15W + 3*gravity)^2) out (may be NULL), resis2 out (gravity variant only, may be NULL). */
int vba_li_ba_damping_iter(vba_ctx *ctx, double *states, double *imus, int gravity, int max_iter, double *hess,
                           double *resis2);

/* Optional iteration trace of the last damping_iter on this context: rows [r1, r2, u, v, q1] as used in each
 * executed iteration.  Returns the number of rows written (<= max_rows). */
int vba_last_lm_trace(vba_ctx *ctx, double *rows, int max_rows);

/* IMU_PRE(bg, ba) + IMU_PRE::push_imu (PI:32-73) with the noise globals noiseMeas / noiseWalk /
 * imupre_scale_gravity (PI:8-9) passed as diagonals: samples t[n], gyr[n][3], acc[n][3] -> imu_pre[304]. */
int vba_imu_preintegrate(int n, const double *t, const double *gyr, const double *acc, const double *bg,
                         const double *ba, const double *noise_meas_diag6, const double *noise_walk_diag6,
                         double scale_gravity, double *imu_pre_out);
/* IMU_PRE::give_evaluate (PI:137-212) / give_evaluate_g (PI:214-294): returns r^T cov^-1 r in *resid;
 * jtj ((30|33)^2) and gg (30|33) are written when jac_enable != 0. */
int vba_imu_give_evaluate(const double *imu_pre, const double *state1, const double *state2, int with_gravity,
                          int jac_enable, double *jtj, double *gg, double *resid);

/* ------------------------------------------------------------------------------------------------
 * Map level — drop-in for the voxel hash map + octree of local mapping.                           */

/* cut_voxel (VM:1896-1949) / cut_voxel_multi (VM:1964-2096) for one scan: pnt_body [n][3] body-frame
 * points (pointVar::pnt), var [n][9] per-point covariance as produced by pvec_update (VH:242-265) or NULL,
 * pose [12] = the scan's pose used for pw = R p + t (the pwld argument), win_count = frame index in the window.
 * multi != 0 applies cut_voxel_multi's "#touched voxels < thread_num -> scan dropped" rule (VM:2044-2045).
 * pnt_body / var may point to HOST or DEVICE (HBM) memory; device buffers are consumed in place. */
int vba_map_cut_voxel(vba_ctx *ctx, int win_count, int n, const double *pnt_body, const double *var,
                      const double *pose, int multi);
/* pvec_update (VH:242-265) fused with cut_voxel[_multi]: var_body [n][9] is the BODY-frame covariance of the scan's
 * points (from var_init); the world-frame covariance var = R var R^T + phat rot_var phat^T + tsl_var is formed on the
 * device from the scan state's covariance cov [225] (x_curr.cov: rot block (0,0), translation block (3,3)). */
int vba_map_pvec_update_cut_voxel(vba_ctx *ctx, int win_count, int n, const double *pnt_body, const double *var_body,
                                  const double *pose, const double *cov, int multi);
/* var_init (VH:210-234) = calcBodyVar (VH:180-200) + extrinsic ext_pose [12]: pnt_in [n][3] -> pnt_out [n][3],
 * var_out [n][9] (host buffers; pnt_out may alias pnt_in). */
int vba_scan_var_init(vba_ctx *ctx, int n, const double *pnt_in, const double *ext_pose, double dept_err, double beam_err,
                      double *pnt_out, double *var_out);
/* down_sampling_voxel (tools.hpp:201-238): voxel-grid centroid filter.  pnt [n][3] (PCL float coordinates carried in
 * doubles) -> pnt_out [<=n][3] centroids rounded to float, count_out = points per voxel (the `curvature` field after the
 * call, TL:221/230), first_out = index of the voxel's first input point (whose intensity/normal fields the reference
 * keeps); *n_out = number of voxels.  Output order = first occurrence (the reference's is unordered_map order).
 * voxel_size < 0.001 returns the input unchanged (TL:203), counts 0.  Buffers may be HOST or DEVICE memory. */
int vba_scan_down_sampling_voxel(vba_ctx *ctx, int n, const double *pnt, double voxel_size, double *pnt_out, int *count_out,
                                 int *first_out, int *n_out);
/* down_sampling_pvec (voxel_map.hpp:39-83): the pointVar form used when a keyframe cloud is made (VS:2385): double
 * coordinates pnt [n][3] and covariances var [n][9] -> per voxel the mean point and the mean covariance DIAGONAL
 * (stored by the reference in normal_x/y/z), both narrowed to float like the PCL points they become. */
int vba_scan_down_sampling_pvec(vba_ctx *ctx, int n, const double *pnt, const double *var, double voxel_size,
                                double *pnt_out, double *vardiag_out, int *count_out, int *n_out);
/* down_sampling_close (tools.hpp:240-298): per voxel the index of the input point closest to the voxel's centroid
 * (first such point; only squared distances < 100 compete, else the voxel's first point — TL:281-295). */
int vba_scan_down_sampling_close(vba_ctx *ctx, int n, const double *pnt, double voxel_size, int *index_out, int *n_out);
/* Undistortion inner loop of IMUEKF::motion_blur (ekf_imu.hpp:137-163).  pnt [n][3] in/out, curv [n] = per-point time
 * offset (PointType::curvature), ascending as pcl_handler leaves them (VH:92-95); imu_poses [m][22] = the imu_poses
 * vector (EK:87): t, R[9], p[3], v[3], angvel_avr[3], acc_imu[3]; end_pose [12] = xc.R, xc.p after propagation
 * (EK:121-123); ext_pose [12] = Lid_rot_to_IMU, Lid_offset_to_IMU. */
int vba_scan_undistort(vba_ctx *ctx, int n, double *pnt, const double *curv, int m, const double *imu_poses,
                       const double *end_pose, const double *ext_pose);
/* cut_voxel(feat_map, PVec&, wdsize, jour) for fixed (already-world) points (VM:2108-2152). */
int vba_map_cut_voxel_fix(vba_ctx *ctx, int n, const double *pnt_world, double jour);
/* multi_recut (VS:1682-1737) when multi != 0, or the loop "recut + tras_opt over surf_map" of motion_init
 * (VS:699-703) when multi == 0: OctoTree::recut (VM:1396-1456) on every root, then OctoTree::tras_opt
 * (VM:1605-1638) fills the context's factor store (voxhess.clear() + win_size, VS:1918-1919, is implied). */
int vba_map_recut(vba_ctx *ctx, int win_count, const double *poses, int multi);
/* multi_margi (VS:1590-1679): OctoTree::margi (VM:1465-1598, mgsize = 1) on every root of the sliding map
 * using the factor store's refined eig/pcr_add, then drops roots with !isexist from the sliding map.
 * jour is stamped on every sliding-map root (VS:1628). */
int vba_map_margi(vba_ctx *ctx, int win_count, const double *poses, double jour);
/* Ring-map rotation mp[i] = (mp[i] + mgsize) mod W (VS:2014-2019). */
int vba_map_slide(vba_ctx *ctx, int mgsize);
/* "Release the features not used for a long time" (VS:1800-1823): roots with int(jour - root.jour) >= dist (700 in
 * the reference) leave surf_map with their subtrees. */
int vba_map_prune(vba_ctx *ctx, double jour, int dist);
/* Destroys the map (system_reset / motion_init teardown, VS:650-661). */
int vba_map_reset(vba_ctx *ctx);
int vba_map_num_roots(vba_ctx *ctx);       /* surf_map.size() */
int vba_map_num_slide_roots(vba_ctx *ctx); /* surf_map_slide.size() */
/* Storage statistics of the device map (the reference new/deletes OctoTree nodes, VS:1787-1823; here pruned roots hand their
 * node storage and hash slots back for reuse): out8 = [node high-water mark, free root nodes, free 8-node child blocks, root
 * table capacity, root table slots in use (live roots + tombstones), live roots, sliding-map roots, fixed points]. */
int vba_map_stats(vba_ctx *ctx, long long *out8);
/* Leaf dump for inspection / parity tests: 39 doubles per leaf
 * [kx,ky,kz, layer, path, N_add, N_fix, is_plane, isexist, opt_state, eig_value(3), eig_vector(9), pcr_add(10),
 *  plane.center(3), plane.normal(3), plane.radius].  out == NULL returns the leaf count. */
int vba_map_dump_leaves(vba_ctx *ctx, double *out, int max_leaves);
/* The two covariance outputs of the map that the leaf dump does not carry: plane.plane_var (6x6; plane_update VM:1344-1388,
 * consumed by OctoTree::match VM:1667-1672) and cov_add (9x9 symmetric; Bf_var VM:106-121 summed by push VM:1138-1140).
 * 86 doubles per leaf: [kx,ky,kz, layer, path, plane_var(36 row-major), cov_add upper triangle (45, row by row)];
 * same leaf set as vba_map_dump_leaves (order not defined).  out == NULL returns the leaf count. */
int vba_map_dump_plane_var(vba_ctx *ctx, double *out, int max_leaves);

/* ------------------------------------------------------------------------------------------------
 * Odometry scan-to-map ("next #1").
 * bool VOXEL_SLAM::lio_state_estimation(PVecPtr pptr) (VS:962-1098): iterated EKF update of x_curr against the voxel map
 * with match() (VM:2167-2205) / OctoTree::match (VM:1649-1721).  pnt_body [n][3] and var_body [n][9] are the scan's
 * body-frame points and covariances (pointVar as produced by var_init, VH:210-234); state [25] and cov [225] = x_curr
 * (IMUST incl. its 15x15 covariance) in/out; *ok receives the bool result (false = degenerate, VS:1090-1097).
 * The per-point loop runs on the device, the 15x15 EKF algebra on the host. */
int vba_odom_lio_state_estimation(vba_ctx *ctx, int n, const double *pnt_body, const double *var_body, double *state,
                                  double *cov, int *ok);

/* void VOXEL_SLAM::lio_state_estimation_kdtree(PVecPtr pptr) (VS:1102-1252), the odometry used while the system initialises:
 * scan points against a point-cloud map (pl_tree, kept by the context) through an exact 5-nearest-neighbour plane fit.
 * While the map holds fewer than 100 points the scan only seeds it (VS:1105-1118, *iterations = 0); otherwise state / cov
 * are updated in place, the scan is appended in the refined pose and the map re-sampled on a 0.5 m grid (VS:1238-1250). */
int vba_odom_lio_state_estimation_kdtree(vba_ctx *ctx, int n, const double *pnt_body, double *state, double *cov,
                                         int *iterations);
int vba_odom_kdtree_reset(vba_ctx *ctx);   /* pl_tree->clear() */
int vba_odom_kdtree_size(vba_ctx *ctx);    /* pl_tree->size() */
int vba_odom_kdtree_points(vba_ctx *ctx, double *xyz_out /* [size][3] */);

/* ------------------------------------------------------------------------------------------------
 * Hierarchical global BA ("next #3"), one keyframe window per call.  vba_hba_add_edge accepts any
 * wdsize >= 2: a window of the context's win_size (the bottom layers use 10, VS:3033) runs on the templated device
 * kernels; any other size — the top-level BA over all submaps, VS:3103-3113 — takes the sparse path (hashed per-keyframe
 * clusters, atomics Hessian, LM loop and dense LDL^T on the host).  vba_gba_build requires wdsize == win_size.
 * Keyframe clouds are passed ragged: pnt_local [offsets[wdsize]][3] holds keyframe i's points (its own frame, PCL float
 * values in doubles) in rows offsets[i]..offsets[i+1]; HOST or DEVICE memory.  gba_eigen_value_array is ALREADY INVERTED
 * (VS:3022-3024).
 *
 * vba_gba_build = OctreeGBA::cut_voxel for every keyframe (LR:439-479) + OctreeGBA_multi_recut (LR:483-537): fills the
 * context's factor store (what `LidarFactor voxhess(wdsize)` holds at VS:2889-2890).
 *
 * vba_hba_add_edge = VOXEL_SLAM::HBA_add_edge (VS:2822-3015) on an already filtered keyframe set: up to max_iter
 * rounds of {octree rebuild, Lidar_BA_Optimizer::damping_iter(xs, voxhess, &hess, resis, 4)} with the convergence
 * ladder of VS:2871-2915 (the last round uses the context's voxel_size / plane_eigen_value_thre / min_eigen_value),
 * poses [wdsize][12] in/out, then one edge per keyframe pair whose six diagonal Hessian entries are all >= 1e-6
 * (VS:2926-2951): edges_out rows = i, j, rot[9] = R_i^T R_j, tra[3] = R_i^T (p_j - p_i), v6[6] = 1/|H| — the arguments
 * of PGO_Edges::push (LR:247).  cloud_out (optional, capacity offsets[wdsize] rows) receives the submap cloud of
 * VS:2954-2989 (all points in keyframe 0's frame, down_sampling_voxel(voxel_size / 8)), cloud_count its per-voxel
 * counts.  resis_log (optional, [max_iter][2]) receives resis[0], resis[1] of every round.
 * Returns VBA_ERR_TOO_FEW_VOXELS where the reference prints "Too Less Voxel" and exits. */
int vba_gba_build(vba_ctx *ctx, int wdsize, const int *offsets, const double *pnt_local, const double *poses,
                  double gba_voxel_size, double gba_min_eigen_value, const double *gba_eigen_value_array);
int vba_hba_add_edge(vba_ctx *ctx, int wdsize, const int *offsets, const double *pnt_local, double *poses,
                     double gba_voxel_size, double gba_min_eigen_value, const double *gba_eigen_value_array, int max_iter,
                     int thread_num, double *edges_out, int *n_edges, double *cloud_out, int *cloud_count, int *n_cloud,
                     double *resis_log, int *n_log);

/* The optimisation work of thd_globalmapping (VS:3018-3141) over one map: bottom-layer windows of wdsize keyframes every
 * mgsize keyframes (10 / 5 in the reference, VS:3033-3034) on the poses x0, each yielding edges (edges1, the reference's
 * gba_edges1) and one submap (first keyframe's x0, the window's down-sampled cloud in that frame, VS:3084-3089); then the
 * top-level HBA_add_edge over all submaps with the keyframes' CURRENT poses poses_now (VS:3096-3110) -> edges2.  Rows as in
 * vba_hba_add_edge with global keyframe indices; cap1 / cap2 = row capacities of the outputs.  Queues, map switching and
 * the GTSAM graph stay with the caller. */
int vba_hba_global(vba_ctx *ctx, int n_kf, const int *offsets, const double *pnt_local, const double *poses_x0,
                   const double *poses_now, double gba_voxel_size, double gba_min_eigen_value,
                   const double *gba_eigen_value_array, int total_max_iter, int wdsize, int mgsize, double *edges1_out, int cap1,
                   int *n_edges1, double *edges2_out, int cap2, int *n_edges2);

/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: voxels are sharded by root-voxel hash bucket; each rank evaluates its
 * shard and the packed [H | g | r] buffer is summed across ranks (the thread-sum of VM:571-581).
 * The reduction is RCCL inside the library (vba_rccl_init) or, for rehearsals without RCCL, a hook
 * supplied by the host program; both are stream-ordered on the context's stream.                  */
typedef int (*vba_allreduce_fn)(void *user, void *buf_dev, size_t n_doubles, void *stream);
int vba_set_allreduce(vba_ctx *ctx, vba_allreduce_fn fn, void *user);
/* RCCL inside the library: the context owns (vba_rccl_init) or adopts (vba_set_rccl_comm, an ncclComm_t) a communicator and issues
 * ncclAllReduce(ncclDouble, ncclSum) / ncclAllGather on its own stream — no host code in the LM loop.  Rank 0 calls
 * vba_rccl_get_unique_id (128 bytes, ncclUniqueId), the host program hands the bytes to every rank (any transport), every rank
 * calls vba_rccl_init, which also applies vba_set_shard(rank, n_ranks).  The hook above stays for CPU-side rehearsals (gloo). */
int vba_rccl_get_unique_id(void *out128);
int vba_rccl_init(vba_ctx *ctx, const void *unique_id128, int rank, int n_ranks);
int vba_set_rccl_comm(vba_ctx *ctx, void *nccl_comm);
/* Which rank owns root voxel (kx,ky,kz) out of n_ranks (pure function, usable without a device). */
int vba_shard_owner(int64_t kx, int64_t ky, int64_t kz, int n_ranks);
int vba_set_shard(vba_ctx *ctx, int rank, int n_ranks);

/* ------------------------------------------------------------------------------------------------
 * Measurement hooks (bench.py): average device time in microseconds of the named kernel family since
 * the last reset, from hipEvents recorded on the context's stream around each launch.             */
int vba_timing_enable(vba_ctx *ctx, int on);
/* Measurement aid for the rocprofv3 PMC passes: one launch that reads exactly (n_bytes rounded down to 32 KiB) bytes with the
 * factor store's access shape; its FETCH_SIZE calibrates the read-side counter correction (tools/prof_summary.py). */
int vba_timing_calibration_read(vba_ctx *ctx, size_t n_bytes);
/* Restrict the event bracketing to one kernel family (NULL / "" = all): two hipEventRecord calls per launch cost host
 * time, so the headline timed region brackets only the kernel whose roofline is reported. */
int vba_timing_select(vba_ctx *ctx, const char *name);
/* Bracket only every n-th launch of the selected family (n <= 1: every launch).  An event pair costs ~5 us of stream time on
 * this runtime, more than the kernel it brackets: the headline timed region samples instead of bracketing every launch. */
int vba_timing_sample_every(vba_ctx *ctx, int n);
/* Records an event pair around no work under the name "null": the overhead that every bracketed launch carries. */
int vba_timing_null_span(vba_ctx *ctx);
int vba_timing_reset(vba_ctx *ctx);
/* name in {"residual","hessian","reduce","solve","insert","recut","margi"}; returns launches in *count. */
int vba_timing_get(vba_ctx *ctx, const char *name, double *total_us, int *count);

/* LM building blocks on device state (used by bench.py to time exactly K LM iterations, and by the
 * damping_iter entry points themselves): begin loads the poses, iterate runs one trip through the
 * loop body VM:441-494 / VM:643-710, end copies the refined poses back (all-NULL outputs: no synchronisation). */
int vba_lm_begin(vba_ctx *ctx, const double *poses, int thd_num);
int vba_lm_refresh_eigen(vba_ctx *ctx); /* device-side residual pass at the begin poses (re-creates eig/pcr_add state) */
/* Measurement aid: enqueues exactly one launch of the Hessian pass kernel (all voxels, at the begin poses, no reduction) on the
 * context's stream.  bench.py replays a batch of them from a HIP graph between one event pair. */
int vba_timing_launch_hessian(vba_ctx *ctx);
int vba_lm_iterate(vba_ctx *ctx, int *accepted, int *stop); /* NULL, NULL: enqueue only (no host synchronisation) */
int vba_lm_end(vba_ctx *ctx, double *poses, double *hess, double *resis2);

/* ------------------------------------------------------------------------------------------------
 * Session-store formats either side of the path.  Host only, no context.
 *   FileReaderWriter::save_pcd (VS:166-179): <session>/<count>.pcd, pcl::io::savePCDFileBinary of PointXYZI (x y z from the
 *     scan's body-frame points, intensity 0); read back with pcl::io::loadPCDFile (VS:337-340).  vba_io_load_pcd reads
 *     DATA binary and DATA ascii files with x y z [intensity] among their fields; *n_out is the point count also when
 *     VBA_ERR_CAPACITY is returned (call with cap 0 to size the buffers).
 *   FileReaderWriter::save_pose (VS:181-204) / read_lidarstate (VH:268-307): alidarState.txt, one line per scan:
 *     t px py pz qx qy qz qw vx vy vz bgx bgy bgz bax bay baz gx gy gz v6[0..5], fixed notation, 6 decimals for t, 7 for the
 *     rest; nothing is written for fewer than 100 scans (VS:183-184); the reader accepts 8-, 20- and 26-column lines.
 *   states: n x 25 doubles (t, R row-major, p, v, bg, ba, g — the layout of vba_odom_*), v6: n x 6. */
int vba_io_save_pcd(const char *path, int n, const double *xyz);
int vba_io_load_pcd(const char *path, int cap, double *xyz, double *intensity /* may be NULL */, int *n_out);
int vba_io_save_pose(const char *path, int n, const double *states, const double *v6);
int vba_io_read_lidarstate(const char *path, int cap, double *states, double *v6 /* may be NULL */, int *n_out);

#ifdef __cplusplus
}
#endif
#endif /* VOXELBA_H */
