/*
 * sslam.h — C-ABI of the MI355X-native semantic_slam hot path (libsslam_hip.so).
 *
 * Plain pointers and sizes only; no torch / g2o / PCL / ROS types.  Every entry point names the
 * reference interface it replaces (paths relative to the reference repository root).
 * The C++ shims that give these the reference's own method names live in
 *   include/ps_graph_slam_amd/graph_slam.hpp              (ps_graph_slam::GraphSLAM)
 *   include/planar_segmentation_amd/point_cloud_segmentation.hpp (point_cloud_segmentation)
 * and INTEGRATION.md shows the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - return value: >= 0 success (ids / counts where stated), < 0 error (sslam_last_error()).
 *   - poses: 7 doubles  t(x,y,z) q(x,y,z,w)   (same order as g2o's VERTEX_SE3:QUAT row)
 *   - information matrices: dense row-major d x d doubles (d = 6 or 3)
 *   - handles are not re-entrant; different handles may be used from different threads.
 *   - all compute runs on the GPU; there is NO CPU fallback: without a HIP device
 *     sslam_graph_optimize()/sslam_seg_segment() fail with SSLAM_ERR_NO_DEVICE.
 */
#ifndef SSLAM_H
#define SSLAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSLAM_OK 0
#define SSLAM_ERR_INVALID (-1)
#define SSLAM_ERR_NO_DEVICE (-2)
#define SSLAM_ERR_HIP (-3)
#define SSLAM_ERR_NUMERIC (-4)      /* non-finite chi2 / factorisation breakdown */
#define SSLAM_ERR_TOO_FEW_EDGES (-5) /* reference: graph_slam.cpp:184-186 returns false */
#define SSLAM_ERR_UNSUPPORTED (-6)
#define SSLAM_ERR_IO (-7)

const char* sslam_last_error(void);
/* number of visible HIP devices (0 on a CPU-only box; never fails) */
int sslam_device_count(void);
/* Page-locked host memory (hipHostMalloc / hipHostFree) for callers that do not link HIP themselves: clouds handed to
 * sslam_seg_submit_batch from such a buffer are copied asynchronously.  NULL (and sslam_last_error) without a device. */
void* sslam_pinned_alloc(size_t bytes);
void sslam_pinned_free(void* p);

/* ============================================================================================
 * Backend: ps_graph_slam::GraphSLAM  (reference include/ps_graph_slam/graph_slam.hpp:37-144)
 * ==========================================================================================*/
typedef struct sslam_graph sslam_graph;

typedef struct sslam_opt_stats {
  int iterations;        /* LM iterations performed (g2o SparseOptimizer::optimize return value) */
  int trials;            /* linear solves (accepted + rejected LM trials) */
  int status;            /* 0 = iteration cap reached, 1 = LM terminated (10 failed trials or rho == 0),
                            <0 = SSLAM_ERR_* */
  int host_plan_us;      /* sslam_graph_optimize: microseconds of host work before the first launch of this call that a structure change
                            cost -- batch tables + symbolic factorisation (0 when the structure of the last call was reused); 0 in batches */
  double chi2_before;    /* graph->chi2() before (graph_slam.cpp:202) */
  double chi2_after;     /* graph->chi2() after  (graph_slam.cpp:211) */
  double lambda;         /* final LM damping */
  double seconds;        /* wall time of the call (graph_slam.cpp:204-215 "time:") */
  int64_t solver_iterations; /* PCG iterations summed over trials (0 for the direct solver) */
} sslam_opt_stats;

/* GraphSLAM::GraphSLAM (graph_slam.cpp:40-97): LM over block-sparse normal equations, sensor
 * offset parameter id 0 = identity.  device = HIP device ordinal. */
sslam_graph* sslam_graph_create(int device);
/* GraphSLAM::~GraphSLAM (graph_slam.cpp:102) */
void sslam_graph_destroy(sslam_graph* g);

/* add_se3_node (graph_slam.cpp:104-115).  The reference fixes the first vertex of the graph;
 * pass fixed = -1 to get exactly that rule, 0/1 to force.  Returns the vertex id (= number of
 * vertices before the call, graph_slam.cpp:106). */
int sslam_graph_add_vertex_se3(sslam_graph* g, const double t_q[7], int fixed);
/* add_point_xyz_node (graph_slam.cpp:127-134) */
int sslam_graph_add_vertex_point(sslam_graph* g, const double p[3]);
/* add_plane_node (commented out in the reference, graph_slam.cpp:117-125; g2o::VertexPlane) */
int sslam_graph_add_vertex_plane(sslam_graph* g, const double n_d[4]);

/* add_se3_edge (graph_slam.cpp:136-148): g2o::EdgeSE3 between SE3 vertices i and j. Returns edge id. */
int sslam_graph_add_edge_se3(sslam_graph* g, int i, int j, const double z_tq[7], const double info[36]);
/* add_se3_point_xyz_edge (graph_slam.cpp:150-166): g2o::EdgeSE3PointXYZ, offset parameter 0.
 * The reference passes an uninitialised robust-kernel pointer (quirk B1): no kernel is applied unless the option
 * "robust_kernel_dcs" asks for the one it names (g2o::RobustKernelDCS). */
int sslam_graph_add_edge_se3_point(sslam_graph* g, int i, int l, const double z[3], const double info[9]);
/* add_se3_plane_edge (commented out, graph_slam.hpp:73-75) -> g2o::EdgeSE3Plane
 * (reference include/g2o/edge_se3_plane.hpp:8-48; numeric Jacobian: g2o's central differences, delta 1e-9 -- evaluated on the device once per
 * edge and linearisation by k_plane_jacobians, a thread per evaluation (round 6); the plane rotation is formed from the normal's components
 * instead of through its two angles: the same matrix to an ulp). */
int sslam_graph_add_edge_se3_plane(sslam_graph* g, int i, int l, const double z[4], const double info[9]);

/* add_point_xyz_point_xyz_edge (graph_slam.cpp:168-180): g2o::EdgePointXYZ between two VertexPointXYZ, e = (p2 - p1) - z.  Declared and
 * defined by the reference, never called there.  With such edges the landmark block of H is no longer block diagonal: solver 2
 * (Schur complement on the landmarks) refuses the graph, the Cholesky solvers and solver 0 take it as it is. */
int sslam_graph_add_edge_point_point(sslam_graph* g, int l1, int l2, const double z[3], const double info[9]);

int sslam_graph_num_vertices(const sslam_graph* g);
int sslam_graph_num_edges(const sslam_graph* g);

/* vertex->estimate() / setEstimate(); out has 7 (SE3), 3 (point) or 4 (plane) doubles */
int sslam_graph_get_vertex(const sslam_graph* g, int id, double* out);
int sslam_graph_set_vertex(sslam_graph* g, int id, const double* in);
/* vertex->hessianIndex() after initializeOptimization (used at semantic_graph_slam.cpp:188-190):
 * scalar offset in g2o's ordering (non-fixed vertices with edges, by id), or -1. */
int sslam_graph_hessian_index(sslam_graph* g, int id);

/* Options (doubles): "solver" 1 = sparse block Cholesky (default; what "lm_var" + csparse selects in the reference),
 * 0 = block-Jacobi PCG on the full system, 2 = Schur complement on the landmark block + PCG on the reduced pose system (matrix-free;
 * BASELINE.json north_star).  SOLVER 2 IS S-SCALE ONLY: its preconditioner is block-Jacobi, and on a long pose chain with the reference's
 * odometry information (1 / 0.00001 on the rotation block against 2.5 on a landmark, config/bucket_detector.yaml:22-27) the reduced
 * system needs ~1,050 CG iterations per damping trial at 5000 poses (157 ms, against 1.5 ms for the direct solver) -- it converges to the
 * same optimum (tests: S config, and since round 6 three LM iterations of the L config against the oracle) but is not a solver for the
 * L configuration; use 1.  Round 6 measured what the missing preconditioner would buy (tools/pcg_preconditioner_experiment.py, the L graph's
 * reduced system at g2o's first lambda, relative residual 1e-10): block-Jacobi 801 CG iterations, the EXACT block-tridiagonal factor of
 * the odometry chain 111, wider bands (2-16 blocks) 111-112 -- what is left is the coupling of poses on different laps through shared
 * landmarks.  A tridiagonal solve is two sequential sweeps over 5000 blocks per CG iteration: ~110 x 1.5 ms per trial on one wave per
 * graph, slower than the 157 ms it would replace, and still above the 100 iterations asked for -> not built; decision closed.  (3, the window-plan Cholesky of round 3 -- register-resident sliding
 * fronts, VALU and FP64-MFMA updates; correct, leaner in traffic, 2-2.7x slower -- was removed from the library in round 5; DESIGN.md
 * section 5 keeps its measurements.)  "pcg_tol" relative residual; "pcg_max_iters";
 * "robust_kernel_dcs" = phi > 0: g2o::RobustKernelDCS(delta = phi) on every landmark edge (EdgeSE3PointXYZ / EdgeSE3Plane), as
 * graph_slam.cpp:155,161 intends (opt-in, phi = 1 is g2o's default delta); 0 = no kernel (default);
 * "fused_small_graph" 1 (default) / 0: a batch of fewer than eight graphs whose elimination tree is narrower than the chip runs the
 * factorisation, both triangular solves and the begin / end halves of a damping trial in ONE dependency-driven launch (k_chol_flow; the
 * reference's per-tick call pattern, semantic_graph_slam.cpp:58-102); 0: the stand-alone kernels -- same results, bitwise.  (The
 * one-workgroup-per-graph kernel k_lm_trial_small of round 4 -- measured slower, DESIGN.md section 5 -- left the library in round 5.)
 * "speculative_trials" 0 / 1 (default since round 5) / 2: a single small graph runs the damping trials of an LM iteration side by side -- g2o's retry
 * lambdas are known when the iteration starts -- in the lanes of ONE launch (k_chol_spec_round) that also replays the accept / reject
 * sequence over their results: bitwise the sequential result, trial counts included.  1: the lanes join once a trial of the iteration
 * has been rejected, and only while every lane gets a workgroup per piece of the tree (about 200 keyframes); 2: every round with all ten
 * lanes.  Mode 1 measured 5.5 vs 6.0 ms per tick at 110 keyframes and 7.4 vs 7.6 at 436 (re-measured
 * in round 5) and is never slower with the lanes-fit rule: the default.  SSLAM_LM_SPEC=0/1/2 in the environment overrides the option for
 * every graph of the process. */
int sslam_graph_set_option(sslam_graph* g, const char* key, double value);

/* GraphSLAM::optimize (graph_slam.cpp:182-219) with the iteration cap as a parameter (the
 * reference hard-codes 1024, graph_slam.cpp:205).  Fewer than 10 edges: returns
 * SSLAM_ERR_TOO_FEW_EDGES and leaves the graph untouched (graph_slam.cpp:184-186). */
int sslam_graph_optimize(sslam_graph* g, int max_iters, sslam_opt_stats* out);

/* graph->chi2() at the current estimates */
int sslam_graph_chi2(sslam_graph* g, double* chi2);

/* computeLandmarkMarginals (graph_slam.cpp:221-234): diagonal blocks of H^-1 at the current
 * linearisation for the listed vertex ids (the caller asks for (idx,idx) pairs only,
 * semantic_graph_slam.cpp:186-191); out = packed row-major d x d blocks. */
int sslam_graph_marginals(sslam_graph* g, const int* ids, int n, double* out_blocks);
/* the same with the reference's own argument: n (row, col) pairs of vertex hessianIndex() values, as built at
 * semantic_graph_slam.cpp:186-191 and handed to g2o::SparseOptimizer::computeMarginals (graph_slam.cpp:225); row_col = 2n ints.
 * Off-diagonal pairs are allowed; out = packed row-major d(row) x d(col) blocks of H^-1. */
int sslam_graph_marginals_by_hessian_index(sslam_graph* g, const int* row_col, int n, double* out_blocks);

/* GraphSLAM::save (graph_slam.cpp:236-239): g2o text format */
int sslam_graph_save_g2o(const sslam_graph* g, const char* path);
/* inverse of save (the reference has no load path) */
int sslam_graph_load_g2o(sslam_graph* g, const char* path);

/* ---- parity / measurement hooks (no reference counterpart) ---------------------------------
 * Linearise at the current estimates (BlockSolver::buildSystem) and copy the normal equations to
 * the host as dense blocks in g2o hessian-index order.  Two-call protocol: pass NULL arrays to
 * obtain counts.  H is returned as COO over scalar entries of the upper triangle. */
int sslam_graph_linearize(sslam_graph* g, int* dim, int64_t* nnz_upper, int32_t* rows, int32_t* cols,
                          double* vals, double* b);
/* solve (H + lambda I) x = b for the current linearisation; x in g2o hessian-index order */
int sslam_graph_solve(sslam_graph* g, double lambda, double* x, int64_t* solver_iterations);
/* x <- x [+] dx for all active vertices (VertexSE3/PointXYZ/Plane::oplus) */
int sslam_graph_oplus(sslam_graph* g, const double* dx);

/* ---- plan introspection (host only: works without a HIP device) ----------------------------
 * The symbolic sparse-Cholesky plan of a batch (ordering, elimination-tree pieces, work items) as flat arrays of int32
 * records, for the CPU-side plan tests (tests/test_chol_plan_cpu.py).  Arrays: "col","blk","upd","item","mb","ilv","piece",
 * "lvl_ptr","lvl_cols","plv_ptr","plv_pieces","tail_ptr","tail_pieces","plv_lds_f","plv_lds_b","scalars".
 * sslam_debug_plan_array returns the byte size of the array (copies it when out != NULL and cap_bytes suffices). */
void* sslam_debug_plan_create(sslam_graph* const* graphs, int n);
void sslam_debug_plan_destroy(void* plan);
int64_t sslam_debug_plan_array(void* plan, const char* name, void* out, int64_t cap_bytes);
/* ---- batched, device-resident form (MI355X extension) ---------------------------------------
 * B independent graphs laid out contiguously in HBM and optimised together: every kernel runs
 * over the union, LM control (rho, lambda, accept/reject) is per graph on the device. */
/* Lifetime and structure: the batch borrows the graphs -- they must outlive it (destroy the batch first) -- and is built for their
 * structure at creation: after a vertex or an edge is added to a member, every sslam_batch_* entry point returns SSLAM_ERR_INVALID
 * until a new batch is created.  All members must carry the same options (sslam_graph_set_option). */
typedef struct sslam_batch sslam_batch;
sslam_batch* sslam_batch_create(sslam_graph* const* graphs, int n);
/* Stream group: the n graphs split into n_streams contiguous parts, each a batch of its own on its own HIP stream; sslam_batch_optimize
 * drives every part from its own host thread.  Results are those of sslam_batch_create (every graph runs its own LM; the parts only
 * change what overlaps on the chip: one part's tree tops and LM endgame run under another part's leaves).  Measured on 512 distinct L
 * graphs: 4 streams = +16 % iterations/s.  upload / download / optimize / info / profiling / time_* work on a group; the edge-sharded
 * mode and sslam_batch_linearize_hb return SSLAM_ERR_UNSUPPORTED.  sslam_batch_create reads the number of streams from the environment
 * (SSLAM_BATCH_STREAMS, default 1). */
sslam_batch* sslam_batch_create_streams(sslam_graph* const* graphs, int n, int n_streams);
void sslam_batch_destroy(sslam_batch* b);
/* re-upload the host graphs' current estimates (resets the device state) */
int sslam_batch_upload(sslam_batch* b);
/* copy optimised estimates back into the host graphs */
int sslam_batch_download(sslam_batch* b);
int sslam_batch_optimize(sslam_batch* b, int max_iters, sslam_opt_stats* out /* [n] */);
/* Edge-sharded mode (SURVEY 8e mode E, BASELINE.json configs[4]): every graph's edge list is split contiguously over `world` ranks
 * (one process per GPU, each holding the whole batch); a rank builds the partial normal equations of its edges, ONE RCCL all-reduce
 * (ncclDouble, sum, over xGMI) of the contiguous [H || b] device buffer gives every rank the full system, and the solve / update /
 * LM control then run replicated and bit-identical.  Rank 0 obtains an id with sslam_comm_unique_id and hands its 128 bytes to the
 * other ranks by whatever the host program uses (torch.distributed, MPI, a file); every rank then calls sslam_batch_comm_init.
 * world == 1 switches the mode off, unless the environment sets SSLAM_FORCE_COMM=1: then a single-rank communicator is created and the
 * whole path (masked kernels, ncclCommInitRank, ncclAllReduce on the batch stream) runs on one GPU (sslam_batch_info "allreduce_calls"
 * counts the collectives issued).  The all-reduce is out of place, partial buffer -> [H || b]: graphs that do not re-linearise in a
 * step keep their old partial system, so the sum stays the system they already had.  sslam_batch_set_edge_shard installs the shard WITHOUT a communicator (the partial systems stay
 * unsummed): with sslam_batch_linearize_hb (-> the [H || b] buffer; NULL returns its length in doubles) this is the parity hook that
 * shows sum over ranks of partial systems == the full system on a single device. */
int sslam_comm_unique_id(char id_out[128]);
int sslam_batch_comm_init(sslam_batch* b, const char id_in[128], int rank, int world);
int sslam_batch_set_edge_shard(sslam_batch* b, int rank, int world);
int64_t sslam_batch_linearize_hb(sslam_batch* b, double* h_and_b, int64_t capacity);
/* run only the Jacobian build (linearise + assemble) `repeats` times; returns mean kernel
 * milliseconds measured with hipEvents on the batch's stream */
int sslam_batch_time_linearize(sslam_batch* b, int repeats, double* ms_per_build);
/* one numeric factorisation (+ fused forward substitution) and one backward substitution of EVERY graph of the batch, `repeats` times:
 * mean kernel milliseconds (hipEvents on the batch's stream); direct solvers only */
int sslam_batch_time_solver(sslam_batch* b, int repeats, double* factor_ms, double* solve_ms);
/* algorithmic bytes of one Jacobian build over the whole batch (SURVEY §8d formula) */
int64_t sslam_batch_linearize_bytes(const sslam_batch* b);
/* structural facts of a batch (doubles): "factor_lnz" (doubles in the Cholesky factor), "factor_levels",
 * "h_doubles" (doubles in H), "dim" (scalar unknowns), "factor_bytes" = algorithmic HBM bytes of one numeric
 * factorisation + fused forward solve: read H and b once, write L and y once */
int sslam_batch_info(sslam_batch* b, const char* key, double* value);
/* per-kernel accumulated hipEvent time since the last reset (profiling must be enabled with
 * sslam_batch_set_profiling); names: "linearize","chi2","spmv","pcg_update","precond","oplus","factor","solve" */
int sslam_batch_set_profiling(sslam_batch* b, int enable);
int sslam_batch_kernel_time(sslam_batch* b, const char* name, double* total_ms, int64_t* launches);

/* ============================================================================================
 * Frontend: point_cloud_segmentation::segmentallPointCloudData
 *           (reference include/planar_segmentation/point_cloud_segmentation.h:105-181)
 * ==========================================================================================*/
typedef struct sslam_seg sslam_seg;

typedef struct sslam_seg_params {
  double num_point_seg;     /* ~num_point_seg   default 500  (plane_segmentation.cpp:7)  */
  double norm_point_thres;  /* ~norm_point_thres default 5000 (plane_segmentation.cpp:8)  */
  double planar_area;       /* ~planar_area     default 0.1  (plane_segmentation.cpp:9)  */
  float max_depth_change_factor; /* 0.03  (plane_segmentation.cpp:99)  */
  float normal_smoothing_size;   /* 20    (plane_segmentation.cpp:100) */
  float angular_threshold;       /* 0.017453*2 rad (plane_segmentation.cpp:140) */
  float distance_threshold;      /* 0.02 m (plane_segmentation.cpp:141) */
  float maximum_curvature;       /* PCL default 0.001 */
  int min_contour_points;        /* > 100 (plane_segmentation.cpp:169) */
  int image_width, image_height; /* 640 x 480 crop bounds (plane_segmentation.cpp:34-35) */
  int reference_quirks;          /* 1: reproduce tools.h:80-81 typo (quirk B2) */
  int device;
} sslam_seg_params;

/* semantic_SLAM::ObjectInfo (msg/ObjectInfo.msg:1-6); class_id indexes SSLAM_CLASS_* below */
typedef struct sslam_box {
  int32_t tl_x, tl_y, width, height;
  int32_t class_id;
  float prob;
} sslam_box;

/* the class whitelist of point_cloud_segmentation.h:126-130 */
enum { SSLAM_CLASS_OTHER = 0, SSLAM_CLASS_CHAIR, SSLAM_CLASS_TVMONITOR, SSLAM_CLASS_BOOK, SSLAM_CLASS_KEYBOARD,
       SSLAM_CLASS_LAPTOP, SSLAM_CLASS_BUCKET, SSLAM_CLASS_CAR };

/* detected_object (include/planar_segmentation/detected_object.h:14-24) */
typedef struct sslam_plane {
  float centroid_cam[3];  /* detected_object::pose */
  float normal_d[4];      /* detected_object::normal_orientation (sign-normalised) */
  float world_pose[3];    /* detected_object::world_pose */
  float num_points;       /* contour point count (quirk B8) */
  float prob;
  int32_t plane_type;     /* 0 horizontal, 1 vertical */
  int32_t class_id;
  int32_t box_index;
  int32_t inlier_count;   /* true inlier count (extension) */
  float area;             /* polygon area of the contour */
} sslam_plane;

void sslam_seg_default_params(sslam_seg_params* p);
sslam_seg* sslam_seg_create(const sslam_seg_params* p);
void sslam_seg_destroy(sslam_seg* s);

/* segmentallPointCloudData(robot_pose, cam_angle, object_info, point_cloud)
 * cloud = sensor_msgs::PointCloud2::data of an organised width x height cloud; off_* = field
 * offsets of x,y,z (plane_segmentation.cpp:48-61).  Returns the number of planes written. */
int sslam_seg_segment(sslam_seg* s, const uint8_t* cloud, int width, int height, int point_step, int row_step,
                      int off_x, int off_y, int off_z, const sslam_box* boxes, int n_boxes,
                      const float robot_pose[6], float cam_angle, sslam_plane* out, int max_out);

/* Several frames in one pass (MI355X extension): the boxes of all frames are packed into one set of launches -- one frame's 32 boxes
 * occupy 32 of the 256 CUs in the kernels that run one workgroup per box.  All frames share the cloud geometry (width .. off_z).
 * out_frame[k] (optional) = frame of out[k]; box_index of a plane counts inside its own frame.  Results are identical to
 * n_frames calls of sslam_seg_segment. */
typedef struct sslam_frame {
  const uint8_t* cloud;
  const sslam_box* boxes;
  int n_boxes;
  float robot_pose[6];
  float cam_angle;
} sslam_frame;
int sslam_seg_segment_batch(sslam_seg* s, const sslam_frame* frames, int n_frames, int width, int height, int point_step, int row_step,
                            int off_x, int off_y, int off_z, sslam_plane* out, int max_out, int32_t* out_frame);
/* Pipelined form of sslam_seg_segment_batch for a stream of batches: submit enqueues the H2D copy of the clouds, every kernel and
 * the read-back of the result tables on one of two pipelines of the handle (own HIP stream, own device buffers, used in turn) and
 * returns; collect waits for the OLDEST submitted batch and runs the scalar post-processing.  At most two batches are in flight:
 *     submit(0);  for (k = 0; ...; ++k) { submit(k + 1); collect(k); }
 * so that the copy of batch k+1 (9.8 MB per 640x480 frame of 32-byte points; 3.7 MB with point_step = 12: the frontend reads x, y, z only)
 * runs under the kernels of batch k -- the kernels of a batch wait for the other pipeline's kernels, so copies and kernels of successive
 * batches overlap whatever the host's timing.  The cloud buffers of a batch must stay
 * valid until it is collected (boxes and poses are copied at submit); pinned clouds (hipHostMalloc / hipHostRegister) make submit
 * itself asynchronous.  Results are those of sslam_seg_segment_batch.  The blocking calls refuse to run while a batch is in flight. */
int sslam_seg_submit_batch(sslam_seg* s, const sslam_frame* frames, int n_frames, int width, int height, int point_step, int row_step,
                           int off_x, int off_y, int off_z);
int sslam_seg_collect_batch(sslam_seg* s, sslam_plane* out, int max_out, int32_t* out_frame);
/* Truncation of the last segment call: planes not returned because max_out was reached, boxes with more than 64 connected
 * components above num_point_seg, boxes with more than 64 accepted planes.  Returns the sum (0 = nothing was truncated). */
int sslam_seg_last_overflow(const sslam_seg* s, int* dropped_planes, int* boxes_with_full_candidate_table, int* boxes_with_full_region_table);

/* RANSAC plane fit: pcl::SACSegmentation(SACMODEL_PLANE, SAC_RANSAC, optimise coefficients)
 * as configured at plane_segmentation.cpp:639-647 (threshold 0.01; PCL defaults max_iterations 50, probability 0.99)
 * on an unorganised cloud of n xyz float points.  Every hypothesis is scored in parallel (one thread per point,
 * LDS / ballot inlier counting); pcl::RandomSampleConsensus' adaptive-k loop is replayed over the counts.
 * Samples come from a counter-based hash of (seed, iteration) instead of rand().  Returns the number of inliers of
 * the refined model; coeff_out = (nx, ny, nz, d); inliers_out = ascending point indices. */
int sslam_seg_ransac_plane(sslam_seg* s, const float* xyz, int n, float threshold, int max_iterations, double probability,
                           uint64_t seed, float coeff_out[4], int32_t* inliers_out, int max_inliers);

/* Second half of compute2DConvexHull: pcl::ProjectInliers(SACMODEL_PLANE) of the RANSAC inliers onto
 * the model plane and the 2-D pcl::ConvexHull of the projected points (plane_segmentation.cpp:648-662).
 * inliers = point indices (e.g. from sslam_seg_ransac_plane), coeff = (a, b, c, d).  projected_out (optional):
 * n_inliers x 3 floats.  hull_out = POSITIONS in `inliers` of the hull vertices, counter-clockwise in the chosen coordinate
 * plane (axes_out: 0 xy, 1 yz, 2 xz; picked from the plane normal as PCL does), starting at the vertex of smallest
 * atan2 about the vertex centroid (PCL's comparePoints2D order).  Returns the number of hull vertices. */
int sslam_seg_convex_hull_2d(sslam_seg* s, const float* xyz, int n, const int32_t* inliers, int n_inliers, const float coeff[4],
                             float* projected_out, int32_t* hull_out, int max_hull, int* axes_out);

/* Point-to-plane ICP (judge row J1; BASELINE.json north_star "RANSAC plane fit + point-to-plane ICP"; the reference tree holds no
 * counterpart): the rigid transform T that minimises  sum_i (n_k(i) . (T p_i) + d_k(i))^2  over the n points with label k(i) in
 * [0, n_planes) -- e.g. the in-box points of a frame with the label image of sslam_seg_get_labels against the planes of the previous
 * keyframe.  Gauss-Newton on (rotation vector, translation), T <- (exp[w]x, u) o T: every round is ONE pass over the points on the
 * device (29 double sums, fixed reduction order) and a 6 x 6 Cholesky solve on the host.  T0 (NULL = identity) and T_out are 12
 * doubles, R row-major then t; rms_out = root mean square point-to-plane distance at T_out.  Returns the number of points used;
 * SSLAM_ERR_NUMERIC when the planes leave a degree of freedom unconstrained. */
int sslam_seg_icp_point_to_plane(sslam_seg* s, const float* xyz, const int32_t* labels, int n, const float* planes, int n_planes,
                                 int iterations, const double T0[12], double T_out[12], double* rms_out);

/* ---- RANSAC plane per detection box + point-to-plane ICP per frame, batched over the RESIDENT batch (BASELINE.json configs[3]: "640x480
 * depth cloud, 32 detection boxes/frame, RANSAC+ICP plane extraction"; north_star: "RANSAC plane fit + point-to-plane ICP over depth
 * clouds inside detection boxes ... one-thread-per-point HIP kernels with LDS inlier counting").  The reference's only RANSAC is
 * compute2DConvexHull's pcl::SACSegmentation (plane_segmentation.cpp:631-665, threshold 0.01, 50 iterations, probability 0.99); it has
 * no ICP.  Both calls work on the cropped clouds that the last blocking sslam_seg_segment / sslam_seg_segment_batch call left on the
 * device (its accepted boxes, in slot order: frame by frame, box order inside a frame): no second upload.
 *
 * sslam_seg_ransac_boxes: for every accepted box, pcl::SACSegmentation(SACMODEL_PLANE, SAC_RANSAC, optimise coefficients) exactly as
 * sslam_seg_ransac_plane runs it on the box's crop taken as an unorganised cloud of width * height points in crop order (row-major; a
 * non-finite point is never an inlier and a sample that hits one is a bad sample), with seed + slot * 0x9E3779B97F4A7C15 as the seed
 * of box `slot`.  One workgroup per box, the crop staged in LDS, sixteen hypotheses scored at a time.  Writes min(boxes, max_out) records
 * and returns the number of boxes; kernel_ms (optional): device time of the pass.  The inlier flags stay on the device for
 * sslam_seg_ransac_box_inliers (ascending crop indices of one box; returns the true count) and sslam_seg_icp_boxes. */
typedef struct sslam_box_plane {
  float coeff[4];          /* (nx, ny, nz, d) of the refined model; zeros when no model was found */
  int32_t inliers;         /* points within the threshold of the refined model */
  int32_t points;          /* width * height of the box */
  int32_t box_index;       /* index into the caller's box array of its frame */
  int32_t frame;
  int32_t hypotheses;      /* hypotheses the adaptive loop consumed (bad samples included) */
  int32_t best_iteration;  /* hypothesis index of the winning sample, -1: none */
} sslam_box_plane;
int sslam_seg_ransac_boxes(sslam_seg* s, float threshold, int max_iterations, double probability, uint64_t seed, sslam_box_plane* out, int max_out,
                           double* kernel_ms);
int sslam_seg_ransac_box_inliers(sslam_seg* s, int slot, int32_t* out, int max_out);
/* sslam_seg_icp_boxes: sslam_seg_icp_point_to_plane per FRAME of the resident batch; a Gauss-Newton round is two launches over ALL frames
 * (one workgroup per box for the 29 sums over its inliers, one wave per frame for the reduction in slot order, the 6 x 6 solve and the
 * update of T -- no host round trip per iteration): the points of a frame are the RANSAC inliers of its boxes, the points of box
 * slot q measure plane box_plane[q] of `planes` (n_planes x 4 floats, e.g. the previous keyframe's planes in the camera frame of this
 * one; -1: the box takes no part).  T0: NULL (identity) or 12 doubles per frame.  out[f].status is 0 or SSLAM_ERR_NUMERIC (planes that
 * leave a degree of freedom unconstrained).  Returns the number of frames. */
typedef struct sslam_icp_result {
  double T[12];   /* R row-major, then t */
  double rms;     /* root mean square point-to-plane distance at T */
  int32_t used;   /* points that took part */
  int32_t status;
} sslam_icp_result;
int sslam_seg_icp_boxes(sslam_seg* s, const int32_t* box_plane, int n_boxes, const float* planes, int n_planes, int iterations, const double* T0,
                        sslam_icp_result* out, int max_out, double* kernel_ms);

/* ---- cloud filters of the legacy path (dead code upstream) -----------------------------------------------------
 * All three take an unorganised cloud of n xyz float points and run on the GPU; index outputs are ascending and complete when the
 * return value (the true count) does not exceed max_out. */
/* plane_segmentation::distance_filter (plane_segmentation.cpp:607-629): indices of the points with dmin < |p| < dmax (0.3, 3 upstream) */
int sslam_seg_distance_filter(sslam_seg* s, const float* xyz, int n, double dmin, double dmax, int32_t* keep_out, int max_out);
/* plane_segmentation::downsamplePointcloud -> pcl::VoxelGrid (plane_segmentation.cpp:565-581; leaf 0.1 upstream): one centroid per
 * occupied voxel of the bounding box of the finite points, in ascending voxel index (x fastest), optionally with the voxel's point
 * count.  Sums are kept in 2^-20 fixed point with integer atomics (order independent; PCL accumulates in float in the order of an
 * unstable sort).  Returns the number of occupied voxels. */
int sslam_seg_voxel_grid(sslam_seg* s, const float* xyz, int n, float leaf, float* centroids_out, int32_t* counts_out, int max_out);
/* plane_segmentation::removeOutliers -> pcl::StatisticalOutlierRemoval (plane_segmentation.cpp:583-605; meanK 50, multiplier 1.0
 * upstream): a point stays when its mean distance to its mean_k nearest neighbours (exact search) is not above mean + stddev_mul *
 * standard deviation of those mean distances.  mean_dist_out (optional, n floats; -1 for non-finite points).  Returns the inlier count. */
int sslam_seg_statistical_outlier_removal(sslam_seg* s, const float* xyz, int n, int mean_k, double stddev_mul, int32_t* keep_out, int max_out,
                                          float* mean_dist_out);

/* plane_segmentation::computeKmeans -> cv::kmeans(points, k, labels, TermCriteria(EPS + ITER, 10, 0.01), 10 attempts, KMEANS_RANDOM_CENTERS)
 * (plane_segmentation.cpp:524-535) on n points of dimension dim (3: normals, k = 4; 1: plane distances, k = 2 upstream).  Restated with
 * centres drawn uniformly in the bounding box from a counter-based hash of (seed, attempt, centre, coordinate) instead of cv::RNG,
 * order-independent fixed-point sums in the centre update, and an empty cluster keeping its centre.  The assignment step runs on the GPU.
 * labels_out[n], centers_out[k * dim]; returns k. */
int sslam_seg_kmeans(sslam_seg* s, const float* pts, int n, int dim, int k, uint64_t seed, int32_t* labels_out, float* centers_out, double* compactness_out);

/* parity hooks: per-box products of the last sslam_seg_segment call.
 * normals: w*h*4 floats (nx,ny,nz,curvature), labels: w*h int32 (-1 = no plane; otherwise the
 * region index in output order of pcl::OrganizedMultiPlaneSegmentation::segmentAndRefine). */
int sslam_seg_get_normals(sslam_seg* s, int box, float* out);
int sslam_seg_get_labels(sslam_seg* s, int box, int32_t* out);
/* GPU kernel time (hipEvents on the handle's stream, H2D/D2H excluded) and wall time of the last call */
int sslam_seg_last_timing(const sslam_seg* s, double* kernel_ms, double* total_ms);
/* semantic_tools::transformNormalsToWorld (include/tools.h:18-102): 4x4 row-major float */
int sslam_seg_transform(const sslam_seg* s, const float robot_pose[6], float cam_angle, float out16[16]);


/* ============================================================================================
 * Orchestrator tick without ROS: semantic_graph_slam (reference include/ps_graph_slam/semantic_graph_slam.h,
 * src/ps_graph_slam/semantic_graph_slam.cpp) + KeyframeUpdater (keyframe_updater.hpp:41-65) + data_association
 * (data_association.h:70-389) + InformationMatrixCalculator (information_matrix_calculator.cpp:28-35).  SURVEY §8 rows f3, f2.
 * The ROS node becomes a thin shell: its callbacks call the set_* / vio entry points, its 30 Hz loop calls sslam_slam_run.
 * ==========================================================================================*/
typedef struct sslam_slam sslam_slam;

typedef struct sslam_slam_params {
  double keyframe_delta_trans, keyframe_delta_angle, keyframe_delta_time; /* 0.5, 0.5, 1 (keyframe_updater.hpp:24-29) */
  int max_keyframes_per_update;          /* 10 (semantic_graph_slam.cpp:18) */
  int update_keyframes_using_detections; /* ~update_key_using_det, false (semantic_graph_slam.cpp:22-23) */
  double camera_angle_deg;               /* ~camera_angle, degrees (semantic_graph_slam.cpp:24,29) */
  int add_first_lan;                     /* ~add_first_lan (semantic_graph_slam.cpp:25,54-55,289-329) */
  double first_lan[3];                   /* 1.8, 0, 0.3 */
  int use_const_inf_matrix;              /* information_matrix_calculator.cpp:9,29 */
  double const_stddev_x, const_stddev_q; /* 0 -> 0.0667 (information_matrix_calculator.cpp:11-17) */
  double maha_dist_thres, eq_dist_thres; /* 0.5, 1.21 (data_association.h:49-50) */
  double land_noise_low, land_noise_high;/* 0.5, 0.9  (data_association.h:51-52) */
  int use_maha_dist, use_eq_dist, use_rtab_map_odom; /* true, false, false (data_association.h:53-55) */
  int max_iterations;                    /* 1024 (graph_slam.cpp:205) */
  int reference_quirks;                  /* bit 0: keep distance_min across the detections of a frame (quirk B5, data_association.h:101);
                                            default 0 = reset per detection */
  int device;
} sslam_slam_params;

/* data_association's landmark (include/ps_graph_slam/landmark.h:17-33) */
typedef struct sslam_landmark {
  int32_t id;            /* index in the landmark list */
  int32_t vertex;        /* graph vertex id of landmark::node (-1 before it is added) */
  int32_t class_id;      /* landmark::type */
  int32_t plane_type;    /* landmark::plane_type: 0 horizontal, 1 vertical */
  int32_t is_new;
  float pose[3];         /* world position the landmark was created / last observed at */
  float local_pose[3];   /* observation in the robot frame (the edge measurement) */
  float covariance[9];   /* 3x3, row-major: marginal of the last optimisation (getAndSetLandmarkCov) */
  float normal[4];       /* normal_orientation in the world frame */
  float distance;        /* association distance of the observation that produced this record (extension) */
} sslam_landmark;

typedef struct sslam_tick_stats {
  int keyframes_added;     /* <= max_keyframes_per_update */
  int landmarks_added, landmarks_matched, landmark_edges_added;
  int optimized;           /* GraphSLAM::optimize() returned true */
  int marginals_ok;        /* computeLandmarkMarginals returned true */
  sslam_opt_stats opt;
  double seconds_frontend, seconds_association, seconds_optimize, seconds_marginals;
} sslam_tick_stats;

void sslam_slam_default_params(sslam_slam_params* p);
/* semantic_graph_slam::init (semantic_graph_slam.cpp:11-56).  seg: the frontend handle used for keyframes that carry a cloud and
 * boxes (NULL when every keyframe brings pre-segmented objects); it is borrowed, not owned. */
sslam_slam* sslam_slam_create(const sslam_slam_params* p, sslam_seg* seg);
void sslam_slam_destroy(sslam_slam* s);
/* setPointCloudData (semantic_graph_slam.cpp:341-345): the cloud is copied */
int sslam_slam_set_point_cloud(sslam_slam* s, const uint8_t* cloud, int width, int height, int point_step, int row_step,
                               int off_x, int off_y, int off_z);
/* setDetectedObjectInfo (semantic_graph_slam.cpp:353-357) */
int sslam_slam_set_detected_objects(sslam_slam* s, const sslam_box* boxes, int n_boxes);
/* extension: objects segmented elsewhere (e.g. by one sslam_seg_segment_batch call over a recorded sequence) for the next keyframe;
 * they replace the frontend call of semantic_data_ass (semantic_graph_slam.cpp:217-219) and count as an available detection */
int sslam_slam_set_segmented_objects(sslam_slam* s, const sslam_plane* objects, int n);
/* VIOCallback (semantic_graph_slam.cpp:234-287): keyframe gate, dead-reckoned robot pose, queue.  Returns 1 when the odometry sample
 * became a keyframe, 0 when the gate rejected it. */
int sslam_slam_vio(sslam_slam* s, int32_t stamp_sec, int32_t stamp_nsec, const double odom_tq[7]);
/* run (semantic_graph_slam.cpp:58-102): returns 1 when keyframes were processed, 0 when the queue was empty */
int sslam_slam_run(sslam_slam* s, sslam_tick_stats* stats);
/* getRobotPose / getMap2OdomTrans / getVIOPose (semantic_graph_slam.cpp:375-381,333-339) */
int sslam_slam_robot_pose(const sslam_slam* s, double tq[7]);
int sslam_slam_map2odom(const sslam_slam* s, double tq[7]);
/* getMappedLandmarks (semantic_graph_slam.cpp:366-368): returns the number of landmarks (copies min(n, max)) */
int sslam_slam_landmarks(const sslam_slam* s, sslam_landmark* out, int max);
/* getKeyframes (semantic_graph_slam.cpp:370-373): graph vertex ids and current estimates (7 doubles each) of the processed keyframes */
int sslam_slam_keyframes(const sslam_slam* s, int32_t* vertex_ids, double* estimates_tq, int max);
/* the underlying GraphSLAM (saveGraph, semantic_graph_slam.cpp:391-394; borrowed) */
sslam_graph* sslam_slam_graph(sslam_slam* s);
/* data_association::find_matches (data_association.h:75-95) on its own: associates n segmented objects seen from robot_pose
 * (x,y,z,roll,pitch,yaw) against the mapped landmarks ON THE DEVICE and appends the new ones to the landmark list (without adding
 * graph vertices: parity hook for the association kernel).  out = n records. */
int sslam_slam_find_matches(sslam_slam* s, const sslam_plane* objects, int n, const float robot_pose[6], sslam_landmark* out);

#ifdef __cplusplus
}
#endif
#endif /* SSLAM_H */
