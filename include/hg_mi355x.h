/* hg_mi355x.h — C ABI of the MI355X (gfx950) TSDF insertion + TSDF scan-matching path.
 *
 * Drop-in boundary behind HectorGrapher's C++ seams (all paths relative to
 * /root/reference/cartographer/):
 *   - hg_grid_*          replaces  mapping/3d/hybrid_grid_tsdf.h:59-134 (HybridGridTSDF storage,
 *                                  codec mapping/2d/tsd_value_converter.h:39-67, indexing
 *                                  mapping/3d/hybrid_grid_base.h:428-446, iteration :304-372)
 *   - hg_grid_insert*    replaces  RangeDataInserterInterface::Insert
 *                                  (mapping/range_data_inserter_interface.h:37-45) as implemented by
 *                                  TSDFRangeDataInserter3D::Insert
 *                                  (mapping/3d/tsdf_range_data_inserter_3d.cc:395-737), with the
 *                                  optional frame change of Submap3D::InsertData
 *                                  (mapping/3d/submap_3d.cc:436-437)
 *   - hg_problem_*       replaces  the ceres::Problem that OptimizingLocalTrajectoryBuilder builds
 *                                  from the TSDF cost functions and hands to ceres::Solve
 *                                  (mapping/internal/3d/optimizing_local_trajectory_builder.cc:
 *                                  323-511,1238-1291; cost functors
 *                                  mapping/internal/3d/scan_matching/{,interpolated_}{,multi_resolution_}tsdf_space_cost_function_3d.h)
 *   - hg_match_*         replaces  CeresScanMatcher3D::{Match,Evaluate} for one TSDF block
 *                                  (mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.cc:72-118)
 *
 * Conventions: every function returns an int status (HG_OK = 0, negative = error); nothing
 * aborts or throws. Handles are not thread-safe; all work of a context runs on one HIP stream.
 * Pointers are plain host pointers unless the parameter says `memspace`, where HG_DEVICE means
 * a device pointer valid on the context's device. Poses are (t.x t.y t.z q.w q.x q.y q.z).
 */
#ifndef HG_MI355X_H_
#define HG_MI355X_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: the declarations below are its whole export list. */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

typedef struct hg_ctx hg_ctx;
typedef struct hg_grid hg_grid;
typedef struct hg_problem hg_problem;

enum {
  HG_OK = 0,
  HG_ERR_INVALID = -1,     /* bad argument */
  HG_ERR_NO_DEVICE = -2,   /* no HIP device / wrong architecture */
  HG_ERR_HIP = -3,         /* HIP runtime error, see hg_last_error */
  HG_ERR_CAPACITY = -4,    /* block pool or hash table full */
  HG_ERR_UNSUPPORTED = -5, /* option not implemented on the device path */
  HG_ERR_RANGE = -6,       /* cell index outside +-8192 (reference: CHECK_LE(new_bits, 8)) */
  HG_ERR_TIME = -7         /* a return's time lies outside the control points (reference: CHECK_LE / CHECK_GE,
                              optimizing_local_trajectory_builder.cc:1358-1359) */
};

enum { HG_HOST = 0, HG_DEVICE = 1 };

/* Insert modes. HG_INSERT_EXACT reproduces the reference's sequential per-voxel update order
 * bit for bit. HG_INSERT_FAST sums the updates a voxel receives in one call (order-free integer
 * sums) and applies them once: UpdateCell (tsdf_range_data_inserter_3d.cc:725-737) in closed form
 * with a single quantisation. Codes differ from the exact mode by the re-quantisation noise the
 * reference accumulates per update (|d tsd| within 1e-4 m per update of the voxel); results do not
 * depend on arrival order. Needs unit update weights (weight_function_epsilon >= 1), no
 * free-space voxels, relative_truncation_distance <= 3; otherwise HG_ERR_UNSUPPORTED. Allocates
 * 8 bytes per voxel of the grid's block pool on first use. */
enum { HG_INSERT_EXACT = 0, HG_INSERT_FAST = 1 };

/* Fields of proto::TSDFRangeDataInserterOptions3D that the path reads
 * (mapping/proto/3d/tsdf_range_data_inserter_options_3d.proto:5-49). */
typedef struct hg_insert_opts {
  double relative_truncation_distance;
  double maximum_weight;
  int32_t num_free_space_voxels;
  int32_t project_sdf_distance_to_scan_normal;
  double weight_function_epsilon;
  double weight_function_sigma;
  double min_range;
  double max_range;
  double insertion_ratio;
  int32_t normal_computation_method; /* 1 = CLOUD_STRUCTURE */
  int32_t normal_computation_horizontal_stride;
  int32_t normal_computation_vertical_stride;
  int32_t reserved;
} hg_insert_opts;

typedef struct hg_insert_stats {
  uint64_t num_hits;    /* returns that reached the ray walk (N_in) */
  uint64_t num_updates; /* UpdateCell calls applied (U) */
  uint64_t num_blocks;  /* allocated 8^3 blocks after the call */
  uint64_t flags;       /* bit0: capacity exceeded, bit1: index out of range */
} hg_insert_stats;

/* Ceres 1.13 Solver::Options fields the path sets or relies on
 * (common/ceres_solver_options.cc:35-43 + Ceres defaults). */
typedef struct hg_solver_opts {
  int32_t max_num_iterations;
  int32_t jacobi_scaling;
  double initial_trust_region_radius;
  double max_trust_region_radius;
  double min_trust_region_radius;
  double min_relative_decrease;
  double min_lm_diagonal;
  double max_lm_diagonal;
  double function_tolerance;
  double gradient_tolerance;
  double parameter_tolerance;
} hg_solver_opts;

/* Subset of ceres::Solver::Summary. */
typedef struct hg_solver_summary {
  double initial_cost, final_cost, final_radius;
  int32_t num_iterations, num_successful_steps, num_unsuccessful_steps;
  int32_t num_cost_evaluations, num_jacobian_evaluations;
  int32_t termination_type;   /* 0 CONVERGENCE, 1 NO_CONVERGENCE, 2 FAILURE */
  int32_t termination_reason; /* 1 gradient, 2 parameter, 3 function tol, 4 max iter, 5 min radius, 6 invalid steps */
  int32_t reserved;
} hg_solver_summary;

/* ---- context ---------------------------------------------------------------------------- */
/* `stream` is a hipStream_t to run on, or NULL to create a private stream.
 * Several contexts of one process (the reference's deployment: a thread and a trajectory builder per trajectory,
 * mapping/map_builder.cc:120-175) run side by side: every private stream is created with a CU mask of all CUs, which
 * gives it a hardware queue of its own. Measured on MI355X (cpp/example_threads, 100k-point registration steps): 1.95x
 * of one context at two, 2.16x at three; at FOUR the first and the fourth context run at half pace (5.4k scans/s in
 * all against 8.8k at three) whatever GPU_MAX_HW_QUEUES says and with ordinary streams of distinct priorities too
 * (7.3k): the device gives a process' compute queues three pipes, the fourth busy queue shares one with the first,
 * and a pipe serves one queue's dependent launch chain at a time. Rule: up to THREE concurrently busy contexts per
 * process and device; beyond that hg_register_scan_batch / hg_problem_solve_batch (one context, shared launches:
 * 9.8k scans/s at eight submaps), or further processes. */
int hg_ctx_create(int device, void* stream, hg_ctx** out);
int hg_ctx_destroy(hg_ctx* ctx);
/* Waits for the context's stream. Also returns the error (HG_ERR_CAPACITY, HG_ERR_RANGE, ...) an
 * insertion that ran without a stats read-back has raised on ANY grid of the context since that grid
 * was last cleared (the flags are kept per grid: hg_grid_clear / hg_grid_destroy reset only the grid's
 * own). */
int hg_ctx_synchronize(hg_ctx* ctx);
/* Tuning and diagnostic switches of a context (round 6: they used to be HG_* environment variables read inside the
 * library; the environment now only supplies the DEFAULT a context starts from, variable HG_<KEY IN UPPER CASE>).
 * None changes what a call computes beyond what its entry states (summation orders of the matcher, launch shapes,
 * the linear solver of the LM step); results stay inside every tolerance the header gives. Keys:
 *   persistent_solve   1: the single-pose registration solve runs as ONE launch whose workgroups loop over the
 *                         evaluations and wait for each other inside the launch (same arithmetic in the same order:
 *                         bitwise the same poses; no launch gaps, no empty launches behind convergence). OFF by
 *                         default: the launch needs all of its workgroups resident at once -- one 512-thread
 *                         workgroup per CU -- which holds while the GPU is this context's alone; the library checks
 *                         what it can see (occupancy x CUs >= workgroups, no other context in this process) and
 *                         every wait inside the launch is bounded (two seconds; the solve then reports FAILURE,
 *                         termination_reason 7, and the context returns to a launch per evaluation), but another
 *                         PROCESS holding CUs is invisible to it. Set it where the device is not shared.
 *                         0: a launch per evaluation (default)
 *   ticket_handover    1: partial sums by acknowledged stores + ticket (rounds 1-4) instead of tagged granules
 *   prepare_kernel     1: a separate head-upload launch in front of a single-pose solve
 *   eager_solve        1: every launch of a general solve enqueued up front (no held-back tail)
 *   lazy_tail          launches of a general solve held back until needed (default 3)
 *   lm_general         1: single-pose problems through the general k_lm path
 *   lm_band / lm_btd_generic / lm_btd_chain / lm_btd_cr   1: the band / generic block / one-wavefront chain /
 *                         cyclic-reduction factorisation of the LM step instead of the default (twisted block)
 *   window_capacity, window_tiles, batch_tiles   launch shapes of the window / batched residual passes (0: automatic)
 *   window_batch       0: windows of a batch solved one after the other
 *   partition_min      batch size from which the level partition is used (default 32; 0: never)
 *   partition_at       LM iteration from which the batched residual launches read the partitioned returns (default 2)
 *   partition_fold     1 (default): the residual launch in front of that iteration classifies; 0: a lookup pass of its own
 *   host_times         1: host-side timing of hg_register_scan_sequence printed to stderr
 *   stream_group       scans of a stream whose front ends and apply pass share launches (default 32, at most 32; 0: off)
 *   stream_slice       records per voxel slice of large bins in a scan stream (default 1024)
 *   stream_merge       0: one apply launch per scan of a stream (rounds 3-5) instead of one per group of
 *                         stream_group scans (units of (block, voxel slice) x scans, applied in scan order; default 1).
 *                         Bit-identical results either way
 *   apply_turns        1: the levels of k_bin_apply take turns
 *   defer_long_chains  0: no deferral of long chains (only in builds with -DHG_DEFER_LONG_CHAINS)
 *   insert_sort        1: exact insertion through the radix-sort path
 *   insert_pipeline    0: small-scan streams on one stream
 * Unknown key: HG_ERR_INVALID. */
int hg_ctx_set_option(hg_ctx* ctx, const char* key, long long value);
int hg_ctx_get_option(hg_ctx* ctx, const char* key, long long* value);
void* hg_ctx_stream(hg_ctx* ctx);
const char* hg_last_error(void);
const char* hg_version(void);

/* ---- per-kernel timing (HIP events on the context's stream) ------------------------------ */
enum {
  HG_K_RAY_COUNT = 0, HG_K_RAY_EXPAND = 1, HG_K_SORT = 2, HG_K_ALLOC = 3, HG_K_APPLY = 4,
  HG_K_RESIDUALS = 5, HG_K_LM = 6, HG_K_SCAN = 7, HG_K_UNWARP = 8, HG_K_COUNT = 9
};
/* on = 1: bracket every launch of the kernels above with hipEvents (costs ~2 event records per
 * launch); on = 2: only the residual family (HG_K_RESIDUALS); 0: off. hg_prof_read synchronises the stream and returns launches / summed milliseconds since
 * the last hg_prof_reset; `units` sums the per-launch work items (points, records). */
int hg_prof_enable(hg_ctx* ctx, int on);
int hg_prof_reset(hg_ctx* ctx);
int hg_prof_read(hg_ctx* ctx, int kernel, uint64_t* launches, double* total_ms, uint64_t* units);

/* ---- grid: HybridGridTSDF --------------------------------------------------------------- */
/* max_blocks = number of 8x8x8-voxel blocks (2 KiB each) the grid may hold, < 2^22. Device memory:
 * the pool has next_pow2(max_blocks) (at most 2^21) directly addressed slots -- a toroidal window of
 * blocks that lets the matcher address voxels without a hash probe -- plus max_blocks overflow slots,
 * i.e. 2..3 * max_blocks * (2 KiB voxels + 24 B bookkeeping): 2^18 -> 1 GiB, 2^16 -> 256 MiB, 2^12 ->
 * 16 MiB per grid (+ 8 B per voxel of the pool after the first HG_INSERT_FAST call). hg_grid_clear
 * memsets all of it. Size grids that only receive an import (hg_grid_import_blocks, hg_grid_from_proto)
 * or a gather by their block count, not by the mapping default. At most 4096 grids per context. */
int hg_grid_create(hg_ctx* ctx, float resolution, float relative_truncation_distance,
                   float max_weight, uint32_t max_blocks, hg_grid** out);
int hg_grid_destroy(hg_grid* grid);
int hg_grid_clear(hg_grid* grid);
float hg_grid_resolution(const hg_grid* grid);
/* Constants of the grid's TSDValueConverter (mapping/2d/tsd_value_converter.h:39-67): getMaxTSD()
 * (= relative_truncation_distance * resolution in float; getMinTSD() = -getMaxTSD()), getMaxWeight(),
 * and the capacity of the block pool. Any output may be NULL. */
int hg_grid_params(const hg_grid* grid, float* resolution, float* max_tsd, float* max_weight,
                   uint32_t* max_blocks);
/* SetCell for m cells (host arrays): ijk[m*3], tsd[m], weight[m] floats are encoded by the codec. */
int hg_grid_set_cells(hg_grid* grid, const int32_t* ijk, size_t m, const float* tsd,
                      const float* weight);
/* Raw codes (update marker included, as the reference stores them). Unknown cells read 0/0. */
int hg_grid_read_cells(hg_grid* grid, const int32_t* ijk, size_t m, uint16_t* tsd,
                       uint16_t* weight);
int hg_grid_count(hg_grid* grid, size_t* count);
/* Non-default voxels in the reference's iterator order (what ToProto emits). */
int hg_grid_export(hg_grid* grid, int32_t* ijk, uint16_t* tsd, uint16_t* weight, size_t cap,
                   size_t* count);
int hg_grid_num_blocks(hg_grid* grid, uint32_t* num_blocks);
/* State of the grid's directly addressed block window (synchronises): out[0] blocks held, out[1]
 * blocks in the overflow area, out[2..4] window size in blocks per axis (x, y, z), out[5..7] extent
 * of the blocks' bounding box in blocks per axis, out[8] = 1 while lookups take the direct path (no
 * overflow block and the bounding box fits the window; otherwise they go through the hash: same
 * results, one more memory round trip). A larger max_blocks gives a larger window. */
int hg_grid_window_status(hg_grid* grid, uint32_t out[9]);
/* Packed device copy of the grid's blocks for the multi-GPU gather: keys[num_blocks] (u64 block
 * keys) and voxels[num_blocks*512] (u32), in the form hg_grid_import_blocks takes. The arrays belong
 * to the grid and stay valid until its next hg_grid_block_arrays / hg_grid_destroy. */
int hg_grid_block_arrays(hg_grid* grid, void** keys_dev, void** voxels_dev, uint32_t* num_blocks);
/* Merge blocks (e.g. received from another rank) into this grid; existing blocks are overwritten. */
int hg_grid_import_blocks(hg_grid* grid, const void* keys, const void* voxels, uint32_t num_blocks,
                          int memspace);

/* proto::HybridGridTSDF wire bytes (mapping/proto/3d/hybrid_grid_tsdf.proto:19-31) as
 * HybridGridTSDF::ToProto / the proto constructor produce and consume them
 * (mapping/3d/hybrid_grid_tsdf.h:69-83,119-134), including the reference's habit of storing
 * getMaxTSD() in the relative_truncation_distance field. buf == NULL: only *len is returned. */
int hg_grid_to_proto(hg_grid* grid, uint8_t* buf, size_t cap, size_t* len);
int hg_grid_from_proto(hg_ctx* ctx, const uint8_t* buf, size_t len, uint32_t max_blocks,
                       hg_grid** out);

/* X-ray texture of the grid, the view Submap3D::ToResponseProto serves: replaces
 * AddToTextureProto(const HybridGridTSDF&, ...) (mapping/3d/submap_3d.cc:245-276) with its helpers
 * ExtractVoxelData (:142-177), AccumulatePixelData (:80-105) and ComputePixelValues (:179-214).
 * global_submap_pose: double[7] (t xyz, q wxyz). cells receives the UNcompressed cell string
 * (value, alpha per pixel; pixel (x, y) at x * width + y, width = y extent, height = x extent);
 * the caller gzips it (common::FastGzipString) and forms slice_pose = global_submap_pose^-1 *
 * Translation(max_index_xy[0] * resolution, max_index_xy[1] * resolution, global z) (:271-275).
 * cells == NULL: only the sizes are computed. cap < *bytes: HG_ERR_CAPACITY with the sizes set.
 * A grid without a voxel above the obstruction limit gives width = height = 0 (the reference's
 * bounding box is undefined there). */
int hg_grid_xray(hg_grid* grid, const double* global_submap_pose, uint8_t* cells, size_t cap,
                 int32_t* width, int32_t* height, int32_t* max_index_xy, size_t* bytes);

/* ---- insertion: TSDFRangeDataInserter3D::Insert ----------------------------------------- */
/* xyz: n x 3 floats in the grid (submap) frame, or — when pose_tq != NULL — in the frame that
 * pose_tq (float[7], = local_pose().inverse().cast<float>()) maps into the grid frame; the origin
 * is transformed the same way. */
int hg_grid_insert(hg_grid* grid, const hg_insert_opts* opts, const float origin[3],
                   const float* xyz, size_t n, size_t width, const float* pose_tq, int mode,
                   int memspace, hg_insert_stats* stats);
/* Stream form: n_scans scans applied in order; scan s owns points [scan_offsets[s], scan_offsets[s+1]).
 * origins: n_scans x 3, poses_tq: n_scans x 7 or NULL (host arrays); xyz per memspace. */
int hg_grid_insert_batch(hg_grid* grid, const hg_insert_opts* opts, const float* origins,
                         const float* xyz, const uint64_t* scan_offsets, size_t n_scans,
                         size_t width, const float* poses_tq, int mode, int memspace,
                         hg_insert_stats* stats);

/* Pyramid form: the same range data inserted into `levels` grids (e.g. the high- and low-resolution
 * grids of Submap3D::InsertData, submap_3d.cc:441-444) with per-level options opts[levels], in one
 * fused device pass. stats: array[levels] or NULL. With stats == NULL the call does not synchronise;
 * errors (capacity, range) raised on the device then surface as the return value of the next
 * hg_pyramid_insert* (stats == NULL) or hg_register_scan* call on one of THESE grids, of
 * hg_ctx_synchronize (any grid of the context) and of hg_grid_status (the kernels leave the sticky
 * flags in a host-mapped word per grid; calls on other grids of the context are not affected). */
int hg_pyramid_insert(hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                      const float origin[3], const float* xyz, size_t n, size_t width,
                      const float* pose_tq, int mode, int memspace, hg_insert_stats* stats);
int hg_pyramid_insert_batch(hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                            const float* origins, const float* xyz, const uint64_t* scan_offsets,
                            size_t n_scans, size_t width, const float* poses_tq, int mode,
                            int memspace, hg_insert_stats* stats);
/* ---- per-point unwarping before insertion ------------------------------------------------ */
/* One sensor::TimedPointCloudData of the accumulation (point_cloud_data_ entries that leave the window). */
typedef struct hg_timed_cloud {
  int64_t time;    /* common::Time of the cloud in universal 100 ns ticks */
  uint64_t begin;  /* its returns are points[begin .. begin + count); the clouds tile [0, n) in order */
  uint64_t count;
  float origin[3]; /* sensor origin in the tracking frame (TimedPointCloudData::origin) */
  float reserved;
} hg_timed_cloud;
/* use_per_point_unwarping branch of OptimizingLocalTrajectoryBuilder::MaybeOptimize
 * (mapping/internal/3d/optimizing_local_trajectory_builder.cc:1331-1379), then the frame changes of
 * AddAccumulatedRangeData (:1437-1440) and Submap3D::InsertData (mapping/3d/submap_3d.cc:436-437), then
 * TSDFRangeDataInserter3D::Insert -- all on the device, no host pass over the returns:
 *   points: n x 4 floats (x y z time), sensor::TimedRangefinderPoint in the tracking frame at its own time,
 *   time in seconds relative to its cloud; every return is moved with
 *   (control_poses[0].inverse() * InterpolateTransform(prev, next, t_prev, t_next, cloud.time +
 *   FromSeconds(time))).cast<float>() for the control points bracketing it (transform/
 *   timestamped_transform.h:41-65), NaN returns are kept as they are (:1342-1345), the origin is the first
 *   unwarped return's transform applied to its cloud's origin (:1370-1374); then
 *   control_poses[0].cast<float>() (optimized_pose = the front control point, :1294-1295) and, when
 *   pose_tq != NULL, pose_tq (float[7] = local_pose().inverse().cast<float>()).
 * control_poses: n_control x 7 doubles (host), control_times: ascending ticks (host). A return whose time
 * lies outside [control_times[0], control_times[n_control - 1]] raises HG_ERR_TIME (the reference
 * CHECK-fails before anything is inserted, :1358-1359), reported like the other sticky insert errors; the
 * failed call inserts NOTHING (on the device every return of such a call is replaced by NaN before the
 * insertion reads the cloud, and NaN returns are dropped by its gates), so the grids hold what they held
 * before. The flag stays set until hg_grid_clear. width / mode / stats as hg_pyramid_insert. */
int hg_pyramid_insert_unwarped(hg_grid* const* grids, const hg_insert_opts* opts, int levels,
                               const float* points, size_t n, size_t width, int memspace,
                               const hg_timed_cloud* clouds, int n_clouds, const double* control_poses,
                               const int64_t* control_times, int n_control, const float* pose_tq, int mode,
                               hg_insert_stats* stats);
/* The unwarping alone. frame 0: accumulated_range_data_in_tracking (:1331-1379); frame 1: additionally
 * moved by control_poses[0].cast<float>() (range_data_in_local) and, if given, pose_tq. xyz_out (n x 3)
 * and origin_out are host arrays or NULL; the device copies stay valid until the context's next unwarp
 * call (hg_unwarp_last_device), e.g. as input of hg_voxel_filter / hg_pyramid_insert with HG_DEVICE.
 * Returns HG_ERR_TIME when a return's time lies outside the control points (every return of the result is
 * then NaN, on the host and in the device copy, so nothing of it can reach a map). With both outputs NULL
 * the call only enqueues: hg_unwarp_status waits for the context's stream and returns the status of its
 * last unwarp call (HG_OK / HG_ERR_TIME). */
int hg_unwarp_range_data(hg_ctx* ctx, const float* points, size_t n, int memspace, const hg_timed_cloud* clouds,
                         int n_clouds, const double* control_poses, const int64_t* control_times, int n_control,
                         int frame, const float* pose_tq, float* xyz_out, float origin_out[3]);
int hg_unwarp_last_device(hg_ctx* ctx, const float** xyz_dev, const float** origin_dev, size_t* count);
int hg_unwarp_status(hg_ctx* ctx);
/* Synchronises and returns the counters of the last insert call + sticky error flags. */
int hg_grid_status(hg_grid* grid, hg_insert_stats* stats);

/* ---- voxel filters (the step before matching) ------------------------------------------- */
/* VoxelFilter(resolution).Filter (sensor/internal/voxel_filter.cc:26-37): keeps the first point
 * of every voxel, input order preserved. pts: n points of `stride` floats (3 = PointCloud, 4 =
 * TimedPointCloud, time ignored). indices_out (host, capacity n, may be NULL) receives the kept
 * indices; *count their number. Cells are keyed by 3 x 21 bits, or, when a cell index reaches
 * +-2^20, by the reference's 3 x 32 bits (voxel_filter.cc:64-69; slower pass). */
int hg_voxel_filter(hg_ctx* ctx, float resolution, const float* pts, size_t n, int stride,
                    int memspace, uint32_t* indices_out, size_t* count);
/* AdaptiveVoxelFilter::Filter (sensor/internal/adaptive_voxel_filter.h:33-110): FilterByMaxRange,
 * then the binary search on the voxel edge length for >= min_num_points points. The trial lengths
 * are >= max_length / 128 and the points lie within max_range, so cells stay inside 3 x 21 bits for
 * max_range / max_length < 8192; beyond that HG_ERR_RANGE. */
int hg_adaptive_voxel_filter(hg_ctx* ctx, float max_length, float min_num_points, float max_range,
                             const float* pts, size_t n, int stride, int memspace,
                             uint32_t* indices_out, size_t* count);
/* Device-resident results of the last filter call of this context (valid until the next one):
 * kept indices and the gathered xyz (3 floats per kept point), e.g. for hg_problem_add_block. */
int hg_filter_last_device(hg_ctx* ctx, const uint32_t** indices_dev, const float** xyz_dev,
                          size_t* count);

/* ---- scan matching: ceres::Problem over TSDF cost functions ----------------------------- */
int hg_problem_create(hg_ctx* ctx, hg_problem** out);
int hg_problem_destroy(hg_problem* p);
/* Drops all poses and blocks but keeps the device buffers (one ceres::Problem per solve in the
 * reference; reuse avoids hipMalloc in the loop). */
int hg_problem_reset(hg_problem* p);
/* Returns the pose index (>= 0) or an error. constant != 0: SetParameterBlockConstant. */
int hg_problem_add_pose(hg_problem* p, const double tq[7], int constant);
int hg_problem_set_pose(hg_problem* p, int index, const double tq[7]);
int hg_problem_get_pose(hg_problem* p, int index, double tq[7]);
/* One residual block. pose_b < 0: [MultiResolution]TSDFSpaceCostFunction3D on pose_a;
 * pose_b >= 0: Interpolated[MultiResolution]TSDFSpaceCostFunction3D between pose_a and pose_b at
 * interpolation_ratio. multi_res != 0 selects the InterpolatedMultiResolutionTSDF lookup over
 * `pyramid` (ascending voxel size), else pyramid[0] with the single-resolution lookup.
 * The points are copied (HG_HOST) or referenced (HG_DEVICE: must stay valid). */
int hg_problem_add_block(hg_problem* p, const float* xyz, size_t n, int memspace,
                         hg_grid* const* pyramid, int levels, int multi_res,
                         double scaling_factor, int pose_a, int pose_b,
                         double interpolation_ratio);
/* Per-point unwarping (use_per_point_unwarping, oltb.cc:513-612 and :613-683): the reference adds
 * one Interpolated[MultiResolution]TSDFSpaceCostFunction3D per subdivision of
 * num_points_per_subdivision returns, and one InterpolatedTSDFPerPointSpaceCostFunction3D
 * (…/scan_matching/interpolated_tsdf_per_point_space_cost_function_3d.h:36-82) per low-resolution
 * return, each with its own interpolation ratio between the SAME two control points. Here all
 * returns bracketed by (pose_a, pose_b) form one block; interpolation_ratios[i] is return i's ratio
 * (returns of one subdivision repeat their subdivision's ratio). Residual order = return order.
 * n >= 1, pose_a != pose_b >= 0. */
int hg_problem_add_unwarped_block(hg_problem* p, const float* xyz, const double* interpolation_ratios,
                                  size_t n, int memspace, hg_grid* const* pyramid, int levels,
                                  int multi_res, double scaling_factor, int pose_a, int pose_b);
/* Sliding-window blocks that are not TSDF lookups (oltb.cc:928-1074). Control point `index` may
 * carry a velocity parameter block (state.h:11-31); constant != 0 = SetParameterBlockConstant.
 * Columns per control point: 6 pose (if free) then 3 velocity (if free). */
int hg_problem_set_velocity(hg_problem* p, int index, const double v[3], int constant);
int hg_problem_get_velocity(hg_problem* p, int index, double v[3]);
/* RelativeTranslationAndYawCostFunction(translation_weight, rotation_weight, delta_pose) between the
 * previous (a) and next (b) control point: 6 residuals
 * (mapping/internal/3d/scan_matching/relative_translation_and_yaw_cost_function.h:41-63). */
int hg_problem_add_odometry_block(hg_problem* p, int pose_a, int pose_b, double translation_weight,
                                  double rotation_weight, const double delta_tq[7]);
/* PredictionImuPreintegrationCostFunctor(translation_w, velocity_w, rotation_w, delta_time,
 * pre-integrated delta_rotation): 9 residuals over (t, v, q) of a and b
 * (…/scan_matching/prediction_imu_preintegration_cost_functor.h:49-101). Needs velocities on both
 * control points. */
int hg_problem_add_imu_block(hg_problem* p, int pose_a, int pose_b, double translation_weight,
                             double velocity_weight, double rotation_weight, double delta_time_seconds,
                             const double delta_rotation_wxyz[4]);
/* Tells that the cloud of TSDF block `block` is a structured scan stored azimuth-major with `width` returns
 * per column (sensor::RangeData::width / TimedPointCloudData::width: vertical neighbours adjacent, as the
 * inserter's CLOUD_STRUCTURE code assumes, tsdf_range_data_inserter_3d.cc:505-508). A performance hint only:
 * the residual kernels then let adjacent lanes take horizontally adjacent returns, whose voxels share cache
 * lines (the residual sums are formed in another order: results agree to rounding, residual positions are
 * unchanged). width = 0, or a return count that is not a multiple of width, keeps the plain order.
 * hg_register_scan* apply their `width` argument to the block that holds the same cloud. */
int hg_problem_set_block_width(hg_problem* p, int block, size_t width);
int hg_problem_num_residuals(hg_problem* p);
int hg_problem_num_columns(hg_problem* p);
/* ceres::Problem::Evaluate shape: cost = 0.5 |r|^2; residuals[num_residuals]; gradient and JtJ in
 * the local (tangent) parameterisation, 6 columns per non-constant pose; any output may be NULL. */
int hg_problem_evaluate(hg_problem* p, double* cost, double* residuals, double* gradient,
                        double* JtJ);
int hg_solver_default_opts(hg_solver_opts* opts);
/* ceres::Solve: Levenberg-Marquardt trust region on the device; poses are updated in place. */
int hg_problem_solve(hg_problem* p, const hg_solver_opts* opts, hg_solver_summary* summary);
/* Solves `count` INDEPENDENT problems of one context, e.g. the scan-to-submap matches a constraint
 * search hands to CeresScanMatcher3D::Match one by one (mapping/internal/3d/scan_matching/
 * ceres_scan_matcher_3d.cc:72-118). Problems of the single-pose shape (one free pose, one TSDF
 * block, no odometry / IMU blocks) share their kernel launches (one grid row per problem); every
 * problem keeps its own solver state; poses and costs agree with hg_problem_solve on each to the rounding of
 * the normal-equation sums (the batched pass sums over smaller workgroups, and from batches of partition_min
 * problems on it re-orders the returns of every problem after its second step -- same terms, another association;
 * observed 1e-12). Iteration counts and termination are those of hg_problem_solve wherever a decision of the
 * trust-region loop does not hinge on that rounding (an accept / reject or tolerance test within ~1e-12 of its
 * threshold can fall the other way; not observed in the tests' batches).
 * Other shapes are solved one after the other. summaries: count entries or NULL. */
int hg_problem_solve_batch(hg_problem* const* problems, int count, const hg_solver_opts* opts,
                           hg_solver_summary* summaries);

/* Enqueue-only solve and the matching read-back (hg_problem_solve = both). Between the two the
 * solved poses live in device memory only. */
int hg_problem_solve_async(hg_problem* p, const hg_solver_opts* opts);
int hg_problem_fetch(hg_problem* p, hg_solver_summary* summary);
/* Enqueue-only form of hg_problem_solve_batch (= this call, then hg_problem_fetch on every problem). The host is
 * free while the batch runs: a constraint search builds the problems of batch k + 1 (another set of hg_problem
 * handles) and enqueues them while batch k is being solved, then fetches batch k -- the device never waits for the
 * host between batches. Any number of batches may be in flight; a problem must be fetched before it is reset or
 * solved again. Batches of general (window-shaped) problems return when their solve is nearly finished (their last
 * iterations are enqueued on demand), single-pose batches at once. */
int hg_problem_solve_batch_async(hg_problem* const* problems, int count, const hg_solver_opts* opts);
/* One registration step (what OptimizingLocalTrajectoryBuilder::AddRangeData does with a scan,
 * optimizing_local_trajectory_builder.cc:1283 then :1437-1499): solve the prepared problem, then
 * insert `xyz` — given in the frame of pose `pose_index` (tracking frame), `origin` likewise —
 * into the pyramid at the SOLVED pose (cast to float as optimized_pose.cast<float>()), without a
 * host round trip in between; one synchronisation at the end returns pose and summary. The call
 * returns when the pose has arrived, while its insertion may still run: an insertion error (block
 * pool exhausted, index out of range) is returned by the NEXT hg_register_scan* / hg_ctx_synchronize
 * call of the context, never lost. */
int hg_register_scan(hg_problem* p, const hg_solver_opts* sopts, int pose_index,
                     hg_grid* const* grids, const hg_insert_opts* iopts, int levels,
                     const float origin[3], const float* xyz, size_t n, size_t width, int memspace,
                     double pose_out[7], hg_solver_summary* summary);
/* The same step with the insert mode chosen: HG_INSERT_EXACT (what hg_register_scan does) or
 * HG_INSERT_FAST — the match is unchanged, the insertion costs the same wherever the sensor stands
 * (no per-voxel chain), the map differs from the reference's within the tolerance of that mode. */
int hg_register_scan_mode(hg_problem* p, const hg_solver_opts* sopts, int pose_index,
                          hg_grid* const* grids, const hg_insert_opts* iopts, int levels,
                          const float origin[3], const float* xyz, size_t n, size_t width, int memspace,
                          int insert_mode, double pose_out[7], hg_solver_summary* summary);

/* The window step with per-point unwarping (MaybeOptimize with use_per_point_unwarping: ceres::Solve
 * :1283, unwarp :1331-1379, AddAccumulatedRangeData :1437-1440, insertion): solve the prepared window
 * problem, then hg_pyramid_insert_unwarped of `points` with the SOLVED control poses taken from device
 * memory -- control point k of the unwarping is pose pose_index[k] of the problem (pose_index[0] = the
 * front of the window = optimized_pose), control_times[k] its time. No host round trip between the solve
 * and the insertion; returns when the poses have arrived (insertion errors as hg_register_scan).
 * poses_out: n_control x 7 or NULL. */
int hg_register_scan_unwarped(hg_problem* p, const hg_solver_opts* sopts, hg_grid* const* grids,
                              const hg_insert_opts* iopts, int levels, const float* points, size_t n,
                              size_t width, int memspace, const hg_timed_cloud* clouds, int n_clouds,
                              const int* pose_index, const int64_t* control_times, int n_control,
                              const float* pose_tq, int insert_mode, double* poses_out,
                              hg_solver_summary* summary);

/* The registration step of `count` INDEPENDENT submaps with shared launches (offline batch mapping puts
 * several submaps on one GPU: one registration chain is latency-bound and fills a fraction of the
 * chip). Submap j: problems[j] prepared as for hg_register_scan (single-pose shape), its pyramid
 * grids[j * levels .. j * levels + levels), its scan xyz[j] (n[j] returns in the frame of pose
 * pose_index[j], HG_DEVICE memory), origin origins + 3 j; iopts[levels] are shared. Poses, iteration
 * counts and termination are those of hg_register_scan on every submap (poses agree to the rounding
 * of the normal-equation sums, see hg_problem_solve_batch); voxel codes are the reference's for the
 * poses returned. Other problem shapes, host memory or count == 1 run one registration after the
 * other. poses_out: count x 7 or NULL; summaries: count entries or NULL. Exact insert mode. */
int hg_register_scan_batch(hg_problem* const* problems, int count, const hg_solver_opts* sopts,
                           const int* pose_index, hg_grid* const* grids, const hg_insert_opts* iopts,
                           int levels, const float* origins, const float* const* xyz, const size_t* n,
                           size_t width, int memspace, double* poses_out, hg_solver_summary* summaries);

/* `count` registration steps of ONE trajectory, one after the other, in one call: what a C++
 * LocalTrajectoryBuilder3D does per scan (AddRangeData: predicted pose -> scan match -> insert,
 * local_trajectory_builder_3d.cc / optimizing_local_trajectory_builder.cc:1283,1437-1499) with the
 * predictions given up front. Step k: problem reset, one free pose = guesses + 7 k, one
 * [MultiResolution]TSDFSpaceCostFunction3D block over scan xyz[k] (n[k] returns, scaling[k]), solve,
 * insert the scan at the solved pose (hg_register_scan). Results are those of calling the steps one by
 * one; hosts whose per-step overhead matters (an interpreter) use this form. prof_every > 0: HIP-event
 * sampling (hg_prof_*) is switched on for every prof_every-th step, for the residual family only
 * except on every (5 prof_every)-th step, and off otherwise; 0 leaves the profiling state alone.
 * poses_out: count x 7 or NULL; summaries: count entries or NULL. */
int hg_register_scan_sequence(hg_problem* p, const hg_solver_opts* sopts, hg_grid* const* grids,
                              const hg_insert_opts* iopts, int levels, int multi_res, const float* origins,
                              const float* const* xyz, const size_t* n, const double* scaling, size_t width,
                              int memspace, int insert_mode, const double* guesses, int count,
                              int prof_every, double* poses_out, hg_solver_summary* summaries);

/* ---- one-block convenience (CeresScanMatcher3D::{Evaluate,Match} shape) ----------------- */
int hg_match_evaluate(hg_ctx* ctx, hg_grid* const* pyramid, int levels, int multi_res,
                      const float* xyz, size_t n, int memspace, double scaling_factor,
                      const double pose0[7], const double* pose1 /* NULL: single pose */,
                      double interpolation_ratio, double* cost, double* JtJ, double* Jtr,
                      double* residuals);
int hg_match_solve(hg_ctx* ctx, hg_grid* const* pyramid, int levels, int multi_res,
                   const float* xyz, size_t n, int memspace, double scaling_factor,
                   double pose0[7], double* pose1, int pose0_constant,
                   double interpolation_ratio, const hg_solver_opts* opts,
                   hg_solver_summary* summary);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* HG_MI355X_H_ */
