// hg_adapter.h — C++ host-side adapter over the C ABI (include/hg_mi355x.h).
//
// Mirrors the reference's C++ seams for the TSDF hot path so a maintainer can swap the CPU
// classes for these behind the same call sites (paths relative to /root/reference/cartographer/):
//   hg_amd::mapping::HybridGridTSDF            mapping/3d/hybrid_grid_tsdf.h:59-134
//   hg_amd::mapping::TSDFRangeDataInserter3D   mapping/3d/tsdf_range_data_inserter_3d.h (Insert)
//                                              via mapping/range_data_inserter_interface.h:37-45
//   hg_amd::mapping::scan_matching::TsdfScanMatcher3D
//                                              mapping/internal/3d/scan_matching/ceres_scan_matcher_3d.h
//                                              (Match / Evaluate shape, TSDF blocks only)
//   hg_amd::mapping::LocalTrajectoryBuilder3D  mapping/internal/3d/local_trajectory_builder_3d.h:46-79
//                                              (AddImuData / AddRangeData / AddOdometryData /
//                                              MatchingResult / InsertionResult)
//   hg_amd::mapping::OptimizingLocalTrajectoryBuilder
//                                              mapping/internal/3d/optimizing_local_trajectory_builder.cc
//                                              (sliding window: TSDF blocks :323-511, IMU pre-integration
//                                              blocks with velocity states :928-1000, odometry :1009-1074)
// Header-only, C++11, no Eigen: poses are std::array<double, 7> (t xyz, q wxyz), points are
// std::array<float, 3>. Errors throw hg_amd::Error (the reference CHECK-aborts instead).
#ifndef HG_ADAPTER_H_
#define HG_ADAPTER_H_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <deque>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/hg_mi355x.h"

namespace hg_amd {

struct Error : std::runtime_error {
  int code;
  Error(const std::string& what, int c) : std::runtime_error(what + ": " + hg_last_error()), code(c) {}
};
inline void Check(int rc, const char* what) {
  if (rc < 0) throw Error(what, rc);
}

using Pose = std::array<double, 7>;  // t.x t.y t.z q.w q.x q.y q.z
using Point = std::array<float, 3>;

namespace transform {
inline Pose Multiply(const Pose& a, const Pose& b) {  // Rigid3d operator* (rigid_transform.h:184-190)
  const double w = a[3], x = a[4], y = a[5], z = a[6];
  auto rot = [&](const double* v, double* o) {
    const double ux = y * v[2] - z * v[1], uy = z * v[0] - x * v[2], uz = x * v[1] - y * v[0];
    const double tx = ux + ux, ty = uy + uy, tz = uz + uz;
    o[0] = v[0] + w * tx + (y * tz - z * ty);
    o[1] = v[1] + w * ty + (z * tx - x * tz);
    o[2] = v[2] + w * tz + (x * ty - y * tx);
  };
  Pose r;
  rot(&b[0], &r[0]);
  for (int i = 0; i < 3; ++i) r[i] += a[i];
  r[3] = w * b[3] - x * b[4] - y * b[5] - z * b[6];
  r[4] = w * b[4] + x * b[3] + y * b[6] - z * b[5];
  r[5] = w * b[5] + y * b[3] + z * b[4] - x * b[6];
  r[6] = w * b[6] + z * b[3] + x * b[5] - y * b[4];
  const double n = std::sqrt(r[3] * r[3] + r[4] * r[4] + r[5] * r[5] + r[6] * r[6]);
  for (int i = 3; i < 7; ++i) r[i] /= n;
  return r;
}
inline Pose Inverse(const Pose& a) {  // Rigid3::inverse (rigid_transform.h:159-163)
  Pose c{{0, 0, 0, a[3], -a[4], -a[5], -a[6]}};
  Pose t{{-a[0], -a[1], -a[2], 1, 0, 0, 0}};
  Pose r = Multiply(c, t);
  r[3] = c[3]; r[4] = c[4]; r[5] = c[5]; r[6] = c[6];
  return r;
}
inline std::array<float, 7> ToFloat(const Pose& p) {
  std::array<float, 7> f;
  for (int i = 0; i < 7; ++i) f[i] = static_cast<float>(p[i]);
  return f;
}
}  // namespace transform

namespace sensor {
struct RangeData {  // sensor/range_data.h:44-57 (misses are unused by the TSDF inserter)
  Point origin{{0.f, 0.f, 0.f}};
  std::vector<Point> returns;
  size_t width = 0;
};
struct TimedPointCloudData {  // sensor/timed_point_cloud_data.h:27-32
  double time = 0.0;
  Point origin{{0.f, 0.f, 0.f}};
  std::vector<std::array<float, 4>> ranges;  // xyz + relative time
};
struct OdometryData { double time; Pose pose; };
struct ImuData { double time; std::array<double, 3> linear_acceleration, angular_velocity; };
}  // namespace sensor

class Context {
 public:
  explicit Context(int device = 0, void* stream = nullptr) { Check(hg_ctx_create(device, stream, &ctx_), "hg_ctx_create"); }
  ~Context() { if (ctx_) hg_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  hg_ctx* get() const { return ctx_; }
 private:
  hg_ctx* ctx_ = nullptr;
};

namespace sensor {
// sensor::VoxelFilter (sensor/internal/voxel_filter.h:30-47): the first point of every voxel, order kept.
class VoxelFilter {
 public:
  VoxelFilter(Context* ctx, float size) : ctx_(ctx), size_(size) {}
  std::vector<Point> Filter(const std::vector<Point>& point_cloud) const {
    std::vector<uint32_t> keep(point_cloud.size());
    size_t n = 0;
    Check(hg_voxel_filter(ctx_->get(), size_, point_cloud.empty() ? nullptr : point_cloud[0].data(), point_cloud.size(),
                          3, HG_HOST, keep.data(), &n), "hg_voxel_filter");
    std::vector<Point> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = point_cloud[keep[i]];
    return out;
  }
 private:
  Context* ctx_;
  float size_;
};

// sensor::AdaptiveVoxelFilter (sensor/internal/adaptive_voxel_filter.h:88-110) with the fields of
// proto::AdaptiveVoxelFilterOptions (max_length, min_num_points, max_range).
class AdaptiveVoxelFilter {
 public:
  AdaptiveVoxelFilter(Context* ctx, float max_length, float min_num_points, float max_range)
      : ctx_(ctx), max_length_(max_length), min_num_points_(min_num_points), max_range_(max_range) {}
  std::vector<Point> Filter(const std::vector<Point>& point_cloud) const {
    std::vector<uint32_t> keep(point_cloud.size());
    size_t n = 0;
    Check(hg_adaptive_voxel_filter(ctx_->get(), max_length_, min_num_points_, max_range_,
                                   point_cloud.empty() ? nullptr : point_cloud[0].data(), point_cloud.size(), 3, HG_HOST,
                                   keep.data(), &n), "hg_adaptive_voxel_filter");
    std::vector<Point> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = point_cloud[keep[i]];
    return out;
  }
 private:
  Context* ctx_;
  float max_length_, min_num_points_, max_range_;
};
}  // namespace sensor

namespace mapping {

class HybridGridTSDF {
 public:
  HybridGridTSDF(Context* ctx, float resolution, float relative_truncation_distance, float max_weight,
                 uint32_t max_blocks = 1u << 18)
      : resolution_(resolution), relative_truncation_distance_(relative_truncation_distance), max_weight_(max_weight) {
    Check(hg_grid_create(ctx->get(), resolution, relative_truncation_distance, max_weight, max_blocks, &grid_),
          "hg_grid_create");
  }
  ~HybridGridTSDF() { if (grid_) hg_grid_destroy(grid_); }
  HybridGridTSDF(const HybridGridTSDF&) = delete;
  HybridGridTSDF& operator=(const HybridGridTSDF&) = delete;
  float resolution() const { return resolution_; }
  float relative_truncation_distance() const { return relative_truncation_distance_; }
  float max_weight() const { return max_weight_; }
  std::array<int, 3> GetCellIndex(const Point& p) const {
    return {{static_cast<int>(std::lround(p[0] / resolution_)), static_cast<int>(std::lround(p[1] / resolution_)),
             static_cast<int>(std::lround(p[2] / resolution_))}};
  }
  void SetCell(const std::array<int, 3>& index, float tsd, float weight) {
    Check(hg_grid_set_cells(grid_, index.data(), 1, &tsd, &weight), "hg_grid_set_cells");
  }
  // Raw TSDFVoxel codes; unknown cells read {0, 0}.
  void ReadCells(const std::vector<std::array<int, 3>>& cells, std::vector<uint16_t>* tsd,
                 std::vector<uint16_t>* weight) const {
    tsd->resize(cells.size());
    weight->resize(cells.size());
    Check(hg_grid_read_cells(grid_, cells.empty() ? nullptr : cells[0].data(), cells.size(), tsd->data(),
                             weight->data()), "hg_grid_read_cells");
  }
  // What `for (auto it : grid)` / ToProto visits, in the reference iteration order.
  size_t Export(std::vector<std::array<int, 3>>* cells, std::vector<uint16_t>* tsd, std::vector<uint16_t>* weight) const {
    size_t n = 0;
    Check(hg_grid_count(grid_, &n), "hg_grid_count");
    cells->resize(n); tsd->resize(n); weight->resize(n);
    if (n) Check(hg_grid_export(grid_, (*cells)[0].data(), tsd->data(), weight->data(), n, &n), "hg_grid_export");
    return n;
  }
  hg_grid* get() const { return grid_; }
 private:
  float resolution_, relative_truncation_distance_, max_weight_;
  hg_grid* grid_ = nullptr;
};

// proto::SubmapQuery::Response::SubmapTexture as AddToTextureProto(const HybridGridTSDF&, ...) fills
// it (submap_3d.cc:245-276); `cells` is the uncompressed cell string, to be passed through
// common::FastGzipString by the caller.
struct SubmapTexture {
  std::string cells;
  int width = 0, height = 0;
  double resolution = 0.;
  Pose slice_pose{{0, 0, 0, 1, 0, 0, 0}};
};

inline SubmapTexture AddToTexture(const HybridGridTSDF& grid, const Pose& global_submap_pose) {
  SubmapTexture texture;
  texture.resolution = grid.resolution();
  int32_t width = 0, height = 0, max_index[2] = {0, 0};
  size_t bytes = 0;
  Check(hg_grid_xray(grid.get(), global_submap_pose.data(), nullptr, 0, &width, &height, max_index, &bytes),
        "hg_grid_xray");
  texture.cells.resize(bytes);
  if (bytes)
    Check(hg_grid_xray(grid.get(), global_submap_pose.data(), reinterpret_cast<uint8_t*>(&texture.cells[0]), bytes,
                       &width, &height, max_index, &bytes), "hg_grid_xray");
  texture.width = width;
  texture.height = height;
  const float resolution = grid.resolution();
  // global_submap_pose.inverse() * Translation(max_index.x * resolution, max_index.y * resolution, global z)
  const Pose translation{{static_cast<double>(max_index[0] * resolution), static_cast<double>(max_index[1] * resolution),
                          global_submap_pose[2], 1, 0, 0, 0}};
  texture.slice_pose = transform::Multiply(transform::Inverse(global_submap_pose), translation);
  return texture;
}

// proto::TSDFRangeDataInserterOptions3D defaults of configuration_files/trajectory_builder_3d.lua:78-93
inline hg_insert_opts DefaultTSDFInserterOptions() {
  hg_insert_opts o{};
  o.relative_truncation_distance = 2.5; o.maximum_weight = 1000.; o.num_free_space_voxels = 0;
  o.project_sdf_distance_to_scan_normal = 0; o.weight_function_epsilon = 1.0; o.weight_function_sigma = 4.;
  o.min_range = 0.4; o.max_range = 15.0; o.insertion_ratio = 1.0; o.normal_computation_method = 1;
  o.normal_computation_horizontal_stride = 5; o.normal_computation_vertical_stride = 1;
  return o;
}

class TSDFRangeDataInserter3D {
 public:
  // mode: HG_INSERT_EXACT (the reference's codes, bit for bit) or HG_INSERT_FAST (order-free tolerance mode)
  explicit TSDFRangeDataInserter3D(const hg_insert_opts& options, int mode = HG_INSERT_EXACT)
      : options_(options), mode_(mode) {}
  // RangeDataInserterInterface::Insert
  void Insert(const sensor::RangeData& range_data, HybridGridTSDF* grid) const {
    Check(hg_grid_insert(grid->get(), &options_, range_data.origin.data(),
                         range_data.returns.empty() ? nullptr : range_data.returns[0].data(),
                         range_data.returns.size(), range_data.width, nullptr, mode_, HG_HOST, nullptr),
          "hg_grid_insert");
  }
  bool RequiresStructuredData() const { return options_.project_sdf_distance_to_scan_normal != 0; }
  const hg_insert_opts& options() const { return options_; }
 private:
  hg_insert_opts options_;
  int mode_;
};

// Submap3D::InsertData for the TSDF grids of one submap (submap_3d.cc:427-452): frame change by
// local_pose().inverse().cast<float>() and both inserters, fused into one device pass.
inline void InsertIntoSubmap(const sensor::RangeData& range_data_in_local, const Pose& submap_local_pose,
                             const TSDFRangeDataInserter3D& high, const TSDFRangeDataInserter3D& low,
                             HybridGridTSDF* high_grid, HybridGridTSDF* low_grid) {
  hg_grid* grids[2] = {high_grid->get(), low_grid->get()};
  const hg_insert_opts opts[2] = {high.options(), low.options()};
  const std::array<float, 7> inv = transform::ToFloat(transform::Inverse(submap_local_pose));
  Check(hg_pyramid_insert(grids, opts, 2, range_data_in_local.origin.data(),
                          range_data_in_local.returns.empty() ? nullptr : range_data_in_local.returns[0].data(),
                          range_data_in_local.returns.size(), range_data_in_local.width, inv.data(), HG_INSERT_EXACT,
                          HG_HOST, nullptr), "hg_pyramid_insert");
}

// Submap3D + ActiveSubmaps3D for TSDF grids (submap_3d.h:58-141, submap_3d.cc:427-514): every range
// data goes into all live submaps (at most two); a new submap starts when the newest one holds
// num_range_data insertions and the oldest is finished — and dropped from the active set — at twice
// that number. The yaw histogram of the rotational scan matcher is outside the TSDF path.
class Submap3D {
 public:
  Submap3D(Context* ctx, float high_resolution, float low_resolution, const Pose& local_pose,
           float relative_truncation_distance, float maximum_weight, uint32_t max_blocks)
      : local_pose_(local_pose),
        high_(new HybridGridTSDF(ctx, high_resolution, relative_truncation_distance, maximum_weight, max_blocks)),
        low_(new HybridGridTSDF(ctx, low_resolution, relative_truncation_distance, maximum_weight, max_blocks)) {}
  const Pose& local_pose() const { return local_pose_; }
  int num_range_data() const { return num_range_data_; }
  bool insertion_finished() const { return finished_; }
  HybridGridTSDF& high_resolution_hybrid_grid() { return *high_; }
  HybridGridTSDF& low_resolution_hybrid_grid() { return *low_; }
  void InsertData(const sensor::RangeData& range_data_in_local, const TSDFRangeDataInserter3D& high,
                  const TSDFRangeDataInserter3D& low) {
    if (finished_) throw Error("Submap3D::InsertData after Finish", HG_ERR_INVALID);  // CHECK(!insertion_finished())
    InsertIntoSubmap(range_data_in_local, local_pose_, high, low, high_.get(), low_.get());
    ++num_range_data_;
  }
  void Finish() { finished_ = true; }

 private:
  Pose local_pose_;
  std::unique_ptr<HybridGridTSDF> high_, low_;
  int num_range_data_ = 0;
  bool finished_ = false;
};

class ActiveSubmaps3D {
 public:
  struct Options {  // proto::SubmapsOptions3D (trajectory_builder_3d.lua:64-117)
    float high_resolution = 0.10f, low_resolution = 0.45f;
    int num_range_data = 160;
    float relative_truncation_distance = 2.5f, maximum_weight = 1000.f;
    uint32_t max_blocks = 1u << 16;
    hg_insert_opts high_resolution_inserter = DefaultTSDFInserterOptions();
    hg_insert_opts low_resolution_inserter = DefaultTSDFInserterOptions();
  };
  ActiveSubmaps3D(Context* ctx, const Options& options)
      : ctx_(ctx), options_(options), high_inserter_(options.high_resolution_inserter),
        low_inserter_(options.low_resolution_inserter) {}
  const std::vector<std::shared_ptr<Submap3D>>& submaps() const { return submaps_; }
  // range_data in the local frame; a new submap is placed at the sensor origin with
  // `local_from_gravity_aligned` as its orientation (submap_3d.cc:496-501)
  const std::vector<std::shared_ptr<Submap3D>>& InsertData(const sensor::RangeData& range_data,
                                                            const std::array<double, 4>& local_from_gravity_aligned) {
    if (submaps_.empty() || submaps_.back()->num_range_data() == options_.num_range_data) {
      if (submaps_.size() > 1) submaps_.erase(submaps_.begin());  // AddSubmap: the finished one leaves (:555-559)
      const Pose origin{{range_data.origin[0], range_data.origin[1], range_data.origin[2], local_from_gravity_aligned[0],
                         local_from_gravity_aligned[1], local_from_gravity_aligned[2], local_from_gravity_aligned[3]}};
      submaps_.push_back(std::make_shared<Submap3D>(ctx_, options_.high_resolution, options_.low_resolution, origin,
                                                    options_.relative_truncation_distance, options_.maximum_weight,
                                                    options_.max_blocks));
    }
    for (auto& submap : submaps_) submap->InsertData(range_data, high_inserter_, low_inserter_);
    if (submaps_.front()->num_range_data() == 2 * options_.num_range_data) submaps_.front()->Finish();
    return submaps_;
  }

 private:
  Context* ctx_;
  Options options_;
  TSDFRangeDataInserter3D high_inserter_, low_inserter_;
  std::vector<std::shared_ptr<Submap3D>> submaps_;
};

namespace scan_matching {

struct PointCloudAndGrid {
  const std::vector<Point>* point_cloud;
  const HybridGridTSDF* grid;
};

class TsdfScanMatcher3D {
 public:
  TsdfScanMatcher3D(Context* ctx, std::vector<double> occupied_space_weights, int max_num_iterations = 12)
      : ctx_(ctx), weights_(std::move(occupied_space_weights)) {
    hg_solver_default_opts(&solver_);
    solver_.max_num_iterations = max_num_iterations;
    Check(hg_problem_create(ctx->get(), &problem_), "hg_problem_create");
  }
  ~TsdfScanMatcher3D() { if (problem_) hg_problem_destroy(problem_); }
  // CeresScanMatcher3D::Match (ceres_scan_matcher_3d.cc:72-95), TSDF blocks.
  void Match(const Pose& initial_pose_estimate, const std::vector<PointCloudAndGrid>& clouds_and_grids,
             Pose* pose_estimate, hg_solver_summary* summary) {
    Setup(initial_pose_estimate, clouds_and_grids);
    Check(hg_problem_solve(problem_, &solver_, summary), "hg_problem_solve");
    Check(hg_problem_get_pose(problem_, 0, pose_estimate->data()), "hg_problem_get_pose");
  }
  // CeresScanMatcher3D::Evaluate (:97-118): cost, residuals, and J^T J / J^T r instead of the dense J.
  void Evaluate(const Pose& pose, const std::vector<PointCloudAndGrid>& clouds_and_grids, double* cost,
                std::vector<double>* residuals, std::array<double, 36>* JtJ, std::array<double, 6>* Jtr) {
    Setup(pose, clouds_and_grids);
    if (residuals) residuals->resize(hg_problem_num_residuals(problem_));
    Check(hg_problem_evaluate(problem_, cost, residuals ? residuals->data() : nullptr, Jtr ? Jtr->data() : nullptr,
                              JtJ ? JtJ->data() : nullptr), "hg_problem_evaluate");
  }
 private:
  void Setup(const Pose& pose, const std::vector<PointCloudAndGrid>& cg) {
    Check(hg_problem_reset(problem_), "hg_problem_reset");
    const int p = hg_problem_add_pose(problem_, pose.data(), 0);
    Check(p, "hg_problem_add_pose");
    for (size_t i = 0; i < cg.size(); ++i) {
      hg_grid* g = cg[i].grid->get();
      const std::vector<Point>& pc = *cg[i].point_cloud;
      // occupied_space_weight_i / sqrt(N)  (ceres_scan_matcher_3d.cc:149-151)
      Check(hg_problem_add_block(problem_, pc.empty() ? nullptr : pc[0].data(), pc.size(), HG_HOST, &g, 1, 0,
                                 weights_.at(i) / std::sqrt(static_cast<double>(pc.size())), p, -1, 0.0),
            "hg_problem_add_block");
    }
  }
  Context* ctx_;
  std::vector<double> weights_;
  hg_solver_opts solver_;
  hg_problem* problem_ = nullptr;
};

// use_per_point_unwarping branch of AddPerPointMatchingResiduals (oltb.cc:513-612): subdivisions of
// `num_points_per_subdivision` consecutive returns, each timed at the mean of its first and last
// return and interpolated between the control points that bracket that time; subdivisions outside
// (front, back) are omitted. Times: universal 100 ns ticks for the cloud and control points, seconds
// relative to the cloud for the returns (TimedRangefinderPoint::time). The subdivisions of this
// cloud that share a control-point pair become one hg_problem_add_unwarped_block.
inline int64_t FromSecondsTicks(double seconds) { return static_cast<int64_t>(seconds * 1e7); }  // common/time.cc:30-33
inline double TicksToSeconds(int64_t ticks) { return static_cast<double>(ticks) / 1e7; }         // :35-38

inline void AddPerPointMatchingResiduals(hg_problem* problem, const std::vector<int>& pose_ids,
                                         const std::vector<int64_t>& control_times, int64_t cloud_time,
                                         const std::vector<Point>& points, const std::vector<float>& point_times,
                                         hg_grid* const* pyramid, int levels, bool multi_res, double weight,
                                         int num_points_per_subdivision) {
  const size_t n = points.size();
  const size_t pairs = control_times.size() < 2 ? 0 : control_times.size() - 1;
  std::vector<std::vector<float>> xyz(pairs);
  std::vector<std::vector<double>> ratios(pairs);
  for (size_t start = 0; start < n; start += num_points_per_subdivision) {
    const size_t end = std::min(start + num_points_per_subdivision - 1, n - 1);
    // float + float, then promoted by the double 0.5 (oltb.cc:537-542)
    const double center = 0.5 * (point_times[start] + point_times[end]);
    const int64_t t = cloud_time + FromSecondsTicks(center);
    if (!(t < control_times.back() && t > control_times.front())) continue;
    size_t next = 1;
    while (control_times[next] <= t) ++next;
    const double duration = TicksToSeconds(control_times[next] - control_times[next - 1]);
    const double ratio = std::min(1.0, std::max(0.0, TicksToSeconds(t - control_times[next - 1]) / duration));
    for (size_t i = start; i <= end; ++i) {
      xyz[next - 1].insert(xyz[next - 1].end(), points[i].begin(), points[i].end());
      ratios[next - 1].push_back(ratio);
    }
  }
  const double scaling = weight / std::sqrt(static_cast<double>(n));  // :573-577
  for (size_t k = 0; k < pairs; ++k) {
    if (ratios[k].empty()) continue;
    Check(hg_problem_add_unwarped_block(problem, xyz[k].data(), ratios[k].data(), ratios[k].size(), HG_HOST,
                                        pyramid, levels, multi_res ? 1 : 0, scaling, pose_ids[k], pose_ids[k + 1]),
          "hg_problem_add_unwarped_block");
  }
}

}  // namespace scan_matching

// LocalTrajectoryBuilder3D-shaped driver for the scan-matching + insertion subset: constant-velocity
// prediction from the last two poses (or odometry deltas when provided), multi-resolution TSDF match on
// the device, then insertion of the scan at the matched pose into every pyramid level.
class LocalTrajectoryBuilder3D {
 public:
  struct InsertionResult {
    std::vector<const HybridGridTSDF*> insertion_grids;
  };
  struct MatchingResult {
    double time;
    Pose local_pose;
    sensor::RangeData range_data_in_local;
    std::unique_ptr<const InsertionResult> insertion_result;  // nullptr if map update is disabled
  };
  struct Options {
    std::vector<float> resolutions{0.05f, 0.10f, 0.20f};
    float relative_truncation_distance = 2.5f, maximum_weight = 1000.f;
    uint32_t max_blocks = 1u << 18;
    hg_insert_opts inserter = DefaultTSDFInserterOptions();
    double high_resolution_grid_weight = 1.0;
    int max_num_iterations = 12;
    float min_range = 1.f, max_range = 60.f;  // trajectory_builder_3d.lua:18-19
    int insert_mode = HG_INSERT_EXACT;        // HG_INSERT_FAST: tolerance insert (see hg_mi355x.h)
  };

  LocalTrajectoryBuilder3D(Context* ctx, const Options& options) : ctx_(ctx), options_(options) {
    for (float r : options.resolutions)
      grids_.emplace_back(new HybridGridTSDF(ctx, r, options.relative_truncation_distance, options.maximum_weight,
                                             options.max_blocks));
    Check(hg_problem_create(ctx->get(), &problem_), "hg_problem_create");
    hg_solver_default_opts(&solver_);
    solver_.max_num_iterations = options.max_num_iterations;
    pose_ = Pose{{0, 0, 0, 1, 0, 0, 0}};
    prev_pose_ = pose_;
  }
  ~LocalTrajectoryBuilder3D() { if (problem_) hg_problem_destroy(problem_); }

  void AddImuData(const sensor::ImuData&) {}  // IMU residuals are outside the TSDF hot path
  void AddOdometryData(const sensor::OdometryData& odom) {
    if (have_odom_) odom_delta_ = transform::Multiply(transform::Inverse(last_odom_.pose), odom.pose);
    last_odom_ = odom;
    have_odom_ = true;
  }
  void SetMapUpdateEnabled(bool enabled) { map_update_enabled_ = enabled; }
  void UseScanMatching(bool use) { use_scan_matching_ = use; }

  std::unique_ptr<MatchingResult> AddRangeData(const std::string& /*sensor_id*/,
                                               const sensor::TimedPointCloudData& data) {
    // range crop + NaN drop (optimizing_local_trajectory_builder.cc:214-227)
    std::vector<Point> cloud;
    cloud.reserve(data.ranges.size());
    for (const auto& p : data.ranges) {
      const float dx = p[0] - data.origin[0], dy = p[1] - data.origin[1], dz = p[2] - data.origin[2];
      const float r = std::sqrt(dx * dx + dy * dy + dz * dz);
      if (r >= options_.min_range && r <= options_.max_range) cloud.push_back(Point{{p[0], p[1], p[2]}});
    }
    if (cloud.empty()) return nullptr;
    // prediction
    Pose predicted = num_scans_ < 2 ? pose_
                     : have_odom_  ? transform::Multiply(pose_, odom_delta_)
                                   : transform::Multiply(pose_, transform::Multiply(transform::Inverse(prev_pose_), pose_));
    Pose estimate = predicted;
    if (use_scan_matching_ && num_scans_ > 0) {
      Check(hg_problem_reset(problem_), "hg_problem_reset");
      const int p = hg_problem_add_pose(problem_, predicted.data(), 0);
      Check(p, "hg_problem_add_pose");
      std::vector<hg_grid*> pyr;
      for (auto& g : grids_) pyr.push_back(g->get());
      Check(hg_problem_add_block(problem_, cloud[0].data(), cloud.size(), HG_HOST, pyr.data(), static_cast<int>(pyr.size()),
                                 pyr.size() > 1, options_.high_resolution_grid_weight / std::sqrt(double(cloud.size())),
                                 p, -1, 0.0), "hg_problem_add_block");
      hg_solver_summary summary;
      Check(hg_problem_solve(problem_, &solver_, &summary), "hg_problem_solve");
      Check(hg_problem_get_pose(problem_, p, estimate.data()), "hg_problem_get_pose");
    }
    prev_pose_ = pose_;
    pose_ = estimate;
    ++num_scans_;
    std::unique_ptr<MatchingResult> result(new MatchingResult);
    result->time = data.time;
    result->local_pose = estimate;
    result->range_data_in_local.origin = data.origin;
    if (map_update_enabled_) {
      std::vector<hg_grid*> pyr;
      std::vector<hg_insert_opts> opts(grids_.size(), options_.inserter);
      for (auto& g : grids_) pyr.push_back(g->get());
      const std::array<float, 7> pf = transform::ToFloat(estimate);
      Check(hg_pyramid_insert(pyr.data(), opts.data(), static_cast<int>(pyr.size()), data.origin.data(), cloud[0].data(),
                              cloud.size(), 0, pf.data(), options_.insert_mode, HG_HOST, nullptr), "hg_pyramid_insert");
      std::unique_ptr<InsertionResult> ins(new InsertionResult);
      for (auto& g : grids_) ins->insertion_grids.push_back(g.get());
      result->insertion_result = std::move(ins);
    }
    return result;
  }
  const std::vector<std::unique_ptr<HybridGridTSDF>>& grids() const { return grids_; }

 private:
  Context* ctx_;
  Options options_;
  std::vector<std::unique_ptr<HybridGridTSDF>> grids_;
  hg_problem* problem_ = nullptr;
  hg_solver_opts solver_;
  Pose pose_, prev_pose_, odom_delta_{{0, 0, 0, 1, 0, 0, 0}};
  sensor::OdometryData last_odom_{};
  bool have_odom_ = false, map_update_enabled_ = true, use_scan_matching_ = true;
  int num_scans_ = 0;
};

// use_per_point_unwarping, second half (optimizing_local_trajectory_builder.cc:1331-1379, then :1437-1440 and
// submap_3d.cc:436-437): the clouds that leave the window are unwarped return by return with the window's solved
// control poses and inserted, all on the device (hg_pyramid_insert_unwarped). Times in seconds as elsewhere in this
// header; they become 100 ns ticks as common::FromSeconds makes them. control_poses.front() is optimized_pose.
inline void InsertUnwarped(const std::vector<HybridGridTSDF*>& grids, const hg_insert_opts& inserter,
                           const std::vector<sensor::TimedPointCloudData>& clouds, size_t width,
                           const std::vector<Pose>& control_poses, const std::vector<double>& control_times,
                           const std::array<float, 7>* submap_from_local = nullptr, int insert_mode = HG_INSERT_EXACT) {
  if (grids.empty() || clouds.empty() || control_poses.size() < 2 || control_poses.size() != control_times.size())
    throw Error("InsertUnwarped: needs grids, clouds and at least two control points", HG_ERR_INVALID);
  std::vector<hg_grid*> pyr;
  for (HybridGridTSDF* g : grids) pyr.push_back(g->get());
  std::vector<hg_insert_opts> opts(grids.size(), inserter);
  std::vector<hg_timed_cloud> table;
  std::vector<float> points;
  for (const auto& c : clouds) {
    hg_timed_cloud t{};
    t.time = static_cast<int64_t>(c.time * 1e7);
    t.begin = points.size() / 4;
    t.count = c.ranges.size();
    for (int k = 0; k < 3; ++k) t.origin[k] = c.origin[k];
    table.push_back(t);
    for (const auto& p : c.ranges) points.insert(points.end(), p.begin(), p.end());
  }
  std::vector<double> poses;
  std::vector<int64_t> times;
  for (size_t k = 0; k < control_poses.size(); ++k) {
    poses.insert(poses.end(), control_poses[k].begin(), control_poses[k].end());
    times.push_back(static_cast<int64_t>(control_times[k] * 1e7));
  }
  Check(hg_pyramid_insert_unwarped(pyr.data(), opts.data(), static_cast<int>(pyr.size()), points.data(), points.size() / 4,
                                   width, HG_HOST, table.data(), static_cast<int>(table.size()), poses.data(), times.data(),
                                   static_cast<int>(times.size()), submap_from_local ? submap_from_local->data() : nullptr,
                                   insert_mode, nullptr),
        "hg_pyramid_insert_unwarped");
}

// Pre-integrated rotation between two control points, the only part of the pre-integration result
// PredictionImuPreintegrationCostFunctor reads (prediction_imu_preintegration_cost_functor.h:81-84):
// the rotation recurrence of IntegrateImuWithTranslationEuler (imu_integration.h:99-131) --
// piecewise-constant angular velocity, delta_rotation *= AngleAxisVectorToRotationQuaternion(w * dt)
// (transform/transform.h:121-135) -- with identity calibration. Host scalar code, as in the reference.
// Returns (w, x, y, z). `imu` is ordered by time; samples before `start` other than the last one are
// ignored, and without a sample at or before `start` the first sample is held.
inline std::array<double, 4> IntegrateImuDeltaRotation(const std::deque<sensor::ImuData>& imu, double start,
                                                       double end) {
  std::array<double, 4> q{{1.0, 0.0, 0.0, 0.0}};
  if (imu.empty() || !(start < end)) return q;
  size_t it = 0;
  while (it + 1 < imu.size() && imu[it + 1].time <= start) ++it;
  double current = start;
  while (current < end) {
    const double next_imu = it + 1 < imu.size() ? imu[it + 1].time : std::numeric_limits<double>::infinity();
    const double next = std::min(next_imu, end);
    const double dt = next - current;
    const double ax = imu[it].angular_velocity[0] * dt, ay = imu[it].angular_velocity[1] * dt,
                 az = imu[it].angular_velocity[2] * dt;
    double scale = 0.5, w = 1.0;
    const double sq = ax * ax + ay * ay + az * az;
    if (sq > 1e-8) {  // kCutoffAngle: linearised below
      const double norm = std::sqrt(sq);
      scale = std::sin(norm / 2.0) / norm;
      w = std::cos(norm / 2.0);
    }
    const double x = scale * ax, y = scale * ay, z = scale * az;
    const std::array<double, 4> r{{q[0] * w - q[1] * x - q[2] * y - q[3] * z,   // Eigen quaternion product q * d
                                   q[0] * x + q[1] * w + q[2] * z - q[3] * y,
                                   q[0] * y + q[2] * w + q[3] * x - q[1] * z,
                                   q[0] * z + q[3] * w + q[1] * y - q[2] * x}};
    q = r;
    current = next;
    if (current == next_imu) ++it;
  }
  return q;
}

// OptimizingLocalTrajectoryBuilder-shaped driver (mapping/internal/3d/optimizing_local_trajectory_builder.cc):
// a sliding window of control points is re-optimised on every scan — AddRangeData (:188-264) queues the
// cloud and adds a control point, MaybeOptimize (:1114-1413) builds one problem over the window (first
// state constant :1268-1275, one TSDF block per scan :323-511, odometry blocks between neighbours
// :1009-1074), solves it on the device, and the scans that leave the window are inserted into the map at
// their optimised poses (:1332-1404). One control point per scan (the reference spaces them by
// ct_window_rate and interpolates; that is hg_problem_add_block's pose_b / interpolation_ratio).
class OptimizingLocalTrajectoryBuilder {
 public:
  typedef LocalTrajectoryBuilder3D::MatchingResult MatchingResult;
  typedef LocalTrajectoryBuilder3D::InsertionResult InsertionResult;
  struct Options : LocalTrajectoryBuilder3D::Options {
    int window = 5;  // control points kept in the window (ct_window_horizon / ct_window_rate)
    double odometry_translation_weight = 1.0, odometry_rotation_weight = 1.0;  // trajectory_builder_3d.lua:126-127
    // imu_cost_term = "PREINTEGRATION" with velocity_in_state (trajectory_builder_3d.lua:123-125,133,144):
    // every neighbouring pair of control points gets a PredictionImuPreintegrationCostFunctor block once
    // IMU data has arrived; all three weights zero switches the blocks off (oltb.cc:928-937)
    double imu_translation_weight = 1.0, imu_velocity_weight = 1.0, imu_rotation_weight = 1.0;
  };

  OptimizingLocalTrajectoryBuilder(Context* ctx, const Options& options) : ctx_(ctx), options_(options) {
    for (float r : options.resolutions)
      grids_.emplace_back(new HybridGridTSDF(ctx, r, options.relative_truncation_distance, options.maximum_weight,
                                             options.max_blocks));
    Check(hg_problem_create(ctx->get(), &problem_), "hg_problem_create");
    hg_solver_default_opts(&solver_);
    solver_.max_num_iterations = options.max_num_iterations;
  }
  ~OptimizingLocalTrajectoryBuilder() { if (problem_) hg_problem_destroy(problem_); }

  // IMU samples are queued (oltb.cc:166-186); MaybeOptimize pre-integrates them between neighbouring
  // control points on the host and hands the delta rotation to hg_problem_add_imu_block (:968-1000)
  void AddImuData(const sensor::ImuData& imu) {
    if (!imu_data_.empty() && imu.time < imu_data_.back().time) throw Error("AddImuData: samples must arrive in time order", HG_ERR_INVALID);
    imu_data_.push_back(imu);
  }
  void AddOdometryData(const sensor::OdometryData& odom) { last_odom_ = odom; have_odom_ = true; }
  int num_imu_blocks_in_last_solve() const { return last_imu_blocks_; }
  std::array<double, 3> velocity(size_t control_point) const { return window_[control_point].velocity; }

  std::unique_ptr<MatchingResult> AddRangeData(const std::string& /*sensor_id*/,
                                               const sensor::TimedPointCloudData& data) {
    ControlPoint cp;
    cp.time = data.time;
    cp.origin = data.origin;
    for (const auto& p : data.ranges) {  // range crop (:214-227)
      const float dx = p[0] - data.origin[0], dy = p[1] - data.origin[1], dz = p[2] - data.origin[2];
      const float r = std::sqrt(dx * dx + dy * dy + dz * dz);
      if (r >= options_.min_range && r <= options_.max_range) cp.cloud.push_back(Point{{p[0], p[1], p[2]}});
    }
    if (cp.cloud.empty()) return nullptr;
    cp.has_odom = have_odom_;
    if (have_odom_) cp.odom = last_odom_.pose;
    // prediction: odometry delta, else constant velocity, else the previous pose (:266-321)
    if (window_.empty()) cp.pose = Pose{{0, 0, 0, 1, 0, 0, 0}};
    else if (cp.has_odom && window_.back().has_odom)
      cp.pose = transform::Multiply(window_.back().pose, transform::Multiply(transform::Inverse(window_.back().odom), cp.odom));
    else if (window_.size() >= 2)
      cp.pose = transform::Multiply(window_.back().pose,
                                    transform::Multiply(transform::Inverse(window_[window_.size() - 2].pose), window_.back().pose));
    else cp.pose = window_.back().pose;
    // velocity of the new state: the translation of the predicted step over its duration (the
    // constant-velocity form of PredictStateOdom, oltb.cc:1641); zero for the first control point
    if (!window_.empty() && cp.time > window_.back().time)
      for (int k = 0; k < 3; ++k) cp.velocity[k] = (cp.pose[k] - window_.back().pose[k]) / (cp.time - window_.back().time);
    window_.push_back(std::move(cp));

    std::vector<hg_grid*> pyr;
    for (auto& g : grids_) pyr.push_back(g->get());
    if (map_has_data_) {
      // MaybeOptimize: one problem over the window
      Check(hg_problem_reset(problem_), "hg_problem_reset");
      const bool imu_blocks = !imu_data_.empty() && window_.size() > 1 &&
                              (options_.imu_translation_weight != 0.0 || options_.imu_velocity_weight != 0.0 ||
                               options_.imu_rotation_weight != 0.0);
      last_imu_blocks_ = 0;
      for (size_t i = 0; i < window_.size(); ++i) {
        const bool first_constant = i == 0 && window_.size() > 1;  // :1268-1275 (velocity too)
        Check(hg_problem_add_pose(problem_, window_[i].pose.data(), first_constant), "hg_problem_add_pose");
        if (imu_blocks)
          Check(hg_problem_set_velocity(problem_, static_cast<int>(i), window_[i].velocity.data(), first_constant),
                "hg_problem_set_velocity");
      }
      for (size_t i = 1; imu_blocks && i < window_.size(); ++i) {
        const std::array<double, 4> dq = IntegrateImuDeltaRotation(imu_data_, window_[i - 1].time, window_[i].time);
        Check(hg_problem_add_imu_block(problem_, static_cast<int>(i) - 1, static_cast<int>(i),
                                       options_.imu_translation_weight, options_.imu_velocity_weight,
                                       options_.imu_rotation_weight, window_[i].time - window_[i - 1].time, dq.data()),
              "hg_problem_add_imu_block");
        ++last_imu_blocks_;
      }
      for (size_t i = (window_.size() > 1 ? 1 : 0); i < window_.size(); ++i) {
        const std::vector<Point>& c = window_[i].cloud;
        Check(hg_problem_add_block(problem_, c[0].data(), c.size(), HG_HOST, pyr.data(), static_cast<int>(pyr.size()),
                                   pyr.size() > 1, options_.high_resolution_grid_weight / std::sqrt(double(c.size())),
                                   static_cast<int>(i), -1, 0.0), "hg_problem_add_block");
        if (i > 0 && window_[i].has_odom && window_[i - 1].has_odom) {
          // RelativeTranslationAndYawCostFunction: delta = inverse(next odometry) * previous (:1025-1029)
          const Pose delta = transform::Multiply(transform::Inverse(window_[i].odom), window_[i - 1].odom);
          Check(hg_problem_add_odometry_block(problem_, static_cast<int>(i) - 1, static_cast<int>(i),
                                              options_.odometry_translation_weight, options_.odometry_rotation_weight,
                                              delta.data()), "hg_problem_add_odometry_block");
        }
      }
      hg_solver_summary& summary = last_summary_;
      Check(hg_problem_solve(problem_, &solver_, &summary), "hg_problem_solve");
      ++num_solves_;
      for (size_t i = 0; i < window_.size(); ++i) {
        Check(hg_problem_get_pose(problem_, static_cast<int>(i), window_[i].pose.data()), "hg_problem_get_pose");
        if (imu_blocks)
          Check(hg_problem_get_velocity(problem_, static_cast<int>(i), window_[i].velocity.data()), "hg_problem_get_velocity");
      }
      // IMU samples older than the sample in force at the window's first control point are done with
      while (imu_data_.size() > 1 && imu_data_[1].time <= window_.front().time) imu_data_.pop_front();
    }
    std::unique_ptr<MatchingResult> result(new MatchingResult);
    result->time = data.time;
    result->local_pose = window_.back().pose;
    result->range_data_in_local.origin = data.origin;
    // scans leaving the window (and the very first scan, which seeds the map) are inserted
    while (!window_.empty() && (!map_has_data_ || static_cast<int>(window_.size()) > options_.window)) {
      ControlPoint& out = window_.front();
      if (!out.inserted) {
        std::vector<hg_insert_opts> opts(grids_.size(), options_.inserter);
        const std::array<float, 7> pf = transform::ToFloat(out.pose);
        Check(hg_pyramid_insert(pyr.data(), opts.data(), static_cast<int>(pyr.size()), out.origin.data(),
                                out.cloud[0].data(), out.cloud.size(), 0, pf.data(), options_.insert_mode, HG_HOST, nullptr),
              "hg_pyramid_insert");
        out.inserted = true;
        last_inserted_pose_ = out.pose;
        std::unique_ptr<InsertionResult> ins(new InsertionResult);
        for (auto& g : grids_) ins->insertion_grids.push_back(g.get());
        result->insertion_result = std::move(ins);
      }
      if (!map_has_data_) {  // keep the seeding scan as the constant first state of the next windows
        map_has_data_ = true;
        break;
      }
      window_.erase(window_.begin());
    }
    return result;
  }
  const std::vector<std::unique_ptr<HybridGridTSDF>>& grids() const { return grids_; }
  size_t window_size() const { return window_.size(); }
  const Pose& pose(size_t control_point) const { return window_[control_point].pose; }
  const hg_solver_summary& last_summary() const { return last_summary_; }  // of the last MaybeOptimize solve
  int num_solves() const { return num_solves_; }
  const Pose& last_inserted_pose() const { return last_inserted_pose_; }  // where the last leaving scan went into the map

 private:
  struct ControlPoint {
    double time = 0;
    Pose pose, odom;
    std::array<double, 3> velocity{{0.0, 0.0, 0.0}};
    bool has_odom = false, inserted = false;
    std::array<float, 3> origin;
    std::vector<Point> cloud;
  };
  Context* ctx_;
  Options options_;
  std::vector<std::unique_ptr<HybridGridTSDF>> grids_;
  hg_problem* problem_ = nullptr;
  hg_solver_opts solver_;
  std::vector<ControlPoint> window_;
  std::deque<sensor::ImuData> imu_data_;
  int last_imu_blocks_ = 0;
  hg_solver_summary last_summary_{};
  int num_solves_ = 0;
  Pose last_inserted_pose_{{0, 0, 0, 1, 0, 0, 0}};
  sensor::OdometryData last_odom_{};
  bool have_odom_ = false, map_has_data_ = false;
};

}  // namespace mapping
}  // namespace hg_amd
#endif  // HG_ADAPTER_H_
