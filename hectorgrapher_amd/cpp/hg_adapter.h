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
//                                              mapping/internal/3d/optimizing_local_trajectory_builder.cc in its
//                                              own shape: control-point sampling :1162-1232, clouds bracketed and
//                                              interpolated between control points :323-511, IMU pre-integration
//                                              blocks with velocity states :928-1000, odometry blocks with adaptive
//                                              weights :1009-1074, MaybeOptimize :1113-1413
//   hg_amd::mapping::SlidingWindowTrajectoryBuilder   a simplified window (one control point per scan)
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

namespace common {
// common::Time / common::Duration (common/time.h:36-42): universal time in 100 ns ticks, integers end to end --
// every comparison of a scan's time with a control point's is exact, as in the reference.
typedef int64_t Time;
typedef int64_t Duration;
inline Duration FromSeconds(double seconds) { return static_cast<int64_t>(seconds * 1e7); }  // duration_cast truncates (time.cc:30-33)
inline double ToSeconds(Duration duration) { return static_cast<double>(duration) / 1e7; }   // (:35-38)
}  // namespace common

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
// Rigid3f * Vector3f: Eigen's Quaternion<float>::_transformVector, then + translation (rigid_transform.h:193-197)
inline Point TransformPoint(const std::array<float, 7>& p, const Point& v) {
  const float qw = p[3], qx = p[4], qy = p[5], qz = p[6];
  float ux = qy * v[2] - qz * v[1], uy = qz * v[0] - qx * v[2], uz = qx * v[1] - qy * v[0];
  ux = ux + ux; uy = uy + uy; uz = uz + uz;
  const float cx = qy * uz - qz * uy, cy = qz * ux - qx * uz, cz = qx * uy - qy * ux;
  const float rx = v[0] + qw * ux + cx, ry = v[1] + qw * uy + cy, rz = v[2] + qw * uz + cz;
  return Point{{rx + p[0], ry + p[1], rz + p[2]}};
}
inline std::array<double, 3> Rotate(const std::array<double, 4>& q, const std::array<double, 3>& v) {  // Quaterniond * Vector3d
  const double w = q[0], x = q[1], y = q[2], z = q[3];
  const double ux = y * v[2] - z * v[1], uy = z * v[0] - x * v[2], uz = x * v[1] - y * v[0];
  const double tx = ux + ux, ty = uy + uy, tz = uz + uz;
  return {{v[0] + w * tx + (y * tz - z * ty), v[1] + w * ty + (z * tx - x * tz), v[2] + w * tz + (x * ty - y * tx)}};
}
// transform::GetAngle (transform.h:34-37) and Eigen's angularDistance to the identity: 2 atan2(|vec|, |w|)
inline double GetAngle(const Pose& p) {
  return 2.0 * std::atan2(std::sqrt(p[4] * p[4] + p[5] * p[5] + p[6] * p[6]), std::abs(p[3]));
}
// InterpolateTransform (timestamped_transform.h:41-51): translation lerp, Eigen 3.3 QuaternionBase::slerp in double
inline Pose InterpolateTransform(const Pose& a, const Pose& b, double factor) {
  Pose r;
  for (int k = 0; k < 3; ++k) r[k] = a[k] + (b[k] - a[k]) * factor;
  const double one = 1.0 - std::numeric_limits<double>::epsilon();
  const double d = (a[4] * b[4] + a[5] * b[5]) + (a[6] * b[6] + a[3] * b[3]);  // coeffs (x, y, z, w): pairwise sum
  const double abs_d = std::abs(d);
  double s0, s1;
  if (abs_d >= one) {
    s0 = 1.0 - factor;
    s1 = factor;
  } else {
    const double theta = std::acos(abs_d), sin_theta = std::sin(theta);
    s0 = std::sin((1.0 - factor) * theta) / sin_theta;
    s1 = std::sin(factor * theta) / sin_theta;
  }
  if (d < 0.0) s1 = -s1;
  for (int k = 3; k < 7; ++k) r[k] = s0 * a[k] + s1 * b[k];
  return r;
}
inline Pose InterpolateTransform(const Pose& a, const Pose& b, common::Time time_a, common::Time time_b, common::Time time) {
  if (!(time_a <= time && time <= time_b)) throw Error("InterpolateTransform: time outside the two transforms", HG_ERR_TIME);
  const double duration = common::ToSeconds(time_b - time_a);
  return InterpolateTransform(a, b, common::ToSeconds(time - time_a) / duration);  // (:59-62)
}

// transform::TransformInterpolationBuffer (transform_interpolation_buffer.cc:40-140): time-ordered transforms with
// interpolated lookups; LookupUntilDelta is the ADAPTIVE control-point sampling's search.
class TransformInterpolationBuffer {
 public:
  void Push(common::Time time, const Pose& transform) {
    if (!entries_.empty() && time < latest_time()) throw Error("TransformInterpolationBuffer: new transform is older than latest", HG_ERR_INVALID);
    entries_.push_back(Entry{time, transform});
  }
  bool empty() const { return entries_.empty(); }
  common::Time earliest_time() const { return entries_.front().time; }
  common::Time latest_time() const { return entries_.back().time; }
  bool Has(common::Time time) const { return !entries_.empty() && earliest_time() <= time && time <= latest_time(); }
  Pose Lookup(common::Time time) const {
    if (!Has(time)) throw Error("TransformInterpolationBuffer: missing transform", HG_ERR_TIME);
    size_t end = LowerBound(time);
    if (entries_[end].time == time) return entries_[end].transform;
    return InterpolateTransform(entries_[end - 1].transform, entries_[end].transform, entries_[end - 1].time, entries_[end].time, time);
  }
  common::Time LookupUntilDelta(common::Time start_time, double max_translation, double max_rotation, double max_duration,
                                double* translation_ratio, double* rotation_ratio, double* time_ratio) const {
    if (!Has(start_time)) throw Error("TransformInterpolationBuffer: missing transform", HG_ERR_TIME);
    size_t candidate = LowerBound(start_time);
    const Pose start_transform = entries_[candidate].time == start_time
                                     ? entries_[candidate].transform
                                     : InterpolateTransform(entries_[candidate - 1].transform, entries_[candidate].transform,
                                                            entries_[candidate - 1].time, entries_[candidate].time, start_time);
    double target_ratio = 1.0;
    while (candidate + 1 < entries_.size()) {
      const Pose delta = Multiply(Inverse(start_transform), entries_[candidate + 1].transform);
      const double translation_distance = std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
      const double rotation_distance = std::abs(GetAngle(delta));
      const double delta_time = std::abs(common::ToSeconds(entries_[candidate + 1].time - start_time));
      ++candidate;
      *translation_ratio = translation_distance / max_translation;
      *rotation_ratio = rotation_distance / max_rotation;
      *time_ratio = delta_time / max_duration;
      target_ratio = std::max(*translation_ratio, std::max(*rotation_ratio, *time_ratio));
      if (target_ratio >= 1.0) break;
    }
    const common::Duration delta_duration = entries_[candidate].time - start_time;
    const common::Duration corrected = target_ratio > 1.0 ? common::FromSeconds(common::ToSeconds(delta_duration) / target_ratio)
                                                          : delta_duration;
    if (target_ratio > 1.0) {
      *translation_ratio /= target_ratio;
      *rotation_ratio /= target_ratio;
      *time_ratio /= target_ratio;
    }
    return start_time + corrected;
  }

 private:
  struct Entry { common::Time time; Pose transform; };
  size_t LowerBound(common::Time time) const {  // first entry with entry.time >= time
    size_t lo = 0, hi = entries_.size();
    while (lo < hi) {
      const size_t mid = (lo + hi) / 2;
      if (entries_[mid].time < time) lo = mid + 1; else hi = mid;
    }
    return lo;
  }
  std::deque<Entry> entries_;
};
}  // namespace transform

namespace sensor {
struct RangeData {  // sensor/range_data.h:44-57 (misses are unused by the TSDF inserter)
  Point origin{{0.f, 0.f, 0.f}};
  std::vector<Point> returns;
  size_t width = 0;
};
struct TimedPointCloudData {  // sensor/timed_point_cloud_data.h:27-32
  common::Time time = 0;
  Point origin{{0.f, 0.f, 0.f}};
  std::vector<std::array<float, 4>> ranges;  // xyz + time relative to `time`, seconds (TimedRangefinderPoint)
  size_t width = 0;
};
struct OdometryData { common::Time time; Pose pose; };
struct ImuData { common::Time time; std::array<double, 3> linear_acceleration, angular_velocity; };
}  // namespace sensor

class Context {
 public:
  explicit Context(int device = 0, void* stream = nullptr) { Check(hg_ctx_create(device, stream, &ctx_), "hg_ctx_create"); }
  ~Context() { if (ctx_) hg_ctx_destroy(ctx_); }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  hg_ctx* get() const { return ctx_; }
  // tuning / diagnostic switches of the context (keys: include/hg_mi355x.h, hg_ctx_set_option)
  void SetOption(const char* key, long long value) { Check(hg_ctx_set_option(ctx_, key, value), "hg_ctx_set_option"); }
  long long GetOption(const char* key) const {
    long long v = 0;
    Check(hg_ctx_get_option(ctx_, key, &v), "hg_ctx_get_option");
    return v;
  }
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
inline int64_t FromSecondsTicks(double seconds) { return common::FromSeconds(seconds); }  // common/time.cc:30-33
inline double TicksToSeconds(int64_t ticks) { return common::ToSeconds(ticks); }         // :35-38

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
    common::Time time;
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
// control poses and inserted, all on the device (hg_pyramid_insert_unwarped). Times are common::Time ticks, as the
// clouds and the control points carry them (a return on a control point's tick IS inside the window).
// control_poses.front() is optimized_pose.
inline void InsertUnwarped(const std::vector<HybridGridTSDF*>& grids, const hg_insert_opts& inserter,
                           const std::vector<sensor::TimedPointCloudData>& clouds, size_t width,
                           const std::vector<Pose>& control_poses, const std::vector<common::Time>& control_times,
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
    t.time = c.time;
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
    times.push_back(control_times[k]);
  }
  Check(hg_pyramid_insert_unwarped(pyr.data(), opts.data(), static_cast<int>(pyr.size()), points.data(), points.size() / 4,
                                   width, HG_HOST, table.data(), static_cast<int>(table.size()), poses.data(), times.data(),
                                   static_cast<int>(times.size()), submap_from_local ? submap_from_local->data() : nullptr,
                                   insert_mode, nullptr),
        "hg_pyramid_insert_unwarped");
}

// IntegrateImuWithTranslationEuler (imu_integration.h:99-154) with identity calibration: piecewise-constant samples,
// per step delta_rotation *= AngleAxisVectorToRotationQuaternion(w dt) (transform/transform.h:121-135),
// delta_velocity += delta_rotation * (a dt) (Eigen's _transformVector), delta_translation += delta_velocity dt.
// `*it` is the reference's iterator: the last sample at or before start_time on entry (its CHECKs are exceptions here),
// advanced past every sample the integration consumes. Host scalar code, as in the reference. Pinned by the
// reference's own known answers, imu_integration_test.cc:30-120 (tests/test_host_logic.py).
struct IntegrateImuWithTranslationResult {
  std::array<double, 3> delta_translation{{0.0, 0.0, 0.0}};
  std::array<double, 4> delta_rotation{{1.0, 0.0, 0.0, 0.0}};  // (w, x, y, z)
  std::array<double, 3> delta_velocity{{0.0, 0.0, 0.0}};
};
inline IntegrateImuWithTranslationResult IntegrateImuWithTranslationEuler(const std::deque<sensor::ImuData>& imu,
                                                                          common::Time start_time, common::Time end_time,
                                                                          size_t* it) {
  if (!(start_time <= end_time)) throw Error("IntegrateImuWithTranslationEuler: start_time > end_time", HG_ERR_INVALID);
  if (*it >= imu.size() || imu[*it].time > start_time) throw Error("IntegrateImuWithTranslationEuler: no sample at or before start_time", HG_ERR_TIME);
  if (*it + 1 < imu.size() && !(imu[*it + 1].time > start_time)) throw Error("IntegrateImuWithTranslationEuler: iterator is not the last sample before start_time", HG_ERR_TIME);
  IntegrateImuWithTranslationResult result;
  std::array<double, 4>& q = result.delta_rotation;
  common::Time current = start_time;
  while (current < end_time) {
    const common::Time next_imu = *it + 1 < imu.size() ? imu[*it + 1].time : std::numeric_limits<common::Time>::max();
    const common::Time next = std::min(next_imu, end_time);
    const double dt = common::ToSeconds(next - current);
    const double ax = imu[*it].angular_velocity[0] * dt, ay = imu[*it].angular_velocity[1] * dt,
                 az = imu[*it].angular_velocity[2] * dt;
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
    // delta_velocity += delta_rotation * (linear_acceleration * dt): uv = 2 (q.vec x v), v + w uv + q.vec x uv
    const double vx = imu[*it].linear_acceleration[0] * dt, vy = imu[*it].linear_acceleration[1] * dt,
                 vz = imu[*it].linear_acceleration[2] * dt;
    double ux = q[2] * vz - q[3] * vy, uy = q[3] * vx - q[1] * vz, uz = q[1] * vy - q[2] * vx;
    ux += ux; uy += uy; uz += uz;
    result.delta_velocity[0] += vx + q[0] * ux + (q[2] * uz - q[3] * uy);
    result.delta_velocity[1] += vy + q[0] * uy + (q[3] * ux - q[1] * uz);
    result.delta_velocity[2] += vz + q[0] * uz + (q[1] * uy - q[2] * ux);
    for (int k = 0; k < 3; ++k) result.delta_translation[k] += result.delta_velocity[k] * dt;
    current = next;
    if (current == next_imu) ++*it;
  }
  return result;
}

// Pre-integrated rotation between two control points, the only part of the pre-integration result
// PredictionImuPreintegrationCostFunctor reads (prediction_imu_preintegration_cost_functor.h:81-84).
// Returns (w, x, y, z). `imu` is ordered by time; samples before `start` other than the last one are
// ignored, and without a sample at or before `start` the first sample is held (the window's first control point may
// precede the first IMU sample; the reference CHECKs there).
inline std::array<double, 4> IntegrateImuDeltaRotation(const std::deque<sensor::ImuData>& imu, common::Time start,
                                                       common::Time end) {
  if (imu.empty() || !(start < end)) return std::array<double, 4>{{1.0, 0.0, 0.0, 0.0}};
  size_t it = 0;
  while (it + 1 < imu.size() && imu[it + 1].time <= start) ++it;
  if (imu[it].time > start) {  // (it == 0) hold the first sample from `start` on
    std::deque<sensor::ImuData> held = imu;
    held[0].time = start;
    return IntegrateImuWithTranslationEuler(held, start, end, &it).delta_rotation;
  }
  return IntegrateImuWithTranslationEuler(imu, start, end, &it).delta_rotation;
}

// A SIMPLIFIED sliding-window driver (one control point per scan; the reference's own shape is
// OptimizingLocalTrajectoryBuilder below): a sliding window of control points is re-optimised on every scan — AddRangeData (:188-264) queues the
// cloud and adds a control point, MaybeOptimize (:1114-1413) builds one problem over the window (first
// state constant :1268-1275, one TSDF block per scan :323-511, odometry blocks between neighbours
// :1009-1074), solves it on the device, and the scans that leave the window are inserted into the map at
// their optimised poses (:1332-1404). One control point per scan (the reference spaces them by
// ct_window_rate and interpolates; that is hg_problem_add_block's pose_b / interpolation_ratio).
class SlidingWindowTrajectoryBuilder {
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

  SlidingWindowTrajectoryBuilder(Context* ctx, const Options& options) : ctx_(ctx), options_(options) {
    for (float r : options.resolutions)
      grids_.emplace_back(new HybridGridTSDF(ctx, r, options.relative_truncation_distance, options.maximum_weight,
                                             options.max_blocks));
    Check(hg_problem_create(ctx->get(), &problem_), "hg_problem_create");
    hg_solver_default_opts(&solver_);
    solver_.max_num_iterations = options.max_num_iterations;
  }
  ~SlidingWindowTrajectoryBuilder() { if (problem_) hg_problem_destroy(problem_); }

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
      for (int k = 0; k < 3; ++k) cp.velocity[k] = (cp.pose[k] - window_.back().pose[k]) / common::ToSeconds(cp.time - window_.back().time);
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
                                       options_.imu_rotation_weight, common::ToSeconds(window_[i].time - window_[i - 1].time), dq.data()),
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
    common::Time time = 0;
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


// proto::TSDFRangeDataInserterOptions3D of the low-resolution inserter (trajectory_builder_3d.lua:94-109)
inline hg_insert_opts DefaultLowResolutionTSDFInserterOptions() {
  hg_insert_opts o = DefaultTSDFInserterOptions();
  o.min_range = 1.0; o.max_range = 60.0; o.insertion_ratio = 0.1;
  o.normal_computation_horizontal_stride = 20; o.normal_computation_vertical_stride = 4;
  return o;
}

// =====================================================================================================
// OptimizingLocalTrajectoryBuilder in the reference's own shape
// (mapping/internal/3d/optimizing_local_trajectory_builder.cc, options of
// configuration_files/trajectory_builder_3d.lua:18-31,56-60,120-146 with grid_type = "TSDF"):
//   * control points are NOT the scans: CONSTANT sampling places one every ct_window_rate behind the odometry
//     (:1169-1176), SYNCED_WITH_RANGE_DATA one per cloud (:1178-1187), ADAPTIVE by odometry motion (:1189-1232);
//     a new control point is predicted from the odometry delta (PredictStateOdom :1596-1656);
//   * every cloud is bracketed by the control points around its time and matched with an interpolation factor:
//     a single-pose block when it sits exactly on one, a two-pose block otherwise (:323-364,:392-502), against the
//     matching submap's two grids -- both clouds on the pyramid (use_multi_resolution_matching) or, the default, the
//     high-resolution cloud on the high-resolution grid AND the low-resolution cloud on the low-resolution grid;
//   * IMU pre-integration blocks with velocity states (:928-1000), odometry blocks from interpolated odometry
//     lookups at the control points' times with the adaptive weights (:1009-1058);
//   * the solve runs in the matching submap's frame (:1248,:1290), first control point constant (:1268-1275);
//   * the clouds that fall out of the window are moved to the tracking frame of the front control point with the
//     pose interpolated at their time (:1382-1404) -- or return by return (use_per_point_unwarping, :1331-1379,
//     on the device) -- and inserted into the active submaps (:1437-1440, ActiveSubmaps3D above).
// Outside (host-side and not on the TSDF path): PoseExtrapolator / ImuTracker (the initial
// gravity orientation is an option here), IMU calibration, the RK4 integrator (an external library; Euler as
// IntegrateImuWithTranslationEuler), the rotational scan matcher histogram, the debug logger.
// =====================================================================================================
struct State {  // mapping/internal/3d/state.h
  std::array<double, 3> translation{{0, 0, 0}};
  std::array<double, 4> rotation{{1, 0, 0, 0}};  // w x y z
  std::array<double, 3> velocity{{0, 0, 0}};
  Pose ToRigid() const { return Pose{{translation[0], translation[1], translation[2], rotation[0], rotation[1], rotation[2], rotation[3]}}; }
};

class OptimizingLocalTrajectoryBuilder {
 public:
  enum ControlPointSampling { CONSTANT = 0, SYNCED_WITH_RANGE_DATA = 1, ADAPTIVE = 2 };
  struct AdaptiveVoxelFilterOptions { float max_length, min_num_points, max_range; };
  struct Options {
    // TRAJECTORY_BUILDER_3D (trajectory_builder_3d.lua:18-31,56-60)
    float min_range = 1.f, max_range = 60.f;
    int num_accumulated_range_data = 1;
    float voxel_filter_size = 0.15f;
    AdaptiveVoxelFilterOptions high_resolution_adaptive_voxel_filter{2.f, 150.f, 15.f};
    AdaptiveVoxelFilterOptions low_resolution_adaptive_voxel_filter{4.f, 200.f, 60.f};
    int max_num_iterations = 12;  // ceres_scan_matcher.ceres_solver_options
    double motion_filter_max_time_seconds = 0.5, motion_filter_max_distance_meters = 0.1, motion_filter_max_angle_radians = 0.004;
    ActiveSubmaps3D::Options submaps;  // grid_type = "TSDF"; the low-resolution inserter defaults are set below
    // optimizing_local_trajectory_builder (:120-146)
    double high_resolution_grid_weight = 1, low_resolution_grid_weight = 1;
    double velocity_weight = 1, translation_weight = 1, rotation_weight = 1;
    double odometry_translation_weight = 1, odometry_rotation_weight = 1;
    double ct_window_horizon = 0.9, ct_window_rate = 0.1;
    double initialization_duration = 3.0;
    bool use_adaptive_odometry_weights = true;
    bool use_per_point_unwarping = false;
    bool use_multi_resolution_matching = false;
    int num_points_per_subdivision = 4;
    ControlPointSampling control_point_sampling = CONSTANT;
    double sampling_max_delta_translation = 0.2, sampling_max_delta_rotation = 0.1;
    double sampling_min_delta_time = 0.025, sampling_max_delta_time = 0.25;
    bool velocity_in_state = true;
    double odometry_translation_normalization = 2.0e-2, odometry_rotation_normalization = 1.0e-1;
    // initialize_map_orientation_with_imu: the first control point's orientation is what the pose extrapolator's
    // EstimateGravityOrientation returns (:271-276); the extrapolator stays with the caller, who hands it in
    std::array<double, 4> initial_orientation{{1, 0, 0, 0}};
    int insert_mode = HG_INSERT_EXACT;
    Options() { submaps.low_resolution_inserter = DefaultLowResolutionTSDFInserterOptions(); }
  };
  struct ControlPoint {
    common::Time time;
    State state;
    double dT, dR, dt;  // sampling ratios of the ADAPTIVE mode (diagnostic, as in the reference)
  };
  struct InsertionResult {
    std::vector<Point> high_resolution_point_cloud, low_resolution_point_cloud;  // TrajectoryNode::Data (tracking frame)
    std::vector<std::shared_ptr<Submap3D>> insertion_submaps;
  };
  struct MatchingResult {
    common::Time time;
    Pose local_pose;
    sensor::RangeData range_data_in_local;
    std::unique_ptr<const InsertionResult> insertion_result;
  };

  OptimizingLocalTrajectoryBuilder(Context* ctx, const Options& options)
      : ctx_(ctx), options_(options), active_submaps_(ctx, options.submaps),
        ct_window_horizon_(common::FromSeconds(options.ct_window_horizon)),
        ct_window_rate_(common::FromSeconds(options.ct_window_rate)),
        initialization_duration_(common::FromSeconds(options.initialization_duration)) {
    Check(hg_problem_create(ctx->get(), &problem_), "hg_problem_create");
    hg_solver_default_opts(&solver_);
    solver_.max_num_iterations = options.max_num_iterations;
  }
  ~OptimizingLocalTrajectoryBuilder() { if (problem_) hg_problem_destroy(problem_); }
  OptimizingLocalTrajectoryBuilder(const OptimizingLocalTrajectoryBuilder&) = delete;
  OptimizingLocalTrajectoryBuilder& operator=(const OptimizingLocalTrajectoryBuilder&) = delete;

  void AddImuData(const sensor::ImuData& imu_data) {  // (:153-168)
    if (!have_imu_) {
      initial_data_time_ = imu_data.time;
      have_imu_ = true;
    }
    imu_data_.push_back(imu_data);
  }
  void AddOdometryData(const sensor::OdometryData& odometry_data) {  // (:170-186)
    if (!have_imu_) return;                                          // "IMU not yet initialized."
    if (!imu_data_.empty() && imu_data_.front().time >= odometry_data.time) return;  // dropped to maintain IMU consistency
    odometer_data_.push_back(odometry_data);
  }
  void SetMapUpdateEnabled(bool enabled) { map_update_enabled_ = enabled; }
  void UseScanMatching(bool use) { use_scan_matching_ = use; }

  std::unique_ptr<MatchingResult> AddRangeData(const std::string& /*sensor_id*/,
                                               const sensor::TimedPointCloudData& range_data_in_tracking) {  // (:188-264)
    if (range_data_in_tracking.ranges.empty()) throw Error("AddRangeData: empty cloud", HG_ERR_INVALID);  // CHECK_GT(size, 0)
    if (!have_imu_ || odometer_data_.empty()) return nullptr;
    PointCloudSet set;
    set.time = range_data_in_tracking.time;
    set.origin = range_data_in_tracking.origin;
    set.original_cloud = range_data_in_tracking.ranges;
    set.width = range_data_in_tracking.width;
    set.min_point_timestamp = std::numeric_limits<float>::max();
    set.max_point_timestamp = std::numeric_limits<float>::min();  // (sic: the smallest positive float, :213)
    for (const auto& hit : range_data_in_tracking.ranges) {
      if (std::isnan(hit[0]) || std::isnan(hit[1]) || std::isnan(hit[2])) continue;
      const float dx = hit[0] - set.origin[0], dy = hit[1] - set.origin[1], dz = hit[2] - set.origin[2];
      const float range = std::sqrt(dx * dx + (dy * dy + dz * dz));  // Eigen's 3-term reduction x0 + (x1 + x2)
      if (range >= options_.min_range && range <= options_.max_range) {
        set.points.push_back(hit);
        if (hit[3] > set.max_point_timestamp) set.max_point_timestamp = hit[3];
        if (hit[3] < set.min_point_timestamp) set.min_point_timestamp = hit[3];
      }
    }
    if (initial_data_time_ > set.StartTime()) return nullptr;           // "Not enough data, skipping this cloud."
    if (odometer_data_.front().time > set.StartTime()) return nullptr;  // "Not enough odom data, ..."
    AdaptiveVoxelFilterOptions high = options_.high_resolution_adaptive_voxel_filter;
    high.min_num_points = high.min_num_points / options_.num_accumulated_range_data;
    AdaptiveFilter(high, set.points, &set.high_resolution_filtered_points, &set.high_resolution_filtered_times);
    AdaptiveVoxelFilterOptions low = options_.low_resolution_adaptive_voxel_filter;
    low.min_num_points = low.min_num_points / options_.num_accumulated_range_data;
    AdaptiveFilter(low, set.points, &set.low_resolution_filtered_points, &set.low_resolution_filtered_times);
    point_cloud_data_.push_back(std::move(set));
    return MaybeOptimize(range_data_in_tracking.time);
  }

  // ---- what the parity tests and a curious host look at ----
  const std::deque<ControlPoint>& control_points() const { return control_points_; }
  size_t num_queued_clouds() const { return point_cloud_data_.size(); }
  const hg_solver_summary& last_summary() const { return last_summary_; }
  int num_optimizations() const { return num_optimizations_; }
  int num_insertions() const { return num_insertions_; }
  // the TSDF blocks of the last solve: {cloud size, control point a, control point b or -1, interpolation factor, level set}
  struct BlockInfo { size_t points; int pose_a, pose_b; double factor; int grid; };  // grid: 0 high, 1 low, 2 pyramid
  const std::vector<BlockInfo>& last_blocks() const { return last_blocks_; }
  int last_num_residuals() const { return last_num_residuals_; }  // of the last solve: TSDF returns + 9 per IMU + 6 per odometry block
  int last_odometry_blocks() const { return last_odometry_blocks_; }
  int last_imu_blocks() const { return last_imu_blocks_; }
  const ActiveSubmaps3D& active_submaps() const { return active_submaps_; }

 private:
  struct PointCloudSet {  // (optimizing_local_trajectory_builder.h:96-116)
    common::Time time;
    Point origin;
    std::vector<std::array<float, 4>> points, original_cloud;
    std::vector<Point> high_resolution_filtered_points, low_resolution_filtered_points;  // positions (what the matching reads)
    std::vector<float> high_resolution_filtered_times, low_resolution_filtered_times;     // their TimedRangefinderPoint::time
    size_t width;
    float min_point_timestamp, max_point_timestamp;
    common::Time StartTime() const { return time + common::FromSeconds(min_point_timestamp); }
    common::Time EndTime() const { return time + common::FromSeconds(max_point_timestamp); }
  };

  // sensor::AdaptiveVoxelFilter(options).Filter(TimedPointCloud) on the device
  void AdaptiveFilter(const AdaptiveVoxelFilterOptions& o, const std::vector<std::array<float, 4>>& cloud,
                      std::vector<Point>* xyz, std::vector<float>* times) const {
    xyz->clear();
    times->clear();
    if (cloud.empty()) return;
    std::vector<uint32_t> keep(cloud.size());
    size_t n = 0;
    Check(hg_adaptive_voxel_filter(ctx_->get(), o.max_length, o.min_num_points, o.max_range, cloud[0].data(), cloud.size(), 4,
                                   HG_HOST, keep.data(), &n), "hg_adaptive_voxel_filter");
    xyz->resize(n);
    times->resize(n);
    for (size_t i = 0; i < n; ++i) {
      (*xyz)[i] = Point{{cloud[keep[i]][0], cloud[keep[i]][1], cloud[keep[i]][2]}};
      (*times)[i] = cloud[keep[i]][3];
    }
  }

  transform::TransformInterpolationBuffer OdometryBuffer() const {
    transform::TransformInterpolationBuffer buffer;
    for (const auto& o : odometer_data_) buffer.Push(o.time, o.pose);
    return buffer;
  }

  // PredictState = PredictStateOdom (:1516-1521, :1596-1656): the odometry delta between the two times, applied as the
  // reference applies it (delta = current^-1 * previous, sic)
  State PredictStateOdom(const State& start_state, common::Time start_time, common::Time end_time) const {
    {  // (:1599-1603: the walk back through the IMU queue only CHECKs that a sample at or before start_time exists)
      size_t it = imu_data_.size() - 1;
      while (imu_data_[it].time > start_time) {
        if (it == 0) throw Error("PredictStateOdom: no IMU sample at or before the start time", HG_ERR_TIME);
        --it;
      }
    }
    const transform::TransformInterpolationBuffer buffer = OdometryBuffer();
    const common::Time earliest = buffer.earliest_time(), latest = buffer.latest_time();
    auto lookup = [&](common::Time t) {
      if (buffer.Has(t)) return buffer.Lookup(t);
      return t < earliest ? buffer.Lookup(earliest) : buffer.Lookup(latest);
    };
    const Pose previous = lookup(start_time), current = lookup(end_time);
    const Pose delta = transform::Multiply(transform::Inverse(current), previous);
    const double delta_time_seconds = common::ToSeconds(end_time - start_time);
    State out;
    for (int k = 0; k < 3; ++k) {
      out.translation[k] = start_state.translation[k] + delta[k];
      out.velocity[k] = (1.0 / delta_time_seconds) * delta[k];
    }
    // start_rotation * delta_pose.rotation(): a plain quaternion product (no normalisation)
    const double w = start_state.rotation[0], x = start_state.rotation[1], y = start_state.rotation[2], z = start_state.rotation[3];
    out.rotation = {{w * delta[3] - x * delta[4] - y * delta[5] - z * delta[6], w * delta[4] + x * delta[3] + y * delta[6] - z * delta[5],
                     w * delta[5] + y * delta[3] + z * delta[4] - x * delta[6], w * delta[6] + z * delta[3] + x * delta[5] - y * delta[4]}};
    return out;
  }

  void AddControlPoint(common::Time t, double dT = 0.0, double dR = 0.0, double dt = 0.0) {  // (:266-321)
    ControlPoint cp{t, State(), dT, dR, dt};
    if (control_points_.empty()) {
      cp.state.rotation = options_.initial_orientation;
    } else if (active_submaps_.submaps().empty()) {
      cp.state = control_points_.back().state;
    } else {
      cp.state = PredictStateOdom(control_points_.back().state, control_points_.back().time, t);
    }
    control_points_.push_back(cp);
  }

  void TransformStates(const Pose& transform) {  // (:1097-1111)
    const std::array<double, 4> q{{transform[3], transform[4], transform[5], transform[6]}};
    for (ControlPoint& cp : control_points_) {
      const Pose np = transform::Multiply(transform, cp.state.ToRigid());
      const std::array<double, 3> nv = transform::Rotate(q, cp.state.velocity);
      cp.state.translation = {{np[0], np[1], np[2]}};
      cp.state.rotation = {{np[3], np[4], np[5], np[6]}};
      cp.state.velocity = nv;
    }
  }

  void RemoveObsoleteSensorData() {  // (:1076-1095)
    if (control_points_.empty()) return;
    while (!point_cloud_data_.empty() && control_points_.size() > 1 &&
           ct_window_horizon_ < control_points_.back().time - control_points_.front().time &&
           control_points_[1].time < point_cloud_data_.front().StartTime())
      control_points_.pop_front();
    while (imu_data_.size() > 1 && imu_data_[1].time <= control_points_.front().time) imu_data_.pop_front();
    while (odometer_data_.size() > 1 && odometer_data_[1].time <= control_points_.front().time) odometer_data_.pop_front();
  }

  // ---- the residual blocks of one solve ----
  void AddScanBlock(const std::vector<Point>& cloud, hg_grid* const* grids, int levels, bool multi_res, double weight,
                    int a, int b, double factor, int grid_tag) {
    Check(hg_problem_add_block(problem_, cloud[0].data(), cloud.size(), HG_HOST, grids, levels, multi_res ? 1 : 0,
                               weight / std::sqrt(static_cast<double>(cloud.size())), a, b, factor), "hg_problem_add_block");
    last_blocks_.push_back(BlockInfo{cloud.size(), a, b, factor, grid_tag});
  }
  void AddPerScanMatchingResiduals(Submap3D* matching_submap) {  // (:323-511)
    hg_grid* high = matching_submap->high_resolution_hybrid_grid().get();
    hg_grid* low = matching_submap->low_resolution_hybrid_grid().get();
    hg_grid* pyramid[2] = {high, low};
    size_t next = 0;
    for (const PointCloudSet& set : point_cloud_data_) {
      if (set.time > control_points_.back().time) break;
      while (control_points_[next].time <= set.time) {
        if (next + 1 == control_points_.size()) break;
        ++next;
      }
      if (next == 0 || !(control_points_[next - 1].time <= set.time && control_points_[next].time >= set.time))
        throw Error("AddPerScanMatchingResiduals: a cloud lies outside the control points", HG_ERR_TIME);  // CHECKs :335-337
      const int a = static_cast<int>(next) - 1, b = static_cast<int>(next);
      const double duration = common::ToSeconds(control_points_[next].time - control_points_[next - 1].time);
      const double factor = common::ToSeconds(set.time - control_points_[next - 1].time) / duration;
      if (options_.use_multi_resolution_matching) {
        if (options_.high_resolution_grid_weight > 0.0 && !set.high_resolution_filtered_points.empty()) {
          if (factor == 0.0 || factor == 1.0)
            AddScanBlock(set.high_resolution_filtered_points, pyramid, 2, true, options_.high_resolution_grid_weight,
                         factor == 0.0 ? a : b, -1, 0.0, 2);
          else
            AddScanBlock(set.high_resolution_filtered_points, pyramid, 2, true, options_.high_resolution_grid_weight, a, b, factor, 2);
        }
        continue;
      }
      const bool on_prev = control_points_[next - 1].time == set.time, on_next = control_points_[next].time == set.time;
      if (options_.high_resolution_grid_weight > 0.0 && !set.high_resolution_filtered_points.empty()) {
        if (on_prev) AddScanBlock(set.high_resolution_filtered_points, &high, 1, false, options_.high_resolution_grid_weight, a, -1, 0.0, 0);
        else if (on_next) AddScanBlock(set.high_resolution_filtered_points, &high, 1, false, options_.high_resolution_grid_weight, b, -1, 0.0, 0);
        else AddScanBlock(set.high_resolution_filtered_points, &high, 1, false, options_.high_resolution_grid_weight, a, b, factor, 0);
      }
      if (options_.low_resolution_grid_weight > 0.0 && !set.low_resolution_filtered_points.empty()) {
        if (on_prev) AddScanBlock(set.low_resolution_filtered_points, &low, 1, false, options_.low_resolution_grid_weight, a, -1, 0.0, 1);
        else if (on_next) AddScanBlock(set.low_resolution_filtered_points, &low, 1, false, options_.low_resolution_grid_weight, b, -1, 0.0, 1);
        else AddScanBlock(set.low_resolution_filtered_points, &low, 1, false, options_.low_resolution_grid_weight, a, b, factor, 1);
      }
    }
  }
  // use_per_point_unwarping (:513-683): the high-resolution cloud in subdivisions of num_points_per_subdivision returns,
  // each interpolated at its own time (on the pyramid, or on the high-resolution grid); without the pyramid the
  // low-resolution cloud as well, return by return on the low-resolution grid (:622-683)
  void AddPerPointResiduals(Submap3D* matching_submap) {
    hg_grid* high = matching_submap->high_resolution_hybrid_grid().get();
    hg_grid* low = matching_submap->low_resolution_hybrid_grid().get();
    hg_grid* pyramid[2] = {high, low};
    std::vector<int> ids(control_points_.size());
    std::vector<int64_t> times(control_points_.size());
    for (size_t i = 0; i < control_points_.size(); ++i) { ids[i] = static_cast<int>(i); times[i] = control_points_[i].time; }
    for (const PointCloudSet& set : point_cloud_data_) {
      if (set.high_resolution_filtered_points.empty()) continue;
      if (options_.use_multi_resolution_matching)
        scan_matching::AddPerPointMatchingResiduals(problem_, ids, times, set.time, set.high_resolution_filtered_points,
                                                    set.high_resolution_filtered_times, pyramid, 2, true,
                                                    options_.high_resolution_grid_weight, options_.num_points_per_subdivision);
      else
        scan_matching::AddPerPointMatchingResiduals(problem_, ids, times, set.time, set.high_resolution_filtered_points,
                                                    set.high_resolution_filtered_times, &high, 1, false,
                                                    options_.high_resolution_grid_weight, options_.num_points_per_subdivision);
    }
    if (!options_.use_multi_resolution_matching && options_.low_resolution_grid_weight > 0)
      for (const PointCloudSet& set : point_cloud_data_) {
        if (set.low_resolution_filtered_points.empty()) continue;
        scan_matching::AddPerPointMatchingResiduals(problem_, ids, times, set.time, set.low_resolution_filtered_points,
                                                    set.low_resolution_filtered_times, &low, 1, false,
                                                    options_.low_resolution_grid_weight, 1);
      }
  }
  void AddIMUResiduals() {  // (:928-1007), imu_cost_term = PREINTEGRATION
    last_imu_blocks_ = 0;
    if (options_.translation_weight == 0.0 && options_.velocity_weight == 0.0 && options_.rotation_weight == 0.0) return;
    if (!options_.velocity_in_state) throw Error("IMU residuals require velocity_in_state", HG_ERR_INVALID);  // CHECK :937
    {  // (:971-975) a sample at or before the first control point must exist
      size_t it = imu_data_.size() - 1;
      while (imu_data_[it].time > control_points_.front().time) {
        if (it == 0) throw Error("AddIMUResiduals: no IMU sample at or before the first control point", HG_ERR_TIME);
        --it;
      }
    }
    for (size_t i = 1; i < control_points_.size(); ++i) {
      const std::array<double, 4> dq = IntegrateImuDeltaRotation(imu_data_, control_points_[i - 1].time, control_points_[i].time);
      Check(hg_problem_add_imu_block(problem_, static_cast<int>(i) - 1, static_cast<int>(i), options_.translation_weight,
                                     options_.velocity_weight, options_.rotation_weight,
                                     common::ToSeconds(control_points_[i].time - control_points_[i - 1].time), dq.data()),
            "hg_problem_add_imu_block");
      ++last_imu_blocks_;
    }
  }
  void AddOdometryResiduals() {  // (:1009-1074)
    last_odometry_blocks_ = 0;
    if (odometer_data_.size() <= 1) return;
    const transform::TransformInterpolationBuffer buffer = OdometryBuffer();
    for (size_t i = 1; i < control_points_.size(); ++i) {
      if (!(buffer.earliest_time() <= control_points_[i - 1].time && control_points_[i].time <= buffer.latest_time())) continue;
      const Pose previous = buffer.Lookup(control_points_[i - 1].time), current = buffer.Lookup(control_points_[i].time);
      const Pose delta = transform::Multiply(transform::Inverse(current), previous);
      const double delta_time = common::ToSeconds(control_points_[i].time - control_points_[i - 1].time);
      double translation_weight = options_.odometry_translation_weight, rotation_weight = options_.odometry_rotation_weight;
      if (options_.use_adaptive_odometry_weights) {
        const double translation_distance = std::abs(std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]));
        const double rotation_distance = std::abs(transform::GetAngle(delta));  // angularDistance(Identity)
        translation_weight = options_.odometry_translation_weight /
                             std::sqrt(translation_distance + options_.odometry_translation_normalization * delta_time);
        rotation_weight = options_.odometry_rotation_weight /
                          std::sqrt(rotation_distance + options_.odometry_rotation_normalization * delta_time);
      }
      Check(hg_problem_add_odometry_block(problem_, static_cast<int>(i) - 1, static_cast<int>(i), translation_weight,
                                          rotation_weight, delta.data()), "hg_problem_add_odometry_block");
      ++last_odometry_blocks_;
    }
  }

  std::unique_ptr<MatchingResult> MaybeOptimize(common::Time time) {  // (:1113-1413)
    if (time - initial_data_time_ < initialization_duration_) return nullptr;
    if (odometer_data_.size() < 2) return nullptr;
    if (control_points_.empty()) AddControlPoint(std::max(initial_data_time_, odometer_data_.front().time));
    bool added_control_point = false;
    switch (options_.control_point_sampling) {
      case CONSTANT:
        while (control_points_.back().time + ct_window_rate_ < odometer_data_.back().time) {
          AddControlPoint(control_points_.back().time + ct_window_rate_);
          added_control_point = true;
        }
        break;
      case SYNCED_WITH_RANGE_DATA:
        for (const PointCloudSet& set : point_cloud_data_)
          if (control_points_.back().time < set.time && set.time < imu_data_.back().time) {
            AddControlPoint(set.time);
            added_control_point = true;
          }
        break;
      case ADAPTIVE: {
        const transform::TransformInterpolationBuffer buffer = OdometryBuffer();
        common::Time candidate_time = control_points_.back().time;
        while (candidate_time < buffer.latest_time()) {
          double translation_ratio = 0.0, rotation_ratio = 0.0, time_ratio = 0.0;
          candidate_time = buffer.LookupUntilDelta(control_points_.back().time, options_.sampling_max_delta_translation,
                                                   options_.sampling_max_delta_rotation, options_.sampling_max_delta_time,
                                                   &translation_ratio, &rotation_ratio, &time_ratio);
          if (common::ToSeconds(candidate_time - control_points_.back().time) < options_.sampling_min_delta_time)
            candidate_time = control_points_.back().time + common::FromSeconds(options_.sampling_min_delta_time);
          if (candidate_time < buffer.latest_time()) {
            AddControlPoint(candidate_time, translation_ratio, rotation_ratio, time_ratio);
            added_control_point = true;
          }
        }
        break;
      }
    }
    if (!added_control_point) return nullptr;

    if (!active_submaps_.submaps().empty()) {
      Submap3D* matching_submap = active_submaps_.submaps().front().get();
      const Pose inv = transform::Inverse(matching_submap->local_pose());
      // "We assume the map is always aligned with the direction of gravity" (:1243-1244, isApprox(Identity, 1e-8))
      if (!(std::abs(std::abs(inv[3]) - 1.0) < 1e-8)) throw Error("matching submap is rotated against the local frame", HG_ERR_UNSUPPORTED);
      TransformStates(inv);
      Check(hg_problem_reset(problem_), "hg_problem_reset");
      last_blocks_.clear();
      for (size_t i = 0; i < control_points_.size(); ++i) {
        const bool constant = i == 0;  // (:1268-1275)
        const Pose p = control_points_[i].state.ToRigid();
        Check(hg_problem_add_pose(problem_, p.data(), constant), "hg_problem_add_pose");
        if (options_.velocity_in_state)
          Check(hg_problem_set_velocity(problem_, static_cast<int>(i), control_points_[i].state.velocity.data(), constant),
                "hg_problem_set_velocity");
      }
      if (use_scan_matching_) {
        if (options_.use_per_point_unwarping) AddPerPointResiduals(matching_submap);
        else AddPerScanMatchingResiduals(matching_submap);
      }
      AddIMUResiduals();
      AddOdometryResiduals();
      last_num_residuals_ = hg_problem_num_residuals(problem_);
      Check(hg_problem_solve(problem_, &solver_, &last_summary_), "hg_problem_solve");
      ++num_optimizations_;
      for (size_t i = 0; i < control_points_.size(); ++i) {
        Pose p;
        Check(hg_problem_get_pose(problem_, static_cast<int>(i), p.data()), "hg_problem_get_pose");
        control_points_[i].state.translation = {{p[0], p[1], p[2]}};
        control_points_[i].state.rotation = {{p[3], p[4], p[5], p[6]}};
        if (options_.velocity_in_state)
          Check(hg_problem_get_velocity(problem_, static_cast<int>(i), control_points_[i].state.velocity.data()), "hg_problem_get_velocity");
      }
      TransformStates(matching_submap->local_pose());
    }

    const Pose optimized_pose = control_points_.front().state.ToRigid();
    const common::Time time_optimized_pose = control_points_.front().time;
    const Pose optimized_inverse = transform::Inverse(optimized_pose);
    sensor::RangeData accumulated;  // accumulated_range_data_in_tracking
    accumulated.width = point_cloud_data_.front().width;
    if (active_submaps_.submaps().empty()) {
      // "To initialize the empty map we add all available range data assuming zero motion." (:1301-1330; the clouds stay queued)
      size_t it = 0;
      for (const PointCloudSet& set : point_cloud_data_) {
        if (!(set.time < control_points_.back().time)) continue;
        while (control_points_[it].time <= set.time) ++it;
        if (it == 0 || it >= control_points_.size()) throw Error("initialisation: a cloud lies outside the control points", HG_ERR_TIME);
        const Pose cloud_pose = transform::InterpolateTransform(control_points_[it - 1].state.ToRigid(), control_points_[it].state.ToRigid(),
                                                                control_points_[it - 1].time, control_points_[it].time, set.time);
        const std::array<float, 7> tf = transform::ToFloat(transform::Multiply(optimized_inverse, cloud_pose));
        for (const auto& p : set.original_cloud) accumulated.returns.push_back(transform::TransformPoint(tf, Point{{p[0], p[1], p[2]}}));
        accumulated.origin = transform::TransformPoint(tf, set.origin);
      }
    } else if (options_.use_per_point_unwarping) {
      UnwarpLeavingClouds(&accumulated);  // (:1331-1379) accumulated_range_data_in_tracking, then the common path below
    } else {
      if (!(control_points_.front().time <= point_cloud_data_.front().time))
        throw Error("the oldest cloud is older than the window", HG_ERR_TIME);  // CHECK :1381
      while (!point_cloud_data_.empty() &&
             ct_window_horizon_ - ct_window_rate_ < control_points_.back().time - point_cloud_data_.front().time) {
        while (control_points_[1].time < point_cloud_data_.front().time) control_points_.pop_front();
        const Pose cloud_pose = transform::InterpolateTransform(control_points_[0].state.ToRigid(), control_points_[1].state.ToRigid(),
                                                                control_points_[0].time, control_points_[1].time,
                                                                point_cloud_data_.front().time);
        const std::array<float, 7> tf = transform::ToFloat(transform::Multiply(optimized_inverse, cloud_pose));
        for (const auto& p : point_cloud_data_.front().points)
          accumulated.returns.push_back(transform::TransformPoint(tf, Point{{p[0], p[1], p[2]}}));
        accumulated.origin = transform::TransformPoint(tf, point_cloud_data_.front().origin);
        point_cloud_data_.pop_front();
      }
    }
    RemoveObsoleteSensorData();
    return AddAccumulatedRangeData(time_optimized_pose, optimized_pose, accumulated);
  }

  // (:1415-1470) voxel filter and adaptive filters for the trajectory node, range data to the local frame, insertion
  std::unique_ptr<MatchingResult> AddAccumulatedRangeData(common::Time time, const Pose& optimized_pose,
                                                          const sensor::RangeData& range_data_in_tracking) {
    if (range_data_in_tracking.returns.empty()) return nullptr;
    const std::vector<Point> filtered = sensor::VoxelFilter(ctx_, options_.voxel_filter_size).Filter(range_data_in_tracking.returns);
    if (filtered.empty()) return nullptr;
    std::unique_ptr<MatchingResult> result(new MatchingResult);
    result->time = time;
    result->local_pose = optimized_pose;
    const std::array<float, 7> to_local = transform::ToFloat(optimized_pose);  // TransformTimedRangeData(range_data_in_tracking, ...): unfiltered
    result->range_data_in_local.width = range_data_in_tracking.width;
    result->range_data_in_local.origin = transform::TransformPoint(to_local, range_data_in_tracking.origin);
    result->range_data_in_local.returns.reserve(range_data_in_tracking.returns.size());
    for (const Point& p : range_data_in_tracking.returns) result->range_data_in_local.returns.push_back(transform::TransformPoint(to_local, p));
    std::unique_ptr<InsertionResult> insertion(new InsertionResult);
    {
      const AdaptiveVoxelFilterOptions& h = options_.high_resolution_adaptive_voxel_filter;
      insertion->high_resolution_point_cloud = sensor::AdaptiveVoxelFilter(ctx_, h.max_length, h.min_num_points, h.max_range).Filter(filtered);
      if (insertion->high_resolution_point_cloud.empty()) return nullptr;  // "Dropped empty high resolution point cloud data."
      const AdaptiveVoxelFilterOptions& l = options_.low_resolution_adaptive_voxel_filter;
      insertion->low_resolution_point_cloud = sensor::AdaptiveVoxelFilter(ctx_, l.max_length, l.min_num_points, l.max_range).Filter(filtered);
      if (insertion->low_resolution_point_cloud.empty()) return nullptr;
    }
    // InsertIntoSubmap (:1472-1514)
    if (MotionFilterIsSimilar(time, optimized_pose)) return result;  // insertion_result stays null
    // gravity_alignment = optimized_pose.rotation(); local_from_gravity_aligned = pose.rotation() * gravity_alignment.inverse()
    const Pose rot{{0, 0, 0, optimized_pose[3], optimized_pose[4], optimized_pose[5], optimized_pose[6]}};
    const Pose lfg = transform::Multiply(rot, transform::Inverse(rot));
    if (map_update_enabled_) active_submaps_.InsertData(result->range_data_in_local, {{lfg[3], lfg[4], lfg[5], lfg[6]}});
    ++num_insertions_;
    insertion->insertion_submaps = active_submaps_.submaps();
    result->insertion_result = std::move(insertion);
    return result;
  }

  // use_per_point_unwarping (:1331-1379): the clouds that leave the window are unwarped return by return on the device
  // (hg_unwarp_range_data, frame 0: every return into the tracking frame of the first control point with the pose
  // interpolated at its own time; NaN returns kept; the origin from the first unwarped return's transform) into
  // accumulated_range_data_in_tracking. What follows is AddAccumulatedRangeData for every mode alike (this
  // path used to insert on its own and skipped the filters, the node's clouds and the map-update / motion-filter
  // rules of :1415-1514).
  void UnwarpLeavingClouds(sensor::RangeData* accumulated) {
    if (!(control_points_.front().time <= point_cloud_data_.front().StartTime()))
      throw Error("the oldest cloud starts before the window", HG_ERR_TIME);  // CHECK :1333-1334
    std::vector<hg_timed_cloud> table;
    std::vector<float> points;
    while (!point_cloud_data_.empty() && ct_window_horizon_ < control_points_.back().time - point_cloud_data_.front().StartTime() &&
           control_points_.back().time > point_cloud_data_.front().EndTime()) {
      const PointCloudSet& set = point_cloud_data_.front();
      hg_timed_cloud t{};
      t.time = set.time;
      t.begin = points.size() / 4;
      t.count = set.points.size();
      for (int k = 0; k < 3; ++k) t.origin[k] = set.origin[k];
      table.push_back(t);
      for (const auto& p : set.points) points.insert(points.end(), p.begin(), p.end());
      point_cloud_data_.pop_front();
    }
    if (points.empty()) return;
    std::vector<double> poses;
    std::vector<int64_t> times;
    for (const ControlPoint& cp : control_points_) {
      const Pose p = cp.state.ToRigid();
      poses.insert(poses.end(), p.begin(), p.end());
      times.push_back(cp.time);
    }
    accumulated->returns.resize(points.size() / 4);
    Check(hg_unwarp_range_data(ctx_->get(), points.data(), points.size() / 4, HG_HOST, table.data(), static_cast<int>(table.size()),
                               poses.data(), times.data(), static_cast<int>(times.size()), 0, nullptr,
                               accumulated->returns[0].data(), accumulated->origin.data()),
          "hg_unwarp_range_data");
  }

  bool MotionFilterIsSimilar(common::Time time, const Pose& pose) {  // (mapping/internal/motion_filter.cc:40-58)
    ++motion_total_;
    if (motion_total_ > 1 && time - motion_last_time_ <= common::FromSeconds(options_.motion_filter_max_time_seconds)) {
      const double dx = pose[0] - motion_last_pose_[0], dy = pose[1] - motion_last_pose_[1], dz = pose[2] - motion_last_pose_[2];
      if (std::sqrt(dx * dx + dy * dy + dz * dz) <= options_.motion_filter_max_distance_meters &&
          transform::GetAngle(transform::Multiply(transform::Inverse(pose), motion_last_pose_)) <= options_.motion_filter_max_angle_radians)
        return true;
    }
    motion_last_time_ = time;
    motion_last_pose_ = pose;
    return false;
  }

  Context* ctx_;
  Options options_;
  ActiveSubmaps3D active_submaps_;
  common::Duration ct_window_horizon_, ct_window_rate_, initialization_duration_;
  hg_problem* problem_ = nullptr;
  hg_solver_opts solver_;
  hg_solver_summary last_summary_{};
  common::Time initial_data_time_ = 0;
  bool have_imu_ = false, map_update_enabled_ = true, use_scan_matching_ = true;
  std::deque<sensor::ImuData> imu_data_;
  std::deque<sensor::OdometryData> odometer_data_;
  std::deque<PointCloudSet> point_cloud_data_;
  std::deque<ControlPoint> control_points_;
  std::vector<BlockInfo> last_blocks_;
  int last_odometry_blocks_ = 0, last_imu_blocks_ = 0, last_num_residuals_ = 0, num_optimizations_ = 0, num_insertions_ = 0;
  int motion_total_ = 0;
  common::Time motion_last_time_ = 0;
  Pose motion_last_pose_{{0, 0, 0, 1, 0, 0, 0}};
};

}  // namespace mapping
}  // namespace hg_amd
#endif  // HG_ADAPTER_H_
