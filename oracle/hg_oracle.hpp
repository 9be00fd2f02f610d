// hg_oracle.hpp — CPU ORACLE. TEST INFRASTRUCTURE ONLY.
//
// Dependency-free C++17 restatement of the HectorGrapher (Cartographer fork)
// local-SLAM hot path: TSDF voxel codec, sparse hybrid grid, TSDF range-data
// inserter, trilinear (multi-resolution) TSDF lookup, the TSDF space cost
// functions (evaluated with a forward-mode Jet, as Ceres autodiff does) and a
// Ceres-1.13-style Levenberg-Marquardt trust-region loop.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use
// this code, and only as the checker / timed CPU baseline. The product path
// (hectorgrapher_amd/csrc) never links or calls it.
//
// PARITY STATUS: the codec, LUT formula, grid indexing and pose interpolation
// are pinned by the reference's own known-answer tests (see tests/). The
// inserter, the cost functions and the solver are NOT pinned by any reference
// test ("parity unpinned" for those) — they follow the cited
// reference lines, and for the third-party parts (Eigen 3.3 reductions /
// slerp / quaternion rotation, Ceres 1.13 Jet arithmetic, trust-region
// minimizer and LM strategy) the published algorithm, restated from memory of
// those versions because neither library is in /root/reference or the image.
//
// All "ref:" citations are relative to /root/reference/cartographer/.
#pragma once

#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <unordered_set>
#include <vector>

namespace hgo {

using uint16 = std::uint16_t;

// ref: common/port.h:40 (std::lround: half away from zero)
inline int RoundToInt(const float x) { return static_cast<int>(std::lround(x)); }
inline int RoundToInt(const double x) { return static_cast<int>(std::lround(x)); }

// ref: common/math.h:31-40
template <typename T>
inline T Clamp(const T value, const T min, const T max) {
  if (value > max) return max;
  if (value < min) return min;
  return value;
}

// ---------------------------------------------------------------------------
// Minimal fixed-size vectors with Eigen-3.3 evaluation order.
// Eigen's fixed-size reduction (Redux.h, redux_novec_unroller<.., 0, 3>) sums
// three terms as x0 + (x1 + x2); four terms as (x0 + x1) + (x2 + x3).
// ---------------------------------------------------------------------------
template <typename T>
struct Vec3 {
  T x, y, z;
  Vec3() : x(T(0)), y(T(0)), z(T(0)) {}
  Vec3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
  Vec3 operator+(const Vec3& o) const { return {x + o.x, y + o.y, z + o.z}; }
  Vec3 operator-(const Vec3& o) const { return {x - o.x, y - o.y, z - o.z}; }
  Vec3 operator-() const { return {-x, -y, -z}; }
  T dot(const Vec3& o) const { return x * o.x + (y * o.y + z * o.z); }
  T squaredNorm() const { return x * x + (y * y + z * z); }
  Vec3 cross(const Vec3& o) const {
    return {y * o.z - z * o.y, z * o.x - x * o.z, x * o.y - y * o.x};
  }
};
template <typename T, typename S>
inline Vec3<T> operator*(const S& s, const Vec3<T>& v) {
  return {s * v.x, s * v.y, s * v.z};
}
template <typename T, typename S>
inline Vec3<T> operator*(const Vec3<T>& v, const S& s) {
  return {v.x * s, v.y * s, v.z * s};
}
using Vec3f = Vec3<float>;
using Vec3d = Vec3<double>;
struct Vec3i {
  int x, y, z;
};
inline float Norm(const Vec3f& v) { return std::sqrt(v.squaredNorm()); }
inline bool HasNaN(const Vec3f& v) {
  return std::isnan(v.x) || std::isnan(v.y) || std::isnan(v.z);
}

// Quaternion, storage order (w, x, y, z) as passed to Ceres by the reference.
template <typename T>
struct Quat {
  T w, x, y, z;
  Vec3<T> vec() const { return {x, y, z}; }
};

// Eigen 3.3 QuaternionBase::_transformVector (generic path).
template <typename T>
inline Vec3<T> Rotate(const Quat<T>& q, const Vec3<T>& v) {
  Vec3<T> uv = q.vec().cross(v);
  uv = uv + uv;
  return v + q.w * uv + q.vec().cross(uv);
}

// Eigen 3.3 quaternion product (generic path), a*b.
template <typename T>
inline Quat<T> QuatMul(const Quat<T>& a, const Quat<T>& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z,
          a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z,
          a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}

// ref: transform/rigid_transform.h:117-197
template <typename T>
struct Rigid3 {
  Vec3<T> t;
  Quat<T> q;
  Vec3<T> operator*(const Vec3<T>& p) const { return Rotate(q, p) + t; }
};

// ---------------------------------------------------------------------------
// Voxel codec. ref: mapping/value_conversion_tables.cc:26-68,
// mapping/2d/tsd_value_converter.{h:39-96,cc:22-32}
// ---------------------------------------------------------------------------
constexpr uint16 kUpdateMarker = 1u << 15;

inline float SlowValueToBoundedFloat(const uint16 value,
                                     const uint16 unknown_value,
                                     const float unknown_result,
                                     const float lower_bound,
                                     const float upper_bound) {
  if (value == unknown_value) return unknown_result;
  const float kScale = (upper_bound - lower_bound) / 32766.f;
  return value * kScale + (lower_bound - kScale);
}

inline std::vector<float> PrecomputeValueToBoundedFloat(
    const uint16 unknown_value, const float unknown_result,
    const float lower_bound, const float upper_bound) {
  std::vector<float> result;
  const size_t num_values = std::numeric_limits<uint16>::max() + 1;
  result.reserve(num_values);
  for (size_t value = 0; value != num_values; ++value) {
    result.push_back(SlowValueToBoundedFloat(
        static_cast<uint16>(value) & ~kUpdateMarker, unknown_value,
        unknown_result, lower_bound, upper_bound));
  }
  return result;
}

class TSDValueConverter {
 public:
  TSDValueConverter(float max_tsd, float max_weight)
      : max_tsd_(max_tsd),
        min_tsd_(-max_tsd),
        max_weight_(max_weight),
        tsd_resolution_(32766.f / (max_tsd_ - min_tsd_)),
        weight_resolution_(32766.f / (max_weight_ - min_weight_)),
        value_to_tsd_(
            PrecomputeValueToBoundedFloat(0, min_tsd_, min_tsd_, max_tsd_)),
        value_to_weight_(PrecomputeValueToBoundedFloat(0, min_weight_,
                                                       min_weight_, max_weight)) {}

  uint16 TSDToValue(const float tsd) const {
    const int value =
        RoundToInt((Clamp(tsd, min_tsd_, max_tsd_) - min_tsd_) * tsd_resolution_) + 1;
    return static_cast<uint16>(value);
  }
  uint16 WeightToValue(const float weight) const {
    const int value = RoundToInt((Clamp(weight, min_weight_, max_weight_) - min_weight_) *
                                 weight_resolution_) +
                      1;
    return static_cast<uint16>(value);
  }
  float ValueToTSD(const uint16 value) const { return value_to_tsd_[value]; }
  float ValueToWeight(const uint16 value) const { return value_to_weight_[value]; }
  float getMaxTSD() const { return max_tsd_; }
  float getMinTSD() const { return min_tsd_; }
  float getMaxWeight() const { return max_weight_; }
  float getMinWeight() const { return min_weight_; }

 private:
  float max_tsd_;
  float min_tsd_;
  float max_weight_;
  float tsd_resolution_;
  float weight_resolution_;
  static constexpr float min_weight_ = 0.f;
  std::vector<float> value_to_tsd_;
  std::vector<float> value_to_weight_;
};

// ---------------------------------------------------------------------------
// Sparse grid. ref: mapping/3d/hybrid_grid_base.h:40-52,69-141,144-246,251-407
// ---------------------------------------------------------------------------
struct TSDFVoxel {
  uint16 discrete_tsd = 0;
  uint16 discrete_weight = 0;
  bool operator==(const TSDFVoxel& r) const {
    return discrete_tsd == r.discrete_tsd && discrete_weight == r.discrete_weight;
  }
};

inline int ToFlatIndex(const Vec3i& index, const int bits) {
  return (((index.z << bits) + index.y) << bits) + index.x;
}
inline Vec3i To3DIndex(const int index, const int bits) {
  const int mask = (1 << bits) - 1;
  return {index & mask, (index >> bits) & mask, (index >> bits) >> bits};
}

struct FlatGrid {  // FlatGrid<TSDFVoxel, 3>
  static constexpr int kBits = 3;
  static int grid_size() { return 1 << kBits; }
  std::array<TSDFVoxel, 512> cells{};
  TSDFVoxel value(const Vec3i& i) const { return cells[ToFlatIndex(i, kBits)]; }
  TSDFVoxel* mutable_value(const Vec3i& i) { return &cells[ToFlatIndex(i, kBits)]; }
};

struct NestedGrid {  // NestedGrid<FlatGrid<TSDFVoxel,3>,3>
  static constexpr int kBits = 3;
  static int grid_size() { return FlatGrid::grid_size() << kBits; }
  std::array<std::unique_ptr<FlatGrid>, 512> meta_cells;
  TSDFVoxel value(const Vec3i& index) const {
    const Vec3i meta{index.x / 8, index.y / 8, index.z / 8};
    const FlatGrid* const cell = meta_cells[ToFlatIndex(meta, kBits)].get();
    if (cell == nullptr) return TSDFVoxel();
    return cell->value({index.x - meta.x * 8, index.y - meta.y * 8, index.z - meta.z * 8});
  }
  TSDFVoxel* mutable_value(const Vec3i& index) {
    const Vec3i meta{index.x / 8, index.y / 8, index.z / 8};
    std::unique_ptr<FlatGrid>& cell = meta_cells[ToFlatIndex(meta, kBits)];
    if (cell == nullptr) cell = std::make_unique<FlatGrid>();
    return cell->mutable_value(
        {index.x - meta.x * 8, index.y - meta.y * 8, index.z - meta.z * 8});
  }
};

class DynamicGrid {
 public:
  DynamicGrid() : bits_(1), meta_cells_(8) {}
  int grid_size() const { return NestedGrid::grid_size() << bits_; }
  int bits() const { return bits_; }

  TSDFVoxel value(const Vec3i& index) const {
    const int half = grid_size() >> 1;
    const Vec3i s{index.x + half, index.y + half, index.z + half};
    const unsigned gs = static_cast<unsigned>(grid_size());
    if (static_cast<unsigned>(s.x) >= gs || static_cast<unsigned>(s.y) >= gs ||
        static_cast<unsigned>(s.z) >= gs) {
      return TSDFVoxel();
    }
    const Vec3i meta{s.x / 64, s.y / 64, s.z / 64};
    const NestedGrid* const cell = meta_cells_[ToFlatIndex(meta, bits_)].get();
    if (cell == nullptr) return TSDFVoxel();
    return cell->value({s.x - meta.x * 64, s.y - meta.y * 64, s.z - meta.z * 64});
  }

  TSDFVoxel* mutable_value(const Vec3i& index) {
    const int half = grid_size() >> 1;
    const Vec3i s{index.x + half, index.y + half, index.z + half};
    const unsigned gs = static_cast<unsigned>(grid_size());
    if (static_cast<unsigned>(s.x) >= gs || static_cast<unsigned>(s.y) >= gs ||
        static_cast<unsigned>(s.z) >= gs) {
      if (!Grow()) return nullptr;
      return mutable_value(index);
    }
    const Vec3i meta{s.x / 64, s.y / 64, s.z / 64};
    std::unique_ptr<NestedGrid>& cell = meta_cells_[ToFlatIndex(meta, bits_)];
    if (cell == nullptr) cell = std::make_unique<NestedGrid>();
    return cell->mutable_value({s.x - meta.x * 64, s.y - meta.y * 64, s.z - meta.z * 64});
  }

  // Visits every non-default voxel in the reference's iterator order
  // (hybrid_grid_base.h:304-372): meta cells z-major, leaves z-major, voxels
  // z-major (x fastest).
  template <typename F>
  void ForEach(F&& f) const {
    const int half = grid_size() >> 1;
    for (size_t m = 0; m < meta_cells_.size(); ++m) {
      const NestedGrid* nested = meta_cells_[m].get();
      if (nested == nullptr) continue;
      const Vec3i mi = To3DIndex(static_cast<int>(m), bits_);
      for (int l = 0; l < 512; ++l) {
        const FlatGrid* flat = nested->meta_cells[l].get();
        if (flat == nullptr) continue;
        const Vec3i li = To3DIndex(l, 3);
        for (int c = 0; c < 512; ++c) {
          const TSDFVoxel& v = flat->cells[c];
          if (v == TSDFVoxel()) continue;
          const Vec3i ci = To3DIndex(c, 3);
          f(Vec3i{mi.x * 64 + li.x * 8 + ci.x - half, mi.y * 64 + li.y * 8 + ci.y - half,
                  mi.z * 64 + li.z * 8 + ci.z - half},
            v);
        }
      }
    }
  }

 private:
  bool Grow() {
    const int new_bits = bits_ + 1;
    if (new_bits > 8) return false;  // reference: CHECK_LE(new_bits, 8)
    std::vector<std::unique_ptr<NestedGrid>> new_meta(8 * meta_cells_.size());
    for (int z = 0; z != (1 << bits_); ++z)
      for (int y = 0; y != (1 << bits_); ++y)
        for (int x = 0; x != (1 << bits_); ++x) {
          const int o = 1 << (bits_ - 1);
          new_meta[ToFlatIndex({x + o, y + o, z + o}, new_bits)] =
              std::move(meta_cells_[ToFlatIndex({x, y, z}, bits_)]);
        }
    meta_cells_ = std::move(new_meta);
    bits_ = new_bits;
    return true;
  }
  int bits_;
  std::vector<std::unique_ptr<NestedGrid>> meta_cells_;
};

// ref: mapping/3d/hybrid_grid_tsdf.h:41-134, hybrid_grid_base.h:410-458
class HybridGridTSDF {
 public:
  HybridGridTSDF(const float resolution, float relative_truncation_distance,
                 float max_weight)
      : resolution_(resolution),
        value_converter_(relative_truncation_distance * resolution, max_weight) {}

  float resolution() const { return resolution_; }
  Vec3i GetCellIndex(const Vec3f& point) const {
    return {RoundToInt(point.x / resolution_), RoundToInt(point.y / resolution_),
            RoundToInt(point.z / resolution_)};
  }
  Vec3f GetCenterOfCell(const Vec3i& index) const {
    return {static_cast<float>(index.x) * resolution_,
            static_cast<float>(index.y) * resolution_,
            static_cast<float>(index.z) * resolution_};
  }
  bool SetCell(const Vec3i& index, const float tsd, const float weight) {
    TSDFVoxel* v = grid_.mutable_value(index);
    if (v == nullptr) return false;
    *v = {static_cast<uint16>(value_converter_.TSDToValue(tsd) + kUpdateMarker),
          value_converter_.WeightToValue(weight)};
    return true;
  }
  void FinishUpdate() {}  // update_indices_ is never filled in the reference (:96-102)
  float GetTSD(const Vec3i& index) const {
    return value_converter_.ValueToTSD(grid_.value(index).discrete_tsd);
  }
  float GetWeight(const Vec3i& index) const {
    return value_converter_.ValueToWeight(grid_.value(index).discrete_weight);
  }
  bool IsKnown(const Vec3i& index) const {
    return grid_.value(index).discrete_weight != 0;
  }
  TSDFVoxel RawValue(const Vec3i& index) const { return grid_.value(index); }
  const TSDValueConverter& ValueConverter() const { return value_converter_; }
  const DynamicGrid& grid() const { return grid_; }

 private:
  const float resolution_;
  TSDValueConverter value_converter_;
  DynamicGrid grid_;
};

// ---------------------------------------------------------------------------
// TSDF inserter. ref: mapping/3d/tsdf_range_data_inserter_3d.cc
// ---------------------------------------------------------------------------
struct InserterOptions {  // proto/3d/tsdf_range_data_inserter_options_3d.proto
  double relative_truncation_distance = 2.5;
  double maximum_weight = 1000.;
  int num_free_space_voxels = 0;
  bool project_sdf_distance_to_scan_normal = false;
  double weight_function_epsilon = 1.0;
  double weight_function_sigma = 4.;
  double min_range = 0.4;
  double max_range = 15.0;
  double insertion_ratio = 1.0;
  int normal_computation_method = 1;  // 0 PCL, 1 CLOUD_STRUCTURE, 2 OPEN3D, 3 TRIANGLE_FILL_IN
  int normal_computation_horizontal_stride = 5;
  int normal_computation_vertical_stride = 1;
};

struct InsertStats {
  uint64_t num_hits = 0;     // returns that reached the ray walk (N_in)
  uint64_t num_updates = 0;  // UpdateCell calls with non-zero weight (U)
};

class TSDFRangeDataInserter3D {
 public:
  explicit TSDFRangeDataInserter3D(const InserterOptions& options) : options_(options) {}

  // ref :725-737
  void UpdateCell(const Vec3i& cell, float update_sdf, float update_weight,
                  HybridGridTSDF* tsdf, InsertStats* stats) const {
    if (update_weight == 0.f) return;
    const float old_weight = tsdf->GetWeight(cell);
    const float old_sdf = tsdf->GetTSD(cell);
    float updated_weight = old_weight + update_weight;
    float updated_sdf = (old_sdf * old_weight + update_sdf * update_weight) / updated_weight;
    updated_weight = std::min(updated_weight, static_cast<float>(options_.maximum_weight));
    tsdf->SetCell(cell, updated_sdf, updated_weight);
    if (stats) ++stats->num_updates;
  }

  // ref :294-342 (default raycast branch; the Amanatides-Woo branch is dead
  // code behind `use_default_raycast = true`).
  void InsertHit(const Vec3f& hit, const Vec3f& origin, HybridGridTSDF* tsdf,
                 InsertStats* stats) const {
    const Vec3f ray = hit - origin;
    const float range = Norm(ray);
    const float truncation_distance =
        static_cast<float>(options_.relative_truncation_distance * tsdf->resolution());
    if (range < truncation_distance) return;
    const float truncation_ratio = truncation_distance / range;
    const bool update_free_space = options_.num_free_space_voxels > 0;
    const Vec3f ray_begin =
        update_free_space ? origin : origin + (1.0f - truncation_ratio) * ray;
    const Vec3f ray_end = origin + (1.0f + truncation_ratio) * ray;

    const Vec3i begin_cell = tsdf->GetCellIndex(ray_begin);
    const Vec3i end_cell = tsdf->GetCellIndex(ray_end);
    const Vec3i delta{end_cell.x - begin_cell.x, end_cell.y - begin_cell.y,
                      end_cell.z - begin_cell.z};
    const int num_samples =
        std::max(std::abs(delta.x), std::max(std::abs(delta.y), std::abs(delta.z)));
    // The reference divides by float(num_samples); num_samples == 0 (NaN cell)
    // cannot occur for range >= truncation distance with 2*tau >= 2.9 cells.
    // Guarded here (documented divergence: such a return is skipped).
    if (num_samples == 0) return;
    if (stats) ++stats->num_hits;
    for (int position = 0; position <= num_samples; ++position) {
      const float fp = static_cast<float>(position);
      const float fn = static_cast<float>(num_samples);
      // (delta.cast<float>() * float(position) / float(num_samples)).round().cast<int>()
      const Vec3i update_cell_index{
          begin_cell.x + static_cast<int>(std::round(static_cast<float>(delta.x) * fp / fn)),
          begin_cell.y + static_cast<int>(std::round(static_cast<float>(delta.y) * fp / fn)),
          begin_cell.z + static_cast<int>(std::round(static_cast<float>(delta.z) * fp / fn))};
      const Vec3f cell_center = tsdf->GetCenterOfCell(update_cell_index);
      const float distance_cell_to_origin = Norm(cell_center - origin);
      float update_tsd = range - distance_cell_to_origin;
      update_tsd = Clamp(update_tsd, -truncation_distance, truncation_distance);
      float update_weight = 1.0;
      const float epsilon = static_cast<float>(options_.weight_function_epsilon);
      const float normalized_update_tsd = update_tsd / truncation_distance;
      if (normalized_update_tsd < -epsilon) {
        const float sigma = static_cast<float>(options_.weight_function_sigma);
        update_weight = static_cast<float>(
            std::exp(-sigma * std::pow(-normalized_update_tsd - epsilon, 2)));
      }
      UpdateCell(update_cell_index, update_tsd, update_weight, tsdf, stats);
    }
  }

  // ref :197-241 (default raycast branch)
  void InsertHitWithNormal(const Vec3f& hit, const Vec3f& origin, const Vec3f& normal,
                           HybridGridTSDF* tsdf, InsertStats* stats) const {
    const Vec3f ray = hit - origin;
    const float range = Norm(ray);
    const float truncation_distance =
        static_cast<float>(options_.relative_truncation_distance * tsdf->resolution());
    if (range < truncation_distance) return;
    float normal_direction = 1.f;
    if (normal.dot(ray) > 0.f) normal_direction = -1.f;
    const Vec3f ray_begin = hit - (normal_direction * truncation_distance) * normal;
    const Vec3f ray_end = hit + (normal_direction * truncation_distance) * normal;
    const Vec3i begin_cell = tsdf->GetCellIndex(ray_begin);
    const Vec3i end_cell = tsdf->GetCellIndex(ray_end);
    const Vec3i delta{end_cell.x - begin_cell.x, end_cell.y - begin_cell.y,
                      end_cell.z - begin_cell.z};
    const int num_samples =
        std::max(std::abs(delta.x), std::max(std::abs(delta.y), std::abs(delta.z)));
    if (num_samples == 0) return;
    if (stats) ++stats->num_hits;
    for (int position = 0; position <= num_samples; ++position) {
      const float fp = static_cast<float>(position);
      const float fn = static_cast<float>(num_samples);
      const Vec3i update_cell_index{
          begin_cell.x + static_cast<int>(std::round(static_cast<float>(delta.x) * fp / fn)),
          begin_cell.y + static_cast<int>(std::round(static_cast<float>(delta.y) * fp / fn)),
          begin_cell.z + static_cast<int>(std::round(static_cast<float>(delta.z) * fp / fn))};
      const Vec3f cell_center = tsdf->GetCenterOfCell(update_cell_index);
      float update_tsd = normal_direction * (cell_center - hit).dot(normal);
      update_tsd = Clamp(update_tsd, -truncation_distance, truncation_distance);
      UpdateCell(update_cell_index, update_tsd, 1.0f, tsdf, stats);
    }
  }

  // ref :395-404,698-723 (default branch). `returns` is N x 3 floats already
  // in the grid (submap) frame.
  void Insert(const Vec3f& origin, const float* returns, size_t n, size_t width,
              HybridGridTSDF* tsdf, InsertStats* stats) const {
    if (options_.project_sdf_distance_to_scan_normal) {
      InsertWithCloudStructureNormals(origin, returns, n, width, tsdf, stats);
      return;
    }
    size_t num_inserted_points = 0;
    size_t num_omitted_points = 0;
    for (size_t i = 0; i < n; ++i) {
      const Vec3f hit{returns[3 * i], returns[3 * i + 1], returns[3 * i + 2]};
      if (double(num_inserted_points) <=
          options_.insertion_ratio * double(num_inserted_points + num_omitted_points)) {
        num_inserted_points++;
      } else {
        num_omitted_points++;
        continue;
      }
      if (HasNaN(hit)) continue;
      const float r0 = Norm(hit - origin);
      if (r0 < options_.min_range) continue;
      if (r0 > options_.max_range) continue;
      InsertHit(hit, origin, tsdf, stats);
    }
    tsdf->FinishUpdate();
  }

  // ref :502-607 (CLOUD_STRUCTURE). The reference's `point_idx - offset < 0`
  // on size_t (:548) is always false and then reads out of bounds for the
  // first `vertical_stride` points. That undefined
  // behaviour is NOT replicated: the lower-vertical search treats a wrapped
  // index as "no neighbour" (offset reduced), which is what the horizontal
  // search in the same function does (:574 `point_idx < offset`).
  void InsertWithCloudStructureNormals(const Vec3f& origin, const float* returns, size_t n,
                                       size_t width, HybridGridTSDF* tsdf,
                                       InsertStats* stats) const {
    auto P = [&](size_t i) { return Vec3f{returns[3 * i], returns[3 * i + 1], returns[3 * i + 2]}; };
    size_t num_inserted_points = 0, num_omitted_points = 0;
    const size_t vertical_stride = options_.normal_computation_vertical_stride;
    const size_t horizontal_stride = options_.normal_computation_horizontal_stride * width;
    for (size_t point_idx = 0; point_idx < n; ++point_idx) {
      if (double(num_inserted_points) <=
          options_.insertion_ratio * double(num_inserted_points + num_omitted_points)) {
        num_inserted_points++;
      } else {
        num_omitted_points++;
        continue;
      }
      const Vec3f p0 = P(point_idx);
      if (HasNaN(p0)) continue;
      const float r0 = Norm(p0 - origin);
      if (r0 < options_.min_range || r0 > options_.max_range) continue;
      const float max_range_delta = 1.f * tsdf->resolution() / 0.05f;
      auto bad = [&](size_t j) {
        return HasNaN(P(j)) || (std::abs(r0 - Norm(P(j) - origin)) > max_range_delta);
      };
      size_t offset = vertical_stride;
      while (offset > 0 && ((point_idx + offset >= n) || bad(point_idx + offset))) --offset;
      const size_t i_vertical_upper = point_idx + offset;
      offset = vertical_stride;
      while (offset > 0 && ((point_idx < offset) || bad(point_idx - offset))) --offset;
      const size_t i_vertical_lower = point_idx - offset;
      if (i_vertical_lower == i_vertical_upper) continue;
      if (width == 0) continue;
      offset = horizontal_stride;
      while (offset > 0 && (point_idx + offset >= n || bad(point_idx + offset))) offset -= width;
      const size_t i_horizontal_upper = point_idx + offset;
      offset = horizontal_stride;
      while (offset > 0 && (point_idx < offset || bad(point_idx - offset))) offset -= width;
      const size_t i_horizontal_lower = point_idx - offset;
      if (i_horizontal_lower == i_horizontal_upper) continue;
      const Vec3f dh = P(i_horizontal_lower) - P(i_horizontal_upper);
      const Vec3f dv = P(i_vertical_lower) - P(i_vertical_upper);
      auto is_zero = [](const Vec3f& v) { return v.x == 0.f && v.y == 0.f && v.z == 0.f; };
      if (is_zero(dh) || is_zero(dv)) continue;
      Vec3f normal = dh.cross(dv);
      const float nn = Norm(normal);
      if (nn > 0.f) normal = Vec3f{normal.x / nn, normal.y / nn, normal.z / nn};
      if (is_zero(normal)) continue;
      InsertHitWithNormal(p0, origin, normal, tsdf, stats);
    }
    tsdf->FinishUpdate();
  }

  const InserterOptions& options() const { return options_; }

 private:
  InserterOptions options_;
};

// ref: sensor/range_data.cc:25-39 + transform/rigid_transform.h:193-197 (float).
inline void TransformPointsF(const Rigid3<float>& T, const float* in, size_t n, float* out) {
  for (size_t i = 0; i < n; ++i) {
    const Vec3f p = T * Vec3f{in[3 * i], in[3 * i + 1], in[3 * i + 2]};
    out[3 * i] = p.x;
    out[3 * i + 1] = p.y;
    out[3 * i + 2] = p.z;
  }
}

// ---------------------------------------------------------------------------
// Voxel filters. ref: sensor/internal/voxel_filter.{h,cc}:26-77,
// sensor/internal/adaptive_voxel_filter.h:33-110
// ---------------------------------------------------------------------------
struct CellKey {
  int x, y, z;
  bool operator==(const CellKey& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct CellKeyHash {
  size_t operator()(const CellKey& k) const {
    uint64_t h = static_cast<uint32_t>(k.x) * 0x9E3779B97F4A7C15ull;
    h ^= static_cast<uint32_t>(k.y) * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= static_cast<uint32_t>(k.z) * 0x165667B19E3779F9ull + (h << 6) + (h >> 2);
    return static_cast<size_t>(h);
  }
};

// VoxelFilter::Filter: indices of the first point that falls into each voxel, in input order.
// `stride` floats per point (3 = PointCloud, 4 = TimedPointCloud; time is ignored).
inline std::vector<uint32_t> VoxelFilterIndices(float resolution, const float* pts, size_t n,
                                                int stride, const std::vector<uint32_t>* subset = nullptr) {
  std::unordered_set<CellKey, CellKeyHash> voxel_set;
  std::vector<uint32_t> out;
  const size_t m = subset ? subset->size() : n;
  for (size_t j = 0; j < m; ++j) {
    const uint32_t i = subset ? (*subset)[j] : static_cast<uint32_t>(j);
    const float* p = pts + static_cast<size_t>(i) * stride;
    const CellKey k{RoundToInt(p[0] / resolution), RoundToInt(p[1] / resolution),
                    RoundToInt(p[2] / resolution)};
    if (voxel_set.insert(k).second) out.push_back(i);
  }
  return out;
}

// AdaptiveVoxelFilter::Filter = AdaptivelyVoxelFiltered(options, FilterByMaxRange(cloud, max_range)).
inline std::vector<uint32_t> AdaptiveVoxelFilterIndices(float max_length, float min_num_points,
                                                        float max_range, const float* pts, size_t n,
                                                        int stride) {
  std::vector<uint32_t> in;  // FilterByMaxRange (:33-44)
  for (size_t i = 0; i < n; ++i) {
    const float* p = pts + i * stride;
    if (Norm(Vec3f{p[0], p[1], p[2]}) <= max_range) in.push_back(static_cast<uint32_t>(i));
  }
  if (in.size() <= min_num_points) return in;  // already sparse enough (:50-53)
  std::vector<uint32_t> result = VoxelFilterIndices(max_length, pts, n, stride, &in);
  if (result.size() >= min_num_points) return result;
  for (float high_length = max_length; high_length > 1e-2f * max_length; high_length /= 2.f) {
    float low_length = high_length / 2.f;
    result = VoxelFilterIndices(low_length, pts, n, stride, &in);
    if (result.size() >= min_num_points) {
      while ((high_length - low_length) / low_length > 1e-1f) {
        const float mid_length = (low_length + high_length) / 2.f;
        const std::vector<uint32_t> candidate = VoxelFilterIndices(mid_length, pts, n, stride, &in);
        if (candidate.size() >= min_num_points) {
          low_length = mid_length;
          result = candidate;
        } else {
          high_length = mid_length;
        }
      }
      return result;
    }
  }
  return result;
}

// ---------------------------------------------------------------------------
// Jet: forward-mode dual number with Ceres 1.13 jet.h arithmetic.
// ---------------------------------------------------------------------------
template <int N>
struct Jet {
  double a;
  double v[N];
  Jet() : a(0.0) { for (int i = 0; i < N; ++i) v[i] = 0.0; }
  Jet(double a_) : a(a_) { for (int i = 0; i < N; ++i) v[i] = 0.0; }  // NOLINT
  Jet(double a_, int k) : a(a_) {
    for (int i = 0; i < N; ++i) v[i] = 0.0;
    v[k] = 1.0;
  }
};
template <int N> inline Jet<N> operator-(const Jet<N>& f) {
  Jet<N> r; r.a = -f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r;
}
template <int N> inline Jet<N> operator+(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r; r.a = f.a + g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] + g.v[i]; return r;
}
template <int N> inline Jet<N> operator+(const Jet<N>& f, double s) {
  Jet<N> r = f; r.a = f.a + s; return r;
}
template <int N> inline Jet<N> operator+(double s, const Jet<N>& f) {
  Jet<N> r = f; r.a = f.a + s; return r;
}
template <int N> inline Jet<N> operator-(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r; r.a = f.a - g.a; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] - g.v[i]; return r;
}
template <int N> inline Jet<N> operator-(const Jet<N>& f, double s) {
  Jet<N> r = f; r.a = f.a - s; return r;
}
template <int N> inline Jet<N> operator-(double s, const Jet<N>& f) {
  Jet<N> r; r.a = s - f.a; for (int i = 0; i < N; ++i) r.v[i] = -f.v[i]; return r;
}
template <int N> inline Jet<N> operator*(const Jet<N>& f, const Jet<N>& g) {
  Jet<N> r; r.a = f.a * g.a;
  for (int i = 0; i < N; ++i) r.v[i] = f.a * g.v[i] + f.v[i] * g.a;
  return r;
}
template <int N> inline Jet<N> operator*(const Jet<N>& f, double s) {
  Jet<N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r;
}
template <int N> inline Jet<N> operator*(double s, const Jet<N>& f) {
  Jet<N> r; r.a = f.a * s; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s; return r;
}
template <int N> inline Jet<N> operator/(const Jet<N>& f, const Jet<N>& g) {
  const double g_a_inverse = 1.0 / g.a;
  const double f_a_by_g_a = f.a * g_a_inverse;
  Jet<N> r; r.a = f_a_by_g_a;
  for (int i = 0; i < N; ++i) r.v[i] = (f.v[i] - f_a_by_g_a * g.v[i]) * g_a_inverse;
  return r;
}
template <int N> inline Jet<N> operator/(const Jet<N>& f, double s) {
  const double s_inverse = 1.0 / s;
  Jet<N> r; r.a = f.a * s_inverse; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * s_inverse;
  return r;
}
template <int N> inline Jet<N> operator/(double s, const Jet<N>& g) {
  const double minus_s_g_a_inverse2 = -s / (g.a * g.a);
  Jet<N> r; r.a = s / g.a;
  for (int i = 0; i < N; ++i) r.v[i] = g.v[i] * minus_s_g_a_inverse2;
  return r;
}
template <int N> inline Jet<N>& operator+=(Jet<N>& f, const Jet<N>& g) { f = f + g; return f; }
template <int N> inline bool operator<(const Jet<N>& f, const Jet<N>& g) { return f.a < g.a; }
template <int N> inline bool operator<(const Jet<N>& f, double g) { return f.a < g; }
template <int N> inline bool operator>=(const Jet<N>& f, const Jet<N>& g) { return f.a >= g.a; }
template <int N> inline Jet<N> jsin(const Jet<N>& f) {
  const double c = std::cos(f.a);
  Jet<N> r; r.a = std::sin(f.a); for (int i = 0; i < N; ++i) r.v[i] = c * f.v[i]; return r;
}
template <int N> inline Jet<N> jacos(const Jet<N>& f) {
  const double tmp = -1.0 / std::sqrt(1.0 - f.a * f.a);
  Jet<N> r; r.a = std::acos(f.a); for (int i = 0; i < N; ++i) r.v[i] = tmp * f.v[i]; return r;
}
template <int N> inline Jet<N> jabs(const Jet<N>& f) { return f.a < 0.0 ? -f : f; }
template <int N> inline Jet<N> jasin(const Jet<N>& f) {
  const double tmp = 1.0 / std::sqrt(1.0 - f.a * f.a);
  Jet<N> r; r.a = std::asin(f.a); for (int i = 0; i < N; ++i) r.v[i] = tmp * f.v[i]; return r;
}
template <int N> inline Jet<N> jatan2(const Jet<N>& g, const Jet<N>& f) {  // atan2(g, f)
  const double tmp = 1.0 / (f.a * f.a + g.a * g.a);
  Jet<N> r; r.a = std::atan2(g.a, f.a);
  for (int i = 0; i < N; ++i) r.v[i] = tmp * (-g.a * f.v[i] + f.a * g.v[i]);
  return r;
}
template <int N> inline Jet<N> jsqrt(const Jet<N>& f) {
  const double tmp = std::sqrt(f.a);
  const double two_a_inverse = 1.0 / (2.0 * tmp);
  Jet<N> r; r.a = tmp; for (int i = 0; i < N; ++i) r.v[i] = f.v[i] * two_a_inverse; return r;
}
inline double jasin(double x) { return std::asin(x); }
inline double jatan2(double y, double x) { return std::atan2(y, x); }
inline double jsqrt(double x) { return std::sqrt(x); }
inline double jsin(double x) { return std::sin(x); }
inline double jacos(double x) { return std::acos(x); }
inline double jabs(double x) { return std::fabs(x); }
inline double ScalarPart(double x) { return x; }
template <int N> inline double ScalarPart(const Jet<N>& x) { return x.a; }

// Eigen 3.3 QuaternionBase::slerp (Geometry/Quaternion.h).
template <typename T>
inline Quat<T> Slerp(const Quat<T>& a, const T& t, const Quat<T>& b) {
  const T one = T(1.0 - std::numeric_limits<double>::epsilon());
  // coeffs() is stored (x, y, z, w); four-term fixed-size reduction.
  const T d = (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w);
  const T absD = jabs(d);
  T scale0, scale1;
  if (absD >= one) {
    scale0 = T(1.0) - t;
    scale1 = t;
  } else {
    const T theta = jacos(absD);
    const T sinTheta = jsin(theta);
    scale0 = jsin((T(1.0) - t) * theta) / sinTheta;
    scale1 = jsin((t * theta)) / sinTheta;
  }
  if (d < T(0.0)) scale1 = -scale1;
  return {scale0 * a.w + scale1 * b.w, scale0 * a.x + scale1 * b.x,
          scale0 * a.y + scale1 * b.y, scale0 * a.z + scale1 * b.z};
}

// ref: transform/timestamped_transform.h:41-51
template <typename T>
inline Rigid3<T> InterpolateTransform(const Rigid3<T>& start, const Rigid3<T>& end,
                                      const double factor) {
  const Vec3<T> origin = start.t + (end.t - start.t) * factor;
  const Quat<T> rotation = Slerp(start.q, T(factor), end.q);
  return {origin, rotation};
}

// ---------------------------------------------------------------------------
// Interpolated TSDF. ref: mapping/internal/3d/scan_matching/interpolated_tsdf.h
// and interpolated_multi_resolution_tsdf.h
// ---------------------------------------------------------------------------
// ref interpolated_tsdf.h:30-46 (min_tsd_const = -0.3) and
// interpolated_multi_resolution_tsdf.h:30-46 (min_tsd_const = getMinTSD()).
template <typename T, typename T2>
inline void InterpolateLinear(const double both_invalid_value, const T2& q1, const T2& q2,
                              const double w1, const double w2, const T& normalized_ratio,
                              T& q, double& w) {
  if (w1 == 0.0 && w2 == 0.0) {
    q = T(both_invalid_value);
    w = 0.0;
  } else if (w1 == 0.0) {
    q = T(q2);
    w = w2;
  } else if (w2 == 0.0) {
    q = T(q1);
    w = w1;
  } else {
    q = (q2 - q1) * normalized_ratio + q1;
    w = w1 + w2;
  }
}

// ref interpolated_tsdf.h:176-192
inline Vec3f CenterOfLowerVoxel(const HybridGridTSDF& tsdf, const double x, const double y,
                                const double z) {
  Vec3f center = tsdf.GetCenterOfCell(tsdf.GetCellIndex(
      Vec3f{static_cast<float>(x), static_cast<float>(y), static_cast<float>(z)}));
  if (center.x > x) center.x -= tsdf.resolution();
  if (center.y > y) center.y -= tsdf.resolution();
  if (center.z > z) center.z -= tsdf.resolution();
  return center;
}

struct LookupStats {
  uint64_t levels_probed = 0;  // sum over lookups of pyramid levels visited
  uint64_t lookups = 0;
};

// Shared body of both GetTSD variants for one level. Returns false if the
// level must be skipped (multi-res) — never for single-res.
template <typename T>
inline bool LevelTSD(const HybridGridTSDF& tsdf, bool multi_res, const T& x, const T& y,
                     const T& z, T* out) {
  const Vec3f lower = CenterOfLowerVoxel(tsdf, ScalarPart(x), ScalarPart(y), ScalarPart(z));
  const double x1 = lower.x, y1 = lower.y, z1 = lower.z;
  const double x2 = lower.x + tsdf.resolution();
  const double y2 = lower.y + tsdf.resolution();
  const double z2 = lower.z + tsdf.resolution();
  const Vec3i i1 = tsdf.GetCellIndex(
      Vec3f{static_cast<float>(x1), static_cast<float>(y1), static_cast<float>(z1)});
  auto W = [&](int dx, int dy, int dz) {
    return static_cast<double>(tsdf.GetWeight({i1.x + dx, i1.y + dy, i1.z + dz}));
  };
  auto Q = [&](int dx, int dy, int dz) {
    return static_cast<double>(tsdf.GetTSD({i1.x + dx, i1.y + dy, i1.z + dz}));
  };
  const double w111 = W(0, 0, 0), w112 = W(0, 0, 1), w121 = W(0, 1, 0), w122 = W(0, 1, 1);
  const double w211 = W(1, 0, 0), w212 = W(1, 0, 1), w221 = W(1, 1, 0), w222 = W(1, 1, 1);
  double both_invalid;
  if (multi_res) {
    const int num_invalid = int(w111 == 0.0) + int(w112 == 0.0) + int(w121 == 0.0) +
                            int(w122 == 0.0) + int(w211 == 0.0) + int(w212 == 0.0) +
                            int(w221 == 0.0) + int(w222 == 0.0);
    if (num_invalid > 0) return false;
    both_invalid = tsdf.ValueConverter().getMinTSD();
  } else {
    if (w111 == 0.0 && w112 == 0.0 && w121 == 0.0 && w122 == 0.0 && w211 == 0.0 &&
        w212 == 0.0 && w221 == 0.0 && w222 == 0.0) {
      *out = T(static_cast<double>(tsdf.ValueConverter().getMinTSD()));
      return true;
    }
    both_invalid = -0.3;
  }
  const double q111 = Q(0, 0, 0), q112 = Q(0, 0, 1), q121 = Q(0, 1, 0), q122 = Q(0, 1, 1);
  const double q211 = Q(1, 0, 0), q212 = Q(1, 0, 1), q221 = Q(1, 1, 0), q222 = Q(1, 1, 1);
  const T normalized_x = (x - x1) / (x2 - x1);
  const T normalized_y = (y - y1) / (y2 - y1);
  const T normalized_z = (z - z1) / (z2 - z1);
  T q11, q12, q21, q22, q1, q2, q;
  double w11, w12, w21, w22, w1, w2, w;
  InterpolateLinear(both_invalid, q111, q112, w111, w112, normalized_z, q11, w11);
  InterpolateLinear(both_invalid, q121, q122, w121, w122, normalized_z, q12, w12);
  InterpolateLinear(both_invalid, q211, q212, w211, w212, normalized_z, q21, w21);
  InterpolateLinear(both_invalid, q221, q222, w221, w222, normalized_z, q22, w22);
  InterpolateLinear(both_invalid, q11, q12, w11, w12, normalized_y, q1, w1);
  InterpolateLinear(both_invalid, q21, q22, w21, w22, normalized_y, q2, w2);
  InterpolateLinear(both_invalid, q1, q2, w1, w2, normalized_x, q, w);
  *out = q;
  return true;
}

// ref interpolated_tsdf.h:72-116
template <typename T>
inline T InterpolatedGetTSD(const HybridGridTSDF& tsdf, const T& x, const T& y, const T& z) {
  T out;
  LevelTSD(tsdf, false, x, y, z, &out);
  return out;
}

// ref interpolated_multi_resolution_tsdf.h:83-137
template <typename T>
inline T InterpolatedMultiResGetTSD(const std::vector<const HybridGridTSDF*>& pyramid,
                                    const T& x, const T& y, const T& z,
                                    LookupStats* stats = nullptr) {
  if (stats) ++stats->lookups;
  for (const HybridGridTSDF* tsdf : pyramid) {
    if (stats) ++stats->levels_probed;
    T out;
    if (LevelTSD(*tsdf, true, x, y, z, &out)) return out;
  }
  return T(static_cast<double>(pyramid.front()->ValueConverter().getMinTSD()));
}

// ---------------------------------------------------------------------------
// Problem: poses (t[3], q[4] wxyz) + TSDF scan-matching residual blocks.
// ref: the six cost-function headers (a12/a13) and
// mapping/internal/3d/optimizing_local_trajectory_builder.cc:75-111,323-511,
// 1238-1291 (block wiring, scaling, constant first control point,
// QuaternionParameterization).
// ---------------------------------------------------------------------------
struct PoseBlock {  // one control point: pose (+ optional velocity, state.h:11-31)
  double t[3];
  double q[4];  // w x y z
  bool constant = false;
  double v[3] = {0.0, 0.0, 0.0};
  bool has_velocity = false;
  bool v_constant = false;
};

// Non-TSDF residual blocks of the sliding window (oltb.cc:928-1074).
struct SmallBlock {
  int type = 0;  // 1: RelativeTranslationAndYawCostFunction (6 residuals), 2: PredictionImuPreintegrationCostFunctor (9)
  int a = 0, b = 0;          // previous / next control point
  double w[3] = {0, 0, 0};   // type 1: translation, rotation scaling; type 2: translation, velocity, rotation
  double dt = 0.0;           // type 2: delta_time_seconds
  double delta[7] = {0, 0, 0, 1, 0, 0, 0};  // type 1: delta pose (t, q wxyz); type 2: delta_rotation in [3..6]
};

// ref transform/rigid_transform.h:159-163,184-190 and transform/transform.h:43-96 on generic T.
template <typename T> inline Quat<T> QuatNormalized(const Quat<T>& q) {
  const T n = jsqrt((q.x * q.x + q.y * q.y) + (q.z * q.z + q.w * q.w));
  return {q.w / n, q.x / n, q.y / n, q.z / n};
}
template <typename T> inline Rigid3<T> RigidInverse(const Rigid3<T>& r) {
  const Quat<T> rc{r.q.w, -r.q.x, -r.q.y, -r.q.z};
  return {-Rotate(rc, r.t), rc};
}
template <typename T> inline Rigid3<T> RigidMul(const Rigid3<T>& a, const Rigid3<T>& b) {
  return {Rotate(a.q, b.t) + a.t, QuatNormalized(QuatMul(a.q, b.q))};
}
// ---------------------------------------------------------------------------
// Per-point unwarping of the accumulated range data.
// ref: mapping/internal/3d/optimizing_local_trajectory_builder.cc:1331-1379 (use_per_point_unwarping
// branch of MaybeOptimize), common/time.cc:30-38, transform/timestamped_transform.h:53-65.
// ---------------------------------------------------------------------------
using int64 = std::int64_t;
struct TimedCloud {      // sensor::TimedPointCloudData
  int64 time;            // common::Time in universal ticks (100 ns)
  Vec3f origin;
  const float* points;   // n x 4: position xyz + time [s relative to `time`] (sensor::TimedRangefinderPoint)
  size_t n;
};
struct ControlPoint {
  int64 time;
  Rigid3<double> pose;   // state.ToRigid()
};
struct UnwarpedRangeData {  // accumulated_range_data_in_tracking
  Vec3f origin{0.f, 0.f, 0.f};  // Eigen::Vector3f::Zero() (:1298-1299)
  std::vector<float> returns;   // n x 3
  bool time_in_range = true;    // false where the reference would CHECK-fail (:1355-1359)
};
// common::FromSeconds (time.cc:30-33): duration_cast<Duration>(duration<double>(s)) = (int64)(s * 1e7).
inline int64 FromSeconds(const double seconds) { return static_cast<int64>(seconds * 1e7); }
// common::ToSeconds (time.cc:35-38): duration_cast<duration<double>>(ticks) = ticks / 1e7.
inline double ToSeconds(const int64 ticks) { return static_cast<double>(ticks) / 1e7; }
// transform/timestamped_transform.h:53-65
inline Rigid3<double> InterpolateTransformTimed(const Rigid3<double>& start, const Rigid3<double>& end,
                                                const int64 time_start, const int64 time_end, const int64 time) {
  const double duration = ToSeconds(time_end - time_start);
  const double factor = ToSeconds(time - time_start) / duration;
  return InterpolateTransform(start, end, factor);
}
// The while loop of :1337-1378 over the clouds that leave the window (the caller decides which: the loop
// condition compares window times only). `control_points` are the window's control points after the solve;
// optimized_pose = control_points.front() (:1294-1295).
inline UnwarpedRangeData UnwarpAccumulatedRangeData(const std::vector<ControlPoint>& control_points,
                                                    const std::vector<TimedCloud>& clouds) {
  UnwarpedRangeData out;
  const Rigid3<double> optimized_pose = control_points.front().pose;
  size_t next_control_point = 0;  // iterator into control_points (:1335)
  bool first_point = true;
  const size_t last = control_points.size() - 1;
  for (const TimedCloud& cloud : clouds) {
    for (size_t i = 0; i < cloud.n; ++i) {
      const Vec3f position{cloud.points[4 * i], cloud.points[4 * i + 1], cloud.points[4 * i + 2]};
      const float time = cloud.points[4 * i + 3];
      if (HasNaN(position)) {  // :1342-1345
        out.returns.insert(out.returns.end(), {position.x, position.y, position.z});
        continue;
      }
      const int64 point_time = cloud.time + FromSeconds(time);  // :1345-1346
      while (control_points[next_control_point].time <= point_time) {  // :1347-1350
        if (next_control_point == last) break;
        ++next_control_point;
      }
      while (next_control_point > 0 && control_points[next_control_point - 1].time > point_time) {  // :1351-1354
        if (next_control_point - 1 == 0) break;
        --next_control_point;
      }
      if (next_control_point == 0) {  // CHECK(next_control_point != control_points_.begin()) (:1355)
        out.time_in_range = false;
        next_control_point = 1;
      }
      const ControlPoint& prev = control_points[next_control_point - 1];
      const ControlPoint& next = control_points[next_control_point];
      if (!(prev.time <= point_time && next.time >= point_time)) out.time_in_range = false;  // :1358-1359
      const Rigid3<double> transform_cloud =
          InterpolateTransformTimed(prev.pose, next.pose, prev.time, next.time, point_time);  // :1361-1365
      const Rigid3<double> rel = RigidMul(RigidInverse(optimized_pose), transform_cloud);
      const Rigid3<float> transform{{static_cast<float>(rel.t.x), static_cast<float>(rel.t.y), static_cast<float>(rel.t.z)},
                                    {static_cast<float>(rel.q.w), static_cast<float>(rel.q.x),
                                     static_cast<float>(rel.q.y), static_cast<float>(rel.q.z)}};  // :1366-1367
      const Vec3f p = transform * position;  // :1368-1369
      out.returns.insert(out.returns.end(), {p.x, p.y, p.z});
      if (first_point) {  // :1370-1374
        out.origin = transform * cloud.origin;
        first_point = false;
      }
    }
  }
  return out;
}

template <typename T> inline T GetRoll(const Quat<T>& q) {
  const T sinr_cosp = T(2.0) * (q.w * q.x + q.y * q.z);
  const T cosr_cosp = T(1.0) - T(2.0) * (q.x * q.x + q.y * q.y);
  return jatan2(sinr_cosp, cosr_cosp);
}
template <typename T> inline T GetPitch(const Quat<T>& q) {
  const T sinp = T(2.0) * (q.w * q.y - q.z * q.x);
  if (ScalarPart(jabs(sinp)) >= 1.0) return T(M_PI / 2);
  return jasin(sinp);
}
template <typename T> inline T GetYaw(const Quat<T>& q) {
  const Vec3<T> d = Rotate(q, Vec3<T>{T(1.0), T(0.0), T(0.0)});
  return jatan2(d.y, d.x);
}

// ref relative_translation_and_yaw_cost_function.h:41-63
template <typename T>
inline void OdometryResiduals(const SmallBlock& sb, const Rigid3<T>& start, const Rigid3<T>& end, T* r) {
  const Rigid3<T> delta = RigidMul(RigidInverse(end), start);
  const Rigid3<T> delta_c{{T(sb.delta[0]), T(sb.delta[1]), T(sb.delta[2])},
                          {T(sb.delta[3]), T(sb.delta[4]), T(sb.delta[5]), T(sb.delta[6])}};
  const Rigid3<T> error = RigidMul(RigidInverse(delta), delta_c);
  r[0] = sb.w[0] * error.t.x;
  r[1] = sb.w[0] * error.t.y;
  r[2] = sb.w[0] * error.t.z;
  r[3] = sb.w[1] * GetRoll(error.q);
  r[4] = sb.w[1] * GetPitch(error.q);
  r[5] = sb.w[1] * GetYaw(error.q);
}

// ref prediction_imu_preintegration_cost_functor.h:49-101 (the live, un-commented form)
template <typename T>
inline void ImuPreintegrationResiduals(const SmallBlock& sb, const Vec3<T>& t0, const Vec3<T>& v0,
                                       const Quat<T>& q0, const Vec3<T>& t1, const Vec3<T>& v1,
                                       const Quat<T>& q1, T* r) {
  const Vec3<T> te = t1 - t0 - T(sb.dt) * v0;
  const Vec3<T> ve = v1 - v0;
  const Quat<T> q1c{q1.w, -q1.x, -q1.y, -q1.z};
  const Quat<T> dq{T(sb.delta[3]), T(sb.delta[4]), T(sb.delta[5]), T(sb.delta[6])};
  const Quat<T> re = QuatMul(QuatMul(q1c, q0), dq);
  r[0] = sb.w[0] * te.x; r[1] = sb.w[0] * te.y; r[2] = sb.w[0] * te.z;
  r[3] = sb.w[1] * ve.x; r[4] = sb.w[1] * ve.y; r[5] = sb.w[1] * ve.z;
  r[6] = sb.w[2] * re.x; r[7] = sb.w[2] * re.y; r[8] = sb.w[2] * re.z;
}

struct ResidualBlock {
  const float* points = nullptr;  // n x 3
  size_t n = 0;
  std::vector<const HybridGridTSDF*> pyramid;  // 1 grid (single-res) or sorted fine->coarse
  bool multi_res = false;
  double scaling_factor = 1.0;
  int pose_a = 0;
  int pose_b = -1;  // -1: single-pose block (a12); else interpolated (a13)
  double interpolation_ratio = 0.0;
};

// Ceres 1.13 QuaternionParameterization::Plus / ComputeJacobian.
inline void QuaternionPlus(const double* x, const double* delta, double* x_plus_delta) {
  const double norm_delta =
      std::sqrt(delta[0] * delta[0] + delta[1] * delta[1] + delta[2] * delta[2]);
  if (norm_delta > 0.0) {
    const double sin_delta_by_delta = (std::sin(norm_delta) / norm_delta);
    double q_delta[4];
    q_delta[0] = std::cos(norm_delta);
    q_delta[1] = sin_delta_by_delta * delta[0];
    q_delta[2] = sin_delta_by_delta * delta[1];
    q_delta[3] = sin_delta_by_delta * delta[2];
    // ceres::QuaternionProduct(q_delta, x, x_plus_delta)
    x_plus_delta[0] = q_delta[0] * x[0] - q_delta[1] * x[1] - q_delta[2] * x[2] - q_delta[3] * x[3];
    x_plus_delta[1] = q_delta[0] * x[1] + q_delta[1] * x[0] + q_delta[2] * x[3] - q_delta[3] * x[2];
    x_plus_delta[2] = q_delta[0] * x[2] - q_delta[1] * x[3] + q_delta[2] * x[0] + q_delta[3] * x[1];
    x_plus_delta[3] = q_delta[0] * x[3] + q_delta[1] * x[2] - q_delta[2] * x[1] + q_delta[3] * x[0];
  } else {
    for (int i = 0; i < 4; ++i) x_plus_delta[i] = x[i];
  }
}
inline void QuaternionPlusJacobian(const double* x, double* jacobian /*4x3 row-major*/) {
  jacobian[0] = -x[1]; jacobian[1] = -x[2]; jacobian[2] = -x[3];
  jacobian[3] = x[0];  jacobian[4] = x[3];  jacobian[5] = -x[2];
  jacobian[6] = -x[3]; jacobian[7] = x[0];  jacobian[8] = x[1];
  jacobian[9] = x[2];  jacobian[10] = -x[1]; jacobian[11] = x[0];
}

struct SolverOptions {  // Ceres 1.13 defaults + trajectory_builder_3d.lua:49-53
  int max_num_iterations = 12;
  double initial_trust_region_radius = 1e4;
  double max_trust_region_radius = 1e16;
  double min_trust_region_radius = 1e-32;
  double min_relative_decrease = 1e-3;
  double min_lm_diagonal = 1e-6;
  double max_lm_diagonal = 1e32;
  int max_num_consecutive_invalid_steps = 5;
  double function_tolerance = 1e-6;
  double gradient_tolerance = 1e-10;
  double parameter_tolerance = 1e-8;
  bool jacobi_scaling = true;
};

enum TerminationType { CONVERGENCE = 0, NO_CONVERGENCE = 1, FAILURE = 2 };

struct SolverSummary {
  double initial_cost = 0, final_cost = 0;
  int num_iterations = 0;  // iterations.size() in Ceres (includes iteration 0)
  int num_successful_steps = 0, num_unsuccessful_steps = 0;
  int num_cost_evaluations = 0, num_jacobian_evaluations = 0;
  int termination_type = NO_CONVERGENCE;
  int termination_reason = 0;  // 1 grad tol, 2 param tol, 3 func tol, 4 max iter, 5 min radius, 6 invalid steps
  double final_radius = 0;
};

class Problem {
 public:
  std::vector<PoseBlock> poses;
  std::vector<ResidualBlock> blocks;
  std::vector<SmallBlock> small_blocks;
  LookupStats lookup_stats;

  static int SmallRows(const SmallBlock& sb) { return sb.type == 1 ? 6 : 9; }
  bool VelFree(int i) const { return poses[i].has_velocity && !poses[i].v_constant; }
  bool SmallActive(const SmallBlock& sb) const {
    if (!poses[sb.a].constant || !poses[sb.b].constant) return true;
    return sb.type == 2 && (VelFree(sb.a) || VelFree(sb.b));
  }
  int NumResiduals() const {
    int n = 0;
    for (const auto& b : blocks) if (BlockActive(b)) n += static_cast<int>(b.n);
    for (const auto& sb : small_blocks) if (SmallActive(sb)) n += SmallRows(sb);
    return n;
  }
  // Column layout of the reduced program, per control point in order: 6 local columns of a
  // non-constant pose (3 translation, 3 rotation tangent), then 3 of a non-constant velocity.
  int NumEffectiveParameters() const {
    int c = 0;
    for (size_t i = 0; i < poses.size(); ++i) {
      if (!poses[i].constant) c += 6;
      if (VelFree(static_cast<int>(i))) c += 3;
    }
    return c;
  }
  std::vector<int> VelocityOffsets() const {
    std::vector<int> off(poses.size(), -1);
    int c = 0;
    for (size_t i = 0; i < poses.size(); ++i) {
      if (!poses[i].constant) c += 6;
      if (VelFree(static_cast<int>(i))) { off[i] = c; c += 3; }
    }
    return off;
  }
  int NumParameters() const {
    int c = 0;
    for (const auto& p : poses) if (!p.constant) c += 7;
    return c;
  }
  std::vector<int> ColumnOffsets() const {
    std::vector<int> off(poses.size(), -1);
    int c = 0;
    for (size_t i = 0; i < poses.size(); ++i) {
      if (!poses[i].constant) { off[i] = c; c += 6; }
      if (VelFree(static_cast<int>(i))) c += 3;
    }
    return off;
  }
  bool BlockActive(const ResidualBlock& b) const {
    // Ceres drops residual blocks whose parameter blocks are all constant
    // (their cost goes to Summary::fixed_cost).
    if (b.n == 0) return false;
    if (!poses[b.pose_a].constant) return true;
    return b.pose_b >= 0 && !poses[b.pose_b].constant;
  }

  // Evaluates 0.5*|r|^2, residuals, the dense local Jacobian (row-major,
  // num_residuals x NumEffectiveParameters) and gradient = J^T r.
  // With jacobian == nullptr the functors are evaluated with T = double as
  // Ceres does for cost-only evaluations.
  void Evaluate(const std::vector<PoseBlock>& x, double* cost, double* residuals,
                double* jacobian, double* gradient) {
    const int ncols = NumEffectiveParameters();
    const std::vector<int> off = ColumnOffsets();
    double c = 0.0;
    int row0 = 0;
    std::vector<double> rbuf;
    if (gradient) std::fill(gradient, gradient + ncols, 0.0);
    for (const auto& b : blocks) {
      if (!BlockActive(b)) continue;
      double* r = residuals ? residuals + row0 : nullptr;
      if (!r) { rbuf.resize(b.n); r = rbuf.data(); }
      double* J = jacobian ? jacobian + static_cast<size_t>(row0) * ncols : nullptr;
      if (J) std::fill(J, J + b.n * static_cast<size_t>(ncols), 0.0);
      if (J) {
        if (b.pose_b < 0) EvalBlockJet<7>(b, x, off, ncols, r, J);
        else EvalBlockJet<14>(b, x, off, ncols, r, J);
      } else {
        EvalBlockDouble(b, x, r);
      }
      for (size_t i = 0; i < b.n; ++i) c += r[i] * r[i];
      if (gradient && J) {
        for (size_t i = 0; i < b.n; ++i)
          for (int k = 0; k < ncols; ++k) gradient[k] += J[i * ncols + k] * r[i];
      }
      row0 += static_cast<int>(b.n);
    }
    const std::vector<int> voff = VelocityOffsets();
    for (const auto& sb : small_blocks) {
      if (!SmallActive(sb)) continue;
      const int rows = SmallRows(sb);
      double rr[9];
      double* r = residuals ? residuals + row0 : rr;
      double* J = jacobian ? jacobian + static_cast<size_t>(row0) * ncols : nullptr;
      if (J) std::fill(J, J + rows * static_cast<size_t>(ncols), 0.0);
      EvalSmall(sb, x, off, voff, ncols, r, J);
      for (int i = 0; i < rows; ++i) c += r[i] * r[i];
      if (gradient && J)
        for (int i = 0; i < rows; ++i)
          for (int k = 0; k < ncols; ++k) gradient[k] += J[i * ncols + k] * r[i];
      row0 += rows;
    }
    *cost = 0.5 * c;
  }

  // Ceres 1.13 TrustRegionMinimizer + LevenbergMarquardtStrategy, dense normal
  // equations solved by Cholesky. See DESIGN.md for the restated flow.
  SolverSummary Solve(const SolverOptions& opt);

 private:
  // Jet<20> layout: t_a(0-2) v_a(3-5) q_a(6-9) t_b(10-12) v_b(13-15) q_b(16-19), as the parameter
  // block order of the IMU functor; the odometry functor uses t/q of both.
  void EvalSmall(const SmallBlock& sb, const std::vector<PoseBlock>& x, const std::vector<int>& off,
                 const std::vector<int>& voff, int ncols, double* r, double* J) {
    const PoseBlock& A = x[sb.a];
    const PoseBlock& B = x[sb.b];
    const int rows = SmallRows(sb);
    if (!J) {
      const Rigid3<double> Ta{{A.t[0], A.t[1], A.t[2]}, {A.q[0], A.q[1], A.q[2], A.q[3]}};
      const Rigid3<double> Tb{{B.t[0], B.t[1], B.t[2]}, {B.q[0], B.q[1], B.q[2], B.q[3]}};
      if (sb.type == 1) OdometryResiduals<double>(sb, Ta, Tb, r);
      else ImuPreintegrationResiduals<double>(sb, Ta.t, Vec3d{A.v[0], A.v[1], A.v[2]}, Ta.q, Tb.t,
                                              Vec3d{B.v[0], B.v[1], B.v[2]}, Tb.q, r);
      return;
    }
    using JT = Jet<20>;
    const Vec3<JT> ta{JT(A.t[0], 0), JT(A.t[1], 1), JT(A.t[2], 2)};
    const Vec3<JT> va{JT(A.v[0], 3), JT(A.v[1], 4), JT(A.v[2], 5)};
    const Quat<JT> qa{JT(A.q[0], 6), JT(A.q[1], 7), JT(A.q[2], 8), JT(A.q[3], 9)};
    const Vec3<JT> tb{JT(B.t[0], 10), JT(B.t[1], 11), JT(B.t[2], 12)};
    const Vec3<JT> vb{JT(B.v[0], 13), JT(B.v[1], 14), JT(B.v[2], 15)};
    const Quat<JT> qb{JT(B.q[0], 16), JT(B.q[1], 17), JT(B.q[2], 18), JT(B.q[3], 19)};
    JT res[9];
    if (sb.type == 1) OdometryResiduals<JT>(sb, Rigid3<JT>{ta, qa}, Rigid3<JT>{tb, qb}, res);
    else ImuPreintegrationResiduals<JT>(sb, ta, va, qa, tb, vb, qb, res);
    double pja[12], pjb[12];
    QuaternionPlusJacobian(A.q, pja);
    QuaternionPlusJacobian(B.q, pjb);
    for (int i = 0; i < rows; ++i) {
      r[i] = res[i].a;
      double* row = J + static_cast<size_t>(i) * ncols;
      auto put_pose = [&](int col, int tbase, int qbase, const double* pj) {
        if (col < 0) return;
        for (int k = 0; k < 3; ++k) row[col + k] = res[i].v[tbase + k];
        for (int k = 0; k < 3; ++k) {
          double s2 = 0.0;
          for (int j = 0; j < 4; ++j) s2 += res[i].v[qbase + j] * pj[j * 3 + k];
          row[col + 3 + k] = s2;
        }
      };
      put_pose(A.constant ? -1 : off[sb.a], 0, 6, pja);
      put_pose(B.constant ? -1 : off[sb.b], 10, 16, pjb);
      if (voff[sb.a] >= 0) for (int k = 0; k < 3; ++k) row[voff[sb.a] + k] = res[i].v[3 + k];
      if (voff[sb.b] >= 0) for (int k = 0; k < 3; ++k) row[voff[sb.b] + k] = res[i].v[13 + k];
    }
  }

  template <typename T>
  T BlockTSD(const ResidualBlock& b, const T& wx, const T& wy, const T& wz) {
    if (b.multi_res) return InterpolatedMultiResGetTSD(b.pyramid, wx, wy, wz, &lookup_stats);
    ++lookup_stats.lookups;
    ++lookup_stats.levels_probed;
    return InterpolatedGetTSD(*b.pyramid.front(), wx, wy, wz);
  }

  void EvalBlockDouble(const ResidualBlock& b, const std::vector<PoseBlock>& x, double* r) {
    Rigid3<double> T;
    const PoseBlock& pa = x[b.pose_a];
    Rigid3<double> Ta{{pa.t[0], pa.t[1], pa.t[2]}, {pa.q[0], pa.q[1], pa.q[2], pa.q[3]}};
    if (b.pose_b < 0) {
      T = Ta;
    } else {
      const PoseBlock& pb = x[b.pose_b];
      Rigid3<double> Tb{{pb.t[0], pb.t[1], pb.t[2]}, {pb.q[0], pb.q[1], pb.q[2], pb.q[3]}};
      T = InterpolateTransform(Ta, Tb, b.interpolation_ratio);
    }
    for (size_t i = 0; i < b.n; ++i) {
      const Vec3d p{static_cast<double>(b.points[3 * i]), static_cast<double>(b.points[3 * i + 1]),
                    static_cast<double>(b.points[3 * i + 2])};
      const Vec3d world = T * p;
      const double tsd = BlockTSD(b, world.x, world.y, world.z);
      r[i] = b.scaling_factor * tsd;
    }
  }

  template <int N>
  void EvalBlockJet(const ResidualBlock& b, const std::vector<PoseBlock>& x,
                    const std::vector<int>& off, int ncols, double* r, double* J) {
    using JT = Jet<N>;
    const PoseBlock& pa = x[b.pose_a];
    Rigid3<JT> Ta{{JT(pa.t[0], 0), JT(pa.t[1], 1), JT(pa.t[2], 2)},
                  {JT(pa.q[0], 3), JT(pa.q[1], 4), JT(pa.q[2], 5), JT(pa.q[3], 6)}};
    Rigid3<JT> T = Ta;
    if (N == 14) {
      const PoseBlock& pb = x[b.pose_b];
      Rigid3<JT> Tb{{JT(pb.t[0], 7 % N), JT(pb.t[1], 8 % N), JT(pb.t[2], 9 % N)},
                    {JT(pb.q[0], 10 % N), JT(pb.q[1], 11 % N), JT(pb.q[2], 12 % N),
                     JT(pb.q[3], 13 % N)}};
      T = InterpolateTransform(Ta, Tb, b.interpolation_ratio);
    }
    // local parameterization Jacobians (4x3) of the rotation blocks
    double pja[12], pjb[12];
    QuaternionPlusJacobian(pa.q, pja);
    if (N == 14) QuaternionPlusJacobian(x[b.pose_b].q, pjb);
    const int ca = pa.constant ? -1 : off[b.pose_a];
    const int cb = (N == 14 && !x[b.pose_b].constant) ? off[b.pose_b] : -1;
    for (size_t i = 0; i < b.n; ++i) {
      const Vec3<JT> p{JT(static_cast<double>(b.points[3 * i])),
                       JT(static_cast<double>(b.points[3 * i + 1])),
                       JT(static_cast<double>(b.points[3 * i + 2]))};
      const Vec3<JT> world = T * p;
      const JT tsd = BlockTSD(b, world.x, world.y, world.z);
      const JT res = b.scaling_factor * tsd;
      r[i] = res.a;
      double* row = J + i * static_cast<size_t>(ncols);
      if (ca >= 0) {
        for (int k = 0; k < 3; ++k) row[ca + k] = res.v[k];
        for (int k = 0; k < 3; ++k) {
          double s = 0.0;
          for (int j = 0; j < 4; ++j) s += res.v[3 + j] * pja[j * 3 + k];
          row[ca + 3 + k] = s;
        }
      }
      if (cb >= 0) {
        for (int k = 0; k < 3; ++k) row[cb + k] = res.v[(7 + k) % N];
        for (int k = 0; k < 3; ++k) {
          double s = 0.0;
          for (int j = 0; j < 4; ++j) s += res.v[(10 + j) % N] * pjb[j * 3 + k];
          row[cb + 3 + k] = s;
        }
      }
    }
  }
};

// Dense symmetric positive definite solve (Cholesky, LLT). Returns false if
// the matrix is not positive definite or the result is not finite.
inline bool CholeskySolve(int n, std::vector<double> A, const double* b, double* x) {
  for (int j = 0; j < n; ++j) {
    double d = A[j * n + j];
    for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
    if (!(d > 0.0) || !std::isfinite(d)) return false;
    const double l = std::sqrt(d);
    A[j * n + j] = l;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i * n + j];
      for (int k = 0; k < j; ++k) s -= A[i * n + k] * A[j * n + k];
      A[i * n + j] = s / l;
    }
  }
  std::vector<double> y(n);
  for (int i = 0; i < n; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= A[i * n + k] * y[k];
    y[i] = s / A[i * n + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < n; ++k) s -= A[k * n + i] * x[k];
    x[i] = s / A[i * n + i];
  }
  for (int i = 0; i < n; ++i) if (!std::isfinite(x[i])) return false;
  return true;
}

inline void PosePlus(const std::vector<PoseBlock>& x, const double* delta,
                     std::vector<PoseBlock>* out) {
  *out = x;
  int c = 0;
  for (size_t i = 0; i < x.size(); ++i) {
    if (!x[i].constant) {
      for (int k = 0; k < 3; ++k) (*out)[i].t[k] = x[i].t[k] + delta[c + k];
      QuaternionPlus(x[i].q, delta + c + 3, (*out)[i].q);
      c += 6;
    }
    if (x[i].has_velocity && !x[i].v_constant) {
      for (int k = 0; k < 3; ++k) (*out)[i].v[k] = x[i].v[k] + delta[c + k];
      c += 3;
    }
  }
}

inline double AmbientNorm(const std::vector<PoseBlock>& x) {
  double s = 0.0;
  for (const auto& p : x) {
    if (!p.constant) {
      for (int k = 0; k < 3; ++k) s += p.t[k] * p.t[k];
      for (int k = 0; k < 4; ++k) s += p.q[k] * p.q[k];
    }
    if (p.has_velocity && !p.v_constant)
      for (int k = 0; k < 3; ++k) s += p.v[k] * p.v[k];
  }
  return std::sqrt(s);
}
inline void AmbientDiffNorms(const std::vector<PoseBlock>& a, const std::vector<PoseBlock>& b,
                             double* l2, double* linf) {
  double s = 0.0, m = 0.0;
  for (size_t i = 0; i < a.size(); ++i) {
    if (!a[i].constant) {
      for (int k = 0; k < 3; ++k) {
        const double d = a[i].t[k] - b[i].t[k];
        s += d * d; m = std::max(m, std::fabs(d));
      }
      for (int k = 0; k < 4; ++k) {
        const double d = a[i].q[k] - b[i].q[k];
        s += d * d; m = std::max(m, std::fabs(d));
      }
    }
    if (a[i].has_velocity && !a[i].v_constant)
      for (int k = 0; k < 3; ++k) {
        const double d = a[i].v[k] - b[i].v[k];
        s += d * d; m = std::max(m, std::fabs(d));
      }
  }
  *l2 = std::sqrt(s);
  *linf = m;
}

inline SolverSummary Problem::Solve(const SolverOptions& opt) {
  SolverSummary sum;
  const int ncols = NumEffectiveParameters();
  const int nres = NumResiduals();
  // fixed cost of dropped blocks is not tracked (not needed for parity of x).
  if (ncols == 0) {
    sum.termination_type = CONVERGENCE;
    sum.num_iterations = 0;
    return sum;
  }
  std::vector<PoseBlock>& x = poses;
  std::vector<PoseBlock> candidate_x;
  std::vector<double> residuals(nres), J(static_cast<size_t>(nres) * ncols), gradient(ncols);
  std::vector<double> scale(ncols, 1.0), diagonal(ncols), lm_diagonal(ncols), step(ncols),
      delta(ncols), model_residuals(nres), rhs(ncols), A(static_cast<size_t>(ncols) * ncols);
  double x_cost = 0.0, candidate_cost = 0.0;
  double radius = opt.initial_trust_region_radius;
  double decrease_factor = 2.0;
  bool reuse_diagonal = false;
  int num_consecutive_invalid_steps = 0;
  int iteration = 0;
  double gradient_max_norm = 0.0;
  bool step_is_successful = false;

  auto EvaluateGradientAndJacobian = [&]() {
    Evaluate(x, &x_cost, residuals.data(), J.data(), gradient.data());
    ++sum.num_cost_evaluations;
    ++sum.num_jacobian_evaluations;
    if (opt.jacobi_scaling) {
      if (iteration == 0) {
        for (int k = 0; k < ncols; ++k) {
          double s = 0.0;
          for (int i = 0; i < nres; ++i) s += J[static_cast<size_t>(i) * ncols + k] * J[static_cast<size_t>(i) * ncols + k];
          scale[k] = 1.0 / (1.0 + std::sqrt(s));
        }
      }
      for (int i = 0; i < nres; ++i)
        for (int k = 0; k < ncols; ++k) J[static_cast<size_t>(i) * ncols + k] *= scale[k];
    }
    // gradient_max_norm = |x - Plus(x, -gradient)|_inf (ambient space)
    std::vector<double> neg(ncols);
    for (int k = 0; k < ncols; ++k) neg[k] = -gradient[k];
    std::vector<PoseBlock> proj;
    PosePlus(x, neg.data(), &proj);
    double l2, linf;
    AmbientDiffNorms(x, proj, &l2, &linf);
    gradient_max_norm = linf;
  };

  // IterationZero
  EvaluateGradientAndJacobian();
  sum.initial_cost = x_cost;
  step_is_successful = true;
  sum.num_iterations = 1;

  auto finish = [&](int type, int reason) {
    sum.termination_type = type;
    sum.termination_reason = reason;
    sum.final_cost = x_cost;
    sum.final_radius = radius;
    return sum;
  };

  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (step_is_successful) ++sum.num_successful_steps; else ++sum.num_unsuccessful_steps;
    if (iteration >= opt.max_num_iterations) return finish(NO_CONVERGENCE, 4);
    if (step_is_successful && gradient_max_norm <= opt.gradient_tolerance)
      return finish(CONVERGENCE, 1);
    if (radius <= opt.min_trust_region_radius) return finish(CONVERGENCE, 5);

    ++iteration;
    ++sum.num_iterations;
    step_is_successful = false;

    // ComputeTrustRegionStep (LevenbergMarquardtStrategy::ComputeStep)
    if (!reuse_diagonal) {
      for (int k = 0; k < ncols; ++k) {
        double s = 0.0;
        for (int i = 0; i < nres; ++i) s += J[static_cast<size_t>(i) * ncols + k] * J[static_cast<size_t>(i) * ncols + k];
        diagonal[k] = std::min(std::max(s, opt.min_lm_diagonal), opt.max_lm_diagonal);
      }
    }
    for (int k = 0; k < ncols; ++k) lm_diagonal[k] = std::sqrt(diagonal[k] / radius);
    // normal equations (J^T J + D^T D) y = J^T r ; step = -y
    std::fill(A.begin(), A.end(), 0.0);
    std::fill(rhs.begin(), rhs.end(), 0.0);
    for (int i = 0; i < nres; ++i) {
      const double* row = &J[static_cast<size_t>(i) * ncols];
      for (int a = 0; a < ncols; ++a) {
        if (row[a] == 0.0) continue;
        rhs[a] += row[a] * residuals[i];
        for (int b2 = 0; b2 < ncols; ++b2) A[a * ncols + b2] += row[a] * row[b2];
      }
    }
    for (int k = 0; k < ncols; ++k) A[k * ncols + k] += lm_diagonal[k] * lm_diagonal[k];
    bool step_is_valid = CholeskySolve(ncols, A, rhs.data(), step.data());
    reuse_diagonal = true;
    double model_cost_change = 0.0;
    if (step_is_valid) {
      for (int k = 0; k < ncols; ++k) step[k] = -step[k];
      // model_cost_change = -model_residuals . (residuals + model_residuals / 2)
      double mcc = 0.0;
      for (int i = 0; i < nres; ++i) {
        double mr = 0.0;
        const double* row = &J[static_cast<size_t>(i) * ncols];
        for (int k = 0; k < ncols; ++k) mr += row[k] * step[k];
        mcc += mr * (residuals[i] + mr / 2.0);
      }
      model_cost_change = -mcc;
      step_is_valid = model_cost_change > 0.0;
    }
    if (!step_is_valid) {
      // HandleInvalidStep
      ++num_consecutive_invalid_steps;
      if (num_consecutive_invalid_steps >= opt.max_num_consecutive_invalid_steps)
        return finish(FAILURE, 6);
      radius = radius / decrease_factor;  // StepIsInvalid == StepRejected
      decrease_factor *= 2.0;
      reuse_diagonal = true;
      continue;
    }
    num_consecutive_invalid_steps = 0;
    for (int k = 0; k < ncols; ++k) delta[k] = step[k] * scale[k];

    // ComputeCandidatePointAndEvaluateCost
    PosePlus(x, delta.data(), &candidate_x);
    Evaluate(candidate_x, &candidate_cost, nullptr, nullptr, nullptr);
    ++sum.num_cost_evaluations;

    // ParameterToleranceReached
    double step_norm, step_inf;
    AmbientDiffNorms(x, candidate_x, &step_norm, &step_inf);
    const double x_norm = AmbientNorm(x);
    if (step_norm <= opt.parameter_tolerance * (x_norm + opt.parameter_tolerance))
      return finish(CONVERGENCE, 2);
    // FunctionToleranceReached
    const double cost_change = x_cost - candidate_cost;
    if (std::fabs(cost_change) <= opt.function_tolerance * x_cost)
      return finish(CONVERGENCE, 3);
    // IsStepSuccessful
    const double relative_decrease = cost_change / model_cost_change;
    if (relative_decrease > opt.min_relative_decrease) {
      // HandleSuccessfulStep
      x = candidate_x;
      EvaluateGradientAndJacobian();
      step_is_successful = true;
      radius = radius / std::max(1.0 / 3.0, 1.0 - std::pow(2.0 * relative_decrease - 1.0, 3));
      radius = std::min(opt.max_trust_region_radius, radius);
      decrease_factor = 2.0;
      reuse_diagonal = false;
    } else {
      // HandleUnsuccessfulStep
      radius = radius / decrease_factor;
      decrease_factor *= 2.0;
      reuse_diagonal = true;
    }
  }
}

// ---------------------------------------------------------------------------
// X-ray texture of a TSDF submap (the view Submap3D::ToResponseProto serves).
// ref: mapping/3d/submap_3d.cc:31-37 (PixelData), :80-105 (AccumulatePixelData), :142-177
// (ExtractVoxelData, TSDF overload), :179-214 (ComputePixelValues), :245-276 (AddToTextureProto,
// TSDF overload; the gzip of the cell string and the slice pose stay with the caller);
// mapping/probability_values.h:32-44,66-69,92-105 and .cc:30-62; mapping/submaps.h:36-52.
// ---------------------------------------------------------------------------
struct XrayTexture {
  int width = 0, height = 0;  // width = y extent, height = x extent (submap_3d.cc:261-262)
  int max_x = 0, max_y = 0;   // max_index, the caller needs it for slice_pose (:271-275)
  std::vector<uint8_t> cells;  // interleaved value, alpha; pixel (x, y) at x * width + y
};

constexpr float kMinProbability = 0.1f;
constexpr float kMaxProbability = 1.f - kMinProbability;

inline uint16 ProbabilityToValue(const float probability) {
  const int value = RoundToInt((Clamp(probability, kMinProbability, kMaxProbability) - kMinProbability) *
                               (32766.f / (kMaxProbability - kMinProbability))) +
                    1;
  return static_cast<uint16>(value);
}

// kValueToProbability (probability_values.cc:30-62), entries [0, 32767]
inline float ValueToProbability(const int value) {
  if (value == 0) return kMinProbability;
  const float kScale = (kMaxProbability - kMinProbability) / (32768 - 2.f);
  return value * kScale + (kMinProbability - kScale);
}

inline float Logit(float probability) { return std::log(probability / (1.f - probability)); }

inline uint8_t ProbabilityToLogOddsInteger(const float probability) {
  const float kMaxLogOdds = Logit(kMaxProbability);
  const float kMinLogOdds = Logit(kMinProbability);
  const int value = RoundToInt((Logit(probability) - kMinLogOdds) * 254.f / (kMaxLogOdds - kMinLogOdds)) + 1;
  return static_cast<uint8_t>(value);
}

inline XrayTexture XrayTextureTSDF(const HybridGridTSDF& grid, const Rigid3<double>& global_submap_pose) {
  XrayTexture out;
  const Rigid3<float> transform{{static_cast<float>(global_submap_pose.t.x),
                                 static_cast<float>(global_submap_pose.t.y),
                                 static_cast<float>(global_submap_pose.t.z)},
                                {static_cast<float>(global_submap_pose.q.w),
                                 static_cast<float>(global_submap_pose.q.x),
                                 static_cast<float>(global_submap_pose.q.y),
                                 static_cast<float>(global_submap_pose.q.z)}};
  // ExtractVoxelData
  struct Entry { int x, y, z, value; };
  std::vector<Entry> entries;
  const float resolution_inverse = 1.f / grid.resolution();
  constexpr float kXrayObstructedCellProbabilityLimit = 0.501f;
  int min_x = INT_MAX, min_y = INT_MAX, max_x = INT_MIN, max_y = INT_MIN;
  grid.grid().ForEach([&](const Vec3i& index, const TSDFVoxel& voxel) {
    const float tsd = grid.ValueConverter().ValueToTSD(voxel.discrete_tsd);
    const float probability = 1.f - std::abs(tsd) / grid.ValueConverter().getMaxTSD();
    const float probability_value = ProbabilityToValue(probability);
    if (probability < kXrayObstructedCellProbabilityLimit) return;
    const Vec3f cell_center_global = transform * grid.GetCenterOfCell(index);
    const Entry e{RoundToInt(cell_center_global.x * resolution_inverse),
                  RoundToInt(cell_center_global.y * resolution_inverse),
                  RoundToInt(cell_center_global.z * resolution_inverse),
                  static_cast<int>(probability_value)};
    entries.push_back(e);
    min_x = std::min(min_x, e.x);
    min_y = std::min(min_y, e.y);
    max_x = std::max(max_x, e.x);
    max_y = std::max(max_y, e.y);
  });
  if (entries.empty()) return out;  // the reference computes INT_MIN - INT_MAX + 1 here (UB)
  out.width = max_y - min_y + 1;
  out.height = max_x - min_x + 1;
  out.max_x = max_x;
  out.max_y = max_y;
  // AccumulatePixelData
  struct PixelData {
    int min_z = INT_MAX, max_z = INT_MIN, count = 0;
    float probability_sum = 0.f, max_probability = 0.5f;
  };
  std::vector<PixelData> pixels(static_cast<size_t>(out.width) * out.height);
  for (const Entry& e : entries) {
    const int x = max_x - e.x, y = max_y - e.y;
    PixelData& pixel = pixels[static_cast<size_t>(x) * out.width + y];
    ++pixel.count;
    pixel.min_z = std::min(pixel.min_z, e.z);
    pixel.max_z = std::max(pixel.max_z, e.z);
    const float probability = ValueToProbability(e.value);
    pixel.probability_sum += probability;
    pixel.max_probability = std::max(pixel.max_probability, probability);
  }
  // ComputePixelValues
  out.cells.reserve(2 * pixels.size());
  constexpr float kMinZDifference = 3.f;
  constexpr float kFreeSpaceWeight = 0.15f;
  for (const PixelData& pixel : pixels) {
    const float z_difference = pixel.count > 0 ? pixel.max_z - pixel.min_z : 0;
    if (z_difference < kMinZDifference) {
      out.cells.push_back(0);
      out.cells.push_back(0);
      continue;
    }
    const float free_space = std::max(z_difference - pixel.count, 0.f);
    const float free_space_weight = kFreeSpaceWeight * free_space;
    const float total_weight = pixel.count + free_space_weight;
    const float free_space_probability = 1.f - pixel.max_probability;
    const float average_probability =
        Clamp((pixel.probability_sum + free_space_probability * free_space_weight) / total_weight,
              kMinProbability, kMaxProbability);
    const int delta = 128 - ProbabilityToLogOddsInteger(average_probability);
    const uint8_t alpha = delta > 0 ? 0 : -delta;
    const uint8_t value = delta > 0 ? delta : 0;
    out.cells.push_back(value);
    out.cells.push_back((value || alpha) ? alpha : 1);
  }
  return out;
}

}  // namespace hgo
