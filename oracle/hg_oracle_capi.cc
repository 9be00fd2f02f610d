// hg_oracle_capi.cc — C entry points over hg_oracle.hpp for ctypes.
// CPU ORACLE, TEST INFRASTRUCTURE ONLY (see hg_oracle.hpp header comment).
#include <chrono>
#include <cstdio>

#include "hg_oracle.hpp"

using namespace hgo;

extern "C" {

struct hgo_insert_opts {
  double relative_truncation_distance;
  double maximum_weight;
  int num_free_space_voxels;
  int project_sdf_distance_to_scan_normal;
  double weight_function_epsilon;
  double weight_function_sigma;
  double min_range;
  double max_range;
  double insertion_ratio;
  int normal_computation_method;
  int normal_computation_horizontal_stride;
  int normal_computation_vertical_stride;
  int reserved;
};

struct hgo_solver_opts {
  int max_num_iterations;
  int jacobi_scaling;
  double initial_trust_region_radius;
  double max_trust_region_radius;
  double min_trust_region_radius;
  double min_relative_decrease;
  double min_lm_diagonal;
  double max_lm_diagonal;
  double function_tolerance;
  double gradient_tolerance;
  double parameter_tolerance;
};

struct hgo_solver_summary {
  double initial_cost, final_cost, final_radius;
  int num_iterations, num_successful_steps, num_unsuccessful_steps;
  int num_cost_evaluations, num_jacobian_evaluations;
  int termination_type, termination_reason;
  int reserved;
};

static InserterOptions ToOpts(const hgo_insert_opts* o) {
  InserterOptions r;
  r.relative_truncation_distance = o->relative_truncation_distance;
  r.maximum_weight = o->maximum_weight;
  r.num_free_space_voxels = o->num_free_space_voxels;
  r.project_sdf_distance_to_scan_normal = o->project_sdf_distance_to_scan_normal != 0;
  r.weight_function_epsilon = o->weight_function_epsilon;
  r.weight_function_sigma = o->weight_function_sigma;
  r.min_range = o->min_range;
  r.max_range = o->max_range;
  r.insertion_ratio = o->insertion_ratio;
  r.normal_computation_method = o->normal_computation_method;
  r.normal_computation_horizontal_stride = o->normal_computation_horizontal_stride;
  r.normal_computation_vertical_stride = o->normal_computation_vertical_stride;
  return r;
}

// ---- codec ----------------------------------------------------------------
void hgo_conversion_table(float unknown_result, float lower, float upper, float* out65536) {
  const std::vector<float> t = PrecomputeValueToBoundedFloat(0, unknown_result, lower, upper);
  std::memcpy(out65536, t.data(), t.size() * sizeof(float));
}

void* hgo_converter_create(float max_tsd, float max_weight) {
  return new TSDValueConverter(max_tsd, max_weight);
}
void hgo_converter_destroy(void* c) { delete static_cast<TSDValueConverter*>(c); }
uint16_t hgo_tsd_to_value(void* c, float tsd) {
  return static_cast<TSDValueConverter*>(c)->TSDToValue(tsd);
}
uint16_t hgo_weight_to_value(void* c, float w) {
  return static_cast<TSDValueConverter*>(c)->WeightToValue(w);
}
float hgo_value_to_tsd(void* c, uint16_t v) {
  return static_cast<TSDValueConverter*>(c)->ValueToTSD(v);
}
float hgo_value_to_weight(void* c, uint16_t v) {
  return static_cast<TSDValueConverter*>(c)->ValueToWeight(v);
}

// ---- grid -----------------------------------------------------------------
void* hgo_grid_create(float resolution, float relative_truncation_distance, float max_weight) {
  return new HybridGridTSDF(resolution, relative_truncation_distance, max_weight);
}
void hgo_grid_destroy(void* g) { delete static_cast<HybridGridTSDF*>(g); }
float hgo_grid_resolution(void* g) { return static_cast<HybridGridTSDF*>(g)->resolution(); }
float hgo_grid_max_tsd(void* g) {
  return static_cast<HybridGridTSDF*>(g)->ValueConverter().getMaxTSD();
}
void hgo_grid_cell_index(void* g, const float* xyz, size_t m, int32_t* ijk) {
  auto* grid = static_cast<HybridGridTSDF*>(g);
  for (size_t i = 0; i < m; ++i) {
    const Vec3i c = grid->GetCellIndex({xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]});
    ijk[3 * i] = c.x; ijk[3 * i + 1] = c.y; ijk[3 * i + 2] = c.z;
  }
}
void hgo_grid_center_of_cell(void* g, const int32_t* ijk, size_t m, float* xyz) {
  auto* grid = static_cast<HybridGridTSDF*>(g);
  for (size_t i = 0; i < m; ++i) {
    const Vec3f c = grid->GetCenterOfCell({ijk[3 * i], ijk[3 * i + 1], ijk[3 * i + 2]});
    xyz[3 * i] = c.x; xyz[3 * i + 1] = c.y; xyz[3 * i + 2] = c.z;
  }
}
int hgo_grid_set_cell(void* g, int x, int y, int z, float tsd, float weight) {
  return static_cast<HybridGridTSDF*>(g)->SetCell({x, y, z}, tsd, weight) ? 0 : -1;
}
void hgo_grid_read_cells(void* g, const int32_t* ijk, size_t m, uint16_t* tsd, uint16_t* weight) {
  auto* grid = static_cast<HybridGridTSDF*>(g);
  for (size_t i = 0; i < m; ++i) {
    const TSDFVoxel v = grid->RawValue({ijk[3 * i], ijk[3 * i + 1], ijk[3 * i + 2]});
    tsd[i] = v.discrete_tsd;
    weight[i] = v.discrete_weight;
  }
}
void hgo_grid_get_float(void* g, const int32_t* ijk, size_t m, float* tsd, float* weight,
                        uint8_t* known) {
  auto* grid = static_cast<HybridGridTSDF*>(g);
  for (size_t i = 0; i < m; ++i) {
    const Vec3i c{ijk[3 * i], ijk[3 * i + 1], ijk[3 * i + 2]};
    tsd[i] = grid->GetTSD(c);
    weight[i] = grid->GetWeight(c);
    known[i] = grid->IsKnown(c) ? 1 : 0;
  }
}
size_t hgo_grid_count(void* g) {
  size_t n = 0;
  static_cast<HybridGridTSDF*>(g)->grid().ForEach([&](const Vec3i&, const TSDFVoxel&) { ++n; });
  return n;
}
// Export in the reference's iterator order (what ToProto would emit).
size_t hgo_grid_export(void* g, int32_t* ijk, uint16_t* tsd, uint16_t* weight, size_t cap) {
  size_t n = 0;
  static_cast<HybridGridTSDF*>(g)->grid().ForEach([&](const Vec3i& c, const TSDFVoxel& v) {
    if (n < cap) {
      ijk[3 * n] = c.x; ijk[3 * n + 1] = c.y; ijk[3 * n + 2] = c.z;
      tsd[n] = v.discrete_tsd;
      weight[n] = v.discrete_weight;
    }
    ++n;
  });
  return n;
}

// X-ray texture (submap_3d.cc:245-276). pose_tq: double[7] global submap pose (t xyz, q wxyz).
// Returns the bytes of the cell string; writes min(cap, bytes) of them.
size_t hgo_grid_xray(void* g, const double* pose_tq, uint8_t* cells, size_t cap, int32_t* width,
                     int32_t* height, int32_t* max_index_xy) {
  const Rigid3<double> T{{pose_tq[0], pose_tq[1], pose_tq[2]}, {pose_tq[3], pose_tq[4], pose_tq[5], pose_tq[6]}};
  const XrayTexture t = XrayTextureTSDF(*static_cast<HybridGridTSDF*>(g), T);
  *width = t.width;
  *height = t.height;
  max_index_xy[0] = t.max_x;
  max_index_xy[1] = t.max_y;
  if (cells && cap) std::memcpy(cells, t.cells.data(), std::min(cap, t.cells.size()));
  return t.cells.size();
}

// ---- inserter -------------------------------------------------------------
// pose_tq: optional float[7] (t xyz, q wxyz) = local_pose().inverse().cast<float>()
// applied to origin and returns first (Submap3D::InsertData, submap_3d.cc:436-437).
int hgo_grid_insert(void* g, const hgo_insert_opts* opts, const float* origin, const float* xyz,
                    size_t n, size_t width, const float* pose_tq, uint64_t* stats2) {
  auto* grid = static_cast<HybridGridTSDF*>(g);
  const TSDFRangeDataInserter3D inserter(ToOpts(opts));
  InsertStats st;
  Vec3f o{origin[0], origin[1], origin[2]};
  std::vector<float> tmp;
  const float* pts = xyz;
  if (pose_tq) {
    Rigid3<float> T{{pose_tq[0], pose_tq[1], pose_tq[2]},
                    {pose_tq[3], pose_tq[4], pose_tq[5], pose_tq[6]}};
    tmp.resize(3 * n);
    TransformPointsF(T, xyz, n, tmp.data());
    pts = tmp.data();
    o = T * o;
  }
  if (opts->project_sdf_distance_to_scan_normal) {
    if (opts->normal_computation_method != 1) return -2;  // only CLOUD_STRUCTURE restated
    inserter.InsertWithCloudStructureNormals(o, pts, n, width, grid, &st);
  } else {
    inserter.Insert(o, pts, n, width, grid, &st);
  }
  if (stats2) { stats2[0] = st.num_hits; stats2[1] = st.num_updates; }
  return 0;
}

// ---- voxel filters --------------------------------------------------------
size_t hgo_voxel_filter(float resolution, const float* pts, size_t n, int stride, uint32_t* out) {
  const std::vector<uint32_t> r = VoxelFilterIndices(resolution, pts, n, stride);
  if (out) std::memcpy(out, r.data(), r.size() * sizeof(uint32_t));
  return r.size();
}
size_t hgo_adaptive_voxel_filter(float max_length, float min_num_points, float max_range,
                                 const float* pts, size_t n, int stride, uint32_t* out) {
  const std::vector<uint32_t> r = AdaptiveVoxelFilterIndices(max_length, min_num_points, max_range, pts, n, stride);
  if (out) std::memcpy(out, r.data(), r.size() * sizeof(uint32_t));
  return r.size();
}

// ---- interpolation --------------------------------------------------------
// value + gradient of (multi-res) interpolated TSD at m double points.
void hgo_interp_tsd(void* const* grids, int levels, int multi_res, const double* xyz, size_t m,
                    double* value, double* grad3) {
  std::vector<const HybridGridTSDF*> pyr;
  for (int l = 0; l < levels; ++l) pyr.push_back(static_cast<const HybridGridTSDF*>(grids[l]));
  for (size_t i = 0; i < m; ++i) {
    const Jet<3> x(xyz[3 * i], 0), y(xyz[3 * i + 1], 1), z(xyz[3 * i + 2], 2);
    const Jet<3> r = multi_res ? InterpolatedMultiResGetTSD(pyr, x, y, z)
                               : InterpolatedGetTSD(*pyr.front(), x, y, z);
    value[i] = r.a;
    if (grad3) { grad3[3 * i] = r.v[0]; grad3[3 * i + 1] = r.v[1]; grad3[3 * i + 2] = r.v[2]; }
  }
}

// tq = (t xyz, q wxyz), doubles.
void hgo_interpolate_transform(const double* a, const double* b, double factor, double* out) {
  Rigid3<double> A{{a[0], a[1], a[2]}, {a[3], a[4], a[5], a[6]}};
  Rigid3<double> B{{b[0], b[1], b[2]}, {b[3], b[4], b[5], b[6]}};
  const Rigid3<double> r = InterpolateTransform(A, B, factor);
  out[0] = r.t.x; out[1] = r.t.y; out[2] = r.t.z;
  out[3] = r.q.w; out[4] = r.q.x; out[5] = r.q.y; out[6] = r.q.z;
}

// Per-point unwarping (oltb.cc:1331-1379). control: K x (time in `control_times`, tq in `control_poses`);
// clouds: times, begin offsets (n_clouds + 1) into `points` (n x 4), origins (n_clouds x 3). Writes the
// accumulated range data in the tracking frame: xyz_out (n x 3), origin_out[3]. Returns 1 when every return's
// time lies inside the control points, else 0 (the reference CHECK-fails).
int hgo_unwarp_range_data(const long long* control_times, const double* control_poses, int n_control,
                          const long long* cloud_times, const unsigned long long* cloud_offsets,
                          const float* cloud_origins, int n_clouds, const float* points, float* xyz_out,
                          float* origin_out) {
  std::vector<ControlPoint> cps(n_control);
  for (int k = 0; k < n_control; ++k) {
    const double* a = control_poses + 7 * k;
    cps[k].time = control_times[k];
    cps[k].pose = Rigid3<double>{{a[0], a[1], a[2]}, {a[3], a[4], a[5], a[6]}};
  }
  std::vector<TimedCloud> clouds(n_clouds);
  for (int c = 0; c < n_clouds; ++c) {
    clouds[c].time = cloud_times[c];
    clouds[c].origin = Vec3f{cloud_origins[3 * c], cloud_origins[3 * c + 1], cloud_origins[3 * c + 2]};
    clouds[c].points = points + 4 * cloud_offsets[c];
    clouds[c].n = static_cast<size_t>(cloud_offsets[c + 1] - cloud_offsets[c]);
  }
  const UnwarpedRangeData r = UnwarpAccumulatedRangeData(cps, clouds);
  std::memcpy(xyz_out, r.returns.data(), r.returns.size() * sizeof(float));
  origin_out[0] = r.origin.x; origin_out[1] = r.origin.y; origin_out[2] = r.origin.z;
  return r.time_in_range ? 1 : 0;
}

// sensor::TransformRangeData / TransformTimedRangeData on the returns (range_data.cc:25-39): float Rigid3f.
void hgo_transform_points(const float* tq, const float* in, size_t n, float* out) {
  const Rigid3<float> T{{tq[0], tq[1], tq[2]}, {tq[3], tq[4], tq[5], tq[6]}};
  TransformPointsF(T, in, n, out);
}

void hgo_quaternion_plus(const double* q, const double* delta, double* out) {
  QuaternionPlus(q, delta, out);
}

// ---- problem --------------------------------------------------------------
void* hgo_problem_create() { return new Problem(); }
void hgo_problem_destroy(void* p) { delete static_cast<Problem*>(p); }
int hgo_problem_add_pose(void* p, const double* tq, int constant) {
  PoseBlock b;
  for (int k = 0; k < 3; ++k) b.t[k] = tq[k];
  for (int k = 0; k < 4; ++k) b.q[k] = tq[3 + k];
  b.constant = constant != 0;
  auto* P = static_cast<Problem*>(p);
  P->poses.push_back(b);
  return static_cast<int>(P->poses.size()) - 1;
}
void hgo_problem_set_pose(void* p, int idx, const double* tq) {
  PoseBlock& b = static_cast<Problem*>(p)->poses[idx];
  for (int k = 0; k < 3; ++k) b.t[k] = tq[k];
  for (int k = 0; k < 4; ++k) b.q[k] = tq[3 + k];
}
void hgo_problem_get_pose(void* p, int idx, double* tq) {
  const PoseBlock& b = static_cast<Problem*>(p)->poses[idx];
  for (int k = 0; k < 3; ++k) tq[k] = b.t[k];
  for (int k = 0; k < 4; ++k) tq[3 + k] = b.q[k];
}
void hgo_problem_set_velocity(void* p, int idx, const double* v, int constant) {
  PoseBlock& b = static_cast<Problem*>(p)->poses[idx];
  for (int k = 0; k < 3; ++k) b.v[k] = v[k];
  b.has_velocity = true;
  b.v_constant = constant != 0;
}
void hgo_problem_get_velocity(void* p, int idx, double* v) {
  const PoseBlock& b = static_cast<Problem*>(p)->poses[idx];
  for (int k = 0; k < 3; ++k) v[k] = b.v[k];
}
// RelativeTranslationAndYawCostFunction(translation_w, rotation_w, delta_pose) between a and b
int hgo_problem_add_odometry_block(void* p, int a, int b, double wt, double wr, const double* delta_tq) {
  SmallBlock sb;
  sb.type = 1; sb.a = a; sb.b = b; sb.w[0] = wt; sb.w[1] = wr;
  for (int k = 0; k < 7; ++k) sb.delta[k] = delta_tq[k];
  auto* P = static_cast<Problem*>(p);
  P->small_blocks.push_back(sb);
  return static_cast<int>(P->small_blocks.size()) - 1;
}
// PredictionImuPreintegrationCostFunctor(translation_w, velocity_w, rotation_w, dt, delta_rotation)
int hgo_problem_add_imu_block(void* p, int a, int b, double wt, double wv, double wr, double dt,
                              const double* delta_q_wxyz) {
  SmallBlock sb;
  sb.type = 2; sb.a = a; sb.b = b; sb.w[0] = wt; sb.w[1] = wv; sb.w[2] = wr; sb.dt = dt;
  for (int k = 0; k < 4; ++k) sb.delta[3 + k] = delta_q_wxyz[k];
  auto* P = static_cast<Problem*>(p);
  P->small_blocks.push_back(sb);
  return static_cast<int>(P->small_blocks.size()) - 1;
}
// points must stay alive while the problem is used (the reference functors
// hold references too).
int hgo_problem_add_block(void* p, const float* xyz, size_t n, void* const* grids, int levels,
                          int multi_res, double scaling_factor, int pose_a, int pose_b,
                          double interpolation_ratio) {
  ResidualBlock b;
  b.points = xyz;
  b.n = n;
  for (int l = 0; l < levels; ++l) b.pyramid.push_back(static_cast<const HybridGridTSDF*>(grids[l]));
  b.multi_res = multi_res != 0;
  b.scaling_factor = scaling_factor;
  b.pose_a = pose_a;
  b.pose_b = pose_b;
  b.interpolation_ratio = interpolation_ratio;
  auto* P = static_cast<Problem*>(p);
  P->blocks.push_back(b);
  return static_cast<int>(P->blocks.size()) - 1;
}
int hgo_problem_num_residuals(void* p) { return static_cast<Problem*>(p)->NumResiduals(); }
int hgo_problem_num_columns(void* p) { return static_cast<Problem*>(p)->NumEffectiveParameters(); }
// Any of residuals / jacobian / gradient may be NULL. jacobian != NULL selects
// the Jet evaluation path, else the double path (as Ceres does).
void hgo_problem_evaluate(void* p, double* cost, double* residuals, double* jacobian,
                          double* gradient) {
  auto* P = static_cast<Problem*>(p);
  std::vector<double> jtmp;
  if (gradient && !jacobian) {
    jtmp.resize(static_cast<size_t>(P->NumResiduals()) * P->NumEffectiveParameters());
    jacobian = jtmp.data();
  }
  P->Evaluate(P->poses, cost, residuals, jacobian, gradient);
}
void hgo_problem_lookup_stats(void* p, uint64_t* out2) {
  auto* P = static_cast<Problem*>(p);
  out2[0] = P->lookup_stats.lookups;
  out2[1] = P->lookup_stats.levels_probed;
}
void hgo_solver_default_opts(hgo_solver_opts* o) {
  SolverOptions d;
  o->max_num_iterations = d.max_num_iterations;
  o->jacobi_scaling = d.jacobi_scaling ? 1 : 0;
  o->initial_trust_region_radius = d.initial_trust_region_radius;
  o->max_trust_region_radius = d.max_trust_region_radius;
  o->min_trust_region_radius = d.min_trust_region_radius;
  o->min_relative_decrease = d.min_relative_decrease;
  o->min_lm_diagonal = d.min_lm_diagonal;
  o->max_lm_diagonal = d.max_lm_diagonal;
  o->function_tolerance = d.function_tolerance;
  o->gradient_tolerance = d.gradient_tolerance;
  o->parameter_tolerance = d.parameter_tolerance;
}
void hgo_problem_solve(void* p, const hgo_solver_opts* o, hgo_solver_summary* s) {
  SolverOptions opt;
  if (o) {
    opt.max_num_iterations = o->max_num_iterations;
    opt.jacobi_scaling = o->jacobi_scaling != 0;
    opt.initial_trust_region_radius = o->initial_trust_region_radius;
    opt.max_trust_region_radius = o->max_trust_region_radius;
    opt.min_trust_region_radius = o->min_trust_region_radius;
    opt.min_relative_decrease = o->min_relative_decrease;
    opt.min_lm_diagonal = o->min_lm_diagonal;
    opt.max_lm_diagonal = o->max_lm_diagonal;
    opt.function_tolerance = o->function_tolerance;
    opt.gradient_tolerance = o->gradient_tolerance;
    opt.parameter_tolerance = o->parameter_tolerance;
  }
  const SolverSummary r = static_cast<Problem*>(p)->Solve(opt);
  if (s) {
    s->initial_cost = r.initial_cost;
    s->final_cost = r.final_cost;
    s->final_radius = r.final_radius;
    s->num_iterations = r.num_iterations;
    s->num_successful_steps = r.num_successful_steps;
    s->num_unsuccessful_steps = r.num_unsuccessful_steps;
    s->num_cost_evaluations = r.num_cost_evaluations;
    s->num_jacobian_evaluations = r.num_jacobian_evaluations;
    s->termination_type = r.termination_type;
    s->termination_reason = r.termination_reason;
    s->reserved = 0;
  }
}

}  // extern "C"
