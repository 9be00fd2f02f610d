// example_oltb.cc — drives hg_amd::mapping::OptimizingLocalTrajectoryBuilder (hg_adapter.h: the reference's own
// window shape, configuration_files/trajectory_builder_3d.lua defaults with TSDF grids) over a deterministic sensor
// stream -- IMU at 100 Hz, odometry at 50 Hz, lidar at 20 Hz with scan times that do NOT fall on control points --
// and writes (1) every sensor message it was fed, bit for bit, and the range data of every insertion to a binary
// file and (2) the window after every step (control point times and states with full precision, the solver
// summary, the TSDF blocks of the solve) to stdout. tests/test_gpu_cpp_oltb.py replays the same messages through an
// independent Python statement of optimizing_local_trajectory_builder.cc over the CPU oracle.
// Usage: example_oltb <dump file> <mode> [scans]
//   mode 0: Lua defaults (CONSTANT sampling, two single-resolution blocks per scan, adaptive odometry weights)
//   mode 1: SYNCED_WITH_RANGE_DATA sampling (clouds sit ON control points: single-pose blocks)
//   mode 2: ADAPTIVE sampling + use_multi_resolution_matching
//   mode 3: CONSTANT sampling + use_per_point_unwarping
//   mode 4: mode 3 with SetMapUpdateEnabled(false) from scan 30 on (the window keeps running on a frozen map)
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include "hg_adapter.h"

using namespace hg_amd;

namespace {

// the sensor's true motion: at rest for 0.4 s, then 0.35 m/s along x with a slow yaw and a small sway
void TruePose(double t, double* x, double* y, double* z, double* yaw) {
  const double s = t > 0.4 ? t - 0.4 : 0.0;
  *x = 0.35 * s;
  *y = 0.2 + 0.03 * std::sin(1.3 * s);
  *z = 0.01 * s;
  *yaw = 0.08 * s;
}

// box room [-5, 6] x [-4, 3] x [-1.2, 2.6] with a pillar, 12 rings x 240 columns, swept over the 40 ms before the
// scan's time stamp (point times <= 0, the sensor moves while it sweeps)
int g_rings = 12, g_cols = 240;  // (argv[4], argv[5]: BASELINE configs[0] is 16 x 625)
sensor::TimedPointCloudData MakeScan(common::Time time) {
  sensor::TimedPointCloudData scan;
  scan.time = time;
  scan.width = static_cast<size_t>(g_rings);
  for (int c = 0; c < g_cols; ++c) {
    const float point_time = -0.04f * (1.0f - static_cast<float>(c) / static_cast<float>(g_cols - 1));
    double sx, sy, sz, yaw;
    TruePose(common::ToSeconds(time) + static_cast<double>(point_time), &sx, &sy, &sz, &yaw);
    for (int r = 0; r < g_rings; ++r) {
      const double el_deg = g_rings == 12 ? -14.0 + 2.5 * r : -15.0 + 30.0 * r / (g_rings - 1);
      const double az = 6.283185307179586 * c / static_cast<double>(g_cols) + yaw, el = el_deg * 0.017453292519943295;
      const double d[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
      double t = 1e9;
      const double o[3] = {sx, sy, sz};
      const double lo[3] = {-5.0, -4.0, -1.2}, hi[3] = {6.0, 3.0, 2.6};
      for (int a = 0; a < 3; ++a) {
        if (d[a] > 1e-9) t = std::fmin(t, (hi[a] - o[a]) / d[a]);
        if (d[a] < -1e-9) t = std::fmin(t, (lo[a] - o[a]) / d[a]);
      }
      {
        const double px = o[0] - 2.5, py = o[1] + 1.3;
        const double A = d[0] * d[0] + d[1] * d[1], B = 2.0 * (px * d[0] + py * d[1]), C = px * px + py * py - 0.16;
        const double disc = B * B - 4.0 * A * C;
        if (A > 1e-12 && disc > 0.0) {
          const double s0 = (-B - std::sqrt(disc)) / (2.0 * A);
          if (s0 > 0.0) t = std::fmin(t, s0);
        }
      }
      const double cs = std::cos(-yaw), sn = std::sin(-yaw);
      const double wx = d[0] * t, wy = d[1] * t, wz = d[2] * t;
      std::array<float, 4> p{{static_cast<float>(cs * wx - sn * wy), static_cast<float>(sn * wx + cs * wy),
                              static_cast<float>(wz), point_time}};
      if ((c * g_rings + r) % 97 == 13) p[1] = std::nanf("");  // a few invalid returns, as a real driver delivers them
      scan.ranges.push_back(p);
    }
  }
  return scan;
}

void PrintWindow(const mapping::OptimizingLocalTrajectoryBuilder& b) {
  for (const auto& cp : b.control_points()) {
    const mapping::State& s = cp.state;
    std::printf("  cp %lld pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g vel %.17g %.17g %.17g\n", static_cast<long long>(cp.time),
                s.translation[0], s.translation[1], s.translation[2], s.rotation[0], s.rotation[1], s.rotation[2], s.rotation[3],
                s.velocity[0], s.velocity[1], s.velocity[2]);
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 3) return 2;
  const int mode = std::atoi(argv[2]);
  const int scans = argc > 3 ? std::atoi(argv[3]) : 36;
  if (argc > 5) { g_rings = std::max(2, std::atoi(argv[4])); g_cols = std::max(2, std::atoi(argv[5])); }
  const bool quiet = argc > 6 && std::atoi(argv[6]) != 0;  // no window dump per step (timed runs)
  double add_range_seconds = 0.0;
  std::FILE* f = std::fopen(argv[1], "wb");
  if (!f) return 2;
  try {
    Context ctx(0);
    mapping::OptimizingLocalTrajectoryBuilder::Options opt;  // trajectory_builder_3d.lua defaults, grid_type = "TSDF"
    opt.initialization_duration = 0.2;                       // (3 s in the Lua file: a shorter wait keeps the run short)
    opt.submaps.num_range_data = 4;                          // (160: the motion filter lets a dozen insertions through; two live submaps)
    opt.submaps.max_blocks = 1u << 14;
    if (mode == 1) opt.control_point_sampling = mapping::OptimizingLocalTrajectoryBuilder::SYNCED_WITH_RANGE_DATA;
    if (mode == 2) {
      opt.control_point_sampling = mapping::OptimizingLocalTrajectoryBuilder::ADAPTIVE;
      opt.use_multi_resolution_matching = true;
      opt.sampling_max_delta_translation = 0.03;  // (the stream moves 0.35 m/s: translation, not time, places the points)
    }
    if (mode == 3 || mode == 4) opt.use_per_point_unwarping = true;
    mapping::OptimizingLocalTrajectoryBuilder builder(&ctx, opt);
    std::fwrite(&mode, sizeof(int), 1, f);
    // the message stream, time-ordered: kind 0 = IMU, 1 = odometry, 2 = scan
    const common::Time end = 130000 + 500000 * static_cast<common::Time>(scans);
    common::Time next_imu = 0, next_odom = 50000, next_scan = 130000;
    int scan_index = 0;
    while (true) {
      const common::Time t = std::min(next_imu, std::min(next_odom, next_scan));
      if (t > end || scan_index >= scans) break;
      if (t == next_imu) {
        double x, y, z, yaw0, yaw1;
        TruePose(common::ToSeconds(t), &x, &y, &z, &yaw0);
        TruePose(common::ToSeconds(t) + 0.01, &x, &y, &z, &yaw1);
        sensor::ImuData imu;
        imu.time = t;
        imu.linear_acceleration = {{0.0, 0.0, 9.80665}};
        imu.angular_velocity = {{0.0003, -0.0002, (yaw1 - yaw0) / 0.01 + 0.002}};  // a gyro with a small bias
        const int kind = 0;
        std::fwrite(&kind, sizeof(int), 1, f);
        std::fwrite(&imu.time, sizeof(common::Time), 1, f);
        std::fwrite(imu.angular_velocity.data(), sizeof(double), 3, f);
        builder.AddImuData(imu);
        next_imu += 100000;
        continue;
      }
      if (t == next_odom) {
        double x, y, z, yaw;
        TruePose(common::ToSeconds(t), &x, &y, &z, &yaw);
        const long long tick = t / 200000;
        const double err = (tick & 1) ? 0.004 : -0.003;  // odometry with an alternating error and a slow drift
        sensor::OdometryData odom;
        odom.time = t;
        odom.pose = Pose{{x + err + 0.002 * common::ToSeconds(t), y - 0.2 + 0.0007 * tick, z, std::cos(0.5 * (yaw + 0.0005 * tick)), 0.0, 0.0,
                          std::sin(0.5 * (yaw + 0.0005 * tick))}};
        const int kind = 1;
        std::fwrite(&kind, sizeof(int), 1, f);
        std::fwrite(&odom.time, sizeof(common::Time), 1, f);
        std::fwrite(odom.pose.data(), sizeof(double), 7, f);
        builder.AddOdometryData(odom);
        next_odom += 200000;
        continue;
      }
      const sensor::TimedPointCloudData scan = MakeScan(t);
      {
        const int kind = 2, n = static_cast<int>(scan.ranges.size());
        std::fwrite(&kind, sizeof(int), 1, f);
        std::fwrite(&scan.time, sizeof(common::Time), 1, f);
        std::fwrite(scan.origin.data(), sizeof(float), 3, f);
        std::fwrite(&n, sizeof(int), 1, f);
        std::fwrite(scan.ranges.data(), sizeof(float) * 4, n, f);
      }
      const int solves_before = builder.num_optimizations(), inserts_before = builder.num_insertions();
      if (mode == 4 && scan_index == 30) builder.SetMapUpdateEnabled(false);
      const auto t_add = std::chrono::steady_clock::now();
      auto result = builder.AddRangeData("lidar", scan);
      add_range_seconds += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_add).count();
      const hg_solver_summary& s = builder.last_summary();
      const int solved = builder.num_optimizations() - solves_before;
      std::printf("scan %d time %lld result %d solved %d iterations %d termination %d %d queued %zu imu_blocks %d odometry_blocks %d residuals %d\n", scan_index,
                  static_cast<long long>(scan.time), result ? 1 : 0, solved, s.num_iterations, s.termination_type, s.termination_reason,
                  builder.num_queued_clouds(), builder.last_imu_blocks(), builder.last_odometry_blocks(), solved ? builder.last_num_residuals() : 0);
      if (solved)
        for (const auto& b : builder.last_blocks())
          std::printf("  block %zu %d %d %.17g %d\n", b.points, b.pose_a, b.pose_b, b.factor, b.grid);
      if (!quiet) PrintWindow(builder);
      // the range data of this step's insertion (kind 3), as it went into the submaps
      int inserted = 0;
      if (result) {
        std::printf("  local_pose %lld %.17g %.17g %.17g %.17g %.17g %.17g %.17g inserted %d submaps %zu\n", static_cast<long long>(result->time),
                    result->local_pose[0], result->local_pose[1], result->local_pose[2], result->local_pose[3], result->local_pose[4],
                    result->local_pose[5], result->local_pose[6], builder.num_insertions() - inserts_before,
                    result->insertion_result ? result->insertion_result->insertion_submaps.size() : size_t(0));
        inserted = builder.num_insertions() - inserts_before;
      }
      {
        const int kind = 3, n = inserted ? static_cast<int>(result->range_data_in_local.returns.size()) : 0;
        std::fwrite(&kind, sizeof(int), 1, f);
        std::fwrite(&n, sizeof(int), 1, f);
        if (n) {
          std::fwrite(result->range_data_in_local.origin.data(), sizeof(float), 3, f);
          std::fwrite(result->range_data_in_local.returns.data(), sizeof(float) * 3, n, f);
        }
      }
      ++scan_index;
      next_scan += 500000;
    }
    // the live submaps at the end: local pose and both grids' voxels (kind 4)
    const auto& submaps = builder.active_submaps().submaps();
    for (const auto& submap : submaps) {
      const int kind = 4;
      std::fwrite(&kind, sizeof(int), 1, f);
      std::fwrite(submap->local_pose().data(), sizeof(double), 7, f);
      const int num = submap->num_range_data();
      std::fwrite(&num, sizeof(int), 1, f);
      for (mapping::HybridGridTSDF* g : {&submap->high_resolution_hybrid_grid(), &submap->low_resolution_hybrid_grid()}) {
        std::vector<std::array<int, 3>> cells;
        std::vector<uint16_t> tsd, weight;
        const int n = static_cast<int>(g->Export(&cells, &tsd, &weight));
        std::fwrite(&n, sizeof(int), 1, f);
        std::fwrite(cells.data(), sizeof(int) * 3, n, f);
        std::fwrite(tsd.data(), sizeof(uint16_t), n, f);
        std::fwrite(weight.data(), sizeof(uint16_t), n, f);
      }
    }
    std::printf("done: %d optimizations, %d insertions, %zu submaps\n", builder.num_optimizations(), builder.num_insertions(), submaps.size());
    std::printf("timing: %d scans of %d returns, %.6f s inside AddRangeData\n", scan_index, g_rings * g_cols, add_range_seconds);
  } catch (const Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    std::fclose(f);
    return 1;
  }
  std::fclose(f);
  return 0;
}
