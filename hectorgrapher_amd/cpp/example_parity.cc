// example_parity.cc — runs the simplified sliding-window builder (SlidingWindowTrajectoryBuilder) of the adapter (hg_adapter.h) over a
// deterministic trajectory and writes (1) every input it was fed, bit for bit, to a binary file and (2) the
// window's solved poses, velocities and solver summaries with full precision to stdout, so that a test can
// replay the same inputs through the CPU oracle with an independent statement of the window wiring
// (tests/test_gpu_cpp_adapter.py::test_cpp_window_builder_against_oracle).
// Usage: example_parity <dump file> [scans]
#include <cstdio>
#include <cstdlib>

#include "hg_adapter.h"

int main(int argc, char** argv) {
  using namespace hg_amd;
  if (argc < 2) return 2;
  const int scans = argc > 2 ? std::atoi(argv[2]) : 9;
  std::FILE* f = std::fopen(argv[1], "wb");
  if (!f) return 2;
  try {
    Context ctx(0);
    mapping::SlidingWindowTrajectoryBuilder::Options opt;
    opt.window = 4;
    opt.resolutions = {0.10f, 0.20f};
    opt.max_blocks = 1u << 15;
    opt.imu_translation_weight = 1.0;
    opt.imu_velocity_weight = 0.05;
    opt.imu_rotation_weight = 2.0;
    opt.odometry_translation_weight = 3.0;
    opt.odometry_rotation_weight = 5.0;
    mapping::SlidingWindowTrajectoryBuilder builder(&ctx, opt);
    std::fwrite(&scans, sizeof(int), 1, f);
    for (int k = 0; k < scans; ++k) {
      // a box room with a pillar, seen from a sensor that rests for two scans, then moves 4 cm per scan along
      // x and yaws 5 mrad per scan
      const double step = k > 1 ? k - 1 : 0;
      const double sx = 0.04 * step, yaw = 0.005 * step;
      sensor::TimedPointCloudData scan;
      scan.time = 1000000 * static_cast<common::Time>(k);  // 0.1 s per scan, in ticks
      for (int c = 0; c < 240; ++c)
        for (int r = 0; r < 12; ++r) {
          const double az = 6.283185307179586 * c / 240.0 + yaw, el = (-14.0 + 2.5 * r) * 0.017453292519943295;
          const double d[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
          double t = 1e9;  // room [-5, 6] x [-4, 3] x [-1.2, 2.6] seen from (sx, 0.2, 0)
          const double o[3] = {sx, 0.2, 0.0};
          const double lo[3] = {-5.0, -4.0, -1.2}, hi[3] = {6.0, 3.0, 2.6};
          for (int a = 0; a < 3; ++a) {
            if (d[a] > 1e-9) t = std::fmin(t, (hi[a] - o[a]) / d[a]);
            if (d[a] < -1e-9) t = std::fmin(t, (lo[a] - o[a]) / d[a]);
          }
          // pillar: cylinder of radius 0.4 m at (2.5, -1.3)
          {
            const double px = o[0] - 2.5, py = o[1] + 1.3;
            const double A = d[0] * d[0] + d[1] * d[1], B = 2.0 * (px * d[0] + py * d[1]), C = px * px + py * py - 0.16;
            const double disc = B * B - 4.0 * A * C;
            if (A > 1e-12 && disc > 0.0) {
              const double s0 = (-B - std::sqrt(disc)) / (2.0 * A);
              if (s0 > 0.0) t = std::fmin(t, s0);
            }
          }
          // the return in the sensor frame (rotate the world direction back by the yaw)
          const double cs = std::cos(-yaw), sn = std::sin(-yaw);
          const double wx = d[0] * t, wy = d[1] * t, wz = d[2] * t;
          scan.ranges.push_back({{static_cast<float>(cs * wx - sn * wy), static_cast<float>(sn * wx + cs * wy),
                                  static_cast<float>(wz), 0.f}});
        }
      sensor::OdometryData odom;
      odom.time = scan.time;
      const double err = (k & 1) ? 0.008 : -0.006;  // odometry with an alternating error
      odom.pose = Pose{{sx + err, 0.2 + 0.0011 * k, 0.0005 * k, std::cos(0.5 * (yaw + 0.001)), 0.0, 0.0, std::sin(0.5 * (yaw + 0.001))}};
      builder.AddOdometryData(odom);
      std::vector<sensor::ImuData> imu_batch;
      for (int j = 0; j < 10; ++j) {
        sensor::ImuData imu;
        imu.time = scan.time - 1000000 + 100000 * (j + 1);
        imu.linear_acceleration = {{0.0, 0.0, 9.80665}};
        imu.angular_velocity = {{0.0002, -0.0001, k > 1 ? 0.05 : 0.0}};
        builder.AddImuData(imu);
        imu_batch.push_back(imu);
      }
      // the inputs of this step, bit for bit
      const int n = static_cast<int>(scan.ranges.size()), n_imu = static_cast<int>(imu_batch.size());
      std::fwrite(&scan.time, sizeof(common::Time), 1, f);  // ticks
      std::fwrite(&n, sizeof(int), 1, f);
      std::fwrite(scan.ranges.data(), sizeof(float) * 4, n, f);
      std::fwrite(odom.pose.data(), sizeof(double), 7, f);
      std::fwrite(&n_imu, sizeof(int), 1, f);
      for (const auto& s : imu_batch) {
        std::fwrite(&s.time, sizeof(common::Time), 1, f);
        std::fwrite(s.angular_velocity.data(), sizeof(double), 3, f);
      }
      const int solves_before = builder.num_solves();
      auto result = builder.AddRangeData("lidar", scan);
      if (!result) continue;
      // AFTER the step the window has dropped the scans it inserted; the line reports the window as it stands
      const hg_solver_summary& s = builder.last_summary();
      std::printf("step %d solved %d iterations %d termination %d %d window %zu\n", k, builder.num_solves() - solves_before,
                  s.num_iterations, s.termination_type, s.termination_reason, builder.window_size());
      if (result->insertion_result) {
        const Pose& p = builder.last_inserted_pose();
        std::printf("  inserted at %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
      }
      for (size_t i = 0; i < builder.window_size(); ++i) {
        const Pose& p = builder.pose(i);
        const std::array<double, 3> v = builder.velocity(i);
        std::printf("  cp %zu pose %.17g %.17g %.17g %.17g %.17g %.17g %.17g vel %.17g %.17g %.17g\n", i, p[0], p[1], p[2], p[3],
                    p[4], p[5], p[6], v[0], v[1], v[2]);
      }
    }
  } catch (const Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    std::fclose(f);
    return 1;
  }
  std::fclose(f);
  return 0;
}
