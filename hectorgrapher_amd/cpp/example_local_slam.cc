// example_local_slam.cc — drives the C++ adapter the way GlobalTrajectoryBuilder drives a local
// trajectory builder (global_trajectory_builder.cc:52-85): AddRangeData per scan, read back poses.
// Build: g++ -std=c++11 -O2 example_local_slam.cc -L.. -lhg_mi355x -Wl,-rpath,'$ORIGIN/..' -o example_local_slam
#include <cstdio>

#include "hg_adapter.h"

int main() {
  using namespace hg_amd;
  try {
    Context ctx(0);
    mapping::LocalTrajectoryBuilder3D::Options options;
    mapping::LocalTrajectoryBuilder3D builder(&ctx, options);
    // a box room seen from a sensor moving along x: 16 rings x 360 columns
    for (int k = 0; k < 5; ++k) {
      sensor::TimedPointCloudData scan;
      scan.time = 1000000 * static_cast<common::Time>(k);  // 0.1 s per scan, in ticks
      const float sx = 0.05f * k;
      for (int c = 0; c < 360; ++c)
        for (int r = 0; r < 16; ++r) {
          const float az = 6.2831853f * c / 360.f, el = (-15.f + 2.f * r) * 0.01745329f;
          const float d[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
          float t = 1e9f;  // room [-5,5] x [-4,4] x [-1,3] seen from (sx, 0, 0)
          const float lo[3] = {-5.f - sx, -4.f, -1.f}, hi[3] = {5.f - sx, 4.f, 3.f};
          for (int a = 0; a < 3; ++a) {
            if (d[a] > 1e-6f) t = std::fmin(t, hi[a] / d[a]);
            if (d[a] < -1e-6f) t = std::fmin(t, lo[a] / d[a]);
          }
          scan.ranges.push_back({{d[0] * t, d[1] * t, d[2] * t, 0.f}});
        }
      auto result = builder.AddRangeData("lidar", scan);
      if (result) std::printf("scan %d pose %.4f %.4f %.4f\n", k, result->local_pose[0], result->local_pose[1], result->local_pose[2]);
    }
    // the same trajectory through the sliding-window builder (SlidingWindowTrajectoryBuilder
    // shape): window of 3 control points, odometry with an alternating +-1 cm error
    mapping::SlidingWindowTrajectoryBuilder::Options wopt;
    wopt.window = 3;
    // the sensor rests for the first two scans (the window's first state is fixed, with zero
    // velocity) and then moves at 0.5 m/s; the functor ties neighbouring velocities together (it reads
    // only the pre-integrated rotation), so the step in velocity gets a small velocity weight
    wopt.imu_translation_weight = 1.0;
    wopt.imu_velocity_weight = 0.01;
    wopt.imu_rotation_weight = 1.0;
    mapping::SlidingWindowTrajectoryBuilder wbuilder(&ctx, wopt);
    for (int k = 0; k < 8; ++k) {
      sensor::TimedPointCloudData scan;
      scan.time = 1000000 * static_cast<common::Time>(k);  // 0.1 s per scan, in ticks
      const float sx = 0.05f * (k > 0 ? k - 1 : 0);
      for (int c = 0; c < 360; ++c)
        for (int r = 0; r < 16; ++r) {
          const float az = 6.2831853f * c / 360.f, el = (-15.f + 2.f * r) * 0.01745329f;
          const float d[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
          float t = 1e9f;
          const float lo[3] = {-5.f - sx, -4.f, -1.f}, hi[3] = {5.f - sx, 4.f, 3.f};
          for (int a = 0; a < 3; ++a) {
            if (d[a] > 1e-6f) t = std::fmin(t, hi[a] / d[a]);
            if (d[a] < -1e-6f) t = std::fmin(t, lo[a] / d[a]);
          }
          scan.ranges.push_back({{d[0] * t, d[1] * t, d[2] * t, 0.f}});
        }
      sensor::OdometryData odom;
      odom.time = scan.time;
      // (a small lateral offset keeps the predictions off the exact symmetry axis of this synthetic
      // room, where returns sit exactly on voxel boundaries and the interpolated TSDF cost jumps)
      odom.pose = Pose{{0.05 * (k > 0 ? k - 1 : 0) + ((k & 1) ? 0.01 : -0.01), 0.0013 * (k + 1), 0.0007 * (k + 1), 1.0, 0.0, 0.0, 0.0}};
      wbuilder.AddOdometryData(odom);
      // a gyro at 100 Hz that reports no rotation (the trajectory is a straight line): the window's
      // control points are tied by PredictionImuPreintegrationCostFunctor blocks with velocity states
      for (int j = 0; j < 10; ++j) {
        sensor::ImuData imu;
        imu.time = scan.time - 1000000 + 100000 * (j + 1);
        imu.linear_acceleration = {{0.0, 0.0, 9.80665}};
        imu.angular_velocity = {{0.0, 0.0, 0.0}};
        wbuilder.AddImuData(imu);
      }
      auto result = wbuilder.AddRangeData("lidar", scan);
      if (result)
        std::printf("window scan %d pose %.4f %.4f %.4f (window %zu, imu blocks %d, v %.3f %.3f %.3f)\n", k,
                    result->local_pose[0], result->local_pose[1], result->local_pose[2], wbuilder.window_size(),
                    wbuilder.num_imu_blocks_in_last_solve(), wbuilder.velocity(wbuilder.window_size() - 1)[0],
                    wbuilder.velocity(wbuilder.window_size() - 1)[1], wbuilder.velocity(wbuilder.window_size() - 1)[2]);
    }
    // the pre-integration itself: 90 degrees about z in 1 s from a constant 100 Hz gyro
    {
      std::deque<sensor::ImuData> gyro;
      for (int j = 0; j <= 100; ++j) gyro.push_back({100000 * static_cast<common::Time>(j), {{0.0, 0.0, 9.8}}, {{0.0, 0.0, 1.5707963267948966}}});
      const std::array<double, 4> dq = mapping::IntegrateImuDeltaRotation(gyro, 0, common::FromSeconds(1.0));
      std::printf("imu delta rotation %.6f %.6f %.6f %.6f\n", dq[0], dq[1], dq[2], dq[3]);
    }
    // ActiveSubmaps3D: two live submaps, a new one every 3 insertions, the old one finished at 6
    mapping::ActiveSubmaps3D::Options sopt;
    sopt.num_range_data = 3;
    sopt.max_blocks = 1u << 14;
    mapping::ActiveSubmaps3D active(&ctx, sopt);
    for (int k = 0; k < 7; ++k) {
      sensor::RangeData rd;
      rd.origin = Point{{0.05f * k, 0.f, 0.f}};
      for (int c = 0; c < 360; ++c) {
        const float az = 6.2831853f * c / 360.f;
        for (int ring = -3; ring <= 3; ++ring)
          rd.returns.push_back(Point{{0.05f * k + 3.f * std::cos(az), 3.f * std::sin(az), 0.1f * ring}});
      }
      const auto& live = active.InsertData(rd, {{1.0, 0.0, 0.0, 0.0}});
      std::printf("submaps after insert %d:", k);
      for (const auto& sm : live) std::printf(" %d%s", sm->num_range_data(), sm->insertion_finished() ? "(finished)" : "");
      std::printf("\n");
    }
    // the filters in front of the matcher (trajectory_builder_3d.lua:21-27 defaults)
    {
      std::vector<Point> cloud;
      for (int c = 0; c < 3600; ++c) {
        const float az = 6.2831853f * c / 3600.f;
        for (int ring = -8; ring <= 8; ++ring)
          cloud.push_back(Point{{4.f * std::cos(az), 4.f * std::sin(az), 0.05f * ring}});
      }
      const std::vector<Point> coarse = sensor::VoxelFilter(&ctx, 0.15f).Filter(cloud);
      const std::vector<Point> matched = sensor::AdaptiveVoxelFilter(&ctx, 2.f, 150.f, 15.f).Filter(coarse);
      std::printf("filters: %zu -> %zu -> %zu points\n", cloud.size(), coarse.size(), matched.size());
    }
    // X-ray texture of the finished submap (Submap3D::ToResponseProto -> AddToTextureProto)
    {
      auto finished = active.submaps().front();
      const mapping::SubmapTexture tex = mapping::AddToTexture(finished->high_resolution_hybrid_grid(), finished->local_pose());
      size_t solid = 0;
      for (size_t i = 1; i < tex.cells.size(); i += 2) solid += tex.cells[i] != 0;
      std::printf("texture %d x %d, %zu pixels with alpha, slice pose %.3f %.3f %.3f\n", tex.width, tex.height, solid,
                  tex.slice_pose[0], tex.slice_pose[1], tex.slice_pose[2]);
    }
  } catch (const Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    return 1;
  }
  return 0;
}
