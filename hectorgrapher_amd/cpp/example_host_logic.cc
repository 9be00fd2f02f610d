// example_host_logic.cc — the host-only pieces of hg_adapter.h (no device call): InterpolateTransform,
// TransformInterpolationBuffer::{Lookup, LookupUntilDelta}, IntegrateImuDeltaRotation, over a deterministic odometry /
// IMU stream; prints with full precision. tests/test_host_logic.py compares with the oracle's InterpolateTransform
// and the Python statement of the same reference functions (tests/oltb_replay.py). Runs without a GPU.
#include <cstdio>

#include "hg_adapter.h"

using namespace hg_amd;

int main() {
  transform::TransformInterpolationBuffer buffer;
  std::deque<sensor::ImuData> imu;
  for (int k = 0; k < 40; ++k) {
    const double t = 0.02 * k, yaw = 0.3 * t * t, pitch = 0.05 * std::sin(3.0 * t);
    const double cy = std::cos(0.5 * yaw), sy = std::sin(0.5 * yaw), cp = std::cos(0.5 * pitch), sp = std::sin(0.5 * pitch);
    // q = q_yaw(z) * q_pitch(y)
    const Pose pose{{0.4 * t + 0.01 * std::sin(9.0 * t), 0.1 * t * t, 0.003 * k, cy * cp, -sy * sp, cy * sp, sy * cp}};
    buffer.Push(50000 + 200000 * static_cast<common::Time>(k), pose);
    std::printf("odom %lld %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", static_cast<long long>(50000 + 200000 * static_cast<common::Time>(k)),
                pose[0], pose[1], pose[2], pose[3], pose[4], pose[5], pose[6]);
  }
  for (int k = 0; k < 80; ++k) {
    sensor::ImuData s;
    s.time = 100000 * static_cast<common::Time>(k);
    s.linear_acceleration = {{0, 0, 9.8}};
    s.angular_velocity = {{0.02 * std::sin(0.3 * k), -0.01 + 0.001 * k, 0.5 * std::cos(0.11 * k)}};
    imu.push_back(s);
    std::printf("imu %lld %.17g %.17g %.17g\n", static_cast<long long>(s.time), s.angular_velocity[0], s.angular_velocity[1], s.angular_velocity[2]);
  }
  for (common::Time t : {50000ll, 1234567ll, 3333333ll, 7850000ll, 250000ll}) {
    const Pose p = buffer.Lookup(t);
    std::printf("lookup %lld %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", static_cast<long long>(t), p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
  }
  const double limits[4][3] = {{0.2, 0.1, 0.25}, {0.03, 0.1, 0.25}, {0.2, 0.004, 0.25}, {5.0, 5.0, 0.07}};
  for (const auto& l : limits)
    for (common::Time t : {50000ll, 1234567ll, 6000000ll}) {
      double tr = 0, rr = 0, dr = 0;
      const common::Time c = buffer.LookupUntilDelta(t, l[0], l[1], l[2], &tr, &rr, &dr);
      std::printf("until %lld %.17g %.17g %.17g -> %lld %.17g %.17g %.17g\n", static_cast<long long>(t), l[0], l[1], l[2], static_cast<long long>(c), tr, rr, dr);
    }
  const common::Time spans[4][2] = {{0, 1000000}, {123456, 2345678}, {3000000, 3050000}, {7000000, 9000000}};
  for (const auto& sp : spans) {
    const std::array<double, 4> q = mapping::IntegrateImuDeltaRotation(imu, sp[0], sp[1]);
    std::printf("imu_delta %lld %lld %.17g %.17g %.17g %.17g\n", static_cast<long long>(sp[0]), static_cast<long long>(sp[1]), q[0], q[1], q[2], q[3]);
  }
  return 0;
}
