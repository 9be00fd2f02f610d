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
  // --- the reference's own known-answer tests for these two pieces, run through the adapter ---
  // transform/transform_interpolation_buffer_test.cc:29-45 (testHas)
  {
    transform::TransformInterpolationBuffer b;
    const Pose identity{{0, 0, 0, 1, 0, 0, 0}};
    std::printf("kat_has %d", b.Has(50) ? 1 : 0);
    b.Push(50, identity);
    std::printf(" %d %d %d", b.Has(25) ? 1 : 0, b.Has(50) ? 1 : 0, b.Has(75) ? 1 : 0);
    b.Push(100, identity);
    std::printf(" %d %d %d %d %d %lld %lld\n", b.Has(25) ? 1 : 0, b.Has(50) ? 1 : 0, b.Has(75) ? 1 : 0, b.Has(100) ? 1 : 0,
                b.Has(125) ? 1 : 0, static_cast<long long>(b.earliest_time()), static_cast<long long>(b.latest_time()));
  }
  // :47-65 (testLookup): identity at 50, translation (10, 10, 10) * Rz(2 rad) at 100, looked up at 75
  {
    transform::TransformInterpolationBuffer b;
    b.Push(50, Pose{{0, 0, 0, 1, 0, 0, 0}});
    b.Push(100, Pose{{10., 10., 10., std::cos(1.0), 0, 0, std::sin(1.0)}});
    const Pose p = b.Lookup(75);
    std::printf("kat_lookup %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
  }
  // :67-74 (testLookupSingleTransform)
  {
    transform::TransformInterpolationBuffer b;
    b.Push(75, Pose{{0, 0, 0, 1, 0, 0, 0}});
    const Pose p = b.Lookup(75);
    std::printf("kat_single %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", p[0], p[1], p[2], p[3], p[4], p[5], p[6]);
  }
  // mapping/internal/3d/imu_integration_test.cc:30-58 (ZeroIntegration): 101 zero samples one second apart
  {
    std::deque<sensor::ImuData> d;
    common::Time t = 0;
    d.push_back(sensor::ImuData{t, {{0, 0, 0}}, {{0, 0, 0}}});
    for (int i = 0; i < 100; ++i) {
      t += common::FromSeconds(1);
      d.push_back(sensor::ImuData{t, {{0, 0, 0}}, {{0, 0, 0}}});
    }
    size_t it = d.size() - 1;
    while (d[it].time > 0) --it;
    const mapping::IntegrateImuWithTranslationResult r = mapping::IntegrateImuWithTranslationEuler(d, 0, t, &it);
    std::printf("kat_imu_zero %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", r.delta_velocity[0], r.delta_velocity[1],
                r.delta_velocity[2], r.delta_translation[0], r.delta_translation[1], r.delta_translation[2], r.delta_rotation[0],
                r.delta_rotation[1], r.delta_rotation[2], r.delta_rotation[3]);
  }
  // :60-116 (ConstantAcceleration): (0, 0, 9.80665) at 100 Hz, integrated from 0 to every sample time
  {
    std::deque<sensor::ImuData> d;
    common::Time t = 0;
    d.push_back(sensor::ImuData{t, {{0, 0, 9.80665}}, {{0, 0, 0}}});
    for (int i = 0; i < 1000; ++i) {
      t += common::FromSeconds(0.01);
      d.push_back(sensor::ImuData{t, {{0, 0, 9.80665}}, {{0, 0, 0}}});
      size_t it = d.size() - 1;
      while (d[it].time > 0) --it;
      const mapping::IntegrateImuWithTranslationResult r = mapping::IntegrateImuWithTranslationEuler(d, 0, t, &it);
      std::printf("kat_imu_const %lld %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", static_cast<long long>(t),
                  r.delta_velocity[0], r.delta_velocity[1], r.delta_velocity[2], r.delta_translation[0], r.delta_translation[1],
                  r.delta_translation[2], r.delta_rotation[0], r.delta_rotation[1], r.delta_rotation[2], r.delta_rotation[3]);
    }
  }
  return 0;
}
