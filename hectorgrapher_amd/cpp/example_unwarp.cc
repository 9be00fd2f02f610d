// example_unwarp.cc — the adapter's InsertUnwarped (per-point unwarping of the clouds that leave the window, then
// insertion: optimizing_local_trajectory_builder.cc:1331-1379, :1437-1440, submap_3d.cc:436-437) on a small
// deterministic case. Writes every input bit for bit and the resulting voxels of both grids to a binary file, so
// that a test can replay the inputs through the CPU oracle (tests/test_gpu_cpp_adapter.py::
// test_cpp_insert_unwarped_against_oracle).
// Usage: example_unwarp <dump file>
#include <cstdio>
#include <cstdlib>

#include "hg_adapter.h"

int main(int argc, char** argv) {
  using namespace hg_amd;
  if (argc < 2) return 2;
  std::FILE* f = std::fopen(argv[1], "wb");
  if (!f) return 2;
  try {
    Context ctx(0);
    const float res[2] = {0.10f, 0.20f};
    mapping::HybridGridTSDF high(&ctx, res[0], 2.5f, 1000.f, 1u << 14), low(&ctx, res[1], 2.5f, 1000.f, 1u << 14);
    // four control points 50 ms apart on a gently curved path
    std::vector<Pose> poses;
    std::vector<common::Time> times;
    for (int k = 0; k < 4; ++k) {
      const double yaw = 0.02 * k, roll = 0.003 * k * k;
      const double cy = std::cos(0.5 * yaw), sy = std::sin(0.5 * yaw), cr = std::cos(0.5 * roll), sr = std::sin(0.5 * roll);
      // q = q_yaw(z) * q_roll(x)
      poses.push_back(Pose{{0.05 * k, 0.02 * k - 0.001 * k * k, 0.004 * k, cy * cr, cy * sr, sy * sr, sy * cr}});
      times.push_back(1000000000 + 500000 * static_cast<common::Time>(k));  // 100 s + 50 ms per control point, ticks
    }
    // two clouds of 16 rings x 180 columns swept over 70 ms each, seen from inside a box room; every 41st return NaN
    std::vector<sensor::TimedPointCloudData> clouds(2);
    const size_t rings = 16, cols = 180;
    for (int c = 0; c < 2; ++c) {
      clouds[c].time = 1000000000 + 100000 + 700000 * static_cast<common::Time>(c);
      clouds[c].origin = {{0.01f * c, -0.02f, 0.1f}};
      for (size_t col = 0; col < cols; ++col)
        for (size_t r = 0; r < rings; ++r) {
          const double az = 6.283185307179586 * col / cols + 0.3 * c, el = (-15.0 + 2.0 * r) * 0.017453292519943295;
          const double d[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
          double t = 1e9;
          const double lo[3] = {-4.0, -3.0, -1.0}, hi[3] = {5.0, 3.5, 2.2};
          for (int a = 0; a < 3; ++a) {
            if (d[a] > 1e-9) t = std::fmin(t, hi[a] / d[a]);
            if (d[a] < -1e-9) t = std::fmin(t, lo[a] / d[a]);
          }
          const size_t i = col * rings + r;
          float x = static_cast<float>(d[0] * t), y = static_cast<float>(d[1] * t), z = static_cast<float>(d[2] * t);
          if (i % 41 == 7) y = std::nanf("");
          clouds[c].ranges.push_back({{x, y, z, static_cast<float>(0.07 * i / (rings * cols))}});
        }
    }
    const std::array<float, 7> submap_from_local{{-0.3f, 0.2f, 0.05f, 0.9998f, 0.f, 0.f, 0.02f}};
    mapping::InsertUnwarped({&high, &low}, mapping::DefaultTSDFInserterOptions(), clouds, rings, poses, times, &submap_from_local);
    // inputs
    const int n_cp = static_cast<int>(poses.size()), n_clouds = static_cast<int>(clouds.size());
    std::fwrite(&n_cp, sizeof(int), 1, f);
    for (int k = 0; k < n_cp; ++k) {
      std::fwrite(&times[k], sizeof(common::Time), 1, f);
      std::fwrite(poses[k].data(), sizeof(double), 7, f);
    }
    std::fwrite(submap_from_local.data(), sizeof(float), 7, f);
    std::fwrite(&n_clouds, sizeof(int), 1, f);
    for (const auto& c : clouds) {
      const int n = static_cast<int>(c.ranges.size());
      std::fwrite(&c.time, sizeof(common::Time), 1, f);
      std::fwrite(c.origin.data(), sizeof(float), 3, f);
      std::fwrite(&n, sizeof(int), 1, f);
      std::fwrite(c.ranges.data(), sizeof(float) * 4, n, f);
    }
    // results: the voxels of both grids in the reference's iteration order
    for (mapping::HybridGridTSDF* g : {&high, &low}) {
      std::vector<std::array<int, 3>> cells;
      std::vector<uint16_t> tsd, weight;
      const int n = static_cast<int>(g->Export(&cells, &tsd, &weight));
      std::fwrite(&n, sizeof(int), 1, f);
      if (n) {
        std::fwrite(cells[0].data(), sizeof(int) * 3, n, f);
        std::fwrite(tsd.data(), sizeof(uint16_t), n, f);
        std::fwrite(weight.data(), sizeof(uint16_t), n, f);
      }
      std::printf("grid %.2f: %d voxels\n", g == &high ? res[0] : res[1], n);
    }
  } catch (const Error& e) {
    std::fprintf(stderr, "error %d: %s\n", e.code, e.what());
    std::fclose(f);
    return 1;
  }
  std::fclose(f);
  return 0;
}
