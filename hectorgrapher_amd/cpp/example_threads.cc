// example_threads — one host thread per trajectory in ONE process, each with its own hg_ctx (the
// reference's deployment: a thread per trajectory builder): registration steps (match + exact insert)
// of 100k-point scans into three TSDFs, T threads at a time. Prints scans/s for T = 1, 2, 4 and the
// gain over one thread. Build: see __graft_entry__.build().
#include <hip/hip_runtime_api.h>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "hg_adapter.h"

namespace {
using hg_amd::Check;

// Box room [-6, 6] x [-5, 5] x [-1.5, 2.5] seen from (sx, sy, 0): 50 rings x 2000 columns, tracking frame.
std::vector<float> RoomScan(float sx, float sy) {
  std::vector<float> xyz;
  xyz.reserve(3 * 100000);
  for (int r = 0; r < 50; ++r)
    for (int c = 0; c < 2000; ++c) {
      const float az = 6.2831853f * (c + 0.37f) / 2000.f, el = (-12.f + 0.5f * r) * 0.01745329f;
      const float d[3] = {std::cos(el) * std::cos(az), std::cos(el) * std::sin(az), std::sin(el)};
      const float lo[3] = {-6.f - sx, -5.f - sy, -1.5f}, hi[3] = {6.f - sx, 5.f - sy, 2.5f};
      float t = 1e9f;
      for (int a = 0; a < 3; ++a) {
        if (d[a] > 1e-6f) t = std::fmin(t, hi[a] / d[a]);
        if (d[a] < -1e-6f) t = std::fmin(t, lo[a] / d[a]);
      }
      xyz.push_back(d[0] * t);
      xyz.push_back(d[1] * t);
      xyz.push_back(d[2] * t);
    }
  return xyz;
}

struct Trajectory {
  hg_ctx* ctx = nullptr;
  hg_grid* grids[3] = {nullptr, nullptr, nullptr};
  hg_problem* problem = nullptr;
  std::vector<float*> scans;  // device
  std::vector<double> x;      // sensor x per scan
  double y = 0.0;
};

void HipOk(hipError_t e) {
  if (e != hipSuccess) { std::fprintf(stderr, "hip: %s\n", hipGetErrorString(e)); std::exit(2); }
}

void Setup(Trajectory* t, int index, int steps) {
  Check(hg_ctx_create(0, nullptr, &t->ctx), "hg_ctx_create");
  const float res[3] = {0.05f, 0.10f, 0.20f};
  for (int l = 0; l < 3; ++l) Check(hg_grid_create(t->ctx, res[l], 2.5f, 1000.f, 1u << 16, &t->grids[l]), "hg_grid_create");
  Check(hg_problem_create(t->ctx, &t->problem), "hg_problem_create");
  t->y = 0.3 * index;
  const hg_insert_opts io = hg_amd::mapping::DefaultTSDFInserterOptions();
  const hg_insert_opts opts[3] = {io, io, io};
  for (int k = 0; k < steps + 3; ++k) {
    const double sx = 0.02 * k;
    const std::vector<float> h = RoomScan(static_cast<float>(sx), static_cast<float>(t->y));
    float* d = nullptr;
    HipOk(hipMalloc(reinterpret_cast<void**>(&d), h.size() * sizeof(float)));
    HipOk(hipMemcpy(d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    t->scans.push_back(d);
    t->x.push_back(sx);
    if (k < 3) {  // the map the first registration starts from
      const float origin[3] = {0.f, 0.f, 0.f};
      const float pose[7] = {static_cast<float>(sx), static_cast<float>(t->y), 0.f, 1.f, 0.f, 0.f, 0.f};
      Check(hg_pyramid_insert(t->grids, opts, 3, origin, d, h.size() / 3, 2000, pose, HG_INSERT_EXACT, HG_DEVICE, nullptr),
            "hg_pyramid_insert");
    }
  }
  Check(hg_ctx_synchronize(t->ctx), "hg_ctx_synchronize");
}

double Run(Trajectory* t, int steps, std::atomic<int>* ready, std::atomic<bool>* go, double* seconds) {
  const hg_insert_opts io = hg_amd::mapping::DefaultTSDFInserterOptions();
  const hg_insert_opts opts[3] = {io, io, io};
  hg_solver_opts so;
  hg_solver_default_opts(&so);
  const float origin[3] = {0.f, 0.f, 0.f};
  double max_err = 0.0;
  ready->fetch_add(1);
  while (!go->load()) {}
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 3; k < steps + 3; ++k) {
    // guess: the true pose displaced by 3 cm / 0.01 rad
    const double guess[7] = {t->x[k] + 0.03, t->y - 0.02, 0.01, std::cos(0.005), 0.0, 0.0, std::sin(0.005)};
    Check(hg_problem_reset(t->problem), "hg_problem_reset");
    const int pi = hg_problem_add_pose(t->problem, guess, 0);
    Check(hg_problem_add_block(t->problem, t->scans[k], 100000, HG_DEVICE, t->grids, 3, 1, 1.0 / std::sqrt(100000.0), pi, -1, 0.0),
          "hg_problem_add_block");
    double pose[7];
    Check(hg_register_scan(t->problem, &so, pi, t->grids, opts, 3, origin, t->scans[k], 100000, 2000, HG_DEVICE, pose, nullptr),
          "hg_register_scan");
    max_err = std::fmax(max_err, std::fabs(pose[0] - t->x[k]) + std::fabs(pose[1] - t->y));
  }
  Check(hg_ctx_synchronize(t->ctx), "hg_ctx_synchronize");
  *seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  return max_err;
}
}  // namespace

int main(int argc, char** argv) {
  const int steps = argc > 1 ? std::atoi(argv[1]) : 60;
  try {
    double base = 0.0;
    for (int threads : {1, 2, 3, 4}) {
      std::vector<Trajectory> tr(threads);
      for (int i = 0; i < threads; ++i) Setup(&tr[i], i, steps);
      std::atomic<int> ready{0};
      std::atomic<bool> go{false};
      std::vector<double> err(threads, 0.0), secs(threads, 0.0);
      std::vector<std::thread> pool;
      for (int i = 0; i < threads; ++i)
        pool.emplace_back([&, i] { err[i] = Run(&tr[i], steps, &ready, &go, &secs[i]); });
      while (ready.load() < threads) {}
      const auto t0 = std::chrono::steady_clock::now();
      go.store(true);
      for (std::thread& th : pool) th.join();
      const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      const double rate = threads * steps / sec;
      if (threads == 1) base = rate;
      double worst = 0.0;
      for (double e : err) worst = std::fmax(worst, e);
      std::printf("threads %d: %.0f scans/s (%.3f ms per step and thread), gain %.2f, max pose error %.4f m; ms per step by thread:", threads,
                  rate, sec / steps * 1e3, rate / base, worst);
      for (double ts : secs) std::printf(" %.3f", ts / steps * 1e3);
      std::printf("\n");
      for (Trajectory& t : tr) {
        for (float* d : t.scans) (void)hipFree(d);
        hg_problem_destroy(t.problem);
        for (hg_grid* g : t.grids) hg_grid_destroy(g);
        hg_ctx_destroy(t.ctx);
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  return 0;
}
