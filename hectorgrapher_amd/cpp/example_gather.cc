// example_gather — the multi-GPU exchange step driven from C++ (hg_gather.h).
//
//   example_gather pipe N        N rank processes on THIS machine (forked before any GPU call), all on
//                                device 0, payload through pipes: what the tests run on a 1-GPU box
//   example_gather rccl          one rank of an N-GPU job: RANK, WORLD_SIZE, LOCAL_RANK from the
//                                environment, the ncclUniqueId through the file HG_NCCL_ID_FILE
//                                (rank 0 writes it); payload GPU to GPU with ncclSend / ncclRecv
//
// Every rank maps its own small submap (three scans of a ring, shifted by the rank), then all pyramids
// are gathered to rank 0, imported into fresh grids and compared with the owners' exports.
#include <sys/wait.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <thread>

#include "hg_gather.h"

namespace {
namespace hg = hg_amd;
using hg_amd::mapping::HybridGridTSDF;

hg::sensor::RangeData RingScan(float cx, float cy) {
  hg::sensor::RangeData rd;
  rd.origin = {{cx, cy, 0.f}};
  for (int ring = -8; ring < 8; ++ring)
    for (int col = 0; col < 600; ++col) {
      const float az = 6.2831853f * static_cast<float>(col) / 600.f, el = 0.03f * static_cast<float>(ring);
      const float r = 3.f + 0.4f * std::sin(3.f * az);
      rd.returns.push_back({{cx + r * std::cos(az) * std::cos(el), cy + r * std::sin(az) * std::cos(el), r * std::sin(el)}});
    }
  rd.width = 600;
  return rd;
}

template <class Transport>
int RunRank(Transport& tr, int device) {
  hg::Context ctx(device);
  const float res[3] = {0.05f, 0.10f, 0.20f};
  std::vector<std::unique_ptr<HybridGridTSDF>> grids;
  std::vector<HybridGridTSDF*> mine;
  for (float r : res) {
    grids.emplace_back(new HybridGridTSDF(&ctx, r, 2.5f, 1000.f, 1u << 14));
    mine.push_back(grids.back().get());
  }
  hg::mapping::TSDFRangeDataInserter3D inserter(hg::mapping::DefaultTSDFInserterOptions());
  for (int k = 0; k < 3; ++k) {
    const hg::sensor::RangeData rd = RingScan(1.5f * static_cast<float>(tr.rank()) + 0.05f * k, 0.02f * k);
    for (HybridGridTSDF* g : mine) inserter.Insert(rd, g);
  }
  hg::mapping::GatherReport rep;
  std::vector<hg::mapping::Pyramid> all = hg::mapping::GatherSubmaps(tr, &ctx, mine, 0, &rep);
  if (tr.rank() == 0) {
    std::printf("gather %s: ranks %d levels %d blocks %llu voxels %llu in %.3f ms\n", rep.ok ? "ok" : "MISMATCH",
                rep.ranks, rep.levels, static_cast<unsigned long long>(rep.blocks),
                static_cast<unsigned long long>(rep.voxels), rep.seconds * 1e3);
    for (size_t src = 0; src < all.size(); ++src) {
      std::printf("rank %zu:", src);
      for (const auto& g : all[src]) {
        uint32_t nb = 0;
        hg_grid_num_blocks(g->get(), &nb);
        std::printf(" %u", nb);
      }
      std::printf(" blocks\n");
    }
    return rep.ok ? 0 : 1;
  }
  return 0;
}

int RunPipe(int world) {
  // pipes between rank 0 and every peer (the gather is a star), created before the fork
  std::vector<std::array<int, 2>> up(world), down(world);  // up[p]: p -> 0, down[p]: 0 -> p
  for (int p = 1; p < world; ++p)
    if (pipe(up[p].data()) != 0 || pipe(down[p].data()) != 0) return 2;
  std::vector<pid_t> kids;
  int rank = 0;
  for (int p = 1; p < world; ++p) {
    const pid_t pid = fork();  // no GPU call has been made yet in this process
    if (pid == 0) { rank = p; break; }
    kids.push_back(pid);
  }
  std::vector<int> rd(world, -1), wr(world, -1);
  if (rank == 0) {
    for (int p = 1; p < world; ++p) { rd[p] = up[p][0]; wr[p] = down[p][1]; }
  } else {
    rd[0] = down[rank][0];
    wr[0] = up[rank][1];
  }
  int rc = 1;
  try {
    hg::mapping::PipeTransport tr(rank, world, rd, wr);
    rc = RunRank(tr, 0);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
  }
  if (rank != 0) _exit(rc);
  for (pid_t k : kids) {
    int st = 0;
    waitpid(k, &st, 0);
    if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = rc ? rc : 3;
  }
  return rc;
}

#ifdef HG_WITH_RCCL
int RunRccl() {
  const char* e;
  const int rank = (e = std::getenv("RANK")) ? std::atoi(e) : 0;
  const int world = (e = std::getenv("WORLD_SIZE")) ? std::atoi(e) : 1;
  const int local = (e = std::getenv("LOCAL_RANK")) ? std::atoi(e) : rank;
  const char* id_file = std::getenv("HG_NCCL_ID_FILE");
  hg::mapping::HipCheck(hipSetDevice(local), "hipSetDevice");
  ncclUniqueId id;
  if (rank == 0) {
    if (ncclGetUniqueId(&id) != ncclSuccess) return 2;
    if (world > 1) {
      if (!id_file) { std::fprintf(stderr, "HG_NCCL_ID_FILE not set\n"); return 2; }
      std::ofstream f(std::string(id_file) + ".tmp", std::ios::binary);
      f.write(reinterpret_cast<const char*>(&id), sizeof(id));
      f.close();
      std::rename((std::string(id_file) + ".tmp").c_str(), id_file);
    }
  } else {
    if (!id_file) { std::fprintf(stderr, "HG_NCCL_ID_FILE not set\n"); return 2; }
    for (int tries = 0;; ++tries) {
      std::ifstream f(id_file, std::ios::binary);
      if (f && f.read(reinterpret_cast<char*>(&id), sizeof(id))) break;
      if (tries > 600) { std::fprintf(stderr, "no id file\n"); return 2; }
      std::this_thread::sleep_for(std::chrono::milliseconds(100));
    }
  }
  ncclComm_t comm;
  if (ncclCommInitRank(&comm, world, id, rank) != ncclSuccess) return 2;
  hipStream_t stream;
  hg::mapping::HipCheck(hipStreamCreate(&stream), "hipStreamCreate");
  int rc = 1;
  try {
    hg::mapping::RcclTransport tr(comm, rank, world, stream);
    rc = RunRank(tr, local);
  } catch (const std::exception& ex) {
    std::fprintf(stderr, "rank %d: %s\n", rank, ex.what());
  }
  ncclCommDestroy(comm);
  return rc;
}
#endif

}  // namespace

int main(int argc, char** argv) {
  const std::string mode = argc > 1 ? argv[1] : "pipe";
  if (mode == "pipe") return RunPipe(argc > 2 ? std::atoi(argv[2]) : 2);
#ifdef HG_WITH_RCCL
  if (mode == "rccl") return RunRccl();
#endif
  std::fprintf(stderr, "usage: example_gather pipe N | rccl\n");
  return 2;
}
