// hg_gather.h — the one exchange step of multi-GPU batch mapping for a C++ host: the finished TSDF
// blocks of every rank's submap gathered to one rank, and the end-to-end check of that gather.
//
// The path shards only across independent submaps: a rank maps its submaps without
// talking to anyone; afterwards the occupied 8^3 blocks (8-byte key + 2 KiB of voxels each) travel to
// the destination rank — counts first (all-gather), then point-to-point payload per peer: variable
// sizes, no reduction (submaps are disjoint). Same protocol as hectorgrapher_amd/distributed.py.
//
// The transport is a template parameter with this interface (all buffers are DEVICE memory):
//   int rank() const; int size() const;
//   void AllGather(const uint64_t* mine, size_t words, uint64_t* all);   // all: size() * words
//   void Send(const void* dev, size_t bytes, int peer);
//   void Recv(void* dev, size_t bytes, int peer);
//   void Begin(); void End();      // bracket a batch of Send / Recv (a RCCL group); End completes them
// RcclTransport (one process per GPU, ncclSend / ncclRecv over xGMI) is the production transport;
// PipeTransport stages the payload through host memory and POSIX pipes, for ranks that are processes
// of one machine (the tests run two ranks on one GPU with it).
#pragma once

#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "hg_adapter.h"

namespace hg_amd {
namespace mapping {

inline void HipCheck(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

// FNV-1a over the export of a grid (cells in iterator order, tsd and weight codes): what two grids
// must share to be the same HybridGridTSDF.
struct ExportDigest {
  uint64_t voxels = 0;
  uint64_t hash = 0;
  bool operator==(const ExportDigest& o) const { return voxels == o.voxels && hash == o.hash; }
};
inline ExportDigest DigestOf(const HybridGridTSDF& grid) {
  std::vector<std::array<int, 3>> cells;
  std::vector<uint16_t> tsd, weight;
  ExportDigest d;
  d.voxels = grid.Export(&cells, &tsd, &weight);
  uint64_t h = 1469598103934665603ull;
  auto mix = [&h](const void* p, size_t bytes) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < bytes; ++i) { h ^= b[i]; h *= 1099511628211ull; }
  };
  if (d.voxels) {
    mix(cells.data(), cells.size() * sizeof(cells[0]));
    mix(tsd.data(), tsd.size() * sizeof(uint16_t));
    mix(weight.data(), weight.size() * sizeof(uint16_t));
  }
  d.hash = h;
  return d;
}

struct GatherReport {
  bool ok = true;           // every imported submap exports exactly what its owner exported
  int ranks = 0, levels = 0;
  uint64_t blocks = 0, voxels = 0;
  double seconds = 0.0;     // of the gather itself (dst rank)
};

// A gathered submap on the destination rank: one grid per pyramid level.
using Pyramid = std::vector<std::unique_ptr<HybridGridTSDF>>;

// Collective: every rank calls it with its own pyramid (same number of levels and grid parameters on
// all ranks). On `dst` it returns the pyramids of all ranks (index = source rank; the entry of `dst`
// itself is a copy made through the same import path) and fills `report`; elsewhere the result is empty.
template <class Transport>
std::vector<Pyramid> GatherSubmaps(Transport& tr, Context* ctx, const std::vector<HybridGridTSDF*>& mine,
                                   int dst, GatherReport* report) {
  const int world = tr.size(), rank = tr.rank(), levels = static_cast<int>(mine.size());
  // 1. block arrays (packed device copies) and the digests of the local exports
  std::vector<void*> keys(levels), voxels(levels);
  std::vector<uint64_t> local(3 * levels);
  for (int l = 0; l < levels; ++l) {
    uint32_t nb = 0;
    Check(hg_grid_block_arrays(mine[l]->get(), &keys[l], &voxels[l], &nb), "hg_grid_block_arrays");
    const ExportDigest d = DigestOf(*mine[l]);
    local[3 * l] = nb;
    local[3 * l + 1] = d.voxels;
    local[3 * l + 2] = d.hash;
  }
  Check(hg_ctx_synchronize(ctx->get()), "hg_ctx_synchronize");
  // 2. counts and digests of every rank
  std::vector<uint64_t> all(static_cast<size_t>(world) * 3 * levels);
  tr.AllGather(local.data(), local.size(), all.data());
  std::vector<Pyramid> out;
  if (rank != dst) {
    // 3a. payload to dst
    tr.Begin();
    for (int l = 0; l < levels; ++l) {
      const uint64_t nb = local[3 * l];
      if (!nb) continue;
      tr.Send(keys[l], nb * sizeof(uint64_t), dst);
      tr.Send(voxels[l], nb * 512u * sizeof(uint32_t), dst);
    }
    tr.End();
    return out;
  }
  // 3b. dst: receive into device buffers, import into fresh grids, compare exports
  hipEvent_t t0, t1;
  HipCheck(hipEventCreate(&t0), "hipEventCreate");
  HipCheck(hipEventCreate(&t1), "hipEventCreate");
  hipStream_t stream = static_cast<hipStream_t>(hg_ctx_stream(ctx->get()));
  HipCheck(hipEventRecord(t0, stream), "hipEventRecord");
  std::vector<std::vector<void*>> rk(world, std::vector<void*>(levels, nullptr)), rv = rk;
  tr.Begin();
  for (int src = 0; src < world; ++src) {
    if (src == dst) continue;
    for (int l = 0; l < levels; ++l) {
      const uint64_t nb = all[(static_cast<size_t>(src) * levels + l) * 3];
      if (!nb) continue;
      HipCheck(hipMalloc(&rk[src][l], nb * sizeof(uint64_t)), "hipMalloc");
      HipCheck(hipMalloc(&rv[src][l], nb * 512u * sizeof(uint32_t)), "hipMalloc");
      tr.Recv(rk[src][l], nb * sizeof(uint64_t), src);
      tr.Recv(rv[src][l], nb * 512u * sizeof(uint32_t), src);
    }
  }
  tr.End();
  HipCheck(hipEventRecord(t1, stream), "hipEventRecord");
  HipCheck(hipEventSynchronize(t1), "hipEventSynchronize");
  float ms = 0.f;
  HipCheck(hipEventElapsedTime(&ms, t0, t1), "hipEventElapsedTime");
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  GatherReport rep;
  rep.ranks = world;
  rep.levels = levels;
  rep.seconds = ms * 1e-3;
  out.resize(world);
  for (int src = 0; src < world; ++src) {
    for (int l = 0; l < levels; ++l) {
      const uint64_t* e = &all[(static_cast<size_t>(src) * levels + l) * 3];
      const uint64_t nb = e[0];
      std::unique_ptr<HybridGridTSDF> g(new HybridGridTSDF(
          ctx, mine[l]->resolution(), mine[l]->relative_truncation_distance(), mine[l]->max_weight(),
          static_cast<uint32_t>(nb < 64 ? 64 : nb)));
      const void* k = src == dst ? keys[l] : rk[src][l];
      const void* v = src == dst ? voxels[l] : rv[src][l];
      if (nb) Check(hg_grid_import_blocks(g->get(), k, v, static_cast<uint32_t>(nb), HG_DEVICE), "hg_grid_import_blocks");
      const ExportDigest d = DigestOf(*g);
      rep.ok = rep.ok && d.voxels == e[1] && d.hash == e[2];
      rep.blocks += nb;
      rep.voxels += d.voxels;
      out[src].push_back(std::move(g));
      if (rk[src][l]) (void)hipFree(rk[src][l]);
      if (rv[src][l]) (void)hipFree(rv[src][l]);
    }
  }
  if (report) *report = rep;
  return out;
}

// Ranks that are processes of one machine, connected by pipes (fds[peer] = {read end from peer, write
// end to peer}); payload staged through host memory. For tests and single-node tools.
class PipeTransport {
 public:
  PipeTransport(int rank, int size, std::vector<int> read_fd, std::vector<int> write_fd)
      : rank_(rank), size_(size), rd_(std::move(read_fd)), wr_(std::move(write_fd)) {}
  int rank() const { return rank_; }
  int size() const { return size_; }
  void Begin() {}
  void End() {}
  void AllGather(const uint64_t* mine, size_t words, uint64_t* all) {
    // rank 0 collects and redistributes
    const size_t bytes = words * sizeof(uint64_t);
    if (rank_ == 0) {
      std::memcpy(all, mine, bytes);
      for (int p = 1; p < size_; ++p) ReadAll(rd_[p], all + p * words, bytes);
      for (int p = 1; p < size_; ++p) WriteAll(wr_[p], all, bytes * size_);
    } else {
      WriteAll(wr_[0], mine, bytes);
      ReadAll(rd_[0], all, bytes * size_);
    }
  }
  void Send(const void* dev, size_t bytes, int peer) {
    std::vector<char> host(bytes);
    HipCheck(hipMemcpy(host.data(), dev, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    WriteAll(wr_[peer], host.data(), bytes);
  }
  void Recv(void* dev, size_t bytes, int peer) {
    std::vector<char> host(bytes);
    ReadAll(rd_[peer], host.data(), bytes);
    HipCheck(hipMemcpy(dev, host.data(), bytes, hipMemcpyHostToDevice), "hipMemcpy H2D");
  }

 private:
  static void WriteAll(int fd, const void* p, size_t bytes) {
    const char* b = static_cast<const char*>(p);
    while (bytes) {
      const ssize_t w = ::write(fd, b, bytes);
      if (w <= 0) throw std::runtime_error("pipe write failed");
      b += w;
      bytes -= static_cast<size_t>(w);
    }
  }
  static void ReadAll(int fd, void* p, size_t bytes) {
    char* b = static_cast<char*>(p);
    while (bytes) {
      const ssize_t r = ::read(fd, b, bytes);
      if (r <= 0) throw std::runtime_error("pipe read failed");
      b += r;
      bytes -= static_cast<size_t>(r);
    }
  }
  int rank_, size_;
  std::vector<int> rd_, wr_;
};

}  // namespace mapping
}  // namespace hg_amd

#ifdef HG_WITH_RCCL
#include <rccl/rccl.h>
namespace hg_amd {
namespace mapping {
// One process per GPU; `comm` spans the ranks, payload goes GPU to GPU (xGMI inside a node) on the
// context's stream. Tens of MB per submap over ~153 GB/s links: latency bound, so all receives of the
// destination are posted in ONE group.
class RcclTransport {
 public:
  RcclTransport(ncclComm_t comm, int rank, int size, hipStream_t stream)
      : comm_(comm), rank_(rank), size_(size), stream_(stream) {}
  int rank() const { return rank_; }
  int size() const { return size_; }
  void Begin() { NcclCheck(ncclGroupStart(), "ncclGroupStart"); }
  void End() {
    NcclCheck(ncclGroupEnd(), "ncclGroupEnd");
    HipCheck(hipStreamSynchronize(stream_), "hipStreamSynchronize");
  }
  void AllGather(const uint64_t* mine, size_t words, uint64_t* all) {
    uint64_t* d = nullptr;
    HipCheck(hipMalloc(reinterpret_cast<void**>(&d), (size_ + 1) * words * sizeof(uint64_t)), "hipMalloc");
    HipCheck(hipMemcpyAsync(d, mine, words * sizeof(uint64_t), hipMemcpyHostToDevice, stream_), "hipMemcpyAsync");
    NcclCheck(ncclAllGather(d, d + words, words, ncclUint64, comm_, stream_), "ncclAllGather");
    HipCheck(hipMemcpyAsync(all, d + words, size_ * words * sizeof(uint64_t), hipMemcpyDeviceToHost, stream_), "hipMemcpyAsync");
    HipCheck(hipStreamSynchronize(stream_), "hipStreamSynchronize");
    (void)hipFree(d);
  }
  void Send(const void* dev, size_t bytes, int peer) {
    NcclCheck(ncclSend(dev, bytes, ncclUint8, peer, comm_, stream_), "ncclSend");
  }
  void Recv(void* dev, size_t bytes, int peer) {
    NcclCheck(ncclRecv(dev, bytes, ncclUint8, peer, comm_, stream_), "ncclRecv");
  }

 private:
  static void NcclCheck(ncclResult_t r, const char* what) {
    if (r != ncclSuccess) throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
  }
  ncclComm_t comm_;
  int rank_, size_;
  hipStream_t stream_;
};
}  // namespace mapping
}  // namespace hg_amd
#endif
