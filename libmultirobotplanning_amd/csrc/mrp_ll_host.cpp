// C-ABI of the MI355X low-level search engine (include/mrp_ll.h): context, map upload, batch packing, launch.
// No CPU fallback: every entry point needs a working HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../../include/mrp_ll.h"
#include "ll_device.h"

extern "C" uint32_t mrp_ll_lds_bytes(uint32_t capNodes, uint32_t rows, uint32_t rowWords, uint32_t pathBytes);
extern "C" hipError_t mrp_ll_launch(const mrp::LaunchParams* P, uint32_t grid, uint32_t ldsBytes, hipStream_t stream);

namespace {

using mrp::DevJob;
using mrp::DevResult;

struct MapRec {
  int32_t dimx, dimy;
  uint32_t wpr, wordOff;
};

// Growable pinned host buffer that the device accesses in place (zero-copy staging, see ll_device.h).
template <typename T>
struct PinnedBuf {
  T* host = nullptr;
  T* dev = nullptr;   // device-side address of the same memory
  size_t cap = 0, size = 0;
  hipError_t reserve(size_t n) {
    if (n <= cap) return hipSuccess;
    size_t ncap = std::max<size_t>(n, cap * 2);
    ncap = std::max<size_t>(ncap, 4096);
    T* nh = nullptr;
    hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&nh), ncap * sizeof(T), hipHostMallocMapped);
    if (e != hipSuccess) return e;
    void* nd = nullptr;
    e = hipHostGetDevicePointer(&nd, nh, 0);
    if (e != hipSuccess) return e;
    if (host) {
      if (size) std::memcpy(nh, host, size * sizeof(T));
      (void)hipHostFree(host);
    }
    host = nh;
    dev = static_cast<T*>(nd);
    cap = ncap;
    return hipSuccess;
  }
  hipError_t resize(size_t n) {
    hipError_t e = reserve(n);
    if (e == hipSuccess) size = n;
    return e;
  }
  hipError_t push(const T& v) {
    if (size == cap) {
      hipError_t e = reserve(size + 1);
      if (e != hipSuccess) return e;
    }
    host[size++] = v;
    return hipSuccess;
  }
  void clear() { size = 0; }
  void release() {
    if (host) (void)hipHostFree(host);
    host = dev = nullptr;
    cap = size = 0;
  }
};

struct Ticket {
  hipStream_t stream = nullptr;
  hipEvent_t evK0 = nullptr, evK1 = nullptr;
  PinnedBuf<DevJob> jobs;
  PinnedBuf<uint32_t> cons;
  PinnedBuf<uint16_t> paths;
  PinnedBuf<DevResult> results;
  PinnedBuf<uint16_t> outPaths;
  uint32_t* queueHead = nullptr;   // device, monotonic
  uint32_t queueBase = 0;
  uint8_t* arena = nullptr;
  bool inFlight = false;
  bool allocFailed = false;
  int32_t nJobs = 0;
  mrp_ll_result* userResults = nullptr;
  std::vector<uint8_t> rejected;  // per job: rejected on the host (MRP_LL_BAD_JOB)
};

}  // namespace

struct mrp_ll_ctx {
  mrp_ll_options opt;
  int device = 0;
  std::string err;
  std::vector<MapRec> maps;
  std::vector<uint32_t> mapWords;  // host copy of all obstacle bitmaps
  uint32_t* mapsDev = nullptr;
  size_t mapsDevCap = 0;
  bool mapsDirty = false;
  uint32_t maxWpr = 1;
  uint32_t arenaRowWords = 0;
  uint64_t arenaStride = 0;
  uint32_t arenaScratchOff = 0;
  uint32_t arenaPathsBytes = 0;
  std::vector<Ticket> tickets;
  mrp_ll_stats stats;
  uint32_t* debugHost = nullptr;  // MRP_LL_DEBUG: host-mapped trace buffer
};

namespace {

static const bool kDebug = std::getenv("MRP_LL_DEBUG") != nullptr;
#define HIPCHK(ctx, call)                                                                         \
  do {                                                                                            \
    if (kDebug) { std::fprintf(stderr, "[mrp_ll] %s\n", #call); std::fflush(stderr); }            \
    hipError_t e__ = (call);                                                                      \
    if (e__ != hipSuccess) {                                                                      \
      (ctx)->err = std::string(#call) + ": " + hipGetErrorString(e__);                            \
      return MRP_LL_E_DEVICE;                                                                     \
    }                                                                                             \
  } while (0)

int actionFromDelta(int dx, int dy) {
  if (dx == 0 && dy == 0) return MRP_LL_ACT_WAIT;
  if (dx == -1 && dy == 0) return MRP_LL_ACT_LEFT;
  if (dx == 1 && dy == 0) return MRP_LL_ACT_RIGHT;
  if (dx == 0 && dy == 1) return MRP_LL_ACT_UP;
  if (dx == 0 && dy == -1) return MRP_LL_ACT_DOWN;
  return -1;
}
// index in the reference's successor order Wait, Left, Right, Up, Down (ecbs.cpp:365-398)
int neighborIndexFromDelta(int dx, int dy) {
  if (dx == 0 && dy == 0) return 0;
  if (dx == -1 && dy == 0) return 1;
  if (dx == 1 && dy == 0) return 2;
  if (dx == 0 && dy == 1) return 3;
  if (dx == 0 && dy == -1) return 4;
  return -1;
}

int syncMaps(mrp_ll_ctx* ctx) {
  if (!ctx->mapsDirty) return MRP_LL_SUCCESS;
  // all tickets must be idle before the maps buffer may move
  size_t need = std::max<size_t>(ctx->mapWords.size(), 1);
  if (need > ctx->mapsDevCap) {
    for (auto& t : ctx->tickets)
      if (t.inFlight) HIPCHK(ctx, hipEventSynchronize(t.evK1));
    if (ctx->mapsDev) HIPCHK(ctx, hipFree(ctx->mapsDev));
    size_t ncap = std::max<size_t>(need * 2, 4096);
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->mapsDev), ncap * sizeof(uint32_t)));
    ctx->mapsDevCap = ncap;
  }
  HIPCHK(ctx, hipMemcpy(ctx->mapsDev, ctx->mapWords.data(), ctx->mapWords.size() * sizeof(uint32_t),
                        hipMemcpyHostToDevice));
  ctx->mapsDirty = false;
  return MRP_LL_SUCCESS;
}

// Pack one job; returns false if the job is rejected (MRP_LL_BAD_JOB).
bool packJob(mrp_ll_ctx* ctx, const mrp_ll_job& j, Ticket& t, DevJob& d) {
  if (j.map_id < 0 || j.map_id >= static_cast<int32_t>(ctx->maps.size())) return false;
  const MapRec& mp = ctx->maps[j.map_id];
  if (j.algo != MRP_LL_ASTAR && j.algo != MRP_LL_ASTAR_EPS) return false;
  auto inGrid = [&](int x, int y) { return x >= 0 && x < mp.dimx && y >= 0 && y < mp.dimy; };
  if (!inGrid(j.start_x, j.start_y)) return false;
  if (j.n_vertex_constraints < 0 || j.n_edge_constraints < 0 || j.n_agents < 0) return false;
  if (j.n_vertex_constraints > 0 && !j.vertex_constraints) return false;
  if (j.n_edge_constraints > 0 && !j.edge_constraints) return false;
  const int horizon = ctx->opt.max_horizon;
  std::memset(&d, 0, sizeof(d));
  d.map_word_off = mp.wordOff;
  d.dimx = mp.dimx;
  d.dimy = mp.dimy;
  d.words_per_row = mp.wpr;
  d.sx = j.start_x;
  d.sy = j.start_y;
  // a goal outside the grid can never be reached; keep the reference behaviour (search until open is exhausted /
  // capped) by parking it on an unreachable coordinate that still fits the 8-bit fields only if in range
  if (!inGrid(j.goal_x, j.goal_y)) return false;
  d.gx = j.goal_x;
  d.gy = j.goal_y;
  d.algo = j.algo;
  d.w = j.w;
  d.max_expansions = j.max_expansions;
  // setLowLevelContext (ecbs.cpp:264-274): last vertex constraint on the goal cell
  int lastGoal = -1;
  d.vc_off = static_cast<uint32_t>(t.cons.size);
  for (int i = 0; i < j.n_vertex_constraints; ++i) {
    const int32_t* v = j.vertex_constraints + 3 * i;
    if (v[1] == j.goal_x && v[2] == j.goal_y) lastGoal = std::max(lastGoal, v[0]);
    if (v[0] < 0 || v[0] >= horizon || !inGrid(v[1], v[2])) continue;  // can never match a generated state
    if (t.cons.push((static_cast<uint32_t>(v[0]) << 16) | static_cast<uint32_t>(v[2] * mp.dimx + v[1])) != hipSuccess)
      t.allocFailed = true;
  }
  d.n_vc = static_cast<uint32_t>(t.cons.size) - d.vc_off;
  d.last_goal_constraint = lastGoal;
  d.ec_off = static_cast<uint32_t>(t.cons.size);
  for (int i = 0; i < j.n_edge_constraints; ++i) {
    const int32_t* e = j.edge_constraints + 5 * i;
    int k = neighborIndexFromDelta(e[3] - e[1], e[4] - e[2]);
    if (k < 0 || e[0] < 0 || e[0] >= horizon || !inGrid(e[1], e[2])) continue;
    if (t.cons.push((static_cast<uint32_t>(e[0]) << 19) | (static_cast<uint32_t>(e[2] * mp.dimx + e[1]) << 3) |
                    static_cast<uint32_t>(k)) != hipSuccess)
      t.allocFailed = true;
  }
  d.n_ec = static_cast<uint32_t>(t.cons.size) - d.ec_off;
  // focal context: time-major table of the other agents' cells, each path extended by its last cell
  d.n_agents_pad = 0;
  d.t_pad = 0;
  d.path_off = 0;
  if (j.algo == MRP_LL_ASTAR_EPS && j.n_agents > 0) {
    if (!j.path_len || !j.path_xy) return false;
    int tpad = 0;
    for (int a = 0; a < j.n_agents; ++a)
      if (a != j.agent_idx && j.path_len[a] > 0) {
        if (!j.path_xy[a]) return false;
        tpad = std::max(tpad, j.path_len[a]);
      }
    if (tpad > 0) {
      uint32_t npad = (static_cast<uint32_t>(j.n_agents) + 15u) & ~15u;
      // 16-byte align the table start
      size_t base = (t.paths.size + 7u) & ~size_t(7);
      if (t.paths.resize(base + static_cast<size_t>(tpad) * npad) != hipSuccess) {
        t.allocFailed = true;
        return false;
      }
      d.path_off = static_cast<uint32_t>(base);
      d.n_agents_pad = npad;
      d.t_pad = static_cast<uint32_t>(tpad);
      uint16_t* tab = t.paths.host + base;
      const uint16_t none = static_cast<uint16_t>(mrp::kEmptyCell);
      for (size_t q = 0; q < static_cast<size_t>(tpad) * npad; ++q) tab[q] = none;
      for (int a = 0; a < j.n_agents; ++a) {
        int len = j.path_len[a];
        if (a == j.agent_idx || len <= 0) continue;
        const int32_t* xy = j.path_xy[a];
        uint16_t cell = none;
        for (int tt = 0; tt < tpad; ++tt) {
          if (tt < len) {
            int x = xy[2 * tt], y = xy[2 * tt + 1];
            cell = inGrid(x, y) ? static_cast<uint16_t>(y * mp.dimx + x) : none;
          }
          tab[static_cast<size_t>(tt) * npad + a] = cell;
        }
      }
    }
  }
  return true;
}

}  // namespace

extern "C" {

const char* mrp_ll_version(void) { return "mrp_ll 0.1 (gfx950, HIP)"; }

const char* mrp_ll_last_error(const mrp_ll_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

int mrp_ll_create(const mrp_ll_options* optIn, mrp_ll_ctx** out) {
  if (!out) return MRP_LL_E_INVALID;
  *out = nullptr;
  mrp_ll_options o;
  std::memset(&o, 0, sizeof(o));
  if (optIn) o = *optIn;
  if (o.n_tickets <= 0) o.n_tickets = 4;
  if (o.slots <= 0) o.slots = 1024;
  if (o.arena_nodes <= 0) o.arena_nodes = 131072;
  if (o.arena_nodes > static_cast<int32_t>(mrp::kMaxArenaNodes)) o.arena_nodes = mrp::kMaxArenaNodes;
  o.arena_nodes &= ~1;
  if (o.max_horizon <= 0) o.max_horizon = 512;
  if (o.max_horizon > static_cast<int32_t>(mrp::kMaxHorizon)) o.max_horizon = mrp::kMaxHorizon;
  if (o.max_cells <= 0) o.max_cells = 4096;
  if (o.max_cells > 255 * 255) o.max_cells = 255 * 255;
  if (o.lds_nodes == 0) o.lds_nodes = 512;
  if (o.lds_nodes < 0) o.lds_nodes = 0;
  o.lds_nodes &= ~1;

  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || o.device < 0 || o.device >= ndev) {
    return MRP_LL_E_DEVICE;  // no fallback: the engine is HIP-only
  }
  mrp_ll_ctx* ctx = new mrp_ll_ctx();
  ctx->opt = o;
  ctx->device = o.device;
  std::memset(&ctx->stats, 0, sizeof(ctx->stats));
  if (hipSetDevice(o.device) != hipSuccess) {
    delete ctx;
    return MRP_LL_E_DEVICE;
  }
  ctx->arenaRowWords = (static_cast<uint32_t>(o.max_cells) + 31u) / 32u;
  uint64_t stride = static_cast<uint64_t>(o.arena_nodes) * 16 + 3ull * (static_cast<uint64_t>(o.arena_nodes) * 8 + 16) +
                    static_cast<uint64_t>(o.max_horizon) * ctx->arenaRowWords * 4;
  stride = (stride + 255) & ~255ull;
  // scratch tail of every slot: [path out: max_horizon u16][constraint copy][path-table copy]
  ctx->arenaScratchOff = static_cast<uint32_t>(stride);
  ctx->arenaPathsBytes = 128 * 1024;
  stride += static_cast<uint64_t>(o.max_horizon) * 2 + mrp::kConsLocalWords * 4 + ctx->arenaPathsBytes;
  stride = (stride + 255) & ~255ull;
  ctx->arenaStride = stride;
  ctx->tickets.resize(o.n_tickets);
  for (auto& t : ctx->tickets) {
    hipError_t e = hipStreamCreateWithFlags(&t.stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&t.evK0);
    if (e == hipSuccess) e = hipEventCreate(&t.evK1);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t.queueHead), 256);
    if (e == hipSuccess) e = hipMemset(t.queueHead, 0, 256);
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&t.arena), stride * static_cast<uint64_t>(o.slots));
    if (e != hipSuccess) {
      ctx->err = std::string("mrp_ll_create: ") + hipGetErrorString(e);
      mrp_ll_destroy(ctx);
      return e == hipErrorOutOfMemory ? MRP_LL_E_NOMEM : MRP_LL_E_DEVICE;
    }
  }
  *out = ctx;
  return MRP_LL_SUCCESS;
}

void mrp_ll_destroy(mrp_ll_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  for (auto& t : ctx->tickets) {
    if (t.inFlight && t.evK1) (void)hipEventSynchronize(t.evK1);
    if (t.stream) (void)hipStreamSynchronize(t.stream);
    t.jobs.release();
    t.cons.release();
    t.paths.release();
    t.results.release();
    t.outPaths.release();
    if (t.queueHead) (void)hipFree(t.queueHead);
    if (t.arena) (void)hipFree(t.arena);
    if (t.evK0) (void)hipEventDestroy(t.evK0);
    if (t.evK1) (void)hipEventDestroy(t.evK1);
    if (t.stream) (void)hipStreamDestroy(t.stream);
  }
  if (ctx->mapsDev) (void)hipFree(ctx->mapsDev);
  delete ctx;
}

int mrp_ll_upload_map(mrp_ll_ctx* ctx, int32_t dimx, int32_t dimy, int32_t nObst, const int32_t* obstXY,
                      int32_t* mapId) {
  if (!ctx || !mapId || dimx <= 0 || dimy <= 0 || dimx > 255 || dimy > 255 || nObst < 0 || (nObst > 0 && !obstXY)) {
    if (ctx) ctx->err = "mrp_ll_upload_map: invalid argument (dimensions must be 1..255)";
    return MRP_LL_E_INVALID;
  }
  if (dimx * dimy > ctx->opt.max_cells) {
    ctx->err = "mrp_ll_upload_map: dimx*dimy exceeds mrp_ll_options.max_cells";
    return MRP_LL_E_INVALID;
  }
  MapRec m;
  m.dimx = dimx;
  m.dimy = dimy;
  m.wpr = (static_cast<uint32_t>(dimx * dimy) + 31u) / 32u;
  // keep every bitmap 16-byte aligned
  while (ctx->mapWords.size() & 3u) ctx->mapWords.push_back(0);
  m.wordOff = static_cast<uint32_t>(ctx->mapWords.size());
  ctx->mapWords.resize(ctx->mapWords.size() + m.wpr, 0u);
  uint32_t* w = ctx->mapWords.data() + m.wordOff;
  // cells past dimx*dimy in the last word are never addressed
  for (int i = 0; i < nObst; ++i) {
    int x = obstXY[2 * i], y = obstXY[2 * i + 1];
    if (x < 0 || x >= dimx || y < 0 || y >= dimy) continue;  // unreachable anyway (stateValid bounds, ecbs.cpp:500)
    uint32_t cell = static_cast<uint32_t>(y * dimx + x);
    w[cell >> 5] |= 1u << (cell & 31);
  }
  ctx->maps.push_back(m);
  ctx->maxWpr = std::max(ctx->maxWpr, m.wpr);
  ctx->mapsDirty = true;
  *mapId = static_cast<int32_t>(ctx->maps.size()) - 1;
  return MRP_LL_SUCCESS;
}

int mrp_ll_submit(mrp_ll_ctx* ctx, int32_t nJobs, const mrp_ll_job* jobs, mrp_ll_result* results, int32_t* ticketOut) {
  if (!ctx || !ticketOut || nJobs < 0 || (nJobs > 0 && (!jobs || !results))) return MRP_LL_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  int ti = -1;
  for (size_t i = 0; i < ctx->tickets.size(); ++i)
    if (!ctx->tickets[i].inFlight) {
      ti = static_cast<int>(i);
      break;
    }
  if (ti < 0) {
    ctx->err = "mrp_ll_submit: all tickets in flight";
    return MRP_LL_E_BUSY;
  }
  int rc = syncMaps(ctx);
  if (rc != MRP_LL_SUCCESS) return rc;
  Ticket& t = ctx->tickets[ti];
  auto packT0 = std::chrono::steady_clock::now();
  t.nJobs = nJobs;
  t.userResults = results;
  t.rejected.assign(nJobs, 0);
  t.allocFailed = false;
  t.jobs.clear();
  t.cons.clear();
  t.paths.clear();
  HIPCHK(ctx, t.jobs.resize(std::max(nJobs, 1)));
  for (int i = 0; i < nJobs; ++i) {
    size_t c0 = t.cons.size, p0 = t.paths.size;
    if (!packJob(ctx, jobs[i], t, t.jobs.host[i])) {
      // rejected: give the device a trivially capped job and remember the rejection
      t.cons.size = c0;
      t.paths.size = p0;
      t.rejected[i] = 1;
      DevJob& d = t.jobs.host[i];
      std::memset(&d, 0, sizeof(d));
      d.dimx = 1; d.dimy = 1; d.words_per_row = 1;
      d.map_word_off = ctx->maps.empty() ? 0 : ctx->maps[0].wordOff;
      d.gx = 0; d.gy = 0; d.algo = 0; d.last_goal_constraint = -1;
      d.max_expansions = 0;
    }
  }
  if (t.allocFailed) {
    ctx->err = "mrp_ll_submit: pinned staging allocation failed";
    return MRP_LL_E_NOMEM;
  }
  *ticketOut = ti;
  t.inFlight = true;
  if (nJobs == 0) return MRP_LL_SUCCESS;
  const uint32_t outStride = static_cast<uint32_t>(ctx->opt.max_horizon);
  HIPCHK(ctx, t.cons.reserve(16));
  HIPCHK(ctx, t.paths.reserve(16));
  HIPCHK(ctx, t.results.resize(nJobs));
  HIPCHK(ctx, t.outPaths.resize(static_cast<size_t>(nJobs) * outStride));
  ctx->stats.pack_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - packT0).count();

  mrp::LaunchParams P;
  std::memset(&P, 0, sizeof(P));
  P.jobs = t.jobs.dev;
  P.results = t.results.dev;
  P.out_paths = t.outPaths.dev;
  P.maps = ctx->mapsDev;
  P.cons = t.cons.dev;
  P.paths = t.paths.dev;
  P.queue_head = t.queueHead;
  P.queue_base = t.queueBase;
  P.arena = t.arena;
  P.arena_stride = ctx->arenaStride;
  P.arena_scratch_off = ctx->arenaScratchOff;
  P.arena_paths_bytes = ctx->arenaPathsBytes;
  P.n_jobs = static_cast<uint32_t>(nJobs);
  P.out_stride = outStride;
  P.arena_nodes = static_cast<uint32_t>(ctx->opt.arena_nodes);
  P.arena_rows = static_cast<uint32_t>(ctx->opt.max_horizon);
  P.arena_row_words = ctx->arenaRowWords;
  // LDS tier geometry: rows sized for the widest uploaded map; a workgroup may take up to the CU's whole 160 KiB
  // (minus the kernel's small static LDS); occupancy is floor(160 KiB / ldsBytes) workgroups per CU
  uint32_t ldsNodes = static_cast<uint32_t>(ctx->opt.lds_nodes);
  uint32_t rowWords = (ctx->maxWpr + 3u) & ~3u;
  uint32_t rows = 0;
  uint32_t ldsBytes = 0;
  const uint32_t ldsPaths = 4096;
  if (ldsNodes) {
    const uint32_t budget = 160 * 1024 - 512;
    uint32_t fixed = mrp_ll_lds_bytes(ldsNodes, 0, rowWords, ldsPaths);
    if (fixed + 16 * rowWords * 4 <= budget) {
      rows = std::min<uint32_t>(64, (budget - fixed) / (rowWords * 4));
      rows = std::min<uint32_t>(rows, static_cast<uint32_t>(ctx->opt.max_horizon));
      ldsBytes = mrp_ll_lds_bytes(ldsNodes, rows, rowWords, ldsPaths);
    } else {
      ldsNodes = 0;
    }
  }
  P.lds_nodes = ldsNodes;
  P.lds_rows = rows;
  P.lds_row_words = rowWords;
  P.lds_paths_bytes = ldsNodes ? ldsPaths : 0;
  if (kDebug) {
    if (!ctx->debugHost) {
      HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->debugHost), 16 * 4 * 4096, hipHostMallocMapped | hipHostMallocCoherent));
    }
    std::memset(ctx->debugHost, 0, 16 * 4 * 4096);
    void* dptr = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer(&dptr, ctx->debugHost, 0));
    P.debug = static_cast<volatile uint32_t*>(dptr);
  }
  uint32_t grid = std::min<uint32_t>(static_cast<uint32_t>(nJobs), static_cast<uint32_t>(ctx->opt.slots));
  t.queueBase += static_cast<uint32_t>(nJobs) + grid;  // every workgroup takes one ticket past the end when it exits
  HIPCHK(ctx, hipEventRecord(t.evK0, t.stream));
  HIPCHK(ctx, mrp_ll_launch(&P, grid, ldsBytes, t.stream));
  HIPCHK(ctx, hipEventRecord(t.evK1, t.stream));
  ctx->stats.launches += 1;
  return MRP_LL_SUCCESS;
}

int mrp_ll_wait(mrp_ll_ctx* ctx, int32_t ticket) {
  if (!ctx || ticket < 0 || ticket >= static_cast<int32_t>(ctx->tickets.size())) return MRP_LL_E_INVALID;
  Ticket& t = ctx->tickets[ticket];
  if (!t.inFlight) return MRP_LL_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (kDebug && ctx->debugHost) {
    for (int spin = 0; spin < 100 && hipEventQuery(t.evK1) == hipErrorNotReady; ++spin) {
      struct timespec ts = {0, 100000000};
      nanosleep(&ts, nullptr);
    }
    if (hipEventQuery(t.evK1) == hipErrorNotReady) {
      std::fprintf(stderr, "[mrp_ll] kernel did not finish within 10 s; trace of the first workgroups:\n");
      for (int b = 0; b < 4; ++b) {
        std::fprintf(stderr, "  wg %d:", b);
        for (int k = 0; k < 16; ++k) std::fprintf(stderr, " %u", ctx->debugHost[b * 16 + k]);
        std::fprintf(stderr, "\n");
      }
      std::fflush(stderr);
      std::_Exit(3);
    }
  }
  t.inFlight = false;
  if (t.nJobs == 0) return MRP_LL_SUCCESS;
  HIPCHK(ctx, hipEventSynchronize(t.evK1));
  if (t.nJobs == 0) return MRP_LL_SUCCESS;
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, t.evK0, t.evK1) == hipSuccess) ctx->stats.kernel_ms += ms;
  const uint32_t outStride = static_cast<uint32_t>(ctx->opt.max_horizon);
  auto unpackT0 = std::chrono::steady_clock::now();
  for (int i = 0; i < t.nJobs; ++i) {
    mrp_ll_result& r = t.userResults[i];
    const DevResult& d = t.results.host[i];
    if (t.rejected[i]) {
      r.status = MRP_LL_BAD_JOB;
      r.cost = r.fmin = r.n_states = 0;
      r.expanded = 0;
      r.tier = 0;
      continue;
    }
    r.status = d.status;
    r.cost = d.cost;
    r.fmin = d.fmin;
    r.n_states = d.status == mrp::ST_OK ? d.n_states : 0;
    r.expanded = d.expanded;
    r.tier = static_cast<int32_t>(d.tier);
    ctx->stats.jobs += 1;
    ctx->stats.expansions += d.expanded;
    ctx->stats.nodes_created += d.nodes_created;
    ctx->stats.migrated += d.tier ? 1 : 0;
    for (int q = 0; q < 8; ++q) ctx->stats.prof[q] += d.prof[q];
    if (d.status == mrp::ST_OK) {
      const uint16_t* p = t.outPaths.host + static_cast<size_t>(i) * outStride;
      int n = d.n_states;
      int lim = std::min(n, r.states_cap);
      if (r.states_txy)
        for (int k = 0; k < lim; ++k) {
          r.states_txy[3 * k] = k;
          r.states_txy[3 * k + 1] = p[k] & 0xFF;
          r.states_txy[3 * k + 2] = p[k] >> 8;
        }
      if (r.actions)
        for (int k = 0; k + 1 < n && k < r.states_cap; ++k)
          r.actions[k] = actionFromDelta((p[k + 1] & 0xFF) - (p[k] & 0xFF), (p[k + 1] >> 8) - (p[k] >> 8));
      if ((r.states_txy || r.actions) && r.states_cap < n) r.status = MRP_LL_PATH_TRUNCATED;
    }
  }
  ctx->stats.unpack_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - unpackT0).count();
  return MRP_LL_SUCCESS;
}

int mrp_ll_search_batch(mrp_ll_ctx* ctx, int32_t nJobs, const mrp_ll_job* jobs, mrp_ll_result* results) {
  int32_t ticket = -1;
  int rc = mrp_ll_submit(ctx, nJobs, jobs, results, &ticket);
  if (rc != MRP_LL_SUCCESS) return rc;
  return mrp_ll_wait(ctx, ticket);
}

int mrp_ll_get_stats(const mrp_ll_ctx* ctx, mrp_ll_stats* out) {
  if (!ctx || !out) return MRP_LL_E_INVALID;
  *out = ctx->stats;
  return MRP_LL_SUCCESS;
}

int mrp_ll_reset_stats(mrp_ll_ctx* ctx) {
  if (!ctx) return MRP_LL_E_INVALID;
  std::memset(&ctx->stats, 0, sizeof(ctx->stats));
  return MRP_LL_SUCCESS;
}

}  // extern "C"
