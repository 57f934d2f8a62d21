// C-ABI of the MI355X low-level search engine (include/mrp_ll.h): context, map upload, batch packing, launch.
// No CPU fallback: every entry point needs a working HIP device.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <climits>
#include <cstring>
#include <ctime>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mrp_ll.h"
#include "ll_device.h"

extern "C" uint32_t mrp_ll_lds_bytes(int kind, uint32_t capNodes, uint32_t rows, uint32_t rowWords, uint32_t pathBytes);
extern "C" int mrp_ll_persistent_occupancy(int kind, uint32_t ldsBytes);
extern "C" int mrp_ll_sipp_persistent_occupancy(void);
extern "C" hipError_t mrp_ll_launch(const mrp::LaunchParams* P, uint32_t grid, uint32_t ldsBytes, int kind,
                                    hipStream_t stream);
extern "C" hipError_t mrp_ll_launch_sipp(const mrp::LaunchParams* P, uint32_t grid, hipStream_t stream);
extern "C" hipError_t mrp_ll_launch_sipp_persistent(const mrp::LaunchParams* P, uint32_t grid, hipStream_t stream);
extern "C" hipError_t mrp_ll_launch_persistent(const mrp::LaunchParams* P, uint32_t grid, uint32_t ldsBytes, int kind,
                                               hipStream_t stream);
extern "C" uint32_t mrp_ll_heavy_lds_bytes(void);
extern "C" hipError_t mrp_ll_launch_front_heavy(const mrp::LaunchParams* P, uint32_t grid, uint32_t ldsBytes, int heavy,
                                                hipStream_t stream);
extern "C" int mrp_ll_front_heavy_occupancy(int heavy, uint32_t ldsBytes);

namespace mrp {
struct ConflictOut {
  int32_t found, time, agent1, agent2, type, x1, y1, x2, y2, count;
};
struct ConflictParams {
  const uint32_t* setFirstAgent;
  const uint32_t* pathFirstState;
  const uint16_t* states;
  ConflictOut* out;
  uint32_t nSets;
};
}  // namespace mrp
extern "C" hipError_t mrp_ll_launch_conflict(const mrp::ConflictParams* P, hipStream_t stream);

namespace {

using mrp::DevJob;
using mrp::DevResult;

struct MapRec {
  int32_t dimx, dimy;
  uint32_t wpr, wordOff;
};
struct HeurRec {  // MRP_LL_ASTAR_TA: shortest-path table of one goal cell, kHeurWords words inside the maps buffer
  int32_t mapId;
  uint32_t wordOff;
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
  std::vector<mrp_ll_sipp_table*> commitTab;  // per job: the table a sipp_commit job reports back to (else null)
  bool sipp = false;              // the batch holds MRP_LL_SIPP jobs (own kernel, own result format)
  int kind = 0;                   // A* batches: 0 = mixed, 1 = all A*-epsilon, 2 = all A*
  std::vector<int32_t> jobDimx;   // SIPP: grid width per job (cell -> x, y when unpacking)
  std::vector<int32_t> jobInit;   // per job: initial_cost (A*) / start_time (SIPP), applied when unpacking
};

// ---- session mode: job ring in coherent pinned host memory --------------------------------------------------
// Two levels: a TICKET ring per lane, consumed strictly in order by the workgroups (entry = generation << 11 | job
// slot), and a pool of job SLOTS (descriptor, constraint words, path table, result, path) handed out from a free
// list.  A slot is tied up until its result has been consumed, a ticket entry only until a workgroup has started the
// job, so one long search never blocks the publication of the searches behind it.
struct Ring {
  static constexpr uint32_t kSlots = mrp::kRingSlots;     // job slots, shared by both lanes (the device masks with the same constant)
  static constexpr uint32_t kReserve1 = 0;                // (round 1: slots kept free for a device-side priority lane)
  static constexpr uint32_t kTickets0 = 1u << 17;         // ticket-ring entries of lane 0 / lane 1
  static constexpr uint32_t kTickets1 = 1u << 14;
  static constexpr uint32_t kTickets = kTickets0 + kTickets1;
  static constexpr uint32_t kSlotConsWords = 1024;        // 4 KB of constraint words per job
  static constexpr uint32_t kSlotPathHalfs = 16 * 1024;   // 32 KB path table per job
  uint8_t* block = nullptr;        // pinned host memory: what the DEVICE writes and the host reads (done words, completion
                                   // queue, results, output paths) — and, without a large BAR, everything else too
  uint8_t* push = nullptr;         // what the HOST writes and the device reads (ticket entries, head / stop / heartbeat words,
                                   // job descriptors, constraint words, path tables): pinned host memory, or — with a large
                                   // BAR and MRP_LL_RING_IN_DEVICE=1 — UNCACHED DEVICE memory the host stores into directly
                                   // (write-combined, posted PCIe writes), so that the resident workgroups never read host
                                   // memory.  Measured the same within 1 % once the per-job cache fences were gone.
  bool pushInDevice = false;
  size_t pushBytes = 0;
  uint64_t lastBeatTsc = 0;
  uint32_t *state = nullptr, *done = nullptr, *stop = nullptr, *headWord = nullptr, *compRing = nullptr;
  uint32_t* compCountDev = nullptr;  // device counter
  unsigned long long* ticksDev = nullptr;  // device [2]: busy / idle ticks of the session's workgroups
  uint64_t compCursor = 0;           // next completion-queue entry the host expects
  uint32_t heartbeat = 0;            // bumped on every submit / poll: resident workgroups leave only when it stands still
  uint32_t emptyPolls = 0;           // consecutive polls that found nothing (liveness check of the resident kernel)
  uint32_t inFlightJobs = 0;         // published, result not consumed yet
  uint32_t idleLimitS = 20;
  DevJob* jobs = nullptr;
  DevResult* results = nullptr;
  uint16_t* outPaths = nullptr;
  uint32_t* cons = nullptr;
  uint16_t* paths = nullptr;
  uint32_t outStride = 0;
  uint64_t head[2] = {0, 0};       // per lane: next ticket number to publish
  std::vector<uint8_t> busy;       // slot holds a job whose result the host has not consumed yet
  std::vector<int32_t> slotTicket; // slot -> session ticket id / job index inside it
  std::vector<int32_t> slotJob;
  std::vector<uint32_t> slotGen;   // value the occupant's done word will show: its ticket number + 1 (valid while busy)
  std::vector<uint32_t> freeSlots;     // job slots not in use (stack)
  std::vector<uint32_t> tkSlot, tkSeq; // per ticket-ring entry: the slot / done value of the job last published there
  uint32_t Q[2] = {kTickets0, kTickets1};  // ticket-ring entries in use (MRP_LL_TICKET_RING shrinks them: wrap tests)
  bool active = false;
  uint32_t grid = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  // Optional second resident launch of the same session (MRP_LL_EXTRA_HBM_WGS): workgroups WITHOUT an LDS tier (their
  // searches live in the HBM arena / L2 from the start) on a stream of their own.  They take tickets from the same
  // rings; the LDS tier caps the first launch at floor(160 KiB / tier bytes) workgroups per CU, these fill SIMD issue
  // slots beyond that.
  uint32_t grid2 = 0;
  hipStream_t stream2 = nullptr;
  hipEvent_t ev2 = nullptr;
  // The heavy workgroups of an A*-epsilon session (mrp_ll_session_begin_tiers; ll_device.h heavy_q): the second launch is
  // then the heavy kernel, the first one the front kernel.
  bool heavy = false;
  unsigned long long* heavyQ = nullptr;   // device: kRingSlots entries
  uint32_t* heavyCtr = nullptr;           // device: [0] written, [16] taken
  uint32_t* heavyAlive = nullptr;         // pinned host: one word per heavy workgroup
  // SIPP sessions (mrp_ll_session_begin_sipp): the jobs' safe-interval tables are far larger than a slot's constraint
  // area, so they get their own pinned buffer, and only the first kSippSlots job slots are used
  static constexpr uint32_t kSippSlots = 512;
  bool sipp = false;
  int kind = 0;                    // A* sessions: 0 = mixed kernel, 1 = A*-epsilon jobs only, 2 = A* jobs only
  uint32_t* sippCons = nullptr;
  uint32_t sippSlotWords = 0;      // capacity per slot the buffer was allocated with
  std::vector<int32_t> slotDimx;   // SIPP: grid width of the slot's job (cell -> x, y when unpacking)
  std::vector<mrp_ll_sipp_table*> slotTable;  // SIPP: the table of the slot's job when the job has to report back to it
  std::vector<uint8_t> slotSippFlags;         //       bit 0: it runs on the device-resident copy, bit 1: sipp_commit
  std::vector<int32_t> slotInit;   // initial_cost (A*) / start_time (SIPP) of the slot's job
  std::vector<int32_t> slotChain;  // MRP_LL_JOB_ROOT_CHAIN: results the slot's job fills (n_agents - agent_idx), else 0
};
// The host's stores into the push block are write-combined when it is device memory: everything written so far leaves
// the core's buffers, in order, before whatever is stored next (x86 SFENCE; a no-op price for pinned host memory).
static inline void pushFence(const Ring& g) {
  if (g.pushInDevice) __builtin_ia32_sfence();
}
struct SessTicket {
  bool used = false;
  int32_t lane = 0;
  int32_t tag = -1;                // mrp_ll_submit_tagged: which of the context's co-workers the ticket belongs to (-1: untagged)
  int32_t n = 0, remaining = 0;
  mrp_ll_result* res = nullptr;
  std::vector<uint8_t> state;      // per job: 0 pending, 1 consumed, 2 rejected on the host
  std::vector<uint32_t> slots;     // per job: its job slot
  std::vector<uint32_t> seq;       // per job: the done value that marks it finished
};

}  // namespace

struct mrp_ll_ctx {
  mrp_ll_options opt;
  int device = 0;
  std::string err;
  std::vector<MapRec> maps;
  std::vector<uint32_t> mapWords;  // host copy of all obstacle bitmaps (and of the heuristic tables of MRP_LL_ASTAR_TA)
  std::vector<HeurRec> heurs;
  uint32_t* mapsDev = nullptr;
  size_t mapsDevCap = 0;
  bool mapsDirty = false;
  uint32_t maxWpr = 1;
  uint32_t extraHbmWgs = 0;       // session mode: additional resident workgroups without an LDS tier (see Ring::grid2)
  uint32_t tierRows = 64, tierPathBytes = 4096;  // LDS tier geometry (mrp_ll_configure_tiers); nodes live in opt.lds_nodes
  uint32_t sessionRowWords = 0;   // LDS bitmap row width the resident kernel was launched with
  uint32_t sessionLdsPathBytes = 0;  // ... and the bytes of its window that hold the focal path table (0: no compact tier)
  uint32_t arenaRowWords = 0;
  uint64_t arenaStride = 0;
  uint32_t arenaScratchOff = 0;
  uint32_t arenaPathsBytes = 0;
  std::vector<Ticket> tickets;
  mrp_ll_stats stats;
  uint32_t* debugHost = nullptr;  // MRP_LL_DEBUG: host-mapped trace buffer
  Ring ring;
  std::vector<SessTicket> sess;
  std::vector<int32_t> sessFree;   // free session-ticket ids (stack)
  // co-workers (mrp_ll_submit_tagged / mrp_ll_poll_any_tagged): two host threads that share this context's session
  static constexpr int kMaxTags = 4;
  std::mutex coMu;
  std::vector<int32_t> coStash[kMaxTags];        // finished tickets another co-worker's poll has come across (not released yet)
  std::atomic<int32_t> coStashCount[kMaxTags];
  void* sippScratch = nullptr;     // SippScratch, created on first use (packSipp)
  // device-resident SIPP tables: chunks of sippTabsPerChunk tables of sippTabStride bytes each (about 64 MB a chunk)
  std::vector<uint8_t*> sippTabChunks;
  std::vector<int32_t> sippTabFree;
  int32_t sippTabNext = 0;
  size_t sippTabStride = 0;
  int32_t sippTabsPerChunk = 64;
  uint16_t* pathStore = nullptr;   // device-resident path store (mrp_ll_path_store_reserve)
  uint32_t pathStoreStride = 0, pathStoreSlots = 0;
  uint8_t* scanDev = nullptr;      // mrp_ll_conflict_scan: device staging (grown on demand)
  size_t scanDevCap = 0;
  hipStream_t scanStream = nullptr;  // ... and its own stream: a session's resident kernel occupies tickets[0].stream
  std::vector<uint16_t> scanStates;
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
    size_t ncap = std::max<size_t>(need * 2, 1u << 20);  // >= 4 MB: room for in-session uploads
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->mapsDev), ncap * sizeof(uint32_t)));
    ctx->mapsDevCap = ncap;
  }
  HIPCHK(ctx, hipMemcpy(ctx->mapsDev, ctx->mapWords.data(), ctx->mapWords.size() * sizeof(uint32_t),
                        hipMemcpyHostToDevice));
  ctx->mapsDirty = false;
  return MRP_LL_SUCCESS;
}

// Where a packed job's constraint words / path table go: a growable batch buffer or a fixed ring-slot area.
struct ConsSinkBuf {
  PinnedBuf<uint32_t>& b;
  bool failed = false;
  size_t size() const { return b.size; }
  bool fits(size_t) const { return true; }
  void push(uint32_t w) {
    if (b.push(w) != hipSuccess) failed = true;
  }
  uint32_t* grow(size_t n) {  // n consecutive words, written by the caller
    const size_t at = b.size;
    if (b.resize(at + n) != hipSuccess) {
      failed = true;
      return nullptr;
    }
    return b.host + at;
  }
};
struct PathSinkBuf {
  PinnedBuf<uint16_t>& b;
  bool failed = false;
  uint16_t* alloc(size_t n, uint32_t& off) {
    size_t base = (b.size + 7u) & ~size_t(7);  // 16-byte aligned table start
    if (b.resize(base + n) != hipSuccess) {
      failed = true;
      return nullptr;
    }
    off = static_cast<uint32_t>(base);
    return b.host + base;
  }
};
struct ConsSinkSlot {
  uint32_t* area;
  uint32_t baseOff, cap, used = 0;
  bool failed = false;
  size_t size() const { return baseOff + used; }
  bool fits(size_t n) const { return used + n <= cap; }
  void push(uint32_t w) {
    if (used >= cap) {
      failed = true;
      return;
    }
    area[used++] = w;
  }
  uint32_t* grow(size_t n) {
    if (used + n > cap) {
      failed = true;
      return nullptr;
    }
    uint32_t* p = area + used;
    used += static_cast<uint32_t>(n);
    return p;
  }
};
struct PathSinkSlot {
  uint16_t* area;
  uint32_t baseOff, cap;
  bool failed = false;
  uint16_t* alloc(size_t n, uint32_t& off) {
    if (n > cap) {
      failed = true;
      return nullptr;
    }
    off = baseOff;
    return area;
  }
};

// SIPP job tables (see runSipp in ll_kernel.hip).  Safe intervals are derived from the collision intervals exactly as
// SIPPEnvironment::setCollisionIntervals does (sipp.hpp:245-284): sort by start; a safe interval [start, ci.start-1]
// in front of every collision interval when non-empty; a final [start, INT_MAX] unless the last one ends at INT_MAX.
struct SippScratch {  // reused across jobs of a context: packSipp allocates nothing in the steady state
  struct Iv { int32_t s, e; };
  std::vector<int32_t> cellIdx;            // cell -> special index + 1
  std::vector<uint32_t> first, count;      // per special cell: block of safe intervals inside `pool`
  std::vector<Iv> pool, ci;
};

SippScratch& sippScratchOf(mrp_ll_ctx* ctx) {
  if (!ctx->sippScratch) ctx->sippScratch = new SippScratch();
  return *static_cast<SippScratch*>(ctx->sippScratch);
}

// setCollisionIntervals for one location (sipp.hpp:245-284): collision intervals sorted by start -> safe intervals
// appended to `out`.  `scratch` holds the sorted copy.
void safeFromCollisions(const int32_t* civ, int cnt, std::vector<SippScratch::Iv>& scratch,
                        std::vector<SippScratch::Iv>& out) {
  typedef SippScratch::Iv Iv;
  scratch.clear();
  bool sorted = true;
  for (int k = 0; k < cnt; ++k) {
    scratch.push_back(Iv{civ[2 * k], civ[2 * k + 1]});
    if (k && scratch[k].s < scratch[k - 1].s) sorted = false;
  }
  if (!sorted) std::stable_sort(scratch.begin(), scratch.end(), [](const Iv& a, const Iv& b) { return a.s < b.s; });
  long long start = 0;
  int32_t lastEnd = 0;
  for (const Iv& c : scratch) {
    if (start <= static_cast<long long>(c.s) - 1) out.push_back(Iv{static_cast<int32_t>(start), c.s - 1});
    start = static_cast<long long>(c.e) + 1;
    lastEnd = c.e;
  }
  if (lastEnd < INT32_MAX) out.push_back(Iv{static_cast<int32_t>(start), INT32_MAX});
}

}  // namespace

// Incrementally maintained safe-interval table of one agent-planning context (include/mrp_ll.h, mrp_ll_sipp_table_*):
// what SIPP::setCollisionIntervals would hold after the same calls, kept per cell so that adding one collision interval
// recomputes one cell's list, and a job only has to be COPIED into its slot instead of being rebuilt from every
// collision interval of the instance.
struct mrp_ll_sipp_table {
  int32_t mapId = -1, dimx = 0, dimy = 0;
  std::vector<int32_t> cellIdx;                       // cell -> special index + 1
  std::vector<uint16_t> cellIdx16;                    // the same as the device reads it
  struct Spec {
    std::vector<int32_t> collisions;                  // [n][2] in the order they were added
    std::vector<SippScratch::Iv> safe;
    bool disjoint = true;                             // no two collision intervals of the cell have overlapped so far
  };
  std::vector<Spec> spec;
  uint32_t totalSafe = 0;
  std::vector<SippScratch::Iv> scratch;
  // device-resident copy (session mode): the jobs carry only the cells that changed since the previous job
  mrp_ll_ctx* ctx = nullptr;
  int32_t devIndex = -1;                              // slot in the engine's table pool (-1: none)
  bool devFresh = true;                               // the device copy has never been written: the next job resets it
  bool inFlight = false;                              // a job is using (and writing) the device copy
  uint32_t epoch = 0;                                 // of the last job (status words of other epochs read as unseen)
  bool overflow = false;                              // some cell has more than kSippCap safe intervals: ship whole tables
  std::vector<int32_t> dirty;                         // cells changed since the last job was packed
  std::vector<uint8_t> isDirty;
  // sipp_commit: stays (cell, start, end) the DEVICE copy already holds and this host copy does not yet; replayed
  // (sippTableSync) before anything reads the host copy
  std::vector<int32_t> log;
};

namespace {
void sippTableAddCell(mrp_ll_sipp_table* t, size_t cell, int32_t start, int32_t end, bool markDirty);
void sippTableSync(mrp_ll_sipp_table* t) {
  for (size_t k = 0; k + 2 < t->log.size(); k += 3)
    sippTableAddCell(t, static_cast<size_t>(t->log[k]), t->log[k + 1], t->log[k + 2], false);
  t->log.clear();
}
// the stays of a raw solution (cell | arrival << 16 per state): one collision interval per state (mrp_ll.h, sipp_commit)
template <class F>
void forEachStay(const uint32_t* raw, int n, F&& f) {
  for (int k = 0; k < n; ++k)
    f(static_cast<int32_t>(raw[k] & 0xFFFFu), static_cast<int32_t>(raw[k] >> 16),
      k + 1 < n ? static_cast<int32_t>(raw[k + 1] >> 16) - 1 : INT32_MAX);
}
// A job on an mrp_ll_sipp_table has come back.  flags: bit 0 = it ran on the device-resident copy, bit 1 = sipp_commit.
void finishSippTableJob(mrp_ll_sipp_table* T, uint32_t flags, const mrp::DevResult& d, const uint16_t* rawPath) {
  if (flags & 1u) T->inFlight = false;
  if (!(flags & 2u) || d.status != mrp::ST_OK) return;
  const uint32_t* raw = reinterpret_cast<const uint32_t*>(rawPath);
  if (flags & 1u) {
    // the workgroup has already put the stays into the device copy; this copy catches up when somebody needs it
    forEachStay(raw, d.n_states, [&](int32_t cell, int32_t s0, int32_t e0) {
      T->log.push_back(cell);
      T->log.push_back(s0);
      T->log.push_back(e0);
    });
    if (d.tier & mrp::kSippTierCommitFailed) {  // ... unless a stay did not fit the fixed layout: redo the table here,
      sippTableSync(T);                         // and from now on it travels whole
      T->overflow = true;
      T->devFresh = true;
    }
  } else {
    if (!T->log.empty()) sippTableSync(T);
    forEachStay(raw, d.n_states, [&](int32_t cell, int32_t s0, int32_t e0) { sippTableAddCell(T, cell, s0, e0, true); });
  }
}
}  // namespace

namespace {

// The table of a job from an mrp_ll_sipp_table: cellIdx[cells], specFirst[K + 1], ivals[total][2] (see runSipp).
template <class ConsSink>
bool packSippFromTable(const mrp_ll_job& j, const MapRec& mp, ConsSink& cs, DevJob& d) {
  if (!j.sipp_table->log.empty()) sippTableSync(const_cast<mrp_ll_sipp_table*>(j.sipp_table));
  const mrp_ll_sipp_table& T = *j.sipp_table;
  if (T.dimx != mp.dimx || T.dimy != mp.dimy) return false;
  const int cells = mp.dimx * mp.dimy;
  const uint32_t K = static_cast<uint32_t>(T.spec.size());
  d.algo = MRP_LL_SIPP;
  d.max_expansions = j.max_expansions;
  d.vc_off = static_cast<uint32_t>(cs.size());
  const size_t cw = (static_cast<size_t>(cells) + 1) / 2;  // cellIdx travels as halfwords
  uint32_t* w = cs.grow(cw + K + 1 + 2 * static_cast<size_t>(T.totalSafe));
  if (!w) return false;
  w[cw - 1] = 0;
  std::memcpy(w, T.cellIdx16.data(), sizeof(uint16_t) * cells);
  w += cw;
  uint32_t run = 0;
  for (uint32_t k = 0; k < K; ++k) {
    w[k] = run;
    run += static_cast<uint32_t>(T.spec[k].safe.size());
  }
  w[K] = run;
  w += K + 1;
  for (uint32_t k = 0; k < K; ++k) {
    std::memcpy(w, T.spec[k].safe.data(), sizeof(SippScratch::Iv) * T.spec[k].safe.size());
    w += 2 * T.spec[k].safe.size();
  }
  d.n_vc = K;
  d.n_ec = T.totalSafe;
  d.ec_off = 0;
  d.n_agents_pad = 0;
  d.path_off = 0;
  const int32_t startTime = j.initial_cost;
  if (startTime > static_cast<int32_t>(mrp::kGMask)) return false;
  d.last_goal_constraint = startTime;
  const int sc = j.start_y * mp.dimx + j.start_x;
  int startIv = -1;
  if (!T.cellIdx[sc]) {
    startIv = 0;
  } else {
    const auto& v = T.spec[T.cellIdx[sc] - 1].safe;
    for (size_t k = 0; k < v.size(); ++k)
      if (v[k].s <= startTime && v[k].e >= startTime) {
        startIv = static_cast<int>(k);
        break;
      }
  }
  d.t_pad = startIv < 0 ? 0xFFFFFFFFu : static_cast<uint32_t>(startIv);
  return !cs.failed;
}


// Session mode: the job carries the delta of a device-resident table (ll_device.h kSippResident).  `T` is updated (its
// dirty list is consumed, it is marked in flight), so the job MUST run — sessionSubmit publishes it right away.
template <class ConsSink>
bool packSippResident(mrp_ll_ctx* ctx, const mrp_ll_job& j, const MapRec& mp, ConsSink& cs, DevJob& d) {
  mrp_ll_sipp_table& T = *const_cast<mrp_ll_sipp_table*>(j.sipp_table);
  if (T.dimx != mp.dimx || T.dimy != mp.dimy) return false;
  const int cells = mp.dimx * mp.dimy;
  const int32_t startTime = j.initial_cost;
  if (startTime > static_cast<int32_t>(mrp::kGMask)) return false;
  const bool fresh = T.devFresh || T.epoch >= mrp::kSippEpochMax;  // epochs used up: start over from a zeroed table
  if (fresh && !T.log.empty()) sippTableSync(&T);
  if (T.overflow) return false;
  const size_t nRec = fresh ? T.spec.size() : T.dirty.size();
  // every record of one job has room for the same number of intervals: the longest list among them, as a power of two >= 2
  size_t longest = 0;
  if (fresh) {
    for (const mrp_ll_sipp_table::Spec& sp : T.spec) longest = std::max(longest, sp.safe.size());
  } else {
    for (int32_t cell : T.dirty) longest = std::max(longest, T.spec[T.cellIdx[cell] - 1].safe.size());
  }
  size_t recIv = 4;  // bounds words per record: whole 16-byte units
  while (recIv < longest) recIv *= 2;
  const size_t hdrWords = (nRec + 3) & ~size_t(3);
  if ((cs.size() & 3u) != 0 || !cs.fits(hdrWords + nRec * recIv)) return false;  // nothing consumed yet
  d.vc_off = static_cast<uint32_t>(cs.size());
  uint32_t* hdr = cs.grow(hdrWords + nRec * recIv);
  if (!hdr) return false;
  uint32_t* body = hdr + hdrWords;
  size_t r = 0;
  auto emit = [&](int32_t cell) {
    const mrp_ll_sipp_table::Spec& sp = T.spec[T.cellIdx[cell] - 1];
    hdr[r] = static_cast<uint32_t>(cell) | (static_cast<uint32_t>(sp.safe.size()) << 16);
    uint32_t* b = body + r * recIv;
    for (size_t q = 0; q < recIv; ++q)  // start | end << 16 (ll_device.h; T.overflow has vouched for the ranges)
      b[q] = q < sp.safe.size() ? static_cast<uint32_t>(sp.safe[q].s) |
                                      (sp.safe[q].e == INT32_MAX ? mrp::kSippEndInf : static_cast<uint32_t>(sp.safe[q].e)) << 16
                                : 0u;
    if (recIv == mrp::kSippRowWords) b[15] = static_cast<uint32_t>(sp.safe.size()) + 1u;  // a whole row: its last word is the count
    r += 1;
  };
  if (fresh) {
    for (int32_t cell = 0; cell < cells; ++cell)
      if (T.cellIdx[cell]) emit(cell);
    T.epoch = 0;
  } else {
    for (int32_t cell : T.dirty) emit(cell);
  }
  for (int32_t cell : T.dirty) T.isDirty[cell] = 0;
  T.dirty.clear();
  T.devFresh = false;
  T.epoch += 1;
  T.inFlight = true;
  const uint64_t addr = reinterpret_cast<uint64_t>(ctx->sippTabChunks[T.devIndex / ctx->sippTabsPerChunk]) +
                        static_cast<uint64_t>(T.devIndex % ctx->sippTabsPerChunk) * ctx->sippTabStride;
  d.algo = MRP_LL_SIPP;
  d.max_expansions = j.max_expansions;
  d.n_agents_pad = static_cast<uint32_t>(addr);
  d.path_off = static_cast<uint32_t>(addr >> 32);
  static const bool noLds = [] {
    const char* e = std::getenv("MRP_LL_SIPP_NO_LDS");  // tier comparison (tests, probes)
    return e && *e == '1';
  }();
  d.ctx_flags = mrp::kSippResident | (noLds ? mrp::kSippNoLds : 0u);
  d.n_ctx = T.epoch;
  d.n_vc = static_cast<uint32_t>(recIv);  // intervals per delta record
  d.n_ec = T.totalSafe;
  d.ec_off = static_cast<uint32_t>(nRec) | (fresh ? 0x80000000u : 0u);
  d.last_goal_constraint = startTime;
  // the start interval (findSafeInterval, sipp.hpp:286-296) is looked up by the workgroup: with sipp_commit the
  // device copy is ahead of this one
  d.t_pad = 0;
  if (j.sipp_commit) d.ctx_flags |= mrp::kSippCommit;
  return true;
}

template <class ConsSink>
bool packSipp(mrp_ll_ctx* ctx, const mrp_ll_job& j, const MapRec& mp, ConsSink& cs, DevJob& d) {
  if (j.sipp_table) {
    const mrp_ll_sipp_table& T = *j.sipp_table;
    // resident form: in a session, for a table of this engine that fits the fixed layout (one job per table in flight)
    if (ctx->ring.active && T.ctx == ctx && T.devIndex >= 0 && !T.overflow && !T.inFlight &&
        packSippResident(ctx, j, mp, cs, d))
      return true;
    return packSippFromTable(j, mp, cs, d);
  }
  const int cells = mp.dimx * mp.dimy;
  if (j.n_collision_locations < 0) return false;
  if (j.n_collision_locations > 0 && (!j.collision_xy || !j.collision_count || !j.collision_intervals)) return false;
  typedef SippScratch::Iv Iv;
  SippScratch& sc0 = sippScratchOf(ctx);
  std::vector<int32_t>& cellIdx = sc0.cellIdx;
  cellIdx.assign(cells, 0);
  sc0.first.clear();
  sc0.count.clear();
  sc0.pool.clear();
  size_t off = 0;
  for (int n = 0; n < j.n_collision_locations; ++n) {
    const int x = j.collision_xy[2 * n], y = j.collision_xy[2 * n + 1];
    const int cnt = j.collision_count[n];
    const int32_t* civ = j.collision_intervals + 2 * off;
    off += cnt;
    if (x < 0 || x >= mp.dimx || y < 0 || y >= mp.dimy) continue;  // never visited
    const int cell = y * mp.dimx + x;
    const uint32_t p0 = static_cast<uint32_t>(sc0.pool.size());
    if (cnt > 0) safeFromCollisions(civ, cnt, sc0.ci, sc0.pool);
    const uint32_t nSafe = static_cast<uint32_t>(sc0.pool.size()) - p0;
    // erase + re-create (sipp.hpp:247-251): an empty list restores the default single interval
    if (cellIdx[cell]) {
      const int k = cellIdx[cell] - 1;
      if (cnt == 0) {
        sc0.first[k] = static_cast<uint32_t>(sc0.pool.size());
        sc0.pool.push_back(Iv{0, INT32_MAX});
        sc0.count[k] = 1;
      } else {
        sc0.first[k] = p0;
        sc0.count[k] = nSafe;
      }
    } else if (cnt > 0) {
      sc0.first.push_back(p0);
      sc0.count.push_back(nSafe);
      cellIdx[cell] = static_cast<int32_t>(sc0.first.size());
    }
  }
  const uint32_t K = static_cast<uint32_t>(sc0.first.size());
  uint32_t total = 0;
  for (uint32_t k = 0; k < K; ++k) total += sc0.count[k];
  d.algo = MRP_LL_SIPP;
  d.max_expansions = j.max_expansions;
  d.vc_off = static_cast<uint32_t>(cs.size());
  {  // cellIdx[cells], specFirst[K + 1], ivals[total][2]
    const size_t cw = (static_cast<size_t>(cells) + 1) / 2;  // cellIdx travels as halfwords
    uint32_t* w = cs.grow(cw + K + 1 + 2 * static_cast<size_t>(total));
    if (!w) return false;
    w[cw - 1] = 0;
    uint16_t* w16 = reinterpret_cast<uint16_t*>(w);
    for (int c = 0; c < cells; ++c) w16[c] = static_cast<uint16_t>(cellIdx[c]);
    w += cw;
    uint32_t run = 0;
    for (uint32_t k = 0; k < K; ++k) {
      w[k] = run;
      run += sc0.count[k];
    }
    w[K] = run;
    w += K + 1;
    for (uint32_t k = 0; k < K; ++k) {
      std::memcpy(w, sc0.pool.data() + sc0.first[k], sizeof(Iv) * sc0.count[k]);
      w += 2 * sc0.count[k];
    }
  }
  d.n_vc = K;
  d.n_ec = total;
  d.ec_off = 0;
  d.n_agents_pad = 0;
  d.path_off = 0;
  // SIPP::search(..., startTime) (sipp.hpp:92-103): carried in the field the A* kernels use for m_lastGoalConstraint
  const int32_t startTime = j.initial_cost;
  if (startTime > static_cast<int32_t>(mrp::kGMask)) return false;
  d.last_goal_constraint = startTime;
  // findSafeInterval(start, startTime) (sipp.hpp:98-100,286-296): no interval -> search() returns false
  const int sc = j.start_y * mp.dimx + j.start_x;
  int startIv = -1;
  if (!cellIdx[sc]) {
    startIv = 0;
  } else {
    const Iv* v = sc0.pool.data() + sc0.first[cellIdx[sc] - 1];
    for (uint32_t k = 0; k < sc0.count[cellIdx[sc] - 1]; ++k)
      if (v[k].s <= startTime && v[k].e >= startTime) {
        startIv = static_cast<int>(k);
        break;
      }
  }
  if (startIv < 0) {  // make the device report NO_SOLUTION: an empty open list cannot be encoded, so cap at 0 ...
    d.t_pad = 0xFFFFFFFFu;
  } else {
    d.t_pad = static_cast<uint32_t>(startIv);
  }
  return !cs.failed;
}

// Pack one job; returns false if the job is rejected (MRP_LL_BAD_JOB).
template <class ConsSink, class PathSink>
bool packJob(mrp_ll_ctx* ctx, const mrp_ll_job& j, ConsSink& cs, PathSink& ps, DevJob& d) {
  if (j.map_id < 0 || j.map_id >= static_cast<int32_t>(ctx->maps.size())) return false;
  const MapRec& mp = ctx->maps[j.map_id];
  if (j.algo != MRP_LL_ASTAR && j.algo != MRP_LL_ASTAR_EPS && j.algo != MRP_LL_SIPP && j.algo != MRP_LL_ASTAR_TA) return false;
  if (j.algo == MRP_LL_ASTAR_EPS && j.initial_cost != 0) return false;  // AStarEpsilon::search has no initialCost
  if (j.initial_cost < 0 || j.initial_cost >= 0x40000000) return false;  // (bit 30 of the per-job word marks MRP_LL_ASTAR_TA, jobInitOf)
  if (j.flags & MRP_LL_JOB_ROOT_CHAIN) {  // the root step of an ECBS conflict tree as one job (mrp_ll.h; ll_device.h kCtxChain)
    const int n = j.n_agents, first = j.agent_idx;
    if (j.algo != MRP_LL_ASTAR_EPS || !ctx->ring.active || ctx->ring.sipp || ctx->ring.kind != 1) return false;
    if (n < 1 || n > static_cast<int>(mrp::kChainMaxAgents) || first < 0 || first >= n) return false;
    if (!j.path_ids || !j.chain_starts_goals_xy || !ctx->pathStore || mp.dimx > 32 || mp.dimy > 32) return false;
    {  // what runChain (ll_kernel.hip) needs of the session: the compact tier, room for its focal table in the window,
       // an arena slot that holds the cameFrom table and the (time, cell) bitmap, room for the output in the job's host area
      const uint32_t npad = static_cast<uint32_t>((n + 15) & ~15);
      if (ctx->opt.lds_nodes == 0 || ctx->sessionLdsPathBytes == 0 || mrp::kChainRows * npad * 2u > ctx->sessionLdsPathBytes ||
          static_cast<uint64_t>(ctx->opt.arena_nodes) * 16u < 64u * 1024u + 8192u ||
          static_cast<uint64_t>(n) * mrp::kChainEntryWords * 2u + static_cast<uint64_t>(n) * 64u > ctx->ring.outStride)
        return false;
    }
    std::memset(&d, 0, sizeof(d));
    d.map_word_off = mp.wordOff;
    d.dimx = mp.dimx;
    d.dimy = mp.dimy;
    d.words_per_row = mp.wpr;
    d.algo = j.algo;
    d.w = j.w;
    d.max_expansions = j.max_expansions;
    d.last_goal_constraint = -1;
    d.ctx_flags = mrp::kCtxChain;
    d.n_ctx = static_cast<uint32_t>(n);
    d.t_pad = static_cast<uint32_t>(first);
    d.n_agents_pad = static_cast<uint32_t>((n + 15) & ~15);
    d.store_out_id = mrp::kNoStoreSlot;
    d.reserved = static_cast<uint32_t>(j.chain_count > 0 ? std::min(n, first + j.chain_count) : n);  // one past the last agent planned
    d.vc_off = static_cast<uint32_t>(cs.size());
    for (int a = 0; a < n; ++a) {
      const int32_t* q = j.chain_starts_goals_xy + 4 * a;
      if (q[0] < 0 || q[0] >= mp.dimx || q[1] < 0 || q[1] >= mp.dimy || q[2] < 0 || q[2] >= mp.dimx || q[3] < 0 || q[3] >= mp.dimy)
        return false;
      cs.push(static_cast<uint32_t>(q[0]) | (static_cast<uint32_t>(q[1]) << 8) | (static_cast<uint32_t>(q[2]) << 16) |
              (static_cast<uint32_t>(q[3]) << 24));
    }
    for (int a = 0; a < n; ++a) {
      if (j.path_ids[a] < 0 || static_cast<uint32_t>(j.path_ids[a]) >= ctx->pathStoreSlots) return false;
      cs.push(static_cast<uint32_t>(j.path_ids[a]));
    }
    return !cs.failed;
  }
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
  const bool taNoGoal = j.algo == MRP_LL_ASTAR_TA && (j.flags & MRP_LL_JOB_NO_GOAL) != 0;
  // a goal outside the grid can never be reached; keep the reference behaviour (search until open is exhausted /
  // capped) by parking it on an unreachable coordinate that still fits the 8-bit fields only if in range
  if (!taNoGoal && !inGrid(j.goal_x, j.goal_y)) return false;
  d.gx = taNoGoal ? 0 : j.goal_x;
  d.gy = taNoGoal ? 0 : j.goal_y;
  d.algo = j.algo;
  d.w = j.w;
  d.max_expansions = j.max_expansions;
  if (j.algo == MRP_LL_SIPP) return packSipp(ctx, j, mp, cs, d);
  if (j.algo == MRP_LL_ASTAR_TA) {
    // (the compact tier serves what fits it — maps up to 32 x 32, 64 + 64 constraints —, the arena tier the rest)
    if (j.initial_cost != 0) return false;
    if (!taNoGoal) {
      if (j.heuristic_id < 0 || j.heuristic_id >= static_cast<int32_t>(ctx->heurs.size()) ||
          ctx->heurs[j.heuristic_id].mapId != j.map_id)
        return false;
      d.path_off = ctx->heurs[j.heuristic_id].wordOff;
    } else {
      d.ctx_flags |= mrp::kTaNoGoal;
    }
  }
  // setLowLevelContext (ecbs.cpp:264-274): last vertex constraint on the goal cell (cbs_ta.cpp:283-303: of ANY cell when
  // the agent has no task)
  int lastGoal = -1;
  d.vc_off = static_cast<uint32_t>(cs.size());
  for (int i = 0; i < j.n_vertex_constraints; ++i) {
    const int32_t* v = j.vertex_constraints + 3 * i;
    if (taNoGoal || (v[1] == j.goal_x && v[2] == j.goal_y)) lastGoal = std::max(lastGoal, v[0]);
    if (v[0] < 0 || v[0] >= horizon || !inGrid(v[1], v[2])) continue;  // can never match a generated state
    cs.push((static_cast<uint32_t>(v[0]) << 16) | (static_cast<uint32_t>(v[2]) << 8) | static_cast<uint32_t>(v[1]));  // t, y, x
  }
  d.n_vc = static_cast<uint32_t>(cs.size()) - d.vc_off;
  d.last_goal_constraint = lastGoal;
  d.ec_off = static_cast<uint32_t>(cs.size());
  for (int i = 0; i < j.n_edge_constraints; ++i) {
    const int32_t* e = j.edge_constraints + 5 * i;
    int k = neighborIndexFromDelta(e[3] - e[1], e[4] - e[2]);
    if (k < 0 || e[0] < 0 || e[0] >= horizon || !inGrid(e[1], e[2])) continue;
    cs.push((static_cast<uint32_t>(e[0]) << 19) | (static_cast<uint32_t>(e[2] * mp.dimx + e[1]) << 3) |
            static_cast<uint32_t>(k));
  }
  d.n_ec = static_cast<uint32_t>(cs.size()) - d.ec_off;
  if (cs.failed) return false;
  if (j.algo == MRP_LL_ASTAR_TA) {
    const uint32_t heurOff = d.path_off;
    d.n_agents_pad = 0;
    d.t_pad = 0;
    d.path_off = heurOff;
    d.store_out_id = mrp::kNoStoreSlot;
    return true;
  }
  // focal context: time-major table of the other agents' cells, each path extended by its last cell
  d.n_agents_pad = 0;
  d.t_pad = 0;
  d.path_off = 0;
  if ((j.flags & MRP_LL_JOB_HEAVY) && j.algo == MRP_LL_ASTAR_EPS) d.ctx_flags |= mrp::kCtxHeavy;
  // the result path also goes to a path-store slot only when the caller says so (a zero-initialised job names no slot)
  const bool storeResult = (j.flags & MRP_LL_JOB_STORE_RESULT) != 0;
  d.store_out_id = (storeResult && j.result_path_id >= 0 && static_cast<uint32_t>(j.result_path_id) < ctx->pathStoreSlots)
                       ? static_cast<uint32_t>(j.result_path_id)
                       : mrp::kNoStoreSlot;
  if (storeResult && d.store_out_id == mrp::kNoStoreSlot) return false;  // no such slot (or no store reserved)
  if (j.algo == MRP_LL_ASTAR_EPS && j.n_agents > 0 && j.path_ids) {
    // f2: the CT node's paths by their path-store slots; the workgroup builds the table (ll_kernel.hip runJob)
    if (!j.path_len || !ctx->pathStore) return false;
    int tpad = 0;
    for (int a = 0; a < j.n_agents; ++a)
      if (a != j.agent_idx && j.path_len[a] > 0) {
        if (j.path_ids[a] < 0 || static_cast<uint32_t>(j.path_ids[a]) >= ctx->pathStoreSlots) return false;
        tpad = std::max(tpad, j.path_len[a]);
      }
    // The workgroup builds the table in LDS or in its arena slot's path area; a table that fits neither (a caller's
    // path_len beyond max_horizon, a very wide solution) is refused here rather than written past the slot.
    if (tpad > horizon ||
        static_cast<uint64_t>(tpad) * ((static_cast<uint32_t>(j.n_agents) + 15u) & ~15u) * 2u > ctx->arenaPathsBytes)
      return false;
    if (tpad > 0) {
      d.path_off = static_cast<uint32_t>(cs.size());
      for (int a = 0; a < j.n_agents; ++a)
        cs.push(a != j.agent_idx && j.path_len[a] > 0 ? static_cast<uint32_t>(j.path_ids[a]) : mrp::kNoStoreSlot);
      if (cs.failed) return false;
      d.ctx_flags |= mrp::kCtxById;
      d.n_ctx = static_cast<uint32_t>(j.n_agents);
      d.n_agents_pad = (static_cast<uint32_t>(j.n_agents) + 15u) & ~15u;
      d.t_pad = static_cast<uint32_t>(tpad);
    }
  } else if (j.algo == MRP_LL_ASTAR_EPS && j.n_agents > 0) {
    if (!j.path_len || !j.path_xy) return false;
    int tpad = 0;
    for (int a = 0; a < j.n_agents; ++a)
      if (a != j.agent_idx && j.path_len[a] > 0) {
        if (!j.path_xy[a]) return false;
        tpad = std::max(tpad, j.path_len[a]);
      }
    if (tpad > 0) {
      uint32_t npad = (static_cast<uint32_t>(j.n_agents) + 15u) & ~15u;
      uint32_t off = 0;
      uint16_t* tab = ps.alloc(static_cast<size_t>(tpad) * npad, off);
      if (!tab) return false;
      d.path_off = off;
      d.n_agents_pad = npad;
      d.t_pad = static_cast<uint32_t>(tpad);
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
            cell = inGrid(x, y) ? static_cast<uint16_t>(x | (y << 8)) : none;
          }
          tab[static_cast<size_t>(tt) * npad + a] = cell;
        }
      }
    }
  }
  return true;
}

void unpackResult(mrp_ll_ctx* ctx, const DevResult& d, const uint16_t* p, bool rejected, mrp_ll_result& r, bool sipp, int dimx,
                  int32_t init);
// The output of a root chain (ll_device.h kCtxChain) -> the caller's per-agent results; `count` = results the job may fill.
void unpackChain(mrp_ll_ctx* ctx, const DevResult& d, const uint16_t* out, bool rejected, mrp_ll_result& r, int32_t count,
                 uint32_t outWords) {
  r.status = rejected ? MRP_LL_BAD_JOB : d.status;
  // the root node's conflicts, when the chain planned every agent of the instance (mrp_ll.h): count and first one, else -1
  r.cost = (rejected || d.status != mrp::ST_OK) ? -1 : d.cost;
  r.fmin = (rejected || d.status != mrp::ST_OK) ? -1 : d.fmin;
  r.tier = 0;
  const int32_t done = (rejected || d.status != mrp::ST_OK) ? 0 : std::min<int32_t>(d.n_states, count);
  r.n_states = done;
  r.expanded = rejected ? 0 : d.expanded;
  if (!rejected)
    for (int q = 0; q < 8; ++q) ctx->stats.prof[q] += d.prof[q];
  if (!r.chain_results) return;
  const uint32_t* w = reinterpret_cast<const uint32_t*>(out);
  for (int32_t i = 0; i < count; ++i) {
    mrp_ll_result& ri = r.chain_results[i];
    if (i < done) {
      const uint32_t* e = w + static_cast<size_t>(i) * mrp::kChainEntryWords;
      DevResult f;
      std::memset(&f, 0, sizeof(f));
      f.status = static_cast<int32_t>(e[0]);
      f.cost = static_cast<int32_t>(e[1]);
      f.fmin = static_cast<int32_t>(e[2]);
      f.n_states = static_cast<int32_t>(e[3]);
      f.expanded = e[4];
      // (the path offset is a word the device wrote: never read past the job's host area)
      const uint32_t pathWords = (static_cast<uint32_t>(std::max(f.n_states, 0)) + 1u) / 2u;
      if (e[5] > outWords || pathWords > outWords - e[5]) {
        f.status = mrp::ST_BAD;
        f.n_states = 0;
      }
      unpackResult(ctx, f, reinterpret_cast<const uint16_t*>(w + (f.status == mrp::ST_BAD ? 0u : e[5])), false, ri, false, 0, 0);
    } else {
      ri.status = MRP_LL_NOT_RUN;
      ri.cost = ri.fmin = ri.n_states = 0;
      ri.expanded = 0;
      ri.tier = 0;
    }
  }
}

// Fills the launch parameters that do not depend on where the jobs live; returns the dynamic LDS size.
// kind: the kernel family the launch uses (0 mixed, 1 A*-epsilon only, 2 A* only): the A*-epsilon-only kernels keep the
// (time, cell) bitmap of the compact tier in the arena slot and take a smaller LDS window.
int fillCommonParams(mrp_ll_ctx* ctx, Ticket& t, mrp::LaunchParams& P, uint32_t& ldsBytesOut, int kind) {
  P.maps = ctx->mapsDev;
  P.queue_head = t.queueHead;
  P.arena = t.arena;
  P.arena_stride = ctx->arenaStride;
  P.arena_scratch_off = ctx->arenaScratchOff;
  P.arena_paths_bytes = ctx->arenaPathsBytes;
  P.out_stride = static_cast<uint32_t>(ctx->opt.max_horizon);
  P.out_host_stride = P.out_stride;  // (sessions: the ring's stride, sessionBegin)
  P.arena_nodes = static_cast<uint32_t>(ctx->opt.arena_nodes);
  P.arena_rows = static_cast<uint32_t>(ctx->opt.max_horizon);
  P.arena_row_words = ctx->arenaRowWords;
  // LDS of a workgroup: the compact tier's window (fixed size, ll_compact.h) + the focal path table; occupancy is
  // floor(160 KiB / ldsBytes) workgroups per CU
  uint32_t ldsNodes = static_cast<uint32_t>(ctx->opt.lds_nodes);
  uint32_t rowWords = (ctx->maxWpr + 3u) & ~3u;
  uint32_t rows = 0;
  uint32_t ldsBytes = 0;
  uint32_t ldsPaths = ctx->tierPathBytes;
  if (const char* e = std::getenv("MRP_LL_LDS_PATHS")) ldsPaths = static_cast<uint32_t>(std::max(0, std::atoi(e))) & ~31u;  // tuning knob
  if (ldsNodes) {
    rows = 64;
    ldsBytes = mrp_ll_lds_bytes(kind, ldsNodes, rows, rowWords, ldsPaths);
    if (ldsBytes > 160u * 1024u - 512u) {
      ldsNodes = 0;
      rows = 0;
    }
  }
  if (!ldsNodes) ldsBytes = mrp_ll_lds_bytes(kind, 0, 0, 0, 0);  // the control block alone
  P.path_store = ctx->pathStore;
  P.path_store_stride = ctx->pathStoreStride;
  P.path_store_slots = ctx->pathStore ? ctx->pathStoreSlots : 0;
  P.lds_nodes = ldsNodes;
  P.lds_rows = rows;
  P.lds_row_words = rowWords;
  P.lds_paths_bytes = ldsNodes ? ldsPaths : 0;
  if (kDebug) {
    if (!ctx->debugHost) {
      HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&ctx->debugHost), 16 * 4 * 4096,
                                hipHostMallocMapped | hipHostMallocCoherent));
    }
    std::memset(ctx->debugHost, 0, 16 * 4 * 4096);
    void* dptr = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer(&dptr, ctx->debugHost, 0));
    P.debug = static_cast<volatile uint32_t*>(dptr);
  }
  ldsBytesOut = ldsBytes;
  return MRP_LL_SUCCESS;
}

// Device result -> caller's mrp_ll_result (+ statistics).
void unpackResult(mrp_ll_ctx* ctx, const DevResult& d, const uint16_t* p, bool rejected, mrp_ll_result& r,
                  bool sipp, int dimx, int32_t init) {
  if (rejected) {
    r.status = MRP_LL_BAD_JOB;
    r.cost = r.fmin = r.n_states = 0;
    r.expanded = 0;
    r.tier = 0;
    return;
  }
  r.status = d.status;
  r.cost = d.cost;
  r.fmin = d.fmin;
  if (init & 0x40000000) {
    // not an initial cost: the goal of an MRP_LL_ASTAR_TA job (used for the action costs below)
  } else if (d.status == mrp::ST_OK && init != 0) {
    if (sipp) {
      r.cost = d.cost - init;  // sipp.hpp:103; fmin stays the A* f value (absolute)
    } else {                   // a_star.hpp:64,78: every node but the start carries initialCost in g and f
      r.cost = d.cost + init;
      if (d.n_states > 1) r.fmin = d.fmin + init;
    }
  }
  r.n_states = d.status == mrp::ST_OK ? d.n_states : 0;
  r.expanded = d.expanded;
  r.tier = static_cast<int32_t>(d.tier & 0xFFu);
  ctx->stats.jobs += 1;
  ctx->stats.expansions += d.expanded;
  ctx->stats.nodes_created += d.nodes_created;
  ctx->stats.migrated += (d.tier & 0xFFu) ? 1 : 0;
  for (int q = 0; q < 8; ++q) ctx->stats.prof[q] += d.prof[q];
  if (d.status == mrp::ST_OK && sipp) {
    // raw A* states (cell | g << 16) -> PlanResult with explicit Wait actions (sipp.hpp:105-128)
    const uint32_t* raw = reinterpret_cast<const uint32_t*>(p);
    const int nRaw = d.n_states;
    int out = 0;
    bool trunc = false;
    auto emit = [&](int cell, int t, int action, int cost, bool hasAction) {
      if (out < r.states_cap) {
        if (r.states_txy) {
          r.states_txy[3 * out] = t;
          r.states_txy[3 * out + 1] = cell % dimx;
          r.states_txy[3 * out + 2] = cell / dimx;
        }
        if (hasAction) {
          if (r.actions) r.actions[out] = action;
          if (r.action_costs) r.action_costs[out] = cost;
        }
      } else {
        trunc = true;
      }
      out += 1;
    };
    for (int k = 0; k + 1 < nRaw; ++k) {
      const int c0 = raw[k] & 0xFFFF, g0 = raw[k] >> 16, c1 = raw[k + 1] & 0xFFFF, g1 = raw[k + 1] >> 16;
      const int motion = actionFromDelta(c1 % dimx - c0 % dimx, c1 / dimx - c0 / dimx);
      const int waitTime = (g1 - g0) - 1;
      if (waitTime == 0) {
        emit(c0, g0, motion, g1 - g0, true);
      } else {
        emit(c0, g0, MRP_LL_ACT_WAIT, waitTime, true);
        emit(c0, g0 + waitTime, motion, 1, true);
      }
    }
    if (nRaw > 0) emit(raw[nRaw - 1] & 0xFFFF, raw[nRaw - 1] >> 16, 0, 0, false);
    r.n_states = out;
    if (trunc && (r.states_txy || r.actions)) r.status = MRP_LL_PATH_TRUNCATED;
    return;
  }
  if (d.status == mrp::ST_OK) {
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
    if (r.action_costs) {
      // MRP_LL_ASTAR_TA (init carries goal and flags, see taInfoOf): a Wait at the goal is free (cbs_ta.cpp:333-338)
      const bool ta = (init & 0x40000000) != 0, noGoal = (init & 0x10000) != 0;
      const int tgx = init & 0xFF, tgy = (init >> 8) & 0xFF;
      for (int k = 0; k + 1 < n && k < r.states_cap; ++k) {
        const bool wait = p[k] == p[k + 1];
        const bool atGoal = noGoal || ((p[k] & 0xFF) == tgx && (p[k] >> 8) == tgy);
        r.action_costs[k] = (ta && wait && atGoal) ? 0 : 1;
      }
    }
    if ((r.states_txy || r.actions) && r.states_cap < n) r.status = MRP_LL_PATH_TRUNCATED;
  }
}

// What unpackResult needs to know about a job besides its device result: initial_cost (A*) / start_time (SIPP), or — for
// MRP_LL_ASTAR_TA, whose initial cost is always 0 — bit 30 + the goal cell and the no-task flag.
int32_t jobInitOf(const mrp_ll_job& j, bool ok) {
  if (!ok) return 0;
  if (j.algo == MRP_LL_ASTAR_TA)
    return 0x40000000 | ((j.flags & MRP_LL_JOB_NO_GOAL) ? 0x10000 : ((j.goal_y & 0xFF) << 8 | (j.goal_x & 0xFF)));
  return j.initial_cost;
}

void trivialRejectedJob(mrp_ll_ctx* ctx, DevJob& d) {
  std::memset(&d, 0, sizeof(d));
  d.dimx = 1; d.dimy = 1; d.words_per_row = 1;
  d.map_word_off = ctx->maps.empty() ? 0 : ctx->maps[0].wordOff;
  d.gx = 0; d.gy = 0; d.algo = 0; d.last_goal_constraint = -1;
  d.max_expansions = 0;
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
  if (o.lds_nodes == 0) o.lds_nodes = mrp::kLdsMaxNodes;
  if (o.lds_nodes < 0) o.lds_nodes = 0;
  o.lds_nodes = std::min<int32_t>(o.lds_nodes, mrp::kLdsMaxNodes);  // the compact tier's open list holds lds_nodes / 2 entries
  o.lds_nodes &= ~3;

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
  if (ctx->ring.active) (void)mrp_ll_session_end(ctx);
  if (ctx->ring.ev0) (void)hipEventDestroy(ctx->ring.ev0);
  if (ctx->ring.ev1) (void)hipEventDestroy(ctx->ring.ev1);
  if (ctx->ring.ev2) (void)hipEventDestroy(ctx->ring.ev2);
  if (ctx->ring.stream2) (void)hipStreamDestroy(ctx->ring.stream2);
  if (ctx->ring.block) (void)hipHostFree(ctx->ring.block);
  if (ctx->ring.push) (void)(ctx->ring.pushInDevice ? hipFree(ctx->ring.push) : hipHostFree(ctx->ring.push));
  if (ctx->ring.sippCons) (void)hipHostFree(ctx->ring.sippCons);
  if (ctx->ring.compCountDev) (void)hipFree(ctx->ring.compCountDev);
  if (ctx->ring.heavyQ) (void)hipFree(ctx->ring.heavyQ);
  if (ctx->ring.heavyCtr) (void)hipFree(ctx->ring.heavyCtr);
  if (ctx->ring.heavyAlive) (void)hipHostFree(ctx->ring.heavyAlive);
  if (ctx->ring.ticksDev) (void)hipFree(ctx->ring.ticksDev);
  delete static_cast<SippScratch*>(ctx->sippScratch);
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
  if (ctx->scanDev) (void)hipFree(ctx->scanDev);
  if (ctx->scanStream) (void)hipStreamDestroy(ctx->scanStream);
  if (ctx->pathStore) (void)hipFree(ctx->pathStore);
  for (uint8_t* c : ctx->sippTabChunks) (void)hipFree(c);
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
  // Every bitmap starts on its own 128-byte line: a map uploaded while a resident kernel runs (below) must not share a
  // cache line with an older one — the kernels read obstacle words with plain cached loads, and an XCD's L2 may still
  // hold the line's previous contents (nothing invalidates it between jobs, ll_kernel.hip residentLoop).
  while (ctx->mapWords.size() & 31u) ctx->mapWords.push_back(0);
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
  if (ctx->ring.active) {
    // a resident kernel is reading the maps buffer: it may be appended to, but neither moved nor re-laid-out
    // (a map wider than the session's LDS rows is served from the HBM tier until the next session)
    if (ctx->mapWords.size() > ctx->mapsDevCap) {
      ctx->maps.pop_back();
      ctx->mapWords.resize(m.wordOff);
      ctx->err = "mrp_ll_upload_map: no room in the device map buffer during a session (upload maps before "
                 "mrp_ll_session_begin, or end the session first)";
      return MRP_LL_E_BUSY;
    }
    HIPCHK(ctx, hipMemcpy(ctx->mapsDev + m.wordOff, ctx->mapWords.data() + m.wordOff, m.wpr * sizeof(uint32_t),
                          hipMemcpyHostToDevice));
  } else {
    ctx->mapsDirty = true;
  }
  *mapId = static_cast<int32_t>(ctx->maps.size()) - 1;
  return MRP_LL_SUCCESS;
}


int mrp_ll_upload_heuristic(mrp_ll_ctx* ctx, int32_t mapId, const int32_t* dist, int32_t* heurId) {
  if (!ctx || !dist || !heurId || mapId < 0 || mapId >= static_cast<int32_t>(ctx->maps.size())) return MRP_LL_E_INVALID;
  if (ctx->ring.active) return MRP_LL_E_BUSY;  // (the maps buffer may have to grow)
  const MapRec& mp = ctx->maps[mapId];
  while (ctx->mapWords.size() & 31u) ctx->mapWords.push_back(0);  // own 128-byte lines, as the bitmaps
  HeurRec h;
  h.mapId = mapId;
  h.wordOff = static_cast<uint32_t>(ctx->mapWords.size());
  // halfwords, 0xFFFF = unreachable (the reference's table holds INT_MAX there): [y * 32 + x] for maps up to 32 x 32 (what
  // the compact tier copies into its window), [y * dimx + x] beyond (arena tier only)
  const bool small = mp.dimx <= 32 && mp.dimy <= 32;
  const int stride = small ? 32 : mp.dimx;
  const size_t words = small ? mrp::kHeurWords : (static_cast<size_t>(mp.dimx) * mp.dimy + 1) / 2;
  ctx->mapWords.resize(ctx->mapWords.size() + words, 0xFFFFFFFFu);
  uint16_t* t16 = reinterpret_cast<uint16_t*>(ctx->mapWords.data() + h.wordOff);
  for (int y = 0; y < mp.dimy; ++y)
    for (int x = 0; x < mp.dimx; ++x) {
      const int32_t v = dist[y * mp.dimx + x];
      t16[y * stride + x] = (v < 0 || v > 0xFFFE) ? 0xFFFFu : static_cast<uint16_t>(v);
    }
  ctx->heurs.push_back(h);
  ctx->mapsDirty = true;
  *heurId = static_cast<int32_t>(ctx->heurs.size()) - 1;
  return MRP_LL_SUCCESS;
}

int mrp_ll_configure_tiers(mrp_ll_ctx* ctx, int32_t ldsNodes, int32_t ldsRows, int32_t ldsPathBytes, int32_t* occOut) {
  if (!ctx) return MRP_LL_E_INVALID;
  if (ctx->ring.active) return MRP_LL_E_BUSY;
  for (const Ticket& t : ctx->tickets)
    if (t.inFlight) return MRP_LL_E_BUSY;
  if (ldsNodes < 0) ctx->opt.lds_nodes = 0;
  // the compact LDS tier holds up to 1023 open entries (lds_nodes / 2) and 64 time steps; its LDS window has a fixed size
  if (ldsNodes > 0) ctx->opt.lds_nodes = std::max(8, std::min<int32_t>(ldsNodes, mrp::kLdsMaxNodes) & ~3);
  if (ldsRows > 0) ctx->tierRows = static_cast<uint32_t>(std::min(std::max(ldsRows, 8), 64));
  if (ldsPathBytes > 0) ctx->tierPathBytes = static_cast<uint32_t>(std::min(ldsPathBytes, 65536)) & ~31u;
  if (occOut) {
    const uint32_t rowWords = (ctx->maxWpr + 3u) & ~3u;
    const uint32_t bytes = ctx->opt.lds_nodes
                               ? mrp_ll_lds_bytes(0, static_cast<uint32_t>(ctx->opt.lds_nodes), ctx->tierRows, rowWords, ctx->tierPathBytes) + 256
                               : 0;
    *occOut = bytes ? static_cast<int32_t>(std::max<uint32_t>(1, std::min<uint32_t>(16, (160u * 1024u) / bytes))) : 16;
  }
  return MRP_LL_SUCCESS;
}

int mrp_ll_session_occupancy(mrp_ll_ctx* ctx, int32_t algo, int32_t* occOut) {
  if (!ctx || !occOut) return MRP_LL_E_INVALID;
  if (algo != MRP_LL_ASTAR && algo != MRP_LL_ASTAR_EPS && algo != MRP_LL_ASTAR_TA && algo != MRP_LL_SIPP) return MRP_LL_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (algo == MRP_LL_SIPP) {  // the resident SIPP kernel: a fixed LDS tier (ll_kernel.hip MRP_LL_SIPP_LDS_NODES)
    const int occS = mrp_ll_sipp_persistent_occupancy();
    *occOut = occS > 0 ? std::min(occS, 32) : 6;
    return MRP_LL_SUCCESS;
  }
  const int kind = algo == MRP_LL_ASTAR_EPS ? 1 : 2;
  const uint32_t rowWords = (ctx->maxWpr + 3u) & ~3u;
  const uint32_t bytes = mrp_ll_lds_bytes(kind, static_cast<uint32_t>(ctx->opt.lds_nodes), ctx->tierRows, rowWords,
                                          ctx->opt.lds_nodes ? ctx->tierPathBytes : 0);
  int occ = mrp_ll_persistent_occupancy(kind, bytes);  // what the runtime grants this kernel with this much dynamic LDS
  if (occ <= 0) occ = static_cast<int>(std::max<uint32_t>(1, std::min<uint32_t>(16, (160u * 1024u) / (bytes + 256u))));
  *occOut = std::min(occ, 16);
  return MRP_LL_SUCCESS;
}

// ---- session mode ---------------------------------------------------------------------------------------------
// kind (A* sessions): 0 = jobs of both A* algorithms, 1 = MRP_LL_ASTAR_EPS only, 2 = MRP_LL_ASTAR only
static int sessionBegin(mrp_ll_ctx* ctx, int32_t workgroups, bool sipp, int kind = 0, int32_t heavyWgs = 0,
                        int32_t* gate = nullptr, int32_t parties = 0) {
  if (!ctx) return MRP_LL_E_INVALID;
  Ring& g = ctx->ring;
  if (g.active) return MRP_LL_E_INVALID;
  if (heavyWgs > 0 && (sipp || kind != 1)) return MRP_LL_E_INVALID;
  if (const char* e = std::getenv("MRP_LL_HEAVY_WGS")) {  // tuning / A-B knob: overrides the caller's heavy workgroups (kind 1)
    if (!sipp && kind == 1 && heavyWgs >= 0) heavyWgs = std::max(0, std::atoi(e));
  }
  if (heavyWgs < 0) heavyWgs = 0;  // (the fallback below: a single all-tier launch, whatever the knob says)
  if (heavyWgs > 0 && (ctx->opt.lds_nodes == 0 || heavyWgs >= ctx->opt.slots)) heavyWgs = 0;  // (no compact tier: one launch serves all)
  HIPCHK(ctx, hipSetDevice(ctx->device));
  Ticket& t = ctx->tickets[0];
  if (t.inFlight) {
    ctx->err = "mrp_ll_session_begin: a batch is still in flight";
    return MRP_LL_E_BUSY;
  }
  int rc = syncMaps(ctx);
  if (rc != MRP_LL_SUCCESS) return rc;
  const uint32_t R = Ring::kSlots;
  // halfwords per job slot of the host output area: a path, or the output of a root chain (ll_device.h kCtxChain)
  // (a chain of 128 agents: 16 halfwords of header + up to 64 states each)
  g.outStride = std::max<uint32_t>(static_cast<uint32_t>(ctx->opt.max_horizon), mrp::kChainMaxAgents * 80u);
  if (!g.block) {
    size_t off = 0;
    auto take = [&](size_t bytes) {
      size_t o = off;
      off += (bytes + 255) & ~size_t(255);
      return o;
    };
    // device -> host
    size_t oDone = take(R * 4), oComp = take(R * 4), oRes = take(R * sizeof(DevResult)),
           oOut = take(static_cast<size_t>(R) * g.outStride * 2);
    const size_t hostBytes = off;
    // host -> device
    off = 0;
    size_t oState = take(Ring::kTickets * 4), oStop = take(256), oHead = take(256), oJobs = take(R * sizeof(DevJob)),
           oCons = take(static_cast<size_t>(R) * Ring::kSlotConsWords * 4),
           oPaths = take(static_cast<size_t>(R) * Ring::kSlotPathHalfs * 2);
    const size_t pushBytes = off;
    g.pushBytes = pushBytes;
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&g.block), hostBytes, hipHostMallocMapped | hipHostMallocCoherent));
    std::memset(g.block, 0, hostBytes);
    {
      hipDeviceProp_t prop;
      std::memset(&prop, 0, sizeof(prop));
      const bool largeBar = hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.isLargeBar != 0;
      // opt-in (MRP_LL_RING_IN_DEVICE=1): with the cache-wide fences gone from the resident loop the two placements
      // measure within 1 % of each other, and pinned host memory does not depend on the BAR
      const char* e = std::getenv("MRP_LL_RING_IN_DEVICE");
      g.pushInDevice = largeBar && e && *e == '1';
    }
    if (g.pushInDevice) {
      void* p = nullptr;
      if (hipExtMallocWithFlags(&p, pushBytes, hipDeviceMallocUncached) == hipSuccess) {
        g.push = static_cast<uint8_t*>(p);
        std::memset(g.push, 0, pushBytes);  // (through the BAR, once per engine; no device-wide synchronisation — other
        __builtin_ia32_sfence();            //  engines' resident kernels may be running)
      } else {
        (void)hipGetLastError();
        g.pushInDevice = false;
      }
    }
    if (!g.pushInDevice) {
      HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&g.push), pushBytes, hipHostMallocMapped | hipHostMallocCoherent));
      std::memset(g.push, 0, pushBytes);
    }
    g.state = reinterpret_cast<uint32_t*>(g.push + oState);
    g.done = reinterpret_cast<uint32_t*>(g.block + oDone);
    g.stop = reinterpret_cast<uint32_t*>(g.push + oStop);
    g.headWord = reinterpret_cast<uint32_t*>(g.push + oHead);
    g.compRing = reinterpret_cast<uint32_t*>(g.block + oComp);
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&g.compCountDev), 256));
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&g.ticksDev), 256));
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&g.heavyQ), static_cast<size_t>(R) * 8));
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&g.heavyCtr), 256));
    HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&g.heavyAlive), static_cast<size_t>(R) * 4,
                              hipHostMallocMapped | hipHostMallocCoherent));
    g.jobs = reinterpret_cast<DevJob*>(g.push + oJobs);
    g.results = reinterpret_cast<DevResult*>(g.block + oRes);
    g.outPaths = reinterpret_cast<uint16_t*>(g.block + oOut);
    g.cons = reinterpret_cast<uint32_t*>(g.push + oCons);
    g.paths = reinterpret_cast<uint16_t*>(g.push + oPaths);
    HIPCHK(ctx, hipEventCreate(&g.ev0));
    HIPCHK(ctx, hipEventCreate(&g.ev1));
  }
  std::memset(g.state, 0, Ring::kTickets * 4);  // (write-combined stores through the BAR when the block is device memory)
  std::memset(g.done, 0, R * 4);
  std::memset(g.compRing, 0, R * 4);
  g.compCursor = 0;
  __atomic_store_n(g.stop, 0u, __ATOMIC_RELEASE);
  __atomic_store_n(g.headWord, 0u, __ATOMIC_RELEASE);
  __atomic_store_n(g.headWord + 16, 0u, __ATOMIC_RELEASE);
  pushFence(g);
  g.head[0] = g.head[1] = 0;
  g.busy.assign(R, 0);
  g.slotTicket.assign(R, -1);
  g.slotJob.assign(R, 0);
  g.slotGen.assign(R, 0);
  g.tkSlot.assign(Ring::kTickets, 0xFFFFFFFFu);
  g.tkSeq.assign(Ring::kTickets, 0);
  g.freeSlots.clear();
  g.sipp = sipp;
  for (uint32_t sl = sipp ? Ring::kSippSlots : Ring::kSlots; sl-- > 0;) g.freeSlots.push_back(sl);
  if (sipp) {
    // table of a job: cellIdx[cells] + specFirst[K + 1] + 2 words per safe interval; 8 words per cell cover about three
    // intervals on every cell
    const uint32_t want = std::max<uint32_t>(8192u, ctx->maxWpr * 32u * 8u);
    if (want > g.sippSlotWords) {
      if (g.sippCons) (void)hipHostFree(g.sippCons);
      g.sippCons = nullptr;
      HIPCHK(ctx, hipHostMalloc(reinterpret_cast<void**>(&g.sippCons), static_cast<size_t>(Ring::kSippSlots) * want * 4,
                                hipHostMallocMapped | hipHostMallocCoherent));
      g.sippSlotWords = want;
    }
    g.slotDimx.assign(R, 0);
    g.slotTable.assign(R, nullptr);
    g.slotSippFlags.assign(R, 0);
  }
  g.slotInit.assign(R, 0);
  g.slotChain.assign(R, 0);
  ctx->sess.clear();
  ctx->sessFree.clear();
  for (int q = 0; q < mrp_ll_ctx::kMaxTags; ++q) {
    ctx->coStash[q].clear();
    ctx->coStashCount[q].store(0, std::memory_order_relaxed);
  }
  auto devPtr = [&](void* hostPtr) {
    const uint8_t* b = static_cast<const uint8_t*>(hostPtr);
    if (g.pushInDevice && b >= g.push && b < g.push + g.pushBytes)
      return hostPtr;  // device memory the host writes through the BAR: one address for both sides
    void* d = nullptr;
    (void)hipHostGetDevicePointer(&d, hostPtr, 0);
    return d;
  };
  mrp::LaunchParams P;
  std::memset(&P, 0, sizeof(P));
  P.jobs = static_cast<const DevJob*>(devPtr(g.jobs));
  P.results = static_cast<DevResult*>(devPtr(g.results));
  P.out_paths = static_cast<uint16_t*>(devPtr(g.outPaths));
  P.cons = static_cast<const uint32_t*>(devPtr(sipp ? g.sippCons : g.cons));
  P.paths = static_cast<const uint16_t*>(devPtr(g.paths));
  P.ring_state = static_cast<uint32_t*>(devPtr(g.state));
  P.ring_done = static_cast<uint32_t*>(devPtr(g.done));
  P.ring_stop = static_cast<uint32_t*>(devPtr(g.stop));
  P.ring_head = static_cast<uint32_t*>(devPtr(g.headWord));
  P.comp_ring = static_cast<uint32_t*>(devPtr(g.compRing));
  P.comp_count = g.compCountDev;
  P.sess_ticks = g.ticksDev;
  g.Q[0] = Ring::kTickets0;
  g.Q[1] = Ring::kTickets1;
  if (const char* e = std::getenv("MRP_LL_TICKET_RING")) {  // test knob: small rings wrap often
    const uint32_t q = static_cast<uint32_t>(std::atoi(e));
    g.Q[0] = std::min(Ring::kTickets0, std::max(q, Ring::kSlots));  // a batch never laps its own entries
    g.Q[1] = std::min(Ring::kTickets1, std::max(q, Ring::kSlots));
  }
  P.ring_size = g.Q[0];
  P.ring_size1 = g.Q[1];
  P.n_slots = Ring::kSlots;
  g.idleLimitS = 20;
  if (const char* e = std::getenv("MRP_LL_IDLE_LIMIT_S")) g.idleLimitS = static_cast<uint32_t>(std::max(1, std::atoi(e)));  // test knob
  P.ring_idle_limit_s = g.idleLimitS;
  g.heartbeat = 0;
  g.emptyPolls = 0;
  g.inFlightJobs = 0;
  __atomic_store_n(g.headWord + mrp::kHeartbeatWord, 0u, __ATOMIC_RELEASE);
  pushFence(g);
  g.lastBeatTsc = __builtin_ia32_rdtsc();
  uint32_t ldsBytes = 0;
  rc = fillCommonParams(ctx, t, P, ldsBytes, sipp ? 0 : kind);
  if (rc != MRP_LL_SUCCESS) return rc;
  P.out_host_stride = g.outStride;
  ctx->sessionRowWords = P.lds_row_words;
  ctx->sessionLdsPathBytes = P.lds_paths_bytes;
  if (P.lds_nodes == 0) heavyWgs = 0;
  heavyWgs = std::min<int32_t>(heavyWgs, static_cast<int32_t>(Ring::kSlots) - 1);  // (the last alive word is the front launch's)
  g.grid = static_cast<uint32_t>(workgroups > 0 ? std::min(workgroups, ctx->opt.slots - heavyWgs) : ctx->opt.slots - heavyWgs);
  HIPCHK(ctx, hipMemsetAsync(t.queueHead, 0, 256, t.stream));  // session tickets of both lanes count from 0
  HIPCHK(ctx, hipMemsetAsync(g.compCountDev, 0, 4, t.stream));
  HIPCHK(ctx, hipMemsetAsync(g.ticksDev, 0, 64, t.stream));
  t.queueBase = 0;
  g.heavy = false;
  g.grid2 = 0;
  if (heavyWgs > 0) {
    // The heavy workgroups go first, on their own stream, so that they find room on the device before the many front
    // workgroups fill it; each reports itself in heavyAlive.  Without at least one of them a search that outgrows the
    // front tier would never run: in that case this session is ended and begun again as a single all-tier launch.
    HIPCHK(ctx, hipMemsetAsync(g.heavyQ, 0, static_cast<size_t>(R) * 8, t.stream));
    HIPCHK(ctx, hipMemsetAsync(g.heavyCtr, 0, 256, t.stream));
    std::memset(g.heavyAlive, 0, static_cast<size_t>(R) * 4);
    HIPCHK(ctx, hipEventRecord(g.ev0, t.stream));
    if (!g.stream2) {
      HIPCHK(ctx, hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
      HIPCHK(ctx, hipEventCreate(&g.ev2));
    }
    void* aliveDev = nullptr;
    HIPCHK(ctx, hipHostGetDevicePointer(&aliveDev, g.heavyAlive, 0));
    P.heavy_q = g.heavyQ;
    P.heavy_ctr = g.heavyCtr;
    P.heavy_alive = static_cast<uint32_t*>(aliveDev);
    P.heavy_wgs = static_cast<uint32_t>(heavyWgs);
    mrp::LaunchParams P2 = P;
    P2.lds_paths_bytes = 0;  // the wide window holds no path table
    P2.arena = t.arena + static_cast<uint64_t>(g.grid) * ctx->arenaStride;  // arena slots behind the front workgroups'
    HIPCHK(ctx, hipStreamWaitEvent(g.stream2, g.ev0, 0));
    HIPCHK(ctx, mrp_ll_launch_front_heavy(&P2, static_cast<uint32_t>(heavyWgs), 0, 1, g.stream2));
    HIPCHK(ctx, hipEventRecord(g.ev2, g.stream2));
    // residency: every heavy workgroup reports itself within microseconds when there is room for it; 200 ms is "never"
    const auto tw0 = std::chrono::steady_clock::now();
    int32_t nAlive = 0;
    for (;;) {
      nAlive = 0;
      for (int32_t q = 0; q < heavyWgs; ++q) nAlive += __atomic_load_n(g.heavyAlive + q, __ATOMIC_ACQUIRE) != 0 ? 1 : 0;
      if (nAlive == heavyWgs || std::chrono::duration<double>(std::chrono::steady_clock::now() - tw0).count() > 0.2) break;
    }
    if (gate) {  // the other contexts' heavy workgroups first, then anybody's front workgroups (mrp_ll.h)
      __atomic_fetch_add(gate, 1, __ATOMIC_ACQ_REL);
      const auto tg0 = std::chrono::steady_clock::now();
      while (__atomic_load_n(gate, __ATOMIC_ACQUIRE) < parties &&
             std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count() < 2.0) {
      }
    }
    HIPCHK(ctx, mrp_ll_launch_front_heavy(&P, g.grid, ldsBytes, 0, t.stream));
    HIPCHK(ctx, hipEventRecord(g.ev1, t.stream));
    ctx->stats.launches += 2;
    g.kind = 1;
    g.grid2 = static_cast<uint32_t>(heavyWgs);
    g.heavy = true;
    g.active = true;
    if (nAlive > 0) {
      // ... and the front launch must really run beside them: two streams of this process that were given the same
      // hardware queue (GPU_MAX_HW_QUEUES smaller than the number of resident kernels) would wait for each other for ever
      const auto tf0 = std::chrono::steady_clock::now();
      while (__atomic_load_n(g.heavyAlive + (R - 1), __ATOMIC_ACQUIRE) == 0) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - tf0).count() > 5.0) {
          (void)mrp_ll_session_end(ctx);
          ctx->err = "mrp_ll_session_begin_tiers: the front workgroups did not start beside the heavy ones (HIP streams share "
                     "hardware queues: export GPU_MAX_HW_QUEUES >= 2 per context + 4 before the HIP runtime initialises)";
          return MRP_LL_E_DEVICE;
        }
      }
      return MRP_LL_SUCCESS;
    }
    // no heavy workgroup runs: a search that outgrows the front tier would wait for ever — one all-tier launch instead
    int rcEnd = mrp_ll_session_end(ctx);
    if (rcEnd != MRP_LL_SUCCESS) return rcEnd;
    ctx->stats.heavy_fallbacks += 1;
    return sessionBegin(ctx, workgroups, false, kind, -1);
  }
  if (gate) __atomic_fetch_add(gate, 1, __ATOMIC_ACQ_REL);  // (a party without heavy workgroups holds nobody up)
  HIPCHK(ctx, hipEventRecord(g.ev0, t.stream));
  if (sipp)
    HIPCHK(ctx, mrp_ll_launch_sipp_persistent(&P, g.grid, t.stream));
  else
    HIPCHK(ctx, mrp_ll_launch_persistent(&P, g.grid, ldsBytes, kind, t.stream));
  g.kind = sipp ? 0 : kind;
  if (!sipp && P.lds_nodes != 0) {
    uint32_t extra = ctx->extraHbmWgs;
    if (const char* e = std::getenv("MRP_LL_EXTRA_HBM_WGS")) extra = static_cast<uint32_t>(std::max(0, std::atoi(e)));
    extra = std::min<uint32_t>(extra, static_cast<uint32_t>(ctx->opt.slots) - g.grid);
    if (extra) {
      if (!g.stream2) {
        HIPCHK(ctx, hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
        HIPCHK(ctx, hipEventCreate(&g.ev2));
      }
      mrp::LaunchParams P2 = P;
      P2.lds_nodes = 0;
      P2.lds_rows = 0;
      P2.lds_paths_bytes = 0;
      P2.arena = t.arena + static_cast<uint64_t>(g.grid) * ctx->arenaStride;  // arena slots behind the first launch's
      // the counters the workgroups take tickets from are zeroed on t.stream: order this launch behind that
      HIPCHK(ctx, hipStreamWaitEvent(g.stream2, g.ev0, 0));
      HIPCHK(ctx, mrp_ll_launch_persistent(&P2, extra, mrp_ll_lds_bytes(kind, 0, 0, 0, 0), kind, g.stream2));
      HIPCHK(ctx, hipEventRecord(g.ev2, g.stream2));
      g.grid2 = extra;
      ctx->stats.launches += 1;
    }
  }
  HIPCHK(ctx, hipEventRecord(g.ev1, t.stream));
  g.active = true;
  ctx->stats.launches += 1;
  return MRP_LL_SUCCESS;
}

int mrp_ll_session_begin(mrp_ll_ctx* ctx, int32_t workgroups) { return sessionBegin(ctx, workgroups, false); }
int mrp_ll_session_begin_algo(mrp_ll_ctx* ctx, int32_t algo, int32_t workgroups) {
  if (algo == MRP_LL_SIPP) return sessionBegin(ctx, workgroups, true);
  if (algo != MRP_LL_ASTAR && algo != MRP_LL_ASTAR_EPS) return MRP_LL_E_INVALID;
  return sessionBegin(ctx, workgroups, false, algo == MRP_LL_ASTAR_EPS ? 1 : 2);
}
int mrp_ll_session_begin_sipp(mrp_ll_ctx* ctx, int32_t workgroups) { return sessionBegin(ctx, workgroups, true); }
int mrp_ll_session_begin_tiers_gated(mrp_ll_ctx* ctx, int32_t algo, int32_t workgroups, int32_t heavyWorkgroups, int32_t* gate,
                                     int32_t parties) {
  if (algo == MRP_LL_SIPP) return MRP_LL_E_INVALID;
  if (algo != MRP_LL_ASTAR && algo != MRP_LL_ASTAR_EPS) return MRP_LL_E_INVALID;
  if (algo != MRP_LL_ASTAR_EPS && heavyWorkgroups > 0) return MRP_LL_E_INVALID;
  return sessionBegin(ctx, workgroups, false, algo == MRP_LL_ASTAR_EPS ? 1 : 2, std::max(heavyWorkgroups, 0), gate, parties);
}
int mrp_ll_session_begin_tiers(mrp_ll_ctx* ctx, int32_t algo, int32_t workgroups, int32_t heavyWorkgroups) {
  return mrp_ll_session_begin_tiers_gated(ctx, algo, workgroups, heavyWorkgroups, nullptr, 0);
}
int mrp_ll_session_tiers_geometry(mrp_ll_ctx* ctx, int32_t* frontOcc, int32_t* frontLds, int32_t* heavyLds) {
  if (!ctx) return MRP_LL_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const uint32_t rowWords = (ctx->maxWpr + 3u) & ~3u;
  const uint32_t bytes = mrp_ll_lds_bytes(1, static_cast<uint32_t>(ctx->opt.lds_nodes), ctx->tierRows, rowWords,
                                          ctx->opt.lds_nodes ? ctx->tierPathBytes : 0);
  if (frontOcc) {
    int occ = mrp_ll_front_heavy_occupancy(0, bytes);
    if (occ <= 0) occ = static_cast<int>(std::max<uint32_t>(1, std::min<uint32_t>(16, (160u * 1024u) / (bytes + 256u))));
    *frontOcc = std::min(occ, 16);
  }
  if (frontLds) *frontLds = static_cast<int32_t>(bytes);
  if (heavyLds) *heavyLds = static_cast<int32_t>(mrp_ll_heavy_lds_bytes());
  return MRP_LL_SUCCESS;
}

int mrp_ll_session_end(mrp_ll_ctx* ctx) {
  if (!ctx) return MRP_LL_E_INVALID;
  Ring& g = ctx->ring;
  if (!g.active) return MRP_LL_E_INVALID;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  __atomic_store_n(g.stop, 1u, __ATOMIC_RELEASE);
  pushFence(g);
  HIPCHK(ctx, hipEventSynchronize(g.ev1));
  if (g.grid2) HIPCHK(ctx, hipEventSynchronize(g.ev2));
  float ms = 0.f;
  if (hipEventElapsedTime(&ms, g.ev0, g.ev1) == hipSuccess) ctx->stats.kernel_ms += ms;
  // (the heavy launch is a launch of its own in stats.launches: its duration counts too — ev0 is where its stream started)
  if (g.grid2 && hipEventElapsedTime(&ms, g.ev0, g.ev2) == hipSuccess) ctx->stats.kernel_ms += ms;
  {
    unsigned long long tk[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (hipMemcpy(tk, g.ticksDev, 64, hipMemcpyDeviceToHost) == hipSuccess) {
      ctx->stats.session_active_wgs += static_cast<int64_t>(tk[2]);
      ctx->stats.session_busy_ms += static_cast<double>(tk[0]) / 1e5;  // 100 MHz ticks
      ctx->stats.session_idle_ms += static_cast<double>(tk[1]) / 1e5;
      ctx->stats.heavy_active_wgs += static_cast<int64_t>(tk[6]);
      ctx->stats.heavy_busy_ms += static_cast<double>(tk[4]) / 1e5;
      ctx->stats.heavy_idle_ms += static_cast<double>(tk[5]) / 1e5;
    }
  }
  g.active = false;
  // a table whose job was abandoned: its device copy is in an unknown state, the next job rewrites it whole
  for (mrp_ll_sipp_table*& tb : g.slotTable)
    if (tb) {
      tb->inFlight = false;
      tb->devFresh = true;  // (a sipp_commit job that was abandoned leaves no trace: its result never reached the caller)
      tb = nullptr;
    }
  // the device counter is past the published tickets: the next batch-mode launch starts from a clean base
  Ticket& t = ctx->tickets[0];
  HIPCHK(ctx, hipMemset(t.queueHead, 0, 256));
  t.queueBase = 0;
  return MRP_LL_SUCCESS;
}

// Every host call into a live session moves the heartbeat word the resident workgroups watch (ll_kernel.hip residentLoop).
// The resident workgroups look at the heartbeat once per idle limit (20 s): a store every 10 ms is plenty, and it is a
// posted PCIe write when the word lives in device memory.
static inline void sessionBeat(Ring& g) {
  const uint64_t tsc = __builtin_ia32_rdtsc();  // (~7 ns; 2^24 reference cycles are 5-8 ms on any current x86)
  if (tsc - g.lastBeatTsc < (1ull << 24)) return;
  g.lastBeatTsc = tsc;
  __atomic_store_n(g.headWord + mrp::kHeartbeatWord, ++g.heartbeat, __ATOMIC_RELAXED);
}

// The resident kernel must still be running while jobs are in flight; checked every few thousand empty polls.
static int sessionAlive(mrp_ll_ctx* ctx) {
  Ring& g = ctx->ring;
  if (g.inFlightJobs == 0 || (++g.emptyPolls & 0xFFF) != 0) return MRP_LL_SUCCESS;
  if (hipEventQuery(g.ev1) != hipErrorNotReady) {
    ctx->err = "session: the resident kernel has exited while jobs were in flight (idle limit or device fault)";
    return MRP_LL_E_DEVICE;
  }
  return MRP_LL_SUCCESS;
}

static int sessionSubmit(mrp_ll_ctx* ctx, int32_t lane, int32_t nJobs, const mrp_ll_job* jobs, mrp_ll_result* results,
                         int32_t* ticketOut) {
  Ring& g = ctx->ring;
  sessionBeat(g);
  const uint32_t nSlotsLane = g.sipp ? Ring::kSippSlots : (lane ? Ring::kSlots : Ring::kSlots - Ring::kReserve1);
  const uint32_t Q = g.Q[lane];
  const uint32_t qBase = lane ? g.Q[0] : 0;
  if (nJobs > static_cast<int32_t>(nSlotsLane)) {
    ctx->err = "mrp_ll_submit (session): batch larger than the ring";
    return MRP_LL_E_INVALID;
  }
  // consume finished tickets first; the bulk lane leaves kReserve1 slots to the priority lane
  if (g.freeSlots.size() < static_cast<size_t>(nJobs) + ((lane || g.sipp) ? 0u : Ring::kReserve1)) return MRP_LL_E_BUSY;
  for (int i = 0; i < nJobs; ++i) {
    // a ticket entry may be overwritten once the job published there a whole ring ago has been consumed
    const uint32_t qi = qBase + static_cast<uint32_t>((g.head[lane] + i) % Q);
    const uint32_t prev = g.tkSlot[qi];
    if (prev != 0xFFFFFFFFu && g.busy[prev] && g.slotGen[prev] == g.tkSeq[qi]) return MRP_LL_E_BUSY;
  }
  auto packT0 = std::chrono::steady_clock::now();
  int ti = -1;
  if (!ctx->sessFree.empty()) {
    ti = ctx->sessFree.back();
    ctx->sessFree.pop_back();
  } else {
    ctx->sess.emplace_back();
    ti = static_cast<int>(ctx->sess.size()) - 1;
  }
  SessTicket& st = ctx->sess[ti];
  st.used = true;
  st.lane = lane;
  st.n = nJobs;
  st.remaining = nJobs;
  st.res = results;
  st.state.assign(nJobs, 0);
  st.slots.resize(nJobs);
  st.seq.resize(nJobs);
  for (int i = 0; i < nJobs; ++i) {
    const uint64_t tk = g.head[lane] + i;
    const uint32_t qi = qBase + static_cast<uint32_t>(tk % Q);
    const uint32_t gen = (static_cast<uint32_t>(tk / Q) + 1) & 0x1FFFFFu;
    const uint32_t slot = g.freeSlots.back();
    g.freeSlots.pop_back();
    ConsSinkSlot cs{g.cons + static_cast<size_t>(slot) * Ring::kSlotConsWords, slot * Ring::kSlotConsWords,
                    Ring::kSlotConsWords};
    PathSinkSlot ps{g.paths + static_cast<size_t>(slot) * Ring::kSlotPathHalfs, slot * Ring::kSlotPathHalfs,
                    Ring::kSlotPathHalfs};
    DevJob d;
    bool ok;
    uint32_t sippWords = 0;
    if (g.sipp) {  // a SIPP session takes SIPP jobs only, an A* / A*-epsilon session none
      ConsSinkSlot csS{g.sippCons + static_cast<size_t>(slot) * g.sippSlotWords, slot * g.sippSlotWords, g.sippSlotWords};
      ok = jobs[i].algo == MRP_LL_SIPP && packJob(ctx, jobs[i], csS, ps, d);
      if (ok) g.slotDimx[slot] = ctx->maps[jobs[i].map_id].dimx;
      sippWords = csS.used;
      {
        const uint32_t fl = !ok || !jobs[i].sipp_table ? 0u
                            : ((d.ctx_flags & mrp::kSippResident) ? 1u : 0u) | (jobs[i].sipp_commit ? 2u : 0u);
        g.slotTable[slot] = fl ? const_cast<mrp_ll_sipp_table*>(jobs[i].sipp_table) : nullptr;
        g.slotSippFlags[slot] = static_cast<uint8_t>(fl);
      }
    } else {
      ok = jobs[i].algo != MRP_LL_SIPP &&
           (g.kind == 0 || (g.kind == 1 ? jobs[i].algo == MRP_LL_ASTAR_EPS
                                        : (jobs[i].algo == MRP_LL_ASTAR || jobs[i].algo == MRP_LL_ASTAR_TA))) &&
           packJob(ctx, jobs[i], cs, ps, d);
    }
    if (!ok) {  // wrong kind of job for this session, or constraint list / table larger than a ring slot
      trivialRejectedJob(ctx, d);
      st.state[i] = 2;
    }
    g.jobs[slot] = d;
    ctx->stats.staged_bytes += static_cast<int64_t>(sizeof(DevJob)) + 4 * static_cast<int64_t>(g.sipp ? sippWords : cs.used) +
                               (g.sipp || (d.ctx_flags & mrp::kCtxById) ? 0 : 2 * static_cast<int64_t>(d.t_pad) * d.n_agents_pad);
    g.slotInit[slot] = jobInitOf(jobs[i], ok);
    g.slotChain[slot] = ((jobs[i].flags & MRP_LL_JOB_ROOT_CHAIN) && !g.sipp) ? std::max(1, jobs[i].n_agents - jobs[i].agent_idx) : 0;
    g.busy[slot] = 1;
    g.slotTicket[slot] = ti;
    g.slotJob[slot] = i;
    g.slotGen[slot] = ((static_cast<uint32_t>(tk) + 1u) & 0x3FFFFFFFu) | 0x40000000u | (lane ? 0x80000000u : 0u);  // never 0
    __atomic_store_n(g.done + slot, 0u, __ATOMIC_RELAXED);  // the previous occupant's done word must not be mistaken
    g.tkSlot[qi] = slot;
    g.tkSeq[qi] = g.slotGen[slot];
    st.slots[i] = slot;
    st.seq[i] = g.slotGen[slot];
    pushFence(g);  // the job data above leaves before its ticket entry
    __atomic_store_n(g.state + qi, (gen << mrp::kRingSlotBits) | slot, __ATOMIC_RELEASE);  // publish: the job data above is visible first
  }
  pushFence(g);    // ... and the entries before the count that covers them
  g.head[lane] += static_cast<uint64_t>(nJobs);
  g.inFlightJobs += static_cast<uint32_t>(nJobs);
  // after every ticket entry
  __atomic_store_n(g.headWord + 16 * lane, static_cast<uint32_t>(g.head[lane]), __ATOMIC_RELEASE);
  pushFence(g);
  ctx->stats.pack_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - packT0).count();
  *ticketOut = ti;
  return MRP_LL_SUCCESS;
}

int mrp_ll_poll(mrp_ll_ctx* ctx, int32_t ticket, int32_t* doneOut) {
  if (!ctx || !doneOut) return MRP_LL_E_INVALID;
  Ring& g = ctx->ring;
  if (!g.active) {  // batch mode: completion of the launch
    if (ticket < 0 || ticket >= static_cast<int32_t>(ctx->tickets.size()) || !ctx->tickets[ticket].inFlight)
      return MRP_LL_E_INVALID;
    if (ctx->tickets[ticket].nJobs > 0 && hipEventQuery(ctx->tickets[ticket].evK1) == hipErrorNotReady) {
      *doneOut = 0;
      return MRP_LL_SUCCESS;
    }
    *doneOut = 1;
    return mrp_ll_wait(ctx, ticket);
  }
  if (ticket < 0 || ticket >= static_cast<int32_t>(ctx->sess.size()) || !ctx->sess[ticket].used) return MRP_LL_E_INVALID;
  SessTicket& st = ctx->sess[ticket];
  sessionBeat(g);
  for (int i = 0; i < st.n && st.remaining > 0; ++i) {
    if (st.state[i] == 1) continue;
    const uint32_t slot = st.slots[i];
    if (__atomic_load_n(g.done + slot, __ATOMIC_ACQUIRE) != st.seq[i]) continue;
    if (g.slotChain[slot])
      unpackChain(ctx, g.results[slot], g.outPaths + static_cast<size_t>(slot) * g.outStride, st.state[i] == 2, st.res[i],
                  g.slotChain[slot], g.outStride / 2u);
    else
      unpackResult(ctx, g.results[slot], g.outPaths + static_cast<size_t>(slot) * g.outStride, st.state[i] == 2,
                   st.res[i], g.sipp, g.sipp ? g.slotDimx[slot] : 0, g.slotInit[slot]);
    if (g.sipp && g.slotTable[slot]) {
      finishSippTableJob(g.slotTable[slot], g.slotSippFlags[slot], g.results[slot],
                         g.outPaths + static_cast<size_t>(slot) * g.outStride);
      g.slotTable[slot] = nullptr;
    }
    st.state[i] = 1;
    st.remaining -= 1;
    g.busy[slot] = 0;
    g.inFlightJobs -= 1;
    g.emptyPolls = 0;
    g.freeSlots.push_back(slot);
  }
  *doneOut = st.remaining == 0 ? 1 : 0;
  if (st.remaining == 0) {
    st.used = false;
    ctx->sessFree.push_back(ticket);
  }
  return MRP_LL_SUCCESS;
}

// Drains the completion queue: every finished job is unpacked into its caller's result; a ticket whose last job this was
// goes to `tickets` (tag < 0: all of them; else: those of this co-worker, the others wait in their owner's stash).
static int32_t drainCompletions(mrp_ll_ctx* ctx, int32_t tag, int32_t* tickets, int32_t cap) {
  Ring& g = ctx->ring;
  const uint32_t R = Ring::kSlots;
  int32_t n = 0;
  // drain the completion queue: entry k holds (k / R + 1) << 11 | slot once the k-th finished job has been published
  while (n < cap) {
    const uint64_t cursor = g.compCursor;
    const uint32_t e = __atomic_load_n(g.compRing + (cursor % R), __ATOMIC_ACQUIRE);
    if ((e >> mrp::kRingSlotBits) != static_cast<uint32_t>(cursor / R) + 1) break;
    __atomic_store_n(&g.compCursor, cursor + 1, __ATOMIC_RELAXED);
    const uint32_t slot = e & mrp::kRingSlotMask;
    if (!g.busy[slot]) continue;  // already consumed through mrp_ll_poll / mrp_ll_wait
    // ... and if the slot has been re-used since, this entry is stale: only the occupant's own done word counts
    if (__atomic_load_n(g.done + slot, __ATOMIC_ACQUIRE) != g.slotGen[slot]) continue;
    SessTicket& st = ctx->sess[g.slotTicket[slot]];
    const int32_t i = g.slotJob[slot];
    if (g.slotChain[slot])
      unpackChain(ctx, g.results[slot], g.outPaths + static_cast<size_t>(slot) * g.outStride, st.state[i] == 2, st.res[i],
                  g.slotChain[slot], g.outStride / 2u);
    else
      unpackResult(ctx, g.results[slot], g.outPaths + static_cast<size_t>(slot) * g.outStride, st.state[i] == 2,
                   st.res[i], g.sipp, g.sipp ? g.slotDimx[slot] : 0, g.slotInit[slot]);
    if (g.sipp && g.slotTable[slot]) {
      finishSippTableJob(g.slotTable[slot], g.slotSippFlags[slot], g.results[slot],
                         g.outPaths + static_cast<size_t>(slot) * g.outStride);
      g.slotTable[slot] = nullptr;
    }
    st.state[i] = 1;
    st.remaining -= 1;
    g.busy[slot] = 0;
    g.inFlightJobs -= 1;
    g.freeSlots.push_back(slot);
    if (st.remaining == 0) {
      const int32_t id = g.slotTicket[slot];
      if (tag >= 0 && st.tag != tag && st.tag >= 0 && st.tag < mrp_ll_ctx::kMaxTags) {
        ctx->coStash[st.tag].push_back(id);  // its owner collects (and releases) it
        ctx->coStashCount[st.tag].fetch_add(1, std::memory_order_release);
      } else {
        st.used = false;
        ctx->sessFree.push_back(id);
        tickets[n++] = id;
      }
    }
  }
  return n;
}

int mrp_ll_poll_any(mrp_ll_ctx* ctx, int32_t* tickets, int32_t cap, int32_t* nOut) {
  if (!ctx || !tickets || !nOut || cap <= 0) return MRP_LL_E_INVALID;
  Ring& g = ctx->ring;
  if (!g.active) return MRP_LL_E_INVALID;
  sessionBeat(g);
  const int32_t n = drainCompletions(ctx, -1, tickets, cap);
  *nOut = n;
  if (n != 0) {
    g.emptyPolls = 0;
    return MRP_LL_SUCCESS;
  }
  return sessionAlive(ctx);
}

int mrp_ll_submit_tagged(mrp_ll_ctx* ctx, int32_t tag, int32_t nJobs, const mrp_ll_job* jobs, mrp_ll_result* results,
                         int32_t* ticketOut) {
  if (!ctx || !ticketOut || tag < 0 || tag >= mrp_ll_ctx::kMaxTags || nJobs < 0 || (nJobs > 0 && (!jobs || !results)))
    return MRP_LL_E_INVALID;
  std::lock_guard<std::mutex> lock(ctx->coMu);
  if (!ctx->ring.active) return MRP_LL_E_INVALID;
  const int rc = sessionSubmit(ctx, 0, nJobs, jobs, results, ticketOut);
  if (rc == MRP_LL_SUCCESS) ctx->sess[*ticketOut].tag = tag;
  return rc;
}

int mrp_ll_poll_any_tagged(mrp_ll_ctx* ctx, int32_t tag, int32_t* tickets, int32_t cap, int32_t* nOut) {
  if (!ctx || !tickets || !nOut || cap <= 0 || tag < 0 || tag >= mrp_ll_ctx::kMaxTags) return MRP_LL_E_INVALID;
  Ring& g = ctx->ring;
  *nOut = 0;
  // a look without the lock: nothing stashed for this co-worker, nothing new in the completion queue
  if (ctx->coStashCount[tag].load(std::memory_order_acquire) == 0) {
    const uint64_t cursor = __atomic_load_n(&g.compCursor, __ATOMIC_RELAXED);
    const uint32_t e = __atomic_load_n(g.compRing + (cursor % Ring::kSlots), __ATOMIC_ACQUIRE);
    const uint64_t tsc = __builtin_ia32_rdtsc();
    const bool beatDue = tsc - __atomic_load_n(&g.lastBeatTsc, __ATOMIC_RELAXED) >= (1ull << 24);
    if ((e >> mrp::kRingSlotBits) != static_cast<uint32_t>(cursor / Ring::kSlots) + 1 && !beatDue) return MRP_LL_SUCCESS;
  }
  std::lock_guard<std::mutex> lock(ctx->coMu);
  if (!g.active) return MRP_LL_E_INVALID;
  sessionBeat(g);
  int32_t n = 0;
  std::vector<int32_t>& mine = ctx->coStash[tag];
  while (n < cap && !mine.empty()) {
    const int32_t id = mine.back();
    mine.pop_back();
    ctx->coStashCount[tag].fetch_sub(1, std::memory_order_relaxed);
    ctx->sess[id].used = false;
    ctx->sessFree.push_back(id);
    tickets[n++] = id;
  }
  n += drainCompletions(ctx, tag, tickets + n, cap - n);
  *nOut = n;
  if (n != 0) {
    g.emptyPolls = 0;
    return MRP_LL_SUCCESS;
  }
  return sessionAlive(ctx);
}

static int sessionWait(mrp_ll_ctx* ctx, int32_t ticket) {
  auto t0 = std::chrono::steady_clock::now();
  for (uint64_t spin = 0;; ++spin) {
    int32_t done = 0;
    int rc = mrp_ll_poll(ctx, ticket, &done);
    if (rc != MRP_LL_SUCCESS) return rc;
    if (done) return MRP_LL_SUCCESS;
    if ((spin & 0xFFF) == 0xFFF) {
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0) {
        ctx->err = "mrp_ll_wait (session): no completion within 120 s";
        return MRP_LL_E_DEVICE;
      }
      if (hipEventQuery(ctx->ring.ev1) != hipErrorNotReady) {
        ctx->err = "mrp_ll_wait (session): the resident kernel has exited (idle limit or fault)";
        return MRP_LL_E_DEVICE;
      }
    }
  }
}

int mrp_ll_submit(mrp_ll_ctx* ctx, int32_t nJobs, const mrp_ll_job* jobs, mrp_ll_result* results, int32_t* ticketOut) {
  if (!ctx || !ticketOut || nJobs < 0 || (nJobs > 0 && (!jobs || !results))) return MRP_LL_E_INVALID;
  if (ctx->ring.active) return sessionSubmit(ctx, 0, nJobs, jobs, results, ticketOut);
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
  {
    int nSipp = 0;
    for (int i = 0; i < nJobs; ++i) nSipp += jobs[i].algo == MRP_LL_SIPP ? 1 : 0;
    if (nSipp != 0 && nSipp != nJobs) {
      ctx->err = "mrp_ll_submit: a batch holds either MRP_LL_SIPP jobs or A-star jobs, not both";
      return MRP_LL_E_INVALID;
    }
    t.sipp = nSipp != 0;
    int nEps = 0;
    for (int i = 0; i < nJobs; ++i) nEps += jobs[i].algo == MRP_LL_ASTAR_EPS ? 1 : 0;
    t.kind = t.sipp ? 0 : nEps == nJobs ? 1 : nEps == 0 ? 2 : 0;  // a one-algorithm batch runs the specialised kernel
  }
  t.jobs.clear();
  t.cons.clear();
  t.paths.clear();
  HIPCHK(ctx, t.jobs.resize(std::max(nJobs, 1)));
  for (int i = 0; i < nJobs; ++i) {
    size_t c0 = t.cons.size, p0 = t.paths.size;
    ConsSinkBuf cs{t.cons};
    PathSinkBuf ps{t.paths};
    bool ok = packJob(ctx, jobs[i], cs, ps, t.jobs.host[i]);
    if (t.sipp) {
      if (static_cast<int>(t.jobDimx.size()) < nJobs) t.jobDimx.resize(nJobs);
      t.jobDimx[i] = ok ? static_cast<int32_t>(t.jobs.host[i].dimx) : 1;
    }
    if (static_cast<int>(t.jobInit.size()) < nJobs) t.jobInit.resize(nJobs);
    t.jobInit[i] = jobInitOf(jobs[i], ok);
    if (t.sipp) {
      if (static_cast<int>(t.commitTab.size()) < nJobs) t.commitTab.resize(nJobs);
      t.commitTab[i] = ok && jobs[i].sipp_table && jobs[i].sipp_commit ? const_cast<mrp_ll_sipp_table*>(jobs[i].sipp_table)
                                                                        : nullptr;
    }
    if (cs.failed || ps.failed) t.allocFailed = true;
    if (!ok) {
      // rejected: give the device a trivially capped job and remember the rejection
      t.cons.size = c0;
      t.paths.size = p0;
      t.rejected[i] = 1;
      trivialRejectedJob(ctx, t.jobs.host[i]);
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
  P.cons = t.cons.dev;
  P.paths = t.paths.dev;
  P.queue_base = t.queueBase;
  P.n_jobs = static_cast<uint32_t>(nJobs);
  uint32_t ldsBytes = 0;
  {
    int rcp = fillCommonParams(ctx, t, P, ldsBytes, t.kind);
    if (rcp != MRP_LL_SUCCESS) return rcp;
  }
  uint32_t grid = std::min<uint32_t>(static_cast<uint32_t>(nJobs), static_cast<uint32_t>(ctx->opt.slots));
  t.queueBase += static_cast<uint32_t>(nJobs) + grid;  // every workgroup takes one ticket past the end when it exits
  HIPCHK(ctx, hipEventRecord(t.evK0, t.stream));
  if (t.sipp)
    HIPCHK(ctx, mrp_ll_launch_sipp(&P, grid, t.stream));
  else
    HIPCHK(ctx, mrp_ll_launch(&P, grid, ldsBytes, t.kind, t.stream));
  HIPCHK(ctx, hipEventRecord(t.evK1, t.stream));
  ctx->stats.launches += 1;
  return MRP_LL_SUCCESS;
}

int mrp_ll_submit_lane(mrp_ll_ctx* ctx, int32_t lane, int32_t nJobs, const mrp_ll_job* jobs, mrp_ll_result* results,
                       int32_t* ticketOut) {
  if (!ctx || !ticketOut || nJobs < 0 || (nJobs > 0 && (!jobs || !results)) || lane < 0 || lane > 1) return MRP_LL_E_INVALID;
  if (!ctx->ring.active) return mrp_ll_submit(ctx, nJobs, jobs, results, ticketOut);  // batch mode has one queue
  // one device queue, first in first out: `lane` is accepted for source compatibility and otherwise ignored — priority is
  // the order in which the caller publishes (see mrp_ll.h)
  return sessionSubmit(ctx, 0, nJobs, jobs, results, ticketOut);
}

int mrp_ll_wait(mrp_ll_ctx* ctx, int32_t ticket) {
  if (ctx && ctx->ring.active) return sessionWait(ctx, ticket);
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
    unpackResult(ctx, t.results.host[i], t.outPaths.host + static_cast<size_t>(i) * outStride, t.rejected[i] != 0,
                 t.userResults[i], t.sipp, t.sipp ? t.jobDimx[i] : 0, t.jobInit[i]);
    if (t.sipp && t.commitTab[i])  // batch mode never uses the device-resident copies: the host adds the stays
      finishSippTableJob(t.commitTab[i], 2u, t.results.host[i], t.outPaths.host + static_cast<size_t>(i) * outStride);
  }
  ctx->stats.unpack_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - unpackT0).count();
  return MRP_LL_SUCCESS;
}

int mrp_ll_sync_maps(mrp_ll_ctx* ctx) {
  if (!ctx) return MRP_LL_E_INVALID;
  if (ctx->ring.active) return MRP_LL_SUCCESS;  // in-session uploads are copied immediately
  HIPCHK(ctx, hipSetDevice(ctx->device));
  return syncMaps(ctx);
}

int mrp_ll_path_store_reserve(mrp_ll_ctx* ctx, int32_t nSlots) {
  if (!ctx || nSlots < 0) return MRP_LL_E_INVALID;
  if (ctx->ring.active) return MRP_LL_E_BUSY;
  for (const Ticket& t : ctx->tickets)
    if (t.inFlight) return MRP_LL_E_BUSY;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  if (ctx->pathStore) HIPCHK(ctx, hipFree(ctx->pathStore));
  ctx->pathStore = nullptr;
  ctx->pathStoreSlots = 0;
  if (nSlots == 0) return MRP_LL_SUCCESS;
  // [len][cells...]: one halfword in front of up to max_horizon states, rounded up to 16 bytes
  ctx->pathStoreStride = (static_cast<uint32_t>(ctx->opt.max_horizon) + 1u + 7u) & ~7u;
  const size_t bytes = static_cast<size_t>(nSlots) * ctx->pathStoreStride * sizeof(uint16_t);
  // A slot is written by one workgroup of a resident kernel and read by workgroups on other CUs / XCDs of the SAME launch.
  // Ordering is carried by the host (a reader's job is published only after the writer's completion was seen: the
  // writer's stores sit in front of a system-scope release); the reader drops stale cached copies with one agent-scope
  // acquire per job (ll_kernel.hip runJob), so ordinary cached device memory is enough — and the paths of a conflict-
  // tree node, read again by job after job, are served from L2.  MRP_LL_STORE_UNCACHED=1: uncached allocation instead
  // (measured: agents100 steps 23 % longer — every table build then reads HBM).
  void* p = nullptr;
  hipError_t e = std::getenv("MRP_LL_STORE_UNCACHED") ? hipExtMallocWithFlags(&p, bytes, hipDeviceMallocUncached)
                                                      : hipMalloc(&p, bytes);
  if (e != hipSuccess) {
    ctx->err = std::string("mrp_ll_path_store_reserve: ") + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? MRP_LL_E_NOMEM : MRP_LL_E_DEVICE;
  }
  ctx->pathStore = static_cast<uint16_t*>(p);
  ctx->pathStoreSlots = static_cast<uint32_t>(nSlots);
  HIPCHK(ctx, hipMemset(ctx->pathStore, 0, bytes));
  return MRP_LL_SUCCESS;
}

int mrp_ll_sipp_table_create(mrp_ll_ctx* ctx, int32_t mapId, mrp_ll_sipp_table** out) {
  if (!ctx || !out || mapId < 0 || mapId >= static_cast<int32_t>(ctx->maps.size())) return MRP_LL_E_INVALID;
  auto* t = new mrp_ll_sipp_table();
  t->mapId = mapId;
  t->dimx = ctx->maps[mapId].dimx;
  t->dimy = ctx->maps[mapId].dimy;
  t->cellIdx.assign(static_cast<size_t>(t->dimx) * t->dimy, 0);
  t->cellIdx16.assign(static_cast<size_t>(t->dimx) * t->dimy, 0);
  t->isDirty.assign(static_cast<size_t>(t->dimx) * t->dimy, 0);
  t->ctx = ctx;
  // a slot of the device-resident table pool (grown by chunks; existing tables never move).  Failure to allocate is not
  // an error: such a table simply ships its whole contents with every job.
  if (ctx->sippTabStride == 0) {
    const size_t maxCells = static_cast<size_t>(ctx->opt.max_cells);
    ctx->sippTabStride = maxCells * mrp::kSippRowWords * 8;  // a bounds row and a status row per cell (ll_device.h)
    ctx->sippTabsPerChunk = static_cast<int32_t>(std::max<size_t>(1, std::min<size_t>(64, (size_t(64) << 20) / ctx->sippTabStride)));
  }
  if (!ctx->sippTabFree.empty()) {
    t->devIndex = ctx->sippTabFree.back();
    ctx->sippTabFree.pop_back();
  } else {
    if (ctx->sippTabNext == static_cast<int32_t>(ctx->sippTabChunks.size()) * ctx->sippTabsPerChunk) {
      // The device-resident SIPP tables live in uncached device memory: consecutive jobs of a table run on different XCDs, whose
      // L2s are not coherent with each other, so cached tables needed an acquire fence at every job start and a release at its
      // end (an invalidate / write-back of the XCD's whole L2, shared with ~190 other searches); uncached, every table access
      // goes to memory and no fence is needed.  Measured on the three prioritized-SIPP legs: 2-3 % faster per expansion
      // (scripts/r4_run24.sh, r4_run25.sh).  The cached form is gone: besides being slower, its commit test (sipp_commit with
      // the fence pair per job) returned ONE expansion count that differed from the oracle's in one of seven otherwise green runs
      // in round 4 — never seen with uncached tables — and a protocol that rests on L2 invalidates across XCDs is not worth
      // keeping as an option nobody measures.  A device without uncached allocations keeps no resident tables (jobs ship whole
      // tables: same results).
      void* c = nullptr;
      if (hipSetDevice(ctx->device) == hipSuccess &&
          hipExtMallocWithFlags(&c, ctx->sippTabStride * ctx->sippTabsPerChunk, hipDeviceMallocUncached) == hipSuccess)
        ctx->sippTabChunks.push_back(static_cast<uint8_t*>(c));
    }
    if (ctx->sippTabNext < static_cast<int32_t>(ctx->sippTabChunks.size()) * ctx->sippTabsPerChunk) t->devIndex = ctx->sippTabNext++;
  }
  *out = t;
  return MRP_LL_SUCCESS;
}

}  // extern "C"

namespace {
void sippTableAddCell(mrp_ll_sipp_table* t, size_t cell, int32_t start, int32_t end, bool markDirty) {
  if (!t->cellIdx[cell]) {
    t->spec.emplace_back();
    t->cellIdx[cell] = static_cast<int32_t>(t->spec.size());
    t->cellIdx16[cell] = static_cast<uint16_t>(t->spec.size());
  }
  mrp_ll_sipp_table::Spec& sp = t->spec[t->cellIdx[cell] - 1];
  const bool first = sp.collisions.empty();
  sp.collisions.push_back(start);
  sp.collisions.push_back(end);
  t->totalSafe -= static_cast<uint32_t>(sp.safe.size());
  // The usual case (a planner adds the stays of a path that avoided every earlier one): the new collision interval lies
  // inside ONE safe interval, and sorting it into the list splits exactly that gap — [a, start - 1] if non-empty and
  // [end + 1, b] if non-empty — which is what setCollisionIntervals' loop (sipp.hpp:258-277) yields for the longer list.
  // Anything else (overlaps, start > end) recomputes the cell from all its collision intervals, as before.
  bool split = false;
  if (first) sp.safe.assign(1, SippScratch::Iv{0, INT32_MAX});
  if (sp.disjoint && start <= end && start >= 0) {
    for (size_t k = 0; k < sp.safe.size(); ++k) {
      const SippScratch::Iv g = sp.safe[k];
      if (g.s <= start && end <= g.e) {
        const bool left = g.s <= start - 1, right = end < g.e;
        if (left && right) {
          sp.safe[k].e = start - 1;
          sp.safe.insert(sp.safe.begin() + k + 1, SippScratch::Iv{end + 1, g.e});
        } else if (left) {
          sp.safe[k].e = start - 1;
        } else if (right) {
          sp.safe[k].s = end + 1;
        } else {
          sp.safe.erase(sp.safe.begin() + k);
        }
        split = true;
        break;
      }
    }
  }
  if (!split) {
    sp.disjoint = false;
    sp.safe.clear();
    safeFromCollisions(sp.collisions.data(), static_cast<int>(sp.collisions.size() / 2), t->scratch, sp.safe);
  }
  t->totalSafe += static_cast<uint32_t>(sp.safe.size());
  // what the resident layout cannot hold — more than kSippCap intervals on a cell, a finite bound that does not fit a
  // halfword: from now on this table travels whole (packSippFromTable)
  if (sp.safe.size() > mrp::kSippCap) t->overflow = true;
  for (const SippScratch::Iv& v : sp.safe)
    if (v.s < 0 || v.s >= static_cast<int32_t>(mrp::kSippEndInf) || (v.e != INT32_MAX && (v.e < 0 || v.e >= static_cast<int32_t>(mrp::kSippEndInf))))
      t->overflow = true;
  if (markDirty && !t->isDirty[cell]) {
    t->isDirty[cell] = 1;
    t->dirty.push_back(static_cast<int32_t>(cell));
  }
}
}  // namespace

extern "C" {

int mrp_ll_sipp_table_add(mrp_ll_sipp_table* t, int32_t x, int32_t y, int32_t start, int32_t end) {
  if (!t) return MRP_LL_E_INVALID;
  if (x < 0 || x >= t->dimx || y < 0 || y >= t->dimy) return MRP_LL_SUCCESS;  // never visited
  if (!t->log.empty()) sippTableSync(t);
  sippTableAddCell(t, static_cast<size_t>(y) * t->dimx + x, start, end, true);
  return MRP_LL_SUCCESS;
}

void mrp_ll_sipp_table_destroy(mrp_ll_sipp_table* t) {
  if (!t) return;
  bool referenced = false;
  if (t->ctx) {
    // a job on this table may still be in flight (its slot reports back to the table, and with sipp_commit a workgroup
    // may still be writing the device copy): forget the slot's reference, and do not hand the device copy to a new
    // table — its pool index is simply retired (0.8 MB of device memory per such destroy, until the engine goes)
    for (mrp_ll_sipp_table*& tb : t->ctx->ring.slotTable)
      if (tb == t) {
        tb = nullptr;
        referenced = true;
      }
    for (Ticket& tk : t->ctx->tickets)
      for (mrp_ll_sipp_table*& tb : tk.commitTab)
        if (tb == t) {
          tb = nullptr;
          referenced = true;
        }
    if (t->devIndex >= 0 && !referenced && !t->inFlight) t->ctx->sippTabFree.push_back(t->devIndex);
  }
  delete t;
}

int mrp_ll_release_maps(mrp_ll_ctx* ctx) {
  if (!ctx) return MRP_LL_E_INVALID;
  if (ctx->ring.active) return MRP_LL_E_BUSY;
  for (const Ticket& t : ctx->tickets)
    if (t.inFlight) return MRP_LL_E_BUSY;
  ctx->maps.clear();
  ctx->heurs.clear();
  ctx->mapWords.clear();
  ctx->mapsDirty = false;  // nothing to copy; the device buffer (and its capacity) is kept for the next uploads
  return MRP_LL_SUCCESS;
}

int mrp_ll_search_batch(mrp_ll_ctx* ctx, int32_t nJobs, const mrp_ll_job* jobs, mrp_ll_result* results) {
  int32_t ticket = -1;
  int rc = mrp_ll_submit(ctx, nJobs, jobs, results, &ticket);
  if (rc != MRP_LL_SUCCESS) return rc;
  return mrp_ll_wait(ctx, ticket);
}

int mrp_ll_conflict_scan(mrp_ll_ctx* ctx, int32_t nSets, const int32_t* setFirstAgent, const int32_t* pathFirstState,
                         const int32_t* statesXY, mrp_ll_conflict* out) {
  static_assert(sizeof(mrp_ll_conflict) == sizeof(mrp::ConflictOut), "mrp_ll_conflict layout");
  if (!ctx || nSets < 0 || (nSets > 0 && (!setFirstAgent || !pathFirstState || !out))) return MRP_LL_E_INVALID;
  if (nSets == 0) return MRP_LL_SUCCESS;
  HIPCHK(ctx, hipSetDevice(ctx->device));
  const int64_t nAgents = setFirstAgent[nSets];
  if (setFirstAgent[0] != 0 || nAgents < 0) return MRP_LL_E_INVALID;
  for (int32_t s = 0; s < nSets; ++s)
    if (setFirstAgent[s + 1] < setFirstAgent[s] || setFirstAgent[s + 1] - setFirstAgent[s] > 65535) return MRP_LL_E_INVALID;
  const int64_t nStates = nAgents ? pathFirstState[nAgents] : 0;
  if (nAgents && (pathFirstState[0] != 0 || !statesXY)) return MRP_LL_E_INVALID;
  for (int64_t a = 0; a < nAgents; ++a)
    if (pathFirstState[a + 1] <= pathFirstState[a]) {  // getState asserts a non-empty path (ecbs.cpp:491)
      ctx->err = "mrp_ll_conflict_scan: every path needs at least one state";
      return MRP_LL_E_INVALID;
    }
  ctx->scanStates.resize(static_cast<size_t>(nStates));
  for (int64_t k = 0; k < nStates; ++k) {
    const int32_t x = statesXY[2 * k], y = statesXY[2 * k + 1];
    if (x < 0 || x > 255 || y < 0 || y > 255) {
      ctx->err = "mrp_ll_conflict_scan: coordinates must be 0..255";
      return MRP_LL_E_INVALID;
    }
    ctx->scanStates[k] = static_cast<uint16_t>(x | (y << 8));
  }
  auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
  const size_t oSet = 0, oPath = al((nSets + 1) * 4), oStates = oPath + al((nAgents + 1) * 4),
               oOut = oStates + al(static_cast<size_t>(nStates) * 2), total = oOut + al(sizeof(mrp_ll_conflict) * nSets);
  if (total > ctx->scanDevCap) {
    if (ctx->scanDev) HIPCHK(ctx, hipFree(ctx->scanDev));
    ctx->scanDev = nullptr;
    ctx->scanDevCap = 0;
    HIPCHK(ctx, hipMalloc(reinterpret_cast<void**>(&ctx->scanDev), total * 2));
    ctx->scanDevCap = total * 2;
  }
  // A stream of its own (non-blocking): during a session tickets[0].stream is held by the resident kernel until
  // mrp_ll_session_end, and a scan queued behind it would never start (while its caller, blocked here, stops moving the
  // session's heartbeat).  The scan kernel runs beside the resident wavefronts: they leave wave slots and registers free.
  if (!ctx->scanStream) HIPCHK(ctx, hipStreamCreateWithFlags(&ctx->scanStream, hipStreamNonBlocking));
  hipStream_t st = ctx->scanStream;
  HIPCHK(ctx, hipMemcpyAsync(ctx->scanDev + oSet, setFirstAgent, (nSets + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(ctx, hipMemcpyAsync(ctx->scanDev + oPath, pathFirstState, (nAgents + 1) * 4, hipMemcpyHostToDevice, st));
  if (nStates)
    HIPCHK(ctx, hipMemcpyAsync(ctx->scanDev + oStates, ctx->scanStates.data(), static_cast<size_t>(nStates) * 2,
                               hipMemcpyHostToDevice, st));
  mrp::ConflictParams P;
  P.setFirstAgent = reinterpret_cast<const uint32_t*>(ctx->scanDev + oSet);
  P.pathFirstState = reinterpret_cast<const uint32_t*>(ctx->scanDev + oPath);
  P.states = reinterpret_cast<const uint16_t*>(ctx->scanDev + oStates);
  P.out = reinterpret_cast<mrp::ConflictOut*>(ctx->scanDev + oOut);
  P.nSets = static_cast<uint32_t>(nSets);
  HIPCHK(ctx, mrp_ll_launch_conflict(&P, st));
  HIPCHK(ctx, hipMemcpyAsync(out, ctx->scanDev + oOut, sizeof(mrp_ll_conflict) * nSets, hipMemcpyDeviceToHost, st));
  HIPCHK(ctx, hipStreamSynchronize(st));
  return MRP_LL_SUCCESS;
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
