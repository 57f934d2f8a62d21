// C-ABI of the host-side conflict-tree drivers (include/mrp_hl.h).  Worker threads each own one low-level engine
// context (mrp_ll_ctx is not thread-safe).  Default schedule (session mode): every worker keeps a resident kernel fed
// through the engine's job ring, draws instances from one shared pool and publishes an instance's next searches the
// moment its previous ones are back (runGroupSession, runSippGroupSession).  Round-based schedule (mode 1 /
// MRP_HL_SIPP_BATCH): every round a thread gathers the ready searches of its instances into one batch (runGroup,
// runSippGroup).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <climits>
#include <cstring>
#include <deque>
#include <map>
#include <queue>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include <pthread.h>
#include <sched.h>

#include "../../../include/mrp_hl.h"
#include "ct_solver.hpp"
#include "grid2d_astar.hpp"
#include "instance_io.hpp"

using namespace mrp_hl;

struct mrp_hl_preloaded {
  mrp_hl_solver* owner = nullptr;
  int32_t nInst = 0;
  const mrp_hl_instance* instances = nullptr;
  std::vector<std::vector<int32_t>> idx, mapIds;  // per worker: instance indices and their map ids on that engine
  std::vector<int32_t> mapBase;                   // per worker: map id of instance 0 on that engine (every engine
                                                  // holds every map, so any worker can take any instance)
};

struct mrp_hl_solver {
  int32_t device = 0;
  std::vector<mrp_ll_ctx*> engines;
  mrp_ll_options llOpt;
  std::string err;
  int32_t nWorkers = 0;    // host worker threads the caller asked for (>= engines.size(): two may share an engine)
  int32_t nPreloaded = 0;  // live mrp_hl_preloaded objects (their maps are released with the last one)
  int32_t pathSlots = 0;   // slots of the engines' device path stores (0: not allocated)
};

namespace {

// A session loop gives up when nothing at all has come back for this long (a dead resident kernel is reported much
// sooner by mrp_ll_poll_any's own liveness check).
constexpr double kNoProgressLimitS = 600.0;

// Two workers on one engine ("co-workers", mrp_ll.h mrp_ll_submit_tagged): the leader (index 0) begins and ends the
// session, the other one waits for it on both sides.
struct CoSync {
  std::atomic<int32_t> begun{0};     // 1: the session runs, -1: it could not be started
  std::atomic<int32_t> finished{0};  // co-workers that have left their loops
};

struct GroupResult {
  int64_t rounds = 0, searches = 0, expansions = 0;  // expansions: of the searches the conflict trees CONSUMED
  int64_t specSearches = 0, specWasted = 0;          // searches issued ahead of their node's pop; expansions that were run but never consumed
  int64_t rootSolved = 0;                            // instances whose root node was conflict-free, written out without a conflict tree
  double buildS = 0, llS = 0, consumeS = 0;
  std::string err;
};

// Several preloaded batches as ONE pool of instances (mrp_hl.h mrp_hl_solver_solve_stream): global index k names instance
// k - first[b] of batch b; its map id on engine e is mapBase[b][e] + that.
struct StreamView {
  std::vector<int32_t> first;                      // first[b] = global index of batch b's instance 0; first[nBatches] = total
  std::vector<const mrp_hl_instance*> inst;        // per batch
  std::vector<mrp_hl_solution*> sols;              // per batch
  std::vector<const std::vector<int32_t>*> mapBase;  // per batch: per engine
  int32_t batchOf(int32_t k) const {
    return static_cast<int32_t>(std::upper_bound(first.begin(), first.end(), k) - first.begin()) - 1;
  }
  mrp_hl_solution& sol(int32_t k) const {
    const int32_t b = batchOf(k);
    return sols[b][k - first[b]];
  }
};

// One low-level job of the C-ABI for request `r` of instance `I`; the focal-context arrays go to the pools (pointers
// are patched in by the caller once the pools have stopped growing).
// `idPool` != nullptr: also name the context paths by their device path-store slots (f2) when every one of them has one.
void fillJob(const Instance& I, const LLRequest& r, mrp_ll_job& j, std::vector<int32_t>& pathLenPool,
             std::vector<const int32_t*>& pathPtrPool, std::vector<int32_t>* idPool = nullptr, bool* idsOk = nullptr) {
  std::memset(&j, 0, sizeof(j));
  j.map_id = I.mapId();
  j.algo = I.algo() == MRP_HL_ECBS ? MRP_LL_ASTAR_EPS : MRP_LL_ASTAR;
  j.w = I.w();
  j.agent_idx = r.agent;
  j.start_x = I.start(r.agent)[0];
  j.start_y = I.start(r.agent)[1];
  j.goal_x = I.goal(r.agent)[0];
  j.goal_y = I.goal(r.agent)[1];
  j.n_vertex_constraints = static_cast<int32_t>(r.constraints->vertex.size() / 3);
  j.vertex_constraints = r.constraints->vertex.data();
  j.n_edge_constraints = static_cast<int32_t>(r.constraints->edge.size() / 5);
  j.edge_constraints = r.constraints->edge.data();
  j.max_expansions = I.remainingLL();
  j.result_path_id = -1;
  if (idsOk) *idsOk = false;
  if (r.context) {
    j.n_agents = static_cast<int32_t>(r.context->size());
    bool all = idPool != nullptr;
    int32_t a = 0;
    for (const PathPtr& p : *r.context) {
      pathLenPool.push_back(p->len());
      pathPtrPool.push_back(p->xy.data());
      if (idPool) {
        const bool needed = a != r.agent && p->len() > 0;
        idPool->push_back(needed ? p->devSlot : -1);
        if (needed && p->devSlot < 0) all = false;
      }
      ++a;
    }
    if (idsOk) *idsOk = all;
  }
}

// `slot` / `pool`: the path-store slot the job was given for its result path (-1: none)
LLAnswer answerOf(const mrp_ll_result& r, int32_t slot = -1, SlotPool* pool = nullptr) {
  LLAnswer a;
  a.status = r.status;
  a.cost = r.cost;
  a.fmin = r.fmin;
  a.expanded = r.expanded;
  if (r.status == MRP_LL_OK) {
    auto p = std::make_shared<Path>();
    p->xy.resize(static_cast<size_t>(r.n_states) * 2);
    uint32_t orAll = 0;
    for (int32_t s = 0; s < r.n_states; ++s) {
      p->xy[2 * s] = r.states_txy[3 * s + 1];
      p->xy[2 * s + 1] = r.states_txy[3 * s + 2];
      orAll |= static_cast<uint32_t>(p->xy[2 * s]) | static_cast<uint32_t>(p->xy[2 * s + 1]);
    }
    p->fits8 = orAll < 256u;
    if (p->fits8) p->packCells();
    p->cost = r.cost;
    p->fmin = r.fmin;
    if (pool && slot >= 0) {
      p->devSlot = slot;
      p->pool = pool;
    }
    a.path = p;
  } else if (pool) {
    pool->give(slot);  // no path came out of this search
  }
  return a;
}

void writeSolution(const Instance& I, mrp_hl_solution& s) {
  s.status = I.status();
  s.n_ll_searches = I.llSearches();
  s.high_level_expanded = I.hlExpanded();
  s.low_level_expanded = I.llExpanded();
  s.cost = 0;
  s.makespan = 0;
  s.schedule_digest = 0;
  if (I.status() == MRP_HL_SOLVED) {
    const auto& sol = I.finalSolution();
    uint64_t h = 14695981039346656037ull;
    auto mix = [&h](uint32_t byte) { h = (h ^ (byte & 0xFFu)) * 1099511628211ull; };
    for (int32_t a = 0; a < I.nAgents(); ++a) {
      const int32_t* q = sol[a]->xy.data();
      for (int32_t k = 0, n = sol[a]->len(); k < n; ++k) {
        mix(static_cast<uint32_t>(q[2 * k]));
        mix(static_cast<uint32_t>(q[2 * k + 1]));
      }
      mix(0xFFu);
    }
    s.schedule_digest = h;
    for (int32_t a = 0; a < I.nAgents(); ++a) {
      s.cost += sol[a]->cost;
      s.makespan = std::max<int64_t>(s.makespan, sol[a]->cost);
      if (s.path_len) s.path_len[a] = sol[a]->len();
      if (s.paths_xy) {
        int32_t m = std::min(sol[a]->len(), s.path_cap);
        std::memcpy(s.paths_xy + static_cast<size_t>(a) * s.path_cap * 2, sol[a]->xy.data(), sizeof(int32_t) * 2 * m);
      }
    }
  }
}

// The solution of an instance whose ROOT node has no conflict, written straight from the results of its root chain
// (mrp_ll.h MRP_LL_JOB_ROOT_CHAIN: the workgroup that planned the agents also scanned their paths): what ECBS::search
// returns when the first node it pops is conflict-free (ecbs.hpp:227-240) — cost = sum of the agents' costs, one
// high-level expansion — without building a single conflict-tree object.
void writeRootSolution(const std::vector<mrp_ll_result>& r, mrp_hl_solution& s) {
  s.status = MRP_HL_SOLVED;
  s.n_ll_searches = static_cast<int32_t>(r.size());
  s.high_level_expanded = 1;
  s.low_level_expanded = 0;
  s.cost = 0;
  s.makespan = 0;
  uint64_t h = 14695981039346656037ull;
  for (size_t a = 0; a < r.size(); ++a) {
    s.cost += r[a].cost;
    s.makespan = std::max<int64_t>(s.makespan, r[a].cost);
    s.low_level_expanded += r[a].expanded;
    const int32_t n = r[a].n_states;
    const int32_t* q = r[a].states_txy;
    if (s.path_len) s.path_len[a] = n;
    int32_t* dst = s.paths_xy ? s.paths_xy + a * static_cast<size_t>(s.path_cap) * 2 : nullptr;
    for (int32_t k = 0; k < n; ++k) {
      const uint32_t x = static_cast<uint32_t>(q[3 * k + 1]), y = static_cast<uint32_t>(q[3 * k + 2]);
      h = (h ^ (x & 0xFFu)) * 1099511628211ull;
      h = (h ^ (y & 0xFFu)) * 1099511628211ull;
      if (dst && k < s.path_cap) {
        dst[2 * k] = static_cast<int32_t>(x);
        dst[2 * k + 1] = static_cast<int32_t>(y);
      }
    }
    h = (h ^ 0xFFu) * 1099511628211ull;
  }
  s.schedule_digest = h;
}

// Speculation width of the conflict-tree machines (ct_solver.hpp): MRP_HL_SPEC=k.  Default 2: measured on the shipped
// 8x8 CBS inputs (scripts/spec_probe.py) one node of look-ahead halves the time of a small batch (agents8 0.74 -> 0.37 s,
// agents10-12 1.78 -> 0.91 s) and wider windows give it back (their searches queue in front of the popped node's own);
// ECBS pops a fresh child next almost every time, so looking ahead buys it 0-6 %.
int32_t specWidthSetting() {
  if (const char* e = std::getenv("MRP_HL_SPEC")) return std::max(1, std::atoi(e));
  return 2;
}

// Drives instances idx[...] to completion on one engine, one mrp_ll_search_batch per round of ready searches.
void runGroup(mrp_ll_ctx* ctx, const mrp_hl_options& opt, const mrp_hl_instance* instIn, mrp_hl_solution* sols,
              const std::vector<int32_t>& idx, const std::vector<int32_t>& mapIds, int32_t horizon, GroupResult& out) {
  const size_t n = idx.size();
  std::vector<std::unique_ptr<Instance>> inst(n);
  for (size_t k = 0; k < n; ++k) inst[k].reset(new Instance(instIn[idx[k]], mapIds[k], opt));
  std::vector<std::vector<LLRequest>> req(n), nextReq(n);
  for (size_t k = 0; k < n; ++k) inst[k]->start(req[k]);

  std::vector<mrp_ll_job> jobs;
  std::vector<mrp_ll_result> results;
  std::vector<int32_t> owner;           // job -> local instance
  std::vector<int32_t> pathLenPool;     // per job: n_agents ints
  std::vector<const int32_t*> pathPtrPool;
  std::vector<size_t> poolOff;
  std::vector<int32_t> statesPool;
  std::vector<LLAnswer> ans;
  const int32_t cap = horizon;
  int64_t ranExpansions = 0;

  auto now = []() { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  for (;;) {
    auto tA = now();
    jobs.clear();
    owner.clear();
    pathLenPool.clear();
    pathPtrPool.clear();
    poolOff.clear();
    for (size_t k = 0; k < n; ++k) {
      for (const LLRequest& r : req[k]) {
        mrp_ll_job j;
        poolOff.push_back(pathLenPool.size());
        fillJob(*inst[k], r, j, pathLenPool, pathPtrPool);
        jobs.push_back(j);
        owner.push_back(static_cast<int32_t>(k));
      }
    }
    if (jobs.empty()) break;
    // pools may have reallocated while growing: patch the pointers now
    for (size_t q = 0; q < jobs.size(); ++q)
      if (jobs[q].n_agents > 0) {
        jobs[q].path_len = pathLenPool.data() + poolOff[q];
        jobs[q].path_xy = pathPtrPool.data() + poolOff[q];
      }
    results.assign(jobs.size(), mrp_ll_result());
    statesPool.resize(jobs.size() * static_cast<size_t>(cap) * 3);
    for (size_t q = 0; q < jobs.size(); ++q) {
      std::memset(&results[q], 0, sizeof(mrp_ll_result));
      results[q].states_txy = statesPool.data() + q * static_cast<size_t>(cap) * 3;
      results[q].actions = nullptr;
      results[q].states_cap = cap;
    }
    auto tB = now();
    int rc = mrp_ll_search_batch(ctx, static_cast<int32_t>(jobs.size()), jobs.data(), results.data());
    auto tC = now();
    if (rc != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_search_batch: ") + mrp_ll_last_error(ctx);
      return;
    }
    out.rounds += 1;
    out.searches += static_cast<int64_t>(jobs.size());
    // answers go back group by group (the requests of one group are consecutive), in request order
    size_t q = 0;
    for (size_t k = 0; k < n; ++k) {
      nextReq[k].clear();
      const std::vector<LLRequest>& rq = req[k];
      for (size_t a = 0; a < rq.size();) {
        size_t b = a;
        ans.clear();
        while (b < rq.size() && rq[b].group == rq[a].group) {
          ranExpansions += results[q].expanded;
          ans.push_back(answerOf(results[q]));
          ++q;
          ++b;
        }
        inst[k]->deliver(rq[a].group, ans, nextReq[k]);
        a = b;
      }
      if (inst[k]->done()) nextReq[k].clear();  // requests of a finished instance point into freed CT nodes
      req[k].swap(nextReq[k]);
    }
    auto tD = now();
    out.buildS += secs(tA, tB);
    out.llS += secs(tB, tC);
    out.consumeS += secs(tC, tD);
  }
  for (size_t k = 0; k < n; ++k) {
    writeSolution(*inst[k], sols[idx[k]]);
    out.expansions += inst[k]->llExpanded();
    out.specSearches += inst[k]->specSearches();
  }
  out.specWasted += ranExpansions - out.expansions;
}

// Session mode: the engine keeps `workgroups` wavefronts resident (mrp_ll_session_begin_algo) and every instance submits
// its next searches the moment the ones they depend on have finished — no instance ever waits for another one's search.
// Every group of requests (the two children of one CT node, or a root step) is one ticket.  While fewer searches are
// in flight than the engine has resident wavefronts, the conflict-tree machines look ahead (ct_solver.hpp,
// "speculative expansion"): idle wavefronts pre-compute the children of the nodes that will probably be popped next.
// `shared` != nullptr: the workers draw instances 0..nTotal-1 from one counter as their own active set drains, so a
// worker whose instances turn out easy takes more of them (map id of instance k on this engine = mapBase + k);
// otherwise the worker owns exactly idx[...].
// MRP_HL_TIMING only: the moment the current batch call started (set by the solve entry points)
std::chrono::steady_clock::time_point& batchEpoch() {
  static std::chrono::steady_clock::time_point t;
  return t;
}

// MRP_HL_PIN="base[,stride]": worker t of a batch call runs on CPU base + t * stride (default: wherever the scheduler
// puts it).  Tuning knob for hosts whose CPUs are spread over sockets / SMT siblings.
void pinWorker(int32_t t) {
  const char* e = std::getenv("MRP_HL_PIN");
  if (!e) return;
  int base = 0, stride = 1;
  if (std::sscanf(e, "%d,%d", &base, &stride) < 1) return;
  cpu_set_t set;
  CPU_ZERO(&set);
  CPU_SET(base + t * stride, &set);
  (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

void runGroupSession(mrp_ll_ctx* ctx, const mrp_hl_options& opt, const mrp_hl_instance* instIn, mrp_hl_solution* sols,
                     const std::vector<int32_t>& idx, const std::vector<int32_t>& mapIds, int32_t horizon,
                     int32_t workgroups, int32_t pathSlots, GroupResult& out, std::atomic<int32_t>* shared = nullptr,
                     int32_t nTotal = 0, int32_t mapBase = 0, int32_t nWorkersIn = 1, int32_t heavyWgs = 0,
                     int32_t* gate = nullptr, int32_t nEngines = 1, int32_t coIndex = 0, int32_t coCount = 1,
                     CoSync* co = nullptr, const StreamView* view = nullptr, int32_t engineIdx = 0) {
  // `view` (shared pool only): the pool is several batches back to back; instIn / sols / mapBase are not used then
  auto solOf = [&](int32_t g) -> mrp_hl_solution& { return view ? view->sol(g) : sols[g]; };
  int32_t viewBatch = 0;  // the counter only grows, so the batch of the next instance is found by walking forward
  const bool shared2 = coCount > 1 && co != nullptr;  // this engine has two workers: tagged calls only
  const size_t nWorkers = static_cast<size_t>(std::max(nWorkersIn, 1));
  const size_t n = shared ? static_cast<size_t>(nTotal) : idx.size();
  struct Live {
    std::unique_ptr<Instance> inst;
    std::vector<LLRequest> req;   // not submitted yet: req[reqHead..)
    size_t reqHead = 0;
    bool queued = false;          // in `backlog`
    bool counted = false;         // its completion has been taken off nActive
    double tAdmit = 0, tDone = 0;  // MRP_HL_TIMING only: seconds since the loop started
    int32_t noChainAgent = -1;    // root agent whose search outgrew the compact tier inside a chain: it goes as its own job
    int64_t hl = 0, ll = 0, spec = 0;  // ... and what the instance had consumed when it was retired
    int32_t searches = 0;
  };
  struct Pending {                // one ticket in flight
    size_t live = 0;
    int32_t group = 0;
    std::vector<mrp_ll_result> res;
    std::vector<int32_t> states;
    std::vector<int32_t> outSlot;  // per job: the path-store slot its result path also goes to (-1: none)
    // a root chain (MRP_LL_JOB_ROOT_CHAIN): ONE job whose result fans out into chainRes, one per agent from chainFirst on
    std::vector<mrp_ll_result> chainRes;
    int32_t chainFirst = -1;
    int32_t chainCount = 0;           // agents the job was asked to plan (MRP_LL_NOT_RUN behind them is not a tier overflow)
    std::vector<LLRequest> chainReq;  // the request the chain was made from (restored if the chain ran nothing)
  };
  // f2: slots of the engine's device-resident path store, handed to the searches of this worker for their result paths;
  // declared before `live` so that it outlives every Path that returns its slot to it
  SlotPool slotPool;
  slotPool.next = shared2 ? coIndex * (pathSlots / coCount) : 0;  // co-workers split the engine's path store
  slotPool.cap = shared2 ? slotPool.next + pathSlots / coCount : pathSlots;
  std::vector<int32_t> idPool;
  std::vector<size_t> idOff;
  std::vector<uint8_t> idOk;
  std::deque<Live> live;        // grows as instances are admitted; references stay valid
  std::vector<int32_t> gidx;    // live entry -> instance index
  std::deque<Pending> pend;
  std::vector<int32_t> pendFree;
  std::vector<int32_t> ticketPend;  // session ticket id -> pend entry
  const int32_t cap = horizon;
  std::vector<mrp_ll_job> jobs;
  std::vector<int32_t> pathLenPool;
  std::vector<const int32_t*> pathPtrPool;
  std::vector<size_t> poolOff;
  std::vector<LLAnswer> ans;

  std::vector<int32_t> chainIds, chainXy;
  const bool chainDebug = std::getenv("MRP_HL_CHAIN_DEBUG") != nullptr;  // one line per chain answer on stderr
  const int32_t chainChunk = std::getenv("MRP_HL_CHAIN_CHUNK") ? std::max(1, std::atoi(std::getenv("MRP_HL_CHAIN_CHUNK"))) : 8;
  // Between 33 and 63 agents the root step goes out one job per search: measured at fifty agents every form of chain
  // (whole, or in jobs of 4 / 8 / 16 searches) is 5-25 % slower than that, at a hundred agents jobs of eight are 15 %
  // faster (scripts/r4_run19.sh, r4_run20.sh)
  const int32_t chainChunkFrom = std::getenv("MRP_HL_CHAIN_CHUNK_FROM") ? std::atoi(std::getenv("MRP_HL_CHAIN_CHUNK_FROM")) : 64;
  // MRP_HL_ROOT_CHAIN=0: every root search is its own job (A/B; results are the same)
  // (not const: an engine that cannot run chains — no compact tier, a window too small for the chain's focal table —
  // rejects the first one, and this worker goes on with one job per root search)
  bool rootChains = pathSlots > 0 && opt.algo == MRP_HL_ECBS &&
                    !(std::getenv("MRP_HL_ROOT_CHAIN") && std::atoi(std::getenv("MRP_HL_ROOT_CHAIN")) == 0);
  const bool timing = std::getenv("MRP_HL_TIMING") != nullptr;
  // MRP_HL_ROOT_FAST=0: every instance goes through its conflict-tree machine (A/B; results are the same)
  const bool rootFastPath = !(std::getenv("MRP_HL_ROOT_FAST") && std::atoi(std::getenv("MRP_HL_ROOT_FAST")) == 0);
  const int32_t specK = specWidthSetting();
  auto tg0 = std::chrono::steady_clock::now();
  // ECBS: front workgroups (the LDS tier alone) + heavy workgroups that take over the searches that outgrow it; all
  // workers' heavy launches go first (the gate), then the front ones (mrp_ll.h mrp_ll_session_begin_tiers_gated)
  if (!shared2 || coIndex == 0) {
    if (mrp_ll_session_begin_tiers_gated(ctx, opt.algo == MRP_HL_ECBS ? MRP_LL_ASTAR_EPS : MRP_LL_ASTAR, workgroups,
                                         opt.algo == MRP_HL_ECBS ? heavyWgs : 0, gate, nEngines) != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_session_begin_tiers: ") + mrp_ll_last_error(ctx);
      if (co) co->begun.store(-1, std::memory_order_release);
      return;
    }
    if (co) co->begun.store(1, std::memory_order_release);
  } else {
    int32_t b;
    while ((b = co->begun.load(std::memory_order_acquire)) == 0) std::this_thread::yield();
    if (b < 0) {
      out.err = "the engine's session could not be started (see the leading worker)";
      return;
    }
  }
  auto llSubmit = [&](int32_t nJ, const mrp_ll_job* jj, mrp_ll_result* rr, int32_t* tk) {
    return shared2 ? mrp_ll_submit_tagged(ctx, coIndex, nJ, jj, rr, tk) : mrp_ll_submit(ctx, nJ, jj, rr, tk);
  };
  const int32_t myWorkgroups = std::max(1, workgroups / std::max(coCount, 1));  // this worker's share of the resident wavefronts
  size_t ticketsOut = 0;
  int64_t jobsOut = 0, ranExpansions = 0;
  // Submits the first group of live[k]'s unsent requests.  Returns 1 submitted, 0 ring full (retry later), -1 error.
  auto submitGroup = [&](size_t k) -> int {
    Live& L = live[k];
    const Instance& I = *L.inst;
    jobs.clear();
    pathLenPool.clear();
    pathPtrPool.clear();
    poolOff.clear();
    idPool.clear();
    idOff.clear();
    idOk.clear();
    const int32_t group = L.req[L.reqHead].group;
    // ---- the root step of an ECBS tree as ONE job (MRP_LL_JOB_ROOT_CHAIN): the workgroup plans this agent and every
    // later one against the paths before them and keeps the focal table in LDS; the host sees one completion instead of
    // ten.  Only when every existing path sits in the device store and there are slots for the new ones.
    if (rootChains && group == kRootGroup && I.algo() == MRP_HL_ECBS && L.req[L.reqHead].context &&
        L.reqHead + 1 == L.req.size() && I.nAgents() >= 2 && I.nAgents() <= 128 && (I.nAgents() <= 32 || I.nAgents() >= chainChunkFrom) &&
        L.req[L.reqHead].agent != L.noChainAgent) {
      const LLRequest& r = L.req[L.reqHead];
      const int32_t nA = I.nAgents(), first = r.agent;
      chainIds.assign(nA, -1);
      bool ok = true;
      for (int32_t a = 0; a < first && ok; ++a) {
        chainIds[a] = (*r.context)[a]->devSlot;
        ok = chainIds[a] >= 0;
      }
      for (int32_t a = first; a < nA && ok; ++a) {
        chainIds[a] = slotPool.take();
        ok = chainIds[a] >= 0;
      }
      if (!ok) {
        for (int32_t a = first; a < nA; ++a) slotPool.give(chainIds[a]);
      } else {
        chainXy.resize(static_cast<size_t>(nA) * 4);
        for (int32_t a = 0; a < nA; ++a) {
          chainXy[4 * a] = I.start(a)[0];
          chainXy[4 * a + 1] = I.start(a)[1];
          chainXy[4 * a + 2] = I.goal(a)[0];
          chainXy[4 * a + 3] = I.goal(a)[1];
        }
        mrp_ll_job j;
        std::memset(&j, 0, sizeof(j));
        j.map_id = I.mapId();
        j.algo = MRP_LL_ASTAR_EPS;
        j.w = I.w();
        j.agent_idx = first;
        j.n_agents = nA;
        j.path_ids = chainIds.data();
        j.chain_starts_goals_xy = chainXy.data();
        j.max_expansions = I.remainingLL();
        j.result_path_id = -1;
        j.flags = MRP_LL_JOB_ROOT_CHAIN;
        // many agents: jobs of at most eight searches — a root step of fifty searches in ONE job holds its wavefront for
        // tens of milliseconds, and the two-search rounds of deep conflict trees queue behind such jobs (measured at fifty
        // agents: the step 26 % longer than with one job per root search)
        j.chain_count = nA >= chainChunkFrom ? chainChunk : 0;
        int32_t pi;
        if (!pendFree.empty()) {
          pi = pendFree.back();
          pendFree.pop_back();
        } else {
          pend.emplace_back();
          pi = static_cast<int32_t>(pend.size()) - 1;
        }
        Pending& P = pend[pi];
        const int32_t cnt = nA - first;
        P.live = k;
        P.group = group;
        P.chainFirst = first;
        P.chainCount = j.chain_count > 0 ? std::min(j.chain_count, nA - first) : nA - first;
        P.outSlot.assign(chainIds.begin() + first, chainIds.end());
        P.chainRes.assign(static_cast<size_t>(cnt), mrp_ll_result());
        P.states.resize(static_cast<size_t>(cnt) * static_cast<size_t>(cap) * 3);
        for (int32_t q = 0; q < cnt; ++q) {
          std::memset(&P.chainRes[q], 0, sizeof(mrp_ll_result));
          P.chainRes[q].states_txy = P.states.data() + static_cast<size_t>(q) * static_cast<size_t>(cap) * 3;
          P.chainRes[q].states_cap = cap;
        }
        P.res.assign(1, mrp_ll_result());
        std::memset(&P.res[0], 0, sizeof(mrp_ll_result));
        P.res[0].chain_results = P.chainRes.data();
        int32_t ticket = -1;
        int rc = llSubmit(1, &j, P.res.data(), &ticket);
        if (rc != MRP_LL_SUCCESS) {
          for (int32_t sl : P.outSlot) slotPool.give(sl);
          P.chainFirst = -1;
          pendFree.push_back(pi);
          if (rc == MRP_LL_E_BUSY) return 0;
          out.err = std::string("mrp_ll_submit (root chain): ") + mrp_ll_last_error(ctx);
          return -1;
        }
        if (static_cast<size_t>(ticket) >= ticketPend.size()) ticketPend.resize(ticket + 1, -1);
        ticketPend[ticket] = pi;
        P.chainReq.assign(1, L.req[L.reqHead]);
        L.req.clear();
        L.reqHead = 0;
        ticketsOut += 1;
        jobsOut += 1;
        out.rounds += 1;
        return 1;
      }
    }
    size_t end = L.reqHead;
    while (end < L.req.size() && L.req[end].group == group) {
      mrp_ll_job j;
      poolOff.push_back(pathLenPool.size());
      idOff.push_back(idPool.size());
      bool ok = false;
      fillJob(I, L.req[end], j, pathLenPool, pathPtrPool, pathSlots > 0 ? &idPool : nullptr, &ok);
      // a root search that ended a chain outgrows the LDS tier: no second attempt there
      if (group == kRootGroup && L.req[end].agent == L.noChainAgent) j.flags |= MRP_LL_JOB_HEAVY;
      idOk.push_back(ok ? 1 : 0);
      jobs.push_back(j);
      ++end;
    }
    for (size_t q = 0; q < jobs.size(); ++q)
      if (jobs[q].n_agents > 0) {
        jobs[q].path_len = pathLenPool.data() + poolOff[q];
        jobs[q].path_xy = pathPtrPool.data() + poolOff[q];
        if (idOk[q]) jobs[q].path_ids = idPool.data() + idOff[q];  // every needed path is in the device store
      }
    int32_t pi;
    if (!pendFree.empty()) {
      pi = pendFree.back();
      pendFree.pop_back();
    } else {
      pend.emplace_back();
      pi = static_cast<int32_t>(pend.size()) - 1;
    }
    Pending& P = pend[pi];
    P.live = k;
    P.group = group;
    P.chainFirst = -1;
    P.outSlot.assign(jobs.size(), -1);
    if (pathSlots > 0)
      for (size_t q = 0; q < jobs.size(); ++q) {
        jobs[q].result_path_id = P.outSlot[q] = slotPool.take();  // -1: store full, later jobs ship this path as a table
        if (P.outSlot[q] >= 0) jobs[q].flags |= MRP_LL_JOB_STORE_RESULT;
      }
    P.res.assign(jobs.size(), mrp_ll_result());
    P.states.resize(jobs.size() * static_cast<size_t>(cap) * 3);
    for (size_t q = 0; q < jobs.size(); ++q) {
      std::memset(&P.res[q], 0, sizeof(mrp_ll_result));
      P.res[q].states_txy = P.states.data() + q * static_cast<size_t>(cap) * 3;
      P.res[q].states_cap = cap;
    }
    int32_t ticket = -1;
    int rc = llSubmit(static_cast<int32_t>(jobs.size()), jobs.data(), P.res.data(), &ticket);
    if (rc != MRP_LL_SUCCESS)
      for (int32_t sl : P.outSlot) slotPool.give(sl);
    if (rc == MRP_LL_E_BUSY) {
      pendFree.push_back(pi);
      return 0;
    }
    if (rc != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_submit: ") + mrp_ll_last_error(ctx);
      return -1;
    }
    if (static_cast<size_t>(ticket) >= ticketPend.size()) ticketPend.resize(ticket + 1, -1);
    ticketPend[ticket] = pi;
    L.reqHead = end;
    if (L.reqHead == L.req.size()) {
      L.req.clear();
      L.reqHead = 0;
    }
    ticketsOut += 1;
    jobsOut += static_cast<int64_t>(jobs.size());
    out.rounds += 1;
    out.searches += static_cast<int64_t>(jobs.size());
    return 1;
  };
  // Scheduling.  The device queue is kept SHALLOW — at most `ringTarget` searches published per engine, enough to hand
  // every resident wavefront its next job the moment it finishes one — and everything else waits in a host-side
  // priority queue ordered by how many low-level expansions the instance has consumed so far.  An instance deep in its
  // conflict tree (or stuck with one huge search) is a long chain of dependent rounds; served first, each of its
  // rounds starts within one job time instead of queueing behind thousands of searches of easy instances, so the
  // chain costs its compute time and not rounds x queue length (that, not throughput, bounded a step before).  Fresh
  // instances have priority 0 and are admitted only when nothing older is waiting.
  int64_t ringTarget = std::max<int64_t>(2 * static_cast<int64_t>(myWorkgroups), 32);
  if (const char* e = std::getenv("MRP_HL_RING_DEPTH"))
    if (std::atoll(e) > 0) ringTarget = std::atoll(e);  // tuning knob
  typedef std::pair<int64_t, size_t> Waiting;  // (priority, live index)
  static const bool prioExp = std::getenv("MRP_HL_PRIO_EXPANSIONS") != nullptr;  // A/B: round 2's priority (expansions only)
  std::priority_queue<Waiting> backlog;
  auto enqueue = [&](size_t k) {
    Live& L = live[k];
    if (!L.queued) {
      L.queued = true;
      // priority: the work an instance has consumed so far, in searches — a long chain of tiny searches (a deadlocked
      // pair of agents grows its conflict tree by two 10-expansion searches per round, thousands of rounds deep) is as
      // latency-critical as one huge search, and its expansions alone would never say so
      backlog.push(Waiting(prioExp ? L.inst->llExpanded() : L.inst->llExpanded() / 64 + L.inst->llSearches(), k));
    }
  };
  // Submits groups of live[k] while the device queue has room; false on error.  Leaves it in the backlog if some remain.
  auto submitAll = [&](size_t k) -> bool {
    Live& L = live[k];
    while (L.reqHead < L.req.size()) {
      if (jobsOut >= ringTarget) {
        enqueue(k);
        return true;
      }
      int r = submitGroup(k);
      if (r < 0) return false;
      if (r == 0) {
        enqueue(k);
        return true;
      }
    }
    return true;
  };

  auto tg1 = std::chrono::steady_clock::now();
  std::vector<int32_t> doneTickets(64);  // small harvest chunks keep the latency of any one instance's chain low
  std::vector<int32_t> donePend;
  // Admission control (MRP_HL_ACTIVE_LIMIT): at most that many instances of this worker are active at a time; the rest
  // wait in the pool.  With the job slots recycled in completion order it costs nothing (measured 1536..3584 at the
  // bench shape: same step time as "everything at once"), and with a shared pool it is what lets the workers balance.
  size_t nextStatic = 0;
  size_t nActive = 0;
  bool exhausted = false;
  // shared pool: no worker may hold more than its fair share at a time, or a small batch is drained by the first few
  size_t activeLimit =
      shared ? std::max<size_t>(1, std::min<size_t>(16384, (static_cast<size_t>(nTotal) + nWorkers - 1) / nWorkers)) : n;
  if (const char* e = std::getenv("MRP_HL_ACTIVE_LIMIT")) activeLimit = std::max(1, std::atoi(e));
  auto admit = [&]() -> bool {  // next instance of the pool, false when it is empty
    int32_t k, mid;
    if (shared) {
      k = shared->fetch_add(1, std::memory_order_relaxed);
      if (k >= nTotal) {
        exhausted = true;
        return false;
      }
      mid = mapBase + k;
      if (view) {
        while (k >= view->first[viewBatch + 1]) ++viewBatch;
        mid = (*view->mapBase[viewBatch])[engineIdx] + (k - view->first[viewBatch]);
      }
    } else {
      if (nextStatic >= n) {
        exhausted = true;
        return false;
      }
      k = idx[nextStatic];
      mid = mapIds[nextStatic];
      nextStatic += 1;
    }
    live.emplace_back();
    gidx.push_back(k);
    live.back().inst.reset(new Instance(view ? view->inst[viewBatch][k - view->first[viewBatch]] : instIn[k], mid, opt));
    if (timing) live.back().tAdmit = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
    return true;
  };
  // look ahead only while the engine has idle wavefronts: speculative searches must not queue in front of real ones
  auto specNow = [&]() -> int32_t { return jobsOut < static_cast<int64_t>(myWorkgroups) ? specK : 1; };
  // A finished instance is written out and FREED here, inside the loop, where the host has slack and the device is busy:
  // the paths, constraint sets and heaps of 16 384 instances are ~1e6 heap blocks per worker, and freeing them after
  // the loop was 70-130 ms of a 930 ms step with the GPU idle (measured, MRP_HL_TIMING).  Searches of the instance that
  // are still in flight (look-ahead) find `inst` empty when they return and are dropped.
  auto retire = [&](size_t k) {
    Live& L = live[k];
    if (!L.counted && L.inst->done()) {
      L.counted = true;
      nActive -= 1;
      L.req.clear();  // requests of a finished instance point into freed CT nodes
      L.reqHead = 0;
      if (timing) L.tDone = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
      writeSolution(*L.inst, solOf(gidx[k]));
      L.hl = L.inst->hlExpanded();
      L.ll = L.inst->llExpanded();
      L.spec = L.inst->specSearches();
      L.searches = L.inst->llSearches();
      out.expansions += L.ll;
      out.specSearches += L.spec;
      L.inst.reset();
    }
  };
  bool failed = false;
  auto t0 = std::chrono::steady_clock::now();
  auto tg2 = t0;
  uint64_t idleSpins = 0;
  bool sinceProgress = false;
  auto lastProgress = t0;
  double tmSubmit = 0, tmPollEmpty = 0, tmPollHit = 0, tmUnpack = 0, tmAdvance = 0;
  uint64_t nPollEmpty = 0, nPollHit = 0;
  auto nowS = []() { return std::chrono::steady_clock::now(); };
  auto secsS = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  while (!failed && (ticketsOut != 0 || !backlog.empty() || !exhausted)) {
    bool progress = false;
    auto tA = nowS();
    // publish waiting searches, deepest instance first, while the device queue has room
    while (!backlog.empty() && jobsOut < ringTarget) {
      const size_t k = backlog.top().second;
      backlog.pop();
      live[k].queued = false;
      const size_t before = ticketsOut;
      if (!submitAll(k)) {
        failed = true;
        break;
      }
      if (ticketsOut != before) progress = true;
      if (live[k].queued) break;  // the ring itself is full
    }
    if (failed) break;
    // nothing older is waiting: start fresh instances
    while (!exhausted && backlog.empty() && jobsOut < ringTarget && nActive < activeLimit && admit()) {
      const size_t k = live.size() - 1;
      Live& L = live[k];
      nActive += 1;
      L.inst->setSpecWidth(specNow());
      L.inst->start(L.req);
      retire(k);
      if (!submitAll(k)) failed = true;
      progress = true;
      if (failed) break;
    }
    if (failed) break;
    auto tB = nowS();
    tmSubmit += secsS(tA, tB);
    // harvest: one pass over the ring's completion words, whatever the number of instances in flight
    int32_t nDone = 0;
    if ((shared2 ? mrp_ll_poll_any_tagged(ctx, coIndex, doneTickets.data(), static_cast<int32_t>(doneTickets.size()), &nDone)
                 : mrp_ll_poll_any(ctx, doneTickets.data(), static_cast<int32_t>(doneTickets.size()), &nDone)) != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_poll_any: ") + mrp_ll_last_error(ctx);
      failed = true;
      break;
    }
    auto tC = nowS();
    if (nDone) {
      tmPollHit += secsS(tB, tC);
      nPollHit += 1;
    } else {
      tmPollEmpty += secsS(tB, tC);
      nPollEmpty += 1;
    }
    // resolve the owners first: a ticket id freed by this harvest can be handed out again by a resubmission below
    donePend.resize(nDone);
    for (int32_t d = 0; d < nDone; ++d) donePend[d] = ticketPend[doneTickets[d]];
    for (int32_t d = 0; d < nDone; ++d) {
      Pending& P = pend[donePend[d]];
      const size_t k = P.live;
      Live& L = live[k];
      progress = true;
      ticketsOut -= 1;
      jobsOut -= static_cast<int64_t>(P.res.size());
      if (P.chainFirst >= 0) {  // a root chain: its answers are delivered one by one, exactly like ten separate jobs
        if (P.res[0].status == MRP_LL_BAD_JOB) {
          // chains are unavailable on this engine (mrp_ll.h MRP_LL_JOB_ROOT_CHAIN: needs the compact tier and room for the
          // focal table): nothing ran; the request goes out again as an ordinary job, and so does every later root search
          rootChains = false;
          for (int32_t sl : P.outSlot) slotPool.give(sl);
          if (L.inst) {
            L.req = P.chainReq;
            L.reqHead = 0;
          }
          P.chainFirst = -1;
          P.chainReq.clear();
          pendFree.push_back(donePend[d]);
          if (L.inst && L.reqHead < L.req.size() && !L.queued && !submitAll(k)) {
            failed = true;
            break;
          }
          continue;
        }
        if (P.res[0].status != MRP_LL_OK) {
          out.err = "root chain failed on the engine (status " + std::to_string(P.res[0].status) + ")";
          failed = true;
          break;
        }
        const size_t cnt = P.chainRes.size();
        // The chain planned every agent and its workgroup found no conflict among the paths: the root node is the
        // solution.  Seven ten-agent instances in ten end here, without a path object, a conflict-tree node or a scan.
        if (rootFastPath && P.chainFirst == 0 && L.inst && !L.counted && static_cast<size_t>(P.res[0].n_states) == cnt &&
            P.res[0].cost == 0 && static_cast<int32_t>(cnt) == L.inst->nAgents() && L.inst->llSearches() == 0 &&
            (opt.max_hl_expansions < 0 || opt.max_hl_expansions >= 1)) {
          writeRootSolution(P.chainRes, solOf(gidx[k]));
          int64_t ll = 0;
          for (const mrp_ll_result& r : P.chainRes) ll += r.expanded;
          ranExpansions += ll;
          out.searches += static_cast<int64_t>(cnt);
          out.expansions += ll;
          out.rootSolved += 1;
          L.counted = true;
          nActive -= 1;
          L.hl = 1;
          L.ll = ll;
          L.searches = static_cast<int32_t>(cnt);
          if (timing) L.tDone = std::chrono::duration<double>(std::chrono::steady_clock::now() - tg0).count();
          L.req.clear();
          L.reqHead = 0;
          L.inst.reset();
          for (int32_t sl : P.outSlot) slotPool.give(sl);
          P.chainFirst = -1;
          P.chainReq.clear();
          pendFree.push_back(donePend[d]);
          continue;
        }
        size_t q = 0;
        for (; q < cnt; ++q) {
          const mrp_ll_result& r = P.chainRes[q];
          if (chainDebug)
            std::fprintf(stderr, "[chain] inst %d agent %d: status %d cost %d fmin %d n %d expanded %lld\n", gidx[k],
                         P.chainFirst + static_cast<int>(q), r.status, r.cost, r.fmin, r.n_states, (long long)r.expanded);
          if (r.status == MRP_LL_NOT_RUN || !L.inst) break;
          ranExpansions += r.expanded;
          out.searches += 1;
          ans.clear();
          ans.push_back(answerOf(r, P.outSlot[q], &slotPool));
          L.req.clear();  // (the request for the next root agent, which the chain has already answered — or not, below)
          L.reqHead = 0;
          L.inst->setSpecWidth(specNow());
          L.inst->deliver(P.group, ans, L.req);
          ans.clear();
          retire(k);
        }
        if (L.inst && q < cnt && static_cast<int32_t>(q) < P.chainCount && P.chainRes[q].status == MRP_LL_NOT_RUN) {
          // the search of this agent did not fit the compact tier: it goes as an ordinary job (any tier), chains resume behind it
          L.noChainAgent = P.chainFirst + static_cast<int32_t>(q);
          if (q == 0) {  // nothing was delivered, so nothing re-created the request
            L.req = P.chainReq;
            L.reqHead = 0;
          }
        }
        for (; q < cnt; ++q) slotPool.give(P.outSlot[q]);  // agents the chain did not reach
        P.chainFirst = -1;
        P.chainReq.clear();
        pendFree.push_back(donePend[d]);
        if (L.inst && L.reqHead < L.req.size() && !L.queued && !submitAll(k)) {
          failed = true;
          break;
        }
        continue;
      }
      auto tu0 = timing ? nowS() : tC;  // (per-ticket clock reads only when somebody will look at them)
      ans.clear();
      for (size_t q = 0; q < P.res.size(); ++q) {
        ranExpansions += P.res[q].expanded;
        ans.push_back(answerOf(P.res[q], P.outSlot[q], &slotPool));
      }
      const int32_t group = P.group;
      pendFree.push_back(donePend[d]);
      auto tu1 = timing ? nowS() : tC;
      if (L.inst) {  // (else: a pre-computed expansion that came back after its instance had finished)
        L.inst->setSpecWidth(specNow());
        L.inst->deliver(group, ans, L.req);
        retire(k);
      }
      ans.clear();   // the paths nobody took go back to the slot pool now
      auto tu2 = timing ? nowS() : tC;
      tmUnpack += secsS(tu0, tu1);
      tmAdvance += secsS(tu1, tu2);
      // publish the follow-up searches at once: a long conflict-tree chain must not wait for the rest of this pass
      if (L.reqHead < L.req.size() && !L.queued && !submitAll(k)) {
        failed = true;
        break;
      }
    }
    if (failed) break;
    auto tD = nowS();
    out.buildS += secsS(tA, tB);
    out.llS += secsS(tB, tC);
    out.consumeS += secsS(tC, tD);
    if (progress) {  // the guard measures the time since the LAST progress, not since the start of the batch
      idleSpins = 0;
      sinceProgress = false;
    } else if ((++idleSpins & 0xFFFF) == 0) {
      const auto nowT = std::chrono::steady_clock::now();
      if (!sinceProgress) {
        sinceProgress = true;
        lastProgress = nowT;
      } else if (std::chrono::duration<double>(nowT - lastProgress).count() > kNoProgressLimitS) {
        out.err = "session: no progress for too long";
        failed = true;
      }
    }
  }
  auto tg3 = std::chrono::steady_clock::now();
  if (shared2) {  // the leader ends the session when every co-worker has left its loop
    co->finished.fetch_add(1, std::memory_order_acq_rel);
    if (coIndex == 0)
      while (co->finished.load(std::memory_order_acquire) < coCount) std::this_thread::yield();
  }
  if ((!shared2 || coIndex == 0) && mrp_ll_session_end(ctx) != MRP_LL_SUCCESS && out.err.empty())
    out.err = std::string("mrp_ll_session_end: ") + mrp_ll_last_error(ctx);
  auto tg4 = std::chrono::steady_clock::now();
  if (timing && (!shared2 || coIndex == 0)) {
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
      return std::chrono::duration<double, std::milli>(b - a).count();
    };
    mrp_ll_stats ls;
    mrp_ll_get_stats(ctx, &ls);
    std::fprintf(stderr, "[mrp_hl] group of %zu: session_begin %.2f ms, build instances %.2f ms, loop %.2f ms, session_end %.2f ms; "
                 "cumulative: active wgs %lld, busy %.0f ms, idle %.0f ms, heavy wgs %lld busy %.0f ms idle %.0f ms, searches %lld, "
                 "expansions %lld\n", live.size(),
                 ms(tg0, tg1), ms(tg1, tg2), ms(tg2, tg3), ms(tg3, tg4), (long long)ls.session_active_wgs,
                 ls.session_busy_ms, ls.session_idle_ms, (long long)ls.heavy_active_wgs, ls.heavy_busy_ms, ls.heavy_idle_ms,
                 (long long)ls.jobs, (long long)ls.expansions);
    std::fprintf(stderr, "[mrp_hl]   host ms: admit+submit %.1f, poll empty %.1f (%llu), poll hit %.1f (%llu), "
                 "unpack %.1f, advance %.1f; tickets %lld searches %lld\n", tmSubmit * 1e3, tmPollEmpty * 1e3,
                 (unsigned long long)nPollEmpty, tmPollHit * 1e3, (unsigned long long)nPollHit, tmUnpack * 1e3,
                 tmAdvance * 1e3, (long long)out.rounds, (long long)out.searches);
    std::vector<size_t> order(live.size());  // the instances this worker finished last
    for (size_t k = 0; k < order.size(); ++k) order[k] = k;
    std::sort(order.begin(), order.end(), [&](size_t a, size_t b) { return live[a].tDone > live[b].tDone; });
    for (size_t q = 0; q < std::min<size_t>(3, order.size()); ++q) {
      const Live& L = live[order[q]];
      std::fprintf(stderr, "[mrp_hl]     last #%zu: instance %d admitted %.1f ms done %.1f ms, HL %lld, LL %lld, searches %d (+%lld ahead)\n", q,
                   gidx[order[q]], L.tAdmit * 1e3, L.tDone * 1e3, (long long)L.hl, (long long)L.ll, L.searches,
                   (long long)L.spec);
    }
  }
  if (failed) return;
  for (size_t k = 0; k < live.size(); ++k)
    if (live[k].inst) {  // (none: the loop ends when every instance has been retired)
      writeSolution(*live[k].inst, solOf(gidx[k]));
      out.expansions += live[k].inst->llExpanded();
      out.specSearches += live[k].inst->specSearches();
    }
  out.specWasted += ranExpansions - out.expansions;
  if (timing) {
    auto tg5 = std::chrono::steady_clock::now();
    live.clear();
    auto tg6 = std::chrono::steady_clock::now();
    std::fprintf(stderr, "[mrp_hl]   thread began %.1f ms after the batch, loop ended at %.1f ms, solutions written at %.1f ms, "
                 "instances freed at %.1f ms\n", std::chrono::duration<double, std::milli>(tg0 - batchEpoch()).count(),
                 std::chrono::duration<double, std::milli>(tg3 - batchEpoch()).count(),
                 std::chrono::duration<double, std::milli>(tg5 - batchEpoch()).count(),
                 std::chrono::duration<double, std::milli>(tg6 - batchEpoch()).count());
  }
}

}  // namespace

extern "C" {

int mrp_hl_solver_create(int32_t device, int32_t nThreads, const mrp_ll_options* llOpt, mrp_hl_solver** out) {
  if (!out) return MRP_LL_E_INVALID;
  *out = nullptr;
  if (nThreads <= 0) {
    unsigned hc = std::thread::hardware_concurrency();
    nThreads = static_cast<int32_t>(hc ? std::min<unsigned>(hc, 16) : 8);
  }
  // Engines: at most eight.  An engine of an ECBS batch keeps two resident kernels (front + heavy workgroups), and the
  // device time-slices a process's hardware queues — idle ones included — beyond about twenty (measured: the LDS tier's
  // 2.0 us per expansion becomes 2.5 with 24 streams in the process, 3.1 with 32).  Worker threads beyond the engines
  // share them two by two (co-workers, mrp_ll.h mrp_ll_submit_tagged): host cores feed conflict trees, engines feed the
  // device.
  int32_t maxEngines = 8;
  if (const char* e = std::getenv("MRP_HL_MAX_ENGINES")) maxEngines = std::max(1, std::atoi(e));
  const int32_t nWorkers = nThreads;
  nThreads = std::min(nThreads, maxEngines);
  // One HIP stream (= one resident kernel) per worker: each needs its own hardware queue, the ROCm default is 4.  Only
  // effective if the HIP runtime has not been initialised yet in this process; never overrides the caller.
  (void)setenv("GPU_MAX_HW_QUEUES", "64", 0);
  auto* s = new mrp_hl_solver();
  s->device = device;
  s->nWorkers = nWorkers;
  std::memset(&s->llOpt, 0, sizeof(s->llOpt));
  if (llOpt) s->llOpt = *llOpt;
  s->llOpt.device = device;
  // One batch in flight per engine.  A second ticket (MRP_HL_TICKETS=2: the round-based prioritized-SIPP schedule then
  // pipelines two half-batches) means a second arena per engine, and with it allocated the resident A*-epsilon searches
  // ran 7 % slower (busy workgroup time 731 s -> 788 s per 65 536 instances, same kernel) — so it is opt-in.
  if (s->llOpt.n_tickets <= 0) s->llOpt.n_tickets = 1;
  if (const char* e = std::getenv("MRP_HL_TICKETS")) s->llOpt.n_tickets = std::max(1, std::atoi(e));
  if (s->llOpt.slots <= 0) s->llOpt.slots = 512;
  if (s->llOpt.arena_nodes <= 0) s->llOpt.arena_nodes = 65536;
  for (int32_t t = 0; t < nThreads; ++t) {
    mrp_ll_ctx* ctx = nullptr;
    int rc = mrp_ll_create(&s->llOpt, &ctx);
    if (rc != MRP_LL_SUCCESS) {  // no GPU, no solver: there is no CPU path
      for (auto* e : s->engines) mrp_ll_destroy(e);
      delete s;
      return rc;
    }
    s->engines.push_back(ctx);
  }
  *out = s;
  return MRP_LL_SUCCESS;
}

void mrp_hl_solver_destroy(mrp_hl_solver* s) {
  if (!s) return;
  for (auto* e : s->engines) mrp_ll_destroy(e);
  delete s;
}

const char* mrp_hl_solver_last_error(const mrp_hl_solver* s) { return s ? s->err.c_str() : "null solver"; }

int mrp_hl_solver_ll_stats(mrp_hl_solver* s, mrp_ll_stats* out, int32_t reset) {
  if (!s || !out) return MRP_LL_E_INVALID;
  std::memset(out, 0, sizeof(*out));
  for (auto* e : s->engines) {
    mrp_ll_stats st;
    mrp_ll_get_stats(e, &st);
    out->launches += st.launches;
    out->jobs += st.jobs;
    out->expansions += st.expansions;
    out->nodes_created += st.nodes_created;
    out->migrated += st.migrated;
    out->kernel_ms += st.kernel_ms;
    out->h2d_ms += st.h2d_ms;
    out->d2h_ms += st.d2h_ms;
    out->pack_ms += st.pack_ms;
    out->staged_bytes += st.staged_bytes;
    out->session_busy_ms += st.session_busy_ms;
    out->session_idle_ms += st.session_idle_ms;
    out->session_active_wgs += st.session_active_wgs;
    out->unpack_ms += st.unpack_ms;
    out->heavy_busy_ms += st.heavy_busy_ms;
    out->heavy_idle_ms += st.heavy_idle_ms;
    out->heavy_active_wgs += st.heavy_active_wgs;
    out->heavy_fallbacks += st.heavy_fallbacks;
    for (int q = 0; q < 8; ++q) out->prof[q] += st.prof[q];
    if (reset) mrp_ll_reset_stats(e);
  }
  return MRP_LL_SUCCESS;
}

int mrp_hl_solver_preload(mrp_hl_solver* s, int32_t nThreadsWanted, int32_t nInst, const mrp_hl_instance* instances,
                          mrp_hl_preloaded** out) {
  if (!s || !out || nInst < 0 || (nInst > 0 && !instances)) return MRP_LL_E_INVALID;
  *out = nullptr;
  int32_t nThreads = static_cast<int32_t>(s->engines.size());
  if (nThreadsWanted > 0) nThreads = std::min(nThreads, nThreadsWanted);
  const int32_t nEng = std::max(1, nThreads);  // engines that receive the maps: all the caller allows, however small the batch
  nThreads = std::max(1, std::min(nThreads, std::max(nInst, 1)));
  auto* p = new mrp_hl_preloaded();
  p->owner = s;
  p->nInst = nInst;
  p->instances = instances;
  p->idx.resize(nThreads);
  p->mapIds.resize(nThreads);
  // Every engine receives every map (a 32x32 bitmap is 128 bytes), so any worker can run any instance: the session
  // driver lets the workers draw instances from one pool.  idx / mapIds keep the interleaved static split (instance k ->
  // thread k % nThreads) for the round-based schedule.
  for (int32_t k = 0; k < nInst; ++k) p->idx[k % nThreads].push_back(k);
  p->mapBase.assign(nEng, 0);
  std::vector<int> rcs(nEng, MRP_LL_SUCCESS);
  {
    std::vector<std::thread> th;
    for (int32_t t = 0; t < nEng; ++t)
      th.emplace_back([&, t]() {
        for (int32_t k = 0; k < nInst; ++k) {
          const mrp_hl_instance& in = instances[k];
          int32_t mid = -1;
          int rc = mrp_ll_upload_map(s->engines[t], in.dimx, in.dimy, in.n_obstacles, in.obstacles_xy, &mid);
          if (rc != MRP_LL_SUCCESS) {
            rcs[t] = rc;
            return;
          }
          if (k == 0) p->mapBase[t] = mid;
        }
        rcs[t] = mrp_ll_sync_maps(s->engines[t]);  // push the bitmaps to the device now, not at the first launch
      });
    for (auto& x : th) x.join();
  }
  for (int32_t t = 0; t < nEng; ++t) {
    if (rcs[t] != MRP_LL_SUCCESS) {
      s->err = std::string("mrp_ll_upload_map / mrp_ll_sync_maps: ") + mrp_ll_last_error(s->engines[t]);
      delete p;
      return rcs[t];
    }
    if (t < nThreads)
      for (int32_t k : p->idx[t]) p->mapIds[t].push_back(p->mapBase[t] + k);
  }
  s->nPreloaded += 1;
  *out = p;
  return MRP_LL_SUCCESS;
}

// The engines keep one map buffer for all preloaded batches alive at a time; when the last one goes, its maps go too
// (otherwise a persistent solver would grow its host and device map buffers with every batch).
void mrp_hl_preloaded_free(mrp_hl_preloaded* p) {
  if (!p) return;
  if (p->owner && --p->owner->nPreloaded == 0)
    for (auto* e : p->owner->engines) (void)mrp_ll_release_maps(e);
  delete p;
}

int mrp_hl_solver_solve(mrp_hl_solver* s, const mrp_hl_options* optIn, int32_t nInst, const mrp_hl_instance* instances,
                        mrp_hl_solution* solutions, mrp_hl_batch_stats* stats) {
  if (!s || !optIn || nInst < 0 || (nInst > 0 && (!instances || !solutions))) return MRP_LL_E_INVALID;
  mrp_hl_preloaded* p = nullptr;
  int rc = mrp_hl_solver_preload(s, optIn->n_threads, nInst, instances, &p);
  if (rc != MRP_LL_SUCCESS) return rc;
  rc = mrp_hl_solver_solve_preloaded(s, optIn, p, solutions, stats);
  mrp_hl_preloaded_free(p);
  return rc;
}

int mrp_hl_solver_solve_preloaded(mrp_hl_solver* s, const mrp_hl_options* optIn, mrp_hl_preloaded* pre,
                                  mrp_hl_solution* solutions, mrp_hl_batch_stats* stats) {
  return mrp_hl_solver_solve_stream(s, optIn, 1, &pre, &solutions, stats);
}

// mrp_hl.h: n batches as one pool of instances.  One batch: every schedule of the drivers (rounds, static split, shared
// pool); several: the shared pool of the session driver (the others keep a barrier between batches by construction).
int mrp_hl_solver_solve_stream(mrp_hl_solver* s, const mrp_hl_options* optIn, int32_t nBatches, mrp_hl_preloaded* const* pres,
                               mrp_hl_solution* const* solsArr, mrp_hl_batch_stats* stats) {
  if (!s || !optIn || nBatches < 1 || !pres || !solsArr) return MRP_LL_E_INVALID;
  for (int32_t b = 0; b < nBatches; ++b)
    if (!pres[b] || pres[b]->owner != s || (pres[b]->nInst > 0 && !solsArr[b])) return MRP_LL_E_INVALID;
  mrp_hl_options opt = *optIn;
  mrp_hl_preloaded* pre = pres[0];
  mrp_hl_solution* solutions = solsArr[0];
  StreamView view;
  int64_t total = 0;
  for (int32_t b = 0; b < nBatches; ++b) {
    view.first.push_back(static_cast<int32_t>(total));
    view.inst.push_back(pres[b]->instances);
    view.sols.push_back(solsArr[b]);
    view.mapBase.push_back(&pres[b]->mapBase);
    total += pres[b]->nInst;
    if (total > INT32_MAX) {  // (checked before the next prefix is narrowed)
      s->err = "mrp_hl_solver_solve_stream: more than 2^31 - 1 instances in one stream";
      return MRP_LL_E_INVALID;
    }
  }
  view.first.push_back(static_cast<int32_t>(total));
  const bool streamed = nBatches > 1;
  if (streamed && (opt.mode == 1 || std::getenv("MRP_HL_STATIC_SPLIT") != nullptr)) {
    s->err = "mrp_hl_solver_solve_stream: several batches need the session driver's shared pool (mode 0, no MRP_HL_STATIC_SPLIT)";
    return MRP_LL_E_INVALID;
  }
  const int32_t nInst = static_cast<int32_t>(total);
  const mrp_hl_instance* instances = pre->instances;
  int32_t nThreads = static_cast<int32_t>(pre->idx.size());
  if (streamed) {  // engines that hold every batch's maps, and not more of them than there are instances
    nThreads = static_cast<int32_t>(pre->mapBase.size());
    for (int32_t b = 1; b < nBatches; ++b) nThreads = std::min(nThreads, static_cast<int32_t>(pres[b]->mapBase.size()));
    nThreads = std::max(1, std::min(nThreads, std::max(nInst, 1)));
  }
  const int32_t horizon = s->llOpt.max_horizon > 0 ? s->llOpt.max_horizon : 512;
  std::vector<std::vector<int32_t>>& idx = pre->idx;
  std::vector<std::vector<int32_t>>& mapIds = pre->mapIds;
  std::vector<GroupResult> gr(nThreads);
  // LDS tier sized for two resident searches per SIMD (8 per CU): one wavefront alone leaves about half of its SIMD's
  // issue slots idle (waiting on LDS / memory), a second one fills them.  The focal path table of a search is
  // [time][agents rounded up to 16] halfwords; 64 time steps of it are kept in LDS, the rest of a longer table lives in
  // the search's arena slot.
  int32_t maxAgents = 1;
  for (int32_t b = 0; b < nBatches; ++b)
    for (int32_t k = 0; k < pres[b]->nInst; ++k) maxAgents = std::max(maxAgents, pres[b]->instances[k].n_agents);
  const int32_t agentsPad = (maxAgents + 15) & ~15;
  int32_t occupancy = 4;
  if (s->llOpt.lds_nodes == 0 && opt.mode != 1) {  // the caller did not choose a geometry: pick one for this batch
    int32_t pathBytes = opt.algo == MRP_HL_ECBS ? std::min(16384, std::max(2048, agentsPad * 2 * 64)) : 32;
    int32_t tierNodes = 2048, tierRows = 64;  // the compact tier at its full size (open list: tierNodes / 2 entries)
    if (const char* e = std::getenv("MRP_HL_TIER")) {  // tuning knob: "nodes,rows,pathBytes"
      int a = 0, b = 0, c = 0;
      if (std::sscanf(e, "%d,%d,%d", &a, &b, &c) == 3) {
        tierNodes = a;
        tierRows = b;
        if (opt.algo == MRP_HL_ECBS) pathBytes = c;
      }
    }
    for (int32_t t = 0; t < nThreads; ++t)
      if (mrp_ll_configure_tiers(s->engines[t], tierNodes, tierRows, pathBytes, &occupancy) != MRP_LL_SUCCESS) {
        s->err = std::string("mrp_ll_configure_tiers: ") + mrp_ll_last_error(s->engines[t]);
        return MRP_LL_E_DEVICE;
      }
  }
  if (opt.mode != 1 && !s->engines.empty()) {  // what the runtime grants the kernel family of this batch's sessions
    int32_t occ = 0;
    if (mrp_ll_session_occupancy(s->engines[0], opt.algo == MRP_HL_ECBS ? MRP_LL_ASTAR_EPS : MRP_LL_ASTAR, &occ) == MRP_LL_SUCCESS && occ > 0)
      occupancy = occ;
  }
  // resident wavefronts per engine: the chip holds 256 CUs x `occupancy` workgroups of this kernel at once
  int32_t sessionWgs = std::max(16, std::min<int32_t>(s->llOpt.slots, (256 * occupancy) / nThreads));
  // ECBS: some of the device's LDS goes to heavy workgroups (wide LDS tier + arena tier: the searches that outgrow the
  // front tier — 6 % of the expansions at ten agents, 17 % at a hundred, scripts/search_stats.py).  One takes the LDS of
  // `displaced` front workgroups; five eighths / three quarters / all of the CUs get one.
  int32_t heavyPer = 0;
  const int32_t nRun = nThreads;  // engines of this batch
  const bool sharedPoolMode = std::getenv("MRP_HL_STATIC_SPLIT") == nullptr;
  if (opt.algo == MRP_HL_ECBS && opt.mode != 1 && !s->engines.empty() && s->llOpt.lds_nodes >= 0) {
    int32_t frontOcc = 0, frontLds = 0, heavyLds = 0;
    if (mrp_ll_session_tiers_geometry(s->engines[0], &frontOcc, &frontLds, &heavyLds) == MRP_LL_SUCCESS && frontOcc > 0 &&
        frontLds > 0) {
      // (measured, eight workers: ten agents 160 > 192 > 128 > 256; fifty 192 > 256 > 128; a hundred 256)
      int32_t heavyTotal = maxAgents <= 16 ? 160 : maxAgents <= 64 ? 192 : 256;
      if (const char* e = std::getenv("MRP_HL_HEAVY_WGS")) heavyTotal = std::max(0, std::atoi(e));  // tuning knob (0: one launch)
      heavyPer = heavyTotal / nRun + (heavyTotal % nRun ? 1 : 0);
      if (heavyPer > 0) {
        const int32_t granule = 512;  // LDS allocation granularity
        // (a window beyond 32 KB takes room as if it had 64 KB — measured, ll_compact.h MRP_CT_WIDE_GROUPS)
        const int32_t fl = (frontLds + granule - 1) / granule * granule;
        const int32_t hl = heavyLds > 32768 ? 65536 : (heavyLds + granule - 1) / granule * granule;
        const int32_t cuLds = 160 * 1024;
        const int32_t frontBeside = std::max(0, (cuLds - hl) / fl);          // front workgroups on a CU that hosts a heavy one
        // (beyond five heavy workgroups per eight CUs some CUs get two, and the second one costs more: measured, 192 / 224 /
        // 256 of them displace 3.6 front workgroups each, 160 exactly 3)
        const int32_t displaced = std::max(0, std::min(frontOcc, cuLds / fl) - frontBeside) + (heavyPer * nRun > 160 ? 1 : 0);
        const int32_t frontTotal = 256 * std::min(frontOcc, cuLds / fl) - displaced * heavyPer * nRun;
        // (never more than fits: a grid the device cannot place completely blocks its hardware pipe, and a launch queued
        // behind it — another worker's front workgroups — does not start until a resident kernel ends)
        sessionWgs = std::max(16, std::min<int32_t>(s->llOpt.slots - heavyPer, frontTotal / nRun - 2));
        if (sessionWgs + heavyPer > s->llOpt.slots || frontTotal <= 0) heavyPer = 0;  // (tiny engines: one launch serves all)
      }
    }
  }
  if (const char* e = std::getenv("MRP_HL_SESSION_WGS")) sessionWgs = std::max(1, std::atoi(e));  // tuning knob
  // f2: the engines' device-resident path stores (ECBS only: CBS's low level has no focal context).  A search leaves its
  // path there, and later jobs name the paths of their CT node by slot instead of shipping a [t][agent] table.
  // Used for conflict trees of up to 128 agents (MRP_HL_STORE_MAX_AGENTS; beyond that a table row no longer fits the
  // kernel's two 64-lane row loads and the job ships its table): bytes staged per search 1134 -> 134 (10 agents),
  // 5900 -> 303 (50), 12 087 -> 572 (100); steps 7 % / 3 % / 2 % faster.
  int32_t pathSlots = 0;
  int32_t storeMaxAgents = 128;
  if (const char* e = std::getenv("MRP_HL_STORE_MAX_AGENTS")) storeMaxAgents = std::atoi(e);
  if (opt.algo == MRP_HL_ECBS && opt.mode != 1 && maxAgents <= storeMaxAgents) {
    // every live conflict-tree node of every active instance holds one slot per replaced path: a worker with 16384
    // active instances (few threads, big batch) needs millions of them, and a slot is max_horizon halfwords
    pathSlots = std::max(1 << 18, std::min(1 << 22, (1 << 22) / std::max(nThreads, 1)));
    if (const char* e = std::getenv("MRP_HL_PATH_SLOTS")) pathSlots = std::max(0, std::atoi(e));  // 0 = ship tables (round 1)
    if (pathSlots != s->pathSlots) {
      for (int32_t t = 0; t < static_cast<int32_t>(s->engines.size()); ++t)
        if (mrp_ll_path_store_reserve(s->engines[t], pathSlots) != MRP_LL_SUCCESS) {
          pathSlots = 0;  // e.g. the CPU test build: fall back to tables everywhere
          for (int32_t u = 0; u <= t; ++u) (void)mrp_ll_path_store_reserve(s->engines[u], 0);
          break;
        }
      s->pathSlots = pathSlots;
    }
  }
  // one pool of instances for all workers (MRP_HL_STATIC_SPLIT=1 restores the fixed interleaved split)
  std::atomic<int32_t> nextInstance(0);
  const std::vector<int32_t> noIdx;  // (a stream has no static split)
  int32_t sessionGate = 0;  // mrp_ll_session_begin_tiers_gated: every worker's heavy launch before anybody's front launch
  const bool sharedPool = sharedPoolMode;
  auto t0 = std::chrono::steady_clock::now();
  batchEpoch() = t0;
  {
    // session mode with the shared pool: worker threads beyond the engines join them as co-workers, two per engine
    int32_t nWork = nRun;
    if (opt.mode != 1 && sharedPool) {
      int32_t want = s->nWorkers;
      if (const char* e = std::getenv("MRP_HL_WORKERS")) want = std::max(1, std::atoi(e));  // tuning knob
      nWork = std::max(nRun, std::min(want, 2 * nRun));
      nWork = std::min(nWork, std::max(nRun, nInst));
    }
    gr.resize(static_cast<size_t>(nWork));
    std::vector<CoSync> coSync(static_cast<size_t>(nRun));
    std::vector<std::thread> th;
    for (int32_t t = 0; t < nWork; ++t)
      th.emplace_back([&, t]() {
        pinWorker(t);
        const int32_t e = t % nRun, coIndex = t / nRun, coCount = 1 + (e + nRun < nWork ? 1 : 0);
        if (opt.mode == 1)
          runGroup(s->engines[t], opt, instances, solutions, idx[t], mapIds[t], horizon, gr[t]);
        else if (sharedPool)
          runGroupSession(s->engines[e], opt, instances, solutions, streamed ? noIdx : idx[e], streamed ? noIdx : mapIds[e],
                          horizon, sessionWgs, pathSlots,
                          gr[t], &nextInstance, nInst, pre->mapBase[e], nWork, heavyPer, &sessionGate, nRun, coIndex, coCount,
                          coCount > 1 ? &coSync[e] : nullptr, streamed ? &view : nullptr, e);
        else
          runGroupSession(s->engines[t], opt, instances, solutions, idx[t], mapIds[t], horizon, sessionWgs, pathSlots,
                          gr[t], nullptr, 0, 0, nRun, heavyPer, &sessionGate, nRun);
      });
    for (auto& x : th) x.join();
  }
  auto t1 = std::chrono::steady_clock::now();
  mrp_hl_batch_stats st;
  std::memset(&st, 0, sizeof(st));
  st.wall_seconds = std::chrono::duration<double>(t1 - t0).count();
  for (auto& g : gr) {
    if (!g.err.empty()) {
      s->err = g.err;
      return MRP_LL_E_DEVICE;
    }
    st.rounds += g.rounds;
    st.ll_searches += g.searches;
    st.ll_expansions += g.expansions;
    st.speculative_searches += g.specSearches;
    st.wasted_ll_expansions += g.specWasted;
    st.root_solved += g.rootSolved;
    st.build_seconds += g.buildS;
    st.ll_call_seconds += g.llS;
    st.consume_seconds += g.consumeS;
  }
  for (int32_t b = 0; b < nBatches; ++b)
    for (int32_t k = 0; k < pres[b]->nInst; ++k) st.solved += solsArr[b][k].status == MRP_HL_SOLVED ? 1 : 0;
  if (stats) *stats = st;
  return MRP_LL_SUCCESS;
}

namespace {

// The loop of mapf_prioritized_sipp.cpp:214-270 for instances idx[...] on one engine: round r plans agent r of every
// instance that still has one (agents of one instance are a chain — each plans against the intervals the earlier ones
// occupy — instances are independent).
// Test knob (MRP_HL_SIPP_MAX_EXPANSIONS): the expansion cap of every search of the prioritized-SIPP drivers, read per
// call; the reference has none (-1), and a capped search ends its instance with that status.
int64_t sippMaxExpansions() {
  const char* e = std::getenv("MRP_HL_SIPP_MAX_EXPANSIONS");
  return e && std::atoll(e) > 0 ? std::atoll(e) : -1;
}

void runSippGroup(mrp_ll_ctx* ctx, int32_t horizon, int32_t nTickets, const mrp_hl_instance* instances,
                  mrp_hl_sipp_solution* sols, const std::vector<int32_t>& idx, GroupResult& out) {
  struct Iv { int32_t s, e; };
  struct Prio {
    int32_t mapId = -1, agent = 0, dimx = 0;
    // allCollisionIntervals (mapf_prioritized_sipp.cpp:215): the reference keeps a std::map keyed by location and hands
    // every entry to setCollisionIntervals, whose result does not depend on the order of the locations; here: one list
    // per cell plus the cells that have one, in first-touch order
    std::vector<std::vector<Iv>> perCell;
    std::vector<int32_t> touched;
    std::vector<int32_t> xy, cnt, ivs;  // flattened for the job of the current round
    void add(int32_t x, int32_t y, Iv iv) {
      std::vector<Iv>& v = perCell[static_cast<size_t>(y) * dimx + x];
      if (v.empty()) touched.push_back(y * dimx + x);
      v.push_back(iv);
    }
  };
  const size_t n = idx.size();
  std::vector<Prio> st(n);
  for (size_t q = 0; q < n; ++q) {
    const mrp_hl_instance& in = instances[idx[q]];
    int rc = mrp_ll_upload_map(ctx, in.dimx, in.dimy, in.n_obstacles, in.obstacles_xy, &st[q].mapId);
    if (rc != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_upload_map: ") + mrp_ll_last_error(ctx);
      return;
    }
    sols[idx[q]].cost = 0;
    sols[idx[q]].low_level_expanded = 0;
    sols[idx[q]].n_planned = 0;
    sols[idx[q]].status = 0;
    st[q].dimx = in.dimx;
    st[q].perCell.assign(static_cast<size_t>(std::max(in.dimx, 0)) * std::max(in.dimy, 0), std::vector<Iv>());
  }
  const int32_t cap = std::max(horizon, 64);
  // Two halves of the instances take turns: while the searches of one half run on the GPU, the host consumes the
  // results of the other half and packs its next round (a round of a half still ends with its slowest search).
  struct Half {
    std::vector<size_t> members;
    std::vector<mrp_ll_job> jobs;
    std::vector<mrp_ll_result> results;
    std::vector<int32_t> owner, statesPool;
    int32_t ticket = -1;
    bool inflight = false;
  };
  Half half[2];
  for (size_t q = 0; q < n; ++q) half[(n >= 64 && nTickets >= 2) ? (q & 1) : 0].members.push_back(q);
  int64_t roundsOf[2] = {0, 0};

  // round of one half: agent p.agent of every member that still has one; returns false when there was nothing to do
  auto launch = [&](Half& H) -> bool {
    H.jobs.clear();
    H.owner.clear();
    for (size_t q : H.members) {
      Prio& p = st[q];
      const mrp_hl_instance& in = instances[idx[q]];
      if (p.agent >= in.n_agents) continue;
      p.xy.clear();
      p.cnt.clear();
      p.ivs.clear();
      for (int32_t cell : p.touched) {  // sipp.setCollisionIntervals(location, intervals) for every location (:224-226)
        const std::vector<Iv>& v = p.perCell[cell];
        p.xy.push_back(cell % p.dimx);
        p.xy.push_back(cell / p.dimx);
        p.cnt.push_back(static_cast<int32_t>(v.size()));
        for (const Iv& iv : v) {
          p.ivs.push_back(iv.s);
          p.ivs.push_back(iv.e);
        }
      }
      mrp_ll_job j;
      std::memset(&j, 0, sizeof(j));
      j.map_id = p.mapId;
      j.algo = MRP_LL_SIPP;
      j.w = 1.0f;
      j.start_x = in.starts_xy[2 * p.agent];
      j.start_y = in.starts_xy[2 * p.agent + 1];
      j.goal_x = in.goals_xy[2 * p.agent];
      j.goal_y = in.goals_xy[2 * p.agent + 1];
      j.max_expansions = sippMaxExpansions();
      j.n_collision_locations = static_cast<int32_t>(p.cnt.size());
      j.collision_xy = p.xy.data();
      j.collision_count = p.cnt.data();
      j.collision_intervals = p.ivs.data();
      H.jobs.push_back(j);
      H.owner.push_back(static_cast<int32_t>(q));
    }
    if (H.jobs.empty()) return false;
    H.results.assign(H.jobs.size(), mrp_ll_result());
    H.statesPool.resize(H.jobs.size() * static_cast<size_t>(cap) * 3);
    for (size_t q = 0; q < H.jobs.size(); ++q) {
      std::memset(&H.results[q], 0, sizeof(mrp_ll_result));
      H.results[q].states_txy = H.statesPool.data() + q * static_cast<size_t>(cap) * 3;
      H.results[q].states_cap = cap;
    }
    int rc = mrp_ll_submit(ctx, static_cast<int32_t>(H.jobs.size()), H.jobs.data(), H.results.data(), &H.ticket);
    if (rc != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_submit: ") + mrp_ll_last_error(ctx);
      return false;
    }
    H.inflight = true;
    out.searches += static_cast<int64_t>(H.jobs.size());
    return true;
  };
  auto consume = [&](Half& H) -> bool {
    int rc = mrp_ll_wait(ctx, H.ticket);
    H.inflight = false;
    if (rc != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_wait: ") + mrp_ll_last_error(ctx);
      return false;
    }
    for (size_t jq = 0; jq < H.jobs.size(); ++jq) {
      Prio& p = st[H.owner[jq]];
      mrp_hl_sipp_solution& so = sols[idx[H.owner[jq]]];
      const mrp_ll_result& r = H.results[jq];
      const int32_t a = p.agent;
      so.low_level_expanded += r.expanded;
      out.expansions += r.expanded;
      if (r.status != MRP_LL_OK && r.status != MRP_LL_NO_SOLUTION) {
        // a capacity status (expansion cap, node arena, horizon) is not the reference's answer for this agent, and every
        // later agent of the instance would plan against the wrong intervals: THIS instance stops here and says so in
        // its status; the other instances of the batch are not affected
        so.status = r.status;
        if (so.planned) so.planned[a] = 0;
        if (so.n_states) so.n_states[a] = 0;
        p.agent = instances[idx[H.owner[jq]]].n_agents;
        continue;
      }
      const bool ok = r.status == MRP_LL_OK;
      if (so.planned) so.planned[a] = ok ? 1 : 0;
      if (so.n_states) so.n_states[a] = ok ? r.n_states : 0;
      if (ok) {
        so.n_planned += 1;
        so.cost += r.cost;
        const int32_t* S = r.states_txy;  // [t, x, y]
        // update collision intervals (:237-246): one interval per maximal stay on a cell
        int32_t lx = S[1], ly = S[2], lt = S[0];
        for (int32_t i = 1; i < r.n_states; ++i) {
          if (S[3 * i + 1] != lx || S[3 * i + 2] != ly) {
            p.add(lx, ly, Iv{lt, S[3 * i] - 1});
            lx = S[3 * i + 1];
            ly = S[3 * i + 2];
            lt = S[3 * i];
          }
        }
        const int32_t last = r.n_states - 1;
        p.add(S[3 * last + 1], S[3 * last + 2], Iv{S[3 * last], INT32_MAX});
        if (so.states_xyt)
          for (int32_t i = 0; i < r.n_states && i < so.state_cap; ++i) {
            int32_t* dst = so.states_xyt + (static_cast<size_t>(a) * so.state_cap + i) * 3;
            dst[0] = S[3 * i + 1];
            dst[1] = S[3 * i + 2];
            dst[2] = S[3 * i];
          }
      }
      p.agent += 1;
    }
    return true;
  };

  for (int hIdx = 0; hIdx < 2; ++hIdx)
    if (launch(half[hIdx])) roundsOf[hIdx] += 1;
  if (!out.err.empty()) return;
  for (int cur = 0; half[0].inflight || half[1].inflight; cur ^= 1) {
    Half& H = half[cur];
    if (!H.inflight) continue;
    if (!consume(H)) {
      if (half[cur ^ 1].inflight) (void)mrp_ll_wait(ctx, half[cur ^ 1].ticket);  // do not leave a batch behind
      return;
    }
    if (launch(H)) roundsOf[cur] += 1;
    if (!out.err.empty()) {
      if (half[cur ^ 1].inflight) (void)mrp_ll_wait(ctx, half[cur ^ 1].ticket);
      return;
    }
  }
  out.rounds = std::max(roundsOf[0], roundsOf[1]);
}

// The same loop in session mode: the SIPP kernel stays resident and every instance publishes its next agent's search the
// moment the previous one has come back — no round barrier, so an instance never waits for another instance's search.
void runSippGroupSession(mrp_ll_ctx* ctx, int32_t horizon, int32_t slots, int32_t nWorkers, const mrp_hl_instance* instances,
                         mrp_hl_sipp_solution* sols, const std::vector<int32_t>& idx, GroupResult& out) {
  struct Iv { int32_t s, e; };
  struct Prio {
    int32_t mapId = -1, agent = 0, ticket = -1;
    // allCollisionIntervals (mapf_prioritized_sipp.cpp:215) lives in the engine: an incrementally maintained table, so a
    // job is a copy of the current table instead of a rebuild from every interval of the instance
    mrp_ll_sipp_table* tab = nullptr;
    std::vector<int32_t> states;
    mrp_ll_result res;
  };
  const size_t n = idx.size();
  const int32_t cap = std::max(horizon, 64);
  std::vector<Prio> st(n);
  for (size_t q = 0; q < n; ++q) {
    const mrp_hl_instance& in = instances[idx[q]];
    int rc = mrp_ll_upload_map(ctx, in.dimx, in.dimy, in.n_obstacles, in.obstacles_xy, &st[q].mapId);
    if (rc != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_upload_map: ") + mrp_ll_last_error(ctx);
      return;
    }
    sols[idx[q]].cost = 0;
    sols[idx[q]].low_level_expanded = 0;
    sols[idx[q]].n_planned = 0;
    sols[idx[q]].status = 0;
    if (mrp_ll_sipp_table_create(ctx, st[q].mapId, &st[q].tab) != MRP_LL_SUCCESS) {
      out.err = "mrp_ll_sipp_table_create failed";
      return;
    }
    st[q].states.resize(static_cast<size_t>(cap) * 3);
  }
  struct TableGuard {  // the tables go when the group is done, whichever way it ends
    std::vector<Prio>& st;
    ~TableGuard() {
      for (Prio& p : st) mrp_ll_sipp_table_destroy(p.tab);
    }
  } tableGuard{st};
  // at most one wavefront per instance, and as many over all workers as the device holds (mrp_ll_session_occupancy:
  // sixteen per CU with the kernel's 9.2 KB LDS tier).  Round 2 found that more resident wavefronts only slowed each other
  // down; that was with cached tables and a fence pair per job — with uncached tables residency pays (ll_kernel.hip,
  // MRP_LL_SIPP_LDS_NODES).
  int32_t occS = 6;
  if (mrp_ll_session_occupancy(ctx, MRP_LL_SIPP, &occS) != MRP_LL_SUCCESS || occS <= 0) occS = 6;
  const size_t perWorker = static_cast<size_t>(std::max(96, 256 * occS / std::max(nWorkers, 1)));
  int32_t wgs = static_cast<int32_t>(std::min<size_t>(std::min<size_t>(std::max<size_t>(n, 16), slots), perWorker));
  if (const char* e = std::getenv("MRP_HL_SIPP_WGS")) wgs = std::max(1, std::atoi(e));          // tuning knob
  if (mrp_ll_session_begin_sipp(ctx, wgs) != MRP_LL_SUCCESS) {
    out.err = std::string("mrp_ll_session_begin_sipp: ") + mrp_ll_last_error(ctx);
    return;
  }
  // 1 published, 0 no free job slot (retry later), -1 error
  auto submit = [&](size_t q) -> int {
    Prio& p = st[q];
    const mrp_hl_instance& in = instances[idx[q]];
    mrp_ll_job j;
    std::memset(&j, 0, sizeof(j));
    j.map_id = p.mapId;
    j.algo = MRP_LL_SIPP;
    j.w = 1.0f;
    j.start_x = in.starts_xy[2 * p.agent];
    j.start_y = in.starts_xy[2 * p.agent + 1];
    j.goal_x = in.goals_xy[2 * p.agent];
    j.goal_y = in.goals_xy[2 * p.agent + 1];
    j.max_expansions = sippMaxExpansions();
    j.sipp_table = p.tab;  // sipp.setCollisionIntervals(location, intervals) for every location (:224-226), kept up to date
    j.sipp_commit = 1;     // ... by the engine: the stays of the path it finds become collision intervals (:237-246)
    std::memset(&p.res, 0, sizeof(p.res));
    p.res.states_txy = p.states.data();
    p.res.states_cap = cap;
    int rc = mrp_ll_submit(ctx, 1, &j, &p.res, &p.ticket);
    if (rc == MRP_LL_E_BUSY) return 0;
    if (rc != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_submit: ") + mrp_ll_last_error(ctx);
      return -1;
    }
    out.searches += 1;
    return 1;
  };
  std::vector<size_t> backlog, ticketOwner, doneOwners;
  std::vector<int32_t> doneTickets(64);
  for (size_t q = n; q-- > 0;)
    if (instances[idx[q]].n_agents > 0) backlog.push_back(q);
  size_t nInflight = 0;
  bool failed = false;
  int64_t maxAgents = 0;
  auto t0 = std::chrono::steady_clock::now();
  uint64_t idleSpins = 0;
  static const bool timing = std::getenv("MRP_HL_SIPP_TIMING") != nullptr;  // where a worker thread's time goes
  double tSubmit = 0, tPoll = 0, tConsume = 0;
  uint64_t nPolls = 0, nEmptyPolls = 0;
  auto clk = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  bool sinceProgress = false;
  auto lastProgress = t0;
  while (!failed && (nInflight != 0 || !backlog.empty())) {
    bool progress = false;
    auto tA = timing ? clk() : t0;
    while (!backlog.empty()) {
      const size_t q = backlog.back();
      int r = submit(q);
      if (r < 0) failed = true;
      if (r <= 0) break;
      backlog.pop_back();
      if (static_cast<size_t>(st[q].ticket) >= ticketOwner.size()) ticketOwner.resize(st[q].ticket + 1, 0);
      ticketOwner[st[q].ticket] = q;
      nInflight += 1;
      progress = true;
    }
    if (failed) break;
    int32_t nDone = 0;
    auto tB = timing ? clk() : t0;
    if (mrp_ll_poll_any(ctx, doneTickets.data(), static_cast<int32_t>(doneTickets.size()), &nDone) != MRP_LL_SUCCESS) {
      out.err = std::string("mrp_ll_poll_any: ") + mrp_ll_last_error(ctx);
      failed = true;
      break;
    }
    auto tC = timing ? clk() : t0;
    if (timing) {
      tSubmit += secs(tA, tB);
      tPoll += secs(tB, tC);
      nPolls += 1;
      nEmptyPolls += nDone == 0 ? 1 : 0;
    }
    doneOwners.resize(nDone);
    for (int32_t d = 0; d < nDone; ++d) doneOwners[d] = ticketOwner[doneTickets[d]];
    for (int32_t d = 0; d < nDone && !failed; ++d) {
      const size_t q = doneOwners[d];
      Prio& p = st[q];
      const mrp_hl_instance& in = instances[idx[q]];
      mrp_hl_sipp_solution& so = sols[idx[q]];
      const mrp_ll_result& r = p.res;
      const int32_t a = p.agent;
      progress = true;
      nInflight -= 1;
      so.low_level_expanded += r.expanded;
      out.expansions += r.expanded;
      if (r.status != MRP_LL_OK && r.status != MRP_LL_NO_SOLUTION) {
        // capacity status: this instance stops here (see runSippGroup), the rest of the batch goes on
        so.status = r.status;
        if (so.planned) so.planned[a] = 0;
        if (so.n_states) so.n_states[a] = 0;
        p.agent = in.n_agents;
        continue;
      }
      const bool ok = r.status == MRP_LL_OK;
      if (so.planned) so.planned[a] = ok ? 1 : 0;
      if (so.n_states) so.n_states[a] = ok ? r.n_states : 0;
      if (ok) {
        so.n_planned += 1;
        so.cost += r.cost;
        const int32_t* S = r.states_txy;  // [t, x, y]
        // (the collision intervals of this path, :237-246 — one per maximal stay on a cell — are already in the table:
        // sipp_commit)
        if (so.states_xyt)
          for (int32_t i = 0; i < r.n_states && i < so.state_cap; ++i) {
            int32_t* dst = so.states_xyt + (static_cast<size_t>(a) * so.state_cap + i) * 3;
            dst[0] = S[3 * i + 1];
            dst[1] = S[3 * i + 2];
            dst[2] = S[3 * i];
          }
      }
      p.agent += 1;
      maxAgents = std::max<int64_t>(maxAgents, p.agent);
      if (p.agent < in.n_agents) {
        int rr = backlog.empty() ? submit(q) : 0;
        if (rr < 0) {
          failed = true;
        } else if (rr == 1) {
          if (static_cast<size_t>(p.ticket) >= ticketOwner.size()) ticketOwner.resize(p.ticket + 1, 0);
          ticketOwner[p.ticket] = q;
          nInflight += 1;
        } else {
          backlog.push_back(q);
        }
      }
    }
    if (timing) tConsume += secs(tC, clk());
    if (progress) {
      idleSpins = 0;
      sinceProgress = false;
    } else if ((++idleSpins & 0xFFFF) == 0) {
      const auto nowT = std::chrono::steady_clock::now();
      if (!sinceProgress) {
        sinceProgress = true;
        lastProgress = nowT;
      } else if (std::chrono::duration<double>(nowT - lastProgress).count() > kNoProgressLimitS) {
        out.err = "prioritized SIPP session: no progress for too long";
        failed = true;
      }
    }
  }
  if (mrp_ll_session_end(ctx) != MRP_LL_SUCCESS && out.err.empty())
    out.err = std::string("mrp_ll_session_end: ") + mrp_ll_last_error(ctx);
  if (timing)
    std::fprintf(stderr, "[mrp_hl] sipp worker: %.1f ms total; submit %.1f ms, poll_any %.1f ms (%llu calls, %llu empty), consume %.1f ms\n",
                 secs(t0, clk()) * 1e3, tSubmit * 1e3, tPoll * 1e3, (unsigned long long)nPolls,
                 (unsigned long long)nEmptyPolls, tConsume * 1e3);
  out.rounds = maxAgents;  // longest chain of dependent searches
}

}  // namespace

int mrp_hl_solver_prioritized_sipp(mrp_hl_solver* s, int32_t nInst, const mrp_hl_instance* instances,
                                   mrp_hl_sipp_solution* sols, mrp_hl_batch_stats* stats) {
  if (!s || nInst < 0 || (nInst > 0 && (!instances || !sols))) return MRP_LL_E_INVALID;
  const int32_t horizon = s->llOpt.max_horizon > 0 ? s->llOpt.max_horizon : 512;
  // instances are independent: one share per worker thread / engine, as for CBS and ECBS
  const int32_t nThreads = std::max(1, std::min<int32_t>(static_cast<int32_t>(s->engines.size()), std::max(nInst, 1)));
  std::vector<std::vector<int32_t>> idx(nThreads);
  for (int32_t k = 0; k < nInst; ++k) idx[k % nThreads].push_back(k);
  std::vector<GroupResult> gr(nThreads);
  const bool batchMode = std::getenv("MRP_HL_SIPP_BATCH") != nullptr;  // the round-based schedule, for comparison
  auto t0 = std::chrono::steady_clock::now();
  {
    std::vector<std::thread> th;
    for (int32_t t = 0; t < nThreads; ++t)
      th.emplace_back([&, t]() {
        if (batchMode)
          runSippGroup(s->engines[t], horizon, s->llOpt.n_tickets, instances, sols, idx[t], gr[t]);
        else
          runSippGroupSession(s->engines[t], horizon, s->llOpt.slots, nThreads, instances, sols, idx[t], gr[t]);
      });
    for (auto& x : th) x.join();
  }
  mrp_hl_batch_stats bs;
  std::memset(&bs, 0, sizeof(bs));
  bs.wall_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  for (auto& g : gr) {
    if (!g.err.empty()) {
      s->err = g.err;
      return MRP_LL_E_DEVICE;
    }
    bs.rounds = std::max(bs.rounds, g.rounds);  // rounds run concurrently on the worker threads: the longest chain
    bs.ll_searches += g.searches;
    bs.ll_expansions += g.expansions;
  }
  for (int32_t k = 0; k < nInst; ++k) bs.solved += sols[k].n_planned == instances[k].n_agents ? 1 : 0;
  if (stats) *stats = bs;
  return MRP_LL_SUCCESS;
}

int mrp_hl_solve_batch(int32_t device, const mrp_hl_options* opt, int32_t nInst, const mrp_hl_instance* instances,
                       mrp_hl_solution* solutions, mrp_hl_batch_stats* stats) {
  mrp_hl_solver* s = nullptr;
  int rc = mrp_hl_solver_create(device, opt ? opt->n_threads : 0, nullptr, &s);
  if (rc != MRP_LL_SUCCESS) return rc;
  rc = mrp_hl_solver_solve(s, opt, nInst, instances, solutions, stats);
  mrp_hl_solver_destroy(s);
  return rc;
}

// ---- one conflict tree, stepped by the caller (include/mrp_hl.h "mrp_hl_ct_*") ---------------------------------
struct mrp_hl_ct {
  std::unique_ptr<Instance> inst;
  std::vector<LLRequest> req;
  std::vector<int32_t> pathLenPool;
  std::vector<const int32_t*> pathPtrPool;
  std::vector<size_t> poolOff;
  std::vector<mrp_ll_job> jobs;
  void rebuild() {  // job views of the pending requests; arrays stay valid until the next deliver
    pathLenPool.clear();
    pathPtrPool.clear();
    poolOff.clear();
    jobs.resize(req.size());
    for (size_t k = 0; k < req.size(); ++k) {
      poolOff.push_back(pathLenPool.size());
      fillJob(*inst, req[k], jobs[k], pathLenPool, pathPtrPool);
    }
    for (size_t k = 0; k < req.size(); ++k)
      if (jobs[k].n_agents > 0) {
        jobs[k].path_len = pathLenPool.data() + poolOff[k];
        jobs[k].path_xy = pathPtrPool.data() + poolOff[k];
      }
  }
};

int mrp_hl_ct_create(const mrp_hl_instance* in, const mrp_hl_options* opt, int32_t mapId, int32_t specWidth,
                     mrp_hl_ct** out) {
  if (!in || !opt || !out || in->n_agents < 0) return MRP_LL_E_INVALID;
  auto* c = new mrp_hl_ct();
  c->inst.reset(new Instance(*in, mapId, *opt));
  c->inst->setSpecWidth(specWidth);
  c->inst->start(c->req);
  if (c->inst->done()) c->req.clear();
  c->rebuild();
  *out = c;
  return MRP_LL_SUCCESS;
}

void mrp_hl_ct_destroy(mrp_hl_ct* c) { delete c; }

int32_t mrp_hl_ct_n_requests(const mrp_hl_ct* c) { return c ? static_cast<int32_t>(c->req.size()) : 0; }

int mrp_hl_ct_request(const mrp_hl_ct* c, int32_t k, mrp_ll_job* job, int32_t* group, int32_t* slot) {
  if (!c || k < 0 || k >= static_cast<int32_t>(c->req.size()) || !job) return MRP_LL_E_INVALID;
  *job = c->jobs[k];
  if (group) *group = c->req[k].group;
  if (slot) *slot = c->req[k].slot;
  return MRP_LL_SUCCESS;
}

int mrp_hl_ct_deliver(mrp_hl_ct* c, int32_t group, int32_t n, const mrp_ll_result* results) {
  if (!c || n < 0 || (n > 0 && !results)) return MRP_LL_E_INVALID;
  // the group's requests are consecutive in the pending list: take them out, keep the others
  size_t a = 0;
  while (a < c->req.size() && c->req[a].group != group) ++a;
  size_t b = a;
  while (b < c->req.size() && c->req[b].group == group) ++b;
  if (a == c->req.size() || static_cast<int32_t>(b - a) != n) return MRP_LL_E_INVALID;
  std::vector<LLAnswer> ans;
  for (int32_t k = 0; k < n; ++k) ans.push_back(answerOf(results[k]));
  c->req.erase(c->req.begin() + static_cast<std::ptrdiff_t>(a), c->req.begin() + static_cast<std::ptrdiff_t>(b));
  c->inst->deliver(group, ans, c->req);
  if (c->inst->done()) c->req.clear();  // requests of a finished instance point into freed CT nodes
  c->rebuild();
  return MRP_LL_SUCCESS;
}

int32_t mrp_hl_ct_done(const mrp_hl_ct* c) { return c && c->inst->done() ? 1 : 0; }

namespace {
constexpr int32_t kRowHdr = 8;
constexpr int32_t kFailedRound = INT32_MIN;  // group word of a failure row (group ids are >= -1: -1 is the root step)
// the pending groups in order, and which rank searches which (group j -> rank j % world)
void roundOwners(const mrp_hl_ct& c, int32_t world, std::vector<int32_t>& groups, std::vector<int32_t>& ownerOfReq) {
  groups.clear();
  ownerOfReq.assign(c.req.size(), 0);
  for (size_t k = 0; k < c.req.size(); ++k) {
    if (groups.empty() || groups.back() != c.req[k].group) groups.push_back(c.req[k].group);
    ownerOfReq[k] = static_cast<int32_t>((groups.size() - 1) % static_cast<size_t>(world));
  }
}
}  // namespace

int32_t mrp_hl_ct_round_mine(mrp_hl_ct* c, mrp_ll_ctx* ll, int32_t rank, int32_t world, int32_t maxStates, int32_t* rows,
                             int32_t capRows, int32_t* rowsPerRank) {
  if (!c || !ll || !rows || !rowsPerRank || world < 1 || rank < 0 || rank >= world || maxStates < 1 || capRows < 1)
    return MRP_LL_E_INVALID;
  const size_t rowWords = static_cast<size_t>(kRowHdr) + static_cast<size_t>(maxStates);
  std::vector<int32_t> groups, owner;
  roundOwners(*c, world, groups, owner);
  for (int32_t r = 0; r < world; ++r) rowsPerRank[r] = 0;
  std::vector<size_t> mine;
  for (size_t k = 0; k < c->req.size(); ++k) {
    rowsPerRank[owner[k]] += 1;
    if (owner[k] == rank) mine.push_back(k);
  }
  auto fail = [&](int rc) {
    std::memset(rows, 0, rowWords * sizeof(int32_t));
    rows[0] = kFailedRound;
    rows[2] = rc;
    return rc;
  };
  if (static_cast<int32_t>(mine.size()) > capRows) return fail(MRP_LL_E_INVALID);
  std::vector<mrp_ll_job> jobs(mine.size());
  std::vector<mrp_ll_result> res(mine.size());
  std::vector<int32_t> states(mine.size() * static_cast<size_t>(maxStates) * 3);
  for (size_t q = 0; q < mine.size(); ++q) {
    jobs[q] = c->jobs[mine[q]];
    std::memset(&res[q], 0, sizeof(mrp_ll_result));
    res[q].states_txy = states.data() + q * static_cast<size_t>(maxStates) * 3;
    res[q].states_cap = maxStates;
  }
  if (!mine.empty()) {
    const int rc = mrp_ll_search_batch(ll, static_cast<int32_t>(jobs.size()), jobs.data(), res.data());
    if (rc != MRP_LL_SUCCESS) return fail(rc);
  }
  for (size_t q = 0; q < mine.size(); ++q) {
    int32_t* row = rows + q * rowWords;
    const mrp_ll_result& r = res[q];
    if (r.status == MRP_LL_PATH_TRUNCATED || r.n_states > maxStates) return fail(MRP_LL_E_INVALID);  // max_states too small
    row[0] = c->req[mine[q]].group;
    row[1] = c->req[mine[q]].slot;
    row[2] = r.status;
    row[3] = r.cost;
    row[4] = r.fmin;
    row[5] = static_cast<int32_t>(r.expanded & 0x7FFFFFFF);
    row[6] = static_cast<int32_t>(r.expanded >> 31);
    row[7] = r.n_states;
    for (int32_t k = 0; k < r.n_states; ++k) row[kRowHdr + k] = r.states_txy[3 * k + 1] | (r.states_txy[3 * k + 2] << 16);
  }
  return static_cast<int32_t>(mine.size());
}

int mrp_hl_ct_deliver_rows(mrp_hl_ct* c, const int32_t* gathered, int32_t world, int32_t rowsStride, int32_t maxStates) {
  if (!c || !gathered || world < 1 || rowsStride < 1 || maxStates < 1) return MRP_LL_E_INVALID;
  const size_t rowWords = static_cast<size_t>(kRowHdr) + static_cast<size_t>(maxStates);
  std::vector<int32_t> groups, owner;
  roundOwners(*c, world, groups, owner);
  std::vector<int32_t> perRank(world, 0);
  for (size_t k = 0; k < c->req.size(); ++k) perRank[owner[k]] += 1;
  // a rank that failed sent one row marked kFailedRound: every rank stops here, together
  for (int32_t r = 0; r < world; ++r)
    if (gathered[static_cast<size_t>(r) * rowsStride * rowWords] == kFailedRound) return MRP_LL_E_DEVICE;
  struct Row {
    int32_t slot;
    const int32_t* w;
  };
  std::map<int32_t, std::vector<Row>> byGroup;
  for (int32_t r = 0; r < world; ++r) {
    if (perRank[r] > rowsStride) return MRP_LL_E_INVALID;
    for (int32_t q = 0; q < perRank[r]; ++q) {
      const int32_t* w = gathered + (static_cast<size_t>(r) * rowsStride + q) * rowWords;
      byGroup[w[0]].push_back(Row{w[1], w});
    }
  }
  std::vector<int32_t> states;
  std::vector<mrp_ll_result> res;
  for (int32_t g : groups) {  // the same order on every rank
    auto it = byGroup.find(g);
    if (it == byGroup.end()) return MRP_LL_E_INVALID;
    std::vector<Row>& rws = it->second;
    std::sort(rws.begin(), rws.end(), [](const Row& a, const Row& b) { return a.slot < b.slot; });
    res.assign(rws.size(), mrp_ll_result());
    states.assign(rws.size() * static_cast<size_t>(maxStates) * 3, 0);
    for (size_t q = 0; q < rws.size(); ++q) {
      const int32_t* w = rws[q].w;
      mrp_ll_result& r = res[q];
      std::memset(&r, 0, sizeof(r));
      r.status = w[2];
      r.cost = w[3];
      r.fmin = w[4];
      r.expanded = static_cast<int64_t>(w[5]) | (static_cast<int64_t>(w[6]) << 31);
      r.n_states = w[7];
      if (r.n_states < 0 || r.n_states > maxStates) return MRP_LL_E_INVALID;
      r.states_txy = states.data() + q * static_cast<size_t>(maxStates) * 3;
      r.states_cap = maxStates;
      for (int32_t k = 0; k < r.n_states; ++k) {
        r.states_txy[3 * k] = k;
        r.states_txy[3 * k + 1] = w[kRowHdr + k] & 0xFFFF;
        r.states_txy[3 * k + 2] = w[kRowHdr + k] >> 16;
      }
    }
    const int rc = mrp_hl_ct_deliver(c, g, static_cast<int32_t>(res.size()), res.data());
    if (rc != MRP_LL_SUCCESS) return rc;
    if (c->inst->done()) break;
  }
  return MRP_LL_SUCCESS;
}

int mrp_hl_ct_solution(const mrp_hl_ct* c, mrp_hl_solution* out) {
  if (!c || !out || !c->inst->done()) return MRP_LL_E_INVALID;
  writeSolution(*c->inst, *out);
  return MRP_LL_SUCCESS;
}

int32_t mrp_hl_astar_grid2d(int32_t dimx, int32_t dimy, const uint8_t* obstacle_mask, int32_t start_x, int32_t start_y,
                            int32_t goal_x, int32_t goal_y, int32_t* states_xy, int32_t cap, int32_t* cost,
                            int64_t* expanded) {
  if (dimx <= 0 || dimy <= 0 || !obstacle_mask || !cost || !expanded || (cap > 0 && !states_xy)) return -1;
  return astarGrid2d(dimx, dimy, obstacle_mask, start_x, start_y, goal_x, goal_y, states_xy, cap, cost, expanded);
}

int mrp_hl_generate_instance(uint64_t seed, int32_t dimx, int32_t dimy, int32_t nObst, int32_t nAgents,
                             int32_t* obstXY, int32_t* startsXY, int32_t* goalsXY) {
  return generateInstance(seed, dimx, dimy, nObst, nAgents, obstXY, startsXY, goalsXY);
}

int mrp_hl_generate_instances(uint64_t seed0, int32_t n, int32_t dimx, int32_t dimy, int32_t nObst, int32_t nAgents,
                              int32_t* obstXY, int32_t* startsXY, int32_t* goalsXY) {
  if (n < 0 || (n > 0 && (!obstXY || !startsXY || !goalsXY))) return -1;
  unsigned hc = std::thread::hardware_concurrency();
  const int32_t nThreads = std::max<int32_t>(1, std::min<int32_t>(static_cast<int32_t>(hc ? hc : 8), std::min(n, 64)));
  std::atomic<int32_t> next(0), bad(0);
  std::vector<std::thread> th;
  for (int32_t t = 0; t < nThreads; ++t)
    th.emplace_back([&]() {
      for (;;) {
        const int32_t k0 = next.fetch_add(64, std::memory_order_relaxed);
        if (k0 >= n) return;
        for (int32_t k = k0; k < std::min(n, k0 + 64); ++k)
          if (generateInstance(seed0 + static_cast<uint64_t>(k), dimx, dimy, nObst, nAgents,
                               obstXY + static_cast<size_t>(k) * nObst * 2, startsXY + static_cast<size_t>(k) * nAgents * 2,
                               goalsXY + static_cast<size_t>(k) * nAgents * 2) != 0)
            bad.fetch_add(1, std::memory_order_relaxed);
      }
    });
  for (auto& x : th) x.join();
  return bad.load() == 0 ? 0 : -1;
}

}  // extern "C"
