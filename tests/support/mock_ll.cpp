// TEST INFRASTRUCTURE ONLY — never shipped, never built into libmultirobotplanning_amd/lib.
// Implements the C-ABI of include/mrp_ll.h on top of the ORACLE's low-level search (oracle_ll_search) so that the
// host-side conflict-tree drivers (csrc/hl/) can be exercised on a machine without a GPU (`-m "not gpu"` tests).
// The product library has no such path: libmrp_ll.so fails with MRP_LL_E_DEVICE when no HIP device exists.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <string>
#include <map>
#include <mutex>
#include <vector>

#include "../../include/mrp_ll.h"

extern "C" void oracle_conflict_scan(int nAgents, const int32_t* pathLen, const int32_t* pathXY, int32_t* out);
extern "C" int oracle_ll_search(int algo, float w, int dimx, int dimy, int nObst, const int32_t* obstXY, int agentIdx,
                                int startX, int startY, int goalX, int goalY, int nVC, const int32_t* vc, int nEC,
                                const int32_t* ec, int nCtx, const int32_t* ctxLen, const int32_t* ctxXY,
                                int64_t capExpansions, int32_t* out, int64_t* expanded, int32_t* statesTXY,
                                int32_t* actions, int cap);

struct MockMap {
  int dimx, dimy;
  std::vector<int32_t> obst;
};
struct mrp_ll_ctx {
  std::vector<MockMap> maps;
  mrp_ll_stats stats;
  std::string err;
  int32_t pendingJobs = 0;
  std::vector<int32_t> doneTickets;  // completed (synchronously in mrp_ll_submit), not yet reported by poll_any
  int32_t nextTicket = 0;
  const mrp_ll_job* jobs = nullptr;
  mrp_ll_result* results = nullptr;
  std::vector<std::vector<int32_t>> store;  // MRP_MOCK_PATH_STORE=1: the "device" path store, slot -> x, y, x, y, ...
};

extern "C" {
const char* mrp_ll_version(void) { return "mrp_ll MOCK (oracle-backed, tests only)"; }
const char* mrp_ll_last_error(const mrp_ll_ctx* c) { return c ? c->err.c_str() : "null"; }
int mrp_ll_create(const mrp_ll_options*, mrp_ll_ctx** out) {
  *out = new mrp_ll_ctx();
  std::memset(&(*out)->stats, 0, sizeof(mrp_ll_stats));
  return MRP_LL_SUCCESS;
}
void mrp_ll_destroy(mrp_ll_ctx* c) { delete c; }
int mrp_ll_upload_map(mrp_ll_ctx* c, int32_t dimx, int32_t dimy, int32_t n, const int32_t* xy, int32_t* id) {
  MockMap m{dimx, dimy, std::vector<int32_t>(xy, xy + 2 * n)};
  c->maps.push_back(m);
  *id = static_cast<int32_t>(c->maps.size()) - 1;
  return MRP_LL_SUCCESS;
}
// MRP_LL_JOB_ROOT_CHAIN on the mock: agent after agent through the oracle, contexts from the mock's path store.
// MRP_MOCK_CHAIN_BREAK=K: a search of more than K expansions "does not fit the LDS tier" and ends the chain in front of it.
static void mockChain(mrp_ll_ctx* c, const mrp_ll_job& j, mrp_ll_result& r) {
  const MockMap& m = c->maps[j.map_id];
  const int n = j.n_agents, first = j.agent_idx;
  const int end = j.chain_count > 0 ? std::min(n, first + j.chain_count) : n;
  const char* bk = std::getenv("MRP_MOCK_CHAIN_BREAK");
  const int64_t breakAt = bk ? std::atoll(bk) : -1;
  int64_t budget = j.max_expansions, total = 0;
  int done = 0;
  bool stopped = false;
  r.status = MRP_LL_OK;
  for (int a = first; a < n; ++a) {
    mrp_ll_result& ri = r.chain_results[a - first];
    ri.status = MRP_LL_NOT_RUN;
    ri.cost = ri.fmin = ri.n_states = 0;
    ri.expanded = 0;
    if (stopped || a >= end) continue;
    std::vector<int32_t> ctxLen, ctxXY;
    for (int b = 0; b < n; ++b) {
      const std::vector<int32_t>* p = b < a ? &c->store[j.path_ids[b]] : nullptr;
      ctxLen.push_back(p ? static_cast<int32_t>(p->size() / 2) : 0);
      if (p) ctxXY.insert(ctxXY.end(), p->begin(), p->end());
    }
    const int32_t* q = j.chain_starts_goals_xy + 4 * a;
    int32_t out[4];
    int64_t expanded = 0;
    std::vector<int32_t> st(3 * 2048), ac(2048);
    int rc = oracle_ll_search(MRP_LL_ASTAR_EPS, j.w, m.dimx, m.dimy, static_cast<int>(m.obst.size() / 2), m.obst.data(), a, q[0],
                              q[1], q[2], q[3], 0, nullptr, 0, nullptr, n, ctxLen.data(), ctxXY.data(), budget, out, &expanded,
                              st.data(), ac.data(), 2048);
    if (breakAt >= 0 && expanded > breakAt) {  // (the device would have found out on the way; the answer is not used)
      stopped = true;
      continue;
    }
    ri.expanded = expanded;
    ri.tier = 0;
    total += expanded;
    done += 1;
    c->stats.jobs += 1;
    c->stats.expansions += expanded;
    if (rc == -1) {
      ri.status = MRP_LL_CAP_EXPANSIONS;
      stopped = true;
      continue;
    }
    ri.status = out[0] ? MRP_LL_OK : MRP_LL_NO_SOLUTION;
    ri.cost = out[1];
    ri.fmin = out[2];
    ri.n_states = out[0] ? out[3] : 0;
    if (!out[0]) {
      stopped = true;
      continue;
    }
    std::vector<int32_t>& slot = c->store[j.path_ids[a]];
    slot.clear();
    for (int k = 0; k < ri.n_states; ++k) {
      if (ri.states_txy && k < ri.states_cap) std::memcpy(ri.states_txy + 3 * k, st.data() + 3 * k, 12);
      slot.push_back(st[3 * k + 1]);
      slot.push_back(st[3 * k + 2]);
    }
    if (budget >= 0) budget = budget > expanded ? budget - expanded : 0;
  }
  r.n_states = done;
  r.expanded = total;
  // the root node's conflicts (mrp_ll.h): the oracle's getFirstConflict + focalHeuristic over the chain's paths
  r.cost = r.fmin = -1;
  if (first == 0 && end == n && done == n && !stopped) {
    std::vector<int32_t> len, xy;
    for (int a = 0; a < n; ++a) {
      const std::vector<int32_t>& p = c->store[j.path_ids[a]];
      len.push_back(static_cast<int32_t>(p.size() / 2));
      xy.insert(xy.end(), p.begin(), p.end());
    }
    int32_t o[10];
    oracle_conflict_scan(n, len.data(), xy.data(), o);
    r.cost = o[9];
    r.fmin = o[0] ? (o[1] << 24) | (o[4] << 16) | (o[2] << 8) | o[3] : -1;
  }
}

int mrp_ll_search_batch(mrp_ll_ctx* c, int32_t n, const mrp_ll_job* jobs, mrp_ll_result* res) {
  for (int i = 0; i < n; ++i) {
    const mrp_ll_job& j = jobs[i];
    mrp_ll_result& r = res[i];
    if ((j.flags & MRP_LL_JOB_ROOT_CHAIN) && j.map_id >= 0 && j.map_id < static_cast<int>(c->maps.size()) && !c->store.empty()) {
      mockChain(c, j, r);
      continue;
    }
    if (j.map_id < 0 || j.map_id >= static_cast<int>(c->maps.size())) {
      r.status = MRP_LL_BAD_JOB;
      continue;
    }
    const MockMap& m = c->maps[j.map_id];
    std::vector<int32_t> ctxLen, ctxXY;
    int nCtx = j.algo == MRP_LL_ASTAR_EPS ? j.n_agents : 0;
    for (int a = 0; a < nCtx; ++a) {
      int len = a == j.agent_idx ? 0 : j.path_len[a];
      ctxLen.push_back(len);
      for (int k = 0; k < 2 * len; ++k) ctxXY.push_back(j.path_xy[a][k]);
    }
    int32_t out[4];
    int64_t expanded = 0;
    std::vector<int32_t> st(3 * 2048), ac(2048);
    int rc = oracle_ll_search(j.algo, j.w, m.dimx, m.dimy, static_cast<int>(m.obst.size() / 2), m.obst.data(),
                              j.agent_idx, j.start_x, j.start_y, j.goal_x, j.goal_y, j.n_vertex_constraints,
                              j.vertex_constraints, j.n_edge_constraints, j.edge_constraints, nCtx, ctxLen.data(),
                              ctxXY.data(), j.max_expansions, out, &expanded, st.data(), ac.data(), 2048);
    r.expanded = expanded;
    r.tier = 0;
    if (rc == -1) {
      r.status = MRP_LL_CAP_EXPANSIONS;
      continue;
    }
    r.status = out[0] ? MRP_LL_OK : MRP_LL_NO_SOLUTION;
    r.cost = out[1];
    r.fmin = out[2];
    r.n_states = out[0] ? out[3] : 0;
    for (int k = 0; k < r.n_states && k < r.states_cap; ++k) {
      if (r.states_txy) std::memcpy(r.states_txy + 3 * k, st.data() + 3 * k, 12);
      if (r.actions && k + 1 < r.n_states) r.actions[k] = ac[k];
    }
    if ((j.flags & MRP_LL_JOB_STORE_RESULT) && j.result_path_id >= 0 && j.result_path_id < static_cast<int>(c->store.size())) {
      std::vector<int32_t>& slot = c->store[j.result_path_id];
      slot.clear();
      for (int k = 0; k < r.n_states; ++k) {
        slot.push_back(st[3 * k + 1]);
        slot.push_back(st[3 * k + 2]);
      }
    }
    c->stats.jobs += 1;
    c->stats.expansions += expanded;
  }
  c->stats.launches += 1;
  return MRP_LL_SUCCESS;
}
int mrp_ll_submit(mrp_ll_ctx* c, int32_t n, const mrp_ll_job* jobs, mrp_ll_result* res, int32_t* ticket) {
  *ticket = c->nextTicket++ & 0xFFFF;
  c->doneTickets.push_back(*ticket);
  return mrp_ll_search_batch(c, n, jobs, res);
}
int mrp_ll_submit_lane(mrp_ll_ctx* c, int32_t, int32_t n, const mrp_ll_job* jobs, mrp_ll_result* res, int32_t* ticket) {
  return mrp_ll_submit(c, n, jobs, res, ticket);
}
// co-workers: two host threads on one context (mrp_ll.h); finished tickets wait per tag
namespace {
std::mutex g_coMu;
std::map<mrp_ll_ctx*, std::vector<int32_t>> g_coDone[4];
}  // namespace
int mrp_ll_submit_tagged(mrp_ll_ctx* c, int32_t tag, int32_t n, const mrp_ll_job* jobs, mrp_ll_result* res, int32_t* ticket) {
  if (tag < 0 || tag > 3) return MRP_LL_E_INVALID;
  std::lock_guard<std::mutex> lock(g_coMu);
  *ticket = c->nextTicket++ & 0xFFFF;
  g_coDone[tag][c].push_back(*ticket);
  return mrp_ll_search_batch(c, n, jobs, res);
}
int mrp_ll_poll_any_tagged(mrp_ll_ctx* c, int32_t tag, int32_t* tickets, int32_t cap, int32_t* n) {
  if (tag < 0 || tag > 3) return MRP_LL_E_INVALID;
  std::lock_guard<std::mutex> lock(g_coMu);
  std::vector<int32_t>& d = g_coDone[tag][c];
  int32_t k = 0;
  while (!d.empty() && k < cap && k < 3) {  // a few at a time, newest first: out of submission order
    tickets[k++] = d.back();
    d.pop_back();
  }
  *n = k;
  return MRP_LL_SUCCESS;
}
// MRP_MOCK_SHUFFLE=<seed>: completed tickets are reported a few at a time in a pseudo-random order, so that the drivers
// see the groups of one instance come back out of submission order (as they do on the GPU).
int mrp_ll_poll_any(mrp_ll_ctx* c, int32_t* tickets, int32_t cap, int32_t* n) {
  int32_t k = 0;
  const char* sh = std::getenv("MRP_MOCK_SHUFFLE");
  if (sh) {
    static thread_local uint64_t x = 0;
    if (x == 0) x = 0x9E3779B97F4A7C15ull ^ static_cast<uint64_t>(std::atoll(sh) + 1) ^ reinterpret_cast<uintptr_t>(c);
    auto rnd = [&]() {
      x ^= x << 13;
      x ^= x >> 7;
      x ^= x << 17;
      return x;
    };
    if (!c->doneTickets.empty() && (rnd() & 3) == 0) {  // sometimes report nothing: work stays in flight
      *n = 0;
      return MRP_LL_SUCCESS;
    }
    const int32_t lim = 1 + static_cast<int32_t>(rnd() % 3);
    while (!c->doneTickets.empty() && k < cap && k < lim) {
      const size_t i = rnd() % c->doneTickets.size();
      tickets[k++] = c->doneTickets[i];
      c->doneTickets[i] = c->doneTickets.back();
      c->doneTickets.pop_back();
    }
    *n = k;
    return MRP_LL_SUCCESS;
  }
  while (!c->doneTickets.empty() && k < cap) {
    tickets[k++] = c->doneTickets.back();
    c->doneTickets.pop_back();
  }
  *n = k;
  return MRP_LL_SUCCESS;
}
int mrp_ll_wait(mrp_ll_ctx*, int32_t) { return MRP_LL_SUCCESS; }
int mrp_ll_sync_maps(mrp_ll_ctx*) { return MRP_LL_SUCCESS; }
int mrp_ll_release_maps(mrp_ll_ctx* c) {
  c->maps.clear();
  return MRP_LL_SUCCESS;
}
int mrp_ll_configure_tiers(mrp_ll_ctx*, int32_t, int32_t, int32_t, int32_t* occ) {
  if (occ) *occ = 4;
  return MRP_LL_SUCCESS;
}
int mrp_ll_session_occupancy(mrp_ll_ctx*, int32_t, int32_t* occ) {
  if (occ) *occ = 4;
  return MRP_LL_SUCCESS;
}
int mrp_ll_session_begin(mrp_ll_ctx*, int32_t) { return MRP_LL_SUCCESS; }
int mrp_ll_session_begin_sipp(mrp_ll_ctx*, int32_t) { return MRP_LL_SUCCESS; }
int mrp_ll_session_begin_algo(mrp_ll_ctx*, int32_t, int32_t) { return MRP_LL_SUCCESS; }
int mrp_ll_session_begin_tiers(mrp_ll_ctx*, int32_t, int32_t, int32_t) { return MRP_LL_SUCCESS; }
int mrp_ll_session_begin_tiers_gated(mrp_ll_ctx*, int32_t, int32_t, int32_t, int32_t* gate, int32_t) {
  if (gate) __atomic_fetch_add(gate, 1, __ATOMIC_ACQ_REL);
  return MRP_LL_SUCCESS;
}
int mrp_ll_session_tiers_geometry(mrp_ll_ctx*, int32_t* occ, int32_t* front, int32_t* heavy) {
  if (occ) *occ = 12;
  if (front) *front = 12928;
  if (heavy) *heavy = 31360;
  return MRP_LL_SUCCESS;
}
int mrp_ll_session_end(mrp_ll_ctx*) { return MRP_LL_SUCCESS; }
int mrp_ll_poll(mrp_ll_ctx*, int32_t, int32_t* done) {
  *done = 1;  // the mock runs every job synchronously inside mrp_ll_submit
  return MRP_LL_SUCCESS;
}
int mrp_ll_conflict_scan(mrp_ll_ctx*, int32_t, const int32_t*, const int32_t*, const int32_t*, mrp_ll_conflict*) {
  return MRP_LL_E_DEVICE;  // the scan kernel has no stand-in: the host drivers do not call it
}
// no store by default: the drivers fall back to tables.  MRP_MOCK_PATH_STORE=1: a host-side one, so that the drivers' path
// ids and root chains run on the CPU too
int mrp_ll_path_store_reserve(mrp_ll_ctx* c, int32_t n) {
  const char* e = std::getenv("MRP_MOCK_PATH_STORE");
  if (!e || *e != '1') return MRP_LL_E_DEVICE;
  c->store.assign(static_cast<size_t>(std::max(n, 0)), std::vector<int32_t>());
  return MRP_LL_SUCCESS;
}
struct mrp_ll_sipp_table {};  // SIPP has no stand-in here: the prioritized-SIPP driver is covered by the GPU tests
int mrp_ll_sipp_table_create(mrp_ll_ctx*, int32_t, mrp_ll_sipp_table** out) {
  *out = new mrp_ll_sipp_table();
  return MRP_LL_SUCCESS;
}
int mrp_ll_sipp_table_add(mrp_ll_sipp_table*, int32_t, int32_t, int32_t, int32_t) { return MRP_LL_SUCCESS; }
void mrp_ll_sipp_table_destroy(mrp_ll_sipp_table* t) { delete t; }
int mrp_ll_get_stats(const mrp_ll_ctx* c, mrp_ll_stats* out) {
  *out = c->stats;
  return MRP_LL_SUCCESS;
}
int mrp_ll_reset_stats(mrp_ll_ctx* c) {
  std::memset(&c->stats, 0, sizeof(c->stats));
  return MRP_LL_SUCCESS;
}
}
