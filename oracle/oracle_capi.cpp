// ORACLE — TEST INFRASTRUCTURE ONLY (see heap_restated.hpp header).
// Plain-C entry points over the restated CPU algorithms so tests/ and bench.py's cpu_baseline leg can
// drive them through ctypes.  Never linked into the product library.
#include <chrono>
#include <cstdint>
#include <cstring>
#include <atomic>
#include <thread>
#include <vector>

#ifdef ORACLE_USE_BOOST_HEAP  // cross-check build (Makefile: liboracle_boost.so)
#include "boost_heap_adapter.hpp"
#ifndef ORACLE_HAVE_BOOST_HEAP
#error "liboracle_boost.so needs <boost/heap/d_ary_heap.hpp>"
#endif
#define ORACLE_HEAP BoostHeap
#endif
#include "grid2d_restated.hpp"
#include "mapf_restated.hpp"
#include "sipp_restated.hpp"
#include "ta_restated.hpp"

using namespace oracle;
using namespace oracle::mapf;

namespace {

Environment makeEnv(int dimx, int dimy, int nObst, const int32_t* obstXY, int nGoals, const int32_t* goalsXY) {
  std::unordered_set<Cell, CellHash> obst;
  for (int i = 0; i < nObst; ++i) obst.insert(Cell{obstXY[2 * i], obstXY[2 * i + 1]});
  std::vector<Cell> goals;
  for (int i = 0; i < nGoals; ++i) goals.push_back(Cell{goalsXY[2 * i], goalsXY[2 * i + 1]});
  return Environment(dimx, dimy, std::move(obst), std::move(goals));
}

int actionCode(Action a) { return static_cast<int>(a); }

}  // namespace

extern "C" {

// 0: the heaps are the restated MutableBinaryHeap; 1: this library was built on the real boost::heap::d_ary_heap
int oracle_heap_kind(void) {
#ifdef ORACLE_USE_BOOST_HEAP
  return 1;
#else
  return 0;
#endif
}

// algo: 0 = CBS (cbs.hpp), 1 = ECBS (ecbs.hpp, bound w as float32).
// stats[0..5] = cost, makespan, highLevelExpanded, lowLevelExpanded, elapsed_ns (search() only), n_ll_searches
// Return: 1 solved, 0 search returned false, -1 harness cap exceeded.
int oracle_mapf_solve(int algo, float w, int dimx, int dimy, int nObst, const int32_t* obstXY, int nAgents,
                      const int32_t* startsXY, const int32_t* goalsXY, int64_t capPerSearch, int64_t capTotal,
                      int64_t capHL, int64_t* stats, int32_t* pathLen, int32_t* pathsXY, int pathCap) {
  Environment env = makeEnv(dimx, dimy, nObst, obstXY, nAgents, goalsXY);
  std::vector<State> starts;
  for (int i = 0; i < nAgents; ++i) starts.emplace_back(0, startsXY[2 * i], startsXY[2 * i + 1]);
  Limits lim;
  lim.maxLowLevelExpansionsPerSearch = capPerSearch;
  lim.maxLowLevelExpansionsTotal = capTotal;
  lim.maxHighLevelExpansions = capHL;
  std::vector<Plan> sol;
  bool ok = false;
  int rc = 0;
  auto t0 = std::chrono::steady_clock::now();
  try {
    if (algo == 0) {
      CBS cbs(env, lim);
      ok = cbs.search(starts, sol);
    } else {
      ECBS ecbs(env, w, lim);
      ok = ecbs.search(starts, sol);
    }
    rc = ok ? 1 : 0;
  } catch (const CapExceeded&) {
    rc = -1;
  }
  auto t1 = std::chrono::steady_clock::now();
  int64_t cost = 0, makespan = 0;
  if (rc == 1) {
    for (int i = 0; i < nAgents; ++i) {
      cost += sol[i].cost;
      makespan = std::max<int64_t>(makespan, sol[i].cost);
      int n = static_cast<int>(sol[i].states.size());
      if (pathLen) pathLen[i] = n;
      if (pathsXY)
        for (int k = 0; k < n && k < pathCap; ++k) {
          pathsXY[(static_cast<int64_t>(i) * pathCap + k) * 2 + 0] = sol[i].states[k].first.x;
          pathsXY[(static_cast<int64_t>(i) * pathCap + k) * 2 + 1] = sol[i].states[k].first.y;
        }
    }
  }
  stats[0] = cost;
  stats[1] = makespan;
  stats[2] = env.highLevelExpanded();
  stats[3] = env.lowLevelExpanded();
  stats[4] = std::chrono::duration_cast<std::chrono::nanoseconds>(t1 - t0).count();
  stats[5] = 0;
  return rc;
}

// The same for n instances of one shape (the arrays mrp_hl_generate_instances fills), one instance per thread on
// `nThreads` threads (the reference is single-threaded: this is "one ./ecbs process per core").
// perInst[k][0..5] = rc, cost, makespan, highLevelExpanded, lowLevelExpanded, elapsed_ns (search() only)
// Returns the wall-clock nanoseconds of the whole pool.
int64_t oracle_mapf_solve_batch(int algo, float w, int n, int dimx, int dimy, int nObst, const int32_t* obstXY,
                                int nAgents, const int32_t* startsXY, const int32_t* goalsXY, int64_t capTotal,
                                int nThreads, int64_t* perInst) {
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto work = [&]() {
    for (;;) {
      const int k = next.fetch_add(1, std::memory_order_relaxed);
      if (k >= n) return;
      int64_t st[6];
      const int rc = oracle_mapf_solve(algo, w, dimx, dimy, nObst, obstXY + static_cast<int64_t>(k) * nObst * 2, nAgents,
                                       startsXY + static_cast<int64_t>(k) * nAgents * 2,
                                       goalsXY + static_cast<int64_t>(k) * nAgents * 2, -1, capTotal, -1, st, nullptr,
                                       nullptr, 0);
      int64_t* o = perInst + static_cast<int64_t>(k) * 6;
      o[0] = rc;
      o[1] = st[0];
      o[2] = st[1];
      o[3] = st[2];
      o[4] = st[3];
      o[5] = st[4];
    }
  };
  if (nThreads <= 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nThreads; ++t) th.emplace_back(work);
    for (auto& x : th) x.join();
  }
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

// The same with a digest of every solved instance's schedule: perInst[k][0..6] = the six words above + the FNV-1a (64 bit)
// of the paths as include/mrp_hl.h mrp_hl_solution::schedule_digest defines it (0 when not solved).  A schedule longer
// than 1024 states per agent makes the instance's digest 1 (never equal to a real one in practice).
int64_t oracle_mapf_solve_batch_digest(int algo, float w, int n, int dimx, int dimy, int nObst, const int32_t* obstXY,
                                       int nAgents, const int32_t* startsXY, const int32_t* goalsXY, int64_t capTotal,
                                       int nThreads, int64_t* perInst) {
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto work = [&]() {
    const int cap = 1024;
    std::vector<int32_t> len(static_cast<size_t>(std::max(nAgents, 1)));
    std::vector<int32_t> xy(static_cast<size_t>(std::max(nAgents, 1)) * cap * 2);
    for (;;) {
      const int k = next.fetch_add(1, std::memory_order_relaxed);
      if (k >= n) return;
      int64_t st[6];
      const int rc = oracle_mapf_solve(algo, w, dimx, dimy, nObst, obstXY + static_cast<int64_t>(k) * nObst * 2, nAgents,
                                       startsXY + static_cast<int64_t>(k) * nAgents * 2,
                                       goalsXY + static_cast<int64_t>(k) * nAgents * 2, -1, capTotal, -1, st, len.data(),
                                       xy.data(), cap);
      int64_t* o = perInst + static_cast<int64_t>(k) * 7;
      o[0] = rc;
      o[1] = st[0];
      o[2] = st[1];
      o[3] = st[2];
      o[4] = st[3];
      o[5] = st[4];
      uint64_t h = 0;
      if (rc == 1) {
        h = 14695981039346656037ull;
        bool tooLong = false;
        for (int a = 0; a < nAgents; ++a) {
          if (len[a] > cap) tooLong = true;
          for (int q = 0; q < std::min(len[a], cap); ++q) {
            h = (h ^ (static_cast<uint32_t>(xy[(static_cast<size_t>(a) * cap + q) * 2]) & 0xFFu)) * 1099511628211ull;
            h = (h ^ (static_cast<uint32_t>(xy[(static_cast<size_t>(a) * cap + q) * 2 + 1]) & 0xFFu)) * 1099511628211ull;
          }
          h = (h ^ 0xFFu) * 1099511628211ull;
        }
        if (tooLong) h = 1;
      }
      o[6] = static_cast<int64_t>(h);
    }
  };
  if (nThreads <= 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nThreads; ++t) th.emplace_back(work);
    for (auto& x : th) x.join();
  }
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

// getFirstConflict (ecbs.cpp:401-452) + focalHeuristic (:315-350) of one solution given as flattened paths.
// out[0..9] = found, time, agent1, agent2, type (0 vertex / 1 edge), x1, y1, x2, y2, focalHeuristic
void oracle_conflict_scan(int nAgents, const int32_t* pathLen, const int32_t* pathXY, int32_t* out) {
  Environment env = makeEnv(256, 256, 0, nullptr, 0, nullptr);
  std::vector<Plan> sol(nAgents);
  int64_t off = 0;
  for (int a = 0; a < nAgents; ++a)
    for (int k = 0; k < pathLen[a]; ++k, ++off)
      sol[a].states.push_back(std::make_pair(State(k, pathXY[2 * off], pathXY[2 * off + 1]), k));
  Conflict c;
  std::memset(&c, 0, sizeof(c));
  const bool found = env.getFirstConflict(sol, c);
  out[0] = found ? 1 : 0;
  out[1] = found ? c.time : 0;
  out[2] = found ? static_cast<int32_t>(c.agent1) : 0;
  out[3] = found ? static_cast<int32_t>(c.agent2) : 0;
  out[4] = found ? (c.type == Conflict::Edge ? 1 : 0) : 0;
  out[5] = found ? c.x1 : 0;
  out[6] = found ? c.y1 : 0;
  out[7] = found && c.type == Conflict::Edge ? c.x2 : 0;
  out[8] = found && c.type == Conflict::Edge ? c.y2 : 0;
  out[9] = env.focalHeuristic(sol);
}

// One low-level search with everything explicit (the same information a mrp_ll_job carries).
// algo 0 = AStar (CBS low level), 1 = AStarEpsilon (ECBS low level).
// ctxLen[nCtx] = number of states of every agent's path in the CT node (0 = empty, skipped);
// ctxXY = concatenation of the paths, [sum(ctxLen)][2].
// out[0..3] = success, cost, fmin, n_states ; statesTXY [cap][3] ; actions [cap] (enum 0..4 = Up,Down,Left,Right,Wait)
// Return 0, or -1 if expansion cap exceeded.
// `initialCost`: AStar::search's third argument (a_star.hpp:63-64); AStarEpsilon::search has none, so it is only
// applied when algo == 0.
int oracle_ll_search_init(int algo, float w, int dimx, int dimy, int nObst, const int32_t* obstXY, int agentIdx,
                          int startX, int startY, int goalX, int goalY, int nVC, const int32_t* vc, int nEC,
                          const int32_t* ec, int nCtx, const int32_t* ctxLen, const int32_t* ctxXY,
                          int64_t capExpansions, int initialCost, int32_t* out, int64_t* expanded, int32_t* statesTXY,
                          int32_t* actions, int cap);

int oracle_ll_search(int algo, float w, int dimx, int dimy, int nObst, const int32_t* obstXY, int agentIdx,
                     int startX, int startY, int goalX, int goalY, int nVC, const int32_t* vc, int nEC,
                     const int32_t* ec, int nCtx, const int32_t* ctxLen, const int32_t* ctxXY, int64_t capExpansions,
                     int32_t* out, int64_t* expanded, int32_t* statesTXY, int32_t* actions, int cap) {
  return oracle_ll_search_init(algo, w, dimx, dimy, nObst, obstXY, agentIdx, startX, startY, goalX, goalY, nVC, vc, nEC,
                               ec, nCtx, ctxLen, ctxXY, capExpansions, 0, out, expanded, statesTXY, actions, cap);
}

int oracle_ll_search_init(int algo, float w, int dimx, int dimy, int nObst, const int32_t* obstXY, int agentIdx,
                          int startX, int startY, int goalX, int goalY, int nVC, const int32_t* vc, int nEC,
                          const int32_t* ec, int nCtx, const int32_t* ctxLen, const int32_t* ctxXY,
                          int64_t capExpansions, int initialCost, int32_t* out, int64_t* expanded, int32_t* statesTXY,
                          int32_t* actions, int cap) {
  int nGoals = std::max(nCtx, agentIdx + 1);
  std::vector<int32_t> goals(2 * nGoals, 0);
  goals[2 * agentIdx] = goalX;
  goals[2 * agentIdx + 1] = goalY;
  Environment env = makeEnv(dimx, dimy, nObst, obstXY, nGoals, goals.data());
  Constraints cons;
  for (int i = 0; i < nVC; ++i) cons.vertex.insert(VertexConstraint{vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]});
  for (int i = 0; i < nEC; ++i)
    cons.edge.insert(EdgeConstraint{ec[5 * i], ec[5 * i + 1], ec[5 * i + 2], ec[5 * i + 3], ec[5 * i + 4]});
  std::vector<Plan> ctx(nCtx);
  int64_t off = 0;
  for (int a = 0; a < nCtx; ++a) {
    for (int k = 0; k < ctxLen[a]; ++k, ++off)
      ctx[a].states.push_back(std::make_pair(State(k, ctxXY[2 * off], ctxXY[2 * off + 1]), k));
    ctx[a].cost = ctxLen[a] ? ctxLen[a] - 1 : 0;
    ctx[a].fmin = 0;
  }
  // reuse the adapters through tiny one-agent HL shells
  struct Shell {
    Environment& env;
    const std::vector<Plan>& ctx;
    int64_t cap;
    int admissibleHeuristic(const State& s) { return env.admissibleHeuristic(s); }
    int focalStateHeuristic(const State& s, int g) { return env.focalStateHeuristic(s, g, ctx); }
    int focalTransitionHeuristic(const State& a, const State& b, int ga, int gb) {
      return env.focalTransitionHeuristic(a, b, ga, gb, ctx);
    }
    bool isSolution(const State& s) { return env.isSolution(s); }
    void getNeighbors(const State& s, std::vector<Neighbor<State, Action, int>>& n) { env.getNeighbors(s, n); }
    void onExpandNode(const State& s, int f, int g) {
      env.onExpandLowLevelNode(s, f, g);
      if (cap >= 0 && env.lowLevelExpanded() > cap) throw CapExceeded();
    }
    void onDiscover(const State&, int, int) {}
  };
  env.setLowLevelContext(agentIdx, &cons);
  Shell shell{env, ctx, capExpansions};
  Plan plan;
  bool ok = false;
  int rc = 0;
  try {
    if (algo == 0) {
      AStar<State, Action, int, Shell, StateHash> ll(shell);
      ok = ll.search(State(0, startX, startY), plan, initialCost);
    } else {
      AStarEpsilon<State, Action, int, Shell, StateHash> ll(shell, w);
      ok = ll.search(State(0, startX, startY), plan);
    }
  } catch (const CapExceeded&) {
    rc = -1;
  }
  *expanded = env.lowLevelExpanded();
  out[0] = ok ? 1 : 0;
  out[1] = ok ? plan.cost : 0;
  out[2] = ok ? plan.fmin : 0;
  int n = ok ? static_cast<int>(plan.states.size()) : 0;
  out[3] = n;
  for (int k = 0; k < n && k < cap; ++k) {
    statesTXY[3 * k + 0] = plan.states[k].first.time;
    statesTXY[3 * k + 1] = plan.states[k].first.x;
    statesTXY[3 * k + 2] = plan.states[k].first.y;
    if (k + 1 < n) actions[k] = actionCode(plan.actions[k].first);
  }
  return rc;
}

// Run CBS/ECBS with the low-level recorder on and serialise every low-level call into `buf` (int32 words):
//   per call: agent, success, cost, fmin, expanded, nVC, nEC, nCtx, nStates,
//             vc[nVC][3], ec[nEC][5], ctxLen[nCtx], ctxXY[sum][2], statesXY[nStates][2]
// Returns the number of words needed (call again with a bigger buffer if > bufWords); *nCalls = #calls.
int64_t oracle_mapf_record(int algo, float w, int dimx, int dimy, int nObst, const int32_t* obstXY, int nAgents,
                           const int32_t* startsXY, const int32_t* goalsXY, int64_t capTotal, int32_t* buf,
                           int64_t bufWords, int32_t* nCalls, int64_t* stats) {
  Environment env = makeEnv(dimx, dimy, nObst, obstXY, nAgents, goalsXY);
  std::vector<State> starts;
  for (int i = 0; i < nAgents; ++i) starts.emplace_back(0, startsXY[2 * i], startsXY[2 * i + 1]);
  Limits lim;
  lim.maxLowLevelExpansionsTotal = capTotal;
  std::vector<LowLevelCall> calls;
  std::vector<Plan> sol;
  int rc = 0;
  try {
    if (algo == 0) {
      CBS cbs(env, lim);
      cbs.recorder = &calls;
      rc = cbs.search(starts, sol) ? 1 : 0;
    } else {
      ECBS ecbs(env, w, lim);
      ecbs.recorder = &calls;
      rc = ecbs.search(starts, sol) ? 1 : 0;
    }
  } catch (const CapExceeded&) {
    rc = -1;
  }
  int64_t cost = 0;
  if (rc == 1)
    for (auto& p : sol) cost += p.cost;
  stats[0] = cost;
  stats[1] = rc;
  stats[2] = env.highLevelExpanded();
  stats[3] = env.lowLevelExpanded();
  std::vector<int32_t> words;
  for (const auto& c : calls) {
    words.push_back(static_cast<int32_t>(c.agent));
    words.push_back(c.success ? 1 : 0);
    words.push_back(c.success ? c.result.cost : 0);
    words.push_back(c.success ? c.result.fmin : 0);
    words.push_back(static_cast<int32_t>(c.expanded));
    words.push_back(static_cast<int32_t>(c.constraints.vertex.size()));
    words.push_back(static_cast<int32_t>(c.constraints.edge.size()));
    words.push_back(static_cast<int32_t>(c.solutionContext.size()));
    words.push_back(c.success ? static_cast<int32_t>(c.result.states.size()) : 0);
    for (const auto& v : c.constraints.vertex) {
      words.push_back(v.time); words.push_back(v.x); words.push_back(v.y);
    }
    for (const auto& e : c.constraints.edge) {
      words.push_back(e.time); words.push_back(e.x1); words.push_back(e.y1); words.push_back(e.x2); words.push_back(e.y2);
    }
    for (std::size_t a = 0; a < c.solutionContext.size(); ++a)
      words.push_back(a == c.agent ? 0 : static_cast<int32_t>(c.solutionContext[a].states.size()));
    for (std::size_t a = 0; a < c.solutionContext.size(); ++a) {
      if (a == c.agent) continue;
      for (const auto& s : c.solutionContext[a].states) {
        words.push_back(s.first.x); words.push_back(s.first.y);
      }
    }
    if (c.success)
      for (const auto& s : c.result.states) {
        words.push_back(s.first.x); words.push_back(s.first.y);
      }
  }
  *nCalls = static_cast<int32_t>(calls.size());
  int64_t need = static_cast<int64_t>(words.size());
  if (need <= bufWords && buf) std::memcpy(buf, words.data(), need * sizeof(int32_t));
  return need;
}

// ---- MutableBinaryHeap self-check hook: replays an op sequence and returns the array layout ----------
// ops: 0 push(key) ; 1 pop ; 2 erase(handle=arg) ; 3 increase(handle=arg, newKey=arg2) ; keys compare as ints
// (max-heap).  walkOut receives the ordered-walk (real std::priority_queue) handle order; walkOut2 the explicit one.
int oracle_heap_replay(int nOps, const int32_t* ops, int32_t* layoutOut, int32_t* walkOut, int32_t* walkOut2) {
  struct KeyLess {
    bool operator()(const int& a, const int& b) const { return a < b; }
  };
  MutableBinaryHeap<int, KeyLess> h;
  for (int i = 0; i < nOps; ++i) {
    int op = ops[3 * i], a = ops[3 * i + 1], b = ops[3 * i + 2];
    if (op == 0) h.push(a);
    else if (op == 1) h.pop();
    else if (op == 2) h.erase(static_cast<std::size_t>(a));
    else if (op == 3) { h[static_cast<std::size_t>(a)] = b; h.increase(static_cast<std::size_t>(a)); }
  }
  int n = static_cast<int>(h.size());
  for (int i = 0; i < n; ++i) layoutOut[i] = static_cast<int32_t>(h.array()[i]);
  int k = 0;
  h.orderedWalk([&](std::size_t hd) { walkOut[k++] = static_cast<int32_t>(hd); return true; });
  k = 0;
  h.orderedWalkExplicit([&](std::size_t hd) { walkOut2[k++] = static_cast<int32_t>(hd); return true; });
  return n;
}

// ---- config 1: 2-D A* on a text map (example/a_star.cpp) ------------------------------------------------
// map: rows of chars, '#' = obstacle; returns number of states in the schedule (0 = planning failed).
int oracle_astar_2d(int dimx, int dimy, const uint8_t* obstacleMask, int sx, int sy, int gx, int gy,
                    int32_t* statesXY, int cap, int64_t* expanded) {
  return grid2d::solve(dimx, dimy, obstacleMask, sx, sy, gx, gy, statesXY, cap, expanded);
}

// ---- config 5: prioritized SIPP (example/mapf_prioritized_sipp.cpp) -------------------------------------
int oracle_prioritized_sipp(int dimx, int dimy, int nObst, const int32_t* obstXY, int nAgents,
                            const int32_t* startsXY, const int32_t* goalsXY, int64_t* stats, int32_t* planned,
                            int32_t* nStates, int32_t* statesXYT, int cap) {
  return sipp::prioritizedPlan(dimx, dimy, nObst, obstXY, nAgents, startsXY, goalsXY, stats, planned, nStates,
                               statesXYT, cap);
}

// The same for n instances of one shape, one instance per thread on `nThreads` threads; timed inside (no wrapper).
// perInst[k][0..3] = n_planned, cost, expanded, elapsed_ns.  Returns the wall-clock nanoseconds of the pool.
int64_t oracle_prioritized_sipp_batch(int n, int dimx, int dimy, int nObst, const int32_t* obstXY, int nAgents,
                                      const int32_t* startsXY, const int32_t* goalsXY, int nThreads, int64_t* perInst) {
  std::atomic<int> next(0);
  auto t0 = std::chrono::steady_clock::now();
  auto work = [&]() {
    std::vector<int32_t> planned(nAgents), nStates(nAgents);
    for (;;) {
      const int k = next.fetch_add(1, std::memory_order_relaxed);
      if (k >= n) return;
      int64_t st[4] = {0, 0, 0, 0};
      auto a = std::chrono::steady_clock::now();
      const int np = sipp::prioritizedPlan(dimx, dimy, nObst, obstXY + static_cast<int64_t>(k) * nObst * 2, nAgents,
                                           startsXY + static_cast<int64_t>(k) * nAgents * 2,
                                           goalsXY + static_cast<int64_t>(k) * nAgents * 2, st, planned.data(),
                                           nStates.data(), nullptr, 0);
      auto b = std::chrono::steady_clock::now();
      int64_t* o = perInst + static_cast<int64_t>(k) * 4;
      o[0] = np;
      o[1] = st[0];
      o[2] = st[1];
      o[3] = std::chrono::duration_cast<std::chrono::nanoseconds>(b - a).count();
    }
  };
  if (nThreads <= 1) {
    work();
  } else {
    std::vector<std::thread> th;
    for (int t = 0; t < nThreads; ++t) th.emplace_back(work);
    for (auto& x : th) x.join();
  }
  return std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
}

// ---- example/sipp.cpp: one agent, collision intervals given as [n][4] = x, y, start, end ----------------
// `startTime`: SIPP::search's fourth argument (sipp.hpp:92); costFmin[0..1] receive solution.cost / solution.fmin.
int oracle_sipp_single_at(int dimx, int dimy, int nObst, const int32_t* obstXY, int sx, int sy, int gx, int gy, int nCI,
                          const int32_t* ci, int startTime, int32_t* statesXYT, int cap, int64_t* expanded,
                          int32_t* costFmin);

int oracle_sipp_single(int dimx, int dimy, int nObst, const int32_t* obstXY, int sx, int sy, int gx, int gy, int nCI,
                       const int32_t* ci, int32_t* statesXYT, int cap, int64_t* expanded) {
  return oracle_sipp_single_at(dimx, dimy, nObst, obstXY, sx, sy, gx, gy, nCI, ci, 0, statesXYT, cap, expanded, nullptr);
}

int oracle_sipp_single_at(int dimx, int dimy, int nObst, const int32_t* obstXY, int sx, int sy, int gx, int gy, int nCI,
                          const int32_t* ci, int startTime, int32_t* statesXYT, int cap, int64_t* expanded,
                          int32_t* costFmin) {
  std::vector<uint8_t> mask(static_cast<std::size_t>(dimx) * dimy, 0);
  for (int i = 0; i < nObst; ++i) mask[obstXY[2 * i + 1] * dimx + obstXY[2 * i]] = 1;
  sipp::GridEnv env(dimx, dimy, mask, sipp::Cell{gx, gy});
  sipp::Sipp planner(env);
  std::map<sipp::Cell, std::vector<sipp::Interval>> byCell;
  std::vector<sipp::Cell> order;  // sipp.cpp applies them in file order, one call per location entry
  for (int i = 0; i < nCI; ++i) {
    sipp::Cell c{ci[4 * i], ci[4 * i + 1]};
    if (!byCell.count(c)) order.push_back(c);
    byCell[c].push_back(sipp::Interval{ci[4 * i + 2], ci[4 * i + 3]});
  }
  for (const auto& c : order) planner.setCollisionIntervals(c, byCell[c]);
  sipp::TimedPlan sol;
  bool ok = planner.search(sipp::Cell{sx, sy}, sol, startTime);
  if (expanded) *expanded = planner.expanded();
  if (costFmin) {
    costFmin[0] = sol.cost;
    costFmin[1] = sol.fmin;
  }
  if (!ok) return 0;
  int n = static_cast<int>(sol.states.size());
  for (int k = 0; k < n && k < cap; ++k) {
    statesXYT[3 * k] = sol.states[k].first.x;
    statesXYT[3 * k + 1] = sol.states[k].first.y;
    statesXYT[3 * k + 2] = sol.states[k].second;
  }
  return n;
}

// ---- task-assignment callers' low level (ta_restated.hpp) ------------------------------------------
// One AStar::search over example/cbs_ta.cpp's Environment.  hasGoal = 0: the agent has no task (cbs_ta.cpp:283-319).
// out[0..3] = success, cost, fmin, n_states; statesTXY [cap][3], actions [cap], actionCosts [cap] (Wait at the goal: 0).
// Returns 0, or -1 when the expansion cap was exceeded.
int oracle_ta_ll_search(int dimx, int dimy, int nObst, const int32_t* obstXY, int startX, int startY, int hasGoal, int goalX,
                        int goalY, int nVC, const int32_t* vc, int nEC, const int32_t* ec, int64_t capExpansions,
                        int32_t* out, int64_t* expanded, int32_t* statesTXY, int32_t* actions, int32_t* actionCosts, int cap) {
  std::unordered_set<Cell, CellHash> obst;
  for (int i = 0; i < nObst; ++i) obst.insert(Cell{obstXY[2 * i], obstXY[2 * i + 1]});
  Constraints cons;
  for (int i = 0; i < nVC; ++i) cons.vertex.insert(VertexConstraint{vc[3 * i], vc[3 * i + 1], vc[3 * i + 2]});
  for (int i = 0; i < nEC; ++i)
    cons.edge.insert(EdgeConstraint{ec[5 * i], ec[5 * i + 1], ec[5 * i + 2], ec[5 * i + 3], ec[5 * i + 4]});
  ta::Environment env(dimx, dimy, obst);
  const Cell goal{goalX, goalY};
  Plan plan;
  bool ok = false;
  int rc = 0;
  try {
    ok = ta::lowLevelSearch(env, 0, cons, hasGoal ? &goal : nullptr, obst, dimx, dimy, State(0, startX, startY), plan,
                            capExpansions);
  } catch (const CapExceeded&) {
    rc = -1;
  }
  *expanded = env.m_llExpandedThisSearch;
  out[0] = ok ? 1 : 0;
  out[1] = ok ? plan.cost : 0;
  out[2] = ok ? plan.fmin : 0;
  const int n = ok ? static_cast<int>(plan.states.size()) : 0;
  out[3] = n;
  for (int k = 0; k < n && k < cap; ++k) {
    statesTXY[3 * k + 0] = plan.states[k].first.time;
    statesTXY[3 * k + 1] = plan.states[k].first.x;
    statesTXY[3 * k + 2] = plan.states[k].first.y;
    if (k + 1 < n) {
      actions[k] = actionCode(plan.actions[k].first);
      actionCosts[k] = plan.actions[k].second;
    }
  }
  return rc;
}

// cbs_ta.hpp:85-215 for ONE fixed assignment (ta::cbsFixedTasks): tasksXY[i] = (-1, -1): agent i has no task.
// stats[0..2] = solved, cost, highLevelExpanded; endTXY [nAgents][3] = last state of every path.
// Every low-level call is serialised into buf (int32 words) as
//   agent, hasTask, taskX, taskY, success, cost, fmin, expanded, nVC, nEC, nStates, vc[nVC][3], ec[nEC][5],
//   statesTXY[nStates][3], actionCosts[nStates - 1 (0 if nStates == 0)]
// Returns the words needed (call again with a bigger buffer if > bufWords); *nCalls = number of calls.
int64_t oracle_ta_cbs_fixed(int dimx, int dimy, int nObst, const int32_t* obstXY, int nAgents, const int32_t* startsXY,
                            const int32_t* tasksXY, int64_t* stats, int32_t* endTXY, int32_t* buf, int64_t bufWords,
                            int32_t* nCalls) {
  std::unordered_set<Cell, CellHash> obst;
  for (int i = 0; i < nObst; ++i) obst.insert(Cell{obstXY[2 * i], obstXY[2 * i + 1]});
  std::vector<State> starts;
  std::vector<Cell> taskCells(nAgents);
  std::vector<const Cell*> tasks(nAgents, nullptr);
  for (int i = 0; i < nAgents; ++i) {
    starts.push_back(State(0, startsXY[2 * i], startsXY[2 * i + 1]));
    if (tasksXY[2 * i] >= 0) {
      taskCells[i] = Cell{tasksXY[2 * i], tasksXY[2 * i + 1]};
      tasks[i] = &taskCells[i];
    }
  }
  std::vector<Plan> sol;
  int cost = 0;
  std::vector<ta::LowLevelCall> rec;
  const bool ok = ta::cbsFixedTasks(dimx, dimy, obst, starts, tasks, sol, cost, &rec);
  stats[0] = ok ? 1 : 0;
  stats[1] = ok ? cost : 0;
  stats[2] = 0;
  if (ok)
    for (int i = 0; i < nAgents; ++i) {
      const State& s = sol[i].states.back().first;
      endTXY[3 * i] = s.time;
      endTXY[3 * i + 1] = s.x;
      endTXY[3 * i + 2] = s.y;
    }
  int64_t w = 0;
  auto put = [&](int32_t v) {
    if (w < bufWords) buf[w] = v;
    ++w;
  };
  for (const auto& c : rec) {
    const int nst = c.ok ? static_cast<int>(c.plan.states.size()) : 0;
    put(static_cast<int32_t>(c.agent)); put(c.hasTask ? 1 : 0); put(c.task.x); put(c.task.y); put(c.ok ? 1 : 0);
    put(c.ok ? c.plan.cost : 0); put(c.ok ? c.plan.fmin : 0); put(static_cast<int32_t>(c.expanded));
    put(static_cast<int32_t>(c.constraints.vertex.size())); put(static_cast<int32_t>(c.constraints.edge.size())); put(nst);
    for (const auto& v : c.constraints.vertex) { put(v.time); put(v.x); put(v.y); }
    for (const auto& e : c.constraints.edge) { put(e.time); put(e.x1); put(e.y1); put(e.x2); put(e.y2); }
    for (int k = 0; k < nst; ++k) { put(c.plan.states[k].first.time); put(c.plan.states[k].first.x); put(c.plan.states[k].first.y); }
    for (int k = 0; k + 1 < nst; ++k) put(c.plan.actions[k].second);
  }
  *nCalls = static_cast<int32_t>(rec.size());
  return w;
}

#ifdef ORACLE_SEARCH_STATS
// Diagnostic build (scripts/search_stats.py): one row of 13 counters per A*-epsilon search since the last call.
int64_t oracle_search_stats_take(int64_t* out, int64_t capRows);
#endif
}  // extern "C"

#ifdef ORACLE_SEARCH_STATS
#include <mutex>
namespace {
std::mutex g_statMu;
std::vector<oracle::SearchStats> g_statRows;
}  // namespace
namespace oracle {
namespace mapf {
void searchStatsSink(const SearchStats& st) {
  std::lock_guard<std::mutex> lock(g_statMu);
  g_statRows.push_back(st);
}
}  // namespace mapf
}  // namespace oracle
extern "C" int64_t oracle_search_stats_take(int64_t* out, int64_t capRows) {
  std::lock_guard<std::mutex> lock(g_statMu);
  const int64_t n = std::min<int64_t>(capRows, static_cast<int64_t>(g_statRows.size()));
  for (int64_t k = 0; k < n; ++k) {
    const oracle::SearchStats& s = g_statRows[k];
    const long v[13] = {s.expansions, s.maxOpen, s.maxFocal, s.maxG, s.maxF, s.maxFocalH, s.walks, s.visited,
                        s.walksEmpty, s.visitedEmpty, s.walksDistinct, s.visitedDistinct, s.bandNodes};
    for (int q = 0; q < 13; ++q) out[k * 13 + q] = v[q];
  }
  const int64_t total = static_cast<int64_t>(g_statRows.size());
  g_statRows.clear();
  return total;
}
#endif
