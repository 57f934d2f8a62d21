// ORACLE — TEST INFRASTRUCTURE ONLY (see heap_restated.hpp header).
//
// CPU restatement of the reference's two low-level searches, written against the same Environment concept
// (admissibleHeuristic / isSolution / getNeighbors / onExpandNode / onDiscover [+ focalStateHeuristic /
// focalTransitionHeuristic]) so the reference's example Environments can be restated 1:1 on top.
//   AStar        follows include/libMultiRobotPlanning/a_star.hpp:63-161 (Node order :168-179)
//   AStarEpsilon follows include/libMultiRobotPlanning/a_star_epsilon.hpp:86-285
//                (open order :312-323, focal order :346-366, float bound m_w :386)
// Value types follow neighbor.hpp:14-25 and planresult.hpp:18-27.
#pragma once
#include <algorithm>
#include <tuple>
#include <unordered_map>
#include <unordered_set>
#include <utility>
#include <vector>

#include "heap_restated.hpp"

// The heap type of every search and conflict-tree loop of the oracle: the restated one, unless a cross-check build
// (boost_heap_adapter.hpp, `make liboracle_boost.so`) names the real Boost.Heap adapter.
#ifndef ORACLE_HEAP
#define ORACLE_HEAP MutableBinaryHeap
#endif

namespace oracle {

#ifdef ORACLE_SEARCH_STATS
// Diagnostic build only (scripts/search_stats.py): what one AStarEpsilon::search did, for sizing the device tiers.
struct SearchStats {
  long expansions = 0, maxOpen = 0, maxFocal = 0, maxG = 0, maxF = 0, maxFocalH = 0;
  long walks = 0, visited = 0;                 // ordered walks, nodes they visited
  long walksEmpty = 0, visitedEmpty = 0;       // walks whose band (old * w, new * w] holds no open node
  long walksDistinct = 0, visitedDistinct = 0; // non-empty band, all band nodes have pairwise distinct (f, g)
  long bandNodes = 0;
};
inline thread_local SearchStats* g_stats = nullptr;
#define ORACLE_STAT(x) do { if (g_stats) { x; } } while (0)
#else
#define ORACLE_STAT(x) do { } while (0)
#endif

template <typename State, typename Action, typename Cost>
struct Neighbor {  // neighbor.hpp:14-25
  Neighbor(const State& s, const Action& a, Cost c) : state(s), action(a), cost(c) {}
  State state;
  Action action;
  Cost cost;
};

template <typename State, typename Action, typename Cost>
struct PlanResult {  // planresult.hpp:18-27
  std::vector<std::pair<State, Cost>> states;
  std::vector<std::pair<Action, Cost>> actions;
  Cost cost;
  Cost fmin;
};

// ---------------------------------------------------------------------------------------------------
template <typename State, typename Action, typename Cost, typename Environment,
          typename StateHasher = std::hash<State>>
class AStar {
  struct OpenRec {
    State state;
    Cost f, g;
  };
  struct OpenLess {  // a_star.hpp:168-179 — max-heap, so "less" == worse: higher f, then lower g
    bool operator()(const OpenRec& a, const OpenRec& b) const {
      if (a.f != b.f) return a.f > b.f;
      return a.g < b.g;
    }
  };
  typedef ORACLE_HEAP<OpenRec, OpenLess> Open;

 public:
  explicit AStar(Environment& env) : m_env(env) {}

  bool search(const State& start, PlanResult<State, Action, Cost>& out, Cost initialCost = 0) {
    out.states.clear();
    out.states.push_back(std::make_pair(start, Cost(0)));
    out.actions.clear();
    out.cost = 0;

    Open open;
    std::unordered_map<State, typename Open::handle_type, StateHasher> inOpen;
    std::unordered_set<State, StateHasher> closed;
    struct Parent {
      State from;
      Action action;
      Cost stepCost, g;
    };
    std::unordered_map<State, Parent, StateHasher> parentOf;

    inOpen.emplace(start, open.push(OpenRec{start, m_env.admissibleHeuristic(start), initialCost}));

    std::vector<Neighbor<State, Action, Cost>> succ;
    succ.reserve(10);

    while (!open.empty()) {
      OpenRec cur = open.top();
      m_env.onExpandNode(cur.state, cur.f, cur.g);

      if (m_env.isSolution(cur.state)) {
        out.states.clear();
        out.actions.clear();
        auto it = parentOf.find(cur.state);
        while (it != parentOf.end()) {
          out.states.push_back(std::make_pair(it->first, it->second.g));
          out.actions.push_back(std::make_pair(it->second.action, it->second.stepCost));
          it = parentOf.find(it->second.from);
        }
        out.states.push_back(std::make_pair(start, initialCost));
        std::reverse(out.states.begin(), out.states.end());
        std::reverse(out.actions.begin(), out.actions.end());
        out.cost = cur.g;
        out.fmin = cur.f;
        return true;
      }

      open.pop();
      inOpen.erase(cur.state);
      closed.insert(cur.state);

      succ.clear();
      m_env.getNeighbors(cur.state, succ);
      for (const auto& nb : succ) {
        if (closed.find(nb.state) != closed.end()) continue;
        Cost g2 = cur.g + nb.cost;
        auto it = inOpen.find(nb.state);
        if (it == inOpen.end()) {
          Cost f2 = g2 + m_env.admissibleHeuristic(nb.state);
          inOpen.emplace(nb.state, open.push(OpenRec{nb.state, f2, g2}));
          m_env.onDiscover(nb.state, f2, g2);
        } else {
          auto h = it->second;
          if (g2 >= open[h].g) continue;  // not an improvement: parent stays as is
          Cost delta = open[h].g - g2;
          open[h].g = g2;
          open[h].f -= delta;
          open.increase(h);
          m_env.onDiscover(nb.state, open[h].f, open[h].g);
        }
        parentOf.erase(nb.state);
        parentOf.emplace(nb.state, Parent{cur.state, nb.action, nb.cost, g2});
      }
    }
    return false;
  }

 private:
  Environment& m_env;
};

// ---------------------------------------------------------------------------------------------------
template <typename State, typename Action, typename Cost, typename Environment,
          typename StateHasher = std::hash<State>>
class AStarEpsilon {
  struct OpenRec {
    State state;
    Cost f, g, focalH;
  };
  struct OpenLess {  // a_star_epsilon.hpp:312-323
    bool operator()(const OpenRec& a, const OpenRec& b) const {
      if (a.f != b.f) return a.f > b.f;
      return a.g < b.g;
    }
  };
  typedef ORACLE_HEAP<OpenRec, OpenLess> Open;
  typedef typename Open::handle_type OpenHandle;
  struct FocalLess {  // a_star_epsilon.hpp:346-366 — compares the open records behind two handles
    const Open* open;
    bool operator()(const OpenHandle& h1, const OpenHandle& h2) const {
      const OpenRec& a = (*open)[h1];
      const OpenRec& b = (*open)[h2];
      if (a.focalH != b.focalH) return a.focalH > b.focalH;
      if (a.f != b.f) return a.f > b.f;
      return a.g < b.g;
    }
  };
  typedef ORACLE_HEAP<OpenHandle, FocalLess> Focal;

 public:
  AStarEpsilon(Environment& env, float w) : m_env(env), m_w(w) {}

  bool search(const State& start, PlanResult<State, Action, Cost>& out) {
    out.states.clear();
    out.states.push_back(std::make_pair(start, Cost(0)));
    out.actions.clear();
    out.cost = 0;

    Open open;
    Focal focal(FocalLess{&open});
    std::unordered_map<State, OpenHandle, StateHasher> inOpen;
    std::unordered_set<State, StateHasher> closed;
    struct Parent {
      State from;
      Action action;
      Cost stepCost, g;
    };
    std::unordered_map<State, Parent, StateHasher> parentOf;

    OpenHandle h0 = open.push(OpenRec{start, m_env.admissibleHeuristic(start), 0, 0});
    inOpen.emplace(start, h0);
    focal.push(h0);

    std::vector<Neighbor<State, Action, Cost>> succ;
    succ.reserve(10);

    Cost bestF = open[h0].f;

    while (!open.empty()) {
      {  // incremental focal update, a_star_epsilon.hpp:134-154. NB: int * float -> binary32 products.
        Cost oldBestF = bestF;
        bestF = open.top().f;
        if (bestF > oldBestF) {
#ifdef ORACLE_SEARCH_STATS
          long vis = 0;
          std::vector<std::pair<Cost, Cost>> band;
#endif
          open.orderedWalk([&](OpenHandle h) {
            Cost val = open[h].f;
            ORACLE_STAT(vis += 1);
            if (val > oldBestF * m_w && val <= bestF * m_w) {
              focal.push(h);
              ORACLE_STAT(band.push_back(std::make_pair(open[h].f, open[h].g)));
            }
            if (val > bestF * m_w) return false;
            return true;
          });
#ifdef ORACLE_SEARCH_STATS
          if (g_stats) {
            g_stats->walks += 1;
            g_stats->visited += vis;
            g_stats->bandNodes += static_cast<long>(band.size());
            if (band.empty()) {
              g_stats->walksEmpty += 1;
              g_stats->visitedEmpty += vis;
            } else {
              std::sort(band.begin(), band.end());
              bool distinct = true;
              for (size_t q = 1; q < band.size(); ++q) distinct = distinct && band[q] != band[q - 1];
              if (distinct) {
                g_stats->walksDistinct += 1;
                g_stats->visitedDistinct += vis;
              }
            }
          }
#endif
        }
      }

      OpenHandle curH = focal.top();
      OpenRec cur = open[curH];
      m_env.onExpandNode(cur.state, cur.f, cur.g);
      ORACLE_STAT(g_stats->expansions += 1; g_stats->maxOpen = std::max<long>(g_stats->maxOpen, (long)open.size());
                  g_stats->maxFocal = std::max<long>(g_stats->maxFocal, (long)focal.size());
                  g_stats->maxG = std::max<long>(g_stats->maxG, (long)cur.g));

      if (m_env.isSolution(cur.state)) {
        out.states.clear();
        out.actions.clear();
        auto it = parentOf.find(cur.state);
        while (it != parentOf.end()) {
          out.states.push_back(std::make_pair(it->first, it->second.g));
          out.actions.push_back(std::make_pair(it->second.action, it->second.stepCost));
          it = parentOf.find(it->second.from);
        }
        out.states.push_back(std::make_pair(start, Cost(0)));
        std::reverse(out.states.begin(), out.states.end());
        std::reverse(out.actions.begin(), out.actions.end());
        out.cost = cur.g;
        out.fmin = open.top().f;
        return true;
      }

      focal.pop();
      open.erase(curH);
      inOpen.erase(cur.state);
      closed.insert(cur.state);

      succ.clear();
      m_env.getNeighbors(cur.state, succ);
      for (const auto& nb : succ) {
        if (closed.find(nb.state) != closed.end()) continue;
        Cost g2 = cur.g + nb.cost;
        auto it = inOpen.find(nb.state);
        if (it == inOpen.end()) {
          Cost f2 = g2 + m_env.admissibleHeuristic(nb.state);
          Cost fh2 = cur.focalH + m_env.focalStateHeuristic(nb.state, g2) +
                     m_env.focalTransitionHeuristic(cur.state, nb.state, cur.g, g2);
          OpenHandle h = open.push(OpenRec{nb.state, f2, g2, fh2});
          ORACLE_STAT(g_stats->maxF = std::max<long>(g_stats->maxF, (long)f2);
                      g_stats->maxFocalH = std::max<long>(g_stats->maxFocalH, (long)fh2));
          if (f2 <= bestF * m_w) focal.push(h);
          inOpen.emplace(nb.state, h);
          m_env.onDiscover(nb.state, f2, g2);
        } else {
          OpenHandle h = it->second;
          if (g2 >= open[h].g) continue;
          Cost lastG = open[h].g;
          Cost lastF = open[h].f;
          Cost delta = lastG - g2;
          open[h].g = g2;
          open[h].f -= delta;
          open.increase(h);
          m_env.onDiscover(nb.state, open[h].f, open[h].g);
          // focal entries are never re-keyed and the node keeps its old focalH (a_star_epsilon.hpp:258-269)
          if (open[h].f <= bestF * m_w && lastF > bestF * m_w) focal.push(h);
        }
        parentOf.erase(nb.state);
        parentOf.emplace(nb.state, Parent{cur.state, nb.action, nb.cost, g2});
      }
    }
    return false;
  }

 private:
  Environment& m_env;
  float m_w;
};

}  // namespace oracle
