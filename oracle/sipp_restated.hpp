// ORACLE — TEST INFRASTRUCTURE ONLY (see heap_restated.hpp header).
// Config 5: restatement of SIPP (include/libMultiRobotPlanning/sipp.hpp:91-134 search + post-processing,
// :171-313 SIPPEnvironment) on the grid Environment of example/mapf_prioritized_sipp.cpp:82-155 and of the
// sequential prioritized loop of its main() (:214-270).  Also serves example/sipp.cpp (single agent with given
// collision intervals, :178-200).
#pragma once
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <map>
#include <unordered_map>
#include <vector>

#include "search_restated.hpp"

namespace oracle {
namespace sipp {

struct Cell {
  int x, y;
  bool operator==(const Cell& o) const { return x == o.x && y == o.y; }
  bool operator!=(const Cell& o) const { return !(*this == o); }
  bool operator<(const Cell& o) const { return x != o.x ? x < o.x : y < o.y; }  // std::tie(x,y) order
};
struct CellHash {
  std::size_t operator()(const Cell& s) const {
    std::size_t seed = 0;
    seed ^= std::hash<int>()(s.x) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    seed ^= std::hash<int>()(s.y) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
  }
};
enum class Action { Up, Down, Left, Right, Wait };

struct Interval {  // sipp.hpp:68-78; ordering by start only
  int start, end;
};

// the user Environment of mapf_prioritized_sipp.cpp:82-155 / sipp.cpp
class GridEnv {
 public:
  GridEnv(int dimx, int dimy, const std::vector<uint8_t>& mask, Cell goal)
      : m_dimx(dimx), m_dimy(dimy), m_mask(mask), m_goal(goal) {}
  int admissibleHeuristic(const Cell& s) const { return std::abs(s.x - m_goal.x) + std::abs(s.y - m_goal.y); }
  bool isSolution(const Cell& s) const { return s == m_goal; }
  void motions(const Cell& s, std::vector<std::pair<Cell, Action>>& out) const {  // Up, Down, Left, Right
    out.clear();
    const Cell cand[4] = {{s.x, s.y + 1}, {s.x, s.y - 1}, {s.x - 1, s.y}, {s.x + 1, s.y}};
    const Action act[4] = {Action::Up, Action::Down, Action::Left, Action::Right};
    for (int k = 0; k < 4; ++k)
      if (valid(cand[k])) out.emplace_back(cand[k], act[k]);
  }
  bool valid(const Cell& s) const {
    return s.x >= 0 && s.x < m_dimx && s.y >= 0 && s.y < m_dimy && !m_mask[s.y * m_dimx + s.x];
  }

 private:
  int m_dimx, m_dimy;
  const std::vector<uint8_t>& m_mask;
  Cell m_goal;
};

struct SState {  // sipp.hpp:138-153
  Cell cell;
  std::size_t interval;
  bool operator==(const SState& o) const { return cell == o.cell && interval == o.interval; }
};
struct SStateHash {
  std::size_t operator()(const SState& s) const {
    std::size_t seed = 0;
    seed ^= CellHash()(s.cell) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    seed ^= std::hash<std::size_t>()(s.interval) + 0x9e3779b9 + (seed << 6) + (seed >> 2);
    return seed;
  }
};
struct SAction {  // sipp.hpp:164-169
  Action action;
  int time;
};

class SippEnv {  // sipp.hpp:172-313
 public:
  explicit SippEnv(GridEnv& env) : m_env(env) {}
  int admissibleHeuristic(const SState& s) { return m_env.admissibleHeuristic(s.cell); }
  bool isSolution(const SState& s) {
    return m_env.isSolution(s.cell) && safe(s.cell).at(s.interval).end == INT_MAX;
  }
  void getNeighbors(const SState& s, std::vector<Neighbor<SState, SAction, int>>& out) {  // :191-223
    std::vector<std::pair<Cell, Action>> mot;
    m_env.motions(s.cell, mot);
    for (const auto& m : mot) {
      const int mTime = 1;
      int startT = m_lastG + mTime;
      int endT = safe(s.cell).at(s.interval).end;
      const auto& sis = safe(m.first);
      for (std::size_t i = 0; i < sis.size(); ++i) {
        const Interval& si = sis[i];
        if (si.start - mTime > endT || si.end < startT) continue;
        int t = std::max<int>(si.start, m_lastG + 1);  // isCommandValid, mapf_prioritized_sipp.cpp:129-142
        out.emplace_back(SState{m.first, i}, SAction{m.second, mTime}, t - m_lastG);
      }
    }
  }
  void onExpandNode(const SState&, int, int g) {
    m_lastG = g;
    ++expanded;
  }
  void onDiscover(const SState&, int, int) {}

  void setCollisionIntervals(const Cell& c, const std::vector<Interval>& ivs) {  // :245-284
    m_safe.erase(c);
    std::vector<Interval> sorted(ivs);
    std::sort(sorted.begin(), sorted.end(), [](const Interval& a, const Interval& b) { return a.start < b.start; });
    if (!ivs.empty()) {
      auto& dst = m_safe[c];
      long long start = 0;
      int lastEnd = 0;
      for (const auto& iv : sorted) {
        if (start <= static_cast<long long>(iv.start) - 1) dst.push_back(Interval{static_cast<int>(start), iv.start - 1});
        start = static_cast<long long>(iv.end) + 1;
        lastEnd = iv.end;
      }
      if (lastEnd < INT_MAX) dst.push_back(Interval{static_cast<int>(start), INT_MAX});
    }
  }
  bool findSafeInterval(const Cell& c, int time, std::size_t& idx) {  // :286-296
    const auto& si = safe(c);
    for (std::size_t i = 0; i < si.size(); ++i)
      if (si[i].start <= time && si[i].end >= time) {
        idx = i;
        return true;
      }
    return false;
  }
  int64_t expanded = 0;

 private:
  const std::vector<Interval>& safe(const Cell& c) {  // :299-307
    static const std::vector<Interval> whole(1, Interval{0, INT_MAX});
    auto it = m_safe.find(c);
    return it == m_safe.end() ? whole : it->second;
  }
  GridEnv& m_env;
  int m_lastG = 0;
  std::unordered_map<Cell, std::vector<Interval>, CellHash> m_safe;
};

struct TimedPlan {
  std::vector<std::pair<Cell, int>> states;     // (cell, t)
  std::vector<std::pair<Action, int>> actions;  // (action, duration)
  int cost = 0, fmin = 0;
};

class Sipp {  // sipp.hpp:66-134
 public:
  explicit Sipp(GridEnv& env) : m_env(env), m_astar(m_env) {}
  void setCollisionIntervals(const Cell& c, const std::vector<Interval>& ivs) { m_env.setCollisionIntervals(c, ivs); }
  bool search(const Cell& start, TimedPlan& out, int startTime = 0) {
    PlanResult<SState, SAction, int> raw;
    out = TimedPlan();
    std::size_t idx;
    if (!m_env.findSafeInterval(start, startTime, idx)) return false;
    bool ok = m_astar.search(SState{start, idx}, raw, startTime);
    out.cost = raw.cost - startTime;
    out.fmin = raw.fmin;
    for (std::size_t i = 0; i < raw.actions.size(); ++i) {
      int waitTime = raw.actions[i].second - raw.actions[i].first.time;
      if (waitTime == 0) {
        out.states.emplace_back(raw.states[i].first.cell, raw.states[i].second);
        out.actions.emplace_back(raw.actions[i].first.action, raw.actions[i].second);
      } else {  // explicit Wait before the move
        out.states.emplace_back(raw.states[i].first.cell, raw.states[i].second);
        out.actions.emplace_back(Action::Wait, waitTime);
        out.states.emplace_back(raw.states[i].first.cell, raw.states[i].second + waitTime);
        out.actions.emplace_back(raw.actions[i].first.action, raw.actions[i].first.time);
      }
    }
    out.states.emplace_back(raw.states.back().first.cell, raw.states.back().second);
    return ok;
  }
  int64_t expanded() const { return m_env.expanded; }

 private:
  SippEnv m_env;
  AStar<SState, SAction, int, SippEnv, SStateHash> m_astar;
};

// main() of mapf_prioritized_sipp.cpp:214-270.  stats = {cost, totalExpanded, elapsed_ns}
inline int prioritizedPlan(int dimx, int dimy, int nObst, const int32_t* obstXY, int nAgents, const int32_t* startsXY,
                           const int32_t* goalsXY, int64_t* stats, int32_t* planned, int32_t* nStates,
                           int32_t* statesXYT, int cap) {
  std::vector<uint8_t> mask(static_cast<std::size_t>(dimx) * dimy, 0);
  for (int i = 0; i < nObst; ++i) {
    int x = obstXY[2 * i], y = obstXY[2 * i + 1];
    if (x >= 0 && x < dimx && y >= 0 && y < dimy) mask[y * dimx + x] = 1;
  }
  std::map<Cell, std::vector<Interval>> all;
  long cost = 0;
  int64_t expanded = 0;
  int nPlanned = 0;
  for (int a = 0; a < nAgents; ++a) {
    GridEnv env(dimx, dimy, mask, Cell{goalsXY[2 * a], goalsXY[2 * a + 1]});
    Sipp sipp(env);
    for (const auto& kv : all) sipp.setCollisionIntervals(kv.first, kv.second);
    TimedPlan sol;
    bool ok = sipp.search(Cell{startsXY[2 * a], startsXY[2 * a + 1]}, sol);
    expanded += sipp.expanded();
    planned[a] = ok ? 1 : 0;
    nStates[a] = 0;
    if (!ok) continue;
    ++nPlanned;
    auto last = sol.states[0];
    for (std::size_t k = 1; k < sol.states.size(); ++k)
      if (sol.states[k].first != last.first) {
        all[last.first].push_back(Interval{last.second, sol.states[k].second - 1});
        last = sol.states[k];
      }
    all[sol.states.back().first].push_back(Interval{sol.states.back().second, INT_MAX});
    cost += sol.cost;
    nStates[a] = static_cast<int32_t>(sol.states.size());
    for (int k = 0; k < nStates[a] && k < cap; ++k) {
      int64_t o = (static_cast<int64_t>(a) * cap + k) * 3;
      statesXYT[o] = sol.states[k].first.x;
      statesXYT[o + 1] = sol.states[k].first.y;
      statesXYT[o + 2] = sol.states[k].second;
    }
  }
  stats[0] = cost;
  stats[1] = expanded;
  return nPlanned;
}

}  // namespace sipp
}  // namespace oracle
