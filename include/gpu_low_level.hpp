// gpu_low_level.hpp — the binding a maintainer of libMultiRobotPlanning adds to run the low-level searches of CBS / ECBS
// on the MI355X engine.  The reference has no FFI at this point: its low-level search is a private
// typedef (cbs.hpp:248, ecbs.hpp:421-422) called as  LowLevelSearch_t(llenv[, w]).search(start, out)
// (cbs.hpp:99-101,155-157; ecbs.hpp:126-129,265-268).  This adapter has the same constructor / search shape and forwards to
// the C-ABI of mrp_ll.h; one search per call (the batched drivers of mrp_hl.h are what bench.py measures).
//
// Template parameters are the example's own types (example/ecbs.cpp / example/cbs.cpp):
//   State       { int time, x, y; }  constructible as State(time, x, y)            (ecbs.cpp:29-47)
//   Action      enum class { Up, Down, Left, Right, Wait }                          (ecbs.cpp:49-55) == MRP_LL_ACT_*
//   Location    { int x, y; }                                                       (ecbs.cpp:216-230)
//   Constraints { set<VertexConstraint{time,x,y}> vertexConstraints; set<EdgeConstraint{time,x1,y1,x2,y2}> edgeConstraints; }
//                                                                                   (ecbs.cpp:108-214)
// tests/test_integration_adapter.py compiles this header against the reference's own planresult.hpp / neighbor.hpp and
// links libmrp_ll.so (in the build container, where /root/reference exists).
#pragma once
#include <mrp_ll.h>

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include <libMultiRobotPlanning/planresult.hpp>

namespace mrp {

template <typename State, typename Action, typename Location, typename Constraints>
class GpuLowLevelSearch {
 public:
  typedef libMultiRobotPlanning::PlanResult<State, Action, int> Plan;

  // ECBS: replaces AStarEpsilon<State, Action, int, LowLevelEnvironment> (ecbs.hpp:421).  `mapId` comes once from
  // mrp_ll_upload_map(ctx, dimx, dimy, nObstacles, obstaclesXY, &mapId); agentIdx / goal / constraints are what
  // Environment::setLowLevelContext stores (ecbs.cpp:264-274); `solution` is the CT node's solution vector as the focal
  // heuristics see it (ecbs.hpp:381-389).
  GpuLowLevelSearch(mrp_ll_ctx* ctx, int mapId, size_t agentIdx, const Location& goal, const Constraints& c,
                    const std::vector<Plan>& solution, float w)
      : ctx_(ctx), mapId_(mapId), agent_(agentIdx), goal_(goal), c_(c), solution_(&solution), w_(w), algo_(MRP_LL_ASTAR_EPS) {}
  // CBS: replaces AStar<State, Action, int, LowLevelEnvironment> (cbs.hpp:248): no focal context, no w.
  GpuLowLevelSearch(mrp_ll_ctx* ctx, int mapId, size_t agentIdx, const Location& goal, const Constraints& c)
      : ctx_(ctx), mapId_(mapId), agent_(agentIdx), goal_(goal), c_(c), solution_(nullptr), w_(1.0f), algo_(MRP_LL_ASTAR) {}

  bool search(const State& start, Plan& out) {
    std::vector<int32_t> vc, ec, len;
    std::vector<std::vector<int32_t>> xy(solution_ ? solution_->size() : 0);
    std::vector<const int32_t*> ptr;
    for (const auto& v : c_.vertexConstraints) vc.insert(vc.end(), {v.time, v.x, v.y});
    for (const auto& e : c_.edgeConstraints) ec.insert(ec.end(), {e.time, e.x1, e.y1, e.x2, e.y2});
    if (solution_)
      for (size_t i = 0; i < solution_->size(); ++i) {
        for (const auto& s : (*solution_)[i].states) xy[i].insert(xy[i].end(), {s.first.x, s.first.y});
        len.push_back(static_cast<int32_t>((*solution_)[i].states.size()));
        ptr.push_back(xy[i].data());
      }
    mrp_ll_job job{};
    job.map_id = mapId_;
    job.algo = algo_;
    job.w = w_;
    job.agent_idx = static_cast<int32_t>(agent_);
    job.start_x = start.x;
    job.start_y = start.y;
    job.goal_x = goal_.x;
    job.goal_y = goal_.y;
    job.n_vertex_constraints = static_cast<int32_t>(vc.size() / 3);
    job.vertex_constraints = vc.data();
    job.n_edge_constraints = static_cast<int32_t>(ec.size() / 5);
    job.edge_constraints = ec.data();
    job.n_agents = static_cast<int32_t>(len.size());
    job.path_len = len.data();
    job.path_xy = ptr.data();
    job.max_expansions = -1;
    std::vector<int32_t> states(3 * 1024), actions(1024);
    mrp_ll_result r{};
    r.states_txy = states.data();
    r.actions = actions.data();
    r.states_cap = 1024;
    if (mrp_ll_search_batch(ctx_, 1, &job, &r) != MRP_LL_SUCCESS) throw std::runtime_error(mrp_ll_last_error(ctx_));
    // reference semantics of a failed search (a_star_epsilon.hpp:88-91,284 / a_star.hpp:65-68,160): states = {start}, cost = 0
    out.states.clear();
    out.actions.clear();
    out.cost = 0;
    if (r.status == MRP_LL_NO_SOLUTION) {
      out.states.emplace_back(start, 0);
      return false;
    }
    if (r.status != MRP_LL_OK) throw std::runtime_error("mrp_ll: capacity status " + std::to_string(r.status));
    for (int k = 0; k < r.n_states; ++k) {
      out.states.emplace_back(State(states[3 * k], states[3 * k + 1], states[3 * k + 2]), k);
      if (k + 1 < r.n_states) out.actions.emplace_back(static_cast<Action>(actions[k]), 1);  // enum order ecbs.cpp:49-55
    }
    out.cost = r.cost;
    out.fmin = r.fmin;
    lowLevelExpanded() += r.expanded;  // what Environment::onExpandLowLevelNode counts (ecbs.cpp:476-479)
    return true;
  }
  static int64_t& lowLevelExpanded() {
    static int64_t n = 0;
    return n;
  }

 private:
  mrp_ll_ctx* ctx_;
  int mapId_;
  size_t agent_;
  Location goal_;
  const Constraints& c_;
  const std::vector<Plan>* solution_;
  float w_;
  int32_t algo_;
};

}  // namespace mrp
