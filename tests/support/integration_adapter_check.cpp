// TEST INFRASTRUCTURE (tests/test_integration_adapter.py, build container only): include/gpu_low_level.hpp — the binding
// INTEGRATION.md §1 shows — instantiated over types with the member names of example/ecbs.cpp's State / Action /
// Location / Constraints (declared here: the example's own definitions sit in a .cpp that needs Boost), against the
// REFERENCE's planresult.hpp and neighbor.hpp, and linked with libmrp_ll.so.  Without a GPU the engine refuses to start
// (MRP_LL_E_DEVICE): that, and the build itself, is what the check is about; with one it runs a search on a 3 x 3 map.
#include <cstdio>
#include <set>
#include <tuple>

#include <libMultiRobotPlanning/neighbor.hpp>
#include <gpu_low_level.hpp>

struct State {
  State(int time, int x, int y) : time(time), x(x), y(y) {}
  int time, x, y;
};
enum class Action { Up, Down, Left, Right, Wait };
struct Location {
  Location(int x, int y) : x(x), y(y) {}
  int x, y;
};
struct VertexConstraint {
  int time, x, y;
  bool operator<(const VertexConstraint& o) const { return std::tie(time, x, y) < std::tie(o.time, o.x, o.y); }
};
struct EdgeConstraint {
  int time, x1, y1, x2, y2;
  bool operator<(const EdgeConstraint& o) const { return std::tie(time, x1, y1, x2, y2) < std::tie(o.time, o.x1, o.y1, o.x2, o.y2); }
};
struct Constraints {
  std::set<VertexConstraint> vertexConstraints;
  std::set<EdgeConstraint> edgeConstraints;
};
typedef mrp::GpuLowLevelSearch<State, Action, Location, Constraints> LowLevelSearch_t;
template class mrp::GpuLowLevelSearch<State, Action, Location, Constraints>;  // every member is compiled
typedef libMultiRobotPlanning::Neighbor<State, Action, int> Neighbor_t;       // (the reference's neighbor.hpp parses too)

int main() {
  mrp_ll_ctx* ctx = nullptr;
  const int rc = mrp_ll_create(nullptr, &ctx);
  if (rc == MRP_LL_E_DEVICE) {
    std::printf("adapter built and linked; no HIP device here (MRP_LL_E_DEVICE), as expected without a GPU\n");
    return 0;
  }
  if (rc != MRP_LL_SUCCESS) return 2;
  int mapId = -1;
  const int32_t obst[2] = {1, 1};
  if (mrp_ll_upload_map(ctx, 3, 3, 1, obst, &mapId) != MRP_LL_SUCCESS) return 3;
  Constraints none;
  std::vector<LowLevelSearch_t::Plan> solution(1);
  LowLevelSearch_t ecbs(ctx, mapId, 0, Location(2, 2), none, solution, 1.3f);
  LowLevelSearch_t::Plan out;
  const bool ok = ecbs.search(State(0, 0, 0), out);
  LowLevelSearch_t cbs(ctx, mapId, 0, Location(2, 2), none);
  LowLevelSearch_t::Plan out2;
  const bool ok2 = cbs.search(State(0, 0, 0), out2);
  std::printf("ecbs ok=%d cost=%d states=%zu; cbs ok=%d cost=%d; expanded %lld\n", (int)ok, out.cost, out.states.size(), (int)ok2,
              out2.cost, (long long)LowLevelSearch_t::lowLevelExpanded());
  mrp_ll_destroy(ctx);
  return ok && ok2 && out.cost == 4 && out2.cost == 4 ? 0 : 4;
}
