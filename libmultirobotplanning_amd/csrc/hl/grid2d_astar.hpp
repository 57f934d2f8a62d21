// Config 1 of BASELINE.json — `./a_star` on a text map (example/a_star.cpp) — is CPU plumbing by definition: a single
// 2-D A* (no time dimension, no constraints) that the reference itself runs in microseconds.  This is the host-side
// restatement used by the `a_star` front-end (cli.py); the GPU engine is not involved.
//   AStar::search            include/libMultiRobotPlanning/a_star.hpp:63-161 (open list order :168-179: lowest f, then
//                            highest g; rediscovery with a smaller g = increase(handle), :139-145)
//   Environment              example/a_star.cpp:72-125 (neighbours in the order Up, Down, Left, Right; unit costs;
//                            Manhattan heuristic; isSolution = goal cell)
// The open list replays boost::heap::d_ary_heap<arity<2>, mutable_<true>> through ExactHeap, so the path returned is
// the reference's path, tie-breaks included.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <vector>

#include "exact_heap.hpp"

namespace mrp_hl {

// mask[y * dimx + x] != 0 = obstacle.  Returns the number of states written to outXY ([cap][2]), 0 = no solution;
// *cost = PlanResult::cost, *expanded = nodes popped (onExpandNode calls).
inline int32_t astarGrid2d(int32_t dimx, int32_t dimy, const uint8_t* mask, int32_t sx, int32_t sy, int32_t gx, int32_t gy,
                           int32_t* outXY, int32_t cap, int32_t* cost, int64_t* expanded) {
  auto valid = [&](int32_t x, int32_t y) { return x >= 0 && x < dimx && y >= 0 && y < dimy && !mask[y * dimx + x]; };
  *cost = 0;
  *expanded = 0;
  if (!valid(sx, sy)) return 0;  // a_star.cpp:190: search only if the start state is valid
  const int32_t cells = dimx * dimy;
  std::vector<int32_t> g(cells, -1), f(cells, 0), parent(cells, -1);
  std::vector<uint8_t> closed(cells, 0), inOpen(cells, 0);
  struct Less {  // Node::operator< (a_star.hpp:168-179)
    const std::vector<int32_t>*f, *g;
    bool operator()(int32_t a, int32_t b) const {
      if ((*f)[a] != (*f)[b]) return (*f)[a] > (*f)[b];
      return (*g)[a] < (*g)[b];
    }
  };
  ExactHeap<Less> open(Less{&f, &g});
  auto h = [&](int32_t x, int32_t y) { return std::abs(x - gx) + std::abs(y - gy); };
  const int32_t s = sy * dimx + sx;
  g[s] = 0;
  f[s] = h(sx, sy);
  open.push(s);
  inOpen[s] = 1;
  static const int32_t dx[4] = {0, 0, -1, 1}, dy[4] = {1, -1, 0, 0};  // Up, Down, Left, Right (a_star.cpp:89-110)
  while (!open.empty()) {
    const int32_t cur = open.top();
    *expanded += 1;
    const int32_t cx = cur % dimx, cy = cur / dimx;
    if (cx == gx && cy == gy) {
      std::vector<int32_t> rev;
      for (int32_t c = cur; c != -1; c = parent[c]) rev.push_back(c);
      int32_t n = static_cast<int32_t>(rev.size());
      for (int32_t k = 0; k < n && k < cap; ++k) {
        outXY[2 * k] = rev[n - 1 - k] % dimx;
        outXY[2 * k + 1] = rev[n - 1 - k] / dimx;
      }
      *cost = g[cur];
      return n;
    }
    open.pop();
    inOpen[cur] = 0;
    closed[cur] = 1;
    for (int k = 0; k < 4; ++k) {
      const int32_t nx = cx + dx[k], ny = cy + dy[k];
      if (!valid(nx, ny)) continue;
      const int32_t nb = ny * dimx + nx;
      if (closed[nb]) continue;
      const int32_t tg = g[cur] + 1;
      if (!inOpen[nb]) {
        g[nb] = tg;
        f[nb] = tg + h(nx, ny);
        open.push(nb);
        inOpen[nb] = 1;
      } else {
        if (tg >= g[nb]) continue;
        f[nb] -= g[nb] - tg;
        g[nb] = tg;
        open.increase(nb);
      }
      parent[nb] = cur;
    }
  }
  return 0;
}

}  // namespace mrp_hl
