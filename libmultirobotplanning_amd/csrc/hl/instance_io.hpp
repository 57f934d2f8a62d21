// Synthetic instance generator — the bench workload.  "32x32_obst204-shaped": obstacles sampled
// uniformly without replacement, agents with pairwise distinct starts and pairwise distinct goals, each goal inside
// the 4-connected free component of its start (as in all 3000 shipped benchmark files; an unreachable goal would
// never terminate in the reference).  PRNG = splitmix64 so every box generates identical sets.
#pragma once
#include <cstdint>
#include <vector>

namespace mrp_hl {

struct SplitMix64 {
  uint64_t s;
  explicit SplitMix64(uint64_t seed) : s(seed) {}
  uint64_t next() {
    uint64_t z = (s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return static_cast<uint32_t>(next() % n); }  // modulo bias is irrelevant here
};

inline int generateInstance(uint64_t seed, int dimx, int dimy, int nObst, int nAgents, int32_t* obstXY,
                            int32_t* startsXY, int32_t* goalsXY) {
  const int cells = dimx * dimy;
  if (dimx <= 0 || dimy <= 0 || nObst < 0 || nAgents < 0 || nObst + nAgents > cells) return -1;
  SplitMix64 rng(seed);
  for (int attempt = 0; attempt < 64; ++attempt) {
    std::vector<int> perm(cells);
    for (int i = 0; i < cells; ++i) perm[i] = i;
    for (int i = 0; i < nObst; ++i) {  // partial Fisher-Yates
      int j = i + static_cast<int>(rng.below(static_cast<uint32_t>(cells - i)));
      std::swap(perm[i], perm[j]);
    }
    std::vector<uint8_t> blocked(cells, 0);
    for (int i = 0; i < nObst; ++i) blocked[perm[i]] = 1;
    // component labels
    std::vector<int> comp(cells, -1);
    std::vector<int> compSize;
    std::vector<int> stack;
    for (int c = 0; c < cells; ++c) {
      if (blocked[c] || comp[c] >= 0) continue;
      int id = static_cast<int>(compSize.size());
      compSize.push_back(0);
      stack.push_back(c);
      comp[c] = id;
      while (!stack.empty()) {
        int u = stack.back();
        stack.pop_back();
        compSize[id]++;
        int ux = u % dimx, uy = u / dimx;
        const int nx[4] = {ux - 1, ux + 1, ux, ux}, ny[4] = {uy, uy, uy - 1, uy + 1};
        for (int k = 0; k < 4; ++k) {
          if (nx[k] < 0 || nx[k] >= dimx || ny[k] < 0 || ny[k] >= dimy) continue;
          int v = ny[k] * dimx + nx[k];
          if (!blocked[v] && comp[v] < 0) {
            comp[v] = id;
            stack.push_back(v);
          }
        }
      }
    }
    // starts: distinct free cells; goals: distinct free cells, same component as the start
    std::vector<uint8_t> startUsed(cells, 0), goalUsed(cells, 0);
    bool ok = true;
    for (int a = 0; a < nAgents && ok; ++a) {
      int s = -1, g = -1;
      for (int tries = 0; tries < 10000; ++tries) {
        int c = static_cast<int>(rng.below(static_cast<uint32_t>(cells)));
        if (!blocked[c] && !startUsed[c]) {
          s = c;
          break;
        }
      }
      if (s < 0) {
        ok = false;
        break;
      }
      for (int tries = 0; tries < 10000; ++tries) {
        int c = static_cast<int>(rng.below(static_cast<uint32_t>(cells)));
        if (!blocked[c] && !goalUsed[c] && comp[c] == comp[s]) {
          g = c;
          break;
        }
      }
      if (g < 0) {
        ok = false;
        break;
      }
      startUsed[s] = 1;
      goalUsed[g] = 1;
      startsXY[2 * a] = s % dimx;
      startsXY[2 * a + 1] = s / dimx;
      goalsXY[2 * a] = g % dimx;
      goalsXY[2 * a + 1] = g / dimx;
    }
    if (!ok) continue;
    for (int i = 0; i < nObst; ++i) {
      obstXY[2 * i] = perm[i] % dimx;
      obstXY[2 * i + 1] = perm[i] / dimx;
    }
    return 0;
  }
  return -1;
}

}  // namespace mrp_hl
