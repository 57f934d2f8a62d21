// High-level conflict scans of the grid MAPF Environment as a HIP kernel for gfx950:
//   Environment::getFirstConflict   example/ecbs.cpp:401-452 (example/cbs.cpp identical)
//   Environment::focalHeuristic     example/ecbs.cpp:315-350
// Both walk t = 0 .. max_t-1 (max_t = longest path - 1: the final time step is never checked) and all agent pairs
// i < j; an agent past the end of its path stays on its last cell (getState, ecbs.cpp:486-495).
//   vertex conflict at t : state_i(t) == state_j(t)
//   edge conflict at t   : state_i(t) == state_j(t+1) && state_i(t+1) == state_j(t)
// getFirstConflict returns the first hit in the order (t ascending; at one t every vertex pair before every edge pair;
// pairs in (i, j) lexicographic order); focalHeuristic counts all hits.
//
// One 256-thread workgroup per solution (CT node).  Wave w takes the time steps t = w, w + 4, ...; lanes are agents j
// (in chunks of 64), a wave-uniform loop runs over i, and one ballot per (i, chunk) tests 64 pairs at once: the count is
// a popcount, the first pair of a time step is the lowest set bit of the first non-empty ballot.  First hits are ordered
// by the 64-bit key  t << 40 | type << 32 | i << 16 | j  and combined with an LDS atomicMin; counts with an atomicAdd.
// Integer work only; results are exact.  Positions are x | y << 8 (the engine's grids are at most 255 x 255).
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mrp {

struct ConflictOut {  // mirrors mrp_ll_conflict of include/mrp_ll.h (10 x int32)
  int32_t found, time, agent1, agent2, type, x1, y1, x2, y2, count;
};

struct ConflictParams {
  const uint32_t* setFirstAgent;   // [nSets + 1]
  const uint32_t* pathFirstState;  // [totalAgents + 1]
  const uint16_t* states;          // [totalStates]  x | y << 8
  ConflictOut* out;                // [nSets]
  uint32_t nSets;
};

__device__ __forceinline__ uint32_t posAt(const uint16_t* st, uint32_t first, uint32_t len, uint32_t t) {
  return st[first + (t < len ? t : len - 1)];
}

extern "C" __global__ void __launch_bounds__(256) mrp_ll_conflict_kernel(ConflictParams P) {
  __shared__ unsigned long long bestKey;
  __shared__ uint32_t total;
  __shared__ uint32_t maxLen;
  const uint32_t s = blockIdx.x;
  if (s >= P.nSets) return;
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t a0 = P.setFirstAgent[s], nAg = P.setFirstAgent[s + 1] - a0;
  if (tid == 0) {
    bestKey = ~0ull;
    total = 0;
    maxLen = 0;
  }
  __syncthreads();
  {  // max_t = max over agents of (path length - 1)
    uint32_t m = 0;
    for (uint32_t a = tid; a < nAg; a += 256) m = max(m, P.pathFirstState[a0 + a + 1] - P.pathFirstState[a0 + a]);
    atomicMax(&maxLen, m);
  }
  __syncthreads();
  const uint32_t T = maxLen ? maxLen - 1 : 0;
  const uint32_t nChunks = (nAg + 63u) / 64u;
  uint32_t count = 0;
  unsigned long long myBest = ~0ull;
  for (uint32_t t = wave; t < T; t += 4) {
    unsigned long long firstV = ~0ull, firstE = ~0ull;  // (i << 16 | j) of the first vertex / edge pair at this t
    for (uint32_t cj = 0; cj < nChunks; ++cj) {
      const uint32_t j = cj * 64u + lane;
      uint32_t curJ = 0xFFFFFFFFu, nxtJ = 0xFFFFFFFEu;  // lanes without an agent never match
      if (j < nAg) {
        const uint32_t f = P.pathFirstState[a0 + j], len = P.pathFirstState[a0 + j + 1] - f;
        curJ = posAt(P.states, f, len, t);
        nxtJ = posAt(P.states, f, len, t + 1);
      }
      // i runs over every agent below the end of this chunk (pairs need i < j)
      const uint32_t iEnd = min(nAg, cj * 64u + 64u);
      for (uint32_t i = 0; i < iEnd; ++i) {
        const uint32_t fi = P.pathFirstState[a0 + i], leni = P.pathFirstState[a0 + i + 1] - fi;  // wave-uniform loads
        const uint32_t curI = posAt(P.states, fi, leni, t), nxtI = posAt(P.states, fi, leni, t + 1);
        const unsigned long long vm = __ballot(j > i && curJ == curI);
        const unsigned long long em = __ballot(j > i && curJ == nxtI && nxtJ == curI);
        count += (uint32_t)__popcll(vm) + (uint32_t)__popcll(em);
        if (vm) {
          const unsigned long long k = ((unsigned long long)i << 16) | (cj * 64u + (uint32_t)__builtin_ctzll(vm));
          firstV = k < firstV ? k : firstV;
        }
        if (em) {
          const unsigned long long k = ((unsigned long long)i << 16) | (cj * 64u + (uint32_t)__builtin_ctzll(em));
          firstE = k < firstE ? k : firstE;
        }
      }
    }
    unsigned long long key = ~0ull;
    if (firstV != ~0ull)
      key = ((unsigned long long)t << 40) | firstV;
    else if (firstE != ~0ull)
      key = ((unsigned long long)t << 40) | (1ull << 32) | firstE;
    myBest = key < myBest ? key : myBest;
  }
  if (lane == 0) {  // count and the keys are wave-uniform
    atomicAdd(&total, count);
    atomicMin(&bestKey, myBest);
  }
  __syncthreads();
  if (tid == 0) {
    ConflictOut o;
    o.count = (int32_t)total;
    o.found = bestKey != ~0ull ? 1 : 0;
    o.time = o.agent1 = o.agent2 = o.type = o.x1 = o.y1 = o.x2 = o.y2 = 0;
    if (o.found) {
      const uint32_t t = (uint32_t)(bestKey >> 40), type = (uint32_t)(bestKey >> 32) & 1u;
      const uint32_t i = (uint32_t)(bestKey >> 16) & 0xFFFFu, j = (uint32_t)bestKey & 0xFFFFu;
      const uint32_t fi = P.pathFirstState[a0 + i], leni = P.pathFirstState[a0 + i + 1] - fi;
      const uint32_t c1 = posAt(P.states, fi, leni, t), c2 = posAt(P.states, fi, leni, t + 1);
      o.time = (int32_t)t;
      o.agent1 = (int32_t)i;
      o.agent2 = (int32_t)j;
      o.type = (int32_t)type;
      o.x1 = (int32_t)(c1 & 0xFF);
      o.y1 = (int32_t)(c1 >> 8);
      if (type) {  // Conflict::Edge carries agent1's move (ecbs.cpp:439-445)
        o.x2 = (int32_t)(c2 & 0xFF);
        o.y2 = (int32_t)(c2 >> 8);
      }
    }
    P.out[s] = o;
  }
}

}  // namespace mrp

extern "C" hipError_t mrp_ll_launch_conflict(const mrp::ConflictParams* P, hipStream_t stream) {
  if (P->nSets == 0) return hipSuccess;
  hipLaunchKernelGGL(mrp::mrp_ll_conflict_kernel, dim3(P->nSets), dim3(256), 0, stream, *P);
  return hipGetLastError();
}
