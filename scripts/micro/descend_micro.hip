// Micro-benchmark (dev tool): cycles of one 6-level sift-down block as the search kernel does it, on an LDS heap.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));
#define DEVI __device__ __forceinline__
typedef __attribute__((address_space(3))) uint64_t* P64;
typedef __attribute__((address_space(3))) u64x2* PPair;
typedef __attribute__((address_space(3))) uint32_t* P32;
DEVI uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }
DEVI uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }

template <int VARIANT>
DEVI uint32_t block(P64 heap, P32 pos, uint32_t n, uint32_t idx, uint32_t xk) {
  const uint32_t lane = threadIdx.x;
  const uint32_t lv = 31u - (uint32_t)__builtin_clz(lane + 1);
  const uint32_t off = (lane + 1) - (1u << lv);
  const uint32_t node = ((idx + 1) << lv) - 1 + off;
  const uint32_t c = 2 * node + 1;
  const bool has = (lane < 63) && (c < n);
  u64x2 pr; pr.x = 0; pr.y = 0;
  if (has) pr = *(PPair)(heap + c);
  const uint32_t kl = (uint32_t)(pr.x >> 32), kr = (uint32_t)(pr.y >> 32);
  const bool hasR = has && (c + 1 < n);
  const bool right = hasR && (kl < kr);
  const uint64_t pe = right ? pr.y : pr.x;
  const uint32_t pk = right ? kr : kl;
  const bool go = has && !(pk < xk);
  const uint64_t goMask = ballot64(go);
  const uint64_t rightMask = ballot64(right);
  uint64_t pathMask = 0;
  uint32_t rel = 0, steps = 0;
  if (VARIANT == 0) {
#pragma unroll 1
    while (steps < 6 && ((goMask >> rel) & 1ull)) {
      pathMask |= 1ull << rel;
      rel = 2 * rel + 1 + (uint32_t)((rightMask >> rel) & 1ull);
      steps += 1;
    }
  } else {  // branch-free, always six steps
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      const uint32_t g = (uint32_t)((goMask >> rel) & 1ull);
      const uint32_t r = (uint32_t)((rightMask >> rel) & 1ull);
      pathMask |= (uint64_t)g << rel;
      rel = g ? 2 * rel + 1 + r : rel;
      steps += g;
    }
  }
  if ((pathMask >> lane) & 1ull) {
    heap[node] = pe;
    if (VARIANT != 2) pos[(uint32_t)pe * 4 + 3] = node;
  }
  return ((idx + 1) << steps) - 1 + (rel + 1 - (1u << steps));
}

template <int VARIANT>
__global__ void __launch_bounds__(64) k(uint64_t* out, const uint64_t* init, uint32_t n, uint32_t iters) {
  __shared__ __attribute__((aligned(16))) uint64_t heapS[2050];
  __shared__ uint32_t posS[4096];
  P64 heap = (P64)(heapS + 1);
  P32 pos = (P32)posS;
  for (uint32_t i = threadIdx.x; i < n; i += 64) heap[i] = init[i];
  __syncthreads();
  uint32_t acc = 0;
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (uint32_t it = 0; it < iters; ++it) {
    uint32_t idx = block<VARIANT>(heap, pos, n, 0, 0);   // xk = 0: always descends the full six levels
    acc += rfl(idx);
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}

int main() {
  const uint32_t n = 2000, iters = 2000;
  std::vector<uint64_t> h(n);
  for (uint32_t i = 0; i < n; ++i) h[i] = ((uint64_t)(1000000u - i * 7u % 1000u) << 32) | (i % 1000);
  uint64_t *dInit, *dOut;
  hipMalloc(&dInit, n * 8); hipMalloc(&dOut, 1024 * 16);
  hipMemcpy(dInit, h.data(), n * 8, hipMemcpyHostToDevice);
  for (int grid : {1, 1024, 2048}) {
    for (int v = 0; v < 3; ++v) {
      if (v == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(64), 0, 0, dOut, dInit, n, iters);
      if (v == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(64), 0, 0, dOut, dInit, n, iters);
      if (v == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(64), 0, 0, dOut, dInit, n, iters);
      hipDeviceSynchronize();
      uint64_t o[2];
      hipMemcpy(o, dOut, 16, hipMemcpyDeviceToHost);
      printf("grid %d variant %d: %.1f cycles per 6-level block (acc %llu)\n", grid, v, (double)o[0] / iters, (unsigned long long)o[1]);
    }
  }
  return 0;
}
