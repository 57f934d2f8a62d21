// Does a wave64 VALU instruction get cheaper when only the first 16 / 32 lanes are enabled (EXEC)?  gfx950 microbenchmark:
// a dependent chain of integer VALU ops under different lane masks, timed with s_memtime.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int LANES>
__global__ void chain(uint32_t* out, uint64_t* cycles, int iters) {
  uint32_t x = threadIdx.x * 2654435761u + 1u, y = threadIdx.x ^ 0x9E3779B9u;
  uint64_t t0 = 0, t1 = 0;
  if (LANES == 64 || threadIdx.x < LANES) {
    t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        x = x * 1664525u + y;   // v_mad / v_mul_lo + v_add: dependent VALU chain
        y = (y ^ x) + 0x85EBCA6Bu;
      }
    }
    t1 = __builtin_amdgcn_s_memtime();
  }
  out[blockIdx.x * 64 + threadIdx.x] = x ^ y;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int LANES>
static void run(const char* name, int wavesPerSimdHint) {
  uint32_t* out; uint64_t* cyc;
  const int blocks = 256 * wavesPerSimdHint * 4;  // fill every SIMD with that many waves
  hipMalloc(&out, sizeof(uint32_t) * 64 * blocks);
  hipMalloc(&cyc, sizeof(uint64_t) * blocks);
  const int iters = 20000;
  hipLaunchKernelGGL(chain<LANES>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL(chain<LANES>, dim3(blocks), dim3(64), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  uint64_t c0; hipMemcpy(&c0, cyc, 8, hipMemcpyDeviceToHost);
  const double ops = double(iters) * 16 * 4;  // ~4 VALU instructions per inner step
  printf("%-28s waves/SIMD %d: %.3f ms wall, %.2f ns per VALU instruction per wave, s_memtime ticks/instr %.3f\n", name,
         wavesPerSimdHint, ms, ms * 1e6 / ops, double(c0) / ops);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 4}) {
    run<64>("all 64 lanes", w);
    run<32>("lanes 0..31 (EXEC half)", w);
    run<16>("lanes 0..15 (EXEC quarter)", w);
  }
  return 0;
}
