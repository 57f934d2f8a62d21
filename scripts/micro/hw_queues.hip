// How many HIP streams can hold a resident (never-ending) kernel at the same time?  Streams that share a hardware queue
// serialise their kernels, which is fatal for resident kernels.  Usage: GPU_MAX_HW_QUEUES=64 ./hw_queues 48
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void spin(volatile unsigned* flag, unsigned* started) {
  if (threadIdx.x == 0) atomicAdd_system(started, 1u);
  while (*flag == 0) __builtin_amdgcn_s_sleep(100);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 32;
  unsigned *flag, *started;
  hipHostMalloc((void**)&flag, 64, hipHostMallocMapped | hipHostMallocCoherent);
  hipHostMalloc((void**)&started, 64, hipHostMallocMapped | hipHostMallocCoherent);
  *flag = 0;
  *started = 0;
  std::vector<hipStream_t> st(n);
  for (auto& s : st) hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  for (auto& s : st) hipLaunchKernelGGL(spin, dim3(4), dim3(64), 0, s, flag, started);
  auto t0 = std::chrono::steady_clock::now();
  unsigned seen = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 3.0) {
    seen = *(volatile unsigned*)started;
    if (seen == (unsigned)n * 4u) break;
    std::this_thread::sleep_for(std::chrono::milliseconds(10));
  }
  printf("streams %d: workgroups started %u of %d after %.2f s (GPU_MAX_HW_QUEUES=%s)\n", n, seen, n * 4,
         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "default");
  *flag = 1;
  hipDeviceSynchronize();
  return 0;
}
