// Can the CPU store directly into device memory (large BAR) on this box?  hipMalloc / fine-grained / uncached allocations:
// the host writes a pattern through the pointer, the device reads it back with a kernel, the host times the stores.
#include <hip/hip_runtime.h>
#include <chrono>
#include <csetjmp>
#include <csignal>
#include <cstdio>
#include <cstring>
#include <cstdint>

static sigjmp_buf jb;
static void onSegv(int) { siglongjmp(jb, 1); }

__global__ void sum(const uint32_t* p, uint32_t n, uint32_t* out) {
  uint32_t s = 0;
  for (uint32_t i = threadIdx.x; i < n; i += 64) s += __hip_atomic_load(p + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  atomicAdd(out, s);
}

static void tryAlloc(const char* name, hipError_t (*alloc)(void**, size_t)) {
  void* p = nullptr;
  const size_t bytes = 1 << 20;
  if (alloc(&p, bytes) != hipSuccess) { std::printf("%-28s alloc failed\n", name); return; }
  uint32_t* out = nullptr;
  hipMalloc((void**)&out, 4);
  hipMemset(out, 0, 4);
  hipMemset(p, 0, bytes);
  hipDeviceSynchronize();
  signal(SIGSEGV, onSegv);
  signal(SIGBUS, onSegv);
  if (sigsetjmp(jb, 1)) { std::printf("%-28s host store FAULTS\n", name); return; }
  volatile uint32_t* w = (volatile uint32_t*)p;
  auto t0 = std::chrono::steady_clock::now();
  for (uint32_t i = 0; i < 65536; ++i) w[i] = i + 1;
  __sync_synchronize();
  auto t1 = std::chrono::steady_clock::now();
  hipLaunchKernelGGL(sum, dim3(1), dim3(64), 0, 0, (const uint32_t*)p, 65536u, out);
  uint32_t got = 0;
  hipMemcpy(&got, out, 4, hipMemcpyDeviceToHost);
  uint64_t want = 0;
  for (uint32_t i = 0; i < 65536; ++i) want += i + 1;
  // host read back (slow over BAR, just to know)
  auto t2 = std::chrono::steady_clock::now();
  uint32_t r = 0;
  for (uint32_t i = 0; i < 1024; ++i) r += w[i];
  auto t3 = std::chrono::steady_clock::now();
  std::printf("%-28s host stores ok: 256 KB in %.1f us (%.2f GB/s); device saw %s; 4 KB host read-back %.1f us (sum %u)\n", name,
              std::chrono::duration<double, std::micro>(t1 - t0).count(),
              262144.0 / std::chrono::duration<double>(t1 - t0).count() / 1e9, got == (uint32_t)want ? "the pattern" : "SOMETHING ELSE",
              std::chrono::duration<double, std::micro>(t3 - t2).count(), r);
}

int main() {
  tryAlloc("hipMalloc", [](void** p, size_t n) { return hipMalloc(p, n); });
  tryAlloc("hipExtMalloc finegrained", [](void** p, size_t n) { return hipExtMallocWithFlags(p, n, hipDeviceMallocFinegrained); });
  tryAlloc("hipExtMalloc uncached", [](void** p, size_t n) { return hipExtMallocWithFlags(p, n, hipDeviceMallocUncached); });
  tryAlloc("hipMallocManaged", [](void** p, size_t n) { return hipMallocManaged(p, n, hipMemAttachGlobal); });
  return 0;
}
