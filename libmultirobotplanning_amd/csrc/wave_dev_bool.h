// The "wave program" vocabulary of the compact search tier (ll_compact.h), as gfx950 code.
//
// ll_compact.h is written against a small set of names — a per-lane 32-bit value `V`, a per-lane predicate `B`, wave-wide
// ballots and lane reads, byte-addressed LDS accesses — so that the SAME source is (a) compiled here into plain per-lane
// HIP (V = uint32_t, B = bool: every name below is one or two instructions) and (b) compiled by tests/support/wave_emu.h
// into a 64-lane lockstep interpretation on the host, where the CPU test-suite replays thousands of harvested searches
// against the oracle before the code ever reaches a GPU.  Control flow in ll_compact.h is wave-uniform by construction
// (it branches only on scalars obtained from ballots / lane reads), which is what makes the lockstep reading exact.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wv {

#define WV_FN __device__ __forceinline__
#define WV_ENTRY __device__ __attribute__((noinline))  // a real function: its own register allocation

typedef uint32_t V;   // one 32-bit value per lane
typedef bool B;       // one predicate per lane
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
struct V2 { V x, y; };
struct V4 { V x, y, z, w; };

typedef __attribute__((address_space(3))) uint8_t* Lds;  // the workgroup's LDS window, byte addressed

// The window starts at LDS address 0: the kernels that host the tier declare no static LDS (ll_kernel.hip checks it), so
// every address inside the window is a compile-time constant of the ds_ instructions.
WV_FN Lds windowBase(Lds) { return (Lds)(uintptr_t)0; }
WV_FN V laneId() { return threadIdx.x; }
WV_FN uint64_t clock64() { return __builtin_amdgcn_s_memtime(); }  // shader cycles (diagnostic builds)
WV_FN V splat(uint32_t s) { return s; }
WV_FN V sel(B c, V a, V b) { return c ? a : b; }
WV_FN B bsplat(bool s) { return s; }
WV_FN uint64_t ballot(B p) { return __builtin_amdgcn_ballot_w64(p); }
WV_FN uint32_t readlane(V v, uint32_t lane) { return __builtin_amdgcn_readlane(v, lane); }
WV_FN uint32_t first(V v) { return __builtin_amdgcn_readfirstlane(v); }
// v with lane `lane` replaced by val (both wave-uniform): v_writelane_b32 (this clang has no builtin for the intrinsic)
extern "C" __device__ int mrp_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
WV_FN V writelane(V v, uint32_t val, uint32_t lane) { return (uint32_t)mrp_llvm_writelane((int)val, (int)lane, (int)v); }
WV_FN V shr1(V v) { return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false); }
// lane i of every 16-lane row receives lane i + 1's value; the last lane of a row receives `fill`
WV_FN V rowShl1(V v, uint32_t fill) { return (uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v, 0x101, 0xF, 0xF, false); }
WV_FN V clz(V v) { return (uint32_t)__builtin_clz(v); }          // v != 0
WV_FN V popc(V v) { return (uint32_t)__builtin_popcount(v); }
WV_FN float uintAsFloat(uint32_t v) { return __uint_as_float(v); }
WV_FN V sad(V a, V b, V c) { return ((a > b ? a : b) - (a > b ? b : a)) + c; }  // |a - b| + c  (v_sad_u32)
WV_FN float fmulRn(float a, float b) { return __fmul_rn(a, b); }  // binary32 product, no contraction
WV_FN V cvtF32(V v) { return __float_as_uint((float)v); }        // (float)v as bits
WV_FN B leF32(V aBits, float b) { return __uint_as_float(aBits) <= b; }

// ---- LDS (byte addresses inside the window) ----
WV_FN V ldsLoad32(Lds l, V addr) { return *(__attribute__((address_space(3))) uint32_t*)(l + addr); }
WV_FN V ldsLoad32m(Lds l, V addr, B m) { return m ? *(__attribute__((address_space(3))) uint32_t*)(l + addr) : 0u; }
WV_FN V ldsLoadU16(Lds l, V addr) { return *(__attribute__((address_space(3))) uint16_t*)(l + addr); }
WV_FN V ldsLoadU8(Lds l, V addr) { return *(__attribute__((address_space(3))) uint8_t*)(l + addr); }
WV_FN V2 ldsLoad64m(Lds l, V addr, B m) {
  V2 r{0u, 0u};
  if (m) {
    const v2u t = *(__attribute__((address_space(3))) v2u*)(l + addr);
    r.x = t.x;
    r.y = t.y;
  }
  return r;
}
WV_FN V2 ldsLoad64(Lds l, V addr) {
  const v2u t = *(__attribute__((address_space(3))) v2u*)(l + addr);
  return V2{t.x, t.y};
}
WV_FN V4 ldsLoad128(Lds l, V addr) {
  const v4u t = *(__attribute__((address_space(3))) v4u*)(l + addr);
  return V4{t.x, t.y, t.z, t.w};
}
WV_FN void ldsStore32m(Lds l, V addr, V val, B m) {
  if (m) *(__attribute__((address_space(3))) uint32_t*)(l + addr) = val;
}
WV_FN void ldsStore32(Lds l, V addr, V val) { *(__attribute__((address_space(3))) uint32_t*)(l + addr) = val; }
WV_FN void ldsStore128(Lds l, V addr, V4 val) {
  v4u t;
  t.x = val.x; t.y = val.y; t.z = val.z; t.w = val.w;
  *(__attribute__((address_space(3))) v4u*)(l + addr) = t;
}
WV_FN void ldsStore128m(Lds l, V addr, V4 val, B m) {
  if (m) ldsStore128(l, addr, val);
}
WV_FN void ldsStore8m(Lds l, V addr, V val, B m) {
  if (m) *(__attribute__((address_space(3))) uint8_t*)(l + addr) = (uint8_t)val;
}
WV_FN void ldsOr32m(Lds l, V addr, V bits, B m) {
  if (m)
    __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)(l + addr), bits, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wave-uniform accesses (every lane the same address; the value comes back as a scalar)
WV_FN uint32_t ldsLoadS(Lds l, uint32_t addr) { return first(*(__attribute__((address_space(3))) uint32_t*)(l + addr)); }
WV_FN void ldsStoreS(Lds l, uint32_t addr, uint32_t val) { *(__attribute__((address_space(3))) uint32_t*)(l + addr) = val; }

// ---- global memory (the pointers name device or host-mapped memory, never LDS: global_ instructions, which count in
// vmcnt only — a flat_ access would also hold up every wait for an LDS read) ----
#define WV_G(T, p) ((__attribute__((address_space(1))) T*)(p))
WV_FN void gStore8m(uint8_t* base, V off, V val, B m) {
  if (m) WV_G(uint8_t, base)[off] = (uint8_t)val;
}
WV_FN V gLoadU16m(const uint16_t* base, V idx, B m) { return m ? (uint32_t)WV_G(const uint16_t, base)[idx] : 0u; }
WV_FN V gLoad32m(const uint32_t* base, V idx, B m) { return m ? WV_G(const uint32_t, base)[idx] : 0u; }
// a load that must see what other lanes of this wave (or other workgroups, earlier) stored: an agent-scope load goes
// past this CU's L1 to the coherent level
WV_FN V gLoad32Coherent(const uint32_t* base, V idx) {
  return __hip_atomic_load(WV_G(const uint32_t, base) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
WV_FN void gStoreU16m(uint16_t* base, V idx, V val, B m) {
  if (m) WV_G(uint16_t, base)[idx] = (uint16_t)val;
}
WV_FN V gLoad32CoherentM(const uint32_t* base, V idx, B m) {
  return m ? __hip_atomic_load(WV_G(const uint32_t, base) + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
}
WV_FN void gStore32m(uint32_t* base, V idx, V val, B m) {
  if (m) WV_G(uint32_t, base)[idx] = val;
}
WV_FN void gStore32and8m(uint32_t* base32, V idx, V val, uint8_t* base8, V off, V val8, B m) {  // two stores, one mask
  if (m) {
    WV_G(uint32_t, base32)[idx] = val;
#ifndef MRP_CT_EXPERIMENT_NO_PARENT_STORE  // throughput experiment only (paths come out wrong): what the cameFrom bytes cost
    WV_G(uint8_t, base8)[off] = (uint8_t)val8;
#else
    (void)base8; (void)off; (void)val8;
#endif
  }
}
WV_FN void gStore128(uint32_t* base, V idx16, V4 val) {  // idx16 counts 16-byte units
  v4u t;
  t.x = val.x; t.y = val.y; t.z = val.z; t.w = val.w;
  WV_G(v4u, base)[idx16] = t;
}
WV_FN void sync() { __syncthreads(); }

}  // namespace wv
