// The "wave program" vocabulary of the compact search tier (ll_compact.h), as gfx950 code.
//
// ll_compact.h is written against a small set of names — a per-lane 32-bit value `V`, a per-lane predicate `B`, wave-wide
// ballots and lane reads, byte-addressed LDS accesses — so that the SAME source is (a) compiled here into plain per-lane
// HIP (V = uint32_t, B = bool: every name below is one or two instructions) and (b) compiled by tests/support/wave_emu.h
// into a 64-lane lockstep interpretation on the host, where the CPU test-suite replays thousands of harvested searches
// against the oracle before the code ever reaches a GPU.  Control flow in ll_compact.h is wave-uniform by construction
// (it branches only on scalars obtained from ballots / lane reads), which is what makes the lockstep reading exact.
#pragma once
#ifdef MRP_WV_BOOL_PREDICATES  // A/B: the vocabulary with B = bool (round 3's form)
#include "wave_dev_bool.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace wv {

#define WV_FN __device__ __forceinline__
#define WV_ENTRY __device__ __attribute__((noinline))  // a real function: its own register allocation

// V: one 32-bit value per lane (a one-word struct: every operator below is one VALU instruction).
// B: one predicate per lane, held as what it is on this hardware — a 64-bit lane MASK in a scalar register pair.  A
// comparison is a v_cmp that writes the pair (ballot of a compare), `&`, `|`, `!` are scalar s_and / s_or / s_not,
// ballot() is free, and a predicated access or a select takes the pair as its exec / condition mask
// (__builtin_amdgcn_inverse_ballot_w64).  (With B = bool the compiler materialised every combined predicate as
// v_cndmask 0, 1 + v_cmp_ne before it could ballot it.)  Control flow around these is wave-uniform, so exec is all ones
// wherever a mask is made.
struct V {
  uint32_t v;
  WV_FN V() {}
  WV_FN V(uint32_t s) : v(s) {}  // a scalar operand is the same value in every lane
};
struct B {
  uint64_t m;
};
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
typedef uint32_t v2u __attribute__((ext_vector_type(2)));
struct V2 { V x, y; };
struct V4 { V x, y, z, w; };
#define WV_BIN(op) WV_FN V operator op(V a, V b) { return V(a.v op b.v); }
WV_BIN(+)
WV_BIN(-)
WV_BIN(*)
WV_BIN(&)
WV_BIN(|)
WV_BIN(^)
#undef WV_BIN
WV_FN V operator<<(V a, V b) { return V(a.v << (b.v & 31u)); }
WV_FN V operator>>(V a, V b) { return V(a.v >> (b.v & 31u)); }
WV_FN V operator~(V a) { return V(~a.v); }
WV_FN V& operator|=(V& a, V b) { return a = a | b; }
WV_FN V& operator&=(V& a, V b) { return a = a & b; }
WV_FN V& operator+=(V& a, V b) { return a = a + b; }
#define WV_CMP(op) WV_FN B operator op(V a, V b) { return B{__builtin_amdgcn_ballot_w64(a.v op b.v)}; }
WV_CMP(<)
WV_CMP(<=)
WV_CMP(>)
WV_CMP(>=)
WV_CMP(==)
WV_CMP(!=)
#undef WV_CMP
WV_FN B operator&(B a, B b) { return B{a.m & b.m}; }
WV_FN B operator|(B a, B b) { return B{a.m | b.m}; }
WV_FN B operator!(B a) { return B{~a.m}; }
WV_FN bool lanePred(B b) { return __builtin_amdgcn_inverse_ballot_w64(b.m); }  // this lane's bit, as the instruction mask

typedef __attribute__((address_space(3))) uint8_t* Lds;  // the workgroup's LDS window, byte addressed

// The window starts at LDS address 0: the kernels that host the tier declare no static LDS (ll_kernel.hip checks it), so
// every address inside the window is a compile-time constant of the ds_ instructions.
WV_FN Lds windowBase(Lds) { return (Lds)(uintptr_t)0; }
WV_FN V laneId() { return V(threadIdx.x); }
WV_FN uint64_t clock64() { return __builtin_amdgcn_s_memtime(); }  // shader cycles (diagnostic builds)
WV_FN V splat(uint32_t s) { return V(s); }
WV_FN V sel(B c, V a, V b) { return V(lanePred(c) ? a.v : b.v); }
WV_FN B bsplat(bool s) { return B{s ? ~0ull : 0ull}; }
WV_FN uint64_t ballot(B p) { return p.m; }
WV_FN uint32_t readlane(V v, uint32_t lane) { return __builtin_amdgcn_readlane(v.v, lane); }
WV_FN uint32_t first(V v) { return __builtin_amdgcn_readfirstlane(v.v); }
// v with lane `lane` replaced by val (both wave-uniform): v_writelane_b32 (this clang has no builtin for the intrinsic)
extern "C" __device__ int mrp_llvm_writelane(int val, int lane, int old) __asm("llvm.amdgcn.writelane.i32");
WV_FN V writelane(V v, uint32_t val, uint32_t lane) { return V((uint32_t)mrp_llvm_writelane((int)val, (int)lane, (int)v.v)); }
WV_FN V shr1(V v) { return V((uint32_t)__builtin_amdgcn_update_dpp((int)v.v, (int)v.v, 0x138, 0xF, 0xF, false)); }
// lane i of every 16-lane row receives lane i + 1's value; the last lane of a row receives `fill`
WV_FN V rowShl1(V v, uint32_t fill) { return V((uint32_t)__builtin_amdgcn_update_dpp((int)fill, (int)v.v, 0x101, 0xF, 0xF, false)); }
WV_FN V clz(V v) { return V((uint32_t)__builtin_clz(v.v)); }          // v != 0
WV_FN V popc(V v) { return V((uint32_t)__builtin_popcount(v.v)); }
WV_FN float uintAsFloat(uint32_t v) { return __uint_as_float(v); }
WV_FN V sad(V a, V b, V c) { return V(((a.v > b.v ? a.v : b.v) - (a.v > b.v ? b.v : a.v)) + c.v); }  // |a - b| + c  (v_sad_u32)
WV_FN float fmulRn(float a, float b) { return __fmul_rn(a, b); }  // binary32 product, no contraction
WV_FN V cvtF32(V v) { return V(__float_as_uint((float)v.v)); }    // (float)v as bits
WV_FN B leF32(V aBits, float b) { return B{__builtin_amdgcn_ballot_w64(__uint_as_float(aBits.v) <= b)}; }

// ---- LDS (byte addresses inside the window) ----
WV_FN V ldsLoad32(Lds l, V addr) { return V(*(__attribute__((address_space(3))) uint32_t*)(l + addr.v)); }
WV_FN V ldsLoad32m(Lds l, V addr, B m) { return V(lanePred(m) ? *(__attribute__((address_space(3))) uint32_t*)(l + addr.v) : 0u); }
WV_FN V ldsLoadU16(Lds l, V addr) { return V(*(__attribute__((address_space(3))) uint16_t*)(l + addr.v)); }
WV_FN V ldsLoadU8(Lds l, V addr) { return V(*(__attribute__((address_space(3))) uint8_t*)(l + addr.v)); }
WV_FN V2 ldsLoad64m(Lds l, V addr, B m) {
  V2 r{V(0u), V(0u)};
  if (lanePred(m)) {
    const v2u t = *(__attribute__((address_space(3))) v2u*)(l + addr.v);
    r.x = V(t.x);
    r.y = V(t.y);
  }
  return r;
}
WV_FN V2 ldsLoad64(Lds l, V addr) {
  const v2u t = *(__attribute__((address_space(3))) v2u*)(l + addr.v);
  return V2{V(t.x), V(t.y)};
}
WV_FN V4 ldsLoad128(Lds l, V addr) {
  const v4u t = *(__attribute__((address_space(3))) v4u*)(l + addr.v);
  return V4{V(t.x), V(t.y), V(t.z), V(t.w)};
}
WV_FN void ldsStore32m(Lds l, V addr, V val, B m) {
  if (lanePred(m)) *(__attribute__((address_space(3))) uint32_t*)(l + addr.v) = val.v;
}
WV_FN void ldsStore32(Lds l, V addr, V val) { *(__attribute__((address_space(3))) uint32_t*)(l + addr.v) = val.v; }
WV_FN void ldsStore128(Lds l, V addr, V4 val) {
  v4u t;
  t.x = val.x.v; t.y = val.y.v; t.z = val.z.v; t.w = val.w.v;
  *(__attribute__((address_space(3))) v4u*)(l + addr.v) = t;
}
WV_FN void ldsStore128m(Lds l, V addr, V4 val, B m) {
  if (lanePred(m)) ldsStore128(l, addr, val);
}
WV_FN void ldsStore8m(Lds l, V addr, V val, B m) {
  if (lanePred(m)) *(__attribute__((address_space(3))) uint8_t*)(l + addr.v) = (uint8_t)val.v;
}
WV_FN void ldsOr32m(Lds l, V addr, V bits, B m) {
  if (lanePred(m))
    __hip_atomic_fetch_or((__attribute__((address_space(3))) uint32_t*)(l + addr.v), bits.v, __ATOMIC_RELAXED,
                          __HIP_MEMORY_SCOPE_WORKGROUP);
}
// wave-uniform accesses (every lane the same address; the value comes back as a scalar)
WV_FN uint32_t ldsLoadS(Lds l, uint32_t addr) { return __builtin_amdgcn_readfirstlane(*(__attribute__((address_space(3))) uint32_t*)(l + addr)); }
WV_FN void ldsStoreS(Lds l, uint32_t addr, uint32_t val) { *(__attribute__((address_space(3))) uint32_t*)(l + addr) = val; }

// ---- global memory (the pointers name device or host-mapped memory, never LDS: global_ instructions, which count in
// vmcnt only — a flat_ access would also hold up every wait for an LDS read) ----
#define WV_G(T, p) ((__attribute__((address_space(1))) T*)(p))
WV_FN void gStore8m(uint8_t* base, V off, V val, B m) {
  if (lanePred(m)) WV_G(uint8_t, base)[off.v] = (uint8_t)val.v;
}
WV_FN V gLoadU16m(const uint16_t* base, V idx, B m) { return V(lanePred(m) ? (uint32_t)WV_G(const uint16_t, base)[idx.v] : 0u); }
WV_FN V gLoad32m(const uint32_t* base, V idx, B m) { return V(lanePred(m) ? WV_G(const uint32_t, base)[idx.v] : 0u); }
// a load that must see what other lanes of this wave (or other workgroups, earlier) stored: an agent-scope load goes
// past this CU's L1 to the coherent level
WV_FN V gLoad32Coherent(const uint32_t* base, V idx) {
  return V(__hip_atomic_load(WV_G(const uint32_t, base) + idx.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
WV_FN void gStoreU16m(uint16_t* base, V idx, V val, B m) {
  if (lanePred(m)) WV_G(uint16_t, base)[idx.v] = (uint16_t)val.v;
}
WV_FN V gLoad32CoherentM(const uint32_t* base, V idx, B m) {
  return V(lanePred(m) ? __hip_atomic_load(WV_G(const uint32_t, base) + idx.v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u);
}
WV_FN void gStore32m(uint32_t* base, V idx, V val, B m) {
  if (lanePred(m)) WV_G(uint32_t, base)[idx.v] = val.v;
}
WV_FN void gStore32and8m(uint32_t* base32, V idx, V val, uint8_t* base8, V off, V val8, B m) {  // two stores, one mask
  if (lanePred(m)) {
    WV_G(uint32_t, base32)[idx.v] = val.v;
#ifndef MRP_CT_EXPERIMENT_NO_PARENT_STORE  // throughput experiment only (paths come out wrong): what the cameFrom bytes cost
    WV_G(uint8_t, base8)[off.v] = (uint8_t)val8.v;
#else
    (void)base8; (void)off; (void)val8;
#endif
  }
}
WV_FN void gStore128(uint32_t* base, V idx16, V4 val) {  // idx16 counts 16-byte units
  v4u t;
  t.x = val.x.v; t.y = val.y.v; t.z = val.z.v; t.w = val.w.v;
  WV_G(v4u, base)[idx16.v] = t;
}
WV_FN void sync() { __syncthreads(); }

}  // namespace wv
#endif  // MRP_WV_BOOL_PREDICATES
