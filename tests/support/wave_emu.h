// TEST INFRASTRUCTURE — host interpretation of the "wave program" vocabulary of libmultirobotplanning_amd/csrc/wave_dev.h.
//
// The compact search tier (csrc/ll_compact.h) is written against the names below.  On the GPU they are single
// instructions on per-lane registers; here a `V` is all 64 lanes of a wavefront at once and every operation is carried
// out for the 64 lanes in lockstep, which is exactly what the hardware does for wave-uniform control flow (ll_compact.h
// branches only on scalars that come out of ballots / lane reads).  tests/support/emu_ll.cpp wraps one search of the
// tier into a C function; tests/test_compact_emu.py replays harvested low-level searches through it and compares every
// result with the oracle — on the CPU, under -fsanitize=address,undefined if asked, before the code reaches a GPU.
// LDS accesses are bounds-checked against the window (an out-of-window read returns 0 as on the hardware, but is counted
// so that tests can insist there was none); shifts take their amount modulo 32 like v_lshlrev / v_lshrrev.
#pragma once
#include <stdint.h>
#include <string.h>

#include <cstdio>
#include <cstdlib>

namespace wv {

#define WV_FN inline
#define WV_ENTRY inline

constexpr int kLanes = 64;

struct B {
  uint64_t m;
};
struct V {
  uint32_t l[kLanes];
  V() {}
  V(uint32_t s) {  // a scalar operand is the same value in every lane
    for (int i = 0; i < kLanes; ++i) l[i] = s;
  }
};
struct V2 { V x, y; };
struct V4 { V x, y, z, w; };

#define WV_BIN(op)                                              \
  inline V operator op(const V& a, const V& b) {                \
    V r;                                                        \
    for (int i = 0; i < kLanes; ++i) r.l[i] = a.l[i] op b.l[i]; \
    return r;                                                   \
  }
WV_BIN(+)
WV_BIN(-)
WV_BIN(*)
WV_BIN(&)
WV_BIN(|)
WV_BIN(^)
#undef WV_BIN
inline V operator<<(const V& a, const V& b) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = a.l[i] << (b.l[i] & 31u);
  return r;
}
inline V operator>>(const V& a, const V& b) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = a.l[i] >> (b.l[i] & 31u);
  return r;
}
inline V operator~(const V& a) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ~a.l[i];
  return r;
}
inline V& operator|=(V& a, const V& b) { return a = a | b; }
inline V& operator&=(V& a, const V& b) { return a = a & b; }
inline V& operator+=(V& a, const V& b) { return a = a + b; }
#define WV_CMP(op)                                                                   \
  inline B operator op(const V& a, const V& b) {                                     \
    B r{0};                                                                          \
    for (int i = 0; i < kLanes; ++i) r.m |= (uint64_t)(a.l[i] op b.l[i] ? 1 : 0) << i; \
    return r;                                                                        \
  }
WV_CMP(<)
WV_CMP(<=)
WV_CMP(>)
WV_CMP(>=)
WV_CMP(==)
WV_CMP(!=)
#undef WV_CMP
inline B operator&(const B& a, const B& b) { return B{a.m & b.m}; }
inline B operator|(const B& a, const B& b) { return B{a.m | b.m}; }
inline B operator!(const B& a) { return B{~a.m}; }

struct LdsWindow {
  uint8_t* mem;
  uint32_t size;
  uint64_t oobReads, oobWrites;
};
typedef LdsWindow* Lds;

WV_FN Lds windowBase(Lds w) { return w; }
WV_FN uint64_t clock64() { return 0; }
WV_FN V laneId() {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = (uint32_t)i;
  return r;
}
WV_FN V splat(uint32_t s) { return V(s); }
WV_FN V sel(const B& c, const V& a, const V& b) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ((c.m >> i) & 1) ? a.l[i] : b.l[i];
  return r;
}
WV_FN B bsplat(bool s) { return B{s ? ~0ull : 0ull}; }
WV_FN uint64_t ballot(const B& p) { return p.m; }
WV_FN uint32_t readlane(const V& v, uint32_t lane) { return v.l[lane & 63u]; }
WV_FN uint32_t first(const V& v) { return v.l[0]; }
WV_FN V writelane(const V& v, uint32_t val, uint32_t lane) {
  V r = v;
  r.l[lane & 63u] = val;
  return r;
}
WV_FN V shr1(const V& v) {
  V r;
  r.l[0] = v.l[0];
  for (int i = 1; i < kLanes; ++i) r.l[i] = v.l[i - 1];
  return r;
}
WV_FN V rowShl1(const V& v, uint32_t fill) {  // DPP row_shl:1
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = (i & 15) == 15 ? fill : v.l[i + 1];
  return r;
}
WV_FN V clz(const V& v) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = v.l[i] ? (uint32_t)__builtin_clz(v.l[i]) : 32u;
  return r;
}
WV_FN V popc(const V& v) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = (uint32_t)__builtin_popcount(v.l[i]);
  return r;
}
WV_FN float uintAsFloat(uint32_t v) {
  float f;
  memcpy(&f, &v, 4);
  return f;
}
WV_FN V sad(const V& a, const V& b, const V& c) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = (a.l[i] > b.l[i] ? a.l[i] - b.l[i] : b.l[i] - a.l[i]) + c.l[i];
  return r;
}
WV_FN float fmulRn(float a, float b) {  // binary32 product, rounded once (compile with -ffp-contract=off)
  volatile float r = a * b;
  return r;
}
WV_FN V cvtF32(const V& v) {
  V r;
  for (int i = 0; i < kLanes; ++i) {
    const float f = (float)v.l[i];
    memcpy(&r.l[i], &f, 4);
  }
  return r;
}
WV_FN B leF32(const V& aBits, float b) {
  B r{0};
  for (int i = 0; i < kLanes; ++i) {
    float f;
    memcpy(&f, &aBits.l[i], 4);
    r.m |= (uint64_t)(f <= b ? 1 : 0) << i;
  }
  return r;
}

// ---- LDS ----
inline bool ldsIn(Lds l, uint32_t addr, uint32_t bytes, bool write) {
  if (addr <= l->size && bytes <= l->size - addr) {
    if (addr % (bytes > 8 ? 16 : bytes) != 0) {
      std::fprintf(stderr, "wave_emu: misaligned LDS access of %u bytes at %u\n", bytes, addr);
      std::abort();
    }
    return true;
  }
  if (write)
    l->oobWrites += 1;
  else
    l->oobReads += 1;
  return false;
}
template <class T>
inline uint32_t ldsRd(Lds l, uint32_t addr) {
  if (!ldsIn(l, addr, sizeof(T), false)) return 0;
  T t;
  memcpy(&t, l->mem + addr, sizeof(T));
  return (uint32_t)t;
}
WV_FN V ldsLoad32(Lds l, const V& addr) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ldsRd<uint32_t>(l, addr.l[i]);
  return r;
}
WV_FN V ldsLoad32m(Lds l, const V& addr, const B& m) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ((m.m >> i) & 1) ? ldsRd<uint32_t>(l, addr.l[i]) : 0u;
  return r;
}
WV_FN V ldsLoadU16(Lds l, const V& addr) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ldsRd<uint16_t>(l, addr.l[i]);
  return r;
}
WV_FN V ldsLoadU8(Lds l, const V& addr) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ldsRd<uint8_t>(l, addr.l[i]);
  return r;
}
WV_FN V2 ldsLoad64m(Lds l, const V& addr, const B& m) {
  V2 r;
  for (int i = 0; i < kLanes; ++i) {
    r.x.l[i] = r.y.l[i] = 0;
    if (((m.m >> i) & 1) && ldsIn(l, addr.l[i], 8, false)) {
      memcpy(&r.x.l[i], l->mem + addr.l[i], 4);
      memcpy(&r.y.l[i], l->mem + addr.l[i] + 4, 4);
    }
  }
  return r;
}
WV_FN V2 ldsLoad64(Lds l, const V& addr) { return ldsLoad64m(l, addr, B{~0ull}); }
WV_FN V4 ldsLoad128(Lds l, const V& addr) {
  V4 r;
  for (int i = 0; i < kLanes; ++i) {
    uint32_t t[4] = {0, 0, 0, 0};
    if (ldsIn(l, addr.l[i], 16, false)) memcpy(t, l->mem + addr.l[i], 16);
    r.x.l[i] = t[0]; r.y.l[i] = t[1]; r.z.l[i] = t[2]; r.w.l[i] = t[3];
  }
  return r;
}
WV_FN void ldsStore32m(Lds l, const V& addr, const V& val, const B& m) {
  for (int i = 0; i < kLanes; ++i)  // (lane order: a higher lane wins a same-address race, as on the hardware)
    if (((m.m >> i) & 1) && ldsIn(l, addr.l[i], 4, true)) memcpy(l->mem + addr.l[i], &val.l[i], 4);
}
WV_FN void ldsStore32(Lds l, const V& addr, const V& val) { ldsStore32m(l, addr, val, B{~0ull}); }
WV_FN void ldsStore128(Lds l, const V& addr, const V4& val) {
  for (int i = 0; i < kLanes; ++i)
    if (ldsIn(l, addr.l[i], 16, true)) {
      const uint32_t t[4] = {val.x.l[i], val.y.l[i], val.z.l[i], val.w.l[i]};
      memcpy(l->mem + addr.l[i], t, 16);
    }
}
WV_FN void ldsStore128m(Lds l, const V& addr, const V4& val, const B& m) {
  for (int i = 0; i < kLanes; ++i)
    if (((m.m >> i) & 1) && ldsIn(l, addr.l[i], 16, true)) {
      const uint32_t t[4] = {val.x.l[i], val.y.l[i], val.z.l[i], val.w.l[i]};
      memcpy(l->mem + addr.l[i], t, 16);
    }
}
WV_FN void ldsStore8m(Lds l, const V& addr, const V& val, const B& m) {
  for (int i = 0; i < kLanes; ++i)
    if (((m.m >> i) & 1) && ldsIn(l, addr.l[i], 1, true)) l->mem[addr.l[i]] = (uint8_t)val.l[i];
}
WV_FN void ldsOr32m(Lds l, const V& addr, const V& bits, const B& m) {
  for (int i = 0; i < kLanes; ++i)
    if (((m.m >> i) & 1) && ldsIn(l, addr.l[i], 4, true)) {
      uint32_t t;
      memcpy(&t, l->mem + addr.l[i], 4);
      t |= bits.l[i];
      memcpy(l->mem + addr.l[i], &t, 4);
    }
}
WV_FN uint32_t ldsLoadS(Lds l, uint32_t addr) { return ldsRd<uint32_t>(l, addr); }
WV_FN void ldsStoreS(Lds l, uint32_t addr, uint32_t val) {
  if (ldsIn(l, addr, 4, true)) memcpy(l->mem + addr, &val, 4);
}

// ---- global memory ----
WV_FN void gStore8m(uint8_t* base, const V& off, const V& val, const B& m) {
  for (int i = 0; i < kLanes; ++i)
    if ((m.m >> i) & 1) base[off.l[i]] = (uint8_t)val.l[i];
}
WV_FN V gLoadU16m(const uint16_t* base, const V& idx, const B& m) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ((m.m >> i) & 1) ? (uint32_t)base[idx.l[i]] : 0u;
  return r;
}
WV_FN V gLoad32m(const uint32_t* base, const V& idx, const B& m) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ((m.m >> i) & 1) ? base[idx.l[i]] : 0u;
  return r;
}
// a load that must see what other lanes of this wave (or other workgroups, earlier) stored: past the CU's L1 on the device
WV_FN V gLoad32Coherent(const uint32_t* base, const V& idx) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = base[idx.l[i]];
  return r;
}
WV_FN void gStoreU16m(uint16_t* base, const V& idx, const V& val, const B& m) {
  for (int i = 0; i < kLanes; ++i)
    if ((m.m >> i) & 1) base[idx.l[i]] = (uint16_t)val.l[i];
}
WV_FN V gLoad32CoherentM(const uint32_t* base, const V& idx, const B& m) {
  V r;
  for (int i = 0; i < kLanes; ++i) r.l[i] = ((m.m >> i) & 1) ? base[idx.l[i]] : 0u;
  return r;
}
WV_FN void gStore32m(uint32_t* base, const V& idx, const V& val, const B& m) {
  for (int i = 0; i < kLanes; ++i)
    if ((m.m >> i) & 1) base[idx.l[i]] = val.l[i];
}
WV_FN void gStore32and8m(uint32_t* base32, const V& idx, const V& val, uint8_t* base8, const V& off, const V& val8, const B& m) {
  gStore32m(base32, idx, val, m);
  gStore8m(base8, off, val8, m);
}
WV_FN void gStore128(uint32_t* base, const V& idx16, const V4& val) {  // idx16 counts 16-byte units
  for (int i = 0; i < kLanes; ++i) {
    uint32_t* d = base + 4u * idx16.l[i];
    d[0] = val.x.l[i]; d[1] = val.y.l[i]; d[2] = val.z.l[i]; d[3] = val.w.l[i];
  }
}
WV_FN void sync() {}

}  // namespace wv
