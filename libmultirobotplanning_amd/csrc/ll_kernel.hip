// Low-level space-time searches of CBS / ECBS as hand-written HIP for gfx950 (MI355X, CDNA4).
//
// One 64-lane wavefront (== one workgroup) runs ONE low-level search at a time and pulls searches from the batch's
// job queue until it is empty.  Inside a search the reference's sequential semantics are replayed verbatim — the
// results of A*-epsilon depend on the tie order of boost::heap::d_ary_heap — so the open list, the
// focal list and the ordered walk are exact array-heap emulations executed wave-uniformly (scalar control flow), while
// the 64 lanes are used for everything whose order cannot be observed:
//   * the five successor probes of an expansion (bounds, obstacle, vertex constraint, closed/open membership — all
//     one bit test in a per-search (time, cell) bitmap; edge constraints by key compare) run on lanes 0..4;
//   * the O(N) focal heuristics (ecbs.cpp:282-312) are two coalesced row loads of the other agents' positions and a
//     ballot + popcount per discovered successor;
//   * bitmap rows are initialised lazily, 64 words per instruction.
// Search state lives in LDS (fast tier); a search that outgrows it migrates to a per-workgroup HBM arena and
// continues with the same code instantiated for global pointers.  The pop and the pushes of one expansion are batched
// (popFocalEraseOpen, PushChains): their loads are issued together and the sequential heap semantics are resolved in
// registers.  Kernels: mrp_ll_search_kernel (one launch per batch), mrp_ll_persistent_kernel (resident, fed through a
// host job ring), mrp_ll_sipp_kernel / mrp_ll_sipp_persistent_kernel (SIPP, sipp.hpp).
//
// Reference semantics implemented here (file:line in /root/reference):
//   AStarEpsilon::search   include/libMultiRobotPlanning/a_star_epsilon.hpp:86-285
//   AStar::search          include/libMultiRobotPlanning/a_star.hpp:63-161
//   Environment (grid)     example/ecbs.cpp:264-312,352-399,497-510  (example/cbs.cpp identical minus focal parts)
//   heap rules             boost::heap::d_ary_heap<arity<2>, mutable_<true>> as restated in oracle/heap_restated.hpp
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <mutex>
#include <type_traits>

#include "ll_device.h"
#include "wave_dev.h"
#include "ll_compact.h"

namespace mrp {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef int32_t i32x2 __attribute__((ext_vector_type(2)));

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

#define DEVI __device__ __forceinline__

// SIPP node "x" word as the search loop sees it: cell | interval << 16 | (interval ends at INT_MAX) << 31.  Packed forms
// (LDS node records, TierXT heap entries) squeeze it to kSippXBits = 16 + kSippIvBits + 1 bits.
constexpr uint32_t kSippIvBits = 4;  // kSippCap = 15 intervals per cell
static_assert((1u << kSippIvBits) > kSippCap, "interval index width");
constexpr uint32_t kSippXBits = 16 + kSippIvBits + 1;                       // 21
constexpr uint32_t kSippXLow = (1u << (16 + kSippIvBits)) - 1u;             // cell and interval
DEVI uint32_t sippPackX(uint32_t x) { return (x & kSippXLow) | (x >> 31) << (16 + kSippIvBits); }
DEVI uint32_t sippUnpackX(uint32_t v) { return (v & kSippXLow) | ((v >> (16 + kSippIvBits)) & 1u) << 31; }
#ifdef MRP_LL_TRACE  // diagnostic build only (-DMRP_LL_TRACE): progress words in a host-mapped buffer
#define DBG(P, slot, val)                                                                      \
  do {                                                                                         \
    if ((P).debug && blockIdx.x < 4096) { /* all lanes store the same word */                  \
      (P).debug[blockIdx.x * 16 + (slot)] = (uint32_t)(val);                                   \
      __threadfence_system();                                                                  \
    }                                                                                          \
  } while (0)
#else
#define DBG(P, slot, val) do { } while (0)
#endif
#ifdef MRP_LL_TRACE
#define PROF_T0() uint64_t prof_t0__ = __builtin_amdgcn_s_memtime()
#define PROF_ADD(res, k) (res).prof[k] += (uint32_t)(__builtin_amdgcn_s_memtime() - prof_t0__)
#define PROF_INC(res, k, v) (res).prof[k] += (uint32_t)(v)
#define PROF_MARK(var) uint64_t var = __builtin_amdgcn_s_memtime()
#define PROF_SINCE(res, k, var) (res).prof[k] += (uint32_t)(__builtin_amdgcn_s_memtime() - var)
#else
#define PROF_MARK(var) do { } while (0)
#define PROF_SINCE(res, k, var) do { } while (0)
#define PROF_T0() do { } while (0)
#define PROF_ADD(res, k) do { } while (0)
#define PROF_INC(res, k, v) do { } while (0)
#endif

DEVI uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }  // no bool -> int -> bool round trip
DEVI uint32_t rfl(uint32_t v) { return __builtin_amdgcn_readfirstlane(v); }
DEVI int32_t rfli(int32_t v) { return (int32_t)__builtin_amdgcn_readfirstlane((uint32_t)v); }
DEVI uint64_t rfl64(uint64_t v) {
  uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return ((uint64_t)hi << 32) | lo;
}

DEVI uint32_t waveShr1(uint32_t v) {  // lane i receives lane i-1's value (lane 0 keeps its own)
  return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false);
}

// ---- memory tiers and their record formats ---------------------------------------------------------------------
// A heap entry carries its sort key and the node id.  A larger key is a BETTER node in the reference's orders:
//   open  (a_star_epsilon.hpp:312-323, a_star.hpp:168-179): lowest f, then highest g      -> keyOpen
//   focal (a_star_epsilon.hpp:346-366): lowest focalH, then lowest f, then highest g      -> keyFocal
// Entries with equal keys compare EQUAL (the id never takes part), exactly like the reference's comparators; which of
// two equal entries comes out first is decided by the heap layout, which the kernels replay verbatim.
//
// TierHbm (global memory, the search's arena slot; also what SIPP uses): 64-bit entries as laid out in ll_device.h,
//   16-byte node records {x | y<<8 | t<<16 | action<<27, parent id, focalH, position in the open array}.
// TierLds (LDS, the fast tier): everything a small search needs in ~11 KB so that twelve searches share a CU:
//   32-bit entries  [31:22] 1023 - focalH   [21:15] 127 - f   [14:9] g   [8:0] node id
//   one word per node  x | y<<8 | parent<<16,  plus a halfword per node for its position in the open array
//   (focalH, f and g of a node are read from its heap entry, never from the node).
//   A search stays in this tier while it has <= 512 nodes, <= 256 open entries, t < 64, f < 128 and focalH < 1024;
//   beyond any of these it migrates to TierHbm (runJob), records converted one to one.
// Heap arrays are stored with a one-element bias so that the two children (2i+1, 2i+2) of any node form one naturally
// aligned pair -> a single ds_read_b64 / global_load_dwordx4.
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct TierHbm {
  static constexpr int AS = 1;   // address space of the heaps (and bitmap rows)
  static constexpr int NAS = 1;  // ... of the node records
  static constexpr bool kWideNodes = true;   // four words per node (the position of its open entry in word 3)
  static constexpr bool kEntryHasX = false;
  static constexpr bool kEntryXy = false;    // A* tiers: the entry also carries the node's x | y << 8 (TierHybXy)
  static constexpr bool kHybrid = false;
  typedef uint64_t E;
  typedef u64x2 Pair;
  static constexpr uint32_t kFhCap = kFhMax;
  DEVI static E pack(uint32_t fh, uint32_t f, uint32_t g, uint32_t id) {
    const uint32_t key = ((kFhMax - fh) << (kGBits + kFBits)) | ((kFMax - f) << kGBits) | g;
    return ((uint64_t)key << 32) | id;
  }
  DEVI static uint32_t keyFocal(E e) { return (uint32_t)(e >> 32); }
  DEVI static uint32_t keyOpen(E e) { return (uint32_t)(e >> 32) & kOpenKeyMask; }
  DEVI static uint32_t id(E e) { return (uint32_t)e; }
  DEVI static uint32_t f(E e) { return kFMax - ((keyFocal(e) >> kGBits) & kFMax); }
  DEVI static uint32_t g(E e) { return keyFocal(e) & kGMask; }
  DEVI static uint32_t fh(E e) { return kFhMax - (keyFocal(e) >> (kGBits + kFBits)); }
  // walk-queue entry of the ordered walk: the open key of an open-array element and its index
  DEVI static E aux(uint32_t openKey, uint32_t idx) { return ((uint64_t)openKey << 32) | idx; }
  DEVI static uint32_t auxIdx(E e) { return (uint32_t)e; }
  DEVI static E first(E v) { return rfl64(v); }
  DEVI static E fromLane(E v, uint32_t srcLane) {
    const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)v, srcLane);
    const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(v >> 32), srcLane);
    return ((uint64_t)hi << 32) | lo;
  }
  DEVI static E shr1(E v) { return ((uint64_t)waveShr1((uint32_t)(v >> 32)) << 32) | waveShr1((uint32_t)v); }
};

template <uint32_t ID, uint32_t GBITS, uint32_t FBITS, uint32_t FHBITS>
struct TierLdsT {
  static constexpr int AS = 3;
  static constexpr int NAS = 3;
  static constexpr bool kWideNodes = false;  // one word per node + a halfword position array
  static constexpr bool kEntryHasX = false;
  static constexpr bool kEntryXy = false;
  static constexpr bool kHybrid = false;
  typedef uint32_t E;
  typedef u32x2 Pair;
  static constexpr uint32_t kIdBits = ID, kGB = GBITS, kFB = FBITS, kFhB = FHBITS;
  static constexpr uint32_t kMaxNodes = 1u << kIdBits, kMaxRows = 1u << kGB, kFCap = (1u << kFB) - 1;
  static constexpr uint32_t kFhCap = (1u << kFhB) - 1;
  DEVI static E pack(uint32_t fh, uint32_t f, uint32_t g, uint32_t id) {
    E e = ((kFCap - f) << (kIdBits + kGB)) | (g << kIdBits) | id;
    if constexpr (kFhB != 0) e |= (kFhCap - fh) << (kIdBits + kGB + kFB);
    return e;
  }
  DEVI static uint32_t keyFocal(E e) { return e >> kIdBits; }
  DEVI static uint32_t keyOpen(E e) { return (e >> kIdBits) & ((1u << (kGB + kFB)) - 1u); }
  DEVI static uint32_t id(E e) { return e & (kMaxNodes - 1u); }
  DEVI static uint32_t f(E e) { return kFCap - ((e >> (kIdBits + kGB)) & kFCap); }
  DEVI static uint32_t g(E e) { return (e >> kIdBits) & (kMaxRows - 1u); }
  DEVI static uint32_t fh(E e) {
    if constexpr (kFhB != 0) return kFhCap - (e >> (kIdBits + kGB + kFB));
    return 0;
  }
  DEVI static E aux(uint32_t openKey, uint32_t idx) { return (openKey << kIdBits) | idx; }  // idx < 256
  DEVI static uint32_t auxIdx(E e) { return e & (kMaxNodes - 1u); }
  DEVI static E first(E v) { return rfl(v); }
  DEVI static E fromLane(E v, uint32_t srcLane) { return __builtin_amdgcn_readlane(v, srcLane); }
  DEVI static E shr1(E v) { return waveShr1(v); }
};
// (the CBS / ECBS fast tier is ll_compact.h: its 32-bit entries name the state itself, there are no node records)
// SIPP fast tier: [31:21] 2047 - f, [20:11] g (arrival time, <= kGMask), [10:0] node — the whole open key of TierHbm
typedef TierLdsT<11, kGBits, kFBits, 0> TierLdsSipp;
// SIPP middle tier, for a search that has outgrown TierLdsSipp's 2047 nodes: the node records go to the arena, the open
// list stays in LDS as 64-bit entries (the whole fast-tier area: 3072 of them).  Only the open key of the entry is ever
// compared, so the 21 bits around it carry the node's x word (cell 16, interval 4, ends-at-INT_MAX 1): an expansion
// then needs no node read at all, like in the fast tier.
template <int HEAP_AS>
struct TierXT : TierHbm {
  static constexpr int AS = HEAP_AS;
  static constexpr int NAS = 1;
  static constexpr bool kEntryHasX = true;
  static constexpr uint32_t kIdBitsMix = 22;  // kMaxArenaNodes
  DEVI static uint32_t id(E e) { return (uint32_t)e & ((1u << kIdBitsMix) - 1u); }
  DEVI static E withX(E e, uint32_t x) {  // x = cell | interval << 16 | endsAtInf << 31
    const uint32_t v = sippPackX(x);                                           // 21 bits: 10 below the key word, 11 above the key
    // (pack() leaves 2047 - focalH = all ones in the eleven bits above the open key: they are cleared first)
    return (e & ~((uint64_t)kFhMax << (32 + kGBits + kFBits))) | ((uint64_t)(v & 0x3FFu) << kIdBitsMix) |
           ((uint64_t)(v >> 10) << (32 + kGBits + kFBits));
  }
  DEVI static uint32_t xOf(E e) {
    const uint32_t v = (((uint32_t)e >> kIdBitsMix) & 0x3FFu) | ((uint32_t)(e >> (32 + kGBits + kFBits)) & kFhMax) << 10;
    return sippUnpackX(v);
  }
};
typedef TierXT<3> TierMix;   // open list in LDS
typedef TierXT<1> TierHbmX;  // ... in the arena: the last tier of a search on a resident table (entries keep their x word)

// ---- the arena tier of the CBS / ECBS kernels: heaps whose first nTop entries (their top levels) live in LDS ----------
// After a search has left the compact LDS tier its 13 KB of LDS would sit idle while every heap operation walks arrays in
// HBM, root first.  TierHyb keeps entries [0, nTop) of the open list, the focal list and the walk queue in that LDS and
// the rest in the arena: the roots, the first sift-down block and most of every sift-up chain cost no memory round trip
// at all, and a heap that has at most nTop entries never leaves LDS.  nTop is odd, so an aligned child pair (c, c + 1),
// c odd, lies on one side.  nTop = 0 (no LDS tier configured) degenerates to the plain arena tier.
struct HybRef {
  __attribute__((address_space(3))) uint64_t* top;
  uint64_t* rest;
  uint32_t nTop, i;
  DEVI operator uint64_t() const { return i < nTop ? top[i] : rest[i]; }
  DEVI void operator=(uint64_t e) const {
    if (i < nTop)
      top[i] = e;
    else
      rest[i] = e;
  }
};
struct HybPtr {
  __attribute__((address_space(3))) uint64_t* top;  // element i at top[i] (biased like the arena pointers)
  uint64_t* rest;
  uint32_t nTop;
  DEVI HybRef operator[](uint32_t i) const { return HybRef{top, rest, nTop, i}; }
};
struct TierHyb : TierHbm {
  static constexpr bool kHybrid = true;
};
// ... and, for arenas of at most 65 536 nodes (the conflict-tree drivers' default), the node id takes 16 bits of the
// entry's low word and the node's x | y << 8 the other 16: the expansion then knows its cell from the entry alone and
// requests its bitmap word and the other agents' rows TOGETHER with the node record (which it still needs for the open
// position) instead of after it — one memory round trip less per expansion.
struct TierHybXy : TierHbm {
  static constexpr bool kHybrid = true;
  static constexpr bool kEntryXy = true;
  DEVI static uint32_t id(E e) { return (uint32_t)e & 0xFFFFu; }
  DEVI static uint32_t xyOf(E e) { return ((uint32_t)e >> 16) & 0xFFFFu; }
  DEVI static E withXy(E e, uint32_t xy) { return e | ((uint64_t)(xy & 0xFFFFu) << 16); }
};

template <class T, bool HYB = T::kHybrid>
struct HeapPtr {
  typedef __attribute__((address_space(T::AS))) typename T::E* type;
};
template <class T>
struct HeapPtr<T, true> {
  typedef HybPtr type;
};

template <class T>
struct Mem {
  typedef typename T::E E;
  typedef typename HeapPtr<T>::type PE;
  typedef __attribute__((address_space(T::AS))) typename T::Pair* PPair;
  typedef __attribute__((address_space(T::AS))) uint32_t* P32;
  typedef __attribute__((address_space(T::AS))) uint16_t* P16;
  typedef __attribute__((address_space(T::NAS))) uint32_t* PN32;
  typedef __attribute__((address_space(T::NAS))) u32x4* PNode4;
  PN32 nodes;     // TierLds: one word per node; TierHbm: four words per node
  P16 pos;       // TierLds only: position of the node's entry in the open array
  P16 gOf;       // TierLdsSipp only: arrival time of the node
  PE open;       // biased: element i at open[i] (the pointer already includes the +1 bias)
  PE focal;
  PE aux;        // std::priority_queue of the ordered walk
  P32 bits;      // (time, cell) bitmap: 1 = obstacle | vertex constraint | already discovered
  uint32_t capNodes, capHeap, capRows, rowWords;  // capHeap: entries per heap array (open / focal / walk queue)
};

template <class T>
DEVI void setPos(Mem<T>& m, uint32_t id, uint32_t idx) {
  if constexpr (!T::kWideNodes)
    m.pos[id] = (uint16_t)idx;
  else
    m.nodes[id * 4 + 3] = idx;
}
// x | y << 8 of a node and the position of its entry in the open array (both wave-uniform)
template <class T>
DEVI void nodeXyPos(Mem<T>& m, uint32_t id, uint32_t& xy, uint32_t& pos) {
  if constexpr (T::AS == 3) {
    const uint32_t w = m.nodes[id];
    const uint32_t p = m.pos[id];
    xy = rfl(w) & 0xFFFFu;
    pos = rfl(p);
  } else {
    const u32x4 nd = ((typename Mem<T>::PNode4)m.nodes)[id];
    xy = rfl(nd.x) & 0xFFFFu;
    pos = rfl(nd.w);
  }
}
template <class T>
DEVI void nodeXyParent(Mem<T>& m, uint32_t id, uint32_t& xy, uint32_t& parent) {
  if constexpr (T::AS == 3) {
    const uint32_t w = rfl(m.nodes[id]);
    xy = w & 0xFFFFu;
    parent = w >> 16;
  } else {
    const u32x4 nd = ((typename Mem<T>::PNode4)m.nodes)[id];
    xy = rfl(nd.x) & 0xFFFFu;
    parent = rfl(nd.y);
  }
}

struct Ctx {  // wave-uniform job context
  uint32_t dimx, dimy, wpr, gx, gy, sx, sy;
  int32_t lastGoal;
  float w;
  uint32_t nVc, nEc;
  const uint32_t* vc;       // generic pointers: LDS, arena copy, or (oversized lists only) host memory
  const uint32_t* ec;
  const uint32_t* obst;     // global obstacle bitmap
  const uint16_t* paths;    // generic: LDS copy, arena copy, or host memory
  __attribute__((address_space(3))) const uint16_t* pathsLds;  // the same table when it is the LDS copy (ds_read), else null
  uint32_t nAgentsPad, tPad;
  int64_t maxExp;
  volatile uint32_t* debug;
};

struct SState {  // wave-uniform search state (kept in SGPRs by construction)
  uint32_t nNodes, nOpen, nFocal, rowsReady;
  int32_t bestF;
  int64_t expansions;
};

enum : int { RUN_MIGRATE_NODES = -1, RUN_MIGRATE_ROWS = -2 };
constexpr int32_t ST_CAP_FOCAL = 7;

template <class T>
DEVI typename T::E ldU(typename Mem<T>::PE p, uint32_t i) { return T::first(p[i]); }

// the aligned pair (i, i + 1), i odd: one load
template <class T>
DEVI typename T::Pair hLoadPair(typename Mem<T>::PE p, uint32_t i) {
  if constexpr (T::kHybrid) {
    if (i < p.nTop) return *(__attribute__((address_space(3))) u64x2*)(p.top + i);
    return *(u64x2*)(p.rest + i);
  } else {
    return *(typename Mem<T>::PPair)(p + i);
  }
}

template <class T>
DEVI void ldPair(typename Mem<T>::PE p, uint32_t i, typename T::E& a, typename T::E& b) {  // i odd -> aligned pair
  const typename T::Pair v = hLoadPair<T>(p, i);
  a = T::first(v.x);
  b = T::first(v.y);
}

// ---- heap primitives ------------------------------------------------------------------------------------------
// The heaps are replayed EXACTLY (same array layout after every operation as boost::heap::d_ary_heap / libstdc++'s
// std::push_heap / std::pop_heap would have), but not one element at a time: the data-independent part of every
// operation is done by all lanes at once so that an operation costs O(1) memory round trips instead of O(log n):
//   * sift-up      : lane k loads the k-th ancestor; one ballot finds where the sequential loop would have stopped;
//                    the ancestors below that point move down one level in a single parallel store.
//   * sift-down    : which child is "the larger one" does not depend on the element being sifted, so 63 lanes load
//                    the child pairs of a whole 6-level subtree in one instruction and every lane then decides from
//                    two ballots whether its node is on the path (followPath: no further memory latency, no scalar
//                    walk); repeated per 6 levels.
//   * erase        : the unconditional bubble-to-root is a one-level shift of the ancestor chain (parallel).
// KEY selects the comparator: 0 = open (f asc, g desc), 1 = focal (focalH, f asc, g desc), 2 = walk queue (an open key
// in the entry's key field).  POS=true maintains handle -> position for the node (open list only).
template <class T, int KEY>
DEVI uint32_t keyOf(typename T::E e) { return KEY == 0 ? T::keyOpen(e) : T::keyFocal(e); }
template <class T, int KEY>
DEVI bool kLess(typename T::E a, typename T::E b) {  // the reference's "operator<": a is WORSE than b
  return keyOf<T, KEY>(a) < keyOf<T, KEY>(b);
}

template <class T, bool POS>
DEVI void heapStore(Mem<T>& m, typename Mem<T>::PE heap, uint32_t idx, typename T::E e) {
  heap[idx] = e;
  if (POS) setPos<T>(m, T::id(e), idx);
}

// boost siftup / libstdc++ __push_heap from position idx: while less(parent, e) the parent moves down.
template <class T, int KEY, bool POS>
DEVI void siftUp(Mem<T>& m, typename Mem<T>::PE heap, uint32_t idx, typename T::E e) {
  typedef typename T::E E;
  const uint32_t lane = threadIdx.x;
  const uint32_t depth = 31u - (uint32_t)__builtin_clz(idx + 1);  // number of ancestors of idx
  uint32_t stop = 0;
  if (depth != 0) {
    const bool act = lane < depth;
    const uint32_t anc = act ? ((idx + 1) >> (lane + 1)) - 1 : 0;     // lane k: k-th ancestor
    const E ae = heap[anc];
    const uint64_t worse = ballot64(act && kLess<T, KEY>(ae, e));
    stop = (uint32_t)__builtin_ctzll(~worse);                          // first ancestor that is not worse than e
    if (lane < stop) {                                                 // ancestors 0..stop-1 move down one level
      const uint32_t dest = ((idx + 1) >> lane) - 1;
      heap[dest] = ae;
      if (POS) setPos<T>(m, T::id(ae), dest);
    }
  }
  heapStore<T, POS>(m, heap, ((idx + 1) >> stop) - 1, e);
}

// Which nodes of a 6-level block lie on the sift-down path, decided by all lanes at once instead of a scalar walk
// over the masks: node l (lane l < 63; 1-based number n = l + 1) is reached iff every ancestor lets the hole pass
// (`go`) and turned towards l (`right` bit == the matching digit of n).  The ancestors of a node of a 63-node tree are
// among its first 31 nodes, so both tests are 32-bit masks that depend on the lane only.
//   anc   : bit a set  <=>  node a is an ancestor of this lane's node
//   needR : bit a set  <=>  ... and the path to this lane's node leaves a through its RIGHT child
struct PathLanes {
  uint32_t anc, needR;
};
DEVI PathLanes pathLanes() {
  const uint32_t n = threadIdx.x + 1;
  PathLanes pl;
  pl.anc = 0;
  pl.needR = 0;
#pragma unroll
  for (uint32_t k = 1; k <= 5; ++k) {
    const uint32_t a = n >> k;  // 1-based number of the k-th ancestor (0: none)
    if (a != 0 && n < 64) {
      pl.anc |= 1u << (a - 1);
      pl.needR |= ((n >> (k - 1)) & 1u) << (a - 1);
    }
  }
  return pl;
}
// Follows the path of one block: `go` / `right` are this lane's answers for its node.  Returns the lanes on the path
// (each pulls its chosen child up), the number of levels descended and the new hole relative to the block's root.
DEVI bool followPath(const PathLanes& pl, bool go, bool right, uint32_t& steps, uint32_t& rel) {
  const uint64_t goMask = ballot64(go);
  const uint64_t rightMask = ballot64(right);
  const uint32_t goLo = (uint32_t)goMask, rLo = (uint32_t)rightMask;
  const bool reached = ((goLo & pl.anc) == pl.anc) && (((rLo ^ pl.needR) & pl.anc) == 0u);
  const bool onPath = reached && go;
  const uint64_t pathMask = ballot64(onPath);
  steps = (uint32_t)__popcll(pathMask);
  rel = 0;
  if (pathMask) {
    const uint32_t d = 63u - (uint32_t)__builtin_clzll(pathMask);  // deepest node on the path (levels are index-ordered)
    rel = 2 * d + 1 + (uint32_t)((rightMask >> d) & 1ull);
  }
  return onPath;
}

// Moves the hole at `idx` down a heap of n elements.
//   STL=false (boost siftdown): prefer the FIRST maximal child; stop in front of a child that is less than x; x is
//             stored at the final hole.
//   STL=true  (libstdc++ __adjust_heap): prefer the right child unless it is less than the left one; always descend
//             to a leaf; the final hole index is returned (the caller then sifts its value up from there).
// Per 6 levels: one pair load per lane (63 lanes = the whole subtree below the hole), two ballots, a scalar walk
// over the two bit masks, and ONE predicated store in which every node on the path pulls its chosen child up.
template <class T, int KEY, bool POS, bool STL>
DEVI uint32_t descend(Mem<T>& m, typename Mem<T>::PE heap, uint32_t n, uint32_t idx, typename T::E x) {
  typedef typename T::E E;
  const uint32_t lane = threadIdx.x;
  const uint32_t lv = 31u - (uint32_t)__builtin_clz(lane + 1);  // level of this lane inside a 6-level subtree
  const uint32_t off = (lane + 1) - (1u << lv);                 // position inside that level
  const uint32_t xk = keyOf<T, KEY>(x);
  const PathLanes pl = pathLanes();
  for (;;) {
    const uint32_t node = ((idx + 1) << lv) - 1 + off;          // lane l < 63 owns this node of the subtree
    const uint32_t c = 2 * node + 1;
    const bool has = (lane < 63) && (c < n);
    typename T::Pair pr;
    pr.x = 0;
    pr.y = 0;
    if (has) pr = hLoadPair<T>(heap, c);                        // children (c, c+1): one aligned load
    const uint32_t kl = keyOf<T, KEY>(pr.x);
    const uint32_t kr = keyOf<T, KEY>(pr.y);
    const bool hasR = has && (c + 1 < n);
    const bool right = hasR && (STL ? !(kr < kl) : (kl < kr));
    const E pe = right ? pr.y : pr.x;
    const uint32_t pk = right ? kr : kl;
    const bool go = has && (STL || !(pk < xk));                 // the hole moves below this node
    uint32_t rel, steps;
    if (followPath(pl, go, right, steps, rel)) {                // every node on the path pulls its chosen child up
      heap[node] = pe;
      if (POS) setPos<T>(m, T::id(pe), node);
    }
    idx = ((idx + 1) << steps) - 1 + (rel + 1 - (1u << steps)); // absolute index of the new hole
    if (steps < 6) break;
  }
  if (!STL) heapStore<T, POS>(m, heap, idx, x);
  return idx;
}

// boost pop: swap(front, back), drop back, siftdown(0)
template <class T, int KEY, bool POS>
DEVI void heapPop(Mem<T>& m, typename Mem<T>::PE heap, uint32_t& n) {
  n -= 1;
  if (n == 0) return;
  const typename T::E last = ldU<T>(heap, n);
  descend<T, KEY, POS, false>(m, heap, n, 0, last);
}

// ---- batched operations of one expansion ------------------------------------------------------------------------
// An expansion pops one element and pushes up to five.  Done one heap operation at a time that is a chain of ~25
// dependent memory round trips; the results of the operations, however, depend on each other only through a handful
// of heap entries, so the loads of ALL of them are issued first and the sequential semantics are then resolved in
// registers:
//   * pushes: the sift-up chain of the k-th new element is held LEVEL-MAJOR — lane L owns the chain's node at tree
//     level L (root = level 0), for every k.  Two chains that pass through the same heap position do so at the same
//     level, i.e. in the same lane, so "what did an earlier push of this expansion leave at this position" is a
//     per-lane select; the one-level move of the ancestors that a sift-up performs is a one-lane shift of the wave
//     (DPP wave_shr:1).  Five pushes into two heaps cost one round trip.
//   * pops: the loads of the focal and the open sift-down (which child is the larger one does not depend on the
//     element being sifted) are issued together, and the moved "last" elements are fetched with them.
constexpr uint32_t kNoPos = 0xFFFFFFFFu;

template <class T>
struct PushChains {          // sift-up chains of the (up to five) pushes of one expansion into one heap
  typedef typename T::E E;
  uint32_t pos[5];           // lane L: heap position of the chain's node at level L (kNoPos: none)
  E val[5];                  // lane L: the entry there before any of these pushes
  // `mask` bit k: successor k is pushed; pushed elements take positions n0, n0+1, ... in ascending k
  DEVI void load(typename Mem<T>::PE heap, uint32_t n0, uint32_t mask) {
    const uint32_t lane = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      pos[k] = kNoPos;
      val[k] = 0;
      if ((mask >> k) & 1u) {
        const uint32_t p = n0 + (uint32_t)__builtin_popcount(mask & ((1u << k) - 1u));
        const uint32_t d = 31u - (uint32_t)__builtin_clz(p + 1);  // level of p == number of ancestors
        if (lane <= d) pos[k] = ((p + 1) >> (d - lane)) - 1;
        if (lane < d) val[k] = heap[pos[k]];
      }
    }
  }
  // boost siftup / libstdc++ __push_heap of e[k] at its position, for k ascending — the same stores a one-at-a-time
  // replay ends with (positions written twice are written in push order).
  template <int KEY, bool POS>
  DEVI void resolve(Mem<T>& m, typename Mem<T>::PE heap, uint32_t n0, uint32_t mask, const E (&e)[5]) {
    const uint32_t lane = threadIdx.x;
    E nv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      nv[k] = 0;
      if ((mask >> k) & 1u) {
        const uint32_t p = n0 + (uint32_t)__builtin_popcount(mask & ((1u << k) - 1u));
        const uint32_t d = 31u - (uint32_t)__builtin_clz(p + 1);
        E v = val[k];
#pragma unroll
        for (int j = 0; j < k; ++j)  // what earlier pushes of this expansion left on this chain
          if (((mask >> j) & 1u) && pos[j] == pos[k] && pos[k] != kNoPos) v = nv[j];
        const uint64_t worse = ballot64(lane < d && kLess<T, KEY>(v, e[k]));
        const uint64_t notWorse = ~worse & ((1ull << d) - 1ull);
        const int32_t sLvl = notWorse ? 63 - (int32_t)__builtin_clzll(notWorse) : -1;  // deepest ancestor that stays
        const E sh = T::shr1(v);
        const E nk = (int32_t)lane <= sLvl ? v : ((int32_t)lane == sLvl + 1 ? e[k] : sh);
        if ((int32_t)lane > sLvl && lane <= d) {
          heap[pos[k]] = nk;
          if (POS) setPos<T>(m, T::id(nk), pos[k]);
        }
        nv[k] = nk;
      }
    }
  }
};

// One 6-level block of a sift-down whose child pairs have been loaded (see descend): follows the path, pulls the
// chosen children up, returns the new hole; `more` = the block was left through its bottom.
template <class T, int KEY, bool POS>
DEVI uint32_t descendBlock(Mem<T>& m, typename Mem<T>::PE heap, const PathLanes& pl, uint32_t idx, uint32_t xk,
                           typename T::Pair pr, uint32_t node, bool has, bool hasR, bool& more) {
  typedef typename T::E E;
  const uint32_t kl = keyOf<T, KEY>(pr.x);
  const uint32_t kr = keyOf<T, KEY>(pr.y);
  const bool right = hasR && (kl < kr);
  const E pe = right ? pr.y : pr.x;
  const uint32_t pk = right ? kr : kl;
  const bool go = has && !(pk < xk);
  uint32_t rel, steps;
  if (followPath(pl, go, right, steps, rel)) {
    heap[node] = pe;
    if (POS) setPos<T>(m, T::id(pe), node);
  }
  more = steps == 6;
  return ((idx + 1) << steps) - 1 + (rel + 1 - (1u << steps));
}

// a_star_epsilon.hpp:191-192 of one expansion: focalSet.pop() and openSet.erase(handle of the same node), with the
// memory traffic of the two heaps overlapped.  curPos = position of the popped node in the open array.
template <class T>
DEVI void popFocalEraseOpen(Mem<T>& m, uint32_t& nFocal, uint32_t& nOpen, uint32_t curPos) {
  typedef typename T::E E;
  typedef typename T::Pair Pair;
  const uint32_t lane = threadIdx.x;
  const uint32_t lv = 31u - (uint32_t)__builtin_clz(lane + 1);
  const uint32_t off = (lane + 1) - (1u << lv);
  const PathLanes pl = pathLanes();
  // ---- loads that depend on nothing but the sizes
  nFocal -= 1;
  const uint32_t nOld = nOpen;
  nOpen -= 1;
  E lastFv = 0, lastOv = 0;
  if (nFocal > 0) lastFv = m.focal[nFocal];
  if (nOpen > 0) lastOv = m.open[nOld - 1];
  const uint32_t depth = 31u - (uint32_t)__builtin_clz(curPos + 1);
  const bool act = lane < depth;
  const uint32_t anc = act ? ((curPos + 1) >> (lane + 1)) - 1 : 0;
  E ae = 0;
  if (depth != 0) ae = m.open[anc];
  // first block of the focal sift-down: does not depend on the element being sifted
  uint32_t idxF = 0, idxO = 0;
  bool moreF = nFocal > 0, moreO = nOpen > 0;
  Pair prF;
  prF.x = 0; prF.y = 0;
  const uint32_t nodeF0 = (1u << lv) - 1 + off;
  const bool hasF0 = moreF && lane < 63 && (2 * nodeF0 + 1 < nFocal);
  if (hasF0) prF = hLoadPair<T>(m.focal, 2 * nodeF0 + 1);
  // ---- open: every ancestor of curPos moves down one level (boost erase = bubble to the root, then pop)
  if (act) {
    const uint32_t dest = ((curPos + 1) >> lane) - 1;
    m.open[dest] = ae;
    setPos<T>(m, T::id(ae), dest);
  }
  // the element that pop() moves to the root: the last one — which the shift above has just overwritten if the erased
  // node WAS the last one (then it is the erased node's parent)
  E lastO = T::first(lastOv);
  if (curPos == nOld - 1 && depth != 0) lastO = T::fromLane(ae, 0);
  const E lastF = T::first(lastFv);
  const uint32_t xkF = T::keyFocal(lastF), xkO = T::keyOpen(lastO);
  // ---- sift-downs, block by block, both heaps per round trip
  bool firstF = true;
  for (;;) {
    Pair prO;
    prO.x = 0; prO.y = 0;
    const uint32_t nodeO = ((idxO + 1) << lv) - 1 + off;
    const bool hasO = moreO && lane < 63 && (2 * nodeO + 1 < nOpen);
    if (hasO) prO = hLoadPair<T>(m.open, 2 * nodeO + 1);
    uint32_t nodeF = nodeF0;
    bool hasF = hasF0;
    if (!firstF) {
      nodeF = ((idxF + 1) << lv) - 1 + off;
      hasF = moreF && lane < 63 && (2 * nodeF + 1 < nFocal);
      prF.x = 0; prF.y = 0;
      if (hasF) prF = hLoadPair<T>(m.focal, 2 * nodeF + 1);
    }
    firstF = false;
    if (moreF)
      idxF = descendBlock<T, 1, false>(m, m.focal, pl, idxF, xkF, prF, nodeF, hasF, hasF && (2 * nodeF + 2 < nFocal), moreF);
    if (moreO)
      idxO = descendBlock<T, 0, true>(m, m.open, pl, idxO, xkO, prO, nodeO, hasO, hasO && (2 * nodeO + 2 < nOpen), moreO);
    if (!moreF && !moreO) break;
  }
  if (nFocal > 0) m.focal[idxF] = lastF;
  if (nOpen > 0) heapStore<T, true>(m, m.open, idxO, lastO);
}

// ---- ordered walk (open.ordered_begin(), a_star_epsilon.hpp:141-152) ----------------------------------------
// libstdc++ std::priority_queue<…> restated: push = __push_heap, pop = __pop_heap/__adjust_heap (bits/stl_heap.h).
template <class T>
DEVI typename T::E auxPop(Mem<T>& m, uint32_t& npq) {
  typedef typename T::E E;
  const E result = ldU<T>(m.aux, 0);
  npq -= 1;
  if (npq > 0) {
    const E value = ldU<T>(m.aux, npq);  // *(last - 1)
    const uint32_t hole = descend<T, 2, false, true>(m, m.aux, npq, 0, value);
    siftUp<T, 2, false>(m, m.aux, hole, value);
  }
  return result;  // open key and index into the open array
}

template <class T>
DEVI void orderedWalk(Mem<T>& m, SState& s, const Ctx& c, int32_t oldBest, DevResult& res) {
  typedef typename T::E E;
  // int * float products in binary32, no contraction (a_star_epsilon.hpp:145,149)
  const float lo = __fmul_rn((float)oldBest, c.w);
  const float hi = __fmul_rn((float)s.bestF, c.w);
  // The queue entry of a visited element carries its open key, so f is known without touching the open array again;
  // the two children are pushed with one round trip (PushChains).  In a long search the walks are most of the time
  // (every bestF increase visits every open node with f <= hi), so a round trip per visited node matters.
  uint32_t npq = 0;
  E curA = T::aux(T::keyOpen(ldU<T>(m.open, 0)), 0);  // index 0
  for (;;) {
    const uint32_t cur = T::auxIdx(curA);
    const uint32_t first = 2 * cur + 1;
    if (first < s.nOpen) {
      E e1, e2;
      ldPair<T>(m.open, first, e1, e2);
      E ee[5];
      ee[0] = T::aux(T::keyOpen(e1), first);
      ee[1] = T::aux(T::keyOpen(e2), first + 1);
      ee[2] = ee[3] = ee[4] = 0;
      const uint32_t pm = first + 1 < s.nOpen ? 3u : 1u;
      PushChains<T> pc;
      pc.load(m.aux, npq, pm);
      pc.template resolve<2, false>(m, m.aux, npq, pm, ee);  // == __push_heap of the children in index order
      npq += pm == 3u ? 2u : 1u;
    }
    PROF_INC(res, 7, 1);

    const float fv = (float)(int32_t)T::f(curA);
    if (fv > lo && fv <= hi) {
      const E e = ldU<T>(m.open, cur);
      siftUp<T, 1, false>(m, m.focal, s.nFocal, e);
      s.nFocal += 1;
    }
    if (fv > hi) break;
    if (npq == 0) break;
    curA = auxPop<T>(m, npq);
  }
}

// ---- lazy bitmap rows: row t = obstacles | vertex constraints at time t | states already discovered --------
template <class T>
DEVI void ensureRows(Mem<T>& m, SState& s, const Ctx& c, uint32_t t1, typename Mem<T>::P32 obstLocal, bool useLocal) {
  if (t1 < s.rowsReady) return;
  const uint32_t lane = threadIdx.x;
  uint32_t r0 = s.rowsReady;
  uint32_t r1 = t1 + 4;
  if (r1 > m.capRows) r1 = m.capRows;
  for (uint32_t r = r0; r < r1; ++r)
    for (uint32_t wd = lane; wd < c.wpr; wd += 64)
      m.bits[r * m.rowWords + wd] = useLocal ? obstLocal[wd] : c.obst[wd];
  __syncthreads();
  for (uint32_t j = lane; j < c.nVc; j += 64) {
    uint32_t v = c.vc[j];  // t << 16 | y << 8 | x
    uint32_t tt = v >> 16, cell = ((v >> 8) & 0xFFu) * c.dimx + (v & 0xFFu);
    if (tt >= r0 && tt < r1)
      __hip_atomic_fetch_or(m.bits + tt * m.rowWords + (cell >> 5), 1u << (cell & 31), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  __syncthreads();
  s.rowsReady = r1;
}

// ---- one search in one tier ------------------------------------------------------------------------------------
template <class T, bool EPS>
DEVI void initSearch(Mem<T>& m, SState& s, const Ctx& c) {
  uint32_t h0 = (c.sx > c.gx ? c.sx - c.gx : c.gx - c.sx) + (c.sy > c.gy ? c.sy - c.gy : c.gy - c.sy);
  s.nNodes = 1;
  s.nOpen = 1;
  s.nFocal = EPS ? 1 : 0;
  s.rowsReady = 0;
  s.bestF = (int32_t)h0;
  s.expansions = 0;
  if constexpr (T::AS == 3) {
    m.nodes[0] = c.sx | (c.sy << 8) | (0xFFFFu << 16);
    m.pos[0] = 0;
  } else {
    u32x4 n0;
    n0.x = c.sx | (c.sy << 8) | (0u << 16) | (7u << 27);
    n0.y = kNoParent;
    n0.z = 0;
    n0.w = 0;
    ((typename Mem<T>::PNode4)m.nodes)[0] = n0;
  }
  typename T::E e0 = T::pack(0, h0, 0, 0);
  if constexpr (T::kEntryXy) e0 = T::withXy(e0, c.sx | (c.sy << 8));
  m.open[0] = e0;
  if (EPS) m.focal[0] = e0;
}

// Returns a status (>= 0) when the search ended, or RUN_MIGRATE_* when this tier is too small to continue.
template <class T, bool EPS>
DEVI int runSearch(Mem<T>& m, SState& s, const Ctx& c, typename Mem<T>::P32 obstLocal, bool useLocal, DevResult& res,
                   uint16_t* outPath) {
  typedef typename T::E E;
  const uint32_t lane = threadIdx.x;
  uint32_t dbgIter = 0;
  // edge-constraint keys, one per lane (lists longer than a wave keep their tail in memory)
  const uint32_t ecReg = lane < c.nEc ? c.ec[lane] : 0xFFFFFFFFu;
  // successor of this lane in the reference's order Wait, Left, Right, Up, Down (ecbs.cpp:365-398) on lanes 0..4
  const int32_t dx = (lane == 2) - (lane == 1);
  const int32_t dy = (lane == 3) - (lane == 4);
  for (;;) {
    DBG(c, 5, ++dbgIter);
    PROF_MARK(profTop);
    if (s.nOpen == 0) return ST_NO_SOLUTION;
    const E topE = ldU<T>(m.open, 0);
    E curE = topE;
    if (EPS) {
      const int32_t oldBest = s.bestF;
      s.bestF = (int32_t)T::f(topE);
      if (s.bestF > oldBest) {
        PROF_T0();
        orderedWalk<T>(m, s, c, oldBest, res);
        PROF_ADD(res, 0);
        PROF_INC(res, 6, 1);
      }
      curE = ldU<T>(m.focal, 0);
    }
    // f, g (== time: every action costs 1) and focalH of the popped node are in its entry
    const uint32_t curId = T::id(curE);
    const uint32_t t = T::g(curE);
    const uint32_t curFh = T::fh(curE);
    uint32_t xy, curPos = 0;
    uint32_t curPosV = 0;  // kEntryXy: the open position as loaded (waited for only where popFocalEraseOpen needs it)
    if constexpr (T::kEntryXy) {
      xy = T::xyOf(curE);
      if (EPS) curPosV = m.nodes[curId * 4 + 3];
    } else {
      nodeXyPos<T>(m, curId, xy, curPos);
    }
    const uint32_t x = xy & 0xFF, y = xy >> 8;
    const bool isGoal = (x == c.gx) && (y == c.gy) && ((int32_t)t > c.lastGoal);
    DBG(c, 6, xy | (t << 16));
    DBG(c, 7, isGoal ? 1 : 2);
    if (!isGoal) {
      if (s.nNodes + 5 > m.capNodes || s.nOpen + 5 > m.capHeap) return RUN_MIGRATE_NODES;
      if (t + 1 >= m.capRows) return RUN_MIGRATE_ROWS;
      // a successor adds at most two conflicts per other agent to focalH: leave the compact tier before its field can
      // overflow (TierHbm reports ST_CAP_FOCAL below instead)
      if (T::AS == 3 && curFh + 2 * c.nAgentsPad > T::kFhCap) return RUN_MIGRATE_NODES;
    }
    // other agents' positions at t and t+1 (issued early; consumed after the heap pops)
    uint32_t a0 = kEmptyCell, b0 = kEmptyCell, a1 = kEmptyCell, b1 = kEmptyCell;
    const uint16_t* rowA = nullptr;
    const uint16_t* rowB = nullptr;
    if (EPS && c.nAgentsPad && !isGoal) {
      const uint32_t ra = t < c.tPad ? t : c.tPad - 1;
      const uint32_t rb = (t + 1) < c.tPad ? (t + 1) : c.tPad - 1;
      rowA = c.paths + (size_t)ra * c.nAgentsPad;
      rowB = c.paths + (size_t)rb * c.nAgentsPad;
      if (c.pathsLds) {  // the usual case: LDS reads proper, not flat loads through the LDS aperture
        if (lane < c.nAgentsPad) {  // rows are n_agents_pad (multiple of 16) entries long
          a0 = c.pathsLds[ra * c.nAgentsPad + lane];
          b0 = c.pathsLds[rb * c.nAgentsPad + lane];
        }
        if (64 + lane < c.nAgentsPad) {
          a1 = c.pathsLds[ra * c.nAgentsPad + 64 + lane];
          b1 = c.pathsLds[rb * c.nAgentsPad + 64 + lane];
        }
      } else {
        if (lane < c.nAgentsPad) {
          a0 = rowA[lane];
          b0 = rowB[lane];
        }
        if (64 + lane < c.nAgentsPad) {
          a1 = rowA[64 + lane];
          b1 = rowB[64 + lane];
        }
      }
    }

    s.expansions += 1;  // onExpandNode (a_star_epsilon.hpp:193 / a_star.hpp:87) — counts the goal pop too
    if (c.maxExp >= 0 && s.expansions > c.maxExp) return ST_CAP_EXP;

    if (isGoal) {
      res.cost = (int32_t)t;
      res.fmin = (int32_t)(EPS ? T::f(topE) : T::f(curE));
      res.n_states = (int32_t)t + 1;
      uint32_t nid = curId;
      for (int32_t k = (int32_t)t; k >= 0; --k) {  // follow cameFrom (a_star_epsilon.hpp:198-208)
        uint32_t pxy, par;
        nodeXyParent<T>(m, nid, pxy, par);
        outPath[k] = (uint16_t)pxy;  // all lanes, same address, same value
        nid = par;
      }
      DBG(c, 8, 77);
      return ST_OK;
    }

    const uint32_t t1 = t + 1;
    ensureRows<T>(m, s, c, t1, obstLocal, useLocal);
    PROF_SINCE(res, 4, profTop);  // loop top -> pops, minus the ordered walk (slot 0)
    // the five successor probes: bounds, then ONE bit of the (time, cell) bitmap = obstacle | vertex constraint |
    // already discovered; the words are requested before the pops below so that their latency hides behind them
    const uint32_t nx = x + (uint32_t)dx, ny = y + (uint32_t)dy;
    const bool inb = (lane < 5) && (nx < c.dimx) && (ny < c.dimy);
    const uint32_t ncell = inb ? ny * c.dimx + nx : 0;
    const uint32_t curCell = y * c.dimx + x;
    const uint32_t bitIdx = t1 * m.rowWords + (ncell >> 5);
    const uint32_t word = m.bits[bitIdx];

    {
      PROF_T0();
      if (EPS) {
        if constexpr (T::kEntryXy) curPos = rfl(curPosV);
        popFocalEraseOpen<T>(m, s.nFocal, s.nOpen, curPos);
      } else {
        heapPop<T, 0, true>(m, m.open, s.nOpen);
      }
      PROF_ADD(res, 1);
    }
    const bool ok = inb && !((word >> (ncell & 31)) & 1u);
    uint32_t mask = (uint32_t)(ballot64(ok) & 0x1Full);
    if (c.nEc) {  // transitionValid (ecbs.cpp:505-510): lane j holds edge-constraint key j = t << 19 | cell << 3 | action
      const uint32_t base = (t << 19) | (curCell << 3);
      const uint32_t d = ecReg - base;
      if (ballot64(d < 5u)) {  // rare: some constraint names a move out of this very state
        uint32_t blocked = 0;
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) blocked |= ballot64(d == k) ? (1u << k) : 0u;
        mask &= ~blocked;
      }
      if (c.nEc > 64) {
        uint32_t blocked = 0;
        for (uint32_t j = 64; j < c.nEc; ++j) {  // lists longer than a wave: the rest one by one
          const uint32_t dd = rfl(c.ec[j]) - base;
          if (dd < 5) blocked |= 1u << dd;
        }
        mask &= ~blocked;
      }
    }
    if (mask == 0) continue;
    PROF_MARK(profEnt);

    // ---- the successors' entries, one per lane 0..4 (order-independent part: heuristics, node records, discovered marks)
    const bool mine = (lane < 5) && ((mask >> lane) & 1u);
    const uint32_t nBase = s.nNodes;
    const uint32_t nid = nBase + (uint32_t)__builtin_popcount(mask & ((1u << lane) - 1u));
    const uint32_t h = (nx > c.gx ? nx - c.gx : c.gx - nx) + (ny > c.gy ? ny - c.gy : c.gy - ny);
    const uint32_t f = t1 + h;
    uint32_t fh = curFh;
    if (EPS && c.nAgentsPad) {
      // focalStateHeuristic (ecbs.cpp:282-295) + focalTransitionHeuristic (ecbs.cpp:298-312): lanes hold the other
      // agents' cells at t (a) and t+1 (b); an agent counts once if it stands on the successor's cell at t+1 and once
      // more if it swaps places with this agent
      // (the path table holds the other agents' cells as x | y << 8)
      const uint32_t nxyL = nx | (ny << 8);
      const uint64_t swap0 = ballot64(b0 == xy);
      const uint64_t swap1 = c.nAgentsPad > 64 ? ballot64(b1 == xy) : 0ull;
      for (uint32_t mm = mask; mm; mm &= mm - 1) {
        const uint32_t k = (uint32_t)__builtin_ctz(mm);
        const uint32_t cc = __builtin_amdgcn_readlane(nxyL, k);
        uint32_t cnt = (uint32_t)__popcll(ballot64(b0 == cc)) + (uint32_t)__popcll(ballot64(a0 == cc) & swap0);
        if (c.nAgentsPad > 64) {
          cnt += (uint32_t)__popcll(ballot64(b1 == cc)) + (uint32_t)__popcll(ballot64(a1 == cc) & swap1);
          for (uint32_t base = 128; base < c.nAgentsPad; base += 64) {
            uint32_t av = kEmptyCell, bv = kEmptyCell;
            if (base + lane < c.nAgentsPad) {
              av = rowA[base + lane];
              bv = rowB[base + lane];
            }
            cnt += (uint32_t)__popcll(ballot64(bv == cc)) + (uint32_t)__popcll(ballot64(av == cc && bv == xy));
          }
        }
        fh = lane == k ? curFh + cnt : fh;
      }
      if (T::AS != 3 && ballot64(mine && fh > T::kFhCap)) return ST_CAP_FOCAL;
    }
    E eMine = T::pack(fh, f, t1, nid);
    if constexpr (T::kEntryXy) eMine = T::withXy(eMine, nx | (ny << 8));
    const float bound = __fmul_rn((float)s.bestF, c.w);  // a_star_epsilon.hpp:240, binary32
    const uint32_t maskF = EPS ? (uint32_t)(ballot64(mine && (float)(int32_t)f <= bound) & 0x1Full) : 0u;
    if (mine) {
      if constexpr (T::AS == 3) {
        m.nodes[nid] = nx | (ny << 8) | (curId << 16);
      } else {
        u32x4 nn;
        nn.x = nx | (ny << 8) | (t1 << 16) | (lane << 27);
        nn.y = curId;
        nn.z = fh;
        nn.w = 0;
        ((typename Mem<T>::PNode4)m.nodes)[nid] = nn;
      }
      // mark (t1, cell) discovered: stands for stateToHeap / closedSet membership (a_star_epsilon.hpp:224-227); the
      // successors of one expansion are distinct cells, so marking them together changes nothing.  LDS: one ds_or.
      if constexpr (T::AS == 3)
        __hip_atomic_fetch_or(m.bits + bitIdx, 1u << (ncell & 31), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    if constexpr (T::AS != 3) {
      // HBM tier: a plain store of the merged word instead of a memory-side atomic per successor (successors that share
      // a bitmap word all store the same merged word)
      const uint32_t myBit = mine ? 1u << (ncell & 31) : 0u;
      uint32_t merged = word;
#pragma unroll
      for (uint32_t k = 0; k < 5; ++k) {
        const uint32_t oi = __builtin_amdgcn_readlane(bitIdx, k);
        const uint32_t ob = __builtin_amdgcn_readlane(myBit, k);
        merged |= oi == bitIdx ? ob : 0u;
      }
      if (mine) m.bits[bitIdx] = merged;
    }
    s.nNodes = nBase + (uint32_t)__builtin_popcount(mask);
    E e[5];
#pragma unroll
    for (uint32_t k = 0; k < 5; ++k) e[k] = T::fromLane(eMine, k);
    PROF_SINCE(res, 3, profEnt);  // successors' entries: heuristics, node records, discovered marks
    // ---- pushes: openSet.push for every successor, focalSet.push for those within the bound, in successor order
    {
      PROF_T0();
      PushChains<T> po, pf;
      po.load(m.open, s.nOpen, mask);
      if (EPS) pf.load(m.focal, s.nFocal, maskF);
      po.template resolve<0, true>(m, m.open, s.nOpen, mask, e);
      s.nOpen += (uint32_t)__builtin_popcount(mask);
      if (EPS) {
        pf.template resolve<1, false>(m, m.focal, s.nFocal, maskF, e);
        s.nFocal += (uint32_t)__builtin_popcount(maskF);
      }
      PROF_ADD(res, 2);
    }
  }
}

// ---- LDS layout ------------------------------------------------------------------------------------------------
// Dynamic LDS of a CBS / ECBS workgroup: the compact tier's window (ll_compact.h: open list, focal list, walk queue,
// (time, cell) bitmap, obstacle row), then the focal path table.  A search that has left the compact tier keeps the
// heaps' top entries in the same window (TierHyb).
// bg: the window of the A*-epsilon-only kernels (ll_compact.h BG: the (time, cell) bitmap lives in the arena slot)
__host__ __device__ inline uint32_t ldsBytes(uint32_t pathBytes, bool bg) { return ct::windowBytes(bg) + pathBytes; }

// A read of the device path store.  The slot was written by another workgroup (another CU, possibly another XCD) of the
// same resident launch before its completion was published; an agent-scope load goes past this CU's L1 to the coherent
// level, so no cache has to be invalidated for it (a per-job acquire fence would drop the whole L1 of the CU under the
// ten other searches that share it).
DEVI uint32_t storeLoad(const uint16_t* p) {
#ifdef MRP_LL_STORE_ACQUIRE_FENCE
  return *p;
#else
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

// Words the HOST wrote (job descriptors, constraint words, id lists, shipped tables) and words the host READS (result
// records, paths): system-scope accesses that go past this XCD's L2 in both directions.  The resident loop publishes and
// consumes jobs without cache-wide fences (residentLoop), so nothing else guarantees that a plain load of a recycled job
// slot does not find the previous occupant in L2, or that a plain store has left it when the done word is written.
DEVI uint32_t hostLoad32(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
DEVI void hostStore32(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// Which tiers a kernel carries:
//   kTiersAll   — compact (narrow) tier, then the arena tier: batch kernels, CBS / mixed sessions, A*-epsilon sessions
//                 without heavy workgroups;
//   kTiersFront — the compact (narrow) tier only: a search it cannot hold is handed to the heavy workgroups (runJob returns
//                 true, nothing of the job has been reported); no arena-tier code in the kernel;
//   kTiersHeavy — the compact tier in its WIDE geometry (3071 open entries, long horizons, ll_compact.h), then the arena tier.
enum : int { kTiersAll = 0, kTiersFront = 1, kTiersHeavy = 2 };

// Returns true when the job has to be handed to the heavy workgroups (kTiersFront only).
template <bool EPS, bool BG, int TIERS>
DEVI bool runJob(const LaunchParams& P, const DevJob& J, uint8_t* smem, uint8_t* arenaSlot, DevResult& res,
                 uint16_t* outPath) {
  typedef typename std::conditional<TIERS == kTiersHeavy, ct::Wide, ct::Narrow>::type Geo;
  constexpr bool kTableMayBeInLds = TIERS != kTiersHeavy;  // the wide window holds no path table
  const uint32_t lane = threadIdx.x;
  const bool heavyHint = (J.ctx_flags & kCtxHeavy) != 0;
  if (TIERS == kTiersFront) {
    // not a search of the narrow tier (the caller says so, or the job's shape does): nothing to set up here
    if (heavyHint || P.lds_nodes == 0 || J.dimx > 32u || J.dimy > 32u || J.n_agents_pad > 128u || J.n_ec > 64u) return true;
  }
  Ctx c;
  c.dimx = J.dimx; c.dimy = J.dimy; c.wpr = J.words_per_row;
  c.gx = J.gx; c.gy = J.gy; c.sx = J.sx; c.sy = J.sy;
  c.lastGoal = J.last_goal_constraint;
  c.w = J.w;
  c.nVc = J.n_vc; c.nEc = J.n_ec;
  c.obst = P.maps + J.map_word_off;
  c.nAgentsPad = J.n_agents_pad; c.tPad = J.t_pad;
  c.maxExp = J.max_expansions;
  c.debug = P.debug;

  // ---- bulk-copy the job's constraint words and path table out of host memory (one pass, coalesced) ----
  uint8_t* scratch = arenaSlot + P.arena_scratch_off;
  uint32_t* consLocal = (uint32_t*)(scratch + (size_t)P.out_stride * 2);
  uint8_t* pathsArena = (uint8_t*)(consLocal + kConsLocalWords);
  {
    const uint32_t* src = P.cons + J.vc_off;          // vertex words, then edge words (contiguous)
    const uint32_t nWords = c.nVc + c.nEc;
    if (nWords <= kConsLocalWords) {
      for (uint32_t i = lane; i < nWords; i += 64) consLocal[i] = hostLoad32(src + i);
      c.vc = consLocal;
      c.ec = consLocal + c.nVc;
    } else {  // (more than 2048 constraint words: the search reads them where the host put them, with plain loads)
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      c.vc = src;
      c.ec = P.cons + J.ec_off;
    }
    const uint32_t pathBytes = c.tPad * c.nAgentsPad * 2;  // multiple of 32
    const uint32_t* psrc = (const uint32_t*)(P.paths + J.path_off);
    uint8_t* ldsPaths = smem + Geo::windowBytes(BG);
    c.pathsLds = nullptr;
    if (pathBytes != 0 && (J.ctx_flags & kCtxById)) {
      // f2: the CT node's paths are named by their slots in the device-resident path store (each was written there by
      // the search that produced it); the time-major table [t][agent] is built here, on the device, instead of being
      // packed by the host and read over PCIe.  One coalesced read per agent (lane = time step).
      const bool inLds = kTableMayBeInLds && P.lds_nodes != 0 && pathBytes <= P.lds_paths_bytes;
      if (!inLds && pathBytes > P.arena_paths_bytes) {  // (the host packer refuses such a job; never write past the slot)
        res.status = ST_BAD;
        res.expanded = 0;
        res.nodes_created = 0;
        return false;
      }
      uint16_t* dst = inLds ? (uint16_t*)ldsPaths : (uint16_t*)pathsArena;
      {
        uint32_t* d32 = (uint32_t*)dst;
        for (uint32_t i = lane; i < pathBytes / 4; i += 64) d32[i] = 0xFFFFFFFFu;  // kEmptyCell everywhere
      }
      // The slots named here were written by OTHER workgroups of this same resident launch (possibly on another XCD,
      // whose L2 is not coherent with ours), each before its job's completion was published (processJob writes them in
      // front of residentLoop's system-scope release).  One agent-scope acquire drops whatever stale copies this CU's L1 /
      // this XCD's L2 may hold from an earlier use of a recycled slot; after it plain, cached, coalesced loads are
      // correct (MI355X_MICROARCH.md "Valid forms": poll -> ONE acquire -> s_waitcnt -> barrier -> plain loads).
#ifdef MRP_LL_STORE_ACQUIRE_FENCE  // A/B: one fence + plain loads instead of agent-scope loads
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __syncthreads();
      const uint32_t* ids = P.cons + J.path_off;
      const uint32_t nCtx = J.n_ctx;
      for (uint32_t a0 = 0; a0 < nCtx; a0 += 64) {
        // this chunk's ids and lengths, one agent per lane (one gather for all lengths)
        uint32_t idL = kNoStoreSlot, lenL = 0;
        if (a0 + lane < nCtx) idL = hostLoad32(ids + a0 + lane);
        if (idL < P.path_store_slots) lenL = storeLoad(P.path_store + (size_t)idL * P.path_store_stride);
        if (lenL > P.path_store_stride - 1) lenL = P.path_store_stride - 1;
        const uint32_t nHere = nCtx - a0 < 64 ? nCtx - a0 : 64;
        if (!inLds) {
          // Table in the arena (global memory): one lane per AGENT, so that a row of the table is one coalesced store
          // (a lane per time step would scatter 2-byte stores 2 * n_agents_pad bytes apart — measured on agents100: the
          // longest conflict-tree chain of a batch a third slower).  The reads gather one cell per slot and stay in L2
          // from row to row; eight rows are in flight at a time.
          const uint32_t lenA = lenL;
          const uint16_t* slotA = P.path_store + (size_t)(idL < P.path_store_slots ? idL : 0) * P.path_store_stride + 1;
          const bool hasA = idL < P.path_store_slots && lenA != 0;
          for (uint32_t t0 = 0; t0 < c.tPad; t0 += 8) {
            uint32_t v[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
              const uint32_t t = t0 + u;
              v[u] = kEmptyCell;
              if (hasA && t < c.tPad) v[u] = storeLoad(slotA + (t < lenA ? t : lenA - 1));
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u)
              if (hasA && t0 + u < c.tPad) dst[(t0 + u) * c.nAgentsPad + a0 + lane] = (uint16_t)v[u];
          }
          continue;
        }
        // table in LDS: a lane per time step (one coalesced read per agent); eight agents' loads are in flight before
        // the first store
        for (uint32_t t0 = 0; t0 < c.tPad; t0 += 64) {
          const uint32_t t = t0 + lane;
          for (uint32_t q0 = 0; q0 < nHere; q0 += 8) {
            uint32_t v[8];
            bool has[8];
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u) {
              uint32_t id = kNoStoreSlot, len = 0;
              if (q0 + u < nHere) {
                id = __builtin_amdgcn_readlane(idL, q0 + u);
                len = __builtin_amdgcn_readlane(lenL, q0 + u);
              }
              has[u] = id < P.path_store_slots && len != 0 && t < c.tPad;  // not: empty path / the searching agent itself
              v[u] = kEmptyCell;
              if (has[u]) v[u] = storeLoad(P.path_store + (size_t)id * P.path_store_stride + 1 + (t < len ? t : len - 1));
            }
#pragma unroll
            for (uint32_t u = 0; u < 8; ++u)
              if (has[u]) dst[t * c.nAgentsPad + a0 + q0 + u] = (uint16_t)v[u];
          }
        }
      }
      c.paths = dst;
      if (inLds) c.pathsLds = (__attribute__((address_space(3))) const uint16_t*)ldsPaths;
    } else if (pathBytes == 0) {
      c.paths = nullptr;
    } else if (kTableMayBeInLds && P.lds_nodes != 0 && pathBytes <= P.lds_paths_bytes) {
      uint32_t* dst = (uint32_t*)ldsPaths;
      for (uint32_t i = lane; i < pathBytes / 4; i += 64) dst[i] = hostLoad32(psrc + i);
      c.paths = (const uint16_t*)ldsPaths;
      c.pathsLds = (__attribute__((address_space(3))) const uint16_t*)ldsPaths;
    } else if (pathBytes <= P.arena_paths_bytes) {
      uint32_t* dst = (uint32_t*)pathsArena;
      for (uint32_t i = lane; i < pathBytes / 4; i += 64) dst[i] = hostLoad32(psrc + i);
      c.paths = (const uint16_t*)pathsArena;
    } else {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      c.paths = P.paths + J.path_off;
    }
  }
  __syncthreads();

  SState s;
  int rc = ST_BAD;
  res.tier = 0;

  // ---- compact tier (ll_compact.h): the whole search in LDS, a state = its 32-bit heap entry.  Maps up to 32 x 32 and
  // up to 128 agents in the focal context; a search that outgrows the tier (open list, time steps, focalH field) comes
  // back as C_OVERFLOW with nothing of it observable, and is run again from the start by the next tier.
  // the wide geometry: as many time steps as the arena slot's node area has room for (cameFrom table + bitmap), in
  // chunks of 64, up to the job's horizon
  uint32_t geoRows = Geo::kRows;
  if (TIERS == kTiersHeavy) {
    const uint64_t room = (uint64_t)P.arena_nodes * 16u / (1024u + ct::kRowBytes);
    geoRows = (uint32_t)(room < P.arena_rows ? room : P.arena_rows) & ~63u;
  }
  const bool compactOk = !(TIERS == kTiersAll && heavyHint) && P.lds_nodes != 0 && c.dimx <= 32u && c.dimy <= 32u &&
                         c.nAgentsPad <= 128u && c.nEc <= 64u && geoRows >= 64u &&
                         (uint64_t)P.arena_nodes * 16u >= Geo::parentBytes(geoRows) + (BG ? Geo::bitsBytes(geoRows) : 0u);
  bool done = false;
  if (compactOk) {
    // the job goes into its block of the LDS window (every lane stores the same words), the result comes back from there:
    // ct::compactSearch is a real function with its own register allocation
    ct::CJob cj;
    cj.dimx = c.dimx; cj.dimy = c.dimy; cj.sx = c.sx; cj.sy = c.sy; cj.gx = c.gx; cj.gy = c.gy;
    cj.lastGoal = c.lastGoal;
    cj.w = c.w;
    cj.nVc = c.nVc; cj.nEc = c.nEc;
    cj.obstWords = c.wpr;
    cj.nAgentsPad = EPS ? c.nAgentsPad : 0u; cj.tPad = c.tPad;
    cj.maxExp = c.maxExp < 0 ? 0xFFFFFFFFu : (c.maxExp > 0xFFFFFFFEll ? 0xFFFFFFFEu : (uint32_t)c.maxExp);
    cj.rows = geoRows;
    if (TIERS == kTiersHeavy) {  // the wide geometry at its full size
      cj.openCap = Geo::kCap;
      cj.maxT = Geo::kMaxT < geoRows - 2u ? Geo::kMaxT : geoRows - 2u;
    } else {
      // mrp_ll_configure_tiers: lds_nodes / 2 = open-list entries, lds_rows = time steps a search may use inside the tier
      cj.openCap = P.lds_nodes / 2u < Geo::kCap ? P.lds_nodes / 2u : Geo::kCap;
      cj.maxT = P.lds_rows >= 3u && P.lds_rows - 2u < Geo::kMaxT ? P.lds_rows - 2u : Geo::kMaxT;
    }
    cj.taNoGoal = 0;
    cj.vc = (uint64_t)c.vc; cj.ec = (uint64_t)c.ec;
    cj.obst = (uint64_t)c.obst;
    cj.pathsG = (uint64_t)c.paths;
    cj.parentTab = (uint64_t)arenaSlot;  // the arena's node area: unused while the search is in this tier
    cj.outPath = (uint64_t)outPath;
    cj.bitsG = (uint64_t)(arenaSlot + Geo::parentBytes(geoRows));  // (BG) ... and its (time, cell) bitmap behind it
    {
      auto w32 = (__attribute__((address_space(3))) uint32_t*)((wv::Lds)smem + ct::oJob);
      const uint32_t* src = (const uint32_t*)&cj;
#pragma unroll
      for (uint32_t q = 0; q < sizeof(ct::CJob) / 4; ++q) w32[q] = src[q];
    }
    const bool tableInLds = !EPS || c.nAgentsPad == 0u || c.pathsLds != nullptr;
#ifndef MRP_LL_TRACE  // (the trace build uses prof[] for its phase counters)
    const uint64_t tl0 = __builtin_amdgcn_s_memrealtime();
#endif
    int32_t crc;
    if constexpr (TIERS == kTiersHeavy)
      crc = ct::compactSearch<EPS, false, BG, ct::Wide>((wv::Lds)smem);
    else
      crc = tableInLds ? ct::compactSearch<EPS, true, BG>((wv::Lds)smem) : ct::compactSearch<EPS, false, BG>((wv::Lds)smem);
    ct::CRes cr;
    {
      auto r32 = (__attribute__((address_space(3))) const uint32_t*)((wv::Lds)smem + ct::oRes);
      cr.status = (int32_t)rfl(r32[0]); cr.cost = (int32_t)rfl(r32[1]); cr.fmin = (int32_t)rfl(r32[2]);
      cr.nStates = (int32_t)rfl(r32[3]); cr.expanded = rfl(r32[4]); cr.nodes = rfl(r32[5]);
    }
#ifndef MRP_LL_TRACE
    // 100 MHz ticks / expansions in the compact tier (of a search that was handed over: until then); the wide geometry
    // reports into the arena tier's pair — "the searches that outgrew the narrow tier"
    res.prof[TIERS == kTiersHeavy ? 2 : 0] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tl0);
    res.prof[TIERS == kTiersHeavy ? 3 : 1] = cr.expanded;
#endif
#ifdef MRP_CT_PROF  // diagnostic build: the compact tier's own phase counters instead of the tier statistics
    {
      auto r32 = (__attribute__((address_space(3))) const uint32_t*)((wv::Lds)smem + ct::oRes + 32u);
      for (uint32_t q = 0; q < 8; ++q) res.prof[q] = rfl(r32[q]);
    }
#endif
    if (crc != ct::C_OVERFLOW) {
      rc = crc;  // C_OK / C_NO_SOLUTION / C_CAP_EXP == ST_OK / ST_NO_SOLUTION / ST_CAP_EXP
      res.cost = cr.cost;
      res.fmin = cr.fmin;
      res.n_states = cr.nStates;
      s.expansions = cr.expanded;
      s.nNodes = cr.nodes;
      done = true;
      if (TIERS == kTiersHeavy) res.tier = 2;
    } else {
#ifndef MRP_LL_TRACE
      res.prof[6] = cr.expanded;  // expansions thrown away with the attempt
      res.prof[7] = 1;
#endif
    }
  }
  if constexpr (TIERS == kTiersFront) {
    if (!done) return true;  // the heavy workgroups run it from the start
  } else {
    if (!done) {
      // HBM tier view of this workgroup's arena slot
      Mem<TierHbm> g;
      {
        uint8_t* p = arenaSlot;
        g.nodes = (Mem<TierHbm>::PN32)p;             p += (size_t)P.arena_nodes * 16;
        g.pos = nullptr;
        g.open = (Mem<TierHbm>::PE)(p + 8);          p += (size_t)P.arena_nodes * 8 + 16;
        g.focal = (Mem<TierHbm>::PE)(p + 8);         p += (size_t)P.arena_nodes * 8 + 16;
        g.aux = (Mem<TierHbm>::PE)(p + 8);           p += (size_t)P.arena_nodes * 8 + 16;
        g.bits = (Mem<TierHbm>::P32)p;
        g.capNodes = P.arena_nodes; g.capHeap = P.arena_nodes; g.capRows = P.arena_rows; g.rowWords = P.arena_row_words;
      }
      // ... and the view the arena tier actually runs on: the same arrays, the heaps' first nTop entries in this
      // workgroup's LDS (the compact tier's area, free once a search has left it)
      Mem<TierHyb> gh;
      {
        const uint32_t area = Geo::windowBytes(BG) - ct::oOpen;  // (the window's control blocks in front of it stay as they are)
        // the open list gets half of the area, the focal list five sixteenths, the walk queue the rest (MRP_LL_TOPS_EQUAL:
        // thirds, as before the A*-epsilon kernels' window shrank)
#ifdef MRP_LL_TOPS_EQUAL
        const uint32_t perO = (area / 3u) & ~15u, perF = perO, perA = perO;
#else
        const uint32_t perO = (area / 2u) & ~15u, perF = (area * 5u / 16u) & ~15u, perA = (area - perO - perF) & ~15u;
#endif
        auto tops = [&](uint32_t per) {
          uint32_t n = per >= 32u ? ((per - 8u) / 8u) : 0u;
          if (n > 4095u) n = 4095u;
          n = n ? ((n - 1u) | 1u) : 0u;  // odd (or 0: no LDS tier configured)
          return P.lds_nodes == 0 ? 0u : n;
        };
        auto l8 = (__attribute__((address_space(3))) uint8_t*)smem + ct::oOpen;
        gh.nodes = g.nodes;
        gh.pos = nullptr;
        gh.gOf = nullptr;
        gh.open = HybPtr{(__attribute__((address_space(3))) uint64_t*)(l8 + 8), (uint64_t*)g.open, tops(perO)};
        gh.focal = HybPtr{(__attribute__((address_space(3))) uint64_t*)(l8 + perO + 8), (uint64_t*)g.focal, tops(perF)};
        gh.aux = HybPtr{(__attribute__((address_space(3))) uint64_t*)(l8 + perO + perF + 8), (uint64_t*)g.aux, tops(perA)};
        gh.bits = g.bits;
        gh.capNodes = g.capNodes; gh.capHeap = g.capHeap; gh.capRows = g.capRows; gh.rowWords = g.rowWords;
      }
      const bool xyEntries = P.arena_nodes <= 65536u;  // TierHybXy: 16-bit node ids leave room for the cell in the entry
      Mem<TierHybXy> ghx;
      ghx.nodes = gh.nodes; ghx.pos = nullptr; ghx.gOf = nullptr;
      ghx.open = gh.open; ghx.focal = gh.focal; ghx.aux = gh.aux; ghx.bits = gh.bits;
      ghx.capNodes = gh.capNodes; ghx.capHeap = gh.capHeap; ghx.capRows = gh.capRows; ghx.rowWords = gh.rowWords;
      res.tier = 1;
      __syncthreads();  // previous job's / the compact attempt's LDS accesses are done
      if (xyEntries)
        initSearch<TierHybXy, EPS>(ghx, s, c);
      else
        initSearch<TierHyb, EPS>(gh, s, c);
      __syncthreads();
#ifndef MRP_LL_TRACE
      const uint64_t th0 = __builtin_amdgcn_s_memrealtime();
#endif
      if (xyEntries)
        rc = runSearch<TierHybXy, EPS>(ghx, s, c, (Mem<TierHybXy>::P32)c.obst, false, res, outPath);
      else
        rc = runSearch<TierHyb, EPS>(gh, s, c, (Mem<TierHyb>::P32)c.obst, false, res, outPath);
#ifndef MRP_LL_TRACE
      if (TIERS != kTiersHeavy) {
        res.prof[2] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - th0);
        res.prof[3] = (uint32_t)s.expansions;
      }
#endif
    }
  }
  if (rc == RUN_MIGRATE_NODES) rc = ST_CAP_NODES;
  if (rc == RUN_MIGRATE_ROWS) rc = ST_CAP_HORIZON;
  res.status = rc;
  res.expanded = s.expansions;
  res.nodes_created = s.nNodes;
  return false;
}


// The root chain of one ECBS conflict tree (ecbs.hpp:118-136; ll_device.h kCtxChain): agent a is planned against the
// paths of the agents in front of it, the focal table [kChainRows][n_agents_pad] stays in the window between the
// searches and gains one column per path.  Everything runs in the compact tier; a search that outgrows it ends the chain
// in front of it.  Written for the A*-epsilon-only kernels (BG window).
DEVI void runChain(const LaunchParams& P, const DevJob& J, uint8_t* smem, uint8_t* arenaSlot, DevResult& res,
                   uint16_t* outPath, uint16_t* hostOut) {
  constexpr bool BG = true;
  const uint32_t lane = threadIdx.x;
  const uint32_t n = J.n_ctx, first = J.t_pad, npad = J.n_agents_pad;
  const uint32_t end = J.reserved > first && J.reserved < n ? J.reserved : n;  // one past the last agent of this job
  res.tier = 0;
  res.n_states = 0;
  res.expanded = 0;
  if (P.lds_nodes == 0 || J.dimx > 32u || J.dimy > 32u || n > kChainMaxAgents || first >= n || npad < n || npad > 128u ||
      (npad & 1u) || kChainRows * npad * 2u > P.lds_paths_bytes ||
      (uint64_t)P.arena_nodes * 16u < ct::kParentBytes + ct::kBitsBytes ||
      (uint64_t)n * kChainEntryWords * 2u + (uint64_t)n * 64u > (uint64_t)P.out_host_stride) {
    res.status = ST_BAD;  // (the host packer refuses such a job)
    return;
  }
  const uint32_t* who = P.cons + J.vc_off;   // starts / goals
  const uint32_t* ids = who + n;             // path-store slots
  uint16_t* table = (uint16_t*)(smem + ldsBytes(0, BG));
  for (uint32_t i = lane; i < kChainRows * npad / 2u; i += 64) ((uint32_t*)table)[i] = 0xFFFFFFFFu;  // nobody anywhere
  __syncthreads();
  for (uint32_t a = 0; a < first; ++a) {  // the paths that exist already: lane = time step
    const uint32_t id = rfl(hostLoad32(ids + a));
    if (id >= P.path_store_slots) continue;
    const uint16_t* slot = P.path_store + (size_t)id * P.path_store_stride;
    uint32_t len = rfl(storeLoad(slot));
    if (len > P.path_store_stride - 1) len = P.path_store_stride - 1;
    if (len == 0) continue;
    table[lane * npad + a] = (uint16_t)storeLoad(slot + 1 + (lane < len ? lane : len - 1));
  }
  __syncthreads();
  int64_t budget = J.max_expansions;  // < 0: unlimited
  uint32_t* hostW = (uint32_t*)hostOut;
  uint32_t pathOff = n * kChainEntryWords;  // words
  uint32_t done = 0, maxLen = 0;
  bool allOk = true;
  int64_t total = 0;
  for (uint32_t a = first; a < end; ++a) {
    const uint32_t sg = rfl(hostLoad32(who + a));
    ct::CJob cj;
    cj.dimx = J.dimx; cj.dimy = J.dimy;
    cj.sx = sg & 0xFFu; cj.sy = (sg >> 8) & 0xFFu; cj.gx = (sg >> 16) & 0xFFu; cj.gy = sg >> 24;
    cj.lastGoal = -1;
    cj.w = J.w;
    cj.nVc = 0; cj.nEc = 0;
    cj.obstWords = J.words_per_row;
    cj.nAgentsPad = npad; cj.tPad = kChainRows;
    cj.maxExp = budget < 0 ? 0xFFFFFFFFu : (budget > 0xFFFFFFFEll ? 0xFFFFFFFEu : (uint32_t)budget);
    cj.openCap = P.lds_nodes / 2u < ct::kCap ? P.lds_nodes / 2u : ct::kCap;
    cj.maxT = P.lds_rows >= 3u && P.lds_rows - 2u < ct::kMaxT ? P.lds_rows - 2u : ct::kMaxT;
    cj.taNoGoal = 0;
    cj.rows = 0;
    cj.vc = 0; cj.ec = 0;
    cj.obst = (uint64_t)(P.maps + J.map_word_off);
    cj.pathsG = 0;
    cj.parentTab = (uint64_t)arenaSlot;
    cj.outPath = (uint64_t)outPath;
    cj.bitsG = (uint64_t)(arenaSlot + ct::kParentBytes);
    __syncthreads();
    {
      auto w32 = (__attribute__((address_space(3))) uint32_t*)((wv::Lds)smem + ct::oJob);
      const uint32_t* src = (const uint32_t*)&cj;
#pragma unroll
      for (uint32_t q = 0; q < sizeof(ct::CJob) / 4; ++q) w32[q] = src[q];
    }
#ifndef MRP_LL_TRACE
    const uint64_t tl0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int32_t crc = ct::compactSearch<true, true, BG>((wv::Lds)smem);
    auto r32 = (__attribute__((address_space(3))) const uint32_t*)((wv::Lds)smem + ct::oRes);
    const int32_t cost = (int32_t)rfl(r32[1]), fmin = (int32_t)rfl(r32[2]), nStates = (int32_t)rfl(r32[3]);
    const uint32_t expanded = rfl(r32[4]);
#ifndef MRP_LL_TRACE
    res.prof[0] += (uint32_t)(__builtin_amdgcn_s_memrealtime() - tl0);
    res.prof[1] += expanded;
#endif
    if (crc == ct::C_OVERFLOW) {
#ifndef MRP_LL_TRACE
      res.prof[6] += expanded;
      res.prof[7] += 1;
#endif
      allOk = false;
      break;  // not a search of this tier: the caller runs it as an ordinary job
    }
    {  // the agent's entry
      uint32_t v = 0;
      v = lane == 0 ? (uint32_t)crc : lane == 1 ? (uint32_t)cost : lane == 2 ? (uint32_t)fmin
          : lane == 3 ? (crc == ct::C_OK ? (uint32_t)nStates : 0u) : lane == 4 ? expanded : lane == 5 ? pathOff : 0u;
      if (lane < kChainEntryWords) hostStore32(hostW + (size_t)done * kChainEntryWords + lane, v);
    }
    done += 1;
    total += expanded;
    if (crc != ct::C_OK) {  // no path / expansion budget: the conflict tree ends with this answer
      allOk = false;
      break;
    }
    const uint32_t len = (uint32_t)nStates;
    maxLen = len > maxLen ? len : maxLen;
    {  // the path: to the host, to its path-store slot, into the table
      const uint32_t words = (len + 1u) / 2u;
      const uint32_t* src = (const uint32_t*)outPath;
      for (uint32_t i = lane; i < words; i += 64) hostStore32(hostW + pathOff + i, src[i]);
      pathOff += words;
      const uint32_t sid = rfl(hostLoad32(ids + a));
      if (sid < P.path_store_slots && len < P.path_store_stride) {
        uint16_t* slot = P.path_store + (size_t)sid * P.path_store_stride;
        for (uint32_t i = lane; i < len; i += 64) __hip_atomic_store(slot + 1 + i, outPath[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(slot, (uint16_t)len, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      table[lane * npad + a] = outPath[lane < len ? lane : len - 1];
    }
    if (budget >= 0) budget = budget > (int64_t)expanded ? budget - (int64_t)expanded : 0;  // Instance::remainingLL()
  }
  __syncthreads();
  res.status = ST_OK;
  res.n_states = (int32_t)done;
  res.expanded = total;
  res.cost = -1;
  res.fmin = -1;
  // ---- the root node's conflicts (SURVEY.md §8 f1 on the drivers' path): when the chain has planned EVERY agent of the
  // instance, the table in the window is the root's whole solution, and this workgroup says at once whether the conflict
  // tree has anything to do — getFirstConflict (ecbs.cpp:401-452) and focalHeuristic (ecbs.cpp:315-350) over t = 0 ..
  // max_t - 1 (the final time step is never checked) and all pairs i < j:
  //   vertex conflict at t: state_i(t) == state_j(t);  edge conflict: state_i(t) == state_j(t+1) && state_i(t+1) == state_j(t)
  // Lane = time step (every path of this tier has at most 63 states); the first conflict is the smallest
  // (t, vertex before edge, i, j).  Seven instances in ten of the ten-agent workload end here: HL 1, no conflict.
  if (first == 0 && end == n && done == n && allOk && maxLen >= 1u && maxLen <= kChainRows) {
    const uint32_t T = maxLen - 1u;  // <= 62
    const bool inT = lane < T;
    const uint32_t rowC = lane * npad, rowN = (lane + 1u < kChainRows ? lane + 1u : kChainRows - 1u) * npad;
    uint32_t cnt = 0, bestV = 0xFFFFu, bestE = 0xFFFFu;  // this lane's (time step's) first vertex / edge pair: i << 8 | j
    for (uint32_t i = 0; i + 1u < n; ++i) {
      const uint32_t ci = table[rowC + i], ni = table[rowN + i];
      for (uint32_t j = i + 1u; j < n; ++j) {
        const uint32_t cj = table[rowC + j], nj = table[rowN + j];
        const bool v = inT && ci == cj, e = inT && ci == nj && ni == cj;
        cnt += (v ? 1u : 0u) + (e ? 1u : 0u);
        if (v && bestV == 0xFFFFu) bestV = (i << 8) | j;
        if (e && bestE == 0xFFFFu) bestE = (i << 8) | j;
      }
    }
    uint32_t key = bestV != 0xFFFFu ? (lane << 24) | bestV : bestE != 0xFFFFu ? (lane << 24) | (1u << 16) | bestE : 0x7FFFFFFFu;
#pragma unroll
    for (uint32_t off = 32; off >= 1; off >>= 1) {
      cnt += (uint32_t)__shfl_xor((int)cnt, (int)off, 64);
      const uint32_t other = (uint32_t)__shfl_xor((int)key, (int)off, 64);
      key = other < key ? other : key;
    }
    res.cost = (int32_t)rfl(cnt);
    res.fmin = rfl(key) == 0x7FFFFFFFu ? -1 : (int32_t)rfl(key);
  }
}

// MRP_LL_ASTAR_TA in the arena tier: AStar::search (a_star.hpp:63-161) over the Environment of example/cbs_ta.cpp:283-372
// for the searches the compact tier cannot hold (more than 1023 open nodes, t > 61, f > 254, more than 64 + 64
// constraints, maps beyond 32 x 32).  Same rules as ct::compactSearchTA — optional task, shortest-path heuristic from the
// uploaded table, a Wait at the goal is free, so a state can be reached again with a smaller g and
// `openSet.increase(handle)` (a_star.hpp:139-145) is live — on the arena's node records:
//   node   {x | y << 8 | t << 16 | action << 27, parent, g, position of its entry in the open array}
//   entry  TierHbm: key = (f asc, g desc), low word = node id
//   status one word per (t, cell) in the (unused) focal + walk-queue areas of the slot: 0 unseen, node + 1 in the open
//          list, bit 31 closed (stateToHeap + closedSet, a_star.hpp:116-117)
//   bits   (time, cell) bitmap: obstacles | vertex constraints (stateValid, cbs_ta.cpp:483-489), rows made on demand
// Time steps: as many as the status table has room for (and the job's horizon); beyond: MRP_LL_CAP_HORIZON.
DEVI void runTaArena(const LaunchParams& P, const DevJob& J, uint8_t* arenaSlot, DevResult& res, uint16_t* outPath,
                     const uint32_t* vc, const uint32_t* ec, const uint16_t* heur, uint32_t heurStride) {
  typedef TierHbm T;
  const uint32_t lane = threadIdx.x;
  Mem<T> g;
  {
    uint8_t* p = arenaSlot;
    g.nodes = (Mem<T>::PN32)p;             p += (size_t)P.arena_nodes * 16;
    g.pos = nullptr;
    g.gOf = nullptr;
    g.open = (Mem<T>::PE)(p + 8);          p += (size_t)P.arena_nodes * 8 + 16;
    g.focal = (Mem<T>::PE)(p + 8);         // (the status table lives from here on)
    g.aux = g.focal;
    g.bits = (Mem<T>::P32)(arenaSlot + (size_t)P.arena_nodes * 16 + 3 * ((size_t)P.arena_nodes * 8 + 16));
    g.capNodes = P.arena_nodes; g.capHeap = P.arena_nodes; g.capRows = P.arena_rows; g.rowWords = P.arena_row_words;
  }
  uint32_t* status = (uint32_t*)(arenaSlot + (size_t)P.arena_nodes * 16 + ((size_t)P.arena_nodes * 8 + 16));
  const uint32_t dimx = J.dimx, dimy = J.dimy, cells = dimx * dimy;
  const uint64_t statusWords = ((uint64_t)P.arena_nodes * 8 + 16) * 2 / 4;
  uint32_t rows = (uint32_t)(statusWords / cells < P.arena_rows ? statusWords / cells : P.arena_rows);
  const bool noGoal = (J.ctx_flags & kTaNoGoal) != 0;
  Ctx c;
  c.dimx = dimx; c.dimy = dimy; c.wpr = J.words_per_row;
  c.gx = J.gx; c.gy = J.gy; c.sx = J.sx; c.sy = J.sy;
  c.lastGoal = J.last_goal_constraint;
  c.w = 1.0f;
  c.nVc = J.n_vc; c.nEc = J.n_ec;
  c.vc = vc; c.ec = ec;
  c.obst = P.maps + J.map_word_off;
  c.paths = nullptr; c.pathsLds = nullptr;
  c.nAgentsPad = 0; c.tPad = 0;
  c.maxExp = J.max_expansions;
  c.debug = P.debug;
  res.tier = 1;
  res.status = ST_NO_SOLUTION;
  if (rows < 2u) {
    res.status = ST_CAP_HORIZON;
    return;
  }
  SState s;
  s.nNodes = 1; s.nOpen = 1; s.nFocal = 0; s.rowsReady = 0; s.bestF = 0; s.expansions = 0;
  {  // the table: nothing seen (rows are zeroed as the bitmap's rows are made, below: `statusReady`)
    const uint32_t sc = J.sy * dimx + J.sx;
    const uint32_t h0 = noGoal ? 0u : heur[J.sy * heurStride + J.sx];
    if (h0 > kFMax - 2u) {  // the task cannot be reached from here (the reference's table says INT_MAX), or not within f's field
      res.status = ST_CAP_HORIZON;
      return;
    }
    u32x4 n0;
    n0.x = J.sx | (J.sy << 8) | (0u << 16) | (7u << 27);
    n0.y = kNoParent;
    n0.z = 0;
    n0.w = 0;
    ((Mem<T>::PNode4)g.nodes)[0] = n0;
    g.open[0] = T::pack(0, h0, 0, 0);
    for (uint32_t i = lane; i < cells; i += 64) status[i] = 0;
    __syncthreads();
    status[sc] = 1u;  // node 0, in the open list
  }
  uint32_t statusReady = 1;  // rows of the status table that have been zeroed
  const uint32_t ecReg = lane < c.nEc ? c.ec[lane] : 0xFFFFFFFFu;
  const int32_t dx = (lane == 2) - (lane == 1);
  const int32_t dy = (lane == 3) - (lane == 4);
  for (;;) {
    if (s.nOpen == 0) {
      res.status = ST_NO_SOLUTION;
      break;
    }
    const T::E curE = ldU<T>(g.open, 0);
    const uint32_t curId = T::id(curE), gcur = T::g(curE), fcur = T::f(curE);
    const u32x4 nd = ((Mem<T>::PNode4)g.nodes)[curId];
    const uint32_t xyt = rfl(nd.x);
    const uint32_t x = xyt & 0xFFu, y = (xyt >> 8) & 0xFFu, t = (xyt >> 16) & 0x7FFu;
    const bool atGoal = noGoal || (x == c.gx && y == c.gy);
    s.expansions += 1;  // onExpandNode (a_star.hpp:87)
    if (c.maxExp >= 0 && s.expansions > c.maxExp) {
      res.status = ST_CAP_EXP;
      break;
    }
    if (atGoal && (int32_t)t > c.lastGoal) {  // isSolution (cbs_ta.cpp:313-319) -> a_star.hpp:89-106
      if (t + 1u > P.out_stride) {
        res.status = ST_CAP_HORIZON;
        break;
      }
      uint32_t nid = curId;
      for (int32_t k = (int32_t)t; k >= 0; --k) {
        const u32x4 pn = ((Mem<T>::PNode4)g.nodes)[nid];
        outPath[k] = (uint16_t)(rfl(pn.x) & 0xFFFFu);
        nid = rfl(pn.y);
      }
      res.status = ST_OK;
      res.cost = (int32_t)gcur;
      res.fmin = (int32_t)fcur;
      res.n_states = (int32_t)t + 1;
      break;
    }
    const uint32_t t1 = t + 1u;
    if (t1 >= rows || t1 >= g.capRows) {
      res.status = ST_CAP_HORIZON;
      break;
    }
    if (s.nNodes + 5u > g.capNodes || s.nOpen + 5u > g.capHeap) {
      res.status = ST_CAP_NODES;
      break;
    }
    heapPop<T, 0, true>(g, g.open, s.nOpen);  // openSet.pop() (a_star.hpp:109)
    status[t * cells + y * dimx + x] = 0x80000000u;  // closedSet.insert (a_star.hpp:110)
    ensureRows<T>(g, s, c, t1, (Mem<T>::P32)c.obst, false);
    while (statusReady <= t1) {  // the status rows of the next time steps: nothing seen
      for (uint32_t i = lane; i < cells; i += 64) status[statusReady * cells + i] = 0;
      statusReady += 1;
    }
    __syncthreads();
    // getNeighbors (cbs_ta.cpp:321-367): Wait, Left, Right, Up, Down on lanes 0..4 — bounds, obstacle | vertex constraint
    // (one bit of the bitmap), edge constraints by key
    const uint32_t nx = x + (uint32_t)dx, ny = y + (uint32_t)dy;
    const bool inb = (lane < 5) && (nx < dimx) && (ny < dimy);
    const uint32_t ncell = inb ? ny * dimx + nx : 0;
    const uint32_t word = g.bits[t1 * g.rowWords + (ncell >> 5)];
    const uint32_t hN = (noGoal || !inb) ? 0u : heur[ny * heurStride + nx];
    const uint32_t stN = inb ? status[t1 * cells + ncell] : 0u;
    uint32_t mask = (uint32_t)(ballot64(inb && !((word >> (ncell & 31)) & 1u)) & 0x1Full);
    if (c.nEc) {  // transitionValid (cbs_ta.cpp:491-496)
      const uint32_t base = (t << 19) | ((y * dimx + x) << 3);
      uint32_t blocked = 0;
      for (uint32_t j0 = 0; j0 < c.nEc; j0 += 64) {
        const uint32_t d = (j0 == 0 ? ecReg : (j0 + lane < c.nEc ? c.ec[j0 + lane] : 0xFFFFFFFFu)) - base;
#pragma unroll
        for (uint32_t k = 0; k < 5; ++k) blocked |= ballot64(d == k) ? (1u << k) : 0u;
      }
      mask &= ~blocked;
    }
    bool fail = false;
    for (uint32_t mm = mask; mm && !fail; mm &= mm - 1) {  // the new / rediscovered / closed cases of a_star.hpp:116-153, in order
      const uint32_t k = (uint32_t)__builtin_ctz(mm);
      const uint32_t st = __builtin_amdgcn_readlane(stN, k);
      if (st & 0x80000000u) continue;  // closed
      const uint32_t nc = __builtin_amdgcn_readlane(ncell, k), h = __builtin_amdgcn_readlane(hN, k);
      const uint32_t nxk = __builtin_amdgcn_readlane(nx, k), nyk = __builtin_amdgcn_readlane(ny, k);
      const uint32_t g2 = gcur + ((k == 0 && atGoal) ? 0u : 1u);  // tentative_gScore (a_star.hpp:118)
      if (st == 0) {  // not in the open list, not closed: a new node (a_star.hpp:120-129)
        if (h > kFMax || g2 + h > kFMax - 2u || g2 > kGMask) {
          res.status = ST_CAP_HORIZON;
          fail = true;
          break;
        }
        const uint32_t nid = s.nNodes++;
        u32x4 nn;
        nn.x = nxk | (nyk << 8) | (t1 << 16) | (k << 27);
        nn.y = curId;
        nn.z = g2;
        nn.w = 0;
        ((Mem<T>::PNode4)g.nodes)[nid] = nn;
        status[t1 * cells + nc] = nid + 1u;
        siftUp<T, 0, true>(g, g.open, s.nOpen, T::pack(0, g2 + h, g2, nid));
        s.nOpen += 1;
      } else {        // still in the open list (a_star.hpp:130-146)
        const uint32_t nid = st - 1u;
        const u32x4 on = ((Mem<T>::PNode4)g.nodes)[nid];
        const uint32_t gOld = rfl(on.z), posOld = rfl(on.w);
        if (g2 >= gOld) continue;  // not an improvement (a_star.hpp:135-137)
        const uint32_t fOld = T::f(ldU<T>(g.open, posOld));
        const uint32_t fNew = fOld - (gOld - g2);  // fScore -= delta (a_star.hpp:141-142)
        g.nodes[nid * 4 + 0] = (rfl(on.x) & 0x07FFFFFFu) | (k << 27);  // cameFrom is replaced (a_star.hpp:150-152)
        g.nodes[nid * 4 + 1] = curId;
        g.nodes[nid * 4 + 2] = g2;
        siftUp<T, 0, true>(g, g.open, posOld, T::pack(0, fNew, g2, nid));  // increase(handle)
      }
    }
    if (fail) break;
  }
  res.expanded = s.expansions;
  res.nodes_created = s.nNodes;
}

// MRP_LL_ASTAR_TA: the low level of the task-assignment callers: the compact tier (ll_compact.h
// compactSearchTA) when the job fits it — a map up to 32 x 32, at most 64 vertex and 64 edge constraints — and the arena
// tier above when it does not, or when the search outgrows the compact tier on the way (its capacity statuses are then
// not an answer).  The goal's shortest-path table sits in the maps buffer (mrp_ll_upload_heuristic): [32][32] halfwords
// for maps up to 32 x 32, [dimy][dimx] beyond.
DEVI void runJobTA(const LaunchParams& P, const DevJob& J, uint8_t* smem, uint8_t* arenaSlot, DevResult& res, uint16_t* outPath) {
  res.tier = 0;
  const uint32_t lane = threadIdx.x;
  // the constraint words leave host memory in one pass (the arena's copy area holds 2048 of them; longer lists are read in place)
  uint32_t* consLocal = (uint32_t*)(arenaSlot + P.arena_scratch_off + (size_t)P.out_stride * 2);
  const uint32_t* vc = consLocal;
  const uint32_t* ec = consLocal + J.n_vc;
  if (J.n_vc + J.n_ec <= kConsLocalWords) {
    for (uint32_t i = lane; i < J.n_vc; i += 64) consLocal[i] = hostLoad32(P.cons + J.vc_off + i);
    for (uint32_t i = lane; i < J.n_ec; i += 64) consLocal[J.n_vc + i] = hostLoad32(P.cons + J.ec_off + i);
  } else {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    vc = P.cons + J.vc_off;
    ec = P.cons + J.ec_off;
  }
  __syncthreads();
  const bool small = J.dimx <= 32u && J.dimy <= 32u;
  const uint16_t* heur = (const uint16_t*)(P.maps + J.path_off);
  const bool compactOk = P.lds_nodes != 0 && P.lds_paths_bytes >= 2048u && small && J.n_vc <= 64u && J.n_ec <= 64u &&
                         (uint64_t)P.arena_nodes * 16u >= ct::kParentBytes;
  if (compactOk) {
    ct::CJob cj;
    cj.dimx = J.dimx; cj.dimy = J.dimy; cj.sx = J.sx; cj.sy = J.sy; cj.gx = J.gx; cj.gy = J.gy;
    cj.lastGoal = J.last_goal_constraint;
    cj.w = 1.0f;
    cj.nVc = J.n_vc; cj.nEc = J.n_ec;
    cj.obstWords = J.words_per_row;
    cj.nAgentsPad = 0; cj.tPad = 0;
    cj.maxExp = J.max_expansions < 0 ? 0xFFFFFFFFu : (J.max_expansions > 0xFFFFFFFEll ? 0xFFFFFFFEu : (uint32_t)J.max_expansions);
    cj.openCap = P.lds_nodes / 2u < ct::kCap ? P.lds_nodes / 2u : ct::kCap;
    cj.maxT = P.lds_rows >= 3u && P.lds_rows - 2u < ct::kMaxT ? P.lds_rows - 2u : ct::kMaxT;
    cj.taNoGoal = (J.ctx_flags & kTaNoGoal) ? 1u : 0u;
    cj.rows = 0;
    cj.vc = (uint64_t)vc; cj.ec = (uint64_t)ec;
    cj.obst = (uint64_t)(P.maps + J.map_word_off);
    cj.pathsG = (uint64_t)heur;
    cj.parentTab = (uint64_t)arenaSlot;
    cj.outPath = (uint64_t)outPath;
    {
      auto w32 = (__attribute__((address_space(3))) uint32_t*)((wv::Lds)smem + ct::oJob);
      const uint32_t* src = (const uint32_t*)&cj;
#pragma unroll
      for (uint32_t q = 0; q < sizeof(ct::CJob) / 4; ++q) w32[q] = src[q];
    }
    const uint64_t tl0 = __builtin_amdgcn_s_memrealtime();
    const int32_t crc = ct::compactSearchTA((wv::Lds)smem);
    auto r32 = (__attribute__((address_space(3))) const uint32_t*)((wv::Lds)smem + ct::oRes);
    res.prof[0] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tl0);
    res.prof[1] = rfl(r32[4]);
    if (crc != ct::C_CAP_NODES && crc != ct::C_CAP_HORIZON) {  // an answer (C_OK / C_NO_SOLUTION / C_CAP_EXP == the ST_ codes)
      res.status = crc;
      res.cost = (int32_t)rfl(r32[1]);
      res.fmin = (int32_t)rfl(r32[2]);
      res.n_states = (int32_t)rfl(r32[3]);
      res.expanded = rfl(r32[4]);
      res.nodes_created = rfl(r32[5]);
      return;
    }
    res.prof[6] = rfl(r32[4]);  // expansions thrown away with the attempt
    res.prof[7] = 1;
    __syncthreads();
  }
  const uint64_t th0 = __builtin_amdgcn_s_memrealtime();
  runTaArena(P, J, arenaSlot, res, outPath, vc, ec, heur, small ? 32u : J.dimx);
  res.prof[2] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - th0);
  res.prof[3] = (uint32_t)res.expanded;
}

// ---- SIPP (config 5): A* over (cell, safe interval) states ---------------------------------------------------
// Reference: SIPP::search sipp.hpp:91-134 -> AStar::search a_star.hpp:63-161 over SIPPState with
// SIPPEnvironment::getNeighbors sipp.hpp:191-223 (motions Up, Down, Left, Right of mapf_prioritized_sipp.cpp:99-121;
// isCommandValid :129-142: arrival t = max(si.start, g + 1), cost t - g; swaps are not checked) and isSolution
// sipp.hpp:185-189 (goal cell AND the interval ends at INT_MAX).  Edge costs vary, so the decrease-key branch
// a_star.hpp:139-145 (`openSet.increase(handle)` == sift-up from the handle's position) is live here.
// Job tables (packed by the host, copied into the arena slot): cellIdx[cells] (halfwords; 0 = single default interval
// [0, INT_MAX], k+1 = special cell k), specFirst[K+1], ivals[total][2].  State id = cell for default cells,
// cells + specFirst[k] + i for interval i of special cell k.  HBM tier only.
constexpr int32_t kIntMax = 0x7FFFFFFF;

// Where runSipp finds a cell's safe intervals and a state's open/closed status.
//   RES = false: the job's own compact table, copied from the host into the arena (layout above); status words
//                (0 unseen, node + 1 in open, bit 31 closed) in the arena too, zeroed per job.
//   RES = true:  the device-resident table of an mrp_ll_sipp_table (ll_device.h kSippResident): per cell a 64-byte row
//                of bounds words + count (`ivals`) and a 64-byte row of status words, tagged with the job's epoch so
//                nothing is zeroed per job.
template <bool RES>
struct SippView {
  static constexpr uint32_t kClosed = RES ? kSippStClosed : 0x80000000u;
  const uint16_t* cellIdx;
  const uint32_t* specFirst;
  const uint8_t* cnt;
  const int32_t* ivals;
  uint32_t* status;
  uint32_t cells, epochBits;
  // nk != 0: the cell has its own interval list, `n` entries from ivals[2 * first]; nk == 0: the default [0, INT_MAX]
  DEVI void lookup(uint32_t cell, uint32_t& nk, uint32_t& first, uint32_t& n) const {
    if constexpr (RES) {
      nk = (uint32_t)ivals[cell * kSippRowWords + 15u];
      first = 0;
      n = nk ? nk - 1 : 1;
    } else {
      nk = cellIdx[cell];
      first = 0;
      n = 1;
      if (nk) {
        first = specFirst[nk - 1];
        n = specFirst[nk] - first;
      }
    }
  }
  DEVI uint32_t sid(uint32_t cell, uint32_t nk, uint32_t first, uint32_t i) const {
    if constexpr (RES) return cell * kSippRowWords + i;
    return nk ? cells + first + i : cell;
  }
  DEVI uint32_t getSt(uint32_t id) const {
    uint32_t v = status[id];
    if constexpr (RES) v = (v >> kSippEpochShift) == (epochBits >> kSippEpochShift) ? (v & ((1u << kSippEpochShift) - 1u)) : 0u;
    return v;
  }
  DEVI void putSt(uint32_t id, uint32_t v) const { status[id] = RES ? (v | epochBits) : v; }
  // a bounds word of the resident table (ll_device.h)
  DEVI static int32_t bStart(uint32_t w) { return (int32_t)(w & 0xFFFFu); }
  DEVI static int32_t bEnd(uint32_t w) { return (w >> 16) == kSippEndInf ? 0x7FFFFFFF : (int32_t)(w >> 16); }
  DEVI static uint32_t bPack(int32_t s, int32_t e) { return (uint32_t)s | (e == 0x7FFFFFFF ? kSippEndInf : (uint32_t)e) << 16; }
};

// SIPP node records.  Arena tier: u32x4 { x = cell | interval << 16 | (RES: interval ends at INT_MAX) << 31, parent, g,
// position of the open entry }.  TierLdsSipp: one word  cell | interval << 16 | endsAtInf << 20 | parent << 21  (interval
// < kSippCap = 16; parent < 2047, 0x7FF = none), g in Mem::gOf, the position in Mem::pos.
constexpr uint32_t kSippNoParentLds = 0x7FFu;
template <class T>
DEVI uint32_t sippNodeX(Mem<T>& m, uint32_t id) {
  if constexpr (T::kWideNodes) {
    return rfl(((typename Mem<T>::PNode4)m.nodes)[id].x);
  } else {
    const uint32_t w = rfl(m.nodes[id]);
    return sippUnpackX(w);
  }
}
template <class T>
DEVI void sippNodeNew(Mem<T>& m, uint32_t id, uint32_t x, uint32_t parent, uint32_t t) {
  if constexpr (T::kWideNodes) {
    u32x4 nn;
    nn.x = x;
    nn.y = parent;
    nn.z = t;
    nn.w = 0;
    ((typename Mem<T>::PNode4)m.nodes)[id] = nn;
  } else {
    m.nodes[id] = sippPackX(x) | (parent & kSippNoParentLds) << kSippXBits;
    m.gOf[id] = (uint16_t)t;
  }
}
// g and open position of a node that is in the open list (decrease-key, a_star.hpp:130-146)
template <class T>
DEVI void sippNodeGPos(Mem<T>& m, uint32_t id, uint32_t& gOld, uint32_t& pos) {
  if constexpr (T::kWideNodes) {
    const u32x4 on = ((typename Mem<T>::PNode4)m.nodes)[id];
    gOld = rfl(on.z);
    pos = rfl(on.w);
  } else {
    gOld = rfl((uint32_t)m.gOf[id]);
    pos = rfl((uint32_t)m.pos[id]);
  }
}
template <class T>
DEVI void sippNodeReparent(Mem<T>& m, uint32_t id, uint32_t parent, uint32_t t) {  // cameFrom update + new g
  if constexpr (T::kWideNodes) {
    m.nodes[id * 4 + 1] = parent;
    m.nodes[id * 4 + 2] = t;
  } else {
    m.nodes[id] = (rfl(m.nodes[id]) & ((1u << kSippXBits) - 1u)) | (parent & kSippNoParentLds) << kSippXBits;
    m.gOf[id] = (uint16_t)t;
  }
}
template <class T>
DEVI void sippNodePath(Mem<T>& m, uint32_t id, uint32_t& cell, uint32_t& gN, uint32_t& parent) {
  if constexpr (T::kWideNodes) {
    const u32x4 pn = ((typename Mem<T>::PNode4)m.nodes)[id];
    cell = rfl(pn.x) & 0xFFFFu;
    gN = rfl(pn.z);
    parent = rfl(pn.y);
  } else {
    const uint32_t w = rfl(m.nodes[id]);
    cell = w & 0xFFFFu;
    gN = rfl((uint32_t)m.gOf[id]);
    parent = (w >> kSippXBits) == kSippNoParentLds ? kNoParent : (w >> kSippXBits);
  }
}

template <class T>
DEVI typename T::E sippEntry(uint32_t f, uint32_t gN, uint32_t id, uint32_t x) {
  typename T::E e = T::pack(0, f, gN, id);
  if constexpr (T::kEntryHasX) e = T::withX(e, x);
  return e;
}

struct SippState {  // wave-uniform
  uint32_t nNodes, nOpen;
  int64_t expansions;
};

// The search loop over one memory tier.  Returns the job's status, or RUN_MIGRATE_NODES when the LDS tier has no room
// for the successors of the next expansion (nothing of that expansion has happened yet: the caller copies nodes and
// open list into the arena and calls the TierHbm instance with the same state).
template <class T, bool RES>
DEVI int32_t sippLoop(const LaunchParams& P, const DevJob& J, Mem<T>& g, const SippView<RES>& tv, SippState& s,
                      DevResult& res, uint16_t* outPath) {
  const uint32_t lane = threadIdx.x;
  const uint32_t dimx = J.dimx, dimy = J.dimy;
  const uint32_t gx = J.gx, gy = J.gy;
  const int64_t maxExp = J.max_expansions;
  const uint32_t* obst = P.maps + J.map_word_off;
  const int32_t* ivals = tv.ivals;
  uint32_t& nNodes = s.nNodes;
  uint32_t& nOpen = s.nOpen;
  int64_t& expansions = s.expansions;
  const uint32_t divMagic = rfl(0xFFFFFFFFu / dimx + 1u);
  for (;;) {
    if (nOpen == 0) {
      res.status = ST_NO_SOLUTION;
      break;
    }
    const typename T::E curE = ldU<T>(g.open, 0);
    const uint32_t curId = T::id(curE);
    uint32_t cw;
    if constexpr (T::kEntryHasX)
      cw = T::xOf(curE);
    else
      cw = sippNodeX<T>(g, curId);
    const uint32_t cell = cw & 0xFFFF, iv = RES ? (cw >> 16) & 0x7FFFu : cw >> 16;
    const uint32_t gcur = T::g(curE);  // == the node's g: every entry is packed with it
    // cell / dimx without the ~30-instruction division sequence: one multiply-high by floor(2^32 / dimx) + 1 is exact for
    // cell < 2^16 and dimx <= 2^16 (the product overshoots cell / dimx by less than 2^-16 < 1 / dimx)
    const uint32_t cy = dimx == 1u ? cell : __umulhi(cell, divMagic), cx = cell - cy * dimx;
    // RES: every table word this expansion needs has an address that follows from (cell, iv) alone — the cell's own
    // list length and interval end, and for the four neighbours (lanes 16 * motion + i) the obstacle word, the list
    // length, interval slot i and its status word: two 64-byte sectors per cell.  All of
    // it is requested here, in ONE round trip, and the heap pop below (LDS tier) runs while it is in flight; slots beyond
    // a list's length hold stale words that are loaded and ignored.  "Ends at INT_MAX", which the goal test needs at
    // once, rides in bit 31 of the node's x.
    uint32_t ck = 0, f0 = 0, nCur = 0;
    int32_t endT = kIntMax;
    uint32_t r_nc = 0, r_obstW = 0xFFFFFFFFu, r_nk = 0, r_st = 0, r_ck = 0, r_h = 0;
    uint32_t r_bw = 0, r_endW = 0;  // raw bounds words: decoded where they are used, BEHIND the heap pop (a decode here
                                    // would make the wavefront wait for the table before it pops)
    bool r_inb = false;
    if constexpr (RES) {
      const uint32_t mm = lane >> 4, i = lane & 15u;
      const uint32_t nx = cx + (mm == 3) - (mm == 2), ny = cy + (mm == 0) - (mm == 1);
      r_inb = nx < dimx && ny < dimy;
      r_nc = r_inb ? ny * dimx + nx : 0;
      r_h = (nx > gx ? nx - gx : gx - nx) + (ny > gy ? ny - gy : gy - ny);
      const int32_t* rowN = ivals + r_nc * kSippRowWords;
      r_ck = (uint32_t)ivals[cell * kSippRowWords + 15u];
      r_endW = (uint32_t)ivals[cell * kSippRowWords + iv];
      r_obstW = obst[r_nc >> 5];
      r_nk = (uint32_t)rowN[15];
      if (i < kSippCap) {
        r_bw = (uint32_t)rowN[i];
        r_st = tv.status[r_nc * kSippRowWords + i];
      }
      if (!(cw >> 31)) endT = 0;  // any finite value: the goal test below only asks whether it is INT_MAX
    } else {
      tv.lookup(cell, ck, f0, nCur);
      ck = rfl(ck);
      f0 = rfl(f0);
      if (ck) endT = rfli(ivals[2 * (f0 + iv) + 1]);
    }
    if constexpr (T::AS == 3) {  // LDS tiers: room for every successor of this expansion, or continue in the next tier
      if (nNodes + 4 * kSippCap > g.capNodes || nOpen + 4 * kSippCap > g.capHeap) return RUN_MIGRATE_NODES;
    }
    expansions += 1;
    if (maxExp >= 0 && expansions > maxExp) {
      res.status = ST_CAP_EXP;
      break;
    }
    if (cx == gx && cy == gy && endT == kIntMax) {
      // raw A* solution: (cell, g) per state; the host inserts the explicit Wait actions (sipp.hpp:105-128)
      uint32_t len = 0;
      for (uint32_t nid = curId; nid != kNoParent;) {
        uint32_t pc, pg, pp;
        sippNodePath<T>(g, nid, pc, pg, pp);
        nid = pp;
        len += 1;
      }
      if (len * 2 > P.out_stride) {
        res.status = ST_CAP_HORIZON;
        break;
      }
      uint32_t* out32 = (uint32_t*)outPath;
      uint32_t nid = curId;
      for (int32_t k = (int32_t)len - 1; k >= 0; --k) {
        uint32_t pc, pg, pp;
        sippNodePath<T>(g, nid, pc, pg, pp);
        out32[k] = pc | (pg << 16);
        nid = pp;
      }
      res.status = ST_OK;
      res.cost = (int32_t)gcur;
      res.fmin = (int32_t)T::f(curE);
      res.n_states = (int32_t)len;
      break;
    }
    heapPop<T, 0, true>(g, g.open, nOpen);
    if constexpr (RES) {
      ck = rfl(r_ck);
      f0 = 0;
      if (ck) endT = SippView<RES>::bEnd(rfl(r_endW));
      else endT = kIntMax;
    }
    const uint32_t curSid = tv.sid(cell, ck, f0, iv);
    tv.putSt(curSid, SippView<RES>::kClosed);
    const uint32_t startT = gcur + 1;
    if (startT > kGMask) {
      res.status = ST_CAP_HORIZON;
      break;
    }
    bool fail = false;
    // ---- neighbours.  Lanes 0..3 probe the four motions Up, Down, Left, Right at once (bounds, obstacle bit, the
    // cell's safe-interval list); when no list is longer than 16 the intervals of all four cells are then evaluated on
    // lanes 16*m + i together with their open/closed status — three dependent global round trips per expansion
    // instead of three to five per motion.  Candidates are consumed in lane order, which IS the reference's order
    // (motion-major, interval-minor, sipp.hpp:205-222).
    uint32_t c4[4] = {0, 0, 0, 0}, f4[4] = {0, 0, 0, 0}, nc4[4] = {0, 0, 0, 0}, nk4[4] = {0, 0, 0, 0}, h4[4] = {0, 0, 0, 0};
    if constexpr (!RES) {
      const uint32_t nxL = cx + (lane == 3) - (lane == 2), nyL = cy + (lane == 0) - (lane == 1);
      const bool inbL = lane < 4 && nxL < dimx && nyL < dimy;
      const uint32_t ncL = inbL ? nyL * dimx + nxL : 0;
      uint32_t obstW = 0xFFFFFFFFu, nkL = 0, firstL = 0, cntL = 0;
      if (inbL) {
        obstW = obst[ncL >> 5];
        tv.lookup(ncL, nkL, firstL, cntL);
      }
      const bool validL = inbL && !((obstW >> (ncL & 31)) & 1u);
      if (!validL) cntL = 0;
      const uint32_t hL = (nxL > gx ? nxL - gx : gx - nxL) + (nyL > gy ? nyL - gy : gy - nyL);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        c4[m] = __builtin_amdgcn_readlane(cntL, m);
        f4[m] = __builtin_amdgcn_readlane(firstL, m);
        nc4[m] = __builtin_amdgcn_readlane(ncL, m);
        nk4[m] = __builtin_amdgcn_readlane(nkL, m);
        h4[m] = __builtin_amdgcn_readlane(hL, m);
      }
    }
    const uint32_t cMax = RES ? 0u : max(max(c4[0], c4[1]), max(c4[2], c4[3]));
    if (cMax <= 16) {
      const uint32_t mm = lane >> 4, i = lane & 15;
      uint32_t cntM, firstM, ncM, nkM, hM, sidL, stL = 0;
      int32_t siS = 0, siE = kIntMax;
      bool act;
      if constexpr (RES) {
        const bool valid = r_inb && !((r_obstW >> (r_nc & 31)) & 1u);
        nkM = r_nk;
        ncM = r_nc;
        hM = r_h;
        firstM = 0;
        cntM = valid ? (nkM ? nkM - 1 : 1u) : 0u;
        act = i < cntM;
        sidL = tv.sid(r_nc, 0, 0, i);
        if (act) {
          if (nkM) {
            siS = SippView<RES>::bStart(r_bw);
            siE = SippView<RES>::bEnd(r_bw);
          }
          stL = (r_st >> kSippEpochShift) == (tv.epochBits >> kSippEpochShift) ? (r_st & ((1u << kSippEpochShift) - 1u)) : 0u;
        }
      } else {
        cntM = mm == 0 ? c4[0] : mm == 1 ? c4[1] : mm == 2 ? c4[2] : c4[3];
        firstM = mm == 0 ? f4[0] : mm == 1 ? f4[1] : mm == 2 ? f4[2] : f4[3];
        ncM = mm == 0 ? nc4[0] : mm == 1 ? nc4[1] : mm == 2 ? nc4[2] : nc4[3];
        nkM = mm == 0 ? nk4[0] : mm == 1 ? nk4[1] : mm == 2 ? nk4[2] : nk4[3];
        hM = mm == 0 ? h4[0] : mm == 1 ? h4[1] : mm == 2 ? h4[2] : h4[3];
        act = i < cntM;
        sidL = tv.sid(ncM, nkM, firstM, i);
        if (act) {
          if (nkM) {
            siS = ivals[2 * (firstM + i)];
            siE = ivals[2 * (firstM + i) + 1];
          }
          stL = tv.getSt(sidL);
        }
      }
      // sipp.hpp:209: skip if si.start - m_time > end_t || si.end < start_t
      const bool cand = act && !((int64_t)siS - 1 > (int64_t)endT || siE < (int32_t)startT);
      const uint32_t tArr = (uint32_t)(siS > (int32_t)startT ? siS : (int32_t)startT);
      const uint64_t candMask = ballot64(cand);
      const uint64_t lateMask = ballot64(cand && tArr > kGMask);
      const uint64_t openMask = ballot64(cand && stL != 0 && !(stL & SippView<RES>::kClosed));   // already in the open list
      const uint64_t newMask = ballot64(cand && stL == 0);
      const uint32_t nNew = (uint32_t)__popcll(newMask);
      if (lateMask) {
        res.status = ST_CAP_HORIZON;
        fail = true;
      } else if (openMask == 0 && nNew <= 5) {
        // the usual case — nothing to re-key: all pushes of the expansion in one round trip (PushChains)
        if (nNodes + nNew > g.capNodes) {
          res.status = ST_CAP_NODES;
          fail = true;
        } else if (nNew) {
          typename T::E e[5];
          uint64_t mk = newMask;
#pragma unroll
          for (uint32_t k = 0; k < 5; ++k) {
            e[k] = 0;
            if (k < nNew) {
              const uint32_t l = (uint32_t)__builtin_ctzll(mk);
              mk &= mk - 1;
              const uint32_t t = __builtin_amdgcn_readlane(tArr, l);
              const uint32_t sid = __builtin_amdgcn_readlane(sidL, l);
              const uint32_t nc = __builtin_amdgcn_readlane(ncM, l);
              const uint32_t hN = __builtin_amdgcn_readlane(hM, l);
              const uint32_t nid = nNodes + k;
              const uint32_t xN = nc | ((l & 15u) << 16) | (RES ? __builtin_amdgcn_readlane(siE == kIntMax ? 1u : 0u, l) << 31 : 0u);
              sippNodeNew<T>(g, nid, xN, curId, t);
              tv.putSt(sid, nid + 1);
              e[k] = sippEntry<T>(t + hN, t, nid, xN);
            }
          }
          const uint32_t pm = (1u << nNew) - 1u;
          PushChains<T> pc;
          pc.load(g.open, nOpen, pm);
          pc.template resolve<0, true>(g, g.open, nOpen, pm, e);
          nNodes += nNew;
          nOpen += nNew;
        }
      } else {
        uint64_t mask = candMask;
        while (mask && !fail) {
          const uint32_t l = (uint32_t)__builtin_ctzll(mask);
          mask &= mask - 1;
          const uint32_t t = __builtin_amdgcn_readlane(tArr, l);
          const uint32_t sid = __builtin_amdgcn_readlane(sidL, l);
          const uint32_t nc = __builtin_amdgcn_readlane(ncM, l);
          const uint32_t hN = __builtin_amdgcn_readlane(hM, l);
          const uint32_t st = __builtin_amdgcn_readlane(stL, l);
          if (st & SippView<RES>::kClosed) continue;                   // closedSet.find (a_star.hpp:117)
          const uint32_t xN = nc | ((l & 15u) << 16) | (RES ? __builtin_amdgcn_readlane(siE == kIntMax ? 1u : 0u, l) << 31 : 0u);
          if (st == 0) {                                   // new state (a_star.hpp:120-129)
            if (nNodes >= g.capNodes) {
              res.status = ST_CAP_NODES;
              fail = true;
              break;
            }
            const uint32_t nid = nNodes++;
            sippNodeNew<T>(g, nid, xN, curId, t);
            tv.putSt(sid, nid + 1);
            siftUp<T, 0, true>(g, g.open, nOpen, sippEntry<T>(t + hN, t, nid, xN));
            nOpen += 1;
          } else {                                         // already in open (a_star.hpp:130-146)
            const uint32_t nid = st - 1;
            uint32_t gOld, posOld;
            sippNodeGPos<T>(g, nid, gOld, posOld);
            if (t >= gOld) continue;
            sippNodeReparent<T>(g, nid, curId, t);
            siftUp<T, 0, true>(g, g.open, posOld, sippEntry<T>(t + hN, t, nid, xN));  // increase(handle)
          }
        }
      }
    } else {
      // a cell with more than 16 safe intervals: one motion at a time, 64 intervals per pass
      for (uint32_t m = 0; m < 4 && !fail; ++m) {  // Up, Down, Left, Right
        const uint32_t nx = cx + (m == 3) - (m == 2), ny = cy + (m == 0) - (m == 1);
        if (nx >= dimx || ny >= dimy) continue;
        const uint32_t nc = ny * dimx + nx;
        if ((rfl(obst[nc >> 5]) >> (nc & 31)) & 1u) continue;
        uint32_t nk, first, cnt;
        tv.lookup(nc, nk, first, cnt);
        nk = rfl(nk);
        first = rfl(first);
        cnt = rfl(cnt);
        const uint32_t hN = (nx > gx ? nx - gx : gx - nx) + (ny > gy ? ny - gy : gy - ny);
        for (uint32_t base = 0; base < cnt && !fail; base += 64) {
          const uint32_t i = base + lane;
          int32_t siS = 0, siE = kIntMax;
          if (nk && i < cnt) {
            siS = ivals[2 * (first + i)];
            siE = ivals[2 * (first + i) + 1];
          }
          // sipp.hpp:209: skip if si.start - m_time > end_t || si.end < start_t
          const bool cand = (i < cnt) && !((int64_t)siS - 1 > (int64_t)endT || siE < (int32_t)startT);
          const uint32_t tArr = (uint32_t)(siS > (int32_t)startT ? siS : (int32_t)startT);
          uint64_t mask = ballot64(cand);
          while (mask) {
            const uint32_t l = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1;
            const uint32_t ii = base + l;
            const uint32_t t = __builtin_amdgcn_readlane(tArr, l);
            if (t > kGMask) {
              res.status = ST_CAP_HORIZON;
              fail = true;
              break;
            }
            const uint32_t sid = tv.sid(nc, nk, first, ii);
            const uint32_t st = rfl(tv.getSt(sid));
            if (st & SippView<RES>::kClosed) continue;                   // closedSet.find (a_star.hpp:117)
            if (st == 0) {                                   // new state (a_star.hpp:120-129)
              if (nNodes >= g.capNodes) {
                res.status = ST_CAP_NODES;
                fail = true;
                break;
              }
              const uint32_t nid = nNodes++;
              sippNodeNew<T>(g, nid, nc | (ii << 16), curId, t);
              tv.putSt(sid, nid + 1);
              siftUp<T, 0, true>(g, g.open, nOpen, T::pack(0, t + hN, t, nid));
              nOpen += 1;
            } else {                                         // already in open (a_star.hpp:130-146)
              const uint32_t nid = st - 1;
              uint32_t gOld, posOld;
              sippNodeGPos<T>(g, nid, gOld, posOld);
              if (t >= gOld) continue;
              sippNodeReparent<T>(g, nid, curId, t);
              siftUp<T, 0, true>(g, g.open, posOld, T::pack(0, t + hN, t, nid));  // increase(handle)
            }
          }
        }
      }
    }
    if (fail) return res.status;
  }
  return res.status;
}

// sipp_commit (mrp_ll.h): the stays of the path just found — state k = (cell, arrival t_k) occupies its cell during
// [t_k, t_{k+1} - 1], the last one during [t_last, INT_MAX] — become collision intervals of the resident table, i.e. each
// splits the safe interval that contains it (what SIPP::setCollisionIntervals, sipp.hpp:245-284, yields for the longer
// collision list; the stay lies inside ONE safe interval because the search kept the agent there).  One lane per state,
// 64 states per pass; lanes whose states share a cell take turns.  Returns false if a cell would need more than kSippCap
// intervals or a stay is not inside a safe interval: the table is then left half-updated and the host redoes it.
DEVI bool sippCommitPath(const SippView<true>& tv, const uint32_t* path, uint32_t len) {
  const uint32_t lane = threadIdx.x;
  bool bad = false;
  for (uint32_t base = 0; base < len; base += 64) {
    const uint32_t k = base + lane;
    const bool act = k < len;
    const uint32_t w = act ? path[k] : 0xFFFFFFFFu;
    const uint32_t wn = (k + 1 < len) ? path[k + 1] : 0;
    const uint32_t cell = w & 0xFFFFu;
    const int32_t s0 = (int32_t)(w >> 16);
    const int32_t e0 = (k + 1 < len) ? (int32_t)(wn >> 16) - 1 : kIntMax;
    // how many earlier states of this pass sit on the same cell (a path may come back to a cell)
    uint32_t rank = 0;
    const uint32_t nAct = min(len - base, 64u);
    for (uint32_t j = 0; j + 1 < nAct; ++j) {
      const uint32_t cj = __builtin_amdgcn_readlane(cell, j);
      rank += (j < lane && cj == cell) ? 1u : 0u;
    }
    uint64_t todo = ballot64(act);
    for (uint32_t turn = 0; todo; ++turn) {
      const bool mine = act && rank == turn;
      if (mine) {
        u32x4* row4 = (u32x4*)(tv.ivals + cell * kSippRowWords);  // bounds words 0 .. 14, the count in word 15
        uint32_t rw[kSippRowWords];
#pragma unroll
        for (uint32_t v4 = 0; v4 < kSippRowWords / 4; ++v4) {
          const u32x4 v = row4[v4];
          rw[4 * v4] = v.x; rw[4 * v4 + 1] = v.y; rw[4 * v4 + 2] = v.z; rw[4 * v4 + 3] = v.w;
        }
        const uint32_t n1 = rw[15];
        const uint32_t n = n1 ? n1 - 1 : 1u;
        int32_t rs[kSippCap], re[kSippCap];
        if (n1) {
#pragma unroll
          for (uint32_t q = 0; q < kSippCap; ++q) {
            rs[q] = SippView<true>::bStart(rw[q]);
            re[q] = SippView<true>::bEnd(rw[q]);
          }
        } else {
#pragma unroll
          for (uint32_t q = 0; q < kSippCap; ++q) { rs[q] = 0; re[q] = -1; }
          rs[0] = 0;
          re[0] = kIntMax;
        }
        uint32_t kk = kSippCap;  // the safe interval that contains the stay
#pragma unroll
        for (uint32_t q = kSippCap; q-- > 0;)
          if (q < n && rs[q] <= s0 && e0 <= re[q]) kk = q;
        int32_t a = 0, b = 0;
#pragma unroll
        for (uint32_t q = 0; q < kSippCap; ++q)
          if (q == kk) { a = rs[q]; b = re[q]; }
        const bool left = a <= s0 - 1, right = e0 < b;
        const int32_t d = (left ? 1 : 0) + (right ? 1 : 0) - 1;
        if (kk == kSippCap || n + d > kSippCap) {
          bad = true;
        } else {
          int32_t ns[kSippCap], ne[kSippCap];
#pragma unroll
          for (uint32_t q = 0; q < kSippCap; ++q) {
            // entry q of the new list: below kk unchanged; at kk the left part, else the right part, else (both gone)
            // the old successor; above kk the old list shifted by d
            int32_t vs = rs[q], ve = re[q];
            if (q >= kk) {
              const int32_t ps = q > 0 ? rs[q - 1] : 0, pe = q > 0 ? re[q - 1] : 0;          // old[q - 1]
              const int32_t fs = q + 1 < kSippCap ? rs[q + 1] : 0, fe = q + 1 < kSippCap ? re[q + 1] : 0;  // old[q + 1]
              if (d == 1) {
                if (q == kk) { vs = a; ve = s0 - 1; }
                else if (q == kk + 1) { vs = e0 + 1; ve = b; }
                else { vs = ps; ve = pe; }
              } else if (d == 0) {
                if (q == kk) { vs = left ? a : e0 + 1; ve = left ? s0 - 1 : b; }
              } else {
                vs = fs; ve = fe;
              }
            }
            ns[q] = vs;
            ne[q] = ve;
          }
          // (slots beyond the new list get whatever the shift brought along; nobody reads them.  The status words are
          // left alone: they belong to this job's epoch, and the next job of the table has another)
#pragma unroll
          for (uint32_t q = 0; q < kSippCap; ++q) rw[q] = SippView<true>::bPack(ns[q] & 0xFFFF, ne[q] == kIntMax ? kIntMax : (ne[q] & 0xFFFF));
          rw[15] = n + d + 1;
#pragma unroll
          for (uint32_t v4 = 0; v4 < kSippRowWords / 4; ++v4) {
            u32x4 v;
            v.x = rw[4 * v4]; v.y = rw[4 * v4 + 1]; v.z = rw[4 * v4 + 2]; v.w = rw[4 * v4 + 3];
            row4[v4] = v;
          }
        }
      }
      todo &= ~ballot64(mine);
      // the next turn reads rows this one wrote (other lanes of the same wave; the table is uncached memory, so a store
      // that has been acknowledged is what a later load sees)
      if (todo) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (base + 64 < len) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  return ballot64(bad) == 0;
}

// LDS of a resident SIPP workgroup: TierLdsSipp's nodes, positions, g and open list — or TierMix's open list alone
// 768 nodes = 9.2 KB = 16 searches per CU.  With the tables in uncached memory and no fence pair per job, residency pays
// (scripts/r4_run35.sh, r4_run36.sh, 100 / 200 agents: 2048 nodes, 6 per CU: 5.5 / 5.5e8 expansions/s; 1536, 8 per CU:
// 6.5 / 6.5e8; 1024, 12 per CU: 7.2 / 7.4e8; 768, 16 per CU: 7.8 / 8.0e8; 512, 16 per CU: 7.6 / 7.1e8) although every
// expansion gets slower (2.2 -> 2.7 us) and more searches continue in the middle tier (open list in LDS, nodes in the arena).
#ifndef MRP_LL_SIPP_LDS_NODES
#define MRP_LL_SIPP_LDS_NODES 768
#endif
constexpr uint32_t kSippLdsCap = MRP_LL_SIPP_LDS_NODES;  // <= TierLdsSipp::kMaxNodes; ids 0 .. kSippLdsCap - 2 are used
static_assert(kSippLdsCap <= TierLdsSipp::kMaxNodes && kSippLdsCap % 4 == 0, "SIPP LDS tier capacity");
constexpr uint32_t kSippLdsBytesC = kSippLdsCap * (4 + 2 + 2) + kSippLdsCap * 4 + 16;

template <bool RES>
DEVI void runSipp(const LaunchParams& P, const DevJob& J, uint8_t* arenaSlot, uint8_t* ldsTier, uint32_t ldsNodes, DevResult& res,
                  uint16_t* outPath) {
  const uint32_t lane = threadIdx.x;
  const uint32_t dimx = J.dimx, dimy = J.dimy, cells = dimx * dimy;
  const uint32_t K = J.n_vc, totalIv = J.n_ec;
  const uint32_t gx = J.gx, gy = J.gy;
  typedef TierHbm T;
  Mem<T> g;
  Mem<T>::PNode4 gNodes;
  {
    uint8_t* p = arenaSlot;
    g.nodes = (Mem<T>::PN32)p;
    gNodes = (Mem<T>::PNode4)p;                  p += (size_t)P.arena_nodes * 16;
    g.pos = nullptr;
    g.open = (Mem<T>::PE)(p + 8);                p += (size_t)P.arena_nodes * 8 + 16;
    g.focal = (Mem<T>::PE)(p + 8);               p += (size_t)P.arena_nodes * 8 + 16;
    g.aux = (Mem<T>::PE)(p + 8);                 p += (size_t)P.arena_nodes * 8 + 16;
    g.bits = (Mem<T>::P32)p;
    g.capNodes = P.arena_nodes; g.capHeap = P.arena_nodes; g.capRows = P.arena_rows; g.rowWords = P.arena_row_words;
  }
  uint8_t* scratch = arenaSlot + P.arena_scratch_off;
  uint32_t* tab = (uint32_t*)((uint32_t*)(scratch + (size_t)P.out_stride * 2) + kConsLocalWords);  // path-table area
  SippView<RES> tv;
  tv.cells = cells;
  tv.epochBits = 0;
  if constexpr (RES) {
    uint8_t* rt = (uint8_t*)((uint64_t)J.n_agents_pad | ((uint64_t)J.path_off << 32));
    uint32_t* rec = (uint32_t*)rt;  // bounds rows, then status rows (ll_device.h)
    tv.cnt = nullptr;
    tv.ivals = (const int32_t*)rec;
    tv.status = rec + (size_t)cells * kSippRowWords;
    tv.epochBits = J.n_ctx << kSippEpochShift;
    tv.cellIdx = nullptr;
    tv.specFirst = nullptr;
    // (the table was last written by another workgroup, possibly on another XCD: it is uncached memory, that workgroup's
    // stores had been acknowledged before it published its job as done, and the host packed this job after seeing that)
    const uint32_t nRec = J.ec_off & 0x7FFFFFFFu;
    if (J.ec_off >> 31) {  // first job of the table (or its epochs are used up): no cell has a list, no state is seen
      u32x4 z;
      z.x = z.y = z.z = z.w = 0;
      u32x4* s4 = (u32x4*)rec;
      for (uint32_t i = lane; i < cells * (2u * kSippRowWords / 4); i += 64) s4[i] = z;
      __syncthreads();
    }
    // the cells whose lists changed since the table's previous job, out of pinned host memory: lane u copies 16 bytes
    // (four bounds words) of record u / recUnits, four rounds in flight
    const uint32_t* hdr = P.cons + J.vc_off;
    const u32x4* recs = (const u32x4*)(hdr + ((nRec + 3u) & ~3u));
    const uint32_t recUnits = J.n_vc / 4;      // 16-byte units per record of this job (a power of two, 1 .. 4)
    const uint32_t recShift = 31u - (uint32_t)__builtin_clz(recUnits | 1u);
    const uint32_t nUnits = nRec * recUnits;
    for (uint32_t u0 = 0; u0 < nUnits; u0 += 256) {
      uint32_t h[4];
      u32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t u = u0 + q * 64 + lane;
        if (u < nUnits) {
          h[q] = __builtin_nontemporal_load(hdr + (u >> recShift));
          v[q] = __builtin_nontemporal_load(recs + u);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t u = u0 + q * 64 + lane;
        if (u < nUnits) {
          const uint32_t cell = h[q] & 0xFFFFu;
          // the unit holds bounds words q0 .. q0 + 3; word 15 of a row is the count, which the host has put into the last
          // word of a 16-word record (packSippResident), and which the lane of unit 0 writes for a shorter one
          const uint32_t q0 = u & (recUnits - 1u);
          ((u32x4*)rec)[cell * (kSippRowWords / 4) + q0] = v[q];
          if (q0 == 0 && recUnits < 4u) rec[cell * kSippRowWords + 15u] = (h[q] >> 16) + 1u;
        }
      }
    }
    __syncthreads();
  } else {
  const uint32_t cw = (cells + 1) / 2;  // cellIdx is a halfword per cell (cells <= 65025, so K + 1 fits)
  const uint32_t tabWords = cw + K + 1 + 2 * totalIv;
  const uint32_t nStates = cells + totalIv;
  if (tabWords * 4 > P.arena_paths_bytes || nStates > P.arena_rows * P.arena_row_words) {
    res.status = ST_CAP_NODES;
    return;
  }
  {
    // The job's safe-interval table (10-30 KB) comes out of pinned HOST memory: every load instruction is a PCIe round
    // trip, so the copy is made of 16-byte lanes with eight loads in flight per lane (8 KB per round trip); dword by
    // dword it was ~100 dependent round trips and the largest part of a job's time.
    const uint32_t* src = P.cons + J.vc_off;
    uint32_t done = 0;
    if ((J.vc_off & 3u) == 0) {  // session slots are 16-byte aligned; a batch's tables start wherever the previous ended
      const u32x4* src4 = (const u32x4*)src;
      u32x4* dst4 = (u32x4*)tab;
      const uint32_t n4 = tabWords / 4;
      uint32_t i = lane;
      for (; i + 7 * 64 < n4; i += 8 * 64) {
        u32x4 v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = __builtin_nontemporal_load(src4 + i + q * 64);
#pragma unroll
        for (int q = 0; q < 8; ++q) dst4[i + q * 64] = v[q];
      }
      for (; i < n4; i += 64) dst4[i] = __builtin_nontemporal_load(src4 + i);
      done = n4 * 4;
    }
    for (uint32_t i = done + lane; i < tabWords; i += 64) tab[i] = src[i];
    u32x4 z;
    z.x = z.y = z.z = z.w = 0;  // status: 0 unseen, node+1 in open, bit 31 closed
    u32x4* st4 = (u32x4*)(uint32_t*)g.bits;
    for (uint32_t i = lane; i < (nStates + 3) / 4; i += 64) st4[i] = z;
  }
  __syncthreads();
  tv.cellIdx = (const uint16_t*)tab;
  tv.specFirst = tab + cw;
  tv.ivals = (const int32_t*)(tab + cw + K + 1);
  tv.status = (uint32_t*)g.bits;
  tv.cnt = nullptr;
  }
  // start interval (findSafeInterval, sipp.hpp:286-296): found by the host for a table that travels with the job, here
  // for a resident one (the host's copy may be behind)
  uint32_t startIv = J.t_pad, startInf = 0;
  if constexpr (RES) {
    const uint32_t sc = J.sy * dimx + J.sx;
    const int32_t st0 = J.last_goal_constraint;
    const uint32_t n1 = rfl((uint32_t)tv.ivals[sc * kSippRowWords + 15u]);
    if (n1 == 0) {
      startIv = 0;
      startInf = 1;
    } else {
      int32_t a = 0, b = -1;
      if (lane < n1 - 1) {
        const uint32_t bw = (uint32_t)tv.ivals[sc * kSippRowWords + lane];
        a = SippView<true>::bStart(bw);
        b = SippView<true>::bEnd(bw);
      }
      const uint64_t hit = ballot64(lane < n1 - 1 && a <= st0 && b >= st0);
      if (hit) {
        startIv = (uint32_t)__builtin_ctzll(hit);
        startInf = __builtin_amdgcn_readlane(b == kIntMax ? 1u : 0u, startIv);
      } else {
        startIv = 0xFFFFFFFFu;
      }
    }
  }
  if (startIv == 0xFFFFFFFFu) {  // no safe interval contains the start time: SIPP::search returns false (sipp.hpp:98-100)
    res.status = ST_NO_SOLUTION; // (after the table update: a resident table must not miss this job's delta)
    return;
  }

  // start node
  SippState s;
  s.nNodes = 1;
  s.nOpen = 1;
  s.expansions = 0;
  u32x4 n0;
  uint64_t e0;
  {
    const uint32_t sc = J.sy * dimx + J.sx;
    const uint32_t si = startIv;
    const uint32_t h0 = (J.sx > gx ? J.sx - gx : gx - J.sx) + (J.sy > gy ? J.sy - gy : gy - J.sy);
    n0.x = sc | (si << 16) | (RES ? startInf << 31 : 0u);  // RES: bit 31 = the start interval ends at INT_MAX
    n0.y = kNoParent;
    // SIPP::search(..., startTime) (sipp.hpp:92-103): the start node's g is startTime, its f is h(start) alone
    // (a_star.hpp:78 pushes Node(start, h, initialCost))
    const uint32_t startTime = (uint32_t)J.last_goal_constraint;
    n0.z = startTime;
    n0.w = 0;
    e0 = TierHbm::pack(0, h0, startTime, 0);
    uint32_t k, f0, n0c;
    tv.lookup(sc, k, f0, n0c);
    tv.putSt(tv.sid(sc, rfl(k), rfl(f0), si), 1);
  }
  int32_t rc = RUN_MIGRATE_NODES;
  if constexpr (RES) {
    if (ldsTier) {
      // fast tier: nodes and open list in LDS (the table and the status words stay in HBM, one round trip per expansion)
      typedef TierLdsSipp TL;
      Mem<TL> gl;
      auto l8 = (__attribute__((address_space(3))) uint8_t*)ldsTier;
      gl.nodes = (Mem<TL>::PN32)l8;
      gl.pos = (Mem<TL>::P16)(l8 + (size_t)ldsNodes * 4);
      gl.gOf = (Mem<TL>::P16)(l8 + (size_t)ldsNodes * 6);
      gl.open = (Mem<TL>::PE)(l8 + (size_t)ldsNodes * 8 + 4);
      gl.focal = nullptr;
      gl.aux = nullptr;
      gl.bits = nullptr;
      gl.capNodes = ldsNodes - 1;  // ids below kSippNoParentLds
      gl.capHeap = ldsNodes; gl.capRows = 0; gl.rowWords = 0;
      sippNodeNew<TL>(gl, 0, n0.x, kNoParent, n0.z);
      gl.pos[0] = 0;
      gl.open[0] = TL::pack(0, TierHbm::f(e0), n0.z, 0);
      __syncthreads();
      const uint64_t tl0 = __builtin_amdgcn_s_memrealtime();
      rc = sippLoop<TL, RES>(P, J, gl, tv, s, res, outPath);
      res.prof[0] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tl0);  // 100 MHz ticks / expansions in the LDS tier
      res.prof[1] = (uint32_t)s.expansions;
      if (rc == RUN_MIGRATE_NODES) {
        __syncthreads();
        for (uint32_t i = lane; i < s.nNodes; i += 64) {
          const uint32_t w = gl.nodes[i];
          u32x4 nn;
          nn.x = sippUnpackX(w);
          nn.y = (w >> kSippXBits) == kSippNoParentLds ? kNoParent : (w >> kSippXBits);
          nn.z = gl.gOf[i];
          nn.w = gl.pos[i];
          gNodes[i] = nn;
        }
        // the open list: 64-bit entries with the node's x word (TierMix), staged through the arena's open array because
        // the new list covers the area the old one and the node records occupy
        for (uint32_t i = lane; i < s.nOpen; i += 64) {
          const uint32_t e = gl.open[i];
          const uint32_t w = gl.nodes[TL::id(e)];
          g.open[i] = TierMix::withX(TierHbm::pack(0, TL::f(e), TL::g(e), TL::id(e)), sippUnpackX(w));
        }
        __syncthreads();
        Mem<TierMix> gm;
        gm.nodes = g.nodes;
        gm.pos = nullptr;
        gm.gOf = nullptr;
        gm.open = (Mem<TierMix>::PE)(l8 + 8);
        gm.focal = nullptr;
        gm.aux = nullptr;
        gm.bits = nullptr;
        gm.capNodes = P.arena_nodes;
        gm.capHeap = (kSippLdsBytesC - 16) / 8;
        gm.capRows = 0; gm.rowWords = 0;
        for (uint32_t i = lane; i < s.nOpen; i += 64) gm.open[i] = g.open[i];
        __syncthreads();
        res.tier = 2;  // started in LDS, the open list still there
        const uint64_t tm0 = __builtin_amdgcn_s_memrealtime();
        const int64_t e0m = s.expansions;
        rc = sippLoop<TierMix, RES>(P, J, gm, tv, s, res, outPath);
        res.prof[6] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tm0);  // ... in the middle tier
        res.prof[7] = (uint32_t)(s.expansions - e0m);
        if (rc == RUN_MIGRATE_NODES) {  // the open list has outgrown LDS too: everything in the arena
          __syncthreads();
          for (uint32_t i = lane; i < s.nOpen; i += 64) g.open[i] = gm.open[i];
          __syncthreads();
          res.tier = 3;
        }
      } else {
        res.tier = 0;
      }
    } else {
      gNodes[0] = n0;
      g.open[0] = TierHbmX::withX(e0, n0.x);
    }
  } else {
    gNodes[0] = n0;
    g.open[0] = e0;
  }
  if (rc == RUN_MIGRATE_NODES) {
    const uint64_t th0 = __builtin_amdgcn_s_memrealtime();
    const int64_t e0h = s.expansions;
    if constexpr (RES) {
      Mem<TierHbmX> gx;
      gx.nodes = g.nodes; gx.pos = nullptr; gx.gOf = nullptr;
      gx.open = (Mem<TierHbmX>::PE)g.open; gx.focal = nullptr; gx.aux = nullptr; gx.bits = nullptr;
      gx.capNodes = g.capNodes; gx.capHeap = g.capHeap; gx.capRows = 0; gx.rowWords = 0;
      rc = sippLoop<TierHbmX, RES>(P, J, gx, tv, s, res, outPath);
    } else {
      rc = sippLoop<TierHbm, RES>(P, J, g, tv, s, res, outPath);
    }
    res.prof[2] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - th0);  // ... in the arena tier
    res.prof[3] = (uint32_t)(s.expansions - e0h);
    if (res.tier == 0) res.tier = 1;
  }
  res.status = rc;
  res.expanded = s.expansions;
  res.nodes_created = s.nNodes;
  if constexpr (RES) {
    if (rc == ST_OK && (J.ctx_flags & kSippCommit)) {
      __syncthreads();
      if (!sippCommitPath(tv, (const uint32_t*)outPath, (uint32_t)res.n_states)) res.tier |= kSippTierCommitFailed;
    }
  }
}

// Runs the job whose descriptor is at `jobSrc` (host memory) and writes result + path to host memory.
// KIND: 0 = the job's own algo field decides (mixed batches / sessions), 1 = A*-epsilon jobs only (ECBS), 2 = A* jobs
// only (CBS).  The specialised kernels carry one search loop per memory tier instead of two, which halves their code
// (instruction-cache footprint) and takes the other algorithm's live ranges out of the register allocation; a job of
// the other kind comes back as ST_BAD.
// Returns true when the job was NOT run here and has to be handed to the heavy workgroups (TIERS == kTiersFront only):
// nothing has been written to the job's result area then.
template <int KIND, int TIERS = kTiersAll>
DEVI bool processJob(const LaunchParams& P, const DevJob* jobSrc, DevResult* resDst, uint16_t* pathDst, uint8_t* smem,
                     uint8_t* arenaSlot, DevJob& jobS, DevResult& resS) {
  static_assert(TIERS == kTiersAll || KIND == 1, "front / heavy workgroups exist for the A*-epsilon sessions");
  const uint32_t lane = threadIdx.x;
  __syncthreads();
  {  // one coalesced read of the 80-byte descriptor from host memory
    const uint32_t* src = (const uint32_t*)jobSrc;
    if (lane < sizeof(DevJob) / 4) ((uint32_t*)&jobS)[lane] = hostLoad32(src + lane);
  }
  __syncthreads();
  const DevJob& J = jobS;
  DevResult res;
  res.status = ST_BAD; res.cost = 0; res.fmin = 0; res.n_states = 0; res.expanded = 0; res.nodes_created = 0;
  res.tier = 0;
  for (int q = 0; q < 8; ++q) res.prof[q] = 0;
  PROF_T0();
#ifndef MRP_LL_TRACE
  const uint64_t tj0 = __builtin_amdgcn_s_memrealtime();
#endif
  uint16_t* outPath = (uint16_t*)(arenaSlot + P.arena_scratch_off);  // device scratch; copied out below
  const uint32_t algo = rfl(J.algo);
  bool handOver = false;
  if constexpr (KIND == 0) {
    if (algo == 1)
      runJob<true, false, kTiersAll>(P, J, smem, arenaSlot, res, outPath);
    else if (algo == 3)
      runJobTA(P, J, smem, arenaSlot, res, outPath);
    else
      runJob<false, false, kTiersAll>(P, J, smem, arenaSlot, res, outPath);
  } else if constexpr (KIND == 1) {  // the A*-epsilon-only kernels: the small window (mrp_ll_lds_bytes(kind = 1))
    if (algo == 1) {
      if (rfl(J.ctx_flags) & kCtxChain) {
        if constexpr (TIERS != kTiersHeavy) runChain(P, J, smem, arenaSlot, res, outPath, pathDst);  // (heavy: stays ST_BAD)
      } else {
        handOver = runJob<true, true, TIERS>(P, J, smem, arenaSlot, res, outPath);
      }
    }
  } else {
    if (algo == 0) runJob<false, false, kTiersAll>(P, J, smem, arenaSlot, res, outPath);
    if (algo == 3) runJobTA(P, J, smem, arenaSlot, res, outPath);
  }
  if (TIERS == kTiersFront && handOver) return true;
  PROF_ADD(res, 5);
#if !defined(MRP_LL_TRACE) && !defined(MRP_CT_PROF)
  res.prof[4] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tj0);  // the whole job on the device (tables, search)
  res.prof[5] = 1;
#endif
  DBG(P, 2, res.status + 100);
  // result + path back to host memory with lane-parallel stores
  __syncthreads();
  resS = res;
  __syncthreads();
  if (lane < sizeof(DevResult) / 4) hostStore32((uint32_t*)resDst + lane, ((const uint32_t*)&resS)[lane]);
  if (res.status == ST_OK && !(KIND == 1 && (rfl(J.ctx_flags) & kCtxChain))) {  // (a root chain wrote its own output)
    const uint32_t words = ((uint32_t)res.n_states + 1) / 2;
    const uint32_t* src = (const uint32_t*)outPath;
    uint32_t* dst = (uint32_t*)pathDst;
    for (uint32_t i = lane; i < words; i += 64) hostStore32(dst + i, src[i]);
    // f2: the path also goes into the device path store (as cells), where the jobs of the conflict-tree nodes that
    // contain it will read it; visible to them because they are published after this job's completion was seen
    const uint32_t sid = rfl(J.store_out_id);
    if (sid < P.path_store_slots && (uint32_t)res.n_states < P.path_store_stride) {
      // (agent-scope stores: through this XCD's L2 to memory, where the readers' agent-scope loads find them — no
      // write-back of the whole L2 is needed to publish them, residentLoop)
      uint16_t* slot = P.path_store + (size_t)sid * P.path_store_stride;
      for (uint32_t i = lane; i < (uint32_t)res.n_states; i += 64)
        __hip_atomic_store(slot + 1 + i, outPath[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // x | y << 8
      __hip_atomic_store(slot, (uint16_t)res.n_states, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  return false;
}

// Batch mode.  One workgroup == one wavefront; pulls jobs from the batch's queue (exit: queue exhausted).
#define MRP_LL_STAGE_PARAMS(P, Parg) const LaunchParams& P = Parg
// The CBS / ECBS kernels declare NO static LDS: their dynamic window then starts at LDS address 0, which is what makes
// every address inside ll_compact.h's window a constant of the ds_ instructions (wave_dev.h windowBase).  The job
// descriptor and the result record they stage through LDS live in the window's control block instead.
static_assert(sizeof(DevJob) + sizeof(DevResult) <= ct::oJob, "control block of the window");
#define MRP_LL_WINDOW_BLOCKS(smem, jobS, resS)                                                        \
  if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint8_t*)smem != 0u) __builtin_trap(); \
  DevJob& jobS = *(DevJob*)(smem + ct::oCtl);                                                         \
  DevResult& resS = *(DevResult*)(smem + ct::oCtl + sizeof(DevJob))
template <int KIND>
DEVI void batchLoop(const LaunchParams& P, uint8_t* smem, DevJob& jobS, DevResult& resS) {
  const uint32_t lane = threadIdx.x;
  uint8_t* arenaSlot = P.arena + (size_t)blockIdx.x * P.arena_stride;
  DBG(P, 0, 1);
  for (;;) {
    // every lane takes part (lane 0 adds 1, the others 0): the kernel deliberately contains no `if (lane == 0)`
    // blocks — hipcc once merged two of them into a wave-divergent wrapper loop that only lane 0 could leave.
    uint32_t j = atomicAdd(P.queue_head, lane == 0 ? 1u : 0u);
    j = rfl(j) - P.queue_base;
    DBG(P, 1, j + 1);
    if (j >= P.n_jobs) break;
    processJob<KIND>(P, P.jobs + j, P.results + j, P.out_paths + (size_t)j * P.out_host_stride, smem, arenaSlot, jobS, resS);
    DBG(P, 3, j + 1);
  }
  DBG(P, 4, 1);
}

extern "C" __global__ void __launch_bounds__(64) mrp_ll_search_kernel(LaunchParams Parg) {  // mixed batches
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  batchLoop<0>(P, smem, jobS, resS);
}
extern "C" __global__ void __launch_bounds__(64) mrp_ll_ecbs_search_kernel(LaunchParams Parg) {  // A*-epsilon jobs only
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  batchLoop<1>(P, smem, jobS, resS);
}
extern "C" __global__ void __launch_bounds__(64) mrp_ll_cbs_search_kernel(LaunchParams Parg) {  // A* jobs only
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  batchLoop<2>(P, smem, jobS, resS);
}

// One SIPP job whose descriptor is at `jobSrc` (host memory): result + raw A* states back to host memory.
// `ldsTier` (sessions): kSippLdsNodes node records + the open list, for jobs on device-resident tables.
constexpr uint32_t kSippLdsNodes = kSippLdsCap;
constexpr uint32_t kSippLdsBytes = kSippLdsBytesC;
DEVI void processSippJob(const LaunchParams& P, const DevJob* jobSrc, DevResult* resDst, uint16_t* pathDst,
                         uint8_t* arenaSlot, uint8_t* ldsTier, DevJob& jobS, DevResult& resS) {
  const uint32_t lane = threadIdx.x;
  __syncthreads();
  {
    const uint32_t* src = (const uint32_t*)jobSrc;
    if (lane < sizeof(DevJob) / 4) ((uint32_t*)&jobS)[lane] = src[lane];
  }
  __syncthreads();
  DevResult res;
  res.status = ST_BAD; res.cost = 0; res.fmin = 0; res.n_states = 0; res.expanded = 0; res.nodes_created = 0;
  res.tier = 1;
  for (int q = 0; q < 8; ++q) res.prof[q] = 0;
  uint16_t* outPath = (uint16_t*)(arenaSlot + P.arena_scratch_off);
  const uint64_t tj0 = __builtin_amdgcn_s_memrealtime();
  if (rfl(jobS.algo) == 2) {  // anything else stays ST_BAD
    if (rfl(jobS.ctx_flags) & kSippResident) {
      if (ldsTier)  // sessions only (else ST_BAD)
        runSipp<true>(P, jobS, arenaSlot, (rfl(jobS.ctx_flags) & kSippNoLds) ? nullptr : ldsTier, kSippLdsNodes, res, outPath);
    } else {
      runSipp<false>(P, jobS, arenaSlot, nullptr, 0, res, outPath);
    }
  }
  res.prof[4] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - tj0);  // the whole of runSipp (table update + search)
  res.prof[5] = 1;
  __syncthreads();
  resS = res;
  __syncthreads();
  if (lane < sizeof(DevResult) / 4) hostStore32((uint32_t*)resDst + lane, ((const uint32_t*)&resS)[lane]);
  if (res.status == ST_OK) {  // one u32 (cell | g << 16) per raw A* state
    const uint32_t* src = (const uint32_t*)outPath;
    uint32_t* dst = (uint32_t*)pathDst;
    for (uint32_t i = lane; i < (uint32_t)res.n_states; i += 64) hostStore32(dst + i, src[i]);
  }
}

// SIPP batches (MRP_LL_SIPP jobs only) run in their own kernels so that the CBS/ECBS kernels' register allocation is
// not widened by a path they never take.  Same queue discipline as mrp_ll_search_kernel.
extern "C" __global__ void __launch_bounds__(64) mrp_ll_sipp_kernel(LaunchParams Parg) {
  __shared__ DevJob jobS;
  __shared__ DevResult resS;
  MRP_LL_STAGE_PARAMS(P, Parg);
  const uint32_t lane = threadIdx.x;
  uint8_t* arenaSlot = P.arena + (size_t)blockIdx.x * P.arena_stride;
  for (;;) {
    uint32_t j = atomicAdd(P.queue_head, lane == 0 ? 1u : 0u);
    j = rfl(j) - P.queue_base;
    if (j >= P.n_jobs) break;
    processSippJob(P, P.jobs + j, P.results + j, P.out_paths + (size_t)j * P.out_host_stride, arenaSlot, nullptr, jobS, resS);
  }
}

// Session mode.  The same workgroups stay resident for a whole solve and are fed through a ticket ring in coherent
// pinned host memory.  An entry is (generation << 11) | job slot; job slots (descriptor, constraint words, path table,
// result) come from a host-side free list, so a slow search holds one slot, not the ring.  A workgroup takes ticket t
// with a device fetch-add and waits until the host has published it.  There is ONE queue and it is first in, first out:
// which search runs next is decided by the HOST, which keeps the queue shallow and publishes in priority order (the
// conflict-tree drivers, csrc/hl/mrp_hl.cpp) — a second, device-side priority ring that every polling wavefront had to
// look at (round 1) cost more in claim traffic than it saved.
// The finished job's slot gets ring_done[slot] = (ticket + 1) & 0x3FFFFFFF | 1 << 30 (never 0) and an entry in the
// completion queue.
// Exit conditions every wave reaches: *ring_stop != 0, or the host's heartbeat word has not moved for
// ring_idle_limit_s seconds (the host is gone).
// A finished job becomes visible to the host: its done word, then its entry in the completion queue.
template <bool SIPP>
DEVI void publishDone(const LaunchParams& P, uint32_t slot, uint32_t doneVal) {
  const uint32_t lane = threadIdx.x;
  const uint32_t compSize = P.n_slots;
#ifdef MRP_LL_SESSION_FENCES
  __threadfence_system();
  __hip_atomic_store(P.ring_done + slot, doneVal, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  // completion queue: the host consumes finished jobs in O(1) each instead of scanning the ring
  uint32_t cidx = atomicAdd(P.comp_count, lane == 0 ? 1u : 0u);
  cidx = rfl(cidx);
  __hip_atomic_store(P.comp_ring + (cidx % compSize), ((cidx / compSize + 1) << kRingSlotBits) | slot, __ATOMIC_RELEASE,
                     __HIP_MEMORY_SCOPE_SYSTEM);
#else
  // Publication without a cache write-back.  What the host reads (result record, path, done word, completion entry)
  // was written with system-scope stores (processJob: hostStore32) that go through the L2 to host memory, and they
  // are complete — in order for the host — once vmcnt says so.  What other workgroups read (the path-store slot) was
  // written with agent-scope stores that go through to memory (processJob).  A system-scope release would add a
  // buffer_wbl2 sc0 sc1: a write-back of EVERY dirty line of this XCD's L2 (the bitmaps, cameFrom bytes, arena nodes and
  // heaps of every search on the XCD), per job.
  // The rule this relies on (LLVM AMDGPU memory model, gfx942 / gfx950): a system-scope (sc0 sc1) store is a write-through
  // store — it is performed at the system coherence point, not held in this XCD's L2 — and `s_waitcnt vmcnt(0)` returns only
  // when every earlier vector-memory operation of the wave, stores included (gfx9 counts them in vmcnt), has been
  // acknowledged there.  Stores of ONE wave that have all been acknowledged before a later store is issued cannot be
  // observed out of order by any agent.  That is the release half of the model's store-release code sequence
  // (`buffer_wbl2 sc0 sc1; s_waitcnt vmcnt(0); store sc0 sc1`) minus the write-back, which exists for data written with
  // weaker scopes — of which the host reads none.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(P.ring_done + slot, doneVal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  // completion queue: the host consumes finished jobs in O(1) each instead of scanning the ring; it looks at the done
  // word of the slot an entry names, so the done word goes first
  uint32_t cidx = atomicAdd(P.comp_count, lane == 0 ? 1u : 0u);
  cidx = rfl(cidx);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __hip_atomic_store(P.comp_ring + (cidx % compSize), ((cidx / compSize + 1) << kRingSlotBits) | slot, __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_SYSTEM);
#endif
}

template <bool SIPP, int KIND, int TIERS = kTiersAll>
DEVI void residentLoop(const LaunchParams& P, uint8_t* smem, DevJob& jobS, DevResult& resS) {
  const uint32_t lane = threadIdx.x;
  uint8_t* arenaSlot = P.arena + (size_t)blockIdx.x * P.arena_stride;
  const uint64_t idleLimit = (uint64_t)P.ring_idle_limit_s * 100000000ull;  // s_memrealtime ticks at 100 MHz
  uint64_t busyTicks = 0, idleTicks = 0;
  const uint32_t q0 = P.ring_size;
  uint32_t* const ring0 = P.ring_state;
  uint32_t* const tickets0 = P.queue_head;        // device counters, 64 bytes apart
  uint32_t* const head0 = P.ring_head;            // host words, 64 bytes apart
  uint32_t* const mirror = P.queue_head + 32;     // device: [0] copy of *head0, [16] copy of *ring_stop (zeroed with the tickets)
  bool haveBulk = false;                          // a bulk ticket is held and not yet served
  uint32_t bulkT = 0;
  uint32_t lastBeat = 0;                          // host heartbeat value seen at the last idle-limit check
  for (;;) {
    uint32_t slot = 0, doneVal = 0;
    bool stop = false;
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t tBeat = t0;
    for (;;) {
      if (!haveBulk) {
        bulkT = rfl(atomicAdd(tickets0, lane == 0 ? 1u : 0u));
        haveBulk = true;
      }
      // Waiting workgroups do NOT poll the host.  The published-ticket count lives in pinned host memory (head0); its
      // device-side MIRROR (tickets0[32], in HBM) is what every waiting workgroup looks at, with agent-scope loads that
      // cost an L2 access.  Exactly one waiting workgroup — the one whose ticket equals the mirror, i.e. the next in line;
      // tickets are consecutive, so while anybody waits there is one — reads the host's word (every ~2 us), copies it
      // into the mirror when it has moved, and copies the host's stop flag likewise.  PCIe reads per engine: one poller
      // instead of one per idle workgroup.  (Round 2 let every idle workgroup poll the host with a back-off; with two
      // engines of 896 workgroups and a host that cannot keep them busy that is ~4e7 uncached reads/s through a handful
      // of L2 channels: measured, the HBM tier ran 14x slower and a 65536-instance step took 6.8 s instead of 1 s.)
      // The others back off in proportion to how far ahead of the mirror their ticket is (next every ~2 us, the k-th
      // every ~2k us, <= ~100 us).
      // The host loads are RELAXED system-scope loads (they go to the host word, not to a cached copy); the ONE acquire
      // sits behind the generation match.  An acquire load per poll is a buffer_inv on this CU's L1 — the cache the
      // ten other searches of the CU are working in — for every look of every idle wavefront (MI355X_MICROARCH.md:
      // "polling with ACQUIRE loads -> correct, 2-3x slower per hop").  -DMRP_LL_RING_POLL_ACQUIRE: the old form (A/B).
#ifdef MRP_LL_RING_POLL_ACQUIRE
      constexpr int kPollOrder = __ATOMIC_ACQUIRE;
#else
      constexpr int kPollOrder = __ATOMIC_RELAXED;
#endif
      uint32_t hd = rfl(__hip_atomic_load(mirror, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      uint32_t sp = 0;
#ifndef MRP_LL_RING_POLL_ALL
      if (hd == bulkT) {  // next in line: look at the host for everybody
#else
      {                   // A/B: every waiting workgroup polls the host (round 2)
#endif
        const uint32_t hh = rfl(__hip_atomic_load(head0, kPollOrder, __HIP_MEMORY_SCOPE_SYSTEM));
        if ((int32_t)(hh - hd) > 0) {
          hd = hh;
#ifndef MRP_LL_RING_POLL_ALL
          __hip_atomic_store(mirror, hh, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // single writer: only T == mirror gets here
#endif
        }
        sp = rfl(__hip_atomic_load(P.ring_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (sp != 0) __hip_atomic_store(mirror + 16, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      if ((int32_t)(hd - bulkT) > 0) {
        const uint32_t gen = (bulkT / q0 + 1) & 0x1FFFFFu;
        const uint32_t e = rfl(__hip_atomic_load(ring0 + bulkT % q0, kPollOrder, __HIP_MEMORY_SCOPE_SYSTEM));
        if ((e >> kRingSlotBits) == gen) {
          // NO acquire fence here.  What the host wrote before publishing (descriptor, constraint words, ids, tables) is
          // read with system-scope loads that go past the caches (hostLoad32: processJob, runJob).  A system-scope
          // acquire FENCE is a buffer_inv sc0 sc1 — it throws away every clean line of this XCD's L2, i.e. the bitmaps,
          // nodes and heaps of the 380 other searches that share it — and round 3 measured what 4e6 of them per second
          // (with the write-backs below) cost: 2.9 instead of 2.5 us per expansion in the compact tier and 11.7
          // instead of 5.1 us in the arena tier on a full chip.  -DMRP_LL_SESSION_FENCES restores them (A/B).
#if defined(MRP_LL_SESSION_FENCES) && !defined(MRP_LL_RING_POLL_ACQUIRE)
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // system scope: the job data the host wrote before publishing
#elif !defined(MRP_LL_RING_POLL_ACQUIRE)
          if (SIPP) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");  // (the SIPP kernels read their tables with plain loads)
#endif
          slot = e & kRingSlotMask;
          doneVal = ((bulkT + 1u) & 0x3FFFFFFFu) | 0x40000000u;
          haveBulk = false;
          break;
        }
      }
      if (sp == 0) sp = rfl(__hip_atomic_load(mirror + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (sp != 0) {
        stop = true;
        break;
      }
      // Safety net for a host that died: the host bumps a heartbeat word on every submit / poll.  A workgroup — which
      // always holds a bulk ticket while it waits — leaves only when that word has not moved for ring_idle_limit_s; a
      // live host that is merely slow to publish (a long tail search elsewhere, a caller pausing between submits while
      // it keeps polling) never loses the workgroup that holds the ticket it will publish next.
      if (__builtin_amdgcn_s_memrealtime() - tBeat > idleLimit) {
        const uint32_t hb = rfl(__hip_atomic_load(P.ring_head + kHeartbeatWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (hb == lastBeat) {
          stop = true;
          break;
        }
        lastBeat = hb;
        tBeat = __builtin_amdgcn_s_memrealtime();
      }
      uint32_t naps = (int32_t)(bulkT - hd) > 0 ? 1 + (bulkT - hd) : 1;
      if (naps > 48) naps = 48;
      for (uint32_t q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(64);
    }
    const uint64_t t1c = __builtin_amdgcn_s_memrealtime();
    idleTicks += t1c - t0;
    if (stop) break;
    bool handOver = false;
    if constexpr (SIPP)
      processSippJob(P, P.jobs + slot, P.results + slot, P.out_paths + (size_t)slot * P.out_host_stride, arenaSlot, smem, jobS,
                     resS);
    else
      handOver = processJob<KIND, TIERS>(P, P.jobs + slot, P.results + slot, P.out_paths + (size_t)slot * P.out_host_stride, smem,
                                         arenaSlot, jobS, resS);
    if (TIERS == kTiersFront && handOver) {
      // The search does not fit the narrow tier: the heavy workgroups take it over (ll_device.h heavy_q).  One 8-byte store
      // carries everything they need — the job slot (whose descriptor still sits in host memory) and the done value.
      const uint32_t ht = rfl(atomicAdd(P.heavy_ctr, lane == 0 ? 1u : 0u));
      const unsigned long long he = ((unsigned long long)doneVal << 32) | (unsigned long long)(((ht / kRingSlots + 1u) << kRingSlotBits) | slot);
      __hip_atomic_store(P.heavy_q + (ht % kRingSlots), he, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      publishDone<SIPP>(P, slot, doneVal);
    }
    busyTicks += __builtin_amdgcn_s_memrealtime() - t1c;
  }
  atomicAdd(P.sess_ticks + 0, lane == 0 ? (unsigned long long)busyTicks : 0ull);
  atomicAdd(P.sess_ticks + 1, lane == 0 ? (unsigned long long)idleTicks : 0ull);
  atomicAdd(P.sess_ticks + 2, (lane == 0 && busyTicks != 0) ? 1ull : 0ull);  // workgroups that ran at least one job
}

extern "C" __global__ void __launch_bounds__(64) mrp_ll_persistent_kernel(LaunchParams Parg) {  // mixed sessions
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  residentLoop<false, 0>(P, smem, jobS, resS);
}
extern "C" __global__ void __launch_bounds__(64) mrp_ll_ecbs_persistent_kernel(LaunchParams Parg) {  // A*-epsilon only
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  residentLoop<false, 1>(P, smem, jobS, resS);
}
// The front / heavy pair of an A*-epsilon session (ll_device.h heavy_q).  Front workgroups are the resident loop above
// with the narrow compact tier alone — no arena-tier code, hence a smaller register allocation; heavy workgroups wait on
// the device-side queue the front ones fill, run each search in the WIDE compact geometry (31.4 KB of LDS: open and focal
// lists of 3071 entries, walk queue) and, beyond even that, in the arena tier.
// (MRP_LL_FRONT_WAVES4: at most 128 VGPRs = four waves per SIMD; the searches' own function needs 118, what is spilled is
// state of the job set-up and of the chain's conflict scan)
#ifdef MRP_LL_FRONT_WAVES4
#define MRP_LL_FRONT_ATTR __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define MRP_LL_FRONT_ATTR
#endif
extern "C" __global__ void __launch_bounds__(64) MRP_LL_FRONT_ATTR mrp_ll_ecbs_front_kernel(LaunchParams Parg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  // "the front launch runs" (the host turns a launch that never starts — streams sharing a hardware queue with a resident
  // kernel — into an error instead of a hang): the last alive word
  if (blockIdx.x == 0) hostStore32(P.heavy_alive + (kRingSlots - 1u), 1u);
  residentLoop<false, 1, kTiersFront>(P, smem, jobS, resS);
}

// Exit conditions every wave reaches: the stop flag (its device mirror, written by the front workgroup that polls the
// host, or the host word itself, looked at every ~0.5 ms), or a host heartbeat that stood still for ring_idle_limit_s.
extern "C" __global__ void __launch_bounds__(64) mrp_ll_ecbs_heavy_kernel(LaunchParams Parg) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  const uint32_t lane = threadIdx.x;
  uint8_t* arenaSlot = P.arena + (size_t)blockIdx.x * P.arena_stride;
  uint32_t* const mirror = P.queue_head + 32;
  const uint64_t idleLimit = (uint64_t)P.ring_idle_limit_s * 100000000ull;
  uint64_t busyTicks = 0, idleTicks = 0;
  hostStore32(P.heavy_alive + blockIdx.x, 1u);  // "this workgroup is resident" (the host checks before it relies on us)
  uint32_t lastBeat = 0;
  for (;;) {
    const uint64_t t0 = __builtin_amdgcn_s_memrealtime();
    uint64_t tBeat = t0;
    const uint32_t ht = rfl(atomicAdd(P.heavy_ctr + 16, lane == 0 ? 1u : 0u));
    const uint32_t gen = ht / kRingSlots + 1u;
    unsigned long long e = 0;
    bool stop = false;
    uint32_t naps = 1, looks = 0;
    for (;;) {
      e = rfl64(__hip_atomic_load(P.heavy_q + (ht % kRingSlots), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if ((((uint32_t)e) >> kRingSlotBits) == gen) break;
      uint32_t sp = rfl(__hip_atomic_load(mirror + 16, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      if (sp == 0 && (++looks & 31u) == 0u) sp = rfl(__hip_atomic_load(P.ring_stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
      if (sp != 0) {
        stop = true;
        break;
      }
      if (__builtin_amdgcn_s_memrealtime() - tBeat > idleLimit) {
        const uint32_t hb = rfl(__hip_atomic_load(P.ring_head + kHeartbeatWord, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM));
        if (hb == lastBeat) {
          stop = true;
          break;
        }
        lastBeat = hb;
        tBeat = __builtin_amdgcn_s_memrealtime();
      }
      for (uint32_t q = 0; q < naps; ++q) __builtin_amdgcn_s_sleep(64);  // ~2 us, doubling to ~16 us
      if (naps < 8) naps *= 2;
    }
    const uint64_t t1c = __builtin_amdgcn_s_memrealtime();
    idleTicks += t1c - t0;
    if (stop) break;
    const uint32_t slot = (uint32_t)e & kRingSlotMask, doneVal = (uint32_t)(e >> 32);
    (void)processJob<1, kTiersHeavy>(P, P.jobs + slot, P.results + slot, P.out_paths + (size_t)slot * P.out_host_stride, smem,
                                     arenaSlot, jobS, resS);
    publishDone<false>(P, slot, doneVal);
    busyTicks += __builtin_amdgcn_s_memrealtime() - t1c;
  }
  atomicAdd(P.sess_ticks + 4, lane == 0 ? (unsigned long long)busyTicks : 0ull);
  atomicAdd(P.sess_ticks + 5, lane == 0 ? (unsigned long long)idleTicks : 0ull);
  atomicAdd(P.sess_ticks + 6, (lane == 0 && busyTicks != 0) ? 1ull : 0ull);
}

extern "C" __global__ void __launch_bounds__(64) mrp_ll_cbs_persistent_kernel(LaunchParams Parg) {  // A* only
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  MRP_LL_WINDOW_BLOCKS(smem, jobS, resS);
  const LaunchParams& P = Parg;
  residentLoop<false, 2>(P, smem, jobS, resS);
}

// The same resident loop for SIPP sessions (jobs of algo MRP_LL_SIPP only).  24.6 KB of LDS per workgroup hold up to 2047
// nodes and the open list of a search on a device-resident table (6 workgroups per CU).
extern "C" __global__ void __launch_bounds__(64) mrp_ll_sipp_persistent_kernel(LaunchParams Parg) {
  __shared__ __attribute__((aligned(16))) uint8_t sippTier[kSippLdsBytes];
  __shared__ DevJob jobS;
  __shared__ DevResult resS;
  MRP_LL_STAGE_PARAMS(P, Parg);
  residentLoop<true, 0>(P, sippTier, jobS, resS);
}

// hipFuncAttributeMaxDynamicSharedMemorySize (a workgroup may take up to the CU's 160 KiB minus the static jobS/resS) for
// kernel number `which` on the calling thread's current device; thread-safe, done once per (kernel, device).
static hipError_t allowFullLds(const void* fn, int which) {
  static std::mutex mu;
  static bool done[8][64] = {};
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lock(mu);
  if (dev >= 0 && dev < 64 && done[which][dev]) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
  if (e == hipSuccess && dev >= 0 && dev < 64) done[which][dev] = true;
  return e;
}

}  // namespace mrp

// ---- host-callable launcher (used by mrp_ll_host.cpp) -----------------------------------------------------------
extern "C" uint32_t mrp_ll_lds_bytes(int kind, uint32_t capNodes, uint32_t rows, uint32_t rowWords, uint32_t pathBytes) {
  (void)rows; (void)rowWords;
  // without the compact tier a workgroup still stages its job descriptor and result through the window's control block
  return capNodes ? mrp::ldsBytes(pathBytes, kind == 1) : mrp::ct::oJob;
}

// kind: 0 = mixed, 1 = A*-epsilon jobs only, 2 = A* jobs only (see processJob)
extern "C" hipError_t mrp_ll_launch(const mrp::LaunchParams* P, uint32_t grid, uint32_t ldsBytes, int kind,
                                    hipStream_t stream) {
  typedef void (*Kern)(mrp::LaunchParams);
  const Kern k = kind == 1 ? mrp::mrp_ll_ecbs_search_kernel : kind == 2 ? mrp::mrp_ll_cbs_search_kernel : mrp::mrp_ll_search_kernel;
  {  // every worker thread launches through here, and the attribute is per device: set it under a lock, once per device
    hipError_t e = mrp::allowFullLds(reinterpret_cast<const void*>(k), kind);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), ldsBytes, stream, *P);
  return hipGetLastError();
}

// Resident workgroups per CU the runtime reports for the persistent kernel of `kind` with `ldsBytes` of dynamic LDS
// (0 on error): what mrp_ll_configure_tiers tells its caller, who sizes a session with it.
extern "C" int mrp_ll_persistent_occupancy(int kind, uint32_t ldsBytes) {
  typedef void (*Kern)(mrp::LaunchParams);
  const Kern k = kind == 1   ? mrp::mrp_ll_ecbs_persistent_kernel
                 : kind == 2 ? mrp::mrp_ll_cbs_persistent_kernel
                             : mrp::mrp_ll_persistent_kernel;
  if (mrp::allowFullLds(reinterpret_cast<const void*>(k), 3 + kind) != hipSuccess) return 0;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), 64, ldsBytes) != hipSuccess) return 0;
  return n;
}

// ... and for the resident SIPP kernel (its LDS is static)
extern "C" int mrp_ll_sipp_persistent_occupancy(void) {
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(mrp::mrp_ll_sipp_persistent_kernel), 64, 0) !=
      hipSuccess)
    return 0;
  return n;
}

// The front / heavy pair (kind 1 sessions with heavy workgroups): `heavy` = false launches the front workgroups with the
// narrow window of `ldsBytes`, true the heavy ones with the wide window.
extern "C" uint32_t mrp_ll_heavy_lds_bytes(void) { return mrp::ct::Wide::windowBytes(true); }
extern "C" hipError_t mrp_ll_launch_front_heavy(const mrp::LaunchParams* P, uint32_t grid, uint32_t ldsBytes, int heavy,
                                                hipStream_t stream) {
  typedef void (*Kern)(mrp::LaunchParams);
  const Kern k = heavy ? mrp::mrp_ll_ecbs_heavy_kernel : mrp::mrp_ll_ecbs_front_kernel;
  {
    hipError_t e = mrp::allowFullLds(reinterpret_cast<const void*>(k), 6 + (heavy ? 1 : 0));
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), heavy ? mrp::ct::Wide::windowBytes(true) : ldsBytes, stream, *P);
  return hipGetLastError();
}
// Resident workgroups per CU of the front kernel (`heavy` = 0) / the heavy kernel alone (0 on error).
extern "C" int mrp_ll_front_heavy_occupancy(int heavy, uint32_t ldsBytes) {
  typedef void (*Kern)(mrp::LaunchParams);
  const Kern k = heavy ? mrp::mrp_ll_ecbs_heavy_kernel : mrp::mrp_ll_ecbs_front_kernel;
  if (mrp::allowFullLds(reinterpret_cast<const void*>(k), 6 + (heavy ? 1 : 0)) != hipSuccess) return 0;
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, reinterpret_cast<const void*>(k), 64,
                                                   heavy ? mrp::ct::Wide::windowBytes(true) : ldsBytes) != hipSuccess)
    return 0;
  return n;
}

extern "C" hipError_t mrp_ll_launch_sipp(const mrp::LaunchParams* P, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(mrp::mrp_ll_sipp_kernel, dim3(grid), dim3(64), 0, stream, *P);
  return hipGetLastError();
}

extern "C" hipError_t mrp_ll_launch_sipp_persistent(const mrp::LaunchParams* P, uint32_t grid, hipStream_t stream) {
  hipLaunchKernelGGL(mrp::mrp_ll_sipp_persistent_kernel, dim3(grid), dim3(64), 0, stream, *P);
  return hipGetLastError();
}

extern "C" hipError_t mrp_ll_launch_persistent(const mrp::LaunchParams* P, uint32_t grid, uint32_t ldsBytes, int kind,
                                               hipStream_t stream) {
  typedef void (*Kern)(mrp::LaunchParams);
  const Kern k = kind == 1   ? mrp::mrp_ll_ecbs_persistent_kernel
                 : kind == 2 ? mrp::mrp_ll_cbs_persistent_kernel
                             : mrp::mrp_ll_persistent_kernel;
  {
    hipError_t e = mrp::allowFullLds(reinterpret_cast<const void*>(k), 3 + kind);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(k, dim3(grid), dim3(64), ldsBytes, stream, *P);
  return hipGetLastError();
}
