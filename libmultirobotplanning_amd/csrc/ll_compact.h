// The compact LDS tier of the CBS / ECBS low-level searches, as a "wave program" (wave_dev.h on gfx950; the CPU test-suite
// compiles the same source against tests/support/wave_emu.h and replays harvested searches against the oracle).
//
// Reference semantics (file:line in /root/reference): AStarEpsilon::search a_star_epsilon.hpp:86-285, AStar::search
// a_star.hpp:63-161, the grid Environment example/ecbs.cpp:264-312,352-399,497-510 (example/cbs.cpp minus the focal
// parts), heap rules boost::heap::d_ary_heap<arity<2>, mutable_<true>> as restated in oracle/heap_restated.hpp.
//
// One wavefront runs one search; the heaps are replayed exactly, but what an expansion costs is cut to the bone:
//   * A state IS its heap entry.  g == time (every action costs 1, ecbs.cpp:369-396) and a state is discovered at most
//     once (the decrease-key branch a_star_epsilon.hpp:254-269 is dead for this Environment), so the 32-bit entry
//       [31:23] 511 - focalH   [22:16] 127 - f   [15:10] g   [9:0] cell = y * 32 + x
//     names the node completely: there are no node records, no node ids and no handle table.  A larger entry (under the
//     list's key mask) is a BETTER node in the reference's orders (open a_star_epsilon.hpp:312-323: f asc, g desc; focal
//     :346-366: focalH, f asc, g desc); equal keys compare equal, the layout decides, and the layout is replayed verbatim.
//   * openSet.erase(handle) (a_star_epsilon.hpp:216) needs the position of the popped node in the open array: one
//     ds_read_b128 per lane looks at 256 entries at once.  No position array, no stores to keep one current.
//   * The open list lives in lanes 0..31 and the focal list in lanes 32..63 of every heap step: the pop of one and the
//     erase from the other (sift-downs, five levels per round trip each), and the two pushes of a successor (sift-ups),
//     are one instruction stream.
//   * Every heap slot that holds no element holds kEmpty, whose key is smaller than any real key, and the word in front
//     of element 0 holds the largest key: a sift-down that looks at the children of a leaf, or past the end of the array,
//     and a sift-up lane beyond the root, read a value that makes them do nothing — no bounds tests, no masked loads.
//   * cameFrom (a_star_epsilon.hpp:275-279) is one byte (the action) per discovered state in a (time, cell) table in the
//     search's arena slot, written and forgotten; the goal branch reads it back eight time steps per round trip.
//   * The (time, cell) bitmap — obstacles | vertex constraints | discovered — answers stateValid (ecbs.cpp:497-503),
//     closedSet.find and stateToHeap.find (a_star_epsilon.hpp:224-227) with one bit; all rows are set up at job start,
//     eight rows per instruction.
//   * The job arrives, and the result leaves, through a block of the LDS window; compactSearch is a real function (not
//     inlined into the kernels), so nothing of the kernels' own state occupies scalar registers while a search runs.
// Limits of the tier (a search that would exceed one returns C_OVERFLOW before touching anything of that expansion and is
// run again by the arena tier, ll_kernel.hip): maps up to 32 x 32, 1023 open entries, t <= 61 for an expanded node,
// focalH <= 511, at most 64 edge constraints, at most 128 agents in the focal context.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace mrp {
namespace ct {

using namespace wv;

// ---- LDS window of one search (byte offsets) ------------------------------------------------------------------
// A heap area: element i at byte 4 * (i + 1), so that the children (2i + 1, 2i + 2) of any node are one aligned 8-byte
// pair and lane L's 16-byte group of 256-entry group g holds elements 256 g + 4L - 1 .. 256 g + 4L + 2.
//
// Two geometries of the same program (template parameter C of compactSearch):
//   Narrow — the tier every search starts in: 1023 open entries, t <= 61, 12.9 KB of LDS with ten agents (12 searches per CU);
//            entry  [31:23] 511 - focalH   [22:16] 127 - f   [15:10] g   [9:0] cell
//   Wide   — the tier of the searches that outgrow it (the "heavy" resident workgroups, ll_kernel.hip): 3071 open entries,
//            31.4 KB, and as many time steps as the job's arena slot has room for (CJob::rows, up to 958): its entry
//            carries h instead of g —  [31:26] 63 - focalH   [25:16] 1023 - f   [15:10] 63 - h   [9:0] cell  — which orders
//            the same way (equal f: g desc <=> h asc, f = g + h) and lets f span 10 bits; g = f - h, a state is named by
//            (f, cell).  Its (time, cell) bitmap lives in device memory and is built 64 rows at a time, as t grows.
constexpr uint32_t kRowBytes = 128;               // (time, cell) bitmap: 32 words (one per y) of 32 bits (x) per time step
constexpr uint32_t oCtl = 0;                      // 256 bytes for the kernel that hosts the tier (job descriptor, result)
constexpr uint32_t oJob = 256;                    // CJob
constexpr uint32_t oRes = 384;                    // CRes (+ eight profile words in the diagnostic build)
constexpr uint32_t oOpen = 448;
constexpr uint32_t kFront = 0xFFFFFFFFu;          // the word in front of element 0: above every key, names no state
// L_BITS: width of the field below f — g (LONGT = false) or 63 - h (LONGT = true, 6 bits)
template <uint32_t GROUPS, uint32_t FH_BITS, uint32_t F_BITS, uint32_t L_BITS, bool LONGT = false>
struct TierCfg {
  static constexpr bool kLongT = LONGT;
  static constexpr uint32_t kGroups = GROUPS;                  // 256-entry groups of the open / focal arrays (one scan instruction each)
  static constexpr uint32_t kCap = 256u * GROUPS - 1u;         // entries of the open / focal list
  static constexpr uint32_t kHeapBytes = 1024u * GROUPS + 16u;
  static constexpr uint32_t kHeapClamp = kCap + 2u;            // odd; elements kCap .. kCap + 3 exist and always hold kEmpty
  static constexpr uint32_t kAuxCap = 128u * GROUPS + 2u;      // walk queue of the ordered walk: at most (n + 1) / 2 + 1 entries
  static constexpr uint32_t kAuxBytes = 512u * GROUPS + 32u;
  static constexpr uint32_t kAuxClamp = kAuxCap + 1u;          // odd; elements kAuxCap .. kAuxCap + 4 always hold kEmpty
  // time steps of the (time, cell) bitmap and of the cameFrom table; LONGT: of one chunk of rows (the job says how many rows)
  static constexpr uint32_t kRows = LONGT ? (kAuxBytes >= 64u * 128u ? 64u : 32u) : (1u << L_BITS);
  static constexpr uint32_t kFShift = 10u + L_BITS, kFhShift = 10u + L_BITS + F_BITS;
  static constexpr uint32_t kGMax = (1u << L_BITS) - 1u, kFMax = (1u << F_BITS) - 1u, kFhMax = (1u << FH_BITS) - 1u;
  static constexpr uint32_t kMO = ((1u << (F_BITS + L_BITS)) - 1u) << 10;  // open key:  f, g (or h)
  static constexpr uint32_t kMF = 0xFFFFFC00u;                             // focal key: focalH, f, g (or h)
  // what names a state: g (== time) and cell; LONGT: f and cell
  static constexpr uint32_t kStateMask = LONGT ? ((kFMax << kFShift) | 1023u) : ((1u << kFShift) - 1u);
  // "no element": its key is below every real key (f <= kFMax - 3 in a tier) and its state bits never name a state
  // (g = kGMax is never pushed: kMaxT; LONGT: nor is f = kFMax)
  static constexpr uint32_t kEmpty = (1u << kFShift) - 1u;
  // the last time step whose nodes are expanded here (LONGT: as far as f = g + h <= kFMax - 3 reaches; the job's rows cut it)
  static constexpr uint32_t kMaxT = LONGT ? kFMax - 3u - 63u : kGMax - 2u;
  // walk-queue entries: the open key of an element above its index in the open array (10 bits, 12 beyond 1023 entries)
  static constexpr uint32_t kAuxShift = GROUPS > 4u ? 2u : 0u;
  static constexpr uint32_t kAuxIdxMask = (1u << (10u + kAuxShift)) - 1u;
  static constexpr uint32_t kMOAux = kMO << kAuxShift;
  static constexpr bool kExactFhCheck = FH_BITS < 9u;          // Narrow: two conflicts per agent always fit or the job leaves first
  static constexpr uint32_t oFocal = oOpen + kHeapBytes;
  static constexpr uint32_t oAux = oFocal + kHeapBytes;
  static constexpr uint32_t oBits = oAux + kAuxBytes;
  static constexpr uint32_t oObst = oBits + kRows * kRowBytes;
  static constexpr uint32_t oPaths = oObst + kRowBytes;        // the focal path table follows (size chosen by the launcher)
  static constexpr uint32_t kBitsBytes = kRows * kRowBytes;
  static constexpr uint32_t kParentBytes = kRows * 1024u;      // cameFrom table in the arena slot
  // bytes of device memory a search of `rows` time steps needs behind CJob::parentTab / CJob::bitsG
  static constexpr uint32_t parentBytes(uint32_t rows) { return (LONGT ? rows : kRows) * 1024u; }
  static constexpr uint32_t bitsBytes(uint32_t rows) { return (LONGT ? rows : kRows) * kRowBytes; }
  // BG ("bitmap in global memory", the A*-epsilon-only kernels): the (time, cell) bitmap lives in the search's arena slot
  // (CJob::bitsG) instead of the window, which then ends right behind the walk queue: obstacle row, path table.
  static constexpr uint32_t obstOff(bool bg) { return bg ? oBits : oObst; }
  static constexpr uint32_t pathsOff(bool bg) { return obstOff(bg) + kRowBytes; }
  static constexpr uint32_t windowBytes(bool bg) { return pathsOff(bg); }
  static_assert(kFhShift + FH_BITS == 32u, "an entry is 32 bits");
  static_assert(!LONGT || L_BITS == 6u, "h <= 62 on a 32 x 32 map");
  // BG (not LONGT) puts the bitmap together in the not yet initialised open + focal areas, kBuildRows rows at a time, and
  // the goal branch stages kStageRows 1 KB rows of the cameFrom table in the areas in front of the obstacle row
  static constexpr uint32_t kBuildRows = LONGT ? kRows : (2u * kHeapBytes >= kRows * kRowBytes ? kRows : kRows / 2u);
  static constexpr uint32_t kStageRows = (2u * kHeapBytes + kAuxBytes) / 1024u >= 8u ? 8u : (2u * kHeapBytes + kAuxBytes) / 1024u;
  static_assert(2u * kHeapBytes >= kBuildRows * kRowBytes && kRows % kBuildRows == 0u, "room for a chunk of bitmap rows");
  static_assert(kStageRows >= 4u, "room for the goal branch's rows");
  static_assert(!LONGT || kAuxBytes >= kBitsBytes, "LONGT builds a chunk of bitmap rows in the walk queue's area");
  static_assert(oFocal % 16 == 0 && oAux % 16 == 0 && oBits % 16 == 0 && oObst % 16 == 0 && oPaths % 16 == 0, "16-byte aligned areas");
  static_assert((kHeapClamp & 1u) == 1u && 4u * (kHeapClamp + 3u) <= kHeapBytes, "clamped child pair stays inside the heap area");
  static_assert((kAuxClamp & 1u) == 1u && 4u * (kAuxClamp + 3u) <= kAuxBytes, "clamped child pair stays inside the walk queue");
  static_assert(256u * GROUPS <= kAuxIdxMask + 1u && 10u + F_BITS + L_BITS + kAuxShift <= 32u, "walk-queue entry: key above index");
  static_assert(kMaxT + 1u + 62u <= kFMax - 3u, "f = g + h (h <= 62 on a 32 x 32 map) fits its field");
};
// (MRP_CT_NARROW_GROUPS: 4 = 1023 open entries, 12.9 KB with ten agents; 3 = 767 entries, 10.1 KB)
#ifndef MRP_CT_NARROW_GROUPS
#define MRP_CT_NARROW_GROUPS 4
#endif
typedef TierCfg<MRP_CT_NARROW_GROUPS, 9, 7, 6> Narrow;
// 12 groups = 3071 open entries in a 31.4 KB window.  (Measured on MI355X, scripts/r4_run10.sh / r4_run11.sh: a workgroup
// with more than 32 KB of LDS takes room on its CU as if it had 64 KB — 16 groups, 41.6 KB, cost 5.2 narrow windows each
// instead of 3 — so the window stays below; the largest open list seen on the benchmark shapes is 2787 entries.)
#ifndef MRP_CT_WIDE_GROUPS
#define MRP_CT_WIDE_GROUPS 12
#endif
typedef TierCfg<MRP_CT_WIDE_GROUPS, 6, 10, 6, true> Wide;
// (names the hosting kernels and the host tests use: the narrow geometry)
constexpr uint32_t kGroups = Narrow::kGroups, kCap = Narrow::kCap, kHeapBytes = Narrow::kHeapBytes, kAuxBytes = Narrow::kAuxBytes;
constexpr uint32_t kRows = Narrow::kRows;
constexpr uint32_t oFocal = Narrow::oFocal, oAux = Narrow::oAux, oBits = Narrow::oBits, oObst = Narrow::oObst, oPaths = Narrow::oPaths;
constexpr uint32_t kLdsBytes = oPaths;
constexpr uint32_t obstOff(bool bg) { return Narrow::obstOff(bg); }
constexpr uint32_t pathsOff(bool bg) { return Narrow::pathsOff(bg); }
constexpr uint32_t windowBytes(bool bg) { return Narrow::windowBytes(bg); }
constexpr uint32_t kBitsBytes = Narrow::kBitsBytes, kParentBytes = Narrow::kParentBytes;
constexpr uint32_t kMO = Narrow::kMO, kMF = Narrow::kMF, kEmpty = Narrow::kEmpty, kMaxT = Narrow::kMaxT;
constexpr uint32_t kHeapClamp = Narrow::kHeapClamp;
static_assert(oOpen % 16 == 0, "LDS areas are 16-byte aligned");

enum : int32_t { C_OK = 0, C_NO_SOLUTION = 1, C_CAP_EXP = 2, C_OVERFLOW = -1 };

struct CJob {              // at oJob of the window; pointers as two words
  uint32_t dimx, dimy, sx, sy, gx, gy;
  int32_t lastGoal;        // Environment::m_lastGoalConstraint (ecbs.cpp:268-273)
  float w;                 // AStarEpsilon::m_w, binary32 (a_star_epsilon.hpp:386)
  uint32_t nVc, nEc;       // nEc <= 64
  uint32_t obstWords;
  uint32_t nAgentsPad, tPad;  // focal context: table [tPad][nAgentsPad] of x | y << 8 halfwords (0xFFFF = nobody); <= 128 agents
  uint32_t maxExp;         // 0xFFFFFFFF = unlimited
  uint32_t openCap, maxT;  // limits of this job inside the tier: <= kCap open entries, expanded nodes at t <= maxT <= kMaxT
  uint32_t taNoGoal;       // compactSearchTA: the agent has no task (cbs_ta.cpp:283-319)
  uint32_t rows;           // Wide geometry: time steps the device memory behind parentTab / bitsG has room for (maxT <= rows - 2)
  uint64_t vc;             // const uint32_t*: t << 16 | y << 8 | x
  uint64_t ec;             // const uint32_t*: t << 19 | (y * dimx + x) << 3 | k   (k = index in Wait, Left, Right, Up, Down)
  uint64_t obst;           // const uint32_t*: the map's obstacle bitmap, bit y * dimx + x
  uint64_t pathsG;         // const uint16_t*: the path table when it is not in the window (PLDS = false);
                           // compactSearchTA: the shortest-path table of the task's cell, [y * 32 + x] halfwords
  uint64_t parentTab;      // uint8_t*: kParentBytes of device memory, action byte per (t, cell)
  uint64_t outPath;        // uint16_t*: x | y << 8 per time step
  uint64_t bitsG;          // uint32_t*: BG instances: kBitsBytes of device memory for the (time, cell) bitmap
};
static_assert(sizeof(CJob) <= oRes - oJob && sizeof(CJob) % 4 == 0, "CJob fits its block of the window");
struct CRes {              // at oRes
  int32_t status, cost, fmin, nStates;
  uint32_t expanded, nodes;
};
// -DMRP_CT_PROF (diagnostic build): shader cycles per phase of the search loop, in eight words behind the CRes:
// 0 loop top + goal test, 1 ordered walk, 2 successor probes / row loads issued, 3 pop + erase, 4 successor entries,
// 5 pushes, 6 nodes visited by walks, 7 walks skipped (no open node in the band); set-up counts as loop top
#ifdef MRP_CT_PROF
#define MRP_CT_PROF_DECL uint32_t prof_[8] = {0, 0, 0, 0, 0, 0, 0, 0}; uint64_t profT_ = wv::clock64()
#define MRP_CT_PROF_MARK(k) do { const uint64_t n_ = wv::clock64(); prof_[k] += (uint32_t)(n_ - profT_); profT_ = n_; } while (0)
#define MRP_CT_PROF_ADD(k, v) prof_[k] += (v)
#define MRP_CT_PROF_STORE(lds) do { for (uint32_t q_ = 0; q_ < 8; ++q_) ldsStoreS(lds, oRes + 32u + 4u * q_, prof_[q_]); } while (0)
#elif defined(MRP_CT_ASM_MARKS)  // dev tool: "; PHASE k" comments in the compiler's assembly (instruction counts per phase)
#define MRP_CT_PROF_DECL do { } while (0)
#define MRP_CT_PROF_MARK(k) asm volatile("; PHASE " #k ::: "memory")
#define MRP_CT_PROF_ADD(k, v) do { } while (0)
#define MRP_CT_PROF_STORE(lds) do { } while (0)
#else
#define MRP_CT_PROF_DECL do { } while (0)
#define MRP_CT_PROF_MARK(k) do { } while (0)
#define MRP_CT_PROF_ADD(k, v) do { } while (0)
#define MRP_CT_PROF_STORE(lds) do { } while (0)
#endif
static_assert(sizeof(CRes) <= 32 && oRes + 64 <= oOpen, "CRes fits its block of the window");

#define MRP_CT_JOB_U32(lds, field) ldsLoadS(lds, oJob + (uint32_t)offsetof(CJob, field))
template <class T>
WV_FN T* jobPtr(Lds lds, uint32_t off) {
  const uint64_t lo = ldsLoadS(lds, oJob + off), hi = ldsLoadS(lds, oJob + off + 4u);
  return (T*)(uintptr_t)(lo | (hi << 32));
}
#define MRP_CT_JOB_PTR(T, lds, field) jobPtr<T>(lds, (uint32_t)offsetof(CJob, field))

WV_FN uint32_t popc64(uint64_t v) { return (uint32_t)__builtin_popcountll(v); }
WV_FN uint32_t ctz64(uint64_t v) { return (uint32_t)__builtin_ctzll(v); }
WV_FN uint32_t ctz32(uint32_t v) { return (uint32_t)__builtin_ctz(v); }
WV_FN uint32_t lg2(uint32_t v) { return 31u - (uint32_t)__builtin_clz(v); }  // v != 0
WV_FN uint32_t lo32(uint64_t v) { return (uint32_t)v; }
WV_FN uint32_t hi32(uint64_t v) { return (uint32_t)(v >> 32); }

// Per-lane constants of the two-heaps-in-one-wave steps: lanes 0..31 ("side A") work on one heap, lanes 32..63 ("side B")
// on another.  Inside a side, lane l5 owns node l5 of a 31-node (five-level) subtree below the sift-down's hole, and
// ancestor l5 of a sift-up chain.
struct Sides {
  V lane, l5;
  B isB;
  V lvl, off1;     // level of node l5 inside the subtree; position inside that level, minus one (lane 31: far outside)
  V anc, needR;    // bit a set <=> node a is an ancestor of node l5 / ... and the way to l5 leaves a through its RIGHT child
};
WV_FN Sides makeSides() {
  Sides s;
  s.lane = laneId();
  s.l5 = s.lane & 31u;
  s.isB = (s.lane >> 5) != 0u;
  const V n = s.l5 + 1u;
  s.lvl = 31u - clz(n);
  s.off1 = sel(s.l5 == 31u, splat(0x100000u), n - (splat(1u) << s.lvl) - 1u);
  s.anc = splat(0u);
  s.needR = splat(0u);
  for (uint32_t k = 1; k <= 4; ++k) {
    const V a = n >> k;  // 1-based number of the k-th ancestor (0: none)
    const B has = (a != 0u) & (n < 32u);
    s.anc |= sel(has, splat(1u) << (a - 1u), splat(0u));
    s.needR |= sel(has, ((n >> (k - 1u)) & 1u) << (a - 1u), splat(0u));
  }
  return s;
}
WV_FN V bothSides(const Sides& S, uint32_t a, uint32_t b) { return sel(S.isB, splat(b), splat(a)); }  // side A: a, side B: b

// Sift-downs of two heaps at once, five levels per LDS round trip.  Each side moves a hole, starting at its root, down
// its heap (element 0 at byte address hb, keys under mask km, child pairs clamped to element cMax) as boost's siftdown
// does: prefer the FIRST maximal child; stop in front of a child that is less than x (key xk).
// Which child is the larger one does not depend on the element being sifted, so a lane decides from two ballots whether
// its node is on the path (every ancestor let the hole pass and turned towards it), and the nodes on the path pull their
// chosen child up in one store.  Slots past a heap's end hold "no element" (a key below every real key): the hole stops
// by itself at a leaf, and a side with nothing (left) to do repeats a block in which nothing moves.  Returns the holes
// (per lane: the hole of the lane's side).
WV_FN V dualDescend(Lds lds, const Sides& S, V hb, V km, V cMax, V xk) {
  V idx = splat(0u);
  for (;;) {
    const V node = ((idx + 1u) << S.lvl) + S.off1;
    V c = node * 2u + 1u;
    c = sel(c < cMax, c, cMax);
    const V2 pr = ldsLoad64(lds, hb + c * 4u);          // children (c, c + 1): one aligned pair
    const V kl = pr.x & km, kr = pr.y & km;
    const B right = kl < kr;
    const V pe = sel(right, pr.y, pr.x);
    const V pk = sel(right, kr, kl);
    const B go = pk >= xk;                              // the hole moves below this node
    const uint64_t goM = ballot(go), rM = ballot(right);
    const V g32 = bothSides(S, lo32(goM), hi32(goM));
    const V r32 = bothSides(S, lo32(rM), hi32(rM));
    const B reached = ((g32 & S.anc) == S.anc) & (((r32 ^ S.needR) & S.anc) == 0u);
    const B onPath = reached & go;
    const uint64_t pM = ballot(onPath);
    ldsStore32m(lds, hb + node * 4u, pe, onPath);       // every node on the path pulls its chosen child up
    // the new hole of this lane's side: below the deepest node on the path (levels are index-ordered), on the side it chose
    const V pm = bothSides(S, lo32(pM), hi32(pM));
    const V steps = popc(pm);
    const V d = 31u - clz(pm | 1u);
    const V rel = d * 2u + 1u + ((r32 >> d) & 1u);
    idx = sel(pm != 0u, ((idx + 1u) << steps) + rel - (splat(1u) << steps), idx);
    if (ballot(steps == 5u) == 0ull) break;
  }
  return idx;
}

// Sift-ups of two heaps at once (boost siftup / libstdc++ __push_heap: while less(parent, e) the parent moves down): a
// side inserts e at position p = p1 - 1 of its heap; a side that has nothing to insert passes p1 = 1 and idle = true.
// Lane l5 of a side loads the chain's ancestor l5 (lanes beyond the root read the word in front of the array, which no
// element is better than); one ballot finds where the sequential loop would have stopped; the ancestors below that point
// move down one level and the new element lands above them — one load, one store.
WV_FN void dualSiftUp(Lds lds, const Sides& S, V hb, V km, V p1, uint32_t e, bool idleA, bool idleB) {
  const V ancPos = (p1 >> (S.l5 + 1u)) - 1u;
  const V ae = ldsLoad32(lds, hb + ancPos * 4u);
  const uint64_t worse = ballot((ae & km) < (km & e));
  // first ancestor that is not worse than e, plus one (0: this side inserts nothing)
  const uint32_t stopA = idleA ? 0u : ctz32(~lo32(worse)) + 1u, stopB = idleB ? 0u : ctz32(~hi32(worse)) + 1u;
  const V stop1 = bothSides(S, stopA, stopB);
  const V dest = (p1 >> S.l5) - 1u;                     // lane l5 < stop: ancestor l5 moves down to here; lane == stop: e
  ldsStore32m(lds, hb + dest * 4u, sel((S.l5 + 1u) == stop1, splat(e), ae), S.l5 < stop1);
}

// Per-lane constants of the four-chains-in-one-wave step: the wave as four rows of 16 lanes.  A sift-up chain of these
// heaps has at most nine ancestors, so a row holds one chain: lane l4 of a row owns ancestor l4.
struct Rows {
  V l4;
  B isY, isSecond;   // rows 2, 3 / rows 1, 3
};
WV_FN Rows makeRows() {
  Rows r;
  const V lane = laneId();
  r.l4 = lane & 15u;
  r.isY = (lane >> 5) != 0u;
  r.isSecond = ((lane >> 4) & 1u) != 0u;
  return r;
}
WV_FN V perRow(const Rows& R, uint32_t x1, uint32_t x2, uint32_t y1, uint32_t y2) {  // rows 0, 1, 2, 3
  return sel(R.isY, sel(R.isSecond, splat(y2), splat(y1)), sel(R.isSecond, splat(x2), splat(x1)));
}

// Up to two consecutive pushes into heap X (elements eX1, eX2 at positions pX, pX + 1: rows 0, 1) and up to two into heap
// Y (rows 2, 3) — each a sift-up as in dualSiftUp — with ONE load and one or two stores.  All chains are loaded before
// anything moves; the second push of a heap must see what the first one left on the part of its chain that the two
// chains share (everything from their lowest common ancestor up; with p = 0 the first element itself): there an ancestor is either untouched, or the first
// push's element, or its own parent moved down one level — all of which the second chain's lanes hold already (the parent
// is the next lane of the row), so the fix-up is a row shift and two selects.
WV_FN void pushPairs(Lds lds, const Rows& R, uint32_t hbX, uint32_t kmX, uint32_t pX, uint32_t nX, uint32_t eX1, uint32_t eX2,
                     uint32_t hbY, uint32_t kmY, uint32_t pY, uint32_t nY, uint32_t eY1, uint32_t eY2) {
  const V hb = sel(R.isY, splat(hbY), splat(hbX));
  const V km = sel(R.isY, splat(kmY), splat(kmX));
  // position + 1 of this row's push (1: the row has nothing to push: no ancestors)
  const V p1 = perRow(R, nX >= 1u ? pX + 1u : 1u, nX >= 2u ? pX + 2u : 1u, nY >= 1u ? pY + 1u : 1u, nY >= 2u ? pY + 2u : 1u);
  const V e = perRow(R, eX1, eX2, eY1, eY2);
  const V ek = e & km;
  const V ancIdx = p1 >> (R.l4 + 1u);
  const V ae = ldsLoad32(lds, hb + ancIdx * 4u - 4u);   // (lanes beyond the root: the word in front of the array)
  const V dest = hb + (p1 >> R.l4) * 4u - 4u;           // lane l4 < stop: ancestor l4 moves down to here; lane == stop: e
  const uint64_t worse1 = ballot((ae & km) < ek);
  // first ancestor that is not worse than the element, plus one (0: nothing to push)
  const uint32_t sX1 = nX >= 1u ? ctz32(~lo32(worse1)) + 1u : 0u, sY1 = nY >= 1u ? ctz32(~hi32(worse1)) + 1u : 0u;
  const V stop1 = sel(R.isY, splat(sY1), splat(sX1));
  ldsStore32m(lds, dest, sel((R.l4 + 1u) == stop1, e, ae), (!R.isSecond) & (R.l4 < stop1));
  if (nX >= 2u || nY >= 2u) {
    // chains of p and p + 1 have the same depth unless p + 2 is a power of two (p + 1 starts a new level)
    const uint32_t ddX = lg2(pX + 2u) - lg2(pX + 1u), ddY = lg2(pY + 2u) - lg2(pY + 1u);
    const V dd = sel(R.isY, splat(ddY), splat(ddX));
    const V e1 = sel(R.isY, splat(eY1), splat(eX1));
    const V j1p1 = R.l4 + 1u - dd;                      // index + 1 of the same position on the first chain
    const B shared = (R.l4 >= dd) & (ancIdx == ((p1 - 1u) >> j1p1));
    const V up = rowShl1(ae, kFront);                   // the parent of this lane's ancestor
    // (p = 0: the first push IS the root, the parent of position 1)
    const V now = sel(shared, sel((j1p1 + 1u) < stop1, up, sel((j1p1 + 1u) == stop1, e1, ae)), sel((p1 == 2u) & (R.l4 == 0u), e1, ae));
    const uint64_t worse2 = ballot((now & km) < ek);
    const uint32_t sX2 = nX >= 2u ? ctz32(~(lo32(worse2) >> 16)) + 1u : 0u, sY2 = nY >= 2u ? ctz32(~(hi32(worse2) >> 16)) + 1u : 0u;
    const V stop2 = sel(R.isY, splat(sY2), splat(sX2));
    ldsStore32m(lds, dest, sel((R.l4 + 1u) == stop2, e, now), R.isSecond & (R.l4 < stop2));
  }
}

// ---- one search ------------------------------------------------------------------------------------------------
// The job is the CJob at oJob of the window, the result the CRes at oRes.  PLDS: the focal path table is in the window at
// oPaths (else at CJob::pathsG); the search loop of a PLDS instance issues no vector-memory LOAD at all — its only
// vector-memory instruction is the cameFrom store — so nothing in it ever waits on vmcnt.
// BG: the (time, cell) bitmap is in device memory (CJob::bitsG): one masked load per expansion, requested before the pops,
// and one merged store per touched word (the Wait / Left / Right successors share theirs).
template <bool EPS, bool PLDS, bool BG = false, class C = Narrow>
WV_ENTRY int32_t compactSearch(Lds window) {
  // the geometry of this instance (the names below hide the narrow tier's at namespace scope)
  constexpr uint32_t kGroups = C::kGroups, kHeapClamp = C::kHeapClamp, kAuxCap = C::kAuxCap, kAuxBytes = C::kAuxBytes,
                     kAuxClamp = C::kAuxClamp, kRows = C::kRows, kBitsBytes = C::kBitsBytes;
  constexpr uint32_t oFocal = C::oFocal, oAux = C::oAux, oBits = C::oBits;
  constexpr uint32_t kMO = C::kMO, kMF = C::kMF, kEmpty = C::kEmpty, kMOAux = C::kMOAux, kAuxShift = C::kAuxShift;
  constexpr uint32_t kFShift = C::kFShift, kFhShift = C::kFhShift, kFMax = C::kFMax, kFhMax = C::kFhMax, kGMax = C::kGMax;
  constexpr uint32_t oObstX = C::obstOff(BG), oPathsX = C::pathsOff(BG);
  constexpr uint32_t oBuild = BG ? oOpen : oBits;   // where the bitmap is put together / where the goal branch stages rows
  const Lds lds = windowBase(window);
  const Sides S = makeSides();
  const Rows Rw = makeRows();
  const V lane = S.lane;
  const V hb = bothSides(S, oOpen + 4u, oFocal + 4u);  // side A = open list, side B = focal list
  const V km = bothSides(S, kMO, kMF);
  const V hbAux = splat(oAux + 4u), kmAux = splat(kMOAux);  // the walk queue: both sides do the same work on it
  const uint32_t dimx = MRP_CT_JOB_U32(lds, dimx), dimy = MRP_CT_JOB_U32(lds, dimy);
  const uint32_t gx = MRP_CT_JOB_U32(lds, gx), gy = MRP_CT_JOB_U32(lds, gy);
  const uint32_t nEc = MRP_CT_JOB_U32(lds, nEc);
  const uint32_t nAgentsPad = EPS ? MRP_CT_JOB_U32(lds, nAgentsPad) : 0u;
  int32_t status = C_NO_SOLUTION, cost = 0, fmin = 0, nStates = 0;
  uint32_t nOpen = 1, nFocal = EPS ? 1u : 0u, nodes = 1, expansions = 0;

  MRP_CT_PROF_DECL;
  // ---- job set-up -------------------------------------------------------------------------------------------
  sync();  // the previous job's LDS reads are done; the CJob block is written
  uint32_t* bitsG = BG ? MRP_CT_JOB_PTR(uint32_t, lds, bitsG) : nullptr;
  {  // obstacle row with a stride of 32 bits per y (the map's bitmap has a stride of dimx): lane y builds word y
    const uint32_t* obst = MRP_CT_JOB_PTR(const uint32_t, lds, obst);
    const uint32_t obstWords = MRP_CT_JOB_U32(lds, obstWords);
    const V y = S.l5;
    const V bitOff = y * dimx;
    const V wi = bitOff >> 5, sh = bitOff & 31u;
    const B rowIn = (y < dimy) & !S.isB;
    const V lo = gLoad32m(obst, wi, rowIn & (wi < obstWords));
    const V hi = gLoad32m(obst, wi + 1u, rowIn & ((wi + 1u) < obstWords));
    V w = sel(sh == 0u, lo, (lo >> sh) | (hi << (splat(32u) - sh)));
    const uint32_t colMask = dimx >= 32u ? 0xFFFFFFFFu : ((1u << dimx) - 1u);
    w = (w & colMask) | ~colMask;                       // columns beyond the map: blocked
    w = sel(rowIn, w, splat(0xFFFFFFFFu));              // rows beyond the map: blocked
    ldsStore32m(lds, splat(oObstX) + y * 4u, w, !S.isB);
  }
  auto initHeaps = [&]() {  // heaps and walk queue: every slot "no element", the words in front of element 0 the largest key
    const V4 none{splat(kEmpty), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
    const V4 head{sel(lane == 0u, splat(kFront), splat(kEmpty)), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
    ldsStore128(lds, splat(oOpen) + lane * 16u, head);
    ldsStore128(lds, splat(oFocal) + lane * 16u, head);
    ldsStore128(lds, splat(oAux) + lane * 16u, head);
    for (uint32_t g = 1; g < kGroups; ++g) {
      ldsStore128(lds, splat(oOpen + g * 1024u) + lane * 16u, none);
      ldsStore128(lds, splat(oFocal + g * 1024u) + lane * 16u, none);
    }
    ldsStore128m(lds, splat(oOpen + kGroups * 1024u), none, lane == 0u);
    ldsStore128m(lds, splat(oFocal + kGroups * 1024u), none, lane == 0u);
    for (uint32_t b = 1024u; b < kAuxBytes; b += 1024u) ldsStore128m(lds, splat(oAux + b) + lane * 16u, none, (lane * 16u + b) < kAuxBytes);
  };
  static_assert(!C::kLongT || BG, "the long-horizon geometry keeps its bitmap in device memory");
  if (!BG || C::kLongT) initHeaps();
  sync();
  // Bitmap rows r0 .. r0 + kRows - 1 (row t = obstacles | vertex constraints at t | states discovered, during the search),
  // put together at `where` in the window.
  auto buildRows = [&](uint32_t where, uint32_t r0, uint32_t nRows) {
    const V4 chunk = ldsLoad128(lds, splat(oObstX) + (lane & 7u) * 16u);
    for (uint32_t i = 0; i < nRows / 8u; ++i)
      ldsStore128(lds, splat(where + i * 8u * kRowBytes) + (lane >> 3) * kRowBytes + (lane & 7u) * 16u, chunk);
    sync();
    // stateValid's vertex constraints (ecbs.cpp:499-502)
    const uint32_t nVc = MRP_CT_JOB_U32(lds, nVc);
    const uint32_t* vc = MRP_CT_JOB_PTR(const uint32_t, lds, vc);
    for (uint32_t j0 = 0; j0 < nVc; j0 += 64u) {
      const B in = (lane + j0) < nVc;
      const V v = gLoad32m(vc, lane + j0, in);
      const V tt = (v >> 16) - r0, yy = (v >> 8) & 0xFFu, xx = v & 0xFFu;  // (rows in front of r0: a huge number)
      ldsOr32m(lds, splat(where) + tt * kRowBytes + yy * 4u, splat(1u) << xx, in & (tt < nRows) & (yy < 32u) & (xx < 32u));
    }
  };
  // LONGT: the rows exist up to rowsReady (exclusive); the next chunk is put together in the walk queue's area — free
  // between walks — copied out to device memory, and the area is given back to the walk queue.
  uint32_t rowsReady = 0;
  auto moreRows = [&]() {
    sync();
    buildRows(oAux, rowsReady, kRows);
    sync();
    for (uint32_t i = 0; i < kBitsBytes / 1024u; ++i)
      gStore128(bitsG, splat(rowsReady * (kRowBytes / 16u) + i * 64u) + lane, ldsLoad128(lds, splat(oAux + i * 1024u) + lane * 16u));
    sync();
    const V4 none{splat(kEmpty), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
    const V4 head{sel(lane == 0u, splat(kFront), splat(kEmpty)), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
    ldsStore128(lds, splat(oAux) + lane * 16u, head);
    for (uint32_t b = 1024u; b < kAuxBytes; b += 1024u) ldsStore128m(lds, splat(oAux + b) + lane * 16u, none, (lane * 16u + b) < kAuxBytes);
    sync();
    rowsReady += kRows;
  };
  if (C::kLongT) {
    moreRows();  // (later chunks: in the loop, as t grows)
  } else if (BG) {  // the finished rows leave for device memory (1 KB stores, coalesced); then the heaps take the area over
    constexpr uint32_t kBuildRows = C::kBuildRows;
    for (uint32_t r0 = 0; r0 < kRows; r0 += kBuildRows) {
      buildRows(oBuild, r0, kBuildRows);
      sync();
      for (uint32_t i = 0; i < kBuildRows * kRowBytes / 1024u; ++i)
        gStore128(bitsG, splat(r0 * (kRowBytes / 16u) + i * 64u) + lane, ldsLoad128(lds, splat(oBuild + i * 1024u) + lane * 16u));
      sync();
    }
    initHeaps();
  } else {
    buildRows(oBuild, 0u, kRows);
  }
  // edge-constraint keys, one per lane.  They pass through the window (the walk queue's area, restored afterwards) so that
  // the loop below holds no register a vector-memory load is still writing.
  V ecReg = splat(0xFFFFFFFFu);
  if (nEc) {
    const uint32_t* ec = MRP_CT_JOB_PTR(const uint32_t, lds, ec);
    ldsStore32(lds, splat(oAux + 16u) + lane * 4u, sel(lane < nEc, gLoad32m(ec, lane, lane < nEc), splat(0xFFFFFFFFu)));
    sync();
    ecReg = ldsLoad32(lds, splat(oAux + 16u) + lane * 4u);
    sync();
    const V4 none{splat(kEmpty), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
    ldsStore128m(lds, splat(oAux + 16u) + lane * 16u, none, lane < 16u);
  }
  // successor of this lane in the reference's order Wait, Left, Right, Up, Down (ecbs.cpp:365-398) on lanes 0..4; the other
  // lanes look at a cell far outside every map
  const V dx = sel(lane == 1u, splat(0xFFFFFFFFu), sel(lane == 2u, splat(1u), sel(lane < 5u, splat(0u), splat(0x4000u))));
  const V dy = sel(lane == 3u, splat(1u), sel(lane == 4u, splat(0xFFFFFFFFu), splat(0u)));
  // focal context: this lane's agents (columns of the path table)
  const B in0 = lane < nAgentsPad, in1 = (lane + 64u) < nAgentsPad;
  const V col0 = sel(in0, lane, splat(nAgentsPad ? nAgentsPad - 1u : 0u));
  const V col1 = sel(in1, lane + 64u, splat(nAgentsPad ? nAgentsPad - 1u : 0u));
  const uint32_t tPad = EPS ? MRP_CT_JOB_U32(lds, tPad) : 0u;
  const uint16_t* pathsG = PLDS ? nullptr : MRP_CT_JOB_PTR(const uint16_t, lds, pathsG);
  const int32_t lastGoal = (int32_t)MRP_CT_JOB_U32(lds, lastGoal);
  const float wBound = uintAsFloat(MRP_CT_JOB_U32(lds, w));
  const uint32_t maxExp = MRP_CT_JOB_U32(lds, maxExp), openCap = MRP_CT_JOB_U32(lds, openCap), maxT = MRP_CT_JOB_U32(lds, maxT);
  uint8_t* parentTab = MRP_CT_JOB_PTR(uint8_t, lds, parentTab);
  const uint32_t goalCell = gx | (gy << 5);
  int32_t bestF;
  {
    const uint32_t sx = MRP_CT_JOB_U32(lds, sx), sy = MRP_CT_JOB_U32(lds, sy);
    const uint32_t h0 = (sx > gx ? sx - gx : gx - sx) + (sy > gy ? sy - gy : gy - sy);
    bestF = (int32_t)h0;
    const uint32_t e0 = (kFhMax << kFhShift) | ((kFMax - h0) << kFShift) | ((C::kLongT ? 63u - h0 : 0u) << 10) | (sx | (sy << 5));
    ldsStoreS(lds, oOpen + 4u, e0);
    if (EPS) ldsStoreS(lds, oFocal + 4u, e0);
  }
  sync();
  MRP_CT_PROF_MARK(0);

  for (;;) {
    if (nOpen == 0u) {
      status = C_NO_SOLUTION;
      break;
    }
    const V tops = ldsLoad32(lds, hb);                   // lanes 0..31: open.top(), lanes 32..63: focal.top()
    const uint32_t topO = readlane(tops, 0);
    uint32_t curE = EPS ? readlane(tops, 32) : topO;
    if (EPS) {
      const int32_t fTop = (int32_t)(kFMax - ((topO >> kFShift) & kFMax));
      const int32_t oldBest = bestF;
      bestF = fTop;                                      // bestFScore = openSet.top().fScore (a_star_epsilon.hpp:136)
      if (fTop > oldBest) {
        MRP_CT_PROF_MARK(0);
        // ---- a_star_epsilon.hpp:134-154: bestFScore grew -> every open node with old * w < f <= new * w joins the focal
        // list, in the order of open.ordered_begin(): a best-first walk of the open array through a std::priority_queue
        // (libstdc++ push_heap / pop_heap restated: push = sift-up, pop = hole down to a leaf, then sift-up)
        const float lo = fmulRn((float)oldBest, wBound), hi = fmulRn((float)fTop, wBound);  // binary32, no contraction
        // The walk changes nothing but the focal list, and only through the nodes of the band old * w < f <= new * w (its
        // queue is dropped afterwards): when no open node has such an f, it is skipped.  f is an integer below 127 here, so
        // the band is f in [floor(lo) + 1, floor(hi)]; one 16-byte read per lane looks at 256 open entries.  Slots that
        // hold no element read as f = kFMax (kEmpty) or f = 0 (the word in front of element 0): never inside.
        // (-DMRP_CT_FORCE_WALK: always walk — the emulator's A/B for "skipping is unobservable")
        bool bandEmpty = false;
#ifndef MRP_CT_FORCE_WALK
        {
          const int32_t fA = (int32_t)lo + 1, fBraw = (int32_t)hi;  // lo, hi >= 0: truncation is floor
          const int32_t fB = fBraw > (int32_t)kFMax - 1 ? (int32_t)kFMax - 1 : fBraw;
          bandEmpty = true;
          if (fB >= fA) {
            const V base = splat(kFMax - (uint32_t)fB), span = splat((uint32_t)(fB - fA));
            for (uint32_t g = 0; g < kGroups && g * 256u <= nOpen; ++g) {  // (element 256 g - 1 belongs to group g)
              const V4 grp = ldsLoad128(lds, splat(oOpen + g * 1024u) + lane * 16u);
              const B in = ((((grp.x >> kFShift) & kFMax) - base) <= span) | ((((grp.y >> kFShift) & kFMax) - base) <= span) |
                           ((((grp.z >> kFShift) & kFMax) - base) <= span) | ((((grp.w >> kFShift) & kFMax) - base) <= span);
              if (ballot(in)) {
                bandEmpty = false;
                break;
              }
            }
          }
        }
#endif
        MRP_CT_PROF_ADD(7, bandEmpty ? 1u : 0u);
        if (!bandEmpty) {
        uint32_t npq = 0, npqHigh = 0;
        uint32_t cur = 0, curKey = topO & kMO, eCur = topO;  // the node being visited: open index, open key, entry
        // lanes 0, 1: the children of the node in the open array (past the end of the list: "no element"); lane 2: the node
        V trio = ldsLoad32(lds, splat(oOpen + 4u) + sel(lane == 2u, splat(cur), splat(2u * cur + 1u) + (lane & 1u)) * 4u);
        for (;;) {
          MRP_CT_PROF_ADD(6, 1u);
          const uint32_t firstC = 2u * cur + 1u;
          const uint32_t nCh = firstC + 1u < nOpen ? 2u : firstC < nOpen ? 1u : 0u;
          if (npq + 2u > kAuxCap) {
            status = C_OVERFLOW;
            cost = 4;
            break;
          }
          // discover the children (index order) before the node is tested; the node joins the focal list if it is in the band
          const uint32_t e1 = readlane(trio, 0), e2 = readlane(trio, 1);
          const float fv = (float)(int32_t)(kFMax - ((curKey >> kFShift) & kFMax));
          const bool inBand = fv > lo && fv <= hi;
          pushPairs(lds, Rw, oAux + 4u, kMOAux, npq, nCh, ((e1 & kMO) << kAuxShift) | firstC, ((e2 & kMO) << kAuxShift) | (firstC + 1u),
                    oFocal + 4u, kMF, nFocal,
                    inBand ? 1u : 0u, eCur, 0u);
          npq += nCh;
          nFocal += inBand ? 1u : 0u;
          npqHigh = npq > npqHigh ? npq : npqHigh;
          if (fv > hi) break;
          if (npq == 0u) break;
          // std::priority_queue::pop == pop_heap: the last element goes into the hole the top leaves behind, i.e. the
          // hole moves down to a leaf (__adjust_heap) and the element up again from there (__push_heap).  The next node
          // (the top) and the last element are fetched together with the first block of the hole's way down, which does
          // not depend on them.
          npq -= 1u;
          const V two = ldsLoad32(lds, splat(oAux + 4u) + sel(lane == 0u, splat(0u), splat(npq)) * 4u);
          ldsStoreS(lds, oAux + 4u + 4u * npq, kEmpty);    // the vacated slot: "no element" (before the hole's way down looks)
          uint32_t hole = 0, above = kFront;  // `above`: what now sits in the hole's parent
          bool first = true;
          uint32_t value = 0;
          for (;;) {
            const V node = ((splat(hole) + 1u) << S.lvl) + S.off1;
            V c = node * 2u + 1u;
            c = sel(c < kAuxClamp, c, splat(kAuxClamp));
            const V2 pr = ldsLoad64(lds, splat(oAux + 4u) + c * 4u);
            if (first) {
              first = false;
              const uint32_t curA = readlane(two, 0);
              value = readlane(two, 1);
              cur = curA & C::kAuxIdxMask;                   // (the bits below the key)
              curKey = (curA >> kAuxShift) & kMO;
              // the next node's entry and children, for the next turn of the loop
              trio = ldsLoad32(lds, splat(oOpen + 4u) + sel(lane == 2u, splat(cur), splat(2u * cur + 1u) + (lane & 1u)) * 4u);
              if (npq == 0u) break;
            }
            const V kl = pr.x & kMOAux, kr = pr.y & kMOAux;
            const B right = kr >= kl;                        // prefer the right child unless it is less than the left one
            const V pe = sel(right, pr.y, pr.x);
            const B go = kl > (kEmpty & kMOAux);             // down to a leaf: as long as there is a child
            const uint32_t goM = lo32(ballot(go)), rM = lo32(ballot(right));
            const B onPath = ((S.anc & goM) == S.anc) & (((S.needR ^ rM) & S.anc) == 0u) & go;
            const uint32_t pm = lo32(ballot(onPath & !S.isB));
            ldsStore32m(lds, splat(oAux + 4u) + node * 4u, pe, onPath);
            if (pm == 0u) break;
            const uint32_t steps = (uint32_t)__builtin_popcount(pm), d = lg2(pm);
            hole = ((hole + 1u) << steps) + 2u * d + 1u + ((rM >> d) & 1u) - (1u << steps);
            above = readlane(pe, d);
            if (steps < 5u) break;
          }
          eCur = readlane(trio, 2);
          if (npq > 0u) {
            if (hole == 0u || !((above & kMOAux) < (value & kMOAux)))
              ldsStoreS(lds, oAux + 4u + 4u * hole, value);  // the usual case: the last element stays at the leaf
            else
              dualSiftUp(lds, S, hbAux, kmAux, splat(hole + 1u), value, false, false);
          }
        }
        // what the walk leaves in its queue is dropped: every slot "no element" again
        for (uint32_t b = 0; b <= 4u * npqHigh; b += 1024u) {
          const V4 none{splat(kEmpty), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
          const V4 head{sel(lane == 0u, splat(kFront), splat(kEmpty)), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
          ldsStore128m(lds, splat(oAux + b) + lane * 16u, b == 0u ? head : none, (lane * 16u + b) < kAuxBytes);
        }
        if (status == C_OVERFLOW) break;
        curE = ldsLoadS(lds, oFocal + 4u);
        }
        MRP_CT_PROF_MARK(1);
      }
    }
    // f, g (== time) and focalH of the popped node are in its entry
    // (LONGT: the field below f holds 63 - h, and g = f - h)
    const uint32_t cell = curE & 1023u, curFh = kFhMax - (curE >> kFhShift);
    const uint32_t t = C::kLongT ? (kFMax - ((curE >> kFShift) & kFMax)) - (63u - ((curE >> 10) & 63u)) : (curE >> 10) & kGMax;
    const uint32_t x = cell & 31u, y = cell >> 5;
    const bool isGoal = cell == goalCell && (int32_t)t > lastGoal;
    if (!isGoal) {
      // (cost names the limit and `expanded` how far the search got: statistics for the caller, not results)
      if (nOpen + 5u > openCap || t > maxT || (EPS && !C::kExactFhCheck && curFh + 2u * nAgentsPad > kFhMax)) {
        status = C_OVERFLOW;
        cost = nOpen + 5u > openCap ? 1 : t > maxT ? 2 : 3;
        break;
      }
    }
    expansions += 1u;  // onExpandNode (a_star_epsilon.hpp:193 / a_star.hpp:87) — counts the goal pop too
    if (expansions > maxExp) {
      status = C_CAP_EXP;
      break;
    }
    if (isGoal) {
      // ---- a_star_epsilon.hpp:195-213: follow cameFrom back to the start.  The action bytes of eight time steps (eight
      // 1 KB rows of the table) are fetched per round trip into the bitmap's LDS area, which the search no longer needs
      status = C_OK;
      cost = (int32_t)t;
      fmin = (int32_t)(kFMax - (((EPS ? topO : curE) >> kFShift) & kFMax));
      nStates = (int32_t)t + 1;
      uint16_t* outPath = MRP_CT_JOB_PTR(uint16_t, lds, outPath);
      sync();  // this wave's action stores have left the CU
      uint32_t c = cell;
      const uint32_t* tab32 = (const uint32_t*)parentTab;
      constexpr int32_t kStage = BG ? (int32_t)C::kStageRows : 8;  // (the bitmap's own area holds eight rows)
      for (int32_t k0 = (int32_t)t; k0 >= 1; k0 -= kStage) {
        V rowW[kStage][4];
        for (int32_t r = 0; r < kStage; ++r)
          for (uint32_t q = 0; q < 4; ++q)
            rowW[r][q] = (k0 - r >= 1) ? gLoad32Coherent(tab32, splat((uint32_t)(k0 - r) * 256u + q * 64u) + lane) : splat(0u);
        for (int32_t r = 0; r < kStage; ++r)
          for (uint32_t q = 0; q < 4; ++q) ldsStore32(lds, splat(oBuild + (uint32_t)r * 1024u + q * 256u) + lane * 4u, rowW[r][q]);
        sync();
        for (int32_t r = 0; r < kStage && k0 - r >= 1; ++r) {
          const uint32_t k = (uint32_t)(k0 - r);
          gStoreU16m(outPath, splat(k), splat((c & 31u) | ((c >> 5) << 8)), lane == 0u);
          const uint32_t a = first(ldsLoadU8(lds, splat(oBuild + (uint32_t)r * 1024u + c)));
          // the parent's cell: undo Wait, Left, Right, Up, Down
          c = a == 1u ? c + 1u : a == 2u ? c - 1u : a == 3u ? c - 32u : a == 4u ? c + 32u : c;
        }
        sync();
      }
      gStoreU16m(outPath, splat(0u), splat((c & 31u) | ((c >> 5) << 8)), lane == 0u);
      break;
    }

    MRP_CT_PROF_MARK(0);
    const uint32_t t1 = t + 1u;
    if (C::kLongT && t1 >= rowsReady) moreRows();  // (t1 <= maxT + 1 < the job's rows; the walk queue is empty here)
    // the five successor probes: bounds, then ONE bit of the (time, cell) bitmap = obstacle | vertex constraint | already
    // discovered; requested before the pops so that the latency hides behind them
    const V nx = splat(x) + dx, ny = splat(y) + dy;
    const B inb = (nx < dimx) & (ny < dimy);
    const V ncell = (nx & 31u) | ((ny & 31u) << 5);
    const V wordAddr = splat(oBits + t1 * kRowBytes) + ((ny & 31u) << 2);
    const V wordIdx = splat(t1 * (kRowBytes / 4u)) + (ny & 31u);
    // (a plain, cached load: the bitmap is written by this wave alone — set-up above, the merged stores below — and a
    // wave's own stores are what its later loads see; an agent-scope load goes past the XCD's L2 to memory and cost
    // 0.9 us per expansion, measured)
    // (unmasked: the lanes that probe nothing look at the word of the Wait successor — the same cache line, no branch)
    const V word = BG ? gLoad32m(bitsG, wordIdx, bsplat(true)) : ldsLoad32(lds, wordAddr);
    // other agents' positions at t (a) and t + 1 (b), one agent per lane
    V a0 = splat(0xFFFFu), b0 = splat(0xFFFFu), a1 = splat(0xFFFFu), b1 = splat(0xFFFFu);
    if (EPS && nAgentsPad) {
      const uint32_t ra = t < tPad ? t : tPad - 1u;
      const uint32_t rb = t1 < tPad ? t1 : tPad - 1u;
      // (lanes beyond a row's end look at the row's last agent instead, and are masked: every read stays inside the table)
      if (PLDS) {
        a0 = sel(in0, ldsLoadU16(lds, splat(oPathsX + ra * nAgentsPad * 2u) + col0 * 2u), splat(0xFFFFu));
        b0 = sel(in0, ldsLoadU16(lds, splat(oPathsX + rb * nAgentsPad * 2u) + col0 * 2u), splat(0xFFFFu));
        if (nAgentsPad > 64u) {
          a1 = sel(in1, ldsLoadU16(lds, splat(oPathsX + ra * nAgentsPad * 2u) + col1 * 2u), splat(0xFFFFu));
          b1 = sel(in1, ldsLoadU16(lds, splat(oPathsX + rb * nAgentsPad * 2u) + col1 * 2u), splat(0xFFFFu));
        }
      } else {
        a0 = sel(in0, gLoadU16m(pathsG, splat(ra * nAgentsPad) + col0, in0), splat(0xFFFFu));
        b0 = sel(in0, gLoadU16m(pathsG, splat(rb * nAgentsPad) + col0, in0), splat(0xFFFFu));
        if (nAgentsPad > 64u) {
          a1 = sel(in1, gLoadU16m(pathsG, splat(ra * nAgentsPad) + col1, in1), splat(0xFFFFu));
          b1 = sel(in1, gLoadU16m(pathsG, splat(rb * nAgentsPad) + col1, in1), splat(0xFFFFu));
        }
      }
    }

    MRP_CT_PROF_MARK(2);
    // ---- a_star_epsilon.hpp:215-216: focalSet.pop(), openSet.erase(handle of the same node)   (a_star.hpp:109: pop)
    {
      const uint32_t nOld = nOpen;
      nOpen -= 1u;
      if (EPS) nFocal -= 1u;
      // the elements that pop() moves to the roots: the last ones
      const V lastAddr = hb + bothSides(S, nOpen, nFocal) * 4u;
      const V lastV = ldsLoad32(lds, lastAddr);
      uint32_t p = 0;
      if (EPS) {  // where is the popped node in the open array?  Lane L looks at elements 4L - 1 .. 4L + 2 of a group.
        const V key = splat(curE & C::kStateMask);
        // (narrow geometry: no early exit — a state is in the open array exactly once, and a loop with one exit compiles to
        // half the scalar bookkeeping; the groups behind the hit cost one LDS read each)
        for (uint32_t g = 0; g < kGroups && g * 256u <= nOld; ++g) {  // (element 256 g - 1 belongs to group g)
          const V4 grp = ldsLoad128(lds, splat(oOpen + g * 1024u) + lane * 16u);
          const B m0 = (grp.x & C::kStateMask) == key, m1 = (grp.y & C::kStateMask) == key, m2 = (grp.z & C::kStateMask) == key,
                  m3 = (grp.w & C::kStateMask) == key;
          const uint64_t any = ballot(m0 | m1 | m2 | m3);
          if (any) {
            const V pos = lane * 4u + sel(m0, splat(0xFFFFFFFFu), sel(m1, splat(0u), sel(m2, splat(1u), splat(2u))));
            p = g * 256u + readlane(pos, ctz64(any));
            if (kGroups > 4u) break;  // (the wide geometry: up to twelve groups, worth leaving early)
          }
        }
      }
      uint32_t lastO = readlane(lastV, 0);
      const uint32_t lastF = EPS ? readlane(lastV, 32) : kFront;  // (no focal list: a key nothing is better than)
      ldsStore32m(lds, lastAddr, splat(kEmpty), S.l5 == 0u);  // the vacated slots: "no element"
      if (EPS) {  // boost erase = bubble to the root (every ancestor of p moves down one level), then pop
        const uint32_t depth = lg2(p + 1u);
        const B act = !S.isB & (S.l5 < depth);
        const V ancPos = (splat(p + 1u) >> (S.l5 + 1u)) - 1u;
        const V ae = ldsLoad32(lds, splat(oOpen + 4u) + ancPos * 4u);
        ldsStore32m(lds, splat(oOpen + 4u) + ((splat(p + 1u) >> S.l5) - 1u) * 4u, ae, act);
        // ... which has just overwritten the last element if the erased node WAS the last one: it is the node's parent
        // then, and the slot it sat in is vacated after all
        if (p == nOld - 1u && depth != 0u) {
          lastO = readlane(ae, 0);
          ldsStoreS(lds, oOpen + 4u + 4u * p, kEmpty);
        }
      }
      const V xk = bothSides(S, lastO & kMO, lastF & kMF);
      const V hole = dualDescend(lds, S, hb, km, splat(kHeapClamp), xk);
      ldsStore32m(lds, hb + hole * 4u, bothSides(S, lastO, lastF), (S.l5 == 0u) & (bothSides(S, nOpen, EPS ? nFocal : 0u) != 0u));
    }

    MRP_CT_PROF_MARK(3);
    const B okV = inb & (((word >> (nx & 31u)) & 1u) == 0u);
    uint32_t mask = lo32(ballot(okV)) & 0x1Fu;
    if (nEc) {  // transitionValid (ecbs.cpp:505-510): lane j holds edge-constraint key j = t << 19 | cell << 3 | action
      const uint32_t base = (t << 19) | ((y * dimx + x) << 3);
      const V d = ecReg - base;
      if (ballot(d < 5u)) {  // rare: some constraint names a move out of this very state
        uint32_t blocked = 0;
        for (uint32_t k = 0; k < 5u; ++k) blocked |= ballot(d == k) ? (1u << k) : 0u;
        mask &= ~blocked;
      }
    }
    if (mask == 0u) {
      MRP_CT_PROF_MARK(4);
      continue;
    }

    // ---- the successors' entries, one per lane 0..4
    const B mine = (lane < 5u) & (((splat(mask) >> lane) & 1u) != 0u);
    const V f = sad(nx, splat(gx), sad(ny, splat(gy), splat(t1)));  // g + admissibleHeuristic (ecbs.cpp:276-279)
    V fhV = splat(curFh);
    if (EPS && nAgentsPad) {
      // focalStateHeuristic (ecbs.cpp:282-295) + focalTransitionHeuristic (ecbs.cpp:298-312): an agent counts once if it
      // stands on the successor's cell at t + 1 and once more if it swaps places with this agent
      const V nxy = nx | (ny << 8);
      const uint32_t curXy = x | (y << 8);
      const uint64_t swap0 = ballot(b0 == curXy);
      const uint64_t swap1 = nAgentsPad > 64u ? ballot(b1 == curXy) : 0ull;
      for (uint32_t mm = mask; mm; mm &= mm - 1u) {
        const uint32_t k = ctz32(mm);
        const uint32_t cc = readlane(nxy, k);
        uint32_t cnt = popc64(ballot(b0 == cc)) + popc64(ballot(a0 == cc) & swap0);
        if (nAgentsPad > 64u) cnt += popc64(ballot(b1 == cc)) + popc64(ballot(a1 == cc) & swap1);
        fhV = writelane(fhV, curFh + cnt, k);
      }
    }
    if (EPS && C::kExactFhCheck && ballot(mine & (fhV > kFhMax))) {  // a focalH beyond the entry's field: not a search of this tier
      status = C_OVERFLOW;
      cost = 3;
      break;
    }
    const V lowKey = C::kLongT ? (splat(63u + t1) - f) << 10 : splat(t1 << 10);  // g, or 63 - h (h = f - g)
    const V eV = ((splat(kFhMax) - fhV) << kFhShift) | ((splat(kFMax) - f) << kFShift) | lowKey | ncell;
    uint32_t maskF = 0;
    if (EPS) {
      const float bound = fmulRn((float)bestF, wBound);  // a_star_epsilon.hpp:240, binary32
      maskF = lo32(ballot(mine & leF32(cvtF32(f), bound)));
    }
    // discovered: stands for stateToHeap / closedSet membership (a_star_epsilon.hpp:224-227)
    if (BG) {  // Wait, Left and Right (lanes 0..2) share the word of row y: each of them stores all of their bits
      const uint32_t bx = 1u << x;
      const uint32_t rowBits = ((mask & 1u) ? bx : 0u) | ((mask & 2u) ? bx >> 1 : 0u) | ((mask & 4u) ? bx << 1 : 0u);
      // ... together with cameFrom (a_star_epsilon.hpp:275-279): the action that led here — one masked region
      gStore32and8m(bitsG, wordIdx, word | sel(lane < 3u, splat(rowBits), splat(1u) << (nx & 31u)), parentTab,
                    splat(t1 << 10) + ncell, lane, mine);
    } else {
      ldsOr32m(lds, wordAddr, splat(1u) << (nx & 31u), mine);
      gStore8m(parentTab, splat(t1 << 10) + ncell, lane, mine);
    }
    nodes += (uint32_t)__builtin_popcount(mask);
    MRP_CT_PROF_MARK(4);
    // ---- openSet.push for every successor, focalSet.push for those within the bound, in successor order — two successors
    // per step (pushes into different lists do not see each other, so the four chains of a step are loaded together)
    for (uint32_t mm = mask; mm;) {
      const uint32_t k1 = ctz32(mm);
      mm &= mm - 1u;
      const bool two = mm != 0u;
      const uint32_t k2 = two ? ctz32(mm) : 0u;
      mm &= mm - 1u;  // (0 stays 0)
      const uint32_t e1 = readlane(eV, k1), e2 = readlane(eV, k2);
      const bool f1 = EPS && ((maskF >> k1) & 1u), f2 = EPS && two && ((maskF >> k2) & 1u);
      const uint32_t nX = two ? 2u : 1u, nY = (f1 ? 1u : 0u) + (f2 ? 1u : 0u);
#ifndef MRP_CT_NO_SINGLE_PUSH
      if (!two)  // one successor (the usual case): its two chains side by side, half the bookkeeping
        dualSiftUp(lds, S, hb, km, bothSides(S, nOpen + 1u, f1 ? nFocal + 1u : 1u), e1, false, !f1);
      else
#endif
        pushPairs(lds, Rw, oOpen + 4u, kMO, nOpen, nX, e1, e2, oFocal + 4u, kMF, nFocal, nY, f1 ? e1 : e2, e2);
      nOpen += nX;
      nFocal += nY;
    }
    MRP_CT_PROF_MARK(5);
  }
  // the result block of the window
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, status), (uint32_t)status);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, cost), (uint32_t)cost);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, fmin), (uint32_t)fmin);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, nStates), (uint32_t)nStates);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, expanded), expansions);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, nodes), nodes);
  MRP_CT_PROF_STORE(lds);
  sync();
  return status;
}

// ---- the low level of the task-assignment callers ----------------------------------------------
// AStar::search (a_star.hpp:63-161) over the Environment of example/cbs_ta.cpp:283-372,483-496 (cbs_ta.hpp:106-109,
// 155-158,196-199; ecbs_ta's low level shares the Environment):
//   * the task (goal) is optional: without one h = 0, every cell is a goal cell, and the search may end as soon as
//     time > the time of the agent's LAST vertex constraint of any cell (setLowLevelContext :283-303, isSolution :313-319);
//   * h = shortest-path distance to the task's cell from an uploaded table (shortest_path_heuristic.hpp:56-60);
//   * Wait costs 0 at the goal cell (everywhere without a task), every other action 1 (getNeighbors :321-367).
// So g != time: a state (time, cell) can be discovered again with a smaller g, and the decrease-key branch
// a_star.hpp:139-145 is live.  Entry:  [30:23] 255 - f   [22:16] g   [15:10] time   [9:0] cell = y * 32 + x  (open order
// a_star.hpp:168-179: f asc, g desc).  The (time, cell) bitmap holds "discovered" only (obstacles: the obstacle row;
// vertex constraints: one key per lane, like the edge constraints), and a discovered successor is looked up in the open
// array by the same scan that openSet.erase uses in compactSearch: found = still open, with its g in the entry (then
// `openSet.increase(handle)` = a sift-up from that position); not found = closed.  cameFrom is the same byte table,
// overwritten when a state is re-parented.  No arena tier behind this one: a search that outgrows the window, or an f above
// 254, ends with a capacity status.
constexpr uint32_t kTaKm = 0x7FFF0000u;
enum : int32_t { C_CAP_NODES = 3, C_CAP_HORIZON = 4 };
WV_ENTRY int32_t compactSearchTA(Lds window) {
  const Lds lds = windowBase(window);
  const Sides S = makeSides();
  const V lane = S.lane;
  const V hbO = splat(oOpen + 4u), kmO = splat(kTaKm);  // one heap: both sides of the wave do the same work on it
  const uint32_t dimx = MRP_CT_JOB_U32(lds, dimx), dimy = MRP_CT_JOB_U32(lds, dimy);
  const uint32_t gx = MRP_CT_JOB_U32(lds, gx), gy = MRP_CT_JOB_U32(lds, gy);
  const uint32_t nEc = MRP_CT_JOB_U32(lds, nEc), nVc = MRP_CT_JOB_U32(lds, nVc);
  const bool noGoal = MRP_CT_JOB_U32(lds, taNoGoal) != 0u;
  int32_t status = C_NO_SOLUTION, cost = 0, fmin = 0, nStates = 0;
  uint32_t nOpen = 1, nodes = 1, expansions = 0;
  sync();
  {  // obstacle row (stride 32) as in compactSearch
    const uint32_t* obst = MRP_CT_JOB_PTR(const uint32_t, lds, obst);
    const uint32_t obstWords = MRP_CT_JOB_U32(lds, obstWords);
    const V y = S.l5;
    const V bitOff = y * dimx;
    const V wi = bitOff >> 5, sh = bitOff & 31u;
    const B rowIn = (y < dimy) & !S.isB;
    const V lo = gLoad32m(obst, wi, rowIn & (wi < obstWords));
    const V hi = gLoad32m(obst, wi + 1u, rowIn & ((wi + 1u) < obstWords));
    V w = sel(sh == 0u, lo, (lo >> sh) | (hi << (splat(32u) - sh)));
    const uint32_t colMask = dimx >= 32u ? 0xFFFFFFFFu : ((1u << dimx) - 1u);
    w = (w & colMask) | ~colMask;
    w = sel(rowIn, w, splat(0xFFFFFFFFu));
    ldsStore32m(lds, splat(oObst) + y * 4u, w, !S.isB);
  }
  {  // open list: every slot "no element"; (time, cell) bitmap: nothing discovered; heuristic table into the window
    const V4 none{splat(kEmpty), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
    const V4 head{sel(lane == 0u, splat(kFront), splat(kEmpty)), splat(kEmpty), splat(kEmpty), splat(kEmpty)};
    const V4 zero{splat(0u), splat(0u), splat(0u), splat(0u)};
    ldsStore128(lds, splat(oOpen) + lane * 16u, head);
    for (uint32_t g = 1; g < kGroups; ++g) ldsStore128(lds, splat(oOpen + g * 1024u) + lane * 16u, none);
    ldsStore128m(lds, splat(oOpen + kGroups * 1024u), none, lane == 0u);
    for (uint32_t i = 0; i < kRows / 8u; ++i) ldsStore128(lds, splat(oBits + i * 1024u) + lane * 16u, zero);
    if (!noGoal) {
      const uint32_t* heur = (const uint32_t*)MRP_CT_JOB_PTR(const uint16_t, lds, pathsG);
      for (uint32_t q = 0; q < 8u; ++q) ldsStore32(lds, splat(oPaths + q * 256u) + lane * 4u, gLoad32m(heur, splat(q * 64u) + lane, bsplat(true)));
    }
  }
  sync();
  // constraint keys, one per lane each: vertex t << 16 | y << 8 | x, edge t << 19 | (y * dimx + x) << 3 | k; they pass
  // through the window so that the loop holds no register a vector-memory load is still writing
  V vcReg = splat(0xFFFFFFFFu), ecReg = splat(0xFFFFFFFFu);
  {
    const uint32_t* vc = MRP_CT_JOB_PTR(const uint32_t, lds, vc);
    const uint32_t* ec = MRP_CT_JOB_PTR(const uint32_t, lds, ec);
    ldsStore32(lds, splat(oAux) + lane * 4u, sel(lane < nVc, gLoad32m(vc, lane, lane < nVc), splat(0xFFFFFFFFu)));
    ldsStore32(lds, splat(oAux + 256u) + lane * 4u, sel(lane < nEc, gLoad32m(ec, lane, lane < nEc), splat(0xFFFFFFFFu)));
    sync();
    vcReg = ldsLoad32(lds, splat(oAux) + lane * 4u);
    ecReg = ldsLoad32(lds, splat(oAux + 256u) + lane * 4u);
    sync();
  }
  const V dx = sel(lane == 1u, splat(0xFFFFFFFFu), sel(lane == 2u, splat(1u), sel(lane < 5u, splat(0u), splat(0x4000u))));
  const V dy = sel(lane == 3u, splat(1u), sel(lane == 4u, splat(0xFFFFFFFFu), splat(0u)));
  const int32_t lastGoal = (int32_t)MRP_CT_JOB_U32(lds, lastGoal);
  const uint32_t maxExp = MRP_CT_JOB_U32(lds, maxExp), openCap = MRP_CT_JOB_U32(lds, openCap), maxT = MRP_CT_JOB_U32(lds, maxT);
  uint8_t* parentTab = MRP_CT_JOB_PTR(uint8_t, lds, parentTab);
  const uint32_t goalCell = gx | (gy << 5);
  {
    const uint32_t sx = MRP_CT_JOB_U32(lds, sx), sy = MRP_CT_JOB_U32(lds, sy);
    const uint32_t sc = sx | (sy << 5);
    const uint32_t h0 = noGoal ? 0u : first(ldsLoadU16(lds, splat(oPaths + sc * 2u)));
    if (h0 > 254u) status = C_CAP_HORIZON;  // (the start cannot reach its task at all, or not within this tier's f)
    ldsStoreS(lds, oOpen + 4u, ((255u - (h0 & 255u)) << 23) | (0u << 16) | (0u << 10) | sc);
  }
  sync();

  while (status == C_NO_SOLUTION) {
    if (nOpen == 0u) break;  // open list exhausted: search() returns false (a_star.hpp:160)
    const uint32_t curE = ldsLoadS(lds, oOpen + 4u);
    const uint32_t cell = curE & 1023u, t = (curE >> 10) & 63u, g = (curE >> 16) & 127u, fCur = 255u - (curE >> 23);
    const uint32_t x = cell & 31u, y = cell >> 5;
    const bool atGoal = noGoal || cell == goalCell;
    expansions += 1u;  // onExpandNode (a_star.hpp:87)
    if (expansions > maxExp) {
      status = C_CAP_EXP;
      break;
    }
    if (atGoal && (int32_t)t > lastGoal) {  // isSolution (cbs_ta.cpp:313-319) -> a_star.hpp:89-106
      status = C_OK;
      cost = (int32_t)g;
      fmin = (int32_t)fCur;
      nStates = (int32_t)t + 1;
      uint16_t* outPath = MRP_CT_JOB_PTR(uint16_t, lds, outPath);
      sync();
      uint32_t c = cell;
      const uint32_t* tab32 = (const uint32_t*)parentTab;
      for (int32_t k0 = (int32_t)t; k0 >= 1; k0 -= 8) {
        V rowW[8][4];
        for (int32_t r = 0; r < 8; ++r)
          for (uint32_t q = 0; q < 4; ++q)
            rowW[r][q] = (k0 - r >= 1) ? gLoad32Coherent(tab32, splat((uint32_t)(k0 - r) * 256u + q * 64u) + lane) : splat(0u);
        for (int32_t r = 0; r < 8; ++r)
          for (uint32_t q = 0; q < 4; ++q) ldsStore32(lds, splat(oBits + (uint32_t)r * 1024u + q * 256u) + lane * 4u, rowW[r][q]);
        sync();
        for (int32_t r = 0; r < 8 && k0 - r >= 1; ++r) {
          gStoreU16m(outPath, splat((uint32_t)(k0 - r)), splat((c & 31u) | ((c >> 5) << 8)), lane == 0u);
          const uint32_t a = first(ldsLoadU8(lds, splat(oBits + (uint32_t)r * 1024u + c)));
          c = a == 1u ? c + 1u : a == 2u ? c - 1u : a == 3u ? c - 32u : a == 4u ? c + 32u : c;
        }
        sync();
      }
      gStoreU16m(outPath, splat(0u), splat((c & 31u) | ((c >> 5) << 8)), lane == 0u);
      break;
    }
    if (nOpen + 5u > openCap) {
      status = C_CAP_NODES;
      break;
    }
    if (t > maxT) {
      status = C_CAP_HORIZON;
      break;
    }
    // openSet.pop() (a_star.hpp:109): the last element goes to the root and sifts down
    {
      nOpen -= 1u;
      const uint32_t last = ldsLoadS(lds, oOpen + 4u + 4u * nOpen);
      ldsStoreS(lds, oOpen + 4u + 4u * nOpen, kEmpty);
      const V hole = dualDescend(lds, S, hbO, kmO, splat(kHeapClamp), splat(last & kTaKm));
      if (nOpen) ldsStoreS(lds, oOpen + 4u + 4u * first(hole), last);
    }
    // getNeighbors (cbs_ta.cpp:321-367): the five probes on lanes 0..4 — bounds, obstacle, vertex constraint (stateValid),
    // edge constraint (transitionValid) — then, in order, the new / rediscovered / closed cases of a_star.hpp:116-153
    const uint32_t t1 = t + 1u;
    const V nx = splat(x) + dx, ny = splat(y) + dy;
    const B inb = (nx < dimx) & (ny < dimy);
    const V ncell = (nx & 31u) | ((ny & 31u) << 5);
    const V obstW = ldsLoad32(lds, splat(oObst) + ((ny & 31u) << 2));
    const V wordAddr = splat(oBits + t1 * kRowBytes) + ((ny & 31u) << 2);
    const V seenW = ldsLoad32(lds, wordAddr);
    const V hN = noGoal ? splat(0u) : ldsLoadU16(lds, splat(oPaths) + ncell * 2u);
    uint32_t mask = lo32(ballot(inb & (((obstW >> (nx & 31u)) & 1u) == 0u))) & 0x1Fu;
    if (nVc) {
      const V nkey = (t1 << 16) | ((ny & 0xFFu) << 8) | (nx & 0xFFu);
      for (uint32_t mm = mask; mm; mm &= mm - 1u) {
        const uint32_t k = ctz32(mm);
        if (ballot(vcReg == readlane(nkey, k))) mask &= ~(1u << k);
      }
    }
    if (nEc) {
      const uint32_t base = (t << 19) | ((y * dimx + x) << 3);
      const V d = ecReg - base;
      if (ballot(d < 5u))
        for (uint32_t k = 0; k < 5u; ++k)
          if (ballot(d == k)) mask &= ~(1u << k);
    }
    const uint32_t seenMask = lo32(ballot(((seenW >> (nx & 31u)) & 1u) != 0u));
    for (uint32_t mm = mask; mm && status == C_NO_SOLUTION; mm &= mm - 1u) {
      const uint32_t k = ctz32(mm);
      const uint32_t nc = readlane(ncell, k), h = readlane(hN, k);
      const uint32_t g2 = g + ((k == 0u && atGoal) ? 0u : 1u);  // tentative_gScore (a_star.hpp:118)
      const uint32_t id = (t1 << 10) | nc;
      if (!((seenMask >> k) & 1u)) {  // not in the open list, not closed: a new node (a_star.hpp:120-129)
        const uint32_t f2 = g2 + h;
        if (h > 254u || f2 > 254u || g2 > 127u) {
          status = C_CAP_HORIZON;
          break;
        }
        ldsOr32m(lds, wordAddr, splat(1u) << (nx & 31u), lane == k);
        gStore8m(parentTab, splat(id), splat(k), lane == 0u);
        dualSiftUp(lds, S, hbO, kmO, splat(nOpen + 1u), ((255u - f2) << 23) | (g2 << 16) | id, false, false);
        nOpen += 1u;
        nodes += 1u;
        continue;
      }
      // discovered before: still in the open list (then its entry says with which g), or closed (a_star.hpp:117)
      uint32_t p = 0xFFFFFFFFu, eOld = 0;
      for (uint32_t grp = 0; grp < kGroups && grp * 256u <= nOpen; ++grp) {  // (element 256 grp - 1 belongs to group grp)
        const V4 q = ldsLoad128(lds, splat(oOpen + grp * 1024u) + lane * 16u);
        const B m0 = (q.x & 0xFFFFu) == id, m1 = (q.y & 0xFFFFu) == id, m2 = (q.z & 0xFFFFu) == id, m3 = (q.w & 0xFFFFu) == id;
        const uint64_t any = ballot(m0 | m1 | m2 | m3);
        if (any) {
          const uint32_t l = ctz64(any);
          const V pos = lane * 4u + sel(m0, splat(0xFFFFFFFFu), sel(m1, splat(0u), sel(m2, splat(1u), splat(2u))));
          const V ent = sel(m0, q.x, sel(m1, q.y, sel(m2, q.z, q.w)));
          p = grp * 256u + readlane(pos, l);
          eOld = readlane(ent, l);
          break;
        }
      }
      if (p == 0xFFFFFFFFu) continue;  // closed
      const uint32_t gOld = (eOld >> 16) & 127u;
      if (g2 >= gOld) continue;        // not an improvement (a_star.hpp:135-137)
      const uint32_t fNew = (255u - (eOld >> 23)) - (gOld - g2);  // fScore -= delta (a_star.hpp:141-142)
      gStore8m(parentTab, splat(id), splat(k), lane == 0u);       // cameFrom is replaced (a_star.hpp:150-152)
      dualSiftUp(lds, S, hbO, kmO, splat(p + 1u), ((255u - fNew) << 23) | (g2 << 16) | id, false, false);  // increase(handle)
    }
  }
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, status), (uint32_t)status);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, cost), (uint32_t)cost);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, fmin), (uint32_t)fmin);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, nStates), (uint32_t)nStates);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, expanded), expansions);
  ldsStoreS(lds, oRes + (uint32_t)offsetof(CRes, nodes), nodes);
  sync();
  return status;
}

}  // namespace ct
}  // namespace mrp
