// Device-side job/result records shared by the HIP kernels (ll_kernel.hip) and the host packer (mrp_ll_host.cpp).
#pragma once
#include <stdint.h>

namespace mrp {

// ---- packed heap entry (uint64) ------------------------------------------------------------------------------
//   high word (the key): [31:21] FH_MAXV - focalH   [20:10] F_MAXV - f   [9:0] g        low word: node id
// A larger key is a BETTER node in the reference's orders:
//   open  (a_star_epsilon.hpp:312-323, a_star.hpp:168-179): lowest f, then highest g      -> key bits [20:0]
//   focal (a_star_epsilon.hpp:346-366): lowest focalH, then lowest f, then highest g      -> key bits [31:0]
// Entries with equal keys compare EQUAL (the id never takes part), exactly like the reference's comparators; which
// of two equal entries comes out first is decided by the heap layout, which the kernels replay verbatim.
constexpr uint32_t kGBits = 10, kFBits = 11, kFhBits = 11;
// compact LDS tier of the CBS / ECBS kernels (ll_compact.h): mrp_ll_options.lds_nodes / 2 = open-list entries a search may
// hold inside it, at most 1023
constexpr uint32_t kLdsMaxNodes = 2048;
constexpr uint32_t kGMask = (1u << kGBits) - 1;
constexpr uint32_t kFMax = (1u << kFBits) - 1;
constexpr uint32_t kFhMax = (1u << kFhBits) - 1;                   // focalH beyond this -> MRP_LL_CAP_FOCAL (loud)
constexpr uint32_t kOpenKeyMask = (1u << (kGBits + kFBits)) - 1;   // applied to the key word
constexpr uint32_t kMaxHorizon = 1024;                             // t <= 1023  (g field has 10 bits)
constexpr uint32_t kMaxArenaNodes = 1u << 22;
constexpr uint32_t kConsLocalWords = 2048;                        // constraint words copied into the arena slot
constexpr uint32_t kNoParent = 0xFFFFFFFFu;
constexpr uint32_t kEmptyCell = 0xFFFFu;                           // "no agent here" in the path table
constexpr uint32_t kRingSlotBits = 11;                            // session mode: job-slot field of a ticket / completion entry
constexpr uint32_t kRingSlots = 1u << kRingSlotBits;              // job slots of a session (host Ring::kSlots == this)
constexpr uint32_t kRingSlotMask = kRingSlots - 1;
constexpr uint32_t kHeartbeatWord = 32;                           // ring_head[32]: host heartbeat (bumped on every submit / poll)

// status codes mirror include/mrp_ll.h
enum : int32_t { ST_OK = 0, ST_NO_SOLUTION = 1, ST_CAP_EXP = 2, ST_CAP_NODES = 3, ST_CAP_HORIZON = 4, ST_BAD = 5 };

struct DevJob {            // 96 bytes, 16-byte aligned
  uint32_t map_word_off;   // offset (uint32 words) of the obstacle bitmap inside the maps buffer
  uint32_t dimx, dimy;
  uint32_t words_per_row;  // ceil(dimx*dimy / 32)
  uint32_t sx, sy, gx, gy;
  uint32_t algo;
  float w;
  int32_t last_goal_constraint;  // Environment::m_lastGoalConstraint (ecbs.cpp:268-273), computed by the packer
  uint32_t n_vc, vc_off;   // vertex constraints: word = t<<16 | cell      (cell = y*dimx + x)
  uint32_t n_ec, ec_off;   // edge constraints:   word = t<<19 | cell<<3 | k   (k = index in Wait,Left,Right,Up,Down)
  uint32_t n_agents_pad;   // path table row length (multiple of 64; 0 = no focal context)
  uint32_t t_pad;          // path table rows (>= 1 when n_agents_pad > 0); row t_pad-1 repeats forever
  uint32_t path_off;       // offset (uint16 units) of the job's table  [t_pad][n_agents_pad]
  int64_t max_expansions;  // < 0: unlimited
  // ---- device-resident path store ----
  uint32_t ctx_flags;      // bit 0: the focal context is a list of path-store ids (n_ctx words at cons[path_off]) instead
                           // of a table at paths[path_off]; the workgroup builds the time-major table itself
  uint32_t n_ctx;          // agents in that list
  uint32_t store_out_id;   // kNoStoreSlot, or the path-store slot that also receives the result path
  uint32_t reserved;
};
constexpr uint32_t kNoStoreSlot = 0xFFFFFFFFu;
constexpr uint32_t kCtxById = 1u;
// ctx_flags bit 5: a ROOT CHAIN of an ECBS conflict tree (ecbs.hpp:118-136; mrp_ll_submit_root_chain): the workgroup plans
// agents t_pad .. n_ctx - 1 of one instance one after the other, each against the paths of the agents before it, and keeps
// the focal table in LDS between the searches.  cons[vc_off + a] = sx | sy << 8 | gx << 16 | gy << 24 of agent a,
// cons[vc_off + n_ctx + a] = its path-store slot (agents before t_pad: where their paths ARE; the others: where theirs go).
// Output (the job slot's host area, words): per planned agent eight words {status, cost, fmin, n_states, expanded,
// offset of its path in words, 0, 0}, then the paths (x | y << 8 halfwords).  DevResult: n_states = entries written,
// expanded = their sum.  The chain stops behind a search that found no path or ran into the expansion budget, and IN
// FRONT OF one that outgrows the compact tier (the caller submits that one as an ordinary job).  `reserved` = one past the
// last agent this job plans (a caller with many agents cuts the root step into jobs of bounded length).
constexpr uint32_t kCtxChain = 32u;
// ctx_flags bit 6 (MRP_LL_JOB_HEAVY): the caller knows that this search outgrows the compact tier (a root chain stopped in
// front of it): no attempt there — a front workgroup hands it to the heavy workgroups at once, an all-tier kernel starts
// it in the arena tier.
constexpr uint32_t kCtxHeavy = 64u;
constexpr uint32_t kChainEntryWords = 8;
constexpr uint32_t kChainMaxAgents = 128;                          // (what the compact tier's focal context holds: two 64-lane row loads)
constexpr uint32_t kChainRows = 64;                                // rows of the chain's focal table (the compact tier ends at t = 62)
// MRP_LL_SIPP with a device-resident table (mrp_ll_sipp_table_* in a session): ctx_flags bit 1.  The table lives in
// device memory at the 64-bit address (n_agents_pad | path_off << 32), in a fixed-capacity layout the search reads directly:
// two 64-byte rows per cell —
//   bounds[cells][16]  words 0 .. 14: interval i as  start | end << 16, both below kSippEndInf, end == kSippEndInf: INT_MAX;
//                      word 15: 0 = the default single interval [0, INT_MAX], n + 1 = n safe intervals (n <= kSippCap)
//   status[cells][16]  word i:  epoch << 24 | closed << 23 | node + 1   (a word of another epoch reads as "unseen")
// — so a neighbour cell costs an expansion one 64-byte sector that the search only reads (count and bounds: it stays in
// the L1 of the CU) and one that it also writes.  (Measured, scripts/r4_run23.sh: ONE 128-byte record per cell with count,
// bounds and status words interleaved is slower than the three separate arrays of round 3, 2.59 against 2.27 us per
// expansion — every status write takes the line with the bounds out of the L1.)
// cons[vc_off] holds only the DELTA since the table's previous job: ec_off & 0x7FFFFFFF records (bit 31: zero the table
// first), as  hdr[nRec]  (cell | count << 16; padded to a multiple of 4 words)  then  nRec x n_vc bounds words
// (n_vc = room per record of this job: the longest list among its records, as a power of two >= 4).
// n_ctx = the job's epoch (1..255).  One job per table in flight.  The workgroup finds the start interval itself
// (findSafeInterval, sipp.hpp:286-296) — the host's copy of the table may be behind the device's (kSippCommit).
constexpr uint32_t kTaNoGoal = 16u;                                // MRP_LL_ASTAR_TA: ctx_flags bit 4: the agent has no task; path_off =
                                                                   // word offset of the goal's heuristic table in the maps buffer
constexpr uint32_t kHeurWords = 512;                               // a heuristic table: [32][32] halfwords (0xFFFF: unreachable)
constexpr uint32_t kSippResident = 2u;
constexpr uint32_t kSippCommit = 8u;                               // ctx_flags bit 3: on success the workgroup adds the path's stays to the table
constexpr uint32_t kSippTierCommitFailed = 0x100u;                 // DevResult.tier flag: a stay did not fit (more than kSippCap intervals,
                                                                   // or no safe interval contains it): the host redoes the table
constexpr uint32_t kSippNoLds = 4u;                                // ctx_flags bit 2: keep nodes and open list in the arena (MRP_LL_SIPP_NO_LDS=1)
constexpr uint32_t kSippCap = 15;                                  // safe intervals per cell the resident layout holds
constexpr uint32_t kSippRowWords = 16;                             // words of a cell's bounds row and of its status row
constexpr uint32_t kSippEndInf = 0xFFFFu;                          // "ends at INT_MAX" in a bounds word; finite bounds are below it
constexpr uint32_t kSippEpochShift = 24, kSippEpochMax = 255;
constexpr uint32_t kSippStClosed = 1u << 23;

struct DevResult {         // 64 bytes
  int32_t status;
  int32_t cost;
  int32_t fmin;
  int32_t n_states;
  int64_t expanded;
  uint32_t nodes_created;
  uint32_t tier;
  uint32_t prof[8];        // -DMRP_LL_TRACE builds only: cycles in walk / pops / pushes / successors / rows / total,
                           // number of walks, nodes visited by walks (zero otherwise)
};

// Zero-copy staging: jobs / cons / paths / results / out_paths live in pinned host memory that the device reads and
// writes directly (no hipMemcpy commands per batch); each workgroup bulk-copies what its job needs into LDS or its
// arena slot at job start and writes the result back with coalesced stores.
struct LaunchParams {
  const DevJob* jobs;         // host-mapped
  DevResult* results;         // host-mapped
  uint16_t* out_paths;        // host-mapped [n_jobs][out_stride] : x | y<<8 for t = 0..n_states-1
  const uint32_t* maps;       // device: obstacle bitmaps
  const uint32_t* cons;       // host-mapped: constraint words of the batch
  const uint16_t* paths;      // host-mapped: path tables of the batch
  uint32_t* queue_head;       // device: monotonic job counter of this ticket (job = counter - queue_base)
  uint8_t* arena;             // HBM tier: per resident workgroup `arena_stride` bytes
  uint64_t arena_stride;
  uint32_t queue_base;
  uint32_t arena_scratch_off; // byte offset inside a slot of [path out][constraint copy][path-table copy]
  uint32_t arena_paths_bytes; // capacity of the path-table copy in the arena slot
  uint32_t lds_paths_bytes;   // capacity of the path-table copy in LDS
  uint32_t n_jobs;
  uint32_t out_stride;        // halfwords of the path scratch in an arena slot (= max_horizon)
  uint32_t out_host_stride;   // halfwords per job of out_paths (sessions: room for a root chain's output)
  uint32_t arena_nodes;       // node capacity in the HBM tier
  uint32_t arena_rows;        // bitmap rows (time steps) in the HBM tier == max_horizon
  uint32_t arena_row_words;   // words per bitmap row the arena was sized for (>= job.words_per_row)
  uint32_t lds_nodes;         // node capacity of the LDS tier (0 = LDS tier disabled)
  uint32_t lds_rows;          // bitmap rows held in LDS
  uint32_t lds_row_words;     // words per row the LDS layout was sized for
  volatile uint32_t* debug;   // host-mapped trace words (MRP_LL_DEBUG only; nullptr otherwise)
  // ---- session mode (mrp_ll_persistent_kernel): host-fed job ring in coherent pinned host memory ----
  uint32_t* ring_state;       // ticket rings, lane 0 then lane 1: host writes (generation << 11) | job slot for ticket
                              // (generation-1)*ring_size + index once the job in that slot is ready
  uint32_t* ring_done;        // [n_slots] device writes ticket+1 when that job's result is in results[slot]
  uint32_t* ring_stop;        // host sets != 0 to end the session
  uint32_t* ring_head;        // host: number of tickets published so far (what waiting workgroups poll)
  uint32_t* comp_ring;        // host [n_slots]: completion queue, entry = (generation << 11) | slot
  uint32_t* comp_count;       // device: completions so far (index into comp_ring)
  unsigned long long* sess_ticks; // device [2]: sum over workgroups of 100 MHz ticks spent in jobs / waiting for jobs
  uint32_t ring_size;         // ticket-ring entries of lane 0
  uint32_t ring_size1;        // ticket-ring entries of lane 1, the priority lane (stored after lane 0's)
  uint32_t n_slots;           // job slots (both lanes); at most 2048 (11-bit slot field)
  uint32_t ring_idle_limit_s; // a workgroup leaves when the host heartbeat (ring_head[kHeartbeatWord]) stood still this long
  // ---- device-resident path store: slot = [len][cell 0][cell 1]... halfwords, cell = y * dimx + x ----
  uint16_t* path_store;       // device (uncached allocation: written by one workgroup, read by others of a resident kernel)
  uint32_t path_store_stride; // halfwords per slot (0 = no store)
  uint32_t path_store_slots;
  // ---- heavy workgroups (A*-epsilon sessions): a second resident launch with the WIDE window (ll_compact.h) that takes
  // over the searches the front workgroups' compact tier cannot hold.  Device-side queue, device memory:
  //   heavy_ctr[0] = entries written so far (front workgroups: fetch-add), heavy_ctr[16] = tickets taken (heavy workgroups),
  //   heavy_q[t % kRingSlots] = done value << 32 | (t / kRingSlots + 1) << 11 | job slot   (one 8-byte store)
  // heavy_alive (host-mapped): word w = 1 once heavy workgroup w runs (the host checks residency before it relies on them).
  unsigned long long* heavy_q;
  uint32_t* heavy_ctr;
  uint32_t* heavy_alive;
  uint32_t heavy_wgs;         // 0: no heavy launch — the resident kernel serves every tier itself
};

}  // namespace mrp
