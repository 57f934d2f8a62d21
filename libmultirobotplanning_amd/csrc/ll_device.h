// Device-side job/result records shared by the HIP kernels (ll_kernel.hip) and the host packer (mrp_ll_host.cpp).
#pragma once
#include <stdint.h>

namespace mrp {

// ---- packed heap entry (uint64) ------------------------------------------------------------------------------
//   [59:42] FH_MAXV - focalH   [41:31] F_MAXV - f   [30:20] g   [19:0] node id
// A larger (entry >> 20) is a BETTER node in the reference's orders:
//   open  (a_star_epsilon.hpp:312-323, a_star.hpp:168-179): lowest f, then highest g      -> bits [41:20]
//   focal (a_star_epsilon.hpp:346-366): lowest focalH, then lowest f, then highest g      -> bits [59:20]
// Entries with equal keys compare EQUAL (the id bits never take part), exactly like the reference's comparators;
// which of two equal entries comes out first is decided by the heap layout, which the kernels replay verbatim.
constexpr uint32_t kIdBits = 20;
constexpr uint32_t kIdMask = (1u << kIdBits) - 1;
constexpr uint32_t kGBits = 11, kFBits = 11, kFhBits = 18;
constexpr uint32_t kFMax = (1u << kFBits) - 1;
constexpr uint32_t kFhMax = (1u << kFhBits) - 1;
constexpr uint32_t kOpenKeyMask = (1u << (kGBits + kFBits)) - 1;  // applied to (entry >> kIdBits)
constexpr uint32_t kMaxHorizon = 1024;                             // t < 1024  (g field has 11 bits)
constexpr uint32_t kNoParent = 0xFFFFFFFFu;
constexpr uint32_t kEmptyCell = 0xFFFFu;                           // "no agent here" in the path table

// status codes mirror include/mrp_ll.h
enum : int32_t { ST_OK = 0, ST_NO_SOLUTION = 1, ST_CAP_EXP = 2, ST_CAP_NODES = 3, ST_CAP_HORIZON = 4, ST_BAD = 5 };

struct DevJob {            // 80 bytes, 16-byte aligned
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
};

struct DevResult {         // 32 bytes
  int32_t status;
  int32_t cost;
  int32_t fmin;
  int32_t n_states;
  int64_t expanded;
  uint32_t nodes_created;
  uint32_t tier;
};

struct LaunchParams {
  const DevJob* jobs;
  DevResult* results;
  uint16_t* out_paths;        // [n_jobs][out_stride] : x | y<<8 for t = 0..n_states-1
  const uint32_t* maps;       // obstacle bitmaps
  const uint32_t* cons;       // constraint words of the batch
  const uint16_t* paths;      // path tables of the batch
  uint32_t* queue_head;       // job counter of this launch (zeroed by the host before the launch)
  uint8_t* arena;             // HBM tier: per resident workgroup `arena_stride` bytes
  uint64_t arena_stride;
  uint32_t n_jobs;
  uint32_t out_stride;
  uint32_t arena_nodes;       // node capacity in the HBM tier
  uint32_t arena_rows;        // bitmap rows (time steps) in the HBM tier == max_horizon
  uint32_t arena_row_words;   // words per bitmap row the arena was sized for (>= job.words_per_row)
  uint32_t lds_nodes;         // node capacity of the LDS tier (0 = LDS tier disabled)
  uint32_t lds_rows;          // bitmap rows held in LDS
  uint32_t lds_row_words;     // words per row the LDS layout was sized for
  volatile uint32_t* debug;   // host-mapped trace words (MRP_LL_DEBUG only; nullptr otherwise)
};

}  // namespace mrp
