/*
 * mrp_hl.h — host-side conflict-tree drivers (libmrp_hl.so) that CALL the low-level C-ABI of mrp_ll.h.
 *
 * These are the callers of the hot path, restated so that many MAPF instances advance in lock-step and every round
 * hands ALL ready low-level searches to the GPU in one mrp_ll_submit:
 *   CBS::search   include/libMultiRobotPlanning/cbs.hpp:85-172   (root: N independent searches; then 2 per CT node)
 *   ECBS::search  include/libMultiRobotPlanning/ecbs.hpp:109-288 (root: chain of N searches, each seeing the paths
 *                 planned so far; then 2 per CT node; HL open by cost, HL focal by (focalHeuristic, cost))
 * with the HL-side Environment methods of example/ecbs.cpp: getFirstConflict :401-452, createConstraintsFromConflict
 * :454-472, focalHeuristic :315-350.  The high-level heaps replay boost::heap::d_ary_heap exactly (tie order decides
 * which CT node is expanded), so cost / makespan / highLevelExpanded / lowLevelExpanded equal the reference's
 * `statistics:` block (example/ecbs.cpp:594-599) — checked against the oracle in tests/.
 *
 * Also here: the seeded synthetic-instance generator used by bench.py and the YAML subset
 * reader / schedule writer of example/ecbs.cpp:554-574,584-617.
 */
#ifndef MRP_HL_H
#define MRP_HL_H

#include <stdint.h>

#include "mrp_ll.h"

#ifdef __cplusplus
extern "C" {
#endif

#define MRP_HL_CBS 0
#define MRP_HL_ECBS 1

/* mrp_hl_solution.status */
#define MRP_HL_SOLVED 0       /* search() returned true                                                        */
#define MRP_HL_NO_SOLUTION 1  /* search() returned false (a root search failed or the CT open list ran empty)   */
#define MRP_HL_CAP 2          /* a harness cap was hit (the reference has none and would keep running)          */
#define MRP_HL_LL_ERROR 3     /* a low-level job came back with a capacity status (MRP_LL_CAP_NODES/HORIZON/..) */

typedef struct mrp_hl_instance {
  int32_t dimx, dimy;
  int32_t n_obstacles;
  const int32_t* obstacles_xy; /* [n][2]  (map.obstacles, ecbs.cpp:564-566) */
  int32_t n_agents;
  const int32_t* starts_xy;    /* [n][2]  (agents[].start, ecbs.cpp:571)     */
  const int32_t* goals_xy;     /* [n][2]  (agents[].goal,  ecbs.cpp:573)     */
} mrp_hl_instance;

typedef struct mrp_hl_solution {
  int32_t status;
  int32_t n_ll_searches;          /* low-level searches issued for this instance                                */
  int64_t cost;                   /* statistics.cost      (sum of PlanResult::cost, ecbs.cpp:586-591)            */
  int64_t makespan;               /* statistics.makespan                                                        */
  int64_t high_level_expanded;    /* statistics.highLevelExpanded                                               */
  int64_t low_level_expanded;     /* statistics.lowLevelExpanded  (the metric's numerator)                      */
  int32_t* path_len;              /* caller buffer [n_agents] or NULL                                           */
  int32_t* paths_xy;              /* caller buffer [n_agents][path_cap][2] or NULL                              */
  int32_t path_cap;
  int32_t reserved;
  /* FNV-1a (64 bit) of the schedule of a SOLVED instance — for every agent in order: the bytes x, y of every state of its
   * path, then the byte 0xFF — computed from the full paths whether or not paths_xy holds (all of) them: a caller that
   * checks a million schedules compares eight bytes each.  0 when status != MRP_HL_SOLVED. */
  uint64_t schedule_digest;
} mrp_hl_solution;

typedef struct mrp_hl_options {
  int32_t algo;                   /* MRP_HL_CBS | MRP_HL_ECBS                                                    */
  float w;                        /* ECBS suboptimality bound (binary32, ecbs.cpp:530,536)                        */
  int64_t max_ll_expansions;      /* per instance, summed over its searches; < 0 unlimited                        */
  int64_t max_hl_expansions;      /* per instance; < 0 unlimited                                                  */
  int32_t n_threads;              /* host worker threads == instance groups in flight (0 = default)               */
  int32_t mode;                   /* 0 = session (resident kernel, every instance advances on its own; default),   */
                                  /* 1 = rounds (one launch per round of ready searches; simpler, slower)          */
} mrp_hl_options;

typedef struct mrp_hl_batch_stats {
  double wall_seconds;            /* timed region: all conflict-tree searches of the batch, first submit to last result */
  int64_t rounds;                 /* mrp_ll_submit calls                                                          */
  int64_t ll_searches;
  int64_t ll_expansions;          /* of the searches the conflict trees consumed (== sum of solutions' low_level_expanded) */
  int64_t solved;
  /* host-side time summed over worker threads (diagnostic): building jobs, inside mrp_ll_search_batch, consuming results */
  double build_seconds, ll_call_seconds, consume_seconds;
  /* look-ahead of the conflict-tree machines (session mode; MRP_HL_SPEC): searches issued before their CT node was popped,
   * and the expansions of searches that were run but whose node was never popped (work, not part of ll_expansions) */
  int64_t speculative_searches, wasted_ll_expansions;
  /* ECBS sessions: instances whose root node had no conflict — found by the workgroup that ran the root chain
   * (mrp_ll.h MRP_LL_JOB_ROOT_CHAIN) — and whose solution was written without a conflict-tree object */
  int64_t root_solved;
} mrp_hl_batch_stats;

/* One engine context per calling thread is created internally for every worker thread on `device`. */
int mrp_hl_solve_batch(int32_t device, const mrp_hl_options* opt, int32_t n_instances, const mrp_hl_instance* instances,
                       mrp_hl_solution* solutions, mrp_hl_batch_stats* stats);

/* Persistent form for benchmarking: engines (and their arenas) are created once and reused across batches. */
typedef struct mrp_hl_solver mrp_hl_solver;
int mrp_hl_solver_create(int32_t device, int32_t n_threads, const mrp_ll_options* ll_opt, mrp_hl_solver** out);
void mrp_hl_solver_destroy(mrp_hl_solver* s);
int mrp_hl_solver_solve(mrp_hl_solver* s, const mrp_hl_options* opt, int32_t n_instances,
                        const mrp_hl_instance* instances, mrp_hl_solution* solutions, mrp_hl_batch_stats* stats);
/* Two-step form: preload uploads the instances' static maps to HBM (the reference builds its Environment before it
 * starts its Timer, example/ecbs.cpp:576-582) and fixes the instance -> worker assignment; solve_preloaded then runs the
 * conflict-tree searches only.  The instance arrays must stay valid until mrp_hl_preloaded_free. */
typedef struct mrp_hl_preloaded mrp_hl_preloaded;
int mrp_hl_solver_preload(mrp_hl_solver* s, int32_t n_threads, int32_t n_instances, const mrp_hl_instance* instances,
                          mrp_hl_preloaded** out);
int mrp_hl_solver_solve_preloaded(mrp_hl_solver* s, const mrp_hl_options* opt, mrp_hl_preloaded* p,
                                  mrp_hl_solution* solutions, mrp_hl_batch_stats* stats);
/* Several preloaded batches (of this solver, preloaded with the same n_threads) as ONE stream: the worker threads draw
 * batch 0's instances, then batch 1's, ... from one pool with no barrier between batches, so the dependent chains that end
 * a batch — deep conflict trees (ecbs.hpp:151-285 is a sequential loop per instance), searches that run to the harness cap —
 * overlap with the root searches of the next batch instead of leaving the device idle.  Instances are independent
 * (one conflict tree each), so solutions[b][k] is exactly what mrp_hl_solver_solve_preloaded(batches[b]) writes; `stats`
 * is the aggregate over the stream (wall_seconds: the whole call).  n_batches = 1 is mrp_hl_solver_solve_preloaded.
 * Several batches need the session driver (mrp_hl_options.mode 0). */
int mrp_hl_solver_solve_stream(mrp_hl_solver* s, const mrp_hl_options* opt, int32_t n_batches,
                               mrp_hl_preloaded* const* batches, mrp_hl_solution* const* solutions,
                               mrp_hl_batch_stats* stats);
void mrp_hl_preloaded_free(mrp_hl_preloaded* p);
int mrp_hl_solver_ll_stats(mrp_hl_solver* s, mrp_ll_stats* out, int32_t reset); /* summed over its engines */
const char* mrp_hl_solver_last_error(const mrp_hl_solver* s);

/* Prioritized planning with SIPP (example/mapf_prioritized_sipp.cpp:214-270): agents are planned one after the other,
 * each against the collision intervals left by the agents before it; an agent that cannot be planned is skipped.
 * Every round plans the next agent of ALL instances in one MRP_LL_SIPP batch.
 * Per instance: cost = statistics.cost (sum over planned agents); planned[a] in {0,1}; n_states[a];
 * states_xyt [n_agents][state_cap][3] = x, y, t (the schedule the reference writes, :256-260). */
typedef struct mrp_hl_sipp_solution {
  int64_t cost;
  int64_t low_level_expanded;
  int32_t n_planned;
  int32_t status;       /* 0: every agent got the reference's answer (planned or "not found"); otherwise the MRP_LL_*
                         * capacity status (expansion cap, node arena, horizon) of the search that stopped THIS
                         * instance — its later agents are not planned; other instances of the batch are unaffected */
  int32_t* planned;     /* caller buffer [n_agents] */
  int32_t* n_states;    /* caller buffer [n_agents] */
  int32_t* states_xyt;  /* caller buffer [n_agents][state_cap][3] or NULL */
  int32_t state_cap;
  int32_t reserved2;
} mrp_hl_sipp_solution;
int mrp_hl_solver_prioritized_sipp(mrp_hl_solver* s, int32_t n_instances, const mrp_hl_instance* instances,
                                   mrp_hl_sipp_solution* solutions, mrp_hl_batch_stats* stats);

/* ---- one conflict tree, stepped by the caller ------------------------------------------------------------------
 * The same state machine the batch drivers run (CBS::search cbs.hpp:85-172 / ECBS::search ecbs.hpp:109-288 cut at the
 * low-level calls), exposed so that a caller can decide WHERE each low-level search runs — e.g. the searches of one
 * round sharded over several GPUs (libmultirobotplanning_amd/ct_sharded.py).  Pending requests come in
 * groups (the root step, or the two children of one conflict-tree node; with spec_width > 1 also the children of the
 * nodes that will probably be popped next); a group is answered as a whole, groups in any order.  The result of an
 * instance that is SOLVED (or has no solution) does not depend on spec_width or on the order of delivery.  An instance
 * that ends in MRP_HL_CAP reports MRP_HL_CAP either way, but its low_level_expanded / n_ll_searches may differ with
 * spec_width: a pre-computed search is issued with the budget left at that moment, before the searches in front of
 * it have been accounted. */
typedef struct mrp_hl_ct mrp_hl_ct;
int mrp_hl_ct_create(const mrp_hl_instance* instance, const mrp_hl_options* opt, int32_t map_id, int32_t spec_width,
                     mrp_hl_ct** out);
void mrp_hl_ct_destroy(mrp_hl_ct* ct);
int32_t mrp_hl_ct_n_requests(const mrp_hl_ct* ct);
/* Request k as a low-level job (map_id as given to create; its arrays stay valid until the next mrp_hl_ct_deliver). */
int mrp_hl_ct_request(const mrp_hl_ct* ct, int32_t k, mrp_ll_job* job, int32_t* group, int32_t* slot);
/* The results of every request of `group`, in slot order (status, cost, fmin, expanded, n_states, states_txy). */
int mrp_hl_ct_deliver(mrp_hl_ct* ct, int32_t group, int32_t n, const mrp_ll_result* results);
int32_t mrp_hl_ct_done(const mrp_hl_ct* ct);
int mrp_hl_ct_solution(const mrp_hl_ct* ct, mrp_hl_solution* out);

/* One round of a conflict tree whose searches are sharded over `world` ranks (ct_sharded.py): every rank
 * holds the same tree, group j of the pending requests belongs to rank j % world.
 *   mrp_hl_ct_round_mine   runs THIS rank's requests on `ll` (mrp_ll_search_batch) and packs their results into `rows`, one
 *                          row of 8 + max_states int32 words per search:
 *                            group, slot, status, cost, fmin, expanded (low 31 bits), expanded >> 31, n_states,
 *                            then one word x | y << 16 per state.
 *                          rows_per_rank[world] receives how many rows every rank produces this round (the same on all
 *                          ranks: the shape of the all-gather); returns the number of rows written (<= cap_rows), or
 *                          a negative MRP_LL_E_* — in which case the rank still writes ONE row with group = INT32_MIN, so that
 *                          the collective can take place and every rank sees the failure.
 *   mrp_hl_ct_deliver_rows takes the gathered rows of all ranks ([world][rows_stride][8 + max_states]) and delivers the
 *                          groups in their pending order.  MRP_LL_E_DEVICE if any rank reported a failure. */
int32_t mrp_hl_ct_round_mine(mrp_hl_ct* ct, mrp_ll_ctx* ll, int32_t rank, int32_t world, int32_t max_states, int32_t* rows,
                             int32_t cap_rows, int32_t* rows_per_rank);
int mrp_hl_ct_deliver_rows(mrp_hl_ct* ct, const int32_t* gathered, int32_t world, int32_t rows_stride, int32_t max_states);

/* BASELINE.json configs[0] — `./a_star` on a text map (example/a_star.cpp:72-125,190-191) — is host plumbing: a single
 * 2-D A* (AStar::search a_star.hpp:63-161, neighbours Up, Down, Left, Right, unit costs, Manhattan heuristic), run on
 * the CPU with the reference's heap tie-breaks.  obstacle_mask[y * dimx + x] != 0 = '#'.  Returns the number of states
 * of the path (written to states_xy [cap][2] up to cap), 0 = "Planning NOT successful!", -1 = bad argument. */
int32_t mrp_hl_astar_grid2d(int32_t dimx, int32_t dimy, const uint8_t* obstacle_mask, int32_t start_x, int32_t start_y,
                            int32_t goal_x, int32_t goal_y, int32_t* states_xy, int32_t cap, int32_t* cost,
                            int64_t* expanded);

/* Seeded synthetic "32x32_obst204-shaped" instance: obstacles uniform without replacement, agents with
 * distinct starts and distinct goals, every goal in the start's 4-connected free component. splitmix64(seed).
 * Buffers: obstacles_xy [n_obstacles][2], starts_xy / goals_xy [n_agents][2]. Returns 0, or -1 if impossible. */
int mrp_hl_generate_instance(uint64_t seed, int32_t dimx, int32_t dimy, int32_t n_obstacles, int32_t n_agents,
                             int32_t* obstacles_xy, int32_t* starts_xy, int32_t* goals_xy);
/* The same for seeds seed0 .. seed0 + n - 1 in one call (host threads): instance k fills obstacles_xy[k][n_obstacles][2],
 * starts_xy[k][n_agents][2], goals_xy[k][n_agents][2].  Returns 0, or -1 if any instance was impossible. */
int mrp_hl_generate_instances(uint64_t seed0, int32_t n, int32_t dimx, int32_t dimy, int32_t n_obstacles,
                              int32_t n_agents, int32_t* obstacles_xy, int32_t* starts_xy, int32_t* goals_xy);

#ifdef __cplusplus
}
#endif
#endif /* MRP_HL_H */
