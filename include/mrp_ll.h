/*
 * mrp_ll.h — C ABI of the MI355X low-level search engine (libmrp_ll.so).
 *
 * Drop-in boundary for ONE path of Sartor02/libMultiRobotPlanning: the per-agent low-level search that the
 * conflict-tree searches call once per agent and once per conflict-tree child.  The reference has no FFI at this
 * point — the boundary is a compile-time template concept — so each entry point below names the reference
 * interface it replaces:
 *
 *   LowLevelSearch_t::search(const State& start, PlanResult& out)
 *       CBS  : AStar<State,Action,Cost,LowLevelEnvironment>            cbs.hpp:248, called at cbs.hpp:99-101,155-157
 *       ECBS : AStarEpsilon<State,Action,Cost,LowLevelEnvironment>(w)  ecbs.hpp:421-422, called at ecbs.hpp:126-129,265-268
 *       search bodies: a_star.hpp:63-161, a_star_epsilon.hpp:86-285
 *   LowLevelEnvironment(env, agentIdx, constraints[, solution])        cbs.hpp:209-217, ecbs.hpp:365-375
 *       -> Environment::setLowLevelContext                              example/ecbs.cpp:264-274, example/cbs.cpp:266-276
 *   Environment::{admissibleHeuristic,isSolution,getNeighbors,stateValid,transitionValid,
 *                 focalStateHeuristic,focalTransitionHeuristic}         example/ecbs.cpp:276-312,352-399,497-510
 *   PlanResult{states,actions,cost,fmin}                                planresult.hpp:18-27
 *   Environment::onExpandLowLevelNode (the metric's numerator)          example/ecbs.cpp:476-479
 *
 * Because a device cannot call back into C++ templates, everything the search reads is passed explicitly in
 * mrp_ll_job: the static map, the agent's start/goal, its vertex/edge constraint sets and (ECBS) the other agents'
 * current paths that the focal heuristics consult.  Jobs of one batch are independent; results are bit-identical to
 * the reference's integers (cost, fmin, path, expansion count) for the grid MAPF Environment of example/ecbs.cpp and
 * example/cbs.cpp.  All buffers are caller-owned; no pointer outlives the call that received it (for
 * mrp_ll_submit: the matching mrp_ll_wait).
 *
 * A context is NOT thread-safe; use one per host thread.  There is no CPU fallback: every entry point fails with
 * MRP_LL_E_DEVICE when no HIP device is usable.
 */
#ifndef MRP_LL_H
#define MRP_LL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes of the API calls ------------------------------------------------------------------------- */
#define MRP_LL_SUCCESS 0
#define MRP_LL_E_INVALID (-1)  /* bad argument (NULL pointer, unknown map id, dimension out of range, ...)   */
#define MRP_LL_E_DEVICE (-2)   /* HIP error / no device; see mrp_ll_last_error()                            */
#define MRP_LL_E_NOMEM (-3)
#define MRP_LL_E_BUSY (-4)     /* no free ticket for mrp_ll_submit                                          */

/* ---- search algorithm of a job ----------------------------------------------------------------------------- */
#define MRP_LL_ASTAR 0     /* a_star.hpp AStar::search          (CBS low level)  */
#define MRP_LL_ASTAR_EPS 1 /* a_star_epsilon.hpp AStarEpsilon   (ECBS low level) */
#define MRP_LL_SIPP 2      /* sipp.hpp SIPP::search over the grid Environment of example/mapf_prioritized_sipp.cpp;  */
                           /* a batch (or a session, mrp_ll_session_begin_sipp) holds either SIPP jobs or A-star     */
                           /* jobs, not both                                                                         */

#define MRP_LL_ASTAR_TA 3  /* a_star.hpp AStar::search over the Environment of example/cbs_ta.cpp:283-372 — the low level of */
                           /* the task-assignment callers (cbs_ta.hpp:106-109,155-158,196-199; ecbs_ta's Environment is the */
                           /* same): optional goal (MRP_LL_JOB_NO_GOAL), h = an uploaded shortest-path table               */
                           /* (mrp_ll_upload_heuristic), Wait costs 0 at the goal, so g != time and decrease-key is live.  */
                           /* Two tiers, like the other searches: the LDS tier (maps up to 32 x 32, at most 64 vertex and  */
                           /* 64 edge constraints, time steps <= 61, f <= 254, 1023 open nodes) and, for everything beyond */
                           /* it, the arena tier (any map the context accepts, any number of constraints; limits:           */
                           /* mrp_ll_options.arena_nodes / max_horizon — MRP_LL_CAP_NODES / MRP_LL_CAP_HORIZON — and as many */
                           /* time steps as arena_nodes x 4 status words hold: 512 on a 32 x 32 map by default).            */

/* ---- per-job status (mrp_ll_result.status) ----------------------------------------------------------------- */
#define MRP_LL_OK 0             /* search() returned true                                                    */
#define MRP_LL_NO_SOLUTION 1    /* search() returned false: open list exhausted (a_star.hpp:160)              */
#define MRP_LL_CAP_EXPANSIONS 2 /* job.max_expansions exceeded (the reference has no cap and would keep going) */
#define MRP_LL_CAP_NODES 3      /* per-search node arena exhausted (mrp_ll_options.arena_nodes)               */
#define MRP_LL_CAP_HORIZON 4    /* a state beyond mrp_ll_options.max_horizon would have been generated         */
#define MRP_LL_BAD_JOB 5        /* job rejected on the host (unknown map, start/goal outside the grid, ...)    */
#define MRP_LL_PATH_TRUNCATED 6 /* solved, but result.states_cap was too small; cost/fmin/expanded are valid    */
#define MRP_LL_NOT_RUN 8        /* root chain (MRP_LL_JOB_ROOT_CHAIN): the chain ended before this agent's search            */
#define MRP_LL_CAP_FOCAL 7      /* a node's focal value exceeded the 11-bit key field (2047 accumulated conflicts)  */

/* Action codes == enum class Action of example/ecbs.cpp:49-55 */
#define MRP_LL_ACT_UP 0
#define MRP_LL_ACT_DOWN 1
#define MRP_LL_ACT_LEFT 2
#define MRP_LL_ACT_RIGHT 3
#define MRP_LL_ACT_WAIT 4

typedef struct mrp_ll_ctx mrp_ll_ctx;

typedef struct mrp_ll_options {
  int32_t device;          /* HIP device ordinal                                                             */
  int32_t n_tickets;       /* batches that may be in flight at once (0 = default 4)                           */
  int32_t slots;           /* resident searches per in-flight batch (0 = default 1024)                        */
  int32_t arena_nodes;     /* HBM node capacity per search before MRP_LL_CAP_NODES (0 = default 131072)       */
  int32_t max_horizon;     /* largest state time + 1 (0 = default 512, max 1024)                              */
  int32_t max_cells;       /* largest dimx*dimy accepted by mrp_ll_upload_map (0 = default 4096, max 65025)   */
  int32_t lds_nodes;       /* 2 x open-list entries of the LDS-resident fast tier (0 = default 2048, <0 = no tier) */
  int32_t reserved;
} mrp_ll_options;

/* One low-level search == one LowLevelEnvironment + one LowLevelSearch_t::search call of the reference. */
typedef struct mrp_ll_job {
  int32_t map_id;  /* from mrp_ll_upload_map                                                                  */
  int32_t algo;    /* MRP_LL_ASTAR | MRP_LL_ASTAR_EPS                                                         */
  float w;         /* suboptimality bound, binary32 exactly as AStarEpsilon::m_w (a_star_epsilon.hpp:386)      */
  int32_t agent_idx; /* index of this agent inside `path_*` (its own entry is ignored, ecbs.cpp:287)          */
  int32_t start_x, start_y; /* start State is (time 0, x, y)  (ecbs.cpp:571)                                  */
  int32_t goal_x, goal_y;   /* m_goals[agentIdx]              (ecbs.cpp:573)                                  */
  int32_t n_vertex_constraints;
  const int32_t* vertex_constraints; /* [n][3] = time, x, y           (ecbs.cpp:108-112)                      */
  int32_t n_edge_constraints;
  const int32_t* edge_constraints;   /* [n][5] = time, x1, y1, x2, y2 (ecbs.cpp:140-147)                      */
  /* ECBS focal context: the CT node's `solution` vector (ecbs.hpp:381-389). Ignored for MRP_LL_ASTAR. */
  int32_t n_agents;                  /* solution.size(); 0 = no context                                       */
  const int32_t* path_len;           /* [n_agents] states per path; 0 = empty path, skipped (ecbs.cpp:287)    */
  const int32_t* const* path_xy;     /* [n_agents] -> [path_len][2] = x, y at time 0,1,2,...                  */
  int64_t max_expansions;            /* < 0: unlimited                                                        */
  /* MRP_LL_SIPP only — SIPP::setCollisionIntervals (sipp.hpp:82-85,245-284), one entry per location; a location
   * given twice keeps the later list.  Ignored by the other algorithms (leave zero). */
  int32_t n_collision_locations;
  const int32_t* collision_xy;        /* [n][2]   x, y                                                        */
  const int32_t* collision_count;     /* [n]      intervals of that location                                  */
  const int32_t* collision_intervals; /* [sum][2] start, end (inclusive; end may be INT32_MAX)                 */
  /* The fork's extra search arguments (both default to 0 in the reference):
   *   MRP_LL_ASTAR     : AStar::search(start, solution, initialCost)   a_star.hpp:63-64,78,100 — the start node's gScore;
   *                      cost and fmin come back offset by it (a start that already is the goal keeps fmin = h(start),
   *                      because the start node's fScore is pushed without it, a_star.hpp:78)
   *   MRP_LL_SIPP      : SIPP::search(start, waitAction, solution, startTime)  sipp.hpp:92-103 — the safe interval of the
   *                      start cell is looked up at this time, state times are absolute, cost = arrival - startTime
   *   MRP_LL_ASTAR_EPS : AStarEpsilon::search has no such argument: a non-zero value is rejected (MRP_LL_BAD_JOB) */
  int32_t initial_cost;
  /* MRP_LL_SIPP with sipp_table: 1 = when the search succeeds, the stays of the path it found become collision
   * intervals of the table — [t_k, t_{k+1} - 1] on the k-th cell of the path, [t_last, INT32_MAX] on the last — exactly
   * the mrp_ll_sipp_table_add calls a prioritized planner makes with the solution (mapf_prioritized_sipp.cpp:237-246),
   * done by the engine (in a session: by the workgroup that ran the search, on the device-resident table).  The table
   * then has ONE user at a time: the next job on it is submitted after this one's result has been collected. */
  int32_t sipp_commit;
  /* MRP_LL_SIPP only, optional: an incrementally maintained table (mrp_ll_sipp_table_*) instead of the collision_*
   * arrays above (which are then ignored).  The table must stay unchanged until the job has been submitted. */
  const struct mrp_ll_sipp_table* sipp_table;
  /* Device-resident path store (mrp_ll_path_store_reserve).  A conflict-tree child differs from its
   * parent in ONE path (ecbs.hpp:253-263 copies all N), and every path was produced by a search of this engine: with
   * result_path_id >= 0 the kernel ALSO leaves the result path in that store slot, and a later MRP_LL_ASTAR_EPS job names
   * the CT node's paths by their slots — path_ids[n_agents], -1 = no path / the searching agent itself — instead of
   * shipping them (path_xy may then be NULL; path_len is still required).  Slot ids are managed by the caller: a slot
   * may be reused once no unfinished job names it.
   * result_path_id is honoured only when `flags` has MRP_LL_JOB_STORE_RESULT: a zero-initialised job (`mrp_ll_job j{}`,
   * memset) stores nothing and touches no slot. */
  const int32_t* path_ids;
  int32_t result_path_id;  /* with MRP_LL_JOB_STORE_RESULT: the slot (0 .. n_slots-1) that also receives the result path */
  int32_t flags;           /* MRP_LL_JOB_* bits; 0 = none */
  /* MRP_LL_ASTAR_TA only: the shortest-path table of this job's goal cell (mrp_ll_upload_heuristic); ignored with
   * MRP_LL_JOB_NO_GOAL.  (Zero-initialised jobs of the other algorithms never look at it.) */
  int32_t heuristic_id;
  int32_t chain_count;     /* MRP_LL_JOB_ROOT_CHAIN: plan at most this many agents (0 = all from agent_idx on): a caller with many
                            * agents cuts the root step into jobs of bounded length; the agents behind come back MRP_LL_NOT_RUN */
  /* MRP_LL_JOB_ROOT_CHAIN only: [n_agents][4] = start x, start y, goal x, goal y of every agent of the instance. */
  const int32_t* chain_starts_goals_xy;
} mrp_ll_job;

#define MRP_LL_JOB_STORE_RESULT 1 /* mrp_ll_job.flags: also leave the result path in path-store slot result_path_id */
/* MRP_LL_JOB_ROOT_CHAIN (sessions of MRP_LL_ASTAR_EPS with a path store; maps up to 32 x 32, at most 128 agents, and room
 * for the 64-row focal table of all agents in the LDS window — mrp_ll_configure_tiers' lds_path_bytes): ONE job
 * that is the root step of an ECBS conflict tree (ecbs.hpp:118-136) from agent `agent_idx` on: agent a = agent_idx ..
 * n_agents - 1 is planned with no constraints against the paths of the agents 0 .. a - 1, one after the other, by one
 * workgroup that keeps the focal context in LDS.  path_ids[n_agents] names the path-store slot of EVERY agent: where the
 * paths of the agents before agent_idx are, and where the others' paths go (all >= 0).  max_expansions is the budget of
 * the whole chain (each search gets what the ones before it left, like a caller that subtracts `expanded` itself).
 * The job's result has chain_results -> [n_agents - agent_idx] ordinary results (each with its own buffers), filled in
 * agent order: the chain ends BEHIND an agent whose search found no path or exceeded the budget, and IN FRONT OF one
 * whose search outgrows the LDS tier — that one and every later agent come back as MRP_LL_NOT_RUN and are the caller's to
 * submit as ordinary jobs (or as another chain).  The job's own n_states = results filled in, expanded = their sum.
 * Every filled-in result is exactly what the ordinary job for that agent would have returned.
 * When agent_idx == 0 and every agent got a path, the workgroup — which holds the whole root solution in LDS — also scans
 * it for conflicts (Environment::getFirstConflict ecbs.cpp:401-452, focalHeuristic ecbs.cpp:315-350): the job's own
 * cost = number of conflicts (0: the root node is the solution, the conflict tree has nothing to do) and fmin = the first
 * one as  time << 24 | type << 16 | agent1 << 8 | agent2  (type 0 vertex / 1 edge), or -1 if there is none; both are -1
 * when the scan did not take place. */
#define MRP_LL_JOB_ROOT_CHAIN 4
/* MRP_LL_JOB_HEAVY (hint, MRP_LL_ASTAR_EPS): the caller knows that this search outgrows the LDS tier every search starts in
 * (e.g. a root chain ended in front of it): no attempt is made there.  Results never depend on it. */
#define MRP_LL_JOB_HEAVY 8
#define MRP_LL_JOB_NO_GOAL 2      /* mrp_ll_job.flags, MRP_LL_ASTAR_TA: the agent has no task (cbs_ta.cpp:283-319: h = 0, every
                                   * cell ends the search once time > the agent's last vertex constraint, every Wait is free) */

typedef struct mrp_ll_result {
  int32_t status;   /* MRP_LL_OK ... */
  int32_t cost;     /* PlanResult::cost  (valid when status == MRP_LL_OK / MRP_LL_PATH_TRUNCATED)             */
  int32_t fmin;     /* PlanResult::fmin  (a_star_epsilon.hpp:210, a_star.hpp:104)                              */
  int32_t n_states; /* PlanResult::states.size(); actions.size() == n_states - 1                              */
  int64_t expanded; /* calls of onExpandNode during this search (counts the goal pop)                         */
  int32_t* states_txy; /* caller buffer [states_cap][3] = time, x, y ; may be NULL                            */
  int32_t* actions;    /* caller buffer [states_cap]    = MRP_LL_ACT_* ; may be NULL                          */
  int32_t states_cap;
  int32_t tier;     /* 0 = finished in the LDS tier, 1 = run by the arena tier, 2 = by a heavy workgroup's wide LDS tier (diagnostic) */
  int32_t* action_costs; /* caller buffer [states_cap] or NULL: PlanResult::actions[k].second (always 1 for
                          * MRP_LL_ASTAR / _EPS; 0 for a Wait at the goal with MRP_LL_ASTAR_TA; Wait durations for
                          * MRP_LL_SIPP, sipp.hpp:105-128)                                                     */
  struct mrp_ll_result* chain_results; /* MRP_LL_JOB_ROOT_CHAIN jobs only: caller array [n_agents - agent_idx]  */
} mrp_ll_result;

typedef struct mrp_ll_stats {
  int64_t launches;          /* kernel launches so far                                                        */
  int64_t jobs;              /* searches run                                                                  */
  int64_t expansions;        /* low-level expansions summed over all searches                                 */
  int64_t nodes_created;     /* heap pushes summed over all searches                                          */
  int64_t migrated;          /* searches that left the LDS tier                                               */
  double kernel_ms;          /* sum of hipEvent-measured kernel durations (on the launching stream)           */
  double h2d_ms, d2h_ms;     /* hipEvent-measured copy durations                                              */
  double session_busy_ms, session_idle_ms; /* session mode: sum over resident workgroups of time in jobs / waiting */
  int64_t session_active_wgs;              /* session mode: workgroups that ran at least one job (summed over sessions) */
  double pack_ms, unpack_ms; /* host time spent packing jobs (mrp_ll_submit) / unpacking results (mrp_ll_wait)           */
  int64_t staged_bytes;      /* bytes of job descriptors, constraint words, id lists and path / interval tables written   */
                             /* to pinned host memory for the device to read over PCIe                                    */
  int64_t prof[8];           /* diagnostic (-DMRP_LL_TRACE library only, else 0): shader cycles in walk, pops,  */
                             /* pushes, successor generation, row init, whole job; #walks; nodes visited by walks */
  double heavy_busy_ms, heavy_idle_ms;     /* mrp_ll_session_begin_tiers: the same two sums over the heavy workgroups    */
  int64_t heavy_active_wgs;                /* ... and how many of them ran at least one search                           */
  int64_t heavy_fallbacks;                 /* sessions that fell back to a single launch (no heavy workgroup became resident) */
} mrp_ll_stats;

int mrp_ll_create(const mrp_ll_options* opt, mrp_ll_ctx** out);
void mrp_ll_destroy(mrp_ll_ctx* ctx);
const char* mrp_ll_last_error(const mrp_ll_ctx* ctx);

/* Static map (Environment ctor, ecbs.cpp:249-259): obstacles as [n][2] = x, y.  Uploaded once, used by any job. */
int mrp_ll_upload_map(mrp_ll_ctx* ctx, int32_t dimx, int32_t dimy, int32_t n_obstacles, const int32_t* obstacles_xy,
                      int32_t* map_id);

/* MRP_LL_ASTAR_TA: the heuristic of one goal cell of map `map_id` — dist[dimy][dimx] (row-major, dist[y * dimx + x]) =
 * ShortestPathHeuristic::getValue(cell, goal) (example/shortest_path_heuristic.hpp:56-60: all-pairs shortest paths on the
 * free cells; INT32_MAX = unreachable).  Computing it is the caller's business (the reference does it once per
 * Environment, cbs_ta.cpp:267); the engine keeps it next to the maps (any map size).  MRP_LL_E_BUSY during a session. */
int mrp_ll_upload_heuristic(mrp_ll_ctx* ctx, int32_t map_id, const int32_t* dist, int32_t* heuristic_id);

/* Copies every map uploaded so far to the device now (otherwise done lazily by the next submit / session_begin). */
int mrp_ll_sync_maps(mrp_ll_ctx* ctx);

/* Allocates the device-resident path store: n_slots slots of one path each (up to mrp_ll_options.max_horizon states).
 * Slot ids 0 .. n_slots-1 are the caller's to hand out (mrp_ll_job.result_path_id / path_ids).  MRP_LL_E_BUSY while a
 * batch or a session is in flight; calling it again re-allocates (contents are lost). */
int mrp_ll_path_store_reserve(mrp_ll_ctx* ctx, int32_t n_slots);

/* ---- incrementally maintained SIPP tables --------------------------------------------------------------------
 * The state SIPP::setCollisionIntervals (sipp.hpp:245-284) builds up, kept by the engine so that a planner which adds a
 * few collision intervals between two searches (example/mapf_prioritized_sipp.cpp:237-246) does not have to hand over —
 * and the engine re-derive — every interval of every location for every job.  mrp_ll_sipp_table_add(t, x, y, s, e)
 * appends [s, e] to that location's collision list and is equivalent to calling setCollisionIntervals(location, list)
 * with the extended list.  A table belongs to one map (its dimensions); it is not thread-safe.
 * In a session (mrp_ll_session_begin_sipp) the table also has a DEVICE-RESIDENT copy: a job on it carries only the cells
 * changed since the table's previous job, and the search reads the copy in place (up to 15 safe intervals per cell, bounds below 65535; a
 * table that needs more, and any table outside a session, travels whole with each job — same results).  One job per
 * table is in flight at a time (a second one simply travels whole).  mrp_ll_job.sipp_commit lets the engine add the
 * found path's stays itself. */
typedef struct mrp_ll_sipp_table mrp_ll_sipp_table;
int mrp_ll_sipp_table_create(mrp_ll_ctx* ctx, int32_t map_id, mrp_ll_sipp_table** out);
int mrp_ll_sipp_table_add(mrp_ll_sipp_table* t, int32_t x, int32_t y, int32_t start, int32_t end);
void mrp_ll_sipp_table_destroy(mrp_ll_sipp_table* t);

/* Forgets every uploaded map (host copy and device buffer contents; map ids start again at 0).  For callers that keep
 * one context across many batches of instances.  MRP_LL_E_BUSY while a batch or a session is in flight. */
int mrp_ll_release_maps(mrp_ll_ctx* ctx);

/* Limits of the LDS-resident fast tier for the launches / sessions that follow.  The tier keeps a whole search in LDS
 * (open list, focal list, walk queue, a (time, cell) bitmap of 64 time steps) in a window of fixed size, plus
 * lds_path_bytes for the focal path table of a search (a larger table is read from the search's arena slot); it serves
 * maps up to 32 x 32 and focal contexts of up to 128 agents.  lds_nodes / 2 = open-list entries a search may hold inside
 * the tier (at most 1023), lds_rows = time steps it may use (at most 64); a search that outgrows a limit is run by the
 * arena tier instead.  0 = keep the current value; lds_nodes < 0 disables the tier.  Results never depend on any of
 * this.  *occupancy_out (may be NULL) receives the resident searches per CU (160 KiB / window + table).  MRP_LL_E_BUSY
 * while a batch or a session is in flight. */
int mrp_ll_configure_tiers(mrp_ll_ctx* ctx, int32_t lds_nodes, int32_t lds_rows, int32_t lds_path_bytes,
                           int32_t* occupancy_out);

/* Resident searches per CU of a session of `algo` (MRP_LL_ASTAR_EPS, MRP_LL_ASTAR, MRP_LL_ASTAR_TA or MRP_LL_SIPP) with the
 * current tier limits, as the HIP runtime grants them: what a caller sizes mrp_ll_session_begin_algo's / _sipp's
 * `workgroups` with (256 CUs x this; the SIPP kernel's LDS tier is fixed: 16 per CU).  The A*-epsilon-only kernels keep the tier's (time, cell) bitmap in device memory and take 8 KB less
 * LDS per search than the others, so the answer depends on the algorithm. */
int mrp_ll_session_occupancy(mrp_ll_ctx* ctx, int32_t algo, int32_t* occupancy_out);

/* Blocking: run n_jobs independent searches, fill results[i] for jobs[i]. */
int mrp_ll_search_batch(mrp_ll_ctx* ctx, int32_t n_jobs, const mrp_ll_job* jobs, mrp_ll_result* results);

/* Asynchronous form of the same call: submit returns a ticket; results are valid after mrp_ll_wait(ticket). */
int mrp_ll_submit(mrp_ll_ctx* ctx, int32_t n_jobs, const mrp_ll_job* jobs, mrp_ll_result* results, int32_t* ticket);
int mrp_ll_wait(mrp_ll_ctx* ctx, int32_t ticket);

/* ---- session mode -------------------------------------------------------------------------------------------
 * Between mrp_ll_session_begin and mrp_ll_session_end the context keeps `workgroups` wavefronts resident on the GPU
 * and feeds them through a job ring in pinned host memory: mrp_ll_submit publishes its jobs immediately (no launch,
 * no stream command) and returns MRP_LL_E_BUSY when the ring has no room for the whole batch (consume finished
 * tickets, then retry); mrp_ll_poll is the non-blocking form of mrp_ll_wait.  Searches of different tickets finish in
 * any order, so a caller can keep thousands of independent conflict trees moving without waiting for the slowest
 * search of a batch.  Results are identical to the batch mode's.
 * Liveness: the resident wavefronts watch a heartbeat that every mrp_ll_submit* / mrp_ll_poll* / mrp_ll_wait call
 * moves, and leave on their own only when it has stood still for 20 s (a caller that died); a caller that keeps
 * polling may pause between submits for as long as it likes.  If the resident kernel is gone while jobs are in flight
 * (that limit, or a device fault), mrp_ll_poll_any / mrp_ll_wait return MRP_LL_E_DEVICE instead of spinning. */
int mrp_ll_session_begin(mrp_ll_ctx* ctx, int32_t workgroups /* 0 = mrp_ll_options.slots */);
/* The same for MRP_LL_SIPP jobs (a session serves one kind: A* / A*-epsilon jobs, or SIPP jobs — the other kind comes
 * back as MRP_LL_BAD_JOB).  At most 512 SIPP jobs are in flight at a time (MRP_LL_E_BUSY beyond that). */
int mrp_ll_session_begin_sipp(mrp_ll_ctx* ctx, int32_t workgroups);
/* A session for jobs of ONE algorithm (MRP_LL_ASTAR, MRP_LL_ASTAR_EPS or MRP_LL_SIPP): the resident kernel is the one
 * specialised for it — half the code and fewer registers than the mixed kernel of mrp_ll_session_begin, same results;
 * jobs of another algorithm come back as MRP_LL_BAD_JOB.  What the conflict-tree drivers use. */
int mrp_ll_session_begin_algo(mrp_ll_ctx* ctx, int32_t algo, int32_t workgroups);
/* An MRP_LL_ASTAR_EPS session as a PAIR of resident launches: `workgroups` front workgroups that run every search in the
 * LDS tier (1023 open entries, 64 time steps) and nothing else, and `heavy_workgroups` heavy ones with a 31.4 KB window
 * (3071 open entries, as many time steps as the arena slot holds — 512 by default; the arena tier behind it) that take over — through a device-side queue, without the
 * host — the searches that outgrow it.  Same jobs, same results as mrp_ll_session_begin_algo; the split keeps the long
 * searches out of the many small windows and gives them an LDS-resident tier of their own.
 * workgroups + heavy_workgroups <= mrp_ll_options.slots (each needs an arena slot).  If the heavy workgroups do not become
 * resident (no room left on the device), the call falls back to mrp_ll_session_begin_algo's single launch.
 * heavy_workgroups = 0 is that call. */
int mrp_ll_session_begin_tiers(mrp_ll_ctx* ctx, int32_t algo, int32_t workgroups, int32_t heavy_workgroups);
/* The same for SEVERAL contexts that begin their sessions on one device at the same time (one per host thread): a heavy
 * workgroup needs 31.4 KB of one CU's LDS in one piece, which no CU has left once the front workgroups of another context
 * have spread over the device.  Every caller passes the same `gate` (zero before the first call) and `parties` (the number
 * of callers): each launches its heavy workgroups, waits until they run, arrives at the gate, and launches its front
 * workgroups only when all parties have arrived (or 2 s have passed).  gate == NULL: no waiting. */
int mrp_ll_session_begin_tiers_gated(mrp_ll_ctx* ctx, int32_t algo, int32_t workgroups, int32_t heavy_workgroups,
                                     int32_t* gate, int32_t parties);
/* Resident workgroups per CU of such a session's front kernel with the current tier limits, and the LDS bytes one heavy
 * workgroup takes: a caller sizes the pair with 256 CUs x (160 KiB - its share of heavy windows) / front window. */
int mrp_ll_session_tiers_geometry(mrp_ll_ctx* ctx, int32_t* front_occupancy_out, int32_t* front_lds_bytes_out,
                                  int32_t* heavy_lds_bytes_out);
int mrp_ll_session_end(mrp_ll_ctx* ctx);
/* Session mode: the same as mrp_ll_submit.  There is one device queue and it is first in, first out; which search
 * starts next is decided by the ORDER in which the caller publishes — keep the queue shallow (about two searches per
 * resident wavefront) and publish the searches of the longest dependent chains first, as the conflict-tree drivers do.
 * `lane` (0 or 1) is accepted for source compatibility with round 1's two-ring design and otherwise ignored. */
int mrp_ll_submit_lane(mrp_ll_ctx* ctx, int32_t lane, int32_t n_jobs, const mrp_ll_job* jobs, mrp_ll_result* results,
                       int32_t* ticket);
int mrp_ll_poll(mrp_ll_ctx* ctx, int32_t ticket, int32_t* done);
/* Session mode only: collect every ticket that has completed since the last call (its results are filled in and the
 * ticket is released, exactly as after mrp_ll_wait).  One pass over the ring's completion words, independent of how
 * many tickets are in flight.  Writes at most `cap` ticket ids to `tickets`, their number to *n. */
int mrp_ll_poll_any(mrp_ll_ctx* ctx, int32_t* tickets, int32_t cap, int32_t* n);

/* Session mode, TWO host threads on one context ("co-workers": the device serves about twenty resident kernels of a
 * process well, a host core feeds about half a kernel's worth of conflict trees).  These two calls — and only these — may
 * be made concurrently by the co-workers of a context; session_begin / session_end / everything else stay with one of
 * them while the other is not inside a call.  `tag` (0 .. 3) names the caller: mrp_ll_poll_any_tagged hands out only the
 * tickets submitted under the same tag (it still unpacks every finished job it comes across into its caller's result
 * buffers, which must therefore stay valid until the ticket has been collected by its owner). */
int mrp_ll_submit_tagged(mrp_ll_ctx* ctx, int32_t tag, int32_t n_jobs, const mrp_ll_job* jobs, mrp_ll_result* results,
                         int32_t* ticket);
int mrp_ll_poll_any_tagged(mrp_ll_ctx* ctx, int32_t tag, int32_t* tickets, int32_t cap, int32_t* n);

/* ---- high-level conflict scans -----------------------------------------------------------
 * Environment::getFirstConflict (example/ecbs.cpp:401-452, example/cbs.cpp identical) and Environment::focalHeuristic
 * (example/ecbs.cpp:315-350) for a batch of solutions (conflict-tree nodes) in one kernel launch.
 * Solution s owns agents set_first_agent[s] .. set_first_agent[s+1]-1 of the flattened agent list; agent a owns states
 * path_first_state[a] .. path_first_state[a+1]-1 of states_xy ([total][2] = x, y at time 0, 1, 2, ...; every path has
 * at least one state, coordinates 0..255).  out[s]: the first conflict in the reference's scan order (time ascending; at
 * one time step vertex pairs before edge pairs; pairs (i, j), i < j, lexicographic; the last time step is never
 * checked) and the number of all conflicts (= focalHeuristic). */
typedef struct mrp_ll_conflict {
  int32_t found;          /* getFirstConflict's return value                                                  */
  int32_t time;           /* Conflict::time                                                                   */
  int32_t agent1, agent2; /* Conflict::agent1 < Conflict::agent2 (indices inside the solution)                */
  int32_t type;           /* 0 = Conflict::Vertex, 1 = Conflict::Edge                                         */
  int32_t x1, y1, x2, y2; /* Vertex: the shared cell in x1, y1; Edge: agent1's move (x1,y1) -> (x2,y2)         */
  int32_t count;          /* focalHeuristic(solution)                                                         */
} mrp_ll_conflict;
int mrp_ll_conflict_scan(mrp_ll_ctx* ctx, int32_t n_sets, const int32_t* set_first_agent,
                         const int32_t* path_first_state, const int32_t* states_xy, mrp_ll_conflict* out);

int mrp_ll_get_stats(const mrp_ll_ctx* ctx, mrp_ll_stats* out);
int mrp_ll_reset_stats(mrp_ll_ctx* ctx);

/* Library version / build info ("gfx950 ..."). */
const char* mrp_ll_version(void);

#ifdef __cplusplus
}
#endif
#endif /* MRP_LL_H */
