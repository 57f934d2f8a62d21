// TEST INFRASTRUCTURE — the compact search tier (libmultirobotplanning_amd/csrc/ll_compact.h, the code the gfx950 kernels
// run) compiled against the host interpretation of its wave vocabulary (wave_emu.h), one search per call.
// Built by tests/test_compact_emu.py into tests/_build/libemu_ll.so; never part of the product.
#include <stdint.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "wave_emu.h"
#include "../../libmultirobotplanning_amd/csrc/ll_compact.h"

namespace {
int neighborIndexFromDelta(int dx, int dy) {  // Wait, Left, Right, Up, Down (ecbs.cpp:365-398)
  if (dx == 0 && dy == 0) return 0;
  if (dx == -1 && dy == 0) return 1;
  if (dx == 1 && dy == 0) return 2;
  if (dx == 0 && dy == 1) return 3;
  if (dx == 0 && dy == -1) return 4;
  return -1;
}
}  // namespace

extern "C" {

// One low-level search through the compact tier (eps: 0 = A*, 1 = A*-epsilon, 2 = A*-epsilon with the bitmap in device
// memory, ll_compact.h BG; 3 = the same in the WIDE geometry — 3071 open entries, t <= 510 — with the path table in
// "device memory" too: what the heavy workgroups run).  Inputs as in include/mrp_ll.h's mrp_ll_job (constraints [n][3] / [n][5],
// context paths flattened: path_len[n_agents], path_xy = all states back to back); lds_path_bytes = room for the focal
// path table in the LDS window (a larger table is read from "global" memory, as on the device); open_cap / max_t > 0:
// tighter limits of the tier for this job (what mrp_ll_configure_tiers' lds_nodes / lds_rows set on the device).
// out[0..5] = status (ct::C_*: 0 ok, 1 no solution, 2 expansion cap, -1 overflow: not a search of this tier), cost, fmin,
// n_states, expanded, nodes created;  out[6] = out-of-window LDS reads, out[7] = out-of-window LDS writes.
int emu_compact_search(int eps, int dimx, int dimy, int n_obst, const int32_t* obst_xy, int sx, int sy, int gx, int gy,
                       float w, int n_vc, const int32_t* vc, int n_ec, const int32_t* ec, int n_agents, int agent_idx,
                       const int32_t* path_len, const int32_t* path_xy, int64_t max_exp, int lds_path_bytes,
                       int open_cap, int max_t, int64_t* out, int32_t* states_xy, int states_cap) {
  using namespace mrp::ct;
  if (dimx < 1 || dimy < 1 || dimx > 32 || dimy > 32) return -2;
  const uint32_t cells = (uint32_t)dimx * dimy;
  std::vector<uint32_t> obst((cells + 31) / 32, 0u);
  for (int i = 0; i < n_obst; ++i) {
    const int x = obst_xy[2 * i], y = obst_xy[2 * i + 1];
    if (x < 0 || x >= dimx || y < 0 || y >= dimy) continue;
    const uint32_t c = (uint32_t)(y * dimx + x);
    obst[c >> 5] |= 1u << (c & 31);
  }
  // packer: the same words libmultirobotplanning_amd/csrc/mrp_ll_host.cpp packJob writes
  std::vector<uint32_t> vcw, ecw;
  int lastGoal = -1;
  for (int i = 0; i < n_vc; ++i) {
    const int32_t* v = vc + 3 * i;
    if (v[1] == gx && v[2] == gy) lastGoal = std::max(lastGoal, v[0]);
    if (v[0] < 0 || v[0] >= 1024 || v[1] < 0 || v[1] >= dimx || v[2] < 0 || v[2] >= dimy) continue;
    vcw.push_back(((uint32_t)v[0] << 16) | ((uint32_t)v[2] << 8) | (uint32_t)v[1]);
  }
  for (int i = 0; i < n_ec; ++i) {
    const int32_t* e = ec + 5 * i;
    const int k = neighborIndexFromDelta(e[3] - e[1], e[4] - e[2]);
    if (k < 0 || e[0] < 0 || e[0] >= 1024 || e[1] < 0 || e[1] >= dimx || e[2] < 0 || e[2] >= dimy) continue;
    ecw.push_back(((uint32_t)e[0] << 19) | ((uint32_t)(e[2] * dimx + e[1]) << 3) | (uint32_t)k);
  }
  vcw.push_back(0);
  ecw.push_back(0);
  // focal context: time-major table of the other agents' cells (x | y << 8), each path extended by its last cell
  std::vector<uint16_t> table;
  uint32_t npad = 0, tpad = 0;
  if (eps && n_agents > 0) {
    int tp = 0;
    for (int a = 0; a < n_agents; ++a)
      if (a != agent_idx && path_len[a] > 0) tp = std::max(tp, path_len[a]);
    if (tp > 0) {
      npad = ((uint32_t)n_agents + 15u) & ~15u;
      tpad = (uint32_t)tp;
      table.assign((size_t)tpad * npad, 0xFFFFu);
      size_t off = 0;
      for (int a = 0; a < n_agents; ++a) {
        const int len = path_len[a];
        if (a != agent_idx && len > 0) {
          uint16_t cell = 0xFFFFu;
          for (uint32_t tt = 0; tt < tpad; ++tt) {
            if ((int)tt < len) {
              const int x = path_xy[2 * (off + tt)], y = path_xy[2 * (off + tt) + 1];
              cell = (x >= 0 && x < dimx && y >= 0 && y < dimy) ? (uint16_t)(x | (y << 8)) : 0xFFFFu;
            }
            table[(size_t)tt * npad + a] = cell;
          }
        }
        off += (size_t)std::max(len, 0);
      }
    }
  }
  if (npad > 128) return -2;  // (the kernel starts such a job in the arena tier)
  const uint32_t tableBytes = tpad * npad * 2u;
  const bool tableInLds = eps != 3 && tableBytes != 0 && tableBytes <= (uint32_t)lds_path_bytes;
  const bool wide = eps == 3;
  const bool bg = eps >= 2;  // the A*-epsilon-only kernels' form: (time, cell) bitmap in "device memory", smaller window
  if (wide) lds_path_bytes = 0;
  std::vector<uint8_t> ldsMem((wide ? Wide::windowBytes(true) : windowBytes(bg)) + (uint32_t)std::max(lds_path_bytes, 0), 0xA5);  // garbage from "the previous job"
  wv::LdsWindow win{ldsMem.data(), (uint32_t)ldsMem.size(), 0, 0};
  if (tableInLds) std::memcpy(ldsMem.data() + pathsOff(bg), table.data(), tableBytes);
  const uint32_t wideRows = 512;  // time steps the "arena slot" of a heavy workgroup has room for (mrp_ll_options.max_horizon)
  std::vector<uint8_t> parentTab(wide ? Wide::parentBytes(wideRows) : kParentBytes, 0xEE);
  std::vector<uint32_t> bitsG((wide ? Wide::bitsBytes(wideRows) : kBitsBytes) / 4u, 0xA5A5A5A5u);
  std::vector<uint16_t> outPath(1024, 0);
  table.resize(table.size() + 256, 0xFFFFu);  // (lanes beyond a row's end are masked, but keep reads in bounds)
  CJob J;
  std::memset(&J, 0, sizeof(J));
  J.dimx = dimx; J.dimy = dimy; J.sx = sx; J.sy = sy; J.gx = gx; J.gy = gy;
  J.lastGoal = lastGoal;
  J.w = w;
  J.nVc = (uint32_t)vcw.size() - 1; J.nEc = (uint32_t)ecw.size() - 1;
  J.vc = (uint64_t)(uintptr_t)vcw.data(); J.ec = (uint64_t)(uintptr_t)ecw.data();
  J.obst = (uint64_t)(uintptr_t)obst.data(); J.obstWords = (uint32_t)obst.size();
  J.nAgentsPad = npad; J.tPad = tpad;
  J.pathsG = (uint64_t)(uintptr_t)table.data();
  J.maxExp = max_exp < 0 ? 0xFFFFFFFFu : (uint32_t)std::min<int64_t>(max_exp, 0xFFFFFFFEll);
  const uint32_t capT = wide ? Wide::kCap : kCap, maxTT = wide ? std::min(Wide::kMaxT, wideRows - 2u) : kMaxT;
  J.rows = wide ? wideRows : 0u;
  J.openCap = open_cap > 0 ? std::min<uint32_t>((uint32_t)open_cap, capT) : capT;
  J.maxT = max_t > 0 ? std::min<uint32_t>((uint32_t)max_t, maxTT) : maxTT;
  J.parentTab = (uint64_t)(uintptr_t)parentTab.data();
  J.outPath = (uint64_t)(uintptr_t)outPath.data();
  J.bitsG = (uint64_t)(uintptr_t)bitsG.data();
  if (J.nEc > 64) return -2;  // (the kernel starts such a job in the arena tier)
  std::memcpy(ldsMem.data() + oJob, &J, sizeof(J));
  const int32_t rc = !eps                ? compactSearch<false, true>(&win)
                     : wide              ? compactSearch<true, false, true, Wide>(&win)
                     : bg && tableInLds  ? compactSearch<true, true, true>(&win)
                     : bg                ? compactSearch<true, false, true>(&win)
                     : tableInLds        ? compactSearch<true, true>(&win)
                                         : compactSearch<true, false>(&win);
  CRes R;
  std::memcpy(&R, ldsMem.data() + oRes, sizeof(R));
  if (rc != R.status) return -3;
  out[0] = rc;
  out[1] = R.cost; out[2] = R.fmin; out[3] = R.nStates; out[4] = R.expanded; out[5] = R.nodes;
  out[6] = (int64_t)win.oobReads;
  out[7] = (int64_t)win.oobWrites;
  if (rc == C_OK)
    for (int k = 0; k < R.nStates && k < states_cap; ++k) {
      states_xy[2 * k] = outPath[k] & 0xFF;
      states_xy[2 * k + 1] = outPath[k] >> 8;
    }
  return 0;
}

// One low-level search of the task-assignment callers (compactSearchTA).  has_goal = 0: no task.  heur: the shortest-path
// table of the goal cell, [dimy][dimx] int32 (INT32_MAX: unreachable), as the caller of mrp_ll_upload_heuristic passes it.
// out as in emu_compact_search (status 3 / 4: capacity of the tier).
int emu_compact_search_ta(int dimx, int dimy, int n_obst, const int32_t* obst_xy, int sx, int sy, int has_goal, int gx, int gy,
                          const int32_t* heur, int n_vc, const int32_t* vc, int n_ec, const int32_t* ec, int64_t max_exp,
                          int open_cap, int max_t, int64_t* out, int32_t* states_xy, int states_cap) {
  using namespace mrp::ct;
  if (dimx < 1 || dimy < 1 || dimx > 32 || dimy > 32) return -2;
  const uint32_t cells = (uint32_t)dimx * dimy;
  std::vector<uint32_t> obst((cells + 31) / 32, 0u);
  for (int i = 0; i < n_obst; ++i) {
    const int x = obst_xy[2 * i], y = obst_xy[2 * i + 1];
    if (x < 0 || x >= dimx || y < 0 || y >= dimy) continue;
    const uint32_t c = (uint32_t)(y * dimx + x);
    obst[c >> 5] |= 1u << (c & 31);
  }
  std::vector<uint32_t> vcw, ecw;
  int lastGoal = -1;
  for (int i = 0; i < n_vc; ++i) {
    const int32_t* v = vc + 3 * i;
    if (!has_goal || (v[1] == gx && v[2] == gy)) lastGoal = std::max(lastGoal, v[0]);  // cbs_ta.cpp:290-301
    if (v[0] < 0 || v[0] >= 1024 || v[1] < 0 || v[1] >= dimx || v[2] < 0 || v[2] >= dimy) continue;
    vcw.push_back(((uint32_t)v[0] << 16) | ((uint32_t)v[2] << 8) | (uint32_t)v[1]);
  }
  for (int i = 0; i < n_ec; ++i) {
    const int32_t* e = ec + 5 * i;
    const int k = neighborIndexFromDelta(e[3] - e[1], e[4] - e[2]);
    if (k < 0 || e[0] < 0 || e[0] >= 1024 || e[1] < 0 || e[1] >= dimx || e[2] < 0 || e[2] >= dimy) continue;
    ecw.push_back(((uint32_t)e[0] << 19) | ((uint32_t)(e[2] * dimx + e[1]) << 3) | (uint32_t)k);
  }
  if (vcw.size() > 64 || ecw.size() > 64) return -2;
  vcw.push_back(0);
  ecw.push_back(0);
  std::vector<uint16_t> table(1024, 0xFFFFu);  // [y * 32 + x], 0xFFFF = unreachable (what mrp_ll_upload_heuristic stores)
  if (has_goal)
    for (int y = 0; y < dimy; ++y)
      for (int x = 0; x < dimx; ++x) {
        const int32_t d = heur[y * dimx + x];
        table[y * 32 + x] = (d < 0 || d > 0xFFFE) ? 0xFFFFu : (uint16_t)d;
      }
  std::vector<uint8_t> ldsMem(kLdsBytes + 2048u, 0xA5);
  wv::LdsWindow win{ldsMem.data(), (uint32_t)ldsMem.size(), 0, 0};
  std::vector<uint8_t> parentTab(kParentBytes, 0xEE);
  std::vector<uint16_t> outPath(1024, 0);
  CJob J;
  std::memset(&J, 0, sizeof(J));
  J.dimx = dimx; J.dimy = dimy; J.sx = sx; J.sy = sy; J.gx = has_goal ? gx : 0; J.gy = has_goal ? gy : 0;
  J.lastGoal = lastGoal;
  J.nVc = (uint32_t)vcw.size() - 1; J.nEc = (uint32_t)ecw.size() - 1;
  J.vc = (uint64_t)(uintptr_t)vcw.data(); J.ec = (uint64_t)(uintptr_t)ecw.data();
  J.obst = (uint64_t)(uintptr_t)obst.data(); J.obstWords = (uint32_t)obst.size();
  J.pathsG = (uint64_t)(uintptr_t)table.data();
  J.taNoGoal = has_goal ? 0u : 1u;
  J.maxExp = max_exp < 0 ? 0xFFFFFFFFu : (uint32_t)std::min<int64_t>(max_exp, 0xFFFFFFFEll);
  const uint32_t capT = kCap, maxTT = kMaxT;
  J.openCap = open_cap > 0 ? std::min<uint32_t>((uint32_t)open_cap, capT) : capT;
  J.maxT = max_t > 0 ? std::min<uint32_t>((uint32_t)max_t, maxTT) : maxTT;
  J.parentTab = (uint64_t)(uintptr_t)parentTab.data();
  J.outPath = (uint64_t)(uintptr_t)outPath.data();
  std::memcpy(ldsMem.data() + oJob, &J, sizeof(J));
  const int32_t rc = compactSearchTA(&win);
  CRes R;
  std::memcpy(&R, ldsMem.data() + oRes, sizeof(R));
  if (rc != R.status) return -3;
  out[0] = rc;
  out[1] = R.cost; out[2] = R.fmin; out[3] = R.nStates; out[4] = R.expanded; out[5] = R.nodes;
  out[6] = (int64_t)win.oobReads;
  out[7] = (int64_t)win.oobWrites;
  if (rc == C_OK)
    for (int k = 0; k < R.nStates && k < states_cap; ++k) {
      states_xy[2 * k] = outPath[k] & 0xFF;
      states_xy[2 * k + 1] = outPath[k] >> 8;
    }
  return 0;
}

}  // extern "C"
