// Conflict-tree searches (CBS, ECBS) restated as resumable per-instance state machines: each call of advance()
// consumes the results of the low-level searches it asked for last time and returns the next ones, so that a driver
// can keep thousands of instances in flight and hand every ready low-level search to the GPU in one batch.
// The ORDER of all observable operations is the reference's:
//   CBS::search   cbs.hpp:85-172     ECBS::search   ecbs.hpp:109-288
// (children are created, searched and pushed in ascending agent order; the two children of a CT node are independent
// of each other — each starts from a copy of the parent — so their searches may run concurrently.)
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "../../../include/mrp_hl.h"
#include "../../../include/mrp_ll.h"
#include "exact_heap.hpp"
#include "grid_mapf.hpp"

namespace mrp_hl {

struct LLRequest {            // one pending low-level search of an instance
  int32_t agent;
  ConsPtr constraints;        // constraint set of `agent` in the (child) node
  const std::vector<PathPtr>* context;  // ECBS: the node's solution vector as the focal heuristics see it
  int32_t slot;               // 0/1: which child (or root step) this answers
};

struct LLAnswer {
  int32_t status;             // MRP_LL_*
  int32_t cost, fmin;
  int64_t expanded;
  PathPtr path;               // valid when status == MRP_LL_OK
};

class Instance {
 public:
  Instance(const mrp_hl_instance& in, int32_t mapId, const mrp_hl_options& opt)
      : mapId_(mapId), algo_(opt.algo), w_(opt.w), capLL_(opt.max_ll_expansions), capHL_(opt.max_hl_expansions),
        openLess_{this}, focalLess_{this}, open_(openLess_), focal_(focalLess_) {
    n_ = in.n_agents;
    starts_.assign(in.starts_xy, in.starts_xy + 2 * n_);
    goals_.assign(in.goals_xy, in.goals_xy + 2 * n_);
    root_.solution.assign(n_, std::make_shared<Path>());  // empty paths: skipped by the focal heuristics (ecbs.cpp:287)
    root_.constraints.assign(n_, std::make_shared<ConstraintSet>());
  }

  bool done() const { return phase_ == DONE; }
  int32_t status() const { return status_; }
  int32_t mapId() const { return mapId_; }
  int32_t nAgents() const { return n_; }
  const int32_t* start(int32_t a) const { return &starts_[2 * a]; }
  const int32_t* goal(int32_t a) const { return &goals_[2 * a]; }
  int32_t algo() const { return algo_; }
  float w() const { return w_; }
  int64_t hlExpanded() const { return hlExpanded_; }
  int64_t llExpanded() const { return llExpanded_; }
  int32_t llSearches() const { return llSearches_; }
  const std::vector<PathPtr>& finalSolution() const { return final_; }
  int64_t remainingLL() const { return capLL_ < 0 ? -1 : std::max<int64_t>(0, capLL_ - llExpanded_); }

  // First call: answers empty.  Returns the searches to run next (empty when done()).
  void advance(const std::vector<LLAnswer>& answers, std::vector<LLRequest>& next) {
    next.clear();
    for (const auto& a : answers) {
      llExpanded_ += a.expanded;
      llSearches_ += 1;
      if (a.status != MRP_LL_OK && a.status != MRP_LL_NO_SOLUTION) {  // capacity statuses: never guess
        finish(a.status == MRP_LL_CAP_EXPANSIONS ? MRP_HL_CAP : MRP_HL_LL_ERROR);
        return;
      }
    }
    if (capLL_ >= 0 && llExpanded_ > capLL_) {
      finish(MRP_HL_CAP);
      return;
    }
    if (phase_ == START) {
      phase_ = ROOT;
      if (n_ == 0) {
        pushRoot();
        popAndBranch(next);
        return;
      }
      if (algo_ == MRP_HL_CBS) {  // cbs.hpp:93-107 — root searches do not depend on each other
        for (int32_t i = 0; i < n_; ++i) next.push_back(LLRequest{i, root_.constraints[i], nullptr, i});
      } else {                    // ecbs.hpp:118-136 — agent i sees the paths of agents < i
        rootAgent_ = 0;
        next.push_back(LLRequest{0, root_.constraints[0], &root_.solution, 0});
      }
      return;
    }
    if (phase_ == ROOT) {
      if (algo_ == MRP_HL_CBS) {
        for (int32_t i = 0; i < n_; ++i) {  // first failing agent makes search() return false (cbs.hpp:102-104)
          if (answers[i].status != MRP_LL_OK) {
            finish(MRP_HL_NO_SOLUTION);
            return;
          }
          root_.solution[i] = answers[i].path;
          root_.cost += answers[i].cost;
        }
      } else {
        const LLAnswer& a = answers[0];
        if (a.status != MRP_LL_OK) {
          finish(MRP_HL_NO_SOLUTION);
          return;
        }
        root_.solution[rootAgent_] = a.path;
        root_.cost += a.cost;
        root_.LB += a.fmin;
        rootAgent_ += 1;
        if (rootAgent_ < n_) {
          next.push_back(LLRequest{rootAgent_, root_.constraints[rootAgent_], &root_.solution, 0});
          return;
        }
      }
      pushRoot();
      popAndBranch(next);
      return;
    }
    // phase_ == BRANCH: the answers belong to children_[0..1]
    for (size_t k = 0; k < children_.size(); ++k) {  // ascending agent order == std::map order (ecbs.hpp:249)
      CTNode& ch = *children_[k];
      const LLAnswer& a = answers[k];
      const int32_t ag = childAgent_[k];
      if (a.status == MRP_LL_OK) {
        const PathPtr oldPath = ch.solution[ag];  // the parent's path of this agent
        const int32_t oldT = maxT(ch.solution);
        ch.solution[ag] = a.path;
        ch.cost += a.cost;
        if (algo_ == MRP_HL_ECBS) {
          ch.LB += a.fmin;
          // focalHeuristic(child) (ecbs.hpp:272): incremental while the scan horizon is unchanged (grid_mapf.hpp)
          const int32_t newT = maxT(ch.solution);
          if (newT == oldT)
            ch.focalHeuristic += conflictsOfAgent(ch.solution, ag, *a.path, newT) -
                                 conflictsOfAgent(ch.solution, ag, *oldPath, oldT);
          else
            ch.focalHeuristic = countConflicts(ch.solution, scratch_);
        }
        int32_t id = storeNode(children_[k]);
        open_.push(id);
        if (algo_ == MRP_HL_ECBS && static_cast<float>(ch.cost) <= static_cast<float>(bestCost_) * w_) focal_.push(id);
      }
      // a failed child is dropped (cbs.hpp:161, ecbs.hpp:274); ++id happens either way
    }
    children_.clear();
    childAgent_.clear();
    popAndBranch(next);
  }

  // public for the heap comparators
  const CTNode& node(int32_t id) const { return *nodes_[id]; }

 private:
  enum Phase { START, ROOT, BRANCH, DONE };
  struct OpenLess {   // HighLevelNode::operator< (cbs.hpp:187-191, ecbs.hpp:321-325)
    const Instance* self;
    bool operator()(int32_t a, int32_t b) const { return self->node(a).cost > self->node(b).cost; }
  };
  struct FocalLess {  // compareFocalHeuristic (ecbs.hpp:344-352)
    const Instance* self;
    bool operator()(int32_t a, int32_t b) const {
      const CTNode& x = self->node(a);
      const CTNode& y = self->node(b);
      if (x.focalHeuristic != y.focalHeuristic) return x.focalHeuristic > y.focalHeuristic;
      return x.cost > y.cost;
    }
  };

  void finish(int32_t st) {
    phase_ = DONE;
    status_ = st;
    children_.clear();
    nodes_.clear();
  }
  int32_t storeNode(const std::shared_ptr<CTNode>& n) {
    nodes_.push_back(n);
    return static_cast<int32_t>(nodes_.size()) - 1;
  }
  void pushRoot() {
    auto r = std::make_shared<CTNode>(root_);
    r->id = 0;
    if (algo_ == MRP_HL_ECBS) r->focalHeuristic = n_ ? countConflicts(r->solution, scratch_) : 0;
    int32_t id = storeNode(r);
    open_.push(id);
    if (algo_ == MRP_HL_ECBS) focal_.push(id);
    bestCost_ = r->cost;
    nextId_ = 1;
  }

  // The body of the reference's `while (!open.empty())` loop up to the two low-level calls.
  void popAndBranch(std::vector<LLRequest>& next) {
    if (open_.empty()) {
      finish(MRP_HL_NO_SOLUTION);
      return;
    }
    int32_t pid;
    if (algo_ == MRP_HL_ECBS) {
      int32_t oldBest = bestCost_;  // ecbs.hpp:170-190: bound is bestCost * w in binary32
      bestCost_ = node(open_.top()).cost;
      if (bestCost_ > oldBest) {
        const float lo = static_cast<float>(oldBest) * w_, hi = static_cast<float>(bestCost_) * w_;
        open_.walkOrdered([&](int32_t id) {
          float val = static_cast<float>(node(id).cost);
          if (val > lo && val <= hi) focal_.push(id);
          return !(val > hi);
        });
      }
      pid = focal_.top();
      focal_.pop();
      open_.erase(pid);
    } else {
      pid = open_.top();
      open_.pop();
    }
    std::shared_ptr<CTNode> P = nodes_[pid];
    nodes_[pid].reset();  // the popped node leaves the heaps for good
    hlExpanded_ += 1;
    if (capHL_ >= 0 && hlExpanded_ > capHL_) {
      finish(MRP_HL_CAP);
      return;
    }
    Conflict c;
    if (!firstConflict(P->solution, c, scratch_)) {
      final_ = P->solution;
      finish(MRP_HL_SOLVED);
      return;
    }
    ConstraintSet add1, add2;
    splitConflict(c, add1, add2);
    const int32_t ags[2] = {c.agent1, c.agent2};
    const ConstraintSet* adds[2] = {&add1, &add2};
    phase_ = BRANCH;
    for (int k = 0; k < 2; ++k) {
      auto ch = std::make_shared<CTNode>(*P);  // shares every path / constraint set with the parent
      ch->id = nextId_++;
      const int32_t ag = ags[k];
      ch->constraints[ag] = withAdded(P->constraints[ag], *adds[k]);
      ch->cost -= P->solution[ag]->cost;
      if (algo_ == MRP_HL_ECBS) ch->LB -= P->solution[ag]->fmin;
      children_.push_back(ch);
      childAgent_.push_back(ag);
      next.push_back(LLRequest{ag, ch->constraints[ag], algo_ == MRP_HL_ECBS ? &ch->solution : nullptr, k});
    }
  }

  int32_t mapId_, algo_;
  float w_;
  int64_t capLL_, capHL_;
  int32_t n_ = 0;
  std::vector<int32_t> starts_, goals_;
  Phase phase_ = START;
  int32_t status_ = MRP_HL_NO_SOLUTION;
  CTNode root_;
  int32_t rootAgent_ = 0;
  std::vector<std::shared_ptr<CTNode>> nodes_;  // id -> node while it is in the heaps
  OpenLess openLess_;
  FocalLess focalLess_;
  ExactHeap<OpenLess> open_;
  ExactHeap<FocalLess> focal_;
  int32_t bestCost_ = 0;
  int32_t nextId_ = 1;
  std::vector<std::shared_ptr<CTNode>> children_;
  std::vector<int32_t> childAgent_;
  std::vector<PathPtr> final_;
  int64_t hlExpanded_ = 0, llExpanded_ = 0;
  int32_t llSearches_ = 0;
  std::vector<int32_t> scratch_;
};

}  // namespace mrp_hl
