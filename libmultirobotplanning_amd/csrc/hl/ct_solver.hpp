// Conflict-tree searches (CBS, ECBS) restated as resumable per-instance state machines: start() / deliver() return the
// low-level searches to run next, so that a driver can keep thousands of instances in flight and hand every ready
// low-level search to the GPU at once.  The ORDER of all observable operations is the reference's:
//   CBS::search   cbs.hpp:85-172     ECBS::search   ecbs.hpp:109-288
// (children are created, searched and pushed in ascending agent order; the two children of a CT node are independent
// of each other — each starts from a copy of the parent — so their searches may run concurrently.)
//
// Speculative expansion.  The two children of a CT node are a pure function of that node (its first conflict, the two
// constraint sets, two low-level searches from the node's own solution; cbs.hpp:126-159, ecbs.hpp:235-272).  While the
// searches of the node that was REALLY popped are in flight, the machine may therefore pre-compute the children of the
// next nodes of the pop order (the current top-k of the focal / open list): requests carry the id of the CT node they
// expand (`group`), answers come back per group, and children are COMMITTED — counted, pushed, made visible to the
// heaps — strictly when their parent is the node the reference would pop next.  A pre-computed node that is never
// popped costs wasted searches, never a different result: cost, makespan, highLevelExpanded, lowLevelExpanded and every
// path are those of the sequential loop for every speculation width.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <unordered_map>
#include <vector>

#include "../../../include/mrp_hl.h"
#include "../../../include/mrp_ll.h"
#include "exact_heap.hpp"
#include "grid_mapf.hpp"

namespace mrp_hl {

struct LLRequest {            // one pending low-level search of an instance
  int32_t agent;
  ConsPtr constraints;        // constraint set of `agent` in the (child) node
  const PathVec* context;  // ECBS: the node's solution vector as the focal heuristics see it
  int32_t slot;               // position of the answer inside its group
  int32_t group;              // kRootGroup, or the id of the CT node these searches expand; requests of one group are
                              // consecutive and are answered together
};
constexpr int32_t kRootGroup = -1;

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
  int64_t specSearches() const { return specSearches_; }  // searches issued ahead of their node's pop
  const PathVec& finalSolution() const { return final_; }
  int64_t remainingLL() const { return capLL_ < 0 ? -1 : std::max<int64_t>(0, capLL_ - llExpanded_); }
  // CT nodes whose children may be computed at the same time (1 = the popped node only, i.e. no speculation)
  void setSpecWidth(int32_t k) { specWidth_ = std::max(1, k); }

  // First call.  Appends the searches to run to `next` (nothing when done()).
  void start(std::vector<LLRequest>& next) {
    phase_ = ROOT;
    if (n_ == 0) {
      pushRoot();
      pump(next);
      return;
    }
    if (algo_ == MRP_HL_CBS) {  // cbs.hpp:93-107 — root searches do not depend on each other
      for (int32_t i = 0; i < n_; ++i) next.push_back(LLRequest{i, root_.constraints[i], nullptr, i, kRootGroup});
    } else {                    // ecbs.hpp:118-136 — agent i sees the paths of agents < i
      rootAgent_ = 0;
      next.push_back(LLRequest{0, root_.constraints[0], &root_.solution, 0, kRootGroup});
    }
  }

  // The answers of every request of `group`, in slot order.  Appends the searches to run next to `next`.
  void deliver(int32_t group, const std::vector<LLAnswer>& answers, std::vector<LLRequest>& next) {
    if (phase_ == DONE) return;  // a pre-computed expansion that came back after the instance had finished
    if (group == kRootGroup) {
      if (!account(answers)) return;
      if (algo_ == MRP_HL_CBS) {
        for (int32_t i = 0; i < n_; ++i) {  // first failing agent makes search() return false (cbs.hpp:102-104)
          if (answers[i].status != MRP_LL_OK) {
            finish(MRP_HL_NO_SOLUTION);
            return;
          }
          root_.solution.set(i, answers[i].path);
          root_.cost += answers[i].cost;
        }
      } else {
        const LLAnswer& a = answers[0];
        if (a.status != MRP_LL_OK) {
          finish(MRP_HL_NO_SOLUTION);
          return;
        }
        root_.solution.set(rootAgent_, a.path);
        root_.cost += a.cost;
        root_.LB += a.fmin;
        rootAgent_ += 1;
        if (rootAgent_ < n_) {
          next.push_back(LLRequest{rootAgent_, root_.constraints[rootAgent_], &root_.solution, 0, kRootGroup});
          return;
        }
      }
      pushRoot();
      pump(next);
      return;
    }
    auto it = branches_.find(group);
    if (it == branches_.end()) return;
    it->second.answers = answers;
    it->second.state = Branch::READY;
    if (popped_ && current_ == group)
      pump(next);
    else
      speculate(next);  // it waits until its node is the one popped; its slot of the look-ahead window is free again
  }

  // public for the heap comparators
  const CTNode& node(int32_t id) const { return *nodes_[id]; }

 private:
  enum Phase { START, ROOT, TREE, DONE };
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
  // The expansion of one CT node: cbs.hpp:126-159 / ecbs.hpp:235-272 up to (not including) the pushes.
  struct Branch {
    enum State { ISSUED, READY } state = ISSUED;
    bool solved = false;                          // the node has no conflict: popping it ends the search
    std::shared_ptr<CTNode> parent;               // keeps the node alive until its children are committed
    std::shared_ptr<CTNode> child[2];
    int32_t agent[2] = {0, 0};
    std::vector<LLAnswer> answers;
  };

  void finish(int32_t st) {
    phase_ = DONE;
    status_ = st;
    branches_.clear();
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
    phase_ = TREE;
  }
  // every answer that is consumed counts, exactly once, in consumption order (ecbs.cpp:476-479 via the low-level search)
  bool account(const std::vector<LLAnswer>& answers) {
    for (const auto& a : answers) {
      llExpanded_ += a.expanded;
      llSearches_ += 1;
      if (a.status != MRP_LL_OK && a.status != MRP_LL_NO_SOLUTION) {  // capacity statuses: never guess
        finish(a.status == MRP_LL_CAP_EXPANSIONS ? MRP_HL_CAP : MRP_HL_LL_ERROR);
        return false;
      }
    }
    if (capLL_ >= 0 && llExpanded_ > capLL_) {
      finish(MRP_HL_CAP);
      return false;
    }
    return true;
  }

  // The reference's `while (!open.empty())` loop: pop (really), make sure that node's children exist, commit them, and
  // so on until a node's searches are still in flight; then look ahead.
  void pump(std::vector<LLRequest>& next) {
    for (;;) {
      if (!popped_) {
        if (open_.empty()) {
          finish(MRP_HL_NO_SOLUTION);
          return;
        }
        current_ = popNext();
        popped_ = true;
        hlExpanded_ += 1;
        if (capHL_ >= 0 && hlExpanded_ > capHL_) {
          finish(MRP_HL_CAP);
          return;
        }
      }
      auto it = branches_.find(current_);
      if (it == branches_.end()) {
        it = branch(current_, next, false);
      }
      Branch& b = it->second;
      if (b.solved) {
        final_ = b.parent->solution;
        finish(MRP_HL_SOLVED);
        return;
      }
      if (b.state != Branch::READY) break;  // wait for deliver(current_)
      if (!commit(b)) return;
      branches_.erase(it);
      popped_ = false;
    }
    speculate(next);
  }

  // The pop of the reference's loop head (ecbs.hpp:170-233 / cbs.hpp:119-124); returns the node id.
  int32_t popNext() {
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
    return pid;
  }

  // Creates the expansion record of node `pid` and issues its two searches.
  std::unordered_map<int32_t, Branch>::iterator branch(int32_t pid, std::vector<LLRequest>& next, bool speculative) {
    Branch b;
    b.parent = nodes_[pid];
    const CTNode& P = *b.parent;
    Conflict c;
    if (!firstConflict(P.solution, c, scratch_)) {
      b.solved = true;
      b.state = Branch::READY;
      return branches_.emplace(pid, std::move(b)).first;
    }
    ConstraintSet add1, add2;
    splitConflict(c, add1, add2);
    const int32_t ags[2] = {c.agent1, c.agent2};
    const ConstraintSet* adds[2] = {&add1, &add2};
    for (int k = 0; k < 2; ++k) {
      auto ch = std::make_shared<CTNode>(P);  // shares every path / constraint set with the parent
      const int32_t ag = ags[k];
      ch->constraints.set(ag, withAdded(P.constraints[ag], *adds[k]));
      ch->cost -= P.solution[ag]->cost;
      if (algo_ == MRP_HL_ECBS) ch->LB -= P.solution[ag]->fmin;
      b.child[k] = ch;
      b.agent[k] = ag;
    }
    auto it = branches_.emplace(pid, std::move(b)).first;
    for (int k = 0; k < 2; ++k) {
      const CTNode& ch = *it->second.child[k];
      next.push_back(LLRequest{it->second.agent[k], ch.constraints[it->second.agent[k]],
                               algo_ == MRP_HL_ECBS ? &ch.solution : nullptr, k, pid});
    }
    if (speculative) specSearches_ += 2;
    return it;
  }

  // The pushes of the reference's loop body for the node that was really popped (ecbs.hpp:264-281, cbs.hpp:155-166).
  bool commit(Branch& b) {
    if (!account(b.answers)) return false;
    for (int k = 0; k < 2; ++k) {  // ascending agent order == std::map order (ecbs.hpp:249)
      const std::shared_ptr<CTNode>& chp = b.child[k];
      CTNode& ch = *chp;
      const LLAnswer& a = b.answers[k];
      const int32_t ag = b.agent[k];
      ch.id = nextId_++;  // ++id happens whether or not the child's search succeeded
      if (a.status == MRP_LL_OK) {
        const PathPtr oldPath = ch.solution[ag];  // the parent's path of this agent
        // focalHeuristic(child) (ecbs.hpp:272): incremental while the scan horizon is unchanged (grid_mapf.hpp).  One walk
        // over the node's paths serves both horizons and both counts (PackedView: plain arrays of packed cells).
        PackedView& pv = PackedView::local();
        const bool packed = algo_ == MRP_HL_ECBS && pv.gather(ch.solution) && a.path->len() > 0 &&
                            static_cast<int32_t>(a.path->cell.size()) == a.path->len();
        int32_t oldT = 0, newT = 0;
        if (packed) {
          int32_t others = 0;  // longest path of the other agents
          for (int32_t j = 0; j < n_; ++j)
            if (j != ag) others = std::max(others, pv.len[j]);
          oldT = std::max(others, pv.len[ag]) - 1;
          newT = std::max(others, a.path->len()) - 1;
        } else if (algo_ == MRP_HL_ECBS) {
          oldT = maxT(ch.solution);
        }
        ch.solution.set(ag, a.path);
        ch.cost += a.cost;
        if (algo_ == MRP_HL_ECBS) {
          ch.LB += a.fmin;
          if (!packed) newT = maxT(ch.solution);
          if (newT != oldT)
            ch.focalHeuristic = countConflicts(ch.solution, scratch_);
          else if (packed)
            ch.focalHeuristic += conflictsOfAgentPacked(pv, ag, *a.path, newT) - conflictsOfAgentPacked(pv, ag, *oldPath, oldT);
          else
            ch.focalHeuristic += conflictsOfAgent(ch.solution, ag, *a.path, newT) -
                                 conflictsOfAgent(ch.solution, ag, *oldPath, oldT);
        }
        int32_t id = storeNode(chp);
        open_.push(id);
        if (algo_ == MRP_HL_ECBS && static_cast<float>(ch.cost) <= static_cast<float>(bestCost_) * w_) focal_.push(id);
      }
      // a failed child is dropped (cbs.hpp:161, ecbs.hpp:274)
    }
    nodes_[current_].reset();  // the popped node left the heaps for good when it was popped
    return true;
  }

  // Look ahead: the nodes the loop would pop next if no child got in front of them — the head of the focal list (ECBS)
  // or of the open list (CBS) in heap order — get their expansion records now.
  void speculate(std::vector<LLRequest>& next) {
    if (specWidth_ <= 1 || phase_ != TREE) return;
    int32_t inFlight = 0;
    for (const auto& kv : branches_) inFlight += kv.second.state == Branch::ISSUED ? 1 : 0;
    if (inFlight >= specWidth_ || static_cast<int32_t>(branches_.size()) >= 4 * specWidth_) return;
    cand_.clear();
    const int32_t want = specWidth_ - inFlight;
    int32_t seen = 0;
    auto visit = [&](int32_t id) {
      if (branches_.find(id) == branches_.end()) cand_.push_back(id);
      seen += 1;
      return static_cast<int32_t>(cand_.size()) < want && seen < 4 * specWidth_;
    };
    if (algo_ == MRP_HL_ECBS)
      focal_.walkOrdered(visit);
    else
      open_.walkOrdered(visit);
    for (int32_t id : cand_) branch(id, next, true);
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
  bool popped_ = false;     // a node has been popped and its children are not committed yet
  int32_t current_ = -1;    // that node (storage index)
  int32_t specWidth_ = 1;
  std::unordered_map<int32_t, Branch> branches_;  // storage index of a CT node -> its expansion
  std::vector<int32_t> cand_;
  PathVec final_;
  int64_t hlExpanded_ = 0, llExpanded_ = 0, specSearches_ = 0;
  int32_t llSearches_ = 0;
  std::vector<int32_t> scratch_;
};

}  // namespace mrp_hl
