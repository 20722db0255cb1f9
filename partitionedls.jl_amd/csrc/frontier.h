// frontier.h — the frontier of the BnB search (host runtime), shared by solvers.hip (the single-context search, the C ABI's
// partls_frontier_*) and multi.hip (partls_fit_bnb_multi: one frontier per rank thread).
#pragma once
#include "ctx.h"
#include <cmath>
#include <queue>
#include <vector>

// fit_BnB (BnB.jl:94-132) as a best-first search: the frontier is ordered by the parent's bound, a round pops the `batch * world` most
// promising nodes (pruned against the incumbent, BnB.jl:102), DEALS them to the ranks, and — once every rank has bounded its share and
// the (bound, branch, snapshot slot) triples have been exchanged — branches the survivors (BnB.jl:117-124).  Every rank runs the same
// frontier on the same data, so it never has to be exchanged; only the triples are.  Dealing: a node whose parent left a tableau
// snapshot goes to the rank that holds it (warm start) up to that rank's quota of the round; what exceeds the quota and every node
// without a snapshot goes to the least loaded ranks and starts cold there (the tree spreads over the ranks by itself).  The frontier
// also keeps the reference counts of the snapshots (two children per branched node) and tells each rank which of ITS slots died.
struct partls_frontier {
    struct Node { double key; uint64_t pat, free_; unsigned long long seq; int owner, slot; };
    struct Cmp { bool operator()(const Node &a, const Node &b) const { return a.key > b.key || (a.key == b.key && a.seq > b.seq); } };
    std::priority_queue<Node, std::vector<Node>, Cmp> heap;
    int rank = 0, world = 1;
    int64_t batch = 1024, bounded = 0;
    unsigned long long seq = 0;
    double mu = INFINITY;
    uint64_t best_pat = 0, best_free = 0;
    std::vector<std::vector<int>> refs;              // [owner][slot]: children of that snapshot still in the frontier or in flight
    std::vector<Node> round;                         // the nodes of the current round, in popping order
    std::vector<int> assign;                         // rank of every node of the round
    std::vector<char> warm;                          // it starts from its parent's snapshot there
    std::vector<int> dead;                           // this rank's slots that lost their last reference

    void unref(int owner, int slot)
    {
        if (owner < 0) return;
        if (--refs[(size_t)owner][(size_t)slot] == 0 && owner == rank) dead.push_back(slot);
    }
    void setref(int owner, int slot, int v)
    {
        std::vector<int> &r = refs[(size_t)owner];
        if ((size_t)slot >= r.size()) r.resize((size_t)slot + 1024, 0);
        r[(size_t)slot] = v;
    }
    // pops and deals the next round; per_rank[world] = nodes of every rank; this rank's share into pat / free / src (capacity: batch)
    int64_t next(int64_t *mine, uint64_t *pat, uint64_t *fre, int32_t *src, int32_t *per_rank)
    {
        round.clear();
        while (!heap.empty() && (int64_t)round.size() < batch * world) {
            const Node nd = heap.top();
            heap.pop();
            if (nd.key >= mu) { unref(nd.owner, nd.slot); continue; }        // its bound can only be >= the parent's
            round.push_back(nd);
        }
        const int64_t cnt = (int64_t)round.size();
        const int64_t quota = (cnt + world - 1) / world;
        std::vector<int64_t> load((size_t)world, 0);
        assign.assign((size_t)cnt, -1);
        warm.assign((size_t)cnt, 0);
        for (int64_t i = 0; i < cnt; ++i) {
            const int o = round[(size_t)i].owner;
            if (o >= 0 && load[(size_t)o] < quota) { assign[(size_t)i] = o; warm[(size_t)i] = 1; ++load[(size_t)o]; }
        }
        for (int64_t i = 0; i < cnt; ++i) {
            if (assign[(size_t)i] >= 0) continue;
            int r = 0;
            for (int q = 1; q < world; ++q) if (load[(size_t)q] < load[(size_t)r]) r = q;
            assign[(size_t)i] = r; ++load[(size_t)r];
        }
        int64_t m = 0;
        for (int64_t i = 0; i < cnt; ++i)
            if (assign[(size_t)i] == rank) {
                pat[m] = round[(size_t)i].pat; fre[m] = round[(size_t)i].free_;
                src[m] = warm[(size_t)i] ? round[(size_t)i].slot : -1;
                ++m;
            }
        for (int q = 0; q < world; ++q) per_rank[q] = (int32_t)load[(size_t)q];
        *mine = m;
        return cnt;
    }
    // results of the round in RANK-MAJOR order (rank 0's nodes in the order next() gave them to rank 0, then rank 1's, ...)
    void ingest(const double *lb, const int32_t *br, const int32_t *dst)
    {
        const int64_t cnt = (int64_t)round.size();
        std::vector<int64_t> at((size_t)world, 0), base((size_t)world + 1, 0);
        for (int64_t i = 0; i < cnt; ++i) ++base[(size_t)assign[(size_t)i] + 1];
        for (int q = 0; q < world; ++q) base[(size_t)q + 1] += base[(size_t)q];
        for (int64_t i = 0; i < cnt; ++i) {
            const Node &nd = round[(size_t)i];
            const int me = assign[(size_t)i];
            const int64_t j = base[(size_t)me] + at[(size_t)me]++;
            ++bounded;
            unref(nd.owner, nd.slot);                                         // this child no longer needs its parent's tableau
            const double l = lb[j];
            const int k = br[j], d = dst[j];
            if (l >= mu || k < 0) {
                if (l < mu) { mu = l; best_pat = nd.pat; best_free = nd.free_; }   // feasible for the original problem (BnB.jl:109-115)
                if (d >= 0 && me == rank) dead.push_back(d);
                continue;
            }
            const uint64_t bit = 1ULL << k;
            if (d >= 0) setref(me, d, 2);                                     // both children start from this node's tableau
            const int o = d >= 0 ? me : -1, sl = d >= 0 ? d : -1;
            heap.push({l, nd.pat | bit, nd.free_ & ~bit, seq++, o, sl});      // alpha_pk >= 0 first (BnB.jl:120,123)
            heap.push({l, nd.pat & ~bit, nd.free_ & ~bit, seq++, o, sl});     // alpha_pk <= 0
        }
        round.clear();
    }
};

