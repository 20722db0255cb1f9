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
    // The open nodes, ordered by (parent's bound, sequence number).  The two children of a branched node carry the same key and
    // consecutive sequence numbers, so they leave the heap together: ONE entry stands for both (`bit` = the branched group; -1: a single
    // node — the root, or a child whose sibling went into the previous round), and the heap is 4-ary (a sift-down touches 3 cache lines per
    // level and half the levels of a binary heap).  Round 4: with std::priority_queue<Node> popping and dealing a round of 1024 nodes cost
    // 170 us against 200 us of device work for the same round.
    struct Entry { double key; unsigned long long seq; uint64_t pat, free_; int owner, slot, bit, pad; };
    struct Heap {
        std::vector<Entry> a;
        static bool less(const Entry &x, const Entry &y) { return x.key < y.key || (x.key == y.key && x.seq < y.seq); }
        bool empty() const { return a.empty(); }
        size_t size() const { return a.size(); }
        const Entry &top() const { return a[0]; }
        void push(const Entry &e)
        {
            a.push_back(e);
            size_t i = a.size() - 1;
            while (i > 0) {
                const size_t par = (i - 1) >> 2;
                if (!less(e, a[par])) break;
                a[i] = a[par];
                i = par;
            }
            a[i] = e;
        }
        void pop()
        {
            const Entry e = a.back();
            a.pop_back();
            const size_t n = a.size();
            if (n == 0) return;
            size_t i = 0;
            for (;;) {
                const size_t c0 = 4 * i + 1;
                if (c0 >= n) break;
                size_t m = c0;
                const size_t ce = c0 + 4 < n ? c0 + 4 : n;
                for (size_t c = c0 + 1; c < ce; ++c) if (less(a[c], a[m])) m = c;
                if (!less(a[m], e)) break;
                a[i] = a[m];
                i = m;
            }
            a[i] = e;
        }
        // compatibility with the round-3 call sites: a single node
        void push_node(const Node &nd) { push(Entry{nd.key, nd.seq, nd.pat, nd.free_, nd.owner, nd.slot, -1, 0}); }
    };
    Heap heap;
    // Entries popped AHEAD of the next round, while the device bounds the current one (prefetch()): a small heap of its own (it stays in
    // cache), merged with the big heap's top by next() — so the round is exactly what popping after the ingest would have given (children
    // pushed meanwhile compete through the big heap), and the expensive sift-downs of the big heap overlap the device's work.
    Heap staged;
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
    std::vector<int64_t> load;                       // nodes dealt to every rank in the current round

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
        const int64_t want = batch * world;
        while ((!heap.empty() || !staged.empty()) && (int64_t)round.size() < want) {
            const bool from_staged = !staged.empty() && (heap.empty() || Heap::less(staged.top(), heap.top()));
            const Entry e = from_staged ? staged.top() : heap.top();
            if (from_staged) staged.pop(); else heap.pop();
            const bool pair = e.bit >= 0;
            if (e.key >= mu) { unref(e.owner, e.slot); if (pair) unref(e.owner, e.slot); continue; }   // a bound can only be >= the parent's
            if (!pair) { round.push_back(Node{e.key, e.pat, e.free_, e.seq, e.owner, e.slot}); continue; }
            const uint64_t bitm = 1ULL << e.bit;
            round.push_back(Node{e.key, e.pat | bitm, e.free_, e.seq, e.owner, e.slot});              // alpha_pk >= 0 first (BnB.jl:120,123)
            const Node second{e.key, e.pat & ~bitm, e.free_, e.seq + 1, e.owner, e.slot};            // alpha_pk <= 0
            if ((int64_t)round.size() < want) round.push_back(second);
            else heap.push_node(second);                                        // the round is full: it competes again, as a single node
        }
        const int64_t cnt = (int64_t)round.size();
        const int64_t quota = (cnt + world - 1) / world;
        load.assign((size_t)world, 0);
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
    // pops up to a round's worth of entries out of the big heap ahead of time (no pruning, no reference counting: next() does both when it
    // takes them); call it between the launch of a round's device work and the wait for it
    void prefetch()
    {
        const size_t want = (size_t)(batch * world);
        while (!heap.empty() && staged.size() < want) { staged.push(heap.top()); heap.pop(); }
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
            heap.push(Entry{l, seq, nd.pat, nd.free_ & ~bit, o, sl, k, 0});   // both children: pat | bit (seq), pat & ~bit (seq + 1)
            seq += 2;
        }
        round.clear();
    }
};

