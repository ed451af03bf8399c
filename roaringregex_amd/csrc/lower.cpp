// lower.cpp — see lower.hpp.
#include "lower.hpp"

#include <algorithm>
#include <cstring>
#include <map>
#include <memory>
#include <queue>
#include <unordered_map>

namespace rrx {

// ------------------------------------------------------------------------------------------ trim
// The useful part of the reference automaton, taken from its UNEXPANDED rows (frontend.hpp: a row is a DAG of shared pieces).
// Reachability and co-reachability are exact and cost the pieces, not the edges.  The rows that come out are the explicit
// rows minus DOMINATED edges: u->w goes when a sibling u->v stays with label(w) a subset of label(v) and L(v) a superset of
// L(w) (the words accepted from it) - then u->w adds no word to L(u), so every state keeps its language whatever else has been
// dropped before.  L(v) >= L(w) is known when w is final only if v is and
//   * row(w) is empty, or the head piece of w is one of the pieces row(v) is made of (v took a copy of w's row: the skipped
//     tail of every nullable fold - x{1,n} leaves n^2/2 such edges, Parser.cpp:123-141), or
//   * to a small depth: every kept edge of w has a kept edge of v with a label above it into the same state or one that
//     dominates it by these same rules ((ab){1,n}: the state after `a` of copy i over that of copy i+1).
// The pruned row of a piece is made from its own edges and the PRUNED rows of its children, children and target rows first
// (explicit stack: chains are tens of thousands deep), so x{1,n} costs O(n) here too.
namespace {

struct PieceTrimmer {
    const RefAutomaton &a;
    const std::vector<uint8_t> &useful;
    std::vector<uint8_t> st;                       // 0 = not visited, 1 = on the stack, 2 = done
    std::vector<std::vector<Edge>> pe;             // pruned expansion per piece (targets: useful reference states, no NUL)
    std::vector<CharSet> first;                    // the characters pe[piece] moves on
    uint64_t pair_tests = 0;
    std::vector<uint32_t> desc_stack;

    PieceTrimmer(const RefAutomaton &a_, const std::vector<uint8_t> &u) : a(a_), useful(u), st(a_.pieces.size(), 0), pe(a_.pieces.size()), first(a_.pieces.size()) {}

    static CharSet strip(CharSet on) { on.w[0] &= ~1ULL; return on; }           // NUL never occurs inside a string (regex.h:157)
    static bool subset(const CharSet &x, const CharSet &y) { return !(x.w[0] & ~y.w[0]) && !(x.w[1] & ~y.w[1]); }

    // "piece `what` is one of the pieces `from` is made of": exact along a spanning forest of the child links (pre / post
    // numbers: what chains of nullable copies give), then a short search for pieces with several parents.  A miss only
    // keeps an edge.
    std::vector<uint32_t> pre, post;
    void number_forest() {
        const uint32_t P = (uint32_t)a.pieces.size();
        pre.assign(P, UINT32_MAX); post.assign(P, 0);
        std::vector<uint8_t> has_parent(P, 0);
        for (uint32_t p = 0; p < P; p++) for (uint32_t c : a.pieces[p].children) has_parent[c] = 1;
        uint32_t clock = 0;
        std::vector<std::pair<uint32_t, uint32_t>> stack;          // (piece, next child)
        for (uint32_t r = 0; r < P; r++) {
            if (has_parent[r] || pre[r] != UINT32_MAX) continue;
            pre[r] = clock++;
            stack.emplace_back(r, 0);
            while (!stack.empty()) {
                const uint32_t p = stack.back().first;
                if (stack.back().second < a.pieces[p].children.size()) {
                    const uint32_t c = a.pieces[p].children[stack.back().second++];
                    if (pre[c] == UINT32_MAX) { pre[c] = clock++; stack.emplace_back(c, 0); }
                } else { post[p] = clock; stack.pop_back(); }
            }
        }
    }
    bool is_descendant(uint32_t from, uint32_t what) {
        if (pre[from] < pre[what] && pre[what] < post[from]) return true;
        if (!a.pieces[what].shared) return false;
        desc_stack.assign(1, from);
        int budget = 12;
        while (!desc_stack.empty() && budget-- > 0) {
            const uint32_t p = desc_stack.back(); desc_stack.pop_back();
            for (uint32_t c : a.pieces[p].children) {
                if (c == what || (pre[c] < pre[what] && pre[what] < post[c])) return true;
                desc_stack.push_back(c);
            }
        }
        return false;
    }
    // The recursive rule is COINDUCTIVE: a pair met again while it is being proven counts as holding (loops inside the
    // repeated operand: (a+b+){1,n} needs "the a-loop of copy i over the a-loop of copy j" to prove itself).  A success that
    // leans on a pair still in progress further up is only as good as that pair: it is handed up (`leans`), never cached;
    // successes that lean on nothing outside their own proof, and all failures (an assumption can only help), are cached.
    std::unordered_map<uint64_t, uint8_t> memo;                 // (v, w) -> 2 yes / 1 no / 16 + d: no within depth d
    std::vector<std::pair<uint32_t, uint32_t>> in_progress;
    static constexpr int kProofDepth = 48;                      // (abcdefghij){1,n}: one level per state of the operand
    // `cut`: set when a limit (depth, an unfinished row, the size caps) stood in for a real "no" somewhere in the proof
    bool dominates(uint32_t v, uint32_t w, int depth, uint32_t &leans, bool &cut) {       // L(v) >= L(w), proven
        leans = UINT32_MAX;
        if (a.is_final[w] && !a.is_final[v]) return false;
        const uint32_t hw = a.head[w], hv = a.head[v];
        if (hw == kNoPiece) return true;
        if (st[hw] == 2 && pe[hw].empty()) return true;
        if (hv == kNoPiece) return false;
        if (is_descendant(hv, hw)) return true;
        if (depth == 0 || st[hv] != 2 || st[hw] != 2 || pair_tests > kTrimPairTests) { cut = true; return false; }
        if (!subset(first[hw], first[hv])) return false;           // w can start with a character v cannot
        for (uint32_t i = 0; i < in_progress.size(); i++) if (in_progress[i].first == v && in_progress[i].second == w) { leans = i; return true; }
        const std::vector<Edge> &rv = pe[hv], &rw = pe[hw];
        if (rw.size() * rv.size() > 1024) { cut = true; return false; }
        const uint64_t key = ((uint64_t)v << 32) | w;
        auto it = memo.find(key);
        if (it != memo.end()) {
            if (it->second == 2) return true;
            if (it->second == 1) return false;
            if (it->second - 16 >= depth) { cut = true; return false; }
        }
        pair_tests += rw.size() * rv.size();
        const uint32_t me = (uint32_t)in_progress.size();
        in_progress.emplace_back(v, w);
        bool all = true, my_cut = false;
        uint32_t lowest = UINT32_MAX;
        for (const Edge &e : rw) {
            bool found = false;
            for (const Edge &f : rv) {
                if (!subset(e.on, f.on)) continue;
                if (f.to == e.to) { found = true; break; }
                uint32_t l;
                if (dominates(f.to, e.to, depth - 1, l, my_cut)) { lowest = std::min(lowest, l); found = true; break; }
            }
            if (!found) { all = false; break; }
        }
        in_progress.pop_back();
        if (!all) {
            memo[key] = my_cut ? (uint8_t)(16 + depth) : (uint8_t)1;
            cut = cut || my_cut;
            return false;
        }
        if (lowest >= me) memo[key] = 2;                        // leans on itself at most
        else leans = lowest;
        return true;
    }
    void raw_expand(uint32_t root, std::vector<Edge> &out) {                    // a child that is still on the stack (a cycle)
        std::vector<uint32_t> stack{root}, seen{root};
        while (!stack.empty()) {
            const uint32_t p = stack.back(); stack.pop_back();
            for (const Edge &e : a.pieces[p].direct) { const CharSet on = strip(e.on); if (useful[e.to] && !on.empty()) out.push_back(Edge{e.to, on}); }
            for (uint32_t c : a.pieces[p].children) if (std::find(seen.begin(), seen.end(), c) == seen.end()) { seen.push_back(c); stack.push_back(c); }
        }
    }
    std::vector<uint32_t> n_parents;               // pieces that list it as a child
    std::vector<uint8_t> heads_useful;             // it is the head piece of a useful state
    uint64_t visited_edges = 0, live_edges = 0;
    void build(uint32_t p) {
        // candidates with their origin: 0 = the piece's own edges, k = its k-th child (whose row is pruned within itself
        // already: only pairs of different origin are compared - an alternation of 1000 keywords is a chain of 1000 pieces,
        // each adding one edge to a row of hundreds)
        std::vector<Edge> cand;
        std::vector<uint16_t> origin;
        for (const Edge &e : a.pieces[p].direct) { const CharSet on = strip(e.on); if (useful[e.to] && !on.empty()) cand.push_back(Edge{e.to, on}); }
        origin.assign(cand.size(), 0);
        uint32_t k = 0;
        for (uint32_t c : a.pieces[p].children) {
            k++;
            if (st[c] == 2) {
                cand.insert(cand.end(), pe[c].begin(), pe[c].end());
                if (n_parents[c] == 1 && !heads_useful[c]) { live_edges -= pe[c].size(); std::vector<Edge>().swap(pe[c]); }    // nobody else reads it
            } else raw_expand(c, cand);
            origin.resize(cand.size(), (uint16_t)std::min<uint32_t>(k, 0xfffe));
        }
        visited_edges += cand.size();
        std::vector<uint32_t> idx(cand.size());
        for (uint32_t i = 0; i < idx.size(); i++) idx[i] = i;
        std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) { return cand[x].to < cand[y].to; });
        std::vector<Edge> merged;
        std::vector<uint16_t> from;                 // 0xffff: the target came from several origins
        for (uint32_t i : idx) {
            if (!merged.empty() && merged.back().to == cand[i].to) { merged.back().on |= cand[i].on; if (from.back() != origin[i]) from.back() = 0xffff; }
            else { merged.push_back(cand[i]); from.push_back(origin[i]); }
        }
        const size_t n = merged.size();
        std::vector<uint8_t> dropped(n, 0);
        if (n > 1) {
            // dominators are tried in this order: the piece's own edges first (the copy taken, not the tail skipped to)
            std::vector<uint32_t> order, order_own;
            for (uint32_t i = 0; i < n; i++) if (from[i] == 0 || from[i] == 0xffff) order.push_back(i);
            order_own = order;
            for (uint32_t i = 0; i < n; i++) if (from[i] != 0 && from[i] != 0xffff) order.push_back(i);
            const size_t per_edge = n <= 64 ? n : std::max<size_t>(8, 4096 / n);
            for (uint32_t w = 0; w < n; w++) {
                size_t tried = 0;
                // an edge that came from the only child has been compared with its fellows there: the own edges are left
                const bool own_only = k == 1 && from[w] == 1;
                for (uint32_t v : own_only ? order_own : order) {
                    if (v == w || dropped[v] || !subset(merged[w].on, merged[v].on)) continue;
                    if (from[v] == from[w] && from[v] != 0 && from[v] != 0xffff) continue;      // compared when that child was built
                    if (++tried > per_edge) break;
                    uint32_t leans;
                    bool cut = false;
                    if (dominates(merged[v].to, merged[w].to, kProofDepth, leans, cut)) { dropped[w] = 1; break; }
                }
            }
        }
        std::vector<Edge> &out = pe[p];
        for (uint32_t i = 0; i < n; i++) if (!dropped[i]) out.push_back(merged[i]);
        live_edges += out.size();
        for (const Edge &e : out) first[p] |= e.on;
        if (live_edges > kTrimBudget || visited_edges > 32 * kTrimBudget)
            throw BudgetError("pattern too large: the automaton's rows exceed the host pipeline's work budget");
    }
    void run(uint32_t root) {
        if (st[root]) return;
        std::vector<uint32_t> stack{root};
        while (!stack.empty()) {
            const uint32_t p = stack.back();
            if (st[p] == 2) { stack.pop_back(); continue; }
            st[p] = 1;
            bool pushed = false;
            for (uint32_t c : a.pieces[p].children) if (st[c] == 0) { stack.push_back(c); pushed = true; }
            for (const Edge &e : a.pieces[p].direct) {
                if (!useful[e.to] || strip(e.on).empty()) continue;
                const uint32_t h = a.head[e.to];
                if (h != kNoPiece && st[h] == 0) { stack.push_back(h); pushed = true; }
            }
            if (pushed) continue;
            build(p);
            st[p] = 2;
            stack.pop_back();
        }
    }
};

}  // namespace

Trimmed trim(const RefAutomaton &a) {
    Trimmed t;
    std::memset(t.cls, 0, sizeof t.cls);
    t.cls_rep.assign(1, 0);
    const uint32_t N = a.states_n, P = (uint32_t)a.pieces.size();
    auto live_edge = [&](const Edge &e) { CharSet on = e.on; on.w[0] &= ~1ULL; return !on.empty() && e.to < N; };
    // ---- reachable from the initial state: every piece is walked once
    std::vector<uint8_t> fwd(N, 0), bwd(N, 0), seen(P, 0);
    std::vector<uint32_t> stack, pstack;
    if (a.initial < N) { fwd[a.initial] = 1; stack.push_back(a.initial); }
    while (!stack.empty()) {
        const uint32_t s = stack.back(); stack.pop_back();
        if (a.head[s] == kNoPiece || seen[a.head[s]]) continue;
        seen[a.head[s]] = 1; pstack.push_back(a.head[s]);
        while (!pstack.empty()) {
            const uint32_t p = pstack.back(); pstack.pop_back();
            for (const Edge &e : a.pieces[p].direct) if (live_edge(e) && !fwd[e.to]) { fwd[e.to] = 1; stack.push_back(e.to); }
            for (uint32_t c : a.pieces[p].children) if (!seen[c]) { seen[c] = 1; pstack.push_back(c); }
        }
    }
    // ---- able to reach a final state: backwards over (target -> pieces that hold it -> pieces that include those -> owners)
    {
        std::vector<std::vector<uint32_t>> holders(N), parents(P), owners(P);
        for (uint32_t p = 0; p < P; p++) {
            for (const Edge &e : a.pieces[p].direct) if (live_edge(e)) holders[e.to].push_back(p);
            for (uint32_t c : a.pieces[p].children) parents[c].push_back(p);
        }
        for (uint32_t s = 0; s < N; s++) if (a.head[s] != kNoPiece) owners[a.head[s]].push_back(s);
        std::vector<uint8_t> live(P, 0);
        for (uint32_t s = 0; s < N; s++) if (a.is_final[s]) { bwd[s] = 1; stack.push_back(s); }
        while (!stack.empty()) {
            const uint32_t s = stack.back(); stack.pop_back();
            for (uint32_t h : holders[s]) {
                if (live[h]) continue;
                live[h] = 1; pstack.push_back(h);
                while (!pstack.empty()) {
                    const uint32_t p = pstack.back(); pstack.pop_back();
                    for (uint32_t o : owners[p]) if (!bwd[o]) { bwd[o] = 1; stack.push_back(o); }
                    for (uint32_t q : parents[p]) if (!live[q]) { live[q] = 1; pstack.push_back(q); }
                }
            }
        }
    }
    if (a.initial >= N || !(fwd[a.initial] && bwd[a.initial])) return t;   // empty language
    std::vector<uint8_t> useful(N, 0);
    std::vector<int64_t> newid(N, -1);
    for (uint32_t s = 0; s < N; s++) if (fwd[s] && bwd[s]) { useful[s] = 1; newid[s] = (int64_t)t.ref_id.size(); t.ref_id.push_back(s); }
    t.n = (uint32_t)t.ref_id.size();
    t.initial = (uint32_t)newid[a.initial];
    t.is_final.assign(t.n, 0);
    t.out.assign(t.n, {});
    PieceTrimmer pt(a, useful);
    pt.number_forest();
    pt.n_parents.assign(P, 0);
    pt.heads_useful.assign(P, 0);
    for (uint32_t p = 0; p < P; p++) for (uint32_t c : a.pieces[p].children) pt.n_parents[c]++;
    for (uint32_t s = 0; s < N; s++) if (useful[s] && a.head[s] != kNoPiece) pt.heads_useful[a.head[s]] = 1;
    if (a.head[a.initial] != kNoPiece) pt.run(a.head[a.initial]);
    for (uint32_t k = 0; k < t.n; k++) {
        const uint32_t s = t.ref_id[k];
        t.is_final[k] = a.is_final[s];
        if (a.head[s] == kNoPiece) continue;
        pt.run(a.head[s]);
        for (const Edge &e : pt.pe[a.head[s]]) t.out[k].push_back(Edge{(uint32_t)newid[e.to], e.on});
    }
    // byte classes = atoms of the boolean algebra generated by the edge labels
    std::vector<CharSet> labels;
    for (auto &v : t.out) for (const Edge &e : v) labels.push_back(e.on);
    std::sort(labels.begin(), labels.end());
    labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
    std::vector<uint32_t> block(128, 0);                 // refine: block id per char
    std::vector<uint8_t> live(128, 0);
    for (const CharSet &L : labels) {
        std::map<std::pair<uint32_t, bool>, uint32_t> remap;
        uint32_t nb = 0;
        for (unsigned c = 0; c < 128; c++) {
            if (L.has(c)) live[c] = 1;
            auto key = std::make_pair(block[c], L.has(c));
            auto it = remap.find(key);
            if (it == remap.end()) it = remap.emplace(key, nb++).first;
            block[c] = it->second;
        }
    }
    std::map<uint32_t, uint32_t> cls_of_block;
    for (unsigned c = 1; c < 128; c++) {
        if (!live[c]) continue;
        auto it = cls_of_block.find(block[c]);
        if (it == cls_of_block.end()) {
            it = cls_of_block.emplace(block[c], (uint32_t)t.cls_rep.size()).first;
            t.cls_rep.push_back((uint8_t)c);
        }
        t.cls[c] = (uint8_t)it->second;
    }
    t.ncls = (uint32_t)t.cls_rep.size();
    return t;
}

// ------------------------------------------------------------------------------------------ NFA lowering
namespace {

using Node = Reduced::Node;

void sort_unique(std::vector<uint32_t> &v) { std::sort(v.begin(), v.end()); v.erase(std::unique(v.begin(), v.end()), v.end()); }

// One bisimulation quotient.  forward: merge nodes with equal (label, fin, successor blocks);
// backward: merge nodes with equal (label, is-initial, predecessor blocks).  Both keep the language.
// Node 0 is the initial node and stays node 0.  Returns true if anything merged.
bool quotient(std::vector<Node> &nodes, bool forward) {
    const uint32_t n = (uint32_t)nodes.size();
    std::vector<std::vector<uint32_t>> nb(n);
    if (forward) for (uint32_t u = 0; u < n; u++) nb[u] = nodes[u].follow;
    else for (uint32_t u = 0; u < n; u++) for (uint32_t v : nodes[u].follow) nb[v].push_back(u);
    std::vector<uint32_t> block(n);
    {
        std::map<std::tuple<uint64_t, uint64_t, bool>, uint32_t> keys;
        for (uint32_t u = 0; u < n; u++) {
            bool flag = forward ? nodes[u].fin : (u == 0);
            auto key = std::make_tuple(nodes[u].label.w[0], nodes[u].label.w[1], flag);
            auto it = keys.find(key);
            if (it == keys.end()) it = keys.emplace(key, (uint32_t)keys.size()).first;
            block[u] = it->second;
        }
    }
    uint32_t count = 0;
    {
        std::vector<uint32_t> tmp = block; sort_unique(tmp); count = (uint32_t)tmp.size();
    }
    // Partition refinement with a worklist: a round only looks at the DIRTY nodes (those with a neighbour that changed
    // block in the round before); the others still carry their block's signature (the set of their neighbours' blocks).
    // The plain "recompute every signature until nothing splits" form took one round per node on a chain of equal labels
    // (x{17000}: 154 s); recomputing only dirty signatures still cost a node of in-degree d its d neighbours in every
    // round it was dirty ((ab?){1,n}: one round per copy, the last node entered from all of them: n^2).  So the signature
    // is kept incrementally: per node a sorted list (neighbour block, how many neighbours in it) and a 64-bit sum of
    // mix(block) over the blocks present; a neighbour's move updates two entries.  Nodes are grouped by (block, sum); the
    // partition that comes out is then CHECKED against the real signatures (below), so a collision cannot merge what must
    // stay apart.  The result is the coarsest stable refinement (unique).
    std::vector<std::vector<uint32_t>> dep(n);             // dep[v]: the nodes whose signature mentions v
    for (uint32_t u = 0; u < n; u++) for (uint32_t v : nb[u]) dep[v].push_back(u);
    std::vector<std::vector<uint32_t>> members(count);
    std::vector<uint32_t> where(n);
    for (uint32_t u = 0; u < n; u++) { where[u] = (uint32_t)members[block[u]].size(); members[block[u]].push_back(u); }
    auto mix = [](uint32_t b) { uint64_t z = (uint64_t)b + 0x9e3779b97f4a7c15ULL; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; return z ^ (z >> 31); };
    std::vector<std::vector<std::pair<uint32_t, uint32_t>>> cnt(n);       // sorted by block
    std::vector<uint64_t> sum(n, 0);
    for (uint32_t u = 0; u < n; u++) {
        std::vector<uint32_t> bl;
        bl.reserve(nb[u].size());
        for (uint32_t v : nb[u]) bl.push_back(block[v]);
        std::sort(bl.begin(), bl.end());
        for (size_t i = 0; i < bl.size();) {
            size_t j = i;
            while (j < bl.size() && bl[j] == bl[i]) j++;
            cnt[u].emplace_back(bl[i], (uint32_t)(j - i));
            sum[u] += mix(bl[i]);
            i = j;
        }
    }
    auto bump = [&](uint32_t u, uint32_t b, int by) {
        auto &c = cnt[u];
        auto it = std::lower_bound(c.begin(), c.end(), std::make_pair(b, 0u));
        if (it != c.end() && it->first == b) {
            it->second = (uint32_t)((int64_t)it->second + by);
            if (!it->second) { c.erase(it); sum[u] -= mix(b); }
        } else { c.insert(it, std::make_pair(b, 1u)); sum[u] += mix(b); }      // (by = +1)
    };
    std::vector<uint8_t> is_dirty(n, 1), queued(n, 0);
    std::vector<uint32_t> dirty(n), next_dirty, moved, moved_from, moved_to;
    for (uint32_t u = 0; u < n; u++) dirty[u] = u;
    while (!dirty.empty()) {
        std::sort(dirty.begin(), dirty.end(), [&](uint32_t x, uint32_t y) {
            return block[x] != block[y] ? block[x] < block[y] : sum[x] != sum[y] ? sum[x] < sum[y] : x < y; });
        next_dirty.clear();
        for (size_t i = 0; i < dirty.size();) {
            const uint32_t B = block[dirty[i]];
            size_t end = i;
            while (end < dirty.size() && block[dirty[end]] == B) end++;
            // which group keeps the id B: the one that agrees with the clean members (they all carry one signature: none of
            // their neighbours moved), or, if every member is dirty, the largest group
            int64_t keep_from = -1;
            if (end - i < members[B].size()) {
                uint32_t clean = UINT32_MAX;
                for (uint32_t u : members[B]) if (!is_dirty[u]) { clean = u; break; }
                keep_from = (int64_t)end;                           // (no dirty group stays, unless one matches)
                for (size_t g = i; g < end; g++) if (sum[dirty[g]] == sum[clean]) { keep_from = (int64_t)g; break; }
            } else {
                size_t best = 0;
                for (size_t g = i; g < end;) {
                    size_t g2 = g + 1;
                    while (g2 < end && sum[dirty[g2]] == sum[dirty[g]]) g2++;
                    if (g2 - g > best) { best = g2 - g; keep_from = (int64_t)g; }
                    g = g2;
                }
            }
            for (size_t g = i; g < end;) {
                size_t g2 = g + 1;
                while (g2 < end && sum[dirty[g2]] == sum[dirty[g]]) g2++;
                if ((int64_t)g != keep_from) {
                    const uint32_t nbk = count++;
                    members.emplace_back();
                    for (size_t q = g; q < g2; q++) {
                        const uint32_t u = dirty[q];
                        std::vector<uint32_t> &mb = members[B];
                        const uint32_t last = mb.back();
                        mb[where[u]] = last; where[last] = where[u]; mb.pop_back();
                        where[u] = (uint32_t)members[nbk].size(); members[nbk].push_back(u);
                        moved.push_back(u); moved_from.push_back(B); moved_to.push_back(nbk);
                    }
                }
                g = g2;
            }
            i = end;
        }
        // the moves take effect together, after every group of the round has been formed on the old partition
        for (size_t q = 0; q < moved.size(); q++) {
            block[moved[q]] = moved_to[q];
            for (uint32_t d : dep[moved[q]]) {
                bump(d, moved_from[q], -1);
                bump(d, moved_to[q], +1);
                if (!queued[d]) { queued[d] = 1; next_dirty.push_back(d); }
            }
        }
        moved.clear(); moved_from.clear(); moved_to.clear();
        for (uint32_t u : dirty) is_dirty[u] = 0;
        dirty.swap(next_dirty);
        for (uint32_t u : dirty) { is_dirty[u] = 1; queued[u] = 0; }
    }
    // the check: within a block every member must carry the same SET of neighbour blocks (the sums only said so)
    for (uint32_t b = 0; b < count; b++) {
        const std::vector<uint32_t> &mb = members[b];
        for (size_t k = 1; k < mb.size(); k++) {
            const auto &x = cnt[mb[0]], &y = cnt[mb[k]];
            bool same = x.size() == y.size();
            for (size_t q = 0; same && q < x.size(); q++) same = x[q].first == y[q].first;
            if (!same) return false;                               // a 64-bit collision: merge nothing (sound, and never seen)
        }
    }
    if (count == n) return false;
    // renumber so that node 0's block is 0 and the rest keep first-occurrence order
    std::vector<int64_t> id(count, -1);
    uint32_t next = 0;
    for (uint32_t u = 0; u < n; u++) if (id[block[u]] < 0) id[block[u]] = next++;
    std::vector<Node> merged(count);
    for (uint32_t u = 0; u < n; u++) {
        Node &m = merged[(size_t)id[block[u]]];
        m.label = nodes[u].label;
        m.fin = m.fin || nodes[u].fin;
        for (uint32_t v : nodes[u].follow) m.follow.push_back((uint32_t)id[block[v]]);
    }
    for (Node &m : merged) sort_unique(m.follow);
    nodes.swap(merged);
    return true;
}

// Removes edges u->w that are covered by a sibling edge u->v whose target simulates w ("little brother"
// pruning): v simulates w iff (fin(w) => fin(v)) and every successor w' of w has a successor v' of v with
// label(w') subset of label(v') and v' simulates w'.  With label(w) subset of label(v) the edge u->w adds no
// word to the language.  The reference's nested-optional construction of x{m,n} (Parser.cpp:133-137) leaves
// many such jump-ahead edges; pruning them turns bounded repeats back into plain shift chains.
// Returns true if an edge was removed.  Nodes that become unreachable are dropped.
// Cheap sufficient form of the same pruning for big graphs: v certainly simulates w when it is final if w is,
// and follow(w) is a subset of follow(v) (v can copy every move of w).  Siblings are tried in order of decreasing
// out-degree, a few per edge; follow sets are compared as bitsets.  Returns true if an edge was removed.
bool prune_by_containment(std::vector<Node> &nodes) {
    const uint32_t n = (uint32_t)nodes.size();
    const uint32_t words = (n + 63) / 64;
    auto subset = [](const CharSet &a, const CharSet &b) { return !(a.w[0] & ~b.w[0]) && !(a.w[1] & ~b.w[1]); };
    std::vector<uint64_t> fs((size_t)n * words, 0);
    for (uint32_t u = 0; u < n; u++) for (uint32_t v : nodes[u].follow) fs[(size_t)u * words + (v >> 6)] |= 1ULL << (v & 63);
    auto contained = [&](uint32_t w, uint32_t v) {              // follow(w) subset of follow(v)
        const uint64_t *a = &fs[(size_t)w * words], *b = &fs[(size_t)v * words];
        for (uint32_t i = 0; i < words; i++) if (a[i] & ~b[i]) return false;
        return true;
    };
    bool removed = false;
    std::vector<uint32_t> order;
    for (uint32_t u = 0; u < n; u++) {
        std::vector<uint32_t> &f = nodes[u].follow;
        if (f.size() < 2) continue;
        order.assign(f.begin(), f.end());
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return nodes[a].follow.size() != nodes[b].follow.size() ? nodes[a].follow.size() > nodes[b].follow.size() : a < b; });
        std::vector<uint32_t> kept;
        for (uint32_t w : f) {
            bool drop = false;
            int tried = 0;
            for (uint32_t v : order) {
                if (v == w) continue;
                if (nodes[v].follow.size() < nodes[w].follow.size()) break;
                if (++tried > 6) break;
                if ((nodes[w].fin && !nodes[v].fin) || !subset(nodes[w].label, nodes[v].label) || !contained(w, v)) continue;
                // mutual cover: keep the lower index
                const bool mutual = nodes[v].follow.size() == nodes[w].follow.size() && nodes[v].fin == nodes[w].fin &&
                                    subset(nodes[v].label, nodes[w].label) && contained(v, w);
                if (mutual && w < v) continue;
                drop = true;
                break;
            }
            if (drop) removed = true; else kept.push_back(w);
        }
        f.swap(kept);                                            // (fs keeps the old rows: still supersets, still sound)
    }
    if (!removed) return false;
    std::vector<int64_t> id(n, -1);
    std::vector<uint32_t> reach{0};
    id[0] = 0;
    for (size_t k = 0; k < reach.size(); k++)
        for (uint32_t v : nodes[reach[k]].follow) if (id[v] < 0) { id[v] = (int64_t)reach.size(); reach.push_back(v); }
    std::vector<Node> out(reach.size());
    for (size_t k = 0; k < reach.size(); k++) {
        out[k] = nodes[reach[k]];
        for (uint32_t &v : out[k].follow) v = (uint32_t)id[v];
        sort_unique(out[k].follow);
    }
    nodes.swap(out);
    return true;
}

bool prune_by_simulation(std::vector<Node> &nodes) {
    const uint32_t n = (uint32_t)nodes.size();
    if (n > 160) return false;                                   // O(n^2 deg^2): small graphs only
    auto subset = [](const CharSet &a, const CharSet &b) { return !(a.w[0] & ~b.w[0]) && !(a.w[1] & ~b.w[1]); };
    // sim[q*n+p] = 1: p simulates q
    std::vector<uint8_t> sim((size_t)n * n, 0);
    for (uint32_t q = 0; q < n; q++) for (uint32_t p2 = 0; p2 < n; p2++) sim[(size_t)q * n + p2] = (!nodes[q].fin || nodes[p2].fin) ? 1 : 0;
    bool changed = true;
    while (changed) {
        changed = false;
        for (uint32_t q = 0; q < n; q++) for (uint32_t p2 = 0; p2 < n; p2++) {
            if (!sim[(size_t)q * n + p2] || q == p2) continue;
            bool ok = true;
            for (uint32_t qs : nodes[q].follow) {
                bool found = false;
                for (uint32_t ps : nodes[p2].follow)
                    if (sim[(size_t)qs * n + ps] && subset(nodes[qs].label, nodes[ps].label)) { found = true; break; }
                if (!found) { ok = false; break; }
            }
            if (!ok) { sim[(size_t)q * n + p2] = 0; changed = true; }
        }
    }
    bool removed = false;
    for (uint32_t u = 0; u < n; u++) {
        std::vector<uint32_t> &f = nodes[u].follow;
        std::vector<uint8_t> drop(f.size(), 0);
        for (size_t i = 0; i < f.size(); i++) {
            const uint32_t w = f[i];
            for (size_t j = 0; j < f.size() && !drop[i]; j++) {
                const uint32_t v = f[j];
                if (i == j || drop[j] || !sim[(size_t)w * n + v] || !subset(nodes[w].label, nodes[v].label)) continue;
                // mutual cover: keep the lower index
                const bool mutual = sim[(size_t)v * n + w] && subset(nodes[v].label, nodes[w].label);
                if (mutual && w < v) continue;
                drop[i] = 1;
            }
        }
        std::vector<uint32_t> kept;
        for (size_t i = 0; i < f.size(); i++) if (!drop[i]) kept.push_back(f[i]); else removed = true;
        f.swap(kept);
    }
    if (!removed) return false;
    // drop nodes no longer reachable from node 0
    std::vector<int64_t> id(n, -1);
    std::vector<uint32_t> order{0};
    id[0] = 0;
    for (size_t k = 0; k < order.size(); k++)
        for (uint32_t v : nodes[order[k]].follow) if (id[v] < 0) { id[v] = (int64_t)order.size(); order.push_back(v); }
    std::vector<Node> kept(order.size());
    for (size_t k = 0; k < order.size(); k++) {
        kept[k] = nodes[order[k]];
        for (uint32_t &v : kept[k].follow) v = (uint32_t)id[v];
        sort_unique(kept[k].follow);
    }
    nodes.swap(kept);
    return true;
}

// Twin merge: nodes with the same finality, the same predecessor SET and the same follow SET differ only in the label
// they are entered on; one node with the union of the labels accepts the same words (every path through the merged node
// on a character of label(u) is a path through u, and the same for v).  This is what re-joins the two halves of a
// character class that the in-label split or an alternation of literals (a|b) made into separate positions, where the
// bisimulation quotients (same label only) cannot.  Node 0 (no label) is never merged.  Returns true if anything merged.
bool merge_twins(std::vector<Node> &nodes) {
    const uint32_t n = (uint32_t)nodes.size();
    std::vector<std::vector<uint32_t>> pred(n);
    for (uint32_t u = 0; u < n; u++) for (uint32_t v : nodes[u].follow) pred[v].push_back(u);
    std::map<std::tuple<bool, std::vector<uint32_t>, std::vector<uint32_t>>, uint32_t> groups;
    std::vector<uint32_t> leader(n);
    bool any = false;
    for (uint32_t u = 0; u < n; u++) {
        leader[u] = u;
        if (u == 0) continue;
        auto key = std::make_tuple((bool)nodes[u].fin, nodes[u].follow, pred[u]);      // both vectors are sorted
        auto it = groups.find(key);
        if (it == groups.end()) groups.emplace(std::move(key), u);
        else { leader[u] = it->second; any = true; }
    }
    if (!any) return false;
    std::vector<int64_t> id(n, -1);
    uint32_t next = 0;
    for (uint32_t u = 0; u < n; u++) if (leader[u] == u) id[u] = next++;
    std::vector<Node> merged(next);
    for (uint32_t u = 0; u < n; u++) {
        Node &m = merged[(size_t)id[leader[u]]];
        m.label.w[0] |= nodes[u].label.w[0]; m.label.w[1] |= nodes[u].label.w[1];
        if (leader[u] != u) continue;
        m.fin = nodes[u].fin;
        for (uint32_t v : nodes[u].follow) m.follow.push_back((uint32_t)id[leader[v]]);
    }
    for (Node &m : merged) sort_unique(m.follow);
    nodes.swap(merged);
    return true;
}

// Maximum bipartite matching (Hopcroft-Karp) between "u as source" and "v as target" over the edges
// u -> v, v != u.  match_succ[u] = v uses edge u->v as a "next bit" edge.
void path_cover(const std::vector<Node> &nodes, std::vector<int64_t> &succ, std::vector<int64_t> &pred) {
    const uint32_t n = (uint32_t)nodes.size();
    succ.assign(n, -1); pred.assign(n, -1);
    std::vector<std::vector<uint32_t>> adj(n);
    for (uint32_t u = 0; u < n; u++) for (uint32_t v : nodes[u].follow) if (v != u) adj[u].push_back(v);
    // greedy seed: prefer targets with a single predecessor candidate (typical chains)
    for (uint32_t u = 0; u < n; u++)
        for (uint32_t v : adj[u]) if (pred[v] < 0) { succ[u] = v; pred[v] = u; break; }
    std::vector<int32_t> dist(n);
    const int32_t INF = 1 << 30;
    auto bfs = [&]() {
        std::queue<uint32_t> q;
        bool found = false;
        for (uint32_t u = 0; u < n; u++) { if (succ[u] < 0) { dist[u] = 0; q.push(u); } else dist[u] = INF; }
        while (!q.empty()) {
            uint32_t u = q.front(); q.pop();
            for (uint32_t v : adj[u]) {
                int64_t w = pred[v];
                if (w < 0) found = true;
                else if (dist[(size_t)w] == INF) { dist[(size_t)w] = dist[u] + 1; q.push((uint32_t)w); }
            }
        }
        return found;
    };
    // iterative DFS to avoid deep recursion on long chains
    std::vector<uint32_t> it(n);
    auto try_augment = [&](uint32_t root) {
        std::vector<uint32_t> path{root};
        while (!path.empty()) {
            uint32_t u = path.back();
            if (it[u] >= adj[u].size()) { dist[u] = INF; path.pop_back(); continue; }
            uint32_t v = adj[u][it[u]++];
            int64_t w = pred[v];
            if (w < 0) {
                // augment along path: each path[i] takes the target it was exploring
                uint32_t tv = v;
                for (size_t i = path.size(); i-- > 0;) {
                    uint32_t pu = path[i];
                    int64_t old = succ[pu];
                    succ[pu] = tv; pred[tv] = pu;
                    if (old < 0) break;
                    tv = (uint32_t)old;
                }
                return true;
            }
            if (dist[(size_t)w] == dist[u] + 1) path.push_back((uint32_t)w);
        }
        return false;
    };
    while (bfs()) {
        std::fill(it.begin(), it.end(), 0);
        for (uint32_t u = 0; u < n; u++) if (succ[u] < 0) try_augment(u);
    }
}

}  // namespace

Reduced reduce(const Trimmed &t) {
    Reduced red;
    std::memcpy(red.cls, t.cls, sizeof red.cls);
    red.ncls = t.ncls;
    red.cls_rep = t.cls_rep;
    if (t.n == 0) return red;
    // ---- split states by in-label: node (state, label); node 0 = the initial state before any input
    std::vector<Node> nodes(1);
    nodes[0].fin = t.is_final[t.initial];
    std::vector<std::map<CharSet, uint32_t>> copies(t.n);
    for (uint32_t s = 0; s < t.n; s++)
        for (const Edge &e : t.out[s]) {
            auto &m = copies[e.to];
            if (m.find(e.on) == m.end()) {
                m.emplace(e.on, (uint32_t)nodes.size());
                Node nd; nd.label = e.on; nd.fin = t.is_final[e.to];
                nodes.push_back(nd);
            }
        }
    std::vector<std::vector<uint32_t>> follow_of_state(t.n);
    for (uint32_t s = 0; s < t.n; s++) {
        for (const Edge &e : t.out[s]) follow_of_state[s].push_back(copies[e.to].at(e.on));
        sort_unique(follow_of_state[s]);
    }
    nodes[0].follow = follow_of_state[t.initial];
    for (uint32_t s = 0; s < t.n; s++) for (auto &kv : copies[s]) nodes[kv.second].follow = follow_of_state[s];
    // ---- the dominated edges trim() dropped may have been the only way into some nodes
    {
        const uint32_t n = (uint32_t)nodes.size();
        std::vector<int64_t> id(n, -1);
        std::vector<uint32_t> reach{0};
        id[0] = 0;
        for (size_t k = 0; k < reach.size(); k++)
            for (uint32_t v : nodes[reach[k]].follow) if (id[v] < 0) { id[v] = (int64_t)reach.size(); reach.push_back(v); }
        if (reach.size() < n) {
            std::vector<Node> kept(reach.size());
            for (size_t k = 0; k < reach.size(); k++) {
                kept[k] = std::move(nodes[reach[k]]);
                for (uint32_t &v : kept[k].follow) v = (uint32_t)id[v];
                sort_unique(kept[k].follow);
            }
            nodes.swap(kept);
        }
    }
    // ---- shrink
    for (int round = 0; round < 8; round++) {
        bool c = prune_by_containment(nodes);              // first: it cuts the edge count the quotients iterate over
        bool a = quotient(nodes, true);
        bool b = quotient(nodes, false);
        bool d = prune_by_simulation(nodes);
        bool e = merge_twins(nodes);
        if (!a && !b && !c && !d && !e) break;
    }
    red.nodes.swap(nodes);
    return red;
}

bool lower_nfa(const Reduced &red, uint32_t max_bits, NfaProgram &p, bool allow_carry, bool gaps) {
    p = NfaProgram();
    if (red.nodes.empty()) {                 // empty language: one dead position
        p.W = 1; p.nbits = 1;
        p.init = {0}; p.fin = {0}; p.chain = {0}; p.self = {0}; p.excm = {0}; p.cgrp = {0}; p.ctgt = {0};
        p.X.assign(1, 0); p.B.assign(256, 0);
        return true;
    }
    const std::vector<Node> &nodes = red.nodes;
    const uint32_t n = (uint32_t)nodes.size();
    if (n > max_bits) return false;
    // ---- layout along a path cover
    std::vector<int64_t> succ, pred;
    path_cover(nodes, succ, pred);
    // `gaps`: one never-entered position in front of every path head but the first, so that the bit shifted out of the
    // end of one path dies there instead of entering the next path: the line-mode lane kernel then needs no CHAIN mask
    // (the gap is in no B row).  order[q] = node at position q, or kGap.
    constexpr uint32_t kGap = UINT32_MAX;
    std::vector<uint32_t> pos(n, UINT32_MAX), order;
    std::vector<uint8_t> placed(n, 0);
    auto lay = [&](uint32_t head) {
        if (gaps && !order.empty()) order.push_back(kGap);
        for (int64_t u = head; u >= 0 && !placed[(size_t)u]; u = succ[(size_t)u]) {
            placed[(size_t)u] = 1; pos[(size_t)u] = (uint32_t)order.size(); order.push_back((uint32_t)u);
        }
    };
    lay(0);                                               // node 0 (entered by nothing) heads a path: position 0
    for (uint32_t u = 0; u < n; u++) if (pred[u] < 0 && !placed[u]) lay(u);
    for (uint32_t u = 0; u < n; u++) if (!placed[u]) {   // a cycle: cut it in front of u
        succ[(size_t)pred[u]] = -1; pred[u] = -1; lay(u);
    }
    const uint32_t N = (uint32_t)order.size();            // positions, gaps included
    if (N > max_bits) return false;
    p.nbits = N;
    p.W = (N + 31) / 32;
    const uint32_t W = p.W;
    p.init.assign(W, 0); p.fin.assign(W, 0); p.chain.assign(W, 0); p.self.assign(W, 0); p.excm.assign(W, 0);
    p.cgrp.assign(W, 0); p.ctgt.assign(W, 0);
    const bool dense_x = N <= kDenseExceptionBits;         // beyond: the CSR form only (N * W words would be GiBs)
    p.X.assign(dense_x ? (size_t)N * W : 0, 0);
    p.B.assign((size_t)256 * W, 0);
    auto setbit = [&](std::vector<uint32_t> &v, size_t base, uint32_t b) { v[base + (b >> 5)] |= 1u << (b & 31); };
    setbit(p.init, 0, pos[0]);
    p.accepts_empty = nodes[0].fin;
    // follow sets by position
    std::vector<std::vector<uint32_t>> fpos(N);
    std::vector<uint8_t> has_self(N, 0);
    for (uint32_t u = 0; u < n; u++)
        for (uint32_t v : nodes[u].follow) {
            if (v == u) has_self[pos[u]] = 1;
            else fpos[pos[u]].push_back(pos[v]);
        }
    for (auto &f : fpos) sort_unique(f);
    // carry groups: target t with a maximal run [lo, t-1] of positions that all lead to t
    std::vector<uint8_t> is_target(N, 0), in_group(N, 0);
    std::vector<int64_t> group_target(N, -1);
    for (uint32_t t = 1; allow_carry && t < N; t++) {
        auto leads = [&](uint32_t q) { return std::binary_search(fpos[q].begin(), fpos[q].end(), t); };
        if (!leads(t - 1)) continue;
        uint32_t lo = t - 1;
        while (lo > 0 && leads(lo - 1) && !is_target[lo - 1] && !in_group[lo - 1]) lo--;
        if (is_target[t - 1] || in_group[t - 1]) continue;
        if (lo + 1 == t) continue;                               // a run of one is just the shift edge
        for (uint32_t q = lo; q < t; q++) { in_group[q] = 1; group_target[q] = t; setbit(p.cgrp, 0, q); }
        is_target[t] = 1;
        setbit(p.ctgt, 0, t);
        p.n_carry++;
    }
    for (uint32_t u = 0; u < n; u++) {
        const uint32_t pu = pos[u];
        if (nodes[u].fin) setbit(p.fin, 0, pu);
        if (pred[u] >= 0) setbit(p.chain, 0, pu);          // entered by the shift from position pu-1
        if (has_self[pu]) setbit(p.self, 0, pu);
        for (unsigned c = 1; c < 128; c++) if (nodes[u].label.has(c)) setbit(p.B, (size_t)c * W, pu);
    }
    p.xoff.assign(N + 1, 0);
    for (uint32_t q = 0; q < N; q++) {
        uint32_t extra = 0;
        for (uint32_t tv : fpos[q]) {
            if (tv == q + 1 && ((p.chain[tv >> 5] >> (tv & 31)) & 1u)) continue;     // the shift edge
            if (group_target[q] == (int64_t)tv) continue;                              // covered by the add-carry
            if (dense_x) setbit(p.X, (size_t)q * W, tv);
            p.xtgt.push_back(tv); extra++;
            p.max_exc_row_words = std::max(p.max_exc_row_words, (tv >> 5) + 1);
        }
        p.xoff[q + 1] = (uint32_t)p.xtgt.size();
        if (extra) { setbit(p.excm, 0, q); p.n_exc++; }
    }
    return true;
}

// ------------------------------------------------------------------------------------------ DFA lowering
// Minimisation (Hopcroft's partition refinement) of a complete table over K classes and its renumbering.  kind[i]: 0 = rejecting,
// 1 = accepting, 2 = the ESCAPE state of a sampled table (kept apart from the dead state it would otherwise merge with).
// State 0 is the dead state, state 1 the start.  (The "recompute every signature until nothing splits" form took one round per
// state on a chain - a{1,n} is n rounds over n states: 0.45 s at n = 2400, minutes at 16000.)
static bool minimise_into(const std::vector<uint32_t> &nxt, const std::vector<uint8_t> &kind, const uint32_t K, DfaProgram &d) {
    const uint32_t D = (uint32_t)kind.size();
    std::vector<uint32_t> block(D);
    uint32_t count = 0;
    {
        // inverse transitions per class, CSR: inv[k][first[k][t] .. first[k][t + 1]) = the states that go to t on class k
        std::vector<uint32_t> first((size_t)K * (D + 1), 0), inv((size_t)K * D);
        for (uint32_t i = 0; i < D; i++) for (uint32_t k = 0; k < K; k++) first[(size_t)k * (D + 1) + nxt[(size_t)i * K + k] + 1]++;
        for (uint32_t k = 0; k < K; k++) for (uint32_t t2 = 0; t2 < D; t2++) first[(size_t)k * (D + 1) + t2 + 1] += first[(size_t)k * (D + 1) + t2];
        {
            std::vector<uint32_t> fill(first);
            for (uint32_t i = 0; i < D; i++)
                for (uint32_t k = 0; k < K; k++) inv[(size_t)k * D + fill[(size_t)k * (D + 1) + nxt[(size_t)i * K + k]]++] = i;
        }
        // the partition: elems holds the states block by block; a block is elems[lo[b] .. hi[b]).  It starts as the kinds present.
        std::vector<uint32_t> elems(D), loc(D), lo, hi, marked;
        {
            uint32_t n_of[3] = {0, 0, 0}, at[3];
            for (uint32_t i = 0; i < D; i++) n_of[kind[i]]++;
            at[0] = 0; at[1] = n_of[0]; at[2] = n_of[0] + n_of[1];
            for (int q = 0; q < 3; q++) if (n_of[q]) { lo.push_back(at[q]); hi.push_back(at[q] + n_of[q]); }
            for (uint32_t i = 0; i < D; i++) { const uint32_t p = at[kind[i]]++; elems[p] = i; loc[i] = p; }
            for (uint32_t b = 0; b < lo.size(); b++) for (uint32_t q = lo[b]; q < hi[b]; q++) block[elems[q]] = b;
        }
        marked.assign(lo.size(), 0);
        std::vector<uint8_t> queued;                    // [block][class]
        std::vector<std::pair<uint32_t, uint32_t>> work;
        auto enqueue = [&](uint32_t b, uint32_t k) {
            if (queued.size() < (size_t)(b + 1) * K) queued.resize((size_t)(b + 1) * K, 0);
            if (!queued[(size_t)b * K + k]) { queued[(size_t)b * K + k] = 1; work.emplace_back(b, k); }
        };
        {
            uint32_t largest = 0;                       // every starting block but the largest is a splitter
            for (uint32_t b = 1; b < lo.size(); b++) if (hi[b] - lo[b] > hi[largest] - lo[largest]) largest = b;
            for (uint32_t b = 0; b < lo.size(); b++) if (b != largest) for (uint32_t k = 0; k < K; k++) enqueue(b, k);
        }
        std::vector<uint32_t> touched, splitter;
        while (!work.empty()) {
            const uint32_t B = work.back().first, k = work.back().second;
            work.pop_back();
            queued[(size_t)B * K + k] = 0;
            splitter.assign(elems.begin() + lo[B], elems.begin() + hi[B]);      // (B itself may split below)
            touched.clear();
            for (uint32_t t2 : splitter)
                for (uint32_t q = first[(size_t)k * (D + 1) + t2]; q < first[(size_t)k * (D + 1) + t2 + 1]; q++) {
                    const uint32_t i = inv[(size_t)k * D + q], b = block[i];
                    const uint32_t at = loc[i], to = lo[b] + marked[b];          // move i into the marked prefix of its block
                    if (at < to) continue;                                       // (already marked)
                    const uint32_t other = elems[to];
                    elems[to] = i; loc[i] = to; elems[at] = other; loc[other] = at;
                    if (!marked[b]++) touched.push_back(b);
                }
            for (uint32_t b : touched) {
                const uint32_t m = marked[b];
                marked[b] = 0;
                if (m == hi[b] - lo[b]) continue;                                // the whole block goes there: no split
                const uint32_t nb = (uint32_t)lo.size();                         // the marked prefix becomes block nb
                lo.push_back(lo[b]); hi.push_back(lo[b] + m); marked.push_back(0);
                lo[b] += m;
                for (uint32_t q = lo[nb]; q < hi[nb]; q++) block[elems[q]] = nb;
                const bool nb_smaller = hi[nb] - lo[nb] <= hi[b] - lo[b];
                for (uint32_t c = 0; c < K; c++) {
                    if (queued.size() >= (size_t)(b + 1) * K && queued[(size_t)b * K + c]) enqueue(nb, c);
                    else enqueue(nb_smaller ? nb : b, c);
                }
            }
        }
        count = (uint32_t)lo.size();
    }
    // renumber: the dead block -> 0, the others in breadth-first order from the start state, classes ascending - a numbering
    // that depends on the language and its byte classes only, not on which equivalent automaton the subset construction saw
    std::vector<int64_t> id(count, -1);
    uint32_t next_id = 0;
    id[block[0]] = next_id++;
    {
        std::vector<uint32_t> rep(count, UINT32_MAX), bfs;
        for (uint32_t i = 0; i < D; i++) if (rep[block[i]] == UINT32_MAX) rep[block[i]] = i;
        if (id[block[1]] < 0) id[block[1]] = next_id++;
        bfs.push_back(block[1]);
        for (size_t h = 0; h < bfs.size(); h++)
            for (uint32_t k = 0; k < K; k++) {
                const uint32_t nb = block[nxt[(size_t)rep[bfs[h]] * K + k]];
                if (id[nb] < 0) { id[nb] = next_id++; bfs.push_back(nb); }
            }
        for (uint32_t b = 0; b < count; b++) if (id[b] < 0) id[b] = next_id++;    // (none: every set is reachable)
    }
    if (count > 65535) return false;
    d.nstates = count;
    d.accepting.assign(count, 0);
    d.next.assign((size_t)count * K, 0);
    bool any_escape = false;
    for (uint32_t i = 0; i < D; i++) any_escape = any_escape || kind[i] == 2;
    if (any_escape) d.escaped.assign(count, 0);
    for (uint32_t i = 0; i < D; i++) {
        uint32_t b = (uint32_t)id[block[i]];
        d.accepting[b] = kind[i] == 1;
        if (kind[i] == 2) d.escaped[b] = 1;
        for (uint32_t k = 0; k < K; k++) d.next[(size_t)b * K + k] = (uint16_t)id[block[nxt[(size_t)i * K + k]]];
    }
    d.start = (uint32_t)id[block[1]];
    d.accepts_empty = d.accepting[d.start];
    return true;
}
// Subset construction + minimisation.  `sticky`: the initial node stays in every set and no byte kills (the
// automaton of "anything, then the pattern": search_dfas below); class 0 then gets a real column.
static bool subset_construct(const Reduced &red, uint32_t max_states, bool sticky, DfaProgram &d) {
    d = DfaProgram();
    std::memcpy(d.cls, red.cls, sizeof d.cls);
    d.ncls = red.ncls;
    if (red.nodes.empty()) {
        d.nstates = 1; d.start = 0; d.accepting = {0}; d.next.assign(d.ncls, 0);
        return true;
    }
    const std::vector<Node> &nodes = red.nodes;
    const uint32_t K = red.ncls, N = (uint32_t)nodes.size();
    // in-class membership of every node: bit k of member[u] words
    std::vector<std::vector<uint8_t>> enters(N, std::vector<uint8_t>(K, 0));
    for (uint32_t u = 1; u < N; u++) for (uint32_t k = 1; k < K; k++) enters[u][k] = nodes[u].label.has(red.cls_rep[k]);
    std::map<std::vector<uint32_t>, uint32_t> ids;
    std::vector<std::vector<uint32_t>> sets;
    std::vector<uint32_t> nxt;                          // raw [state][class]
    size_t budget = (size_t)24 << 20;                   // total set elements: give up early on exploding automata
    auto intern = [&](std::vector<uint32_t> &&v) -> int64_t {
        auto it = ids.find(v);
        if (it != ids.end()) return it->second;
        if (sets.size() >= max_states || v.size() > budget) return -1;
        budget -= v.size();
        uint32_t id = (uint32_t)sets.size();
        ids.emplace(v, id);
        sets.push_back(std::move(v));
        return id;
    };
    intern({});                                         // 0 = dead (unreachable when sticky)
    intern({0});                                        // 1 = start: the initial node
    std::vector<uint32_t> all, seen_at(N, 0);
    uint32_t epoch = 0;
    for (uint32_t cur = 0; cur < sets.size(); cur++) {
        nxt.resize((size_t)(cur + 1) * K, 0);
        all.clear();                                    // the union of the follow sets: each node once, then sorted
        epoch++;
        for (uint32_t u : sets[cur]) for (uint32_t v : nodes[u].follow) if (seen_at[v] != epoch) { seen_at[v] = epoch; all.push_back(v); }
        std::sort(all.begin(), all.end());
        for (uint32_t k = sticky ? 0 : 1; k < K; k++) {
            std::vector<uint32_t> v;
            if (sticky && cur != 0) v.push_back(0);     // node 0 sorts first; it is entered on nothing
            if (k) for (uint32_t x : all) if (enters[x][k]) v.push_back(x);
            int64_t id = intern(std::move(v));
            if (id < 0) return false;
            nxt[(size_t)cur * K + k] = (uint32_t)id;
        }
    }
    const uint32_t D = (uint32_t)sets.size();
    std::vector<uint8_t> kind(D, 0);
    for (uint32_t i = 0; i < D; i++) for (uint32_t u : sets[i]) if (nodes[u].fin) { kind[i] = 1; break; }
    return minimise_into(nxt, kind, K, d);
}
bool lower_dfa(const Reduced &red, uint32_t max_states, DfaProgram &d) { return subset_construct(red, max_states, false, d); }

bool lower_dfa_sampled(const Reduced &red, const uint8_t *sample, uint32_t pieces, uint32_t piece_bytes, uint32_t max_states, DfaProgram &d,
                       SampledTableStats *stats) {
    d = DfaProgram();
    std::memcpy(d.cls, red.cls, sizeof d.cls);
    d.ncls = red.ncls;
    if (red.nodes.empty() || !sample || max_states < 4) return false;
    const std::vector<Node> &nodes = red.nodes;
    const uint32_t K = red.ncls, N = (uint32_t)nodes.size();
    std::vector<std::vector<uint8_t>> enters(N, std::vector<uint8_t>(K, 0));
    for (uint32_t u = 1; u < N; u++) for (uint32_t k = 1; k < K; k++) enters[u][k] = nodes[u].label.has(red.cls_rep[k]);
    constexpr uint32_t kOpen = UINT32_MAX, kEscape = UINT32_MAX - 1;
    std::map<std::vector<uint32_t>, uint32_t> ids;
    std::vector<std::vector<uint32_t>> sets, unions;    // unions[i]: the union of the follow sets of set i (made when first needed)
    std::vector<uint8_t> has_union;
    std::vector<uint32_t> nxt;                          // [set][class]: set id, kOpen, kEscape
    const uint32_t budget = max_states - 1;             // (one row is the escape state)
    size_t elements = (size_t)8 << 20;
    auto add_set = [&](std::vector<uint32_t> &&v) -> uint32_t {
        const uint32_t id = (uint32_t)sets.size();
        ids.emplace(v, id);
        elements -= std::min(elements, v.size());
        sets.push_back(std::move(v));
        unions.emplace_back(); has_union.push_back(0);
        nxt.resize((size_t)(id + 1) * K, kOpen);
        nxt[(size_t)id * K] = 0;                        // class 0: nothing moves - the dead set
        return id;
    };
    add_set({});                                        // 0 = dead
    for (uint32_t k = 0; k < K; k++) nxt[k] = 0;
    add_set({0});                                       // 1 = start
    std::vector<uint32_t> seen_at(N, 0);
    uint32_t epoch = 0;
    auto resolve = [&](uint32_t cur, uint32_t k) -> uint32_t {      // the transition (cur, k): an id, or kEscape when the budget is spent
        uint32_t &slot = nxt[(size_t)cur * K + k];
        if (slot != kOpen) return slot;
        if (!has_union[cur]) {
            epoch++;
            for (uint32_t u : sets[cur]) for (uint32_t v : nodes[u].follow) if (seen_at[v] != epoch) { seen_at[v] = epoch; unions[cur].push_back(v); }
            std::sort(unions[cur].begin(), unions[cur].end());
            has_union[cur] = 1;
        }
        std::vector<uint32_t> v;
        for (uint32_t x : unions[cur]) if (enters[x][k]) v.push_back(x);
        auto it = ids.find(v);
        uint32_t to;
        if (it != ids.end()) to = it->second;
        else if (sets.size() < budget && v.size() <= elements) to = add_set(std::move(v));
        else to = kEscape;
        return nxt[(size_t)cur * K + k] = to;           // (add_set may have moved nxt: index again)
    };
    SampledTableStats st;
    // ---- the sets the sample reaches
    for (uint32_t p = 0; p < pieces; p++) {
        const uint8_t *t = sample + (size_t)p * piece_bytes;
        uint32_t i = 0;
        while (i < piece_bytes && t[i] != '\n') i++;    // a piece begins inside somebody's line: enter at the first line start
        i++;
        uint32_t cur = 1;
        if (i < piece_bytes) st.sample_lines++;
        for (; i < piece_bytes; i++) {
            const uint8_t c = t[i];
            if (c == '\n') { cur = 1; if (i + 1 < piece_bytes) st.sample_lines++; continue; }
            if (cur == kEscape) continue;               // until the end of the line
            st.sample_bytes_stepped++;
            cur = resolve(cur, c < 0x80 ? red.cls[c] : 0);
            if (cur == kEscape) st.sample_escapes++;
        }
    }
    st.sets_from_sample = (uint32_t)sets.size();
    if (!st.sample_bytes_stepped) return false;
    // ---- closure: the transitions still open, breadth-first over the sets in the order they were found
    for (uint32_t cur = 0; cur < sets.size(); cur++)
        for (uint32_t k = 1; k < K; k++) (void)resolve(cur, k);
    st.sets_from_closure = (uint32_t)sets.size() - st.sets_from_sample;
    // ---- the table: the escape state last; it leaves on class 0 only (whatever the set was, that byte kills it: a plain reject)
    const uint32_t D = (uint32_t)sets.size() + 1, esc = D - 1;
    std::vector<uint32_t> full((size_t)D * K);
    std::vector<uint8_t> kind(D, 0);
    for (uint32_t i = 0; i + 1 < D; i++) {
        for (uint32_t u : sets[i]) if (nodes[u].fin) { kind[i] = 1; break; }
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t to = nxt[(size_t)i * K + k];
            if (to == kEscape) st.open_transitions++;
            full[(size_t)i * K + k] = to == kEscape ? esc : to;
        }
    }
    kind[esc] = 2;
    full[(size_t)esc * K] = 0;
    for (uint32_t k = 1; k < K; k++) full[(size_t)esc * K + k] = esc;
    if (stats) *stats = st;
    if (!st.open_transitions) kind[esc] = 0;            // the closure closed everything: an ordinary table (the escape row is unreachable)
    return minimise_into(full, kind, K, d);
}

// The pattern read right to left, as a graph of the same shape: node w stands for "a byte of label(w) has just been
// consumed backwards and the automaton is in the predecessors of w"; it is final iff the initial node is a predecessor.
static Reduced reversed(const Reduced &r) {
    Reduced o;
    o.ncls = r.ncls;
    std::memcpy(o.cls, r.cls, sizeof o.cls);
    o.cls_rep = r.cls_rep;
    const size_t N = r.nodes.size();
    if (!N) return o;
    o.nodes.resize(N);
    o.nodes[0].fin = r.nodes[0].fin;
    for (size_t w = 1; w < N; w++) {
        o.nodes[w].label = r.nodes[w].label;
        if (r.nodes[w].fin) o.nodes[0].follow.push_back((uint32_t)w);
    }
    for (size_t u = 0; u < N; u++)
        for (uint32_t w : r.nodes[u].follow) {
            if (u == 0) o.nodes[w].fin = true;
            else o.nodes[w].follow.push_back((uint32_t)u);
        }
    for (auto &n : o.nodes) sort_unique(n.follow);
    return o;
}
bool search_dfas(const Reduced &r, uint32_t max_states, DfaProgram &fwd, DfaProgram &rev) {
    return subset_construct(r, max_states, true, fwd) && subset_construct(reversed(r), max_states, false, rev);
}

// Line-mode search table: the product of the sticky forward table (where does the first match END) and the anchored table
// of the pattern itself (is the line's prefix up to here accepted: then the match STARTS at the line start and no walk back
// is needed).  Rows = reachable pairs + SKIP (last), columns = byte classes + '\n' (last).
bool lower_search_line(const DfaProgram &fwd, const DfaProgram *anchored, uint32_t max_rows, SearchLineProgram &o) {
    o = SearchLineProgram();
    const uint32_t K = fwd.ncls;
    if ((anchored && anchored->ncls != K) || fwd.accepting[fwd.start]) return false;   // (patterns that accept "" take another path)
    // without the anchored table: a one-state stand-in that accepts nothing - the rows are the forward states, no hit is anchored
    DfaProgram none;
    if (!anchored) { none.nstates = 1; none.ncls = K; none.start = 0; none.accepting.assign(1, 0); none.next.assign(K, 0); }
    const DfaProgram &anch = anchored ? *anchored : none;
    std::map<std::pair<uint32_t, uint32_t>, uint32_t> ids;               // (anchored state, sticky state) -> row
    std::vector<std::pair<uint32_t, uint32_t>> pairs;
    auto intern = [&](uint32_t a, uint32_t f) -> int64_t {
        auto it = ids.find({a, f});
        if (it != ids.end()) return it->second;
        if (pairs.size() + 1 >= max_rows) return -1;
        ids.emplace(std::make_pair(a, f), (uint32_t)pairs.size());
        pairs.push_back({a, f});
        return (int64_t)pairs.size() - 1;
    };
    if (intern(anch.start, fwd.start) < 0) return false;
    std::vector<uint32_t> raw;                                            // [pair][K]: next pair | flags, SKIP = 0xffff
    for (size_t cur = 0; cur < pairs.size(); cur++) {
        const uint32_t a = pairs[cur].first, f = pairs[cur].second;
        raw.resize((cur + 1) * K);
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t a2 = k ? anch.next[(size_t)a * K + k] : 0u;  // class 0 kills the anchored run
            const uint32_t f2 = fwd.next[(size_t)f * K + k];
            if (fwd.accepting[f2]) raw[cur * K + k] = 0xffffu | kSearchHit | (anch.accepting[a2] ? kSearchAnchored : 0u);
            else {
                const int64_t id = intern(a2, f2);
                if (id < 0) return false;
                raw[cur * K + k] = (uint32_t)id;
            }
        }
    }
    const uint32_t P = (uint32_t)pairs.size();
    o.nrows = P + 1; o.ncols = K + 1; o.start = 0; o.skip = P;
    o.table.assign((size_t)o.nrows * o.ncols, 0);
    for (uint32_t r = 0; r < P; r++) {
        for (uint32_t k = 0; k < K; k++) {
            const uint32_t v = raw[(size_t)r * K + k];
            o.table[(size_t)r * o.ncols + k] = ((v & 0xffffu) == 0xffffu ? P : (v & 0xffffu)) | (v & 0xffff0000u);
        }
        o.table[(size_t)r * o.ncols + K] = o.start | kSearchNewline;
    }
    for (uint32_t k = 0; k < K; k++) o.table[(size_t)P * o.ncols + k] = P;
    o.table[(size_t)P * o.ncols + K] = o.start | kSearchNewline;
    return true;
}

// Stride-2 form of the line-mode search table: both forms (first match: a hit leads to SKIP; all matches: back to the start
// row) composed with themselves, the pair columns shared between them so that one pair table serves both.
bool lower_search_line2(const SearchLineProgram &s, const uint32_t *column, uint32_t max_cols, SearchLine2Program &o) {
    o = SearchLine2Program();
    const uint32_t R = s.nrows, C = s.ncols;
    if (!R || (uint64_t)C * C * R > ((uint64_t)1 << 28)) return false;              // (host work: two passes over C * C * R entries)
    o.nrows = R; o.start = s.start; o.skip = s.skip;
    auto flags_of = [](uint32_t v) -> uint32_t { return (v & kSearchNewline) ? 1u : (v & kSearchHit) ? ((v & kSearchAnchored) ? 3u : 2u) : 0u; };
    // one step of either form: -> next row, flags
    auto step = [&](uint32_t row, uint32_t sym, bool restart, uint32_t &f) -> uint32_t {
        const uint32_t v = s.table[(size_t)row * C + sym];
        f = flags_of(v);
        return (restart && (f & 2u)) ? s.start : (v & 0xffffu);
    };
    std::unordered_map<uint64_t, std::vector<uint32_t>> by_hash;                    // hash of a column -> columns with that hash
    std::vector<std::vector<uint32_t>> col_data;                                    // [column][2 * R]: the first form, then the restart form
    std::vector<uint32_t> sym_pair_col((size_t)C * C);
    std::vector<uint32_t> col(2 * (size_t)R);
    for (uint32_t a = 0; a < C; a++)
        for (uint32_t b = 0; b < C; b++) {
            uint64_t h = 1469598103934665603ull;
            for (uint32_t form = 0; form < 2; form++)
                for (uint32_t r = 0; r < R; r++) {
                    uint32_t f1, f2;
                    const uint32_t r1 = step(r, a, form != 0, f1);
                    const uint32_t r2 = step(r1, b, form != 0, f2);
                    const uint32_t v = r2 | (f1 << 2 | f2) << 24;
                    col[(size_t)form * R + r] = v;
                    h = (h ^ v) * 1099511628211ull;
                }
            uint32_t id = UINT32_MAX;
            for (uint32_t cand : by_hash[h])
                if (col_data[cand] == col) { id = cand; break; }
            if (id == UINT32_MAX) {
                if (col_data.size() >= max_cols) return false;
                id = (uint32_t)col_data.size();
                by_hash[h].push_back(id);
                col_data.push_back(col);
            }
            sym_pair_col[(size_t)a * C + b] = id;
        }
    o.ncols = (uint32_t)col_data.size();
    o.first.assign((size_t)R * o.ncols, 0);
    o.all.assign((size_t)R * o.ncols, 0);
    for (uint32_t c = 0; c < o.ncols; c++)
        for (uint32_t r = 0; r < R; r++) {
            o.first[(size_t)r * o.ncols + c] = col_data[c][r];
            o.all[(size_t)r * o.ncols + c] = col_data[c][(size_t)R + r];
        }
    o.pair_col.assign(128 * 128, 0);
    for (unsigned c1 = 0; c1 < 128; c1++)
        for (unsigned c2 = 0; c2 < 128; c2++) o.pair_col[c1 * 128 + c2] = (uint16_t)sym_pair_col[(size_t)column[c1] * C + column[c2]];
    return true;
}

// ------------------------------------------------------------------------------------------ stride-2 lowering
bool lower_dfa2(const DfaProgram &d, uint32_t max_cols, Dfa2Program &o, bool items) {
    o = Dfa2Program();
    const uint32_t D = d.nstates, K = d.ncls, NL = K;            // symbols 0..K-1 = byte classes, K = '\n'
    o.nstates = D; o.start = d.start; o.accepts_empty = d.accepts_empty;
    const bool two_bit = !d.escaped.empty();                     // a line end reports (accepted, escaped)
    const uint32_t kb = two_bit ? 2u : 1u;
    auto step = [&](uint32_t s, uint32_t sym, uint32_t &line, uint32_t &verdict) -> uint32_t {
        if (sym == NL) { line = 1; verdict = two_bit ? (uint32_t)d.accepting[s] << 1 | d.escaped[s] : d.accepting[s]; return d.start; }
        line = 0; verdict = 0;
        return d.next[(size_t)s * K + sym];
    };
    std::map<std::vector<uint32_t>, uint32_t> cols;
    std::vector<std::vector<uint32_t>> col_data;
    std::vector<uint32_t> sym_pair_col((size_t)(K + 1) * (K + 1));
    for (uint32_t a = 0; a <= K; a++)
        for (uint32_t b = 0; b <= K; b++) {
            std::vector<uint32_t> col(D);
            for (uint32_t s = 0; s < D; s++) {
                uint32_t l1, v1, l2, v2;
                const uint32_t s1 = step(s, a, l1, v1);
                const uint32_t s2 = step(s1, b, l2, v2);
                const uint32_t verdicts = (l1 && l2) ? (v1 << kb | v2) : l1 ? v1 : v2;
                col[s] = s2 | ((l1 + l2) * kb) << 16 | verdicts << 24;
            }
            auto it = cols.find(col);
            if (it == cols.end()) {
                if (col_data.size() >= max_cols) return false;
                it = cols.emplace(col, (uint32_t)col_data.size()).first;
                col_data.push_back(std::move(col));
            }
            sym_pair_col[(size_t)a * (K + 1) + b] = it->second;
        }
    o.ncols = (uint32_t)col_data.size();
    o.next2.assign((size_t)D * o.ncols, 0);
    for (uint32_t c = 0; c < o.ncols; c++) for (uint32_t s = 0; s < D; s++) o.next2[(size_t)s * o.ncols + c] = col_data[c][s];
    const unsigned dim = items ? 129u : 128u;
    o.pair_dim = dim;
    o.pair_col.assign((size_t)dim * dim, 0);
    auto sym = [&](unsigned c) -> uint32_t { return items ? (c == 128 ? NL : d.cls[c]) : (c == '\n' ? NL : d.cls[c]); };
    for (unsigned c1 = 0; c1 < dim; c1++)
        for (unsigned c2 = 0; c2 < dim; c2++) o.pair_col[c1 * dim + c2] = (uint16_t)sym_pair_col[(size_t)sym(c1) * (K + 1) + sym(c2)];
    return true;
}

// ------------------------------------------------------------------------------------------ stride-2 table order
Dfa2OrderStats order_dfa2(const Dfa2Program &d, const uint8_t *sample, uint32_t lanes, uint32_t bytes_per_lane,
                          std::vector<uint32_t> &row_slot, std::vector<uint32_t> &col_slot) {
    Dfa2OrderStats st;
    const uint32_t D = d.nstates, C = d.ncols, P = C | 1u;
    row_slot.resize(D); col_slot.resize(C);
    for (uint32_t i = 0; i < D; i++) row_slot[i] = i;
    for (uint32_t i = 0; i < C; i++) col_slot[i] = i;
    const uint32_t groups = lanes / 32, steps = bytes_per_lane / 2;
    if (!groups || !steps || D < 2 || C < 2) return st;
    // ---- the half-waves: per group and step the DISTINCT (row, column) pairs its 32 lanes look up
    struct Entry { uint16_t row, col; };
    std::vector<Entry> entries;
    std::vector<uint32_t> first;                                  // half-wave h: entries[first[h] .. first[h + 1])
    std::vector<uint64_t> freq_row(D, 0), freq_col(C, 0);
    std::vector<uint32_t> state(32);
    for (uint32_t g = 0; g < groups; g++) {
        std::fill(state.begin(), state.end(), 0u);                // dead: a stripe begins inside somebody else's line
        for (uint32_t k = 0; k < steps; k++) {
            first.push_back((uint32_t)entries.size());
            Entry hw[32];
            uint32_t n = 0;
            for (uint32_t l = 0; l < 32; l++) {
                const uint8_t *t = sample + ((size_t)g * 32 + l) * bytes_per_lane + 2 * k;
                const uint32_t c1 = t[0] < 0x80 ? t[0] : 0, c2 = t[1] < 0x80 ? t[1] : 0;       // (a corpus with bytes >= 0x80 does not get here)
                const uint32_t col = d.pair_col[c1 * 128 + c2], row = state[l];
                freq_row[row]++; freq_col[col]++;
                bool seen = false;
                for (uint32_t q = 0; q < n && !seen; q++) seen = hw[q].row == row && hw[q].col == col;
                if (!seen) hw[n++] = Entry{(uint16_t)row, (uint16_t)col};
                state[l] = d.next2[(size_t)row * C + col] & 0xffffu;
            }
            entries.insert(entries.end(), hw, hw + n);
        }
    }
    first.push_back((uint32_t)entries.size());
    const uint32_t H = (uint32_t)first.size() - 1;
    st.half_waves = H;
    auto cost_of = [&](uint32_t h) -> uint32_t {                  // the fullest bank of half-wave h
        uint8_t bank[32] = {0};
        uint32_t worst = 0;
        for (uint32_t q = first[h]; q < first[h + 1]; q++) {
            const uint32_t b = (row_slot[entries[q].row] * P + col_slot[entries[q].col]) & 31u;
            if (++bank[b] > worst) worst = bank[b];
        }
        return worst;
    };
    std::vector<uint8_t> hw_cost(H);
    uint64_t best = 0;
    for (uint32_t h = 0; h < H; h++) { hw_cost[h] = (uint8_t)cost_of(h); best += hw_cost[h]; }
    st.before = (double)best / H;
    // a swap only changes the half-waves that hold an entry of one of the two rows (columns): their lists
    std::vector<std::vector<uint32_t>> in_row(D), in_col(C);
    for (uint32_t h = 0; h < H; h++)
        for (uint32_t q = first[h]; q < first[h + 1]; q++) {
            std::vector<uint32_t> &r = in_row[entries[q].row], &c = in_col[entries[q].col];
            if (r.empty() || r.back() != h) r.push_back(h);
            if (c.empty() || c.back() != h) c.push_back(h);
        }
    std::vector<uint32_t> stamp(H, 0), touched;
    uint32_t epoch = 0;
    constexpr uint32_t kTop = 32;                                 // the hottest rows, then the hottest columns
    for (int pass = 0; pass < 2; pass++) {                        // pass 0: rows, pass 1: columns
        std::vector<uint32_t> &slot = pass == 0 ? row_slot : col_slot;
        const std::vector<uint64_t> &freq = pass == 0 ? freq_row : freq_col;
        const std::vector<std::vector<uint32_t>> &in = pass == 0 ? in_row : in_col;
        const uint32_t n = pass == 0 ? D : C, mult = pass == 0 ? P : 1u;
        std::vector<uint32_t> order(n);
        for (uint32_t i = 0; i < n; i++) order[i] = i;
        std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return freq[a] != freq[b] ? freq[a] > freq[b] : a < b; });
        // the change of the total if k and j trade slots (the slots are left traded; the caller trades back or keeps)
        auto trade = [&](uint32_t k, uint32_t j) -> int64_t {
            std::swap(slot[k], slot[j]);
            epoch++;
            touched.clear();
            int64_t delta = 0;
            for (const std::vector<uint32_t> *list : {&in[k], &in[j]})
                for (uint32_t h : *list) {
                    if (stamp[h] == epoch) continue;
                    stamp[h] = epoch;
                    touched.push_back(h);
                    delta += (int64_t)cost_of(h) - hw_cost[h];
                }
            st.evaluations++;
            return delta;
        };
        for (uint32_t oi = 0; oi < n && oi < kTop; oi++) {
            const uint32_t k = order[oi];
            if (!freq[k] || (pass == 0 && k == 0)) continue;
            // candidates: for every other bank residue, the least used member that sits there
            int64_t cand[32];
            for (int r = 0; r < 32; r++) cand[r] = -1;
            for (uint32_t j = 0; j < n; j++) {
                if (j == k || (pass == 0 && j == 0)) continue;
                const uint32_t r = (slot[j] * mult) & 31u;
                if (cand[r] < 0 || freq[j] < freq[(size_t)cand[r]]) cand[r] = j;
            }
            int64_t pick = -1, pick_delta = 0;
            const uint32_t mine = (slot[k] * mult) & 31u;
            for (int r = 0; r < 32; r++) {
                if (cand[r] < 0 || (uint32_t)r == mine) continue;
                const uint32_t j = (uint32_t)cand[r];
                const int64_t d = trade(k, j);
                if (d < pick_delta) { pick_delta = d; pick = j; }
                std::swap(slot[k], slot[j]);
            }
            if (pick >= 0) {
                (void)trade(k, (uint32_t)pick);
                for (uint32_t h : touched) hw_cost[h] = (uint8_t)cost_of(h);
                best = (uint64_t)((int64_t)best + pick_delta);
            }
        }
    }
    st.after = (double)best / H;
    return st;
}

// ------------------------------------------------------------------------------------------ work beside the caller
bool OnceTask::start(std::function<void()> job, bool background) {
    std::unique_lock<std::mutex> lock(mu_);        // held from the decision to the thread being in place: wait() sees both or neither
    int expected = kIdle;
    if (!state_.compare_exchange_strong(expected, kRunning, std::memory_order_acq_rel)) return false;
    auto body = [this](std::function<void()> fn) {
        fn();
        state_.store(kDone, std::memory_order_release);
    };
    if (!background) { lock.unlock(); body(std::move(job)); return true; }
    thread_ = std::thread(body, std::move(job));
    return true;
}
bool OnceTask::skip() {
    int expected = kIdle;
    return state_.compare_exchange_strong(expected, kSkipped, std::memory_order_acq_rel);
}
void OnceTask::wait() {
    std::lock_guard<std::mutex> lock(mu_);
    if (thread_.joinable()) thread_.join();
}
bool TableOrderSearch::start(const Dfa2Program &d, std::vector<uint8_t> sample, uint32_t lanes, uint32_t bytes_per_lane, bool background, Apply apply) {
    // (shared_ptr: std::function wants a copyable callable)
    auto text = std::make_shared<std::vector<uint8_t>>(std::move(sample));
    return OnceTask::start([&d, text, lanes, bytes_per_lane, apply]() {
        std::vector<uint32_t> rows, cols;
        const Dfa2OrderStats st = order_dfa2(d, text->data(), lanes, bytes_per_lane, rows, cols);
        apply(std::move(rows), std::move(cols), st);
    }, background);
}

}  // namespace rrx
