// frontend.cpp — pattern -> reference-numbered automaton.  See frontend.hpp.
//
// Structure: (1) a lexer that turns the pattern bytes into tokens, reproducing the byte-level rules of
// the reference parser (escapes Parser.cpp:88-91, bracket expressions Parser.cpp:16-39, "{m,n}"
// Parser.cpp:123-141, anchors Parser.cpp:142-146); (2) a two-stack machine that evaluates the tokens with
// the reference's deferred-folding discipline (Parser.cpp:49-83), on (3) an edge-labelled digraph with the
// reference's eps-free algebra (NFA.cc:42-71, 108-185).
#include "frontend.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

namespace rrx {
namespace {

// ------------------------------------------------------------------------------------------ graph
// The automaton under construction, rows as DAGs of pieces (frontend.hpp).  Every operation costs what it touches, not
// what the reference's eager copies would: x{1,n} is O(n) pieces where the explicit graph has n^2/2 edges.
class Graph {
public:
    std::vector<RowPiece> pieces;
    std::vector<uint32_t> head;                  // per state
    std::vector<std::vector<uint32_t>> occ;      // occ[t]: the pieces that hold a direct edge to t (each once)
    uint64_t work = 0;                           // pieces + direct edges made so far

    void ensure(uint64_t upto) {
        if (upto > kMaxStates) throw PatternError("too many states (> 65536)");
        if (head.size() < upto) { head.resize(upto, kNoPiece); occ.resize(upto); }
    }
    void spend(uint64_t units) {
        work += units;
        if (work > kFrontEndBudget) throw BudgetError("pattern too large: the automaton exceeds the front end's work budget");
    }
    uint32_t new_piece() {
        spend(1);
        pieces.emplace_back();
        return (uint32_t)pieces.size() - 1;
    }
    void add_direct(uint32_t piece, uint32_t t, const CharSet &on) {
        if (on.empty()) return;
        auto &d = pieces[piece].direct;
        auto it = std::lower_bound(d.begin(), d.end(), t, [](const Edge &e, uint32_t v) { return e.to < v; });
        if (it != d.end() && it->to == t) it->on |= on;
        else {
            spend(1);
            d.insert(it, Edge{t, on});
            occ[t].push_back(piece);
        }
    }
    // an edge out of a state nobody has copied from yet (atoms: NFA.cc:50-71)
    void connect_fresh(uint32_t s, uint32_t t, const CharSet &on) {
        if (head[s] == kNoPiece) head[s] = new_piece();
        add_direct(head[s], t, on);
    }
    // NFA.cc:108-121 skip<true>(n,k): n takes over every out-edge k has NOW ("eps n->k", resolved eagerly).
    void inherit_out(uint32_t n, uint32_t k) {
        if (n == k || head[k] == kNoPiece) return;
        const uint32_t hk = head[k];
        if (head[n] == kNoPiece) head[n] = new_piece();
        else if (pieces[head[n]].shared) {       // somebody holds a snapshot of row(n): it must not grow with n
            const uint32_t old = head[n], fresh = new_piece();
            pieces[fresh].children.push_back(old);
            head[n] = fresh;
        }
        auto &ch = pieces[head[n]].children;
        if (std::find(ch.begin(), ch.end(), hk) == ch.end()) { spend(1); ch.push_back(hk); }
        pieces[hk].shared = true;
    }
    // NFA.cc:108-121 skip<false>(n,k): whatever enters n also enters k, on the same characters.  A state j enters n on
    // the OR of the labels n carries in the pieces of row(j); adding k beside every such entry gives j exactly that.
    void mirror_in(uint32_t n, uint32_t k) {
        if (n == k) return;
        const std::vector<uint32_t> holders = occ[n];
        for (uint32_t piece : holders) {
            const auto &d = pieces[piece].direct;
            auto it = std::lower_bound(d.begin(), d.end(), n, [](const Edge &e, uint32_t v) { return e.to < v; });
            const CharSet on = it->on;
            add_direct(piece, k, on);
        }
    }
    // NFA.cc:177-185: the states [lo, lo+size) again at [lo+rot, ...).  A fragment on the stack is self-contained (no edge
    // leaves its state range, nobody outside holds one of its pieces), so its pieces are copied one for one, sharing kept.
    void copy_shifted(uint32_t lo, uint32_t size, uint32_t rot) {
        std::vector<uint32_t> stack;
        auto copy_of = [&](uint32_t old) -> int64_t {
            auto it = copied.find(old);
            return it == copied.end() ? -1 : (int64_t)it->second;
        };
        copied.clear();
        for (uint32_t s = lo; s < lo + size; s++) {
            if (head[s] == kNoPiece) continue;
            stack.push_back(head[s]);
            while (!stack.empty()) {                              // post-order without recursion: chains are thousands deep
                const uint32_t p = stack.back();
                if (copy_of(p) >= 0) { stack.pop_back(); continue; }
                bool ready = true;
                for (uint32_t c : pieces[p].children) if (copy_of(c) < 0) { stack.push_back(c); ready = false; }
                if (!ready) continue;
                stack.pop_back();
                const uint32_t q = new_piece();
                spend(pieces[p].direct.size() + pieces[p].children.size());
                pieces[q].shared = pieces[p].shared;
                pieces[q].direct = pieces[p].direct;
                for (Edge &e : pieces[q].direct) { e.to += rot; occ[e.to].push_back(q); }
                pieces[q].children.reserve(pieces[p].children.size());
                for (uint32_t c : pieces[p].children) pieces[q].children.push_back((uint32_t)copy_of(c));
                copied.emplace(p, q);
            }
            head[s + rot] = (uint32_t)copy_of(head[s]);
        }
    }
private:
    std::unordered_map<uint32_t, uint32_t> copied;
};

// A sub-automaton occupying states [initial, initial+size): regex.h:78-96 + final_states (regex.h:177).
struct Frag {
    uint32_t initial = 0, size = 0;
    std::vector<uint32_t> finals;                // sorted
    bool nullable() const { return std::binary_search(finals.begin(), finals.end(), initial); }
    void mark_final(uint32_t s) {
        auto it = std::lower_bound(finals.begin(), finals.end(), s);
        if (it == finals.end() || *it != s) finals.insert(it, s);
    }
};

// ------------------------------------------------------------------------------------------ tokens
struct Token {
    enum Kind { ATOM, OPEN, CLOSE, ALT, STAR, PLUS, OPT, REPEAT } kind;
    CharSet on;            // ATOM
    long m = 0, n = 0;     // REPEAT
    bool has_tail = false; // REPEAT: a second number was parsed ("{m?n}")
};

class Lexer {
    const char *p;          // NUL-terminated pattern (the byte rules read the terminator, like the reference does)
    size_t ps;
public:
    explicit Lexer(const std::string &s) : p(s.c_str()), ps(s.size()) {}

    // Parser.cpp:16-39.  s0 = index of '['.  Returns the index the cursor is left on.
    size_t bracket(size_t s0, CharSet &set) const {
        bool escaped = false;
        size_t q = s0 + 1;
        const bool negate = p[q] == '^';                       // the '^' itself is NOT consumed (Parser.cpp:18)
        while (q + 1 < ps && !(p[q] == ']' && !escaped)) {
            const unsigned char cur = (unsigned char)p[q];
            if (cur >= 0x80) throw PatternError("non-ASCII byte in pattern");
            if (!escaped) {
                const char nx = p[q + 1];
                if (nx != ']' && q + 2 < ps) {
                    const unsigned char hi = (unsigned char)p[q + 2];
                    if (nx == '-' && hi != ']') {
                        if (hi >= 0x7f) throw PatternError("bracket range end out of range");
                        for (unsigned c = cur; c <= hi; c++) set.add(c);
                        q += 3;
                        continue;
                    }
                }
            }
            escaped = !escaped && cur == '\\';
            set.add(cur);
            q++;
        }
        if (q == ps) throw PatternError("invalid expression!");   // Parser.cpp:36
        if (negate) set = set.inverted();                         // over all 128 codes, BitSet.cc:42-56
        return q;
    }

    std::vector<Token> run() const {
        std::vector<Token> toks;
        size_t cp = 0;
        bool escaped = false;
        do {                                                       // Parser.cpp:87-153 (runs once even for "")
            const unsigned char ch = (unsigned char)p[cp];
            if (ch >= 0x80) throw PatternError("non-ASCII byte in pattern");
            if (!escaped && ch == '\\') { escaped = true; continue; }
            Token t{};
            switch (escaped ? 0 : ch) {
            case '[': t.kind = Token::ATOM; cp = bracket(cp, t.on); break;
            case '(': t.kind = Token::OPEN; break;
            case ')': t.kind = Token::CLOSE; break;
            case '|': t.kind = Token::ALT; break;
            case '*': t.kind = Token::STAR; break;
            case '+': t.kind = Token::PLUS; break;
            case '?': t.kind = Token::OPT; break;
            case '.': t.kind = Token::ATOM; t.on = CharSet::all(); break;          // Parser.cpp:106-109
            case '^':
            case '$': t.kind = Token::ATOM; t.on = CharSet::single(0); break;      // Parser.cpp:142-146
            case '{': {                                                            // Parser.cpp:123-141
                t.kind = Token::REPEAT;
                char *e1;
                t.m = std::strtol(p + cp + 1, &e1, 10);
                if (*e1 != '}') {
                    if (*e1 == '\0') throw PatternError("invalid expression (unterminated {)");
                    char *e2;
                    t.n = std::strtol(e1 + 1, &e2, 10);
                    t.has_tail = true;
                    cp = (size_t)(e2 - p);
                } else cp = (size_t)(e1 - p);
                if (t.m > (long)kMaxStates || t.n > (long)kMaxStates) throw PatternError("too many states (> 65536)");
                break;
            }
            default: t.kind = Token::ATOM; t.on = CharSet::single(ch); break;      // Parser.cpp:147-150
            }
            toks.push_back(t);
            escaped = false;
        } while (++cp < ps);
        return toks;
    }
};

// ------------------------------------------------------------------------------------------ machine
class Machine {
    enum Op { CONCAT, GROUP, ALTERNATE };                     // regex.h:15
    Graph g;
    std::vector<Frag> operands;                               // Parser.cpp:43
    std::vector<Op> operators;                                // Parser.cpp:44

    [[noreturn]] static void underflow() { throw PatternError("invalid expression (stack underflow)"); }
    Frag &top() { if (operands.empty()) underflow(); return operands.back(); }
    void pop_operator() { if (operators.empty()) underflow(); operators.pop_back(); }
    void pop_operand() { if (operands.empty()) underflow(); operands.pop_back(); }
    uint32_t next_free() const {                              // Parser.cpp:84-86
        return operands.empty() ? 0 : operands.back().initial + operands.back().size;
    }

    // ---- atoms, NFA.cc:42-71
    Frag empty_frag(uint32_t at) {
        g.ensure((uint64_t)at + 1);
        Frag f; f.initial = at; f.size = 1; f.finals = {at};
        return f;
    }
    Frag atom(uint32_t at, const CharSet &on) {
        g.ensure((uint64_t)at + 2);
        g.connect_fresh(at, at + 1, on);
        Frag f; f.initial = at; f.size = 2; f.finals = {at + 1};
        return f;
    }
    // ---- NFA.cc:122-137
    void concat(Frag &a, const Frag &b) {
        a.size += b.size;
        for (uint32_t f : a.finals) g.mirror_in(f, b.initial);
        const bool an = a.nullable();
        if (an) g.inherit_out(a.initial, b.initial);
        const bool both = an && b.nullable();
        a.finals = b.finals;
        if (both) a.mark_final(a.initial);
    }
    // ---- NFA.cc:138-149
    void unite(Frag &a, const Frag &b) {
        a.size += b.size;
        for (uint32_t f : b.finals) a.mark_final(f);
        g.inherit_out(a.initial, b.initial);
        if (b.nullable()) a.mark_final(a.initial);
    }
    // ---- NFA.cc:150-157
    void star(Frag &a) {
        for (uint32_t f : a.finals) g.mirror_in(f, a.initial);
        a.mark_final(a.initial);
    }
    // ---- NFA.cc:177-185 (a fragment on the stack is self-contained: no edge leaves its state range)
    Frag shifted_copy(const Frag &a) {
        const uint32_t rot = a.size;
        g.ensure((uint64_t)a.initial + rot + a.size);
        g.copy_shifted(a.initial, a.size, rot);
        Frag r; r.initial = a.initial + rot; r.size = a.size;
        r.finals.reserve(a.finals.size());
        for (uint32_t f : a.finals) r.finals.push_back(f + rot);
        return r;
    }
    void duplicate_top() {                                    // Parser.cpp:80-83 repeat()
        Frag c = shifted_copy(top());
        operands.push_back(std::move(c));
        operators.push_back(CONCAT);
    }
    void make_top_optional() {                                // Parser.cpp:121 / 135
        Frag e = empty_frag(next_free());
        unite(top(), e);
    }
    void push_atom(const CharSet &on) {
        operands.push_back(atom(next_free(), on));
        operators.push_back(CONCAT);
    }

    // Parser.cpp:49-79.  Folds the operand stack down to the innermost open group.
    void fold() {
        Frag cur = std::move(top());
        if (!operators.empty()) {
            operators.pop_back();
            while (operators.size() > 1 && operators.back() != GROUP) {
                if (operators.back() == CONCAT) {
                    pop_operand(); pop_operator();
                    concat(top(), cur);
                    cur = std::move(top());
                } else {
                    Frag right = std::move(cur);
                    pop_operator();
                    pop_operator();
                    pop_operand();
                    cur = std::move(top());
                    while (operators.size() > 1 && operators.back() == CONCAT) {
                        pop_operand(); pop_operator();
                        concat(top(), cur);
                        cur = std::move(top());
                    }
                    unite(cur, right);
                }
            }
        }
        pop_operator();
        operators.push_back(CONCAT);
        pop_operand();
        operands.push_back(std::move(cur));
    }

public:
    RefAutomaton run(const std::vector<Token> &toks) {
        operators.push_back(GROUP);                           // Parser.cpp:45
        for (const Token &t : toks) {
            switch (t.kind) {
            case Token::ATOM: push_atom(t.on); break;
            case Token::OPEN: operators.push_back(GROUP); break;
            case Token::ALT: operators.push_back(ALTERNATE); break;
            case Token::CLOSE: fold(); break;
            case Token::STAR: star(top()); break;
            case Token::PLUS: duplicate_top(); star(top()); break;             // Parser.cpp:116-119
            case Token::OPT: make_top_optional(); break;
            case Token::REPEAT:
                for (long i = 0; i < t.m - 1; i++) duplicate_top();
                if (t.has_tail) {
                    if (!t.n) { duplicate_top(); star(top()); }
                    else if (t.n > t.m) {
                        duplicate_top();
                        make_top_optional();
                        for (long k = t.m + 1; k < t.n; k++) duplicate_top();
                    }
                }
                break;
            }
        }
        fold();                                               // Parser.cpp:154
        if (operands.size() != 1) throw PatternError("invalid expression");   // Parser.cpp:155
        const Frag &f = operands.back();
        RefAutomaton a;
        a.states_n = f.size;
        a.initial = f.initial;
        g.ensure(a.states_n);
        a.is_final.assign(a.states_n, 0);
        for (uint32_t s : f.finals) if (s < a.states_n) a.is_final[s] = 1;
        a.pieces = std::move(g.pieces);
        a.head = std::move(g.head);
        a.head.resize(a.states_n, kNoPiece);
        return a;
    }
};

}  // namespace

namespace {
// the pieces of row(state), each once
template <class F> void for_each_piece(const RefAutomaton &a, uint32_t state, F &&f) {
    if (state >= a.states_n || a.head[state] == kNoPiece) return;
    thread_local std::vector<uint32_t> stamp;
    thread_local uint32_t epoch = 0;
    if (stamp.size() < a.pieces.size()) stamp.resize(a.pieces.size(), 0);
    if (++epoch == 0) { std::fill(stamp.begin(), stamp.end(), 0u); epoch = 1; }
    std::vector<uint32_t> stack{a.head[state]};
    stamp[a.head[state]] = epoch;
    while (!stack.empty()) {
        const uint32_t p = stack.back(); stack.pop_back();
        f(a.pieces[p]);
        for (uint32_t c : a.pieces[p].children) if (stamp[c] != epoch) { stamp[c] = epoch; stack.push_back(c); }
    }
}
}  // namespace

std::vector<uint32_t> RefAutomaton::row(uint32_t state, unsigned c) const {
    std::vector<uint32_t> r;
    if (c >= 128) return r;
    for_each_piece(*this, state, [&](const RowPiece &p) { for (const Edge &e : p.direct) if (e.on.has(c)) r.push_back(e.to); });
    std::sort(r.begin(), r.end());
    r.erase(std::unique(r.begin(), r.end()), r.end());
    return r;
}

std::vector<Edge> RefAutomaton::expand_row(uint32_t state) const {
    std::vector<Edge> r;
    for_each_piece(*this, state, [&](const RowPiece &p) { r.insert(r.end(), p.direct.begin(), p.direct.end()); });
    std::sort(r.begin(), r.end(), [](const Edge &x, const Edge &y) { return x.to < y.to; });
    size_t w = 0;
    for (size_t i = 0; i < r.size(); i++) {
        if (w && r[w - 1].to == r[i].to) r[w - 1].on |= r[i].on;
        else r[w++] = r[i];
    }
    r.resize(w);
    return r;
}

RefAutomaton build_reference_automaton(const std::string &pattern) {
    if (pattern.find('\0') != std::string::npos) throw PatternError("NUL byte in pattern");
    Lexer lx(pattern);
    Machine m;
    return m.run(lx.run());
}

}  // namespace rrx
