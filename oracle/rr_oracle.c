/*
 * rr_oracle.c — CPU oracle for the RoaringRegex hot path.  TEST INFRASTRUCTURE ONLY (see rr_oracle.h).
 *
 * Plain-C restatement of the reference algorithm; citations are file:line under /root/reference/src.
 * Parity: pinned for <=256 states by tests/golden/kat.json (answers recorded from the reference's own
 * code); the >256-state class is "parity unpinned" (reference broken there) and follows
 * the intended semantics: the same construction with a full-width state index instead of NFA.cc:10's
 * (uint8_t) cast.
 *
 * Storage differs from the reference on purpose (sparse sorted rows during construction, so that the
 * 7786-state keyword automaton does not need a 1.9 GB dense table); the dense word-bitset execution
 * tables of the <=256-state classes are rebuilt from them and stepped exactly like NFA.cc:86-100.
 */
#include "rr_oracle.h"
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define RRO_MAX_STATES 65536u

/* ------------------------------------------------------------------ sparse exact sets ---- */
typedef struct { uint32_t *v; uint32_t n, cap; } set_t;

static void set_free(set_t *s) { free(s->v); s->v = NULL; s->n = s->cap = 0; }
static void set_reserve(set_t *s, uint32_t need) {
    if (need <= s->cap) return;
    uint32_t c = s->cap ? s->cap : 4;
    while (c < need) c *= 2;
    s->v = (uint32_t *)realloc(s->v, (size_t)c * sizeof(uint32_t));
    s->cap = c;
}
static int set_contains(const set_t *s, uint32_t x) {
    uint32_t lo = 0, hi = s->n;
    while (lo < hi) { uint32_t m = (lo + hi) / 2; if (s->v[m] < x) lo = m + 1; else hi = m; }
    return lo < s->n && s->v[lo] == x;
}
static void set_add(set_t *s, uint32_t x) {
    uint32_t lo = 0, hi = s->n;
    while (lo < hi) { uint32_t m = (lo + hi) / 2; if (s->v[m] < x) lo = m + 1; else hi = m; }
    if (lo < s->n && s->v[lo] == x) return;
    set_reserve(s, s->n + 1);
    memmove(s->v + lo + 1, s->v + lo, (size_t)(s->n - lo) * sizeof(uint32_t));
    s->v[lo] = x; s->n++;
}
static void set_or(set_t *d, const set_t *o) {           /* d |= o */
    if (!o->n) return;
    if (d == o) return;
    const uint32_t room = d->n + o->n;
    uint32_t *m = (uint32_t *)malloc((size_t)room * sizeof(uint32_t));
    uint32_t i = 0, j = 0, k = 0;
    while (i < d->n && j < o->n) {
        if (d->v[i] < o->v[j]) m[k++] = d->v[i++];
        else if (d->v[i] > o->v[j]) m[k++] = o->v[j++];
        else { m[k++] = d->v[i++]; j++; }
    }
    while (i < d->n) m[k++] = d->v[i++];
    while (j < o->n) m[k++] = o->v[j++];
    free(d->v); d->v = m; d->n = k; d->cap = room;
}
static void set_assign(set_t *d, const set_t *o) {
    if (d == o) return;
    d->n = 0; set_reserve(d, o->n);
    if (o->n) memcpy(d->v, o->v, (size_t)o->n * sizeof(uint32_t));
    d->n = o->n;
}
static void set_assign_shifted(set_t *d, const set_t *o, uint32_t rot) { /* NFA.cc:172-176 */
    d->n = 0; set_reserve(d, o->n);
    for (uint32_t i = 0; i < o->n; i++) d->v[i] = o->v[i] + rot;
    d->n = o->n;
}

/* ------------------------------------------------------------------ automaton ---- */
typedef struct { set_t r[256]; } state_rows;   /* r[2*c + !fwd], c in [0,128): regex.h:33-35 + NFA.cc:9-12 */

struct rro_nfa {
    uint32_t states_n, initial, size;
    state_rows **st;          /* per-state row blocks (storage only; idx() order is a layout detail) */
    uint32_t st_cap;
    set_t finals;
    int cls;                  /* Parser.cpp:165-168 */
    uint64_t *dense;          /* forward rows [c][state][W] for cls 1/2/4 (regex.h:33-35 forward half) */
    uint64_t dfinal[4];
    /* scratch of the sparse class (regex.h:52-58) */
    uint8_t *mark;
};

typedef struct { uint32_t initial, size; set_t finals; } sub_t;   /* regex.h:78-96 + 177 */

typedef struct {
    rro_nfa *n;
    sub_t *nfas; int nn, ncap;          /* Parser.cpp:43 */
    int *ops; int no, ocap;             /* Parser.cpp:44 */
    char err[160];
    int failed;
} build_t;

enum { OP_CONCATENATION, OP_BRACKETS, OP_OR };   /* regex.h:15 */

static void fail(build_t *b, const char *fmt, ...) {
    if (b->failed) return;
    va_list ap; va_start(ap, fmt); vsnprintf(b->err, sizeof b->err, fmt, ap); va_end(ap);
    b->failed = 1;
}

static set_t *row(rro_nfa *n, uint32_t state, unsigned c, int fwd) {
    if (state >= n->st_cap) {
        uint32_t c2 = n->st_cap ? n->st_cap : 16;
        while (c2 <= state) c2 *= 2;
        n->st = (state_rows **)realloc(n->st, (size_t)c2 * sizeof(*n->st));
        memset(n->st + n->st_cap, 0, (size_t)(c2 - n->st_cap) * sizeof(*n->st));
        n->st_cap = c2;
    }
    if (!n->st[state]) n->st[state] = (state_rows *)calloc(1, sizeof(state_rows));
    return &n->st[state]->r[2 * c + (fwd ? 0 : 1)];
}
static const set_t *crow(const rro_nfa *n, uint32_t state, unsigned c, int fwd) {
    static const set_t empty = {0, 0, 0};
    if (state >= n->st_cap || !n->st[state]) return &empty;
    return &n->st[state]->r[2 * c + (fwd ? 0 : 1)];
}

/* ---- stacks ---- */
static void push_nfa(build_t *b, sub_t s) {
    if (b->nn == b->ncap) { b->ncap = b->ncap ? 2 * b->ncap : 16; b->nfas = (sub_t *)realloc(b->nfas, (size_t)b->ncap * sizeof(sub_t)); }
    b->nfas[b->nn++] = s;
}
static void push_op(build_t *b, int op) {
    if (b->no == b->ocap) { b->ocap = b->ocap ? 2 * b->ocap : 16; b->ops = (int *)realloc(b->ops, (size_t)b->ocap * sizeof(int)); }
    b->ops[b->no++] = op;
}
static uint32_t next_initial(build_t *b) {     /* Parser.cpp:84-86 */
    return b->nn ? b->nfas[b->nn - 1].initial + b->nfas[b->nn - 1].size : 0;
}
static int check_states(build_t *b, uint64_t upto) {
    if (upto > RRO_MAX_STATES) { fail(b, "too many states (> %u)", RRO_MAX_STATES); return 0; }
    return 1;
}

/* ---- atoms: NFA.cc:42-71 ---- */
static sub_t atom_empty(build_t *b, uint32_t cur) {            /* NFA.cc:42-49 */
    sub_t s; memset(&s, 0, sizeof s);
    s.initial = cur; s.size = 1;
    if (!check_states(b, (uint64_t)cur + 1)) return s;
    (void)row(b->n, cur, 0, 1);
    set_add(&s.finals, cur);
    return s;
}
static sub_t atom_set(build_t *b, uint32_t cur, const uint8_t member[128]) {   /* NFA.cc:50-71 */
    sub_t s; memset(&s, 0, sizeof s);
    s.initial = cur; s.size = 2;
    if (!check_states(b, (uint64_t)cur + 2)) return s;
    for (unsigned c = 0; c < 128; c++) if (member[c]) {
        set_add(row(b->n, cur, c, 1), cur + 1);       /* NFA.cc:52 / 63 */
        set_add(row(b->n, cur + 1, c, 0), cur);       /* NFA.cc:53 / 64 */
    }
    (void)row(b->n, cur + 1, 0, 1);
    set_add(&s.finals, cur + 1);
    return s;
}
static sub_t atom_char(build_t *b, uint32_t cur, unsigned c) {
    uint8_t m[128]; memset(m, 0, sizeof m); m[c] = 1;
    return atom_set(b, cur, m);
}

/* ---- NFA.cc:108-121  skip<fwd>(n,k) ---- */
static void skip(rro_nfa *nf, int fwd, uint32_t n, uint32_t k) {
    uint32_t to_skip = fwd ? k : n, fixed = fwd ? n : k;
    for (unsigned c = 0; c < 0x80; c++) {
        const set_t *src = crow(nf, to_skip, c, fwd);
        if (!src->n) continue;
        set_or(row(nf, fixed, c, fwd), src);                          /* NFA.cc:114 */
        src = crow(nf, to_skip, c, fwd);                              /* row() may have reallocated st */
        for (uint32_t j = 0; j < src->n; j++)                         /* NFA.cc:115-119 */
            set_add(row(nf, src->v[j], c, !fwd), fixed);
    }
}
/* ---- NFA.cc:122-137  concatenation ---- */
static void nfa_concat(build_t *b, sub_t *a, const sub_t *o) {
    a->size += o->size;                                               /* regex.h:94 */
    for (uint32_t i = 0; i < a->finals.n; i++) skip(b->n, 0, a->finals.v[i], o->initial);
    int a_null = set_contains(&a->finals, a->initial);
    if (a_null) skip(b->n, 1, a->initial, o->initial);
    if (a_null && set_contains(&o->finals, o->initial)) {
        set_assign(&a->finals, &o->finals);
        set_add(&a->finals, a->initial);
    } else set_assign(&a->finals, &o->finals);
}
/* ---- NFA.cc:138-149  union ---- */
static void nfa_union(build_t *b, sub_t *a, const sub_t *o) {
    a->size += o->size;                                               /* regex.h:93 */
    set_or(&a->finals, &o->finals);
    skip(b->n, 1, a->initial, o->initial);
    if (set_contains(&o->finals, o->initial)) set_add(&a->finals, a->initial);
}
/* ---- NFA.cc:150-157  Kleene ---- */
static void nfa_star(build_t *b, sub_t *a) {
    for (uint32_t i = 0; i < a->finals.n; i++) skip(b->n, 0, a->finals.v[i], a->initial);
    set_add(&a->finals, a->initial);
}
/* ---- NFA.cc:177-185  shifted copy ---- */
static sub_t nfa_shifted(build_t *b, const sub_t *in, uint32_t rot) {
    sub_t r; memset(&r, 0, sizeof r);
    r.initial = in->initial + rot; r.size = in->size;
    if (!check_states(b, (uint64_t)r.initial + r.size)) return r;
    for (uint32_t i = in->initial; i < in->initial + in->size; i++) {
        (void)row(b->n, i + rot, 0, 1);
        for (unsigned c = 0; c < 128; c++) for (int fwd = 0; fwd < 2; fwd++) {
            const set_t *src = crow(b->n, i, c, fwd);
            if (src->n) { set_t *dst = row(b->n, i + rot, c, fwd); src = crow(b->n, i, c, fwd); set_assign_shifted(dst, src, rot); }
        }
    }
    set_assign_shifted(&r.finals, &in->finals, rot);
    return r;
}

/* ---- Parser.cpp:49-79  clear_stack ---- */
#define NEED_NFA(b, k) do { if ((b)->nn < (k)) { fail(b, "invalid expression (operand stack underflow)"); return; } } while (0)
#define NEED_OP(b, k)  do { if ((b)->no < (k)) { fail(b, "invalid expression (operator stack underflow)"); return; } } while (0)
static void clear_stack(build_t *b) {
    NEED_NFA(b, 1);
    sub_t cur = b->nfas[b->nn - 1];                 /* moved-from slot stays on the stack, as in the reference */
    memset(&b->nfas[b->nn - 1].finals, 0, sizeof(set_t));
    if (b->no) {
        b->no--;
        while (b->no > 1 && b->ops[b->no - 1] != OP_BRACKETS) {
            if (b->ops[b->no - 1] == OP_CONCATENATION) {
                b->nn--; b->no--;
                if (b->nn < 1) { fail(b, "invalid expression (operand stack underflow)"); set_free(&cur.finals); return; }
                nfa_concat(b, &b->nfas[b->nn - 1], &cur);
                set_free(&cur.finals);
                cur = b->nfas[b->nn - 1];
                memset(&b->nfas[b->nn - 1].finals, 0, sizeof(set_t));
            } else { /* OP_OR */
                sub_t inter = cur;
                b->no--;
                if (b->no < 1) { fail(b, "invalid expression (operator stack underflow)"); set_free(&inter.finals); return; }
                b->no--;
                b->nn--;
                if (b->nn < 1) { fail(b, "invalid expression (operand stack underflow)"); set_free(&inter.finals); return; }
                cur = b->nfas[b->nn - 1];
                memset(&b->nfas[b->nn - 1].finals, 0, sizeof(set_t));
                while (b->no > 1 && b->ops[b->no - 1] == OP_CONCATENATION) {
                    b->nn--; b->no--;
                    if (b->nn < 1) { fail(b, "invalid expression (operand stack underflow)"); set_free(&inter.finals); set_free(&cur.finals); return; }
                    nfa_concat(b, &b->nfas[b->nn - 1], &cur);
                    set_free(&cur.finals);
                    cur = b->nfas[b->nn - 1];
                    memset(&b->nfas[b->nn - 1].finals, 0, sizeof(set_t));
                }
                nfa_union(b, &cur, &inter);
                set_free(&inter.finals);
            }
            if (b->failed) { set_free(&cur.finals); return; }
        }
    }
    if (b->no < 1) { fail(b, "invalid expression (operator stack underflow)"); set_free(&cur.finals); return; }
    b->no--;
    push_op(b, OP_CONCATENATION);
    set_free(&b->nfas[b->nn - 1].finals);
    b->nn--;
    push_nfa(b, cur);
}
/* ---- Parser.cpp:80-83  repeat ---- */
static void repeat(build_t *b) {
    NEED_NFA(b, 1);
    sub_t top = b->nfas[b->nn - 1];
    sub_t r = nfa_shifted(b, &top, top.size);
    if (b->failed) { set_free(&r.finals); return; }
    push_nfa(b, r);
    push_op(b, OP_CONCATENATION);
}
static void make_optional(build_t *b) {          /* Parser.cpp:121 / 135 */
    NEED_NFA(b, 1);
    sub_t e = atom_empty(b, next_initial(b));
    if (!b->failed) nfa_union(b, &b->nfas[b->nn - 1], &e);
    set_free(&e.finals);
}

/* ---- Parser.cpp:16-39  bracket_expression.  q indexes the pattern; returns new q (at ']' or last char). ---- */
static size_t bracket_expression(build_t *b, const char *p, size_t ps, size_t s0, uint8_t member[128]) {
    int escaped = 0;
    size_t q = s0 + 1;
    int complement = (p[q] == '^');                  /* Parser.cpp:18: the '^' is NOT skipped */
    memset(member, 0, 128);
    while (q + 1 < ps && !(p[q] == ']' && !escaped)) {
        if ((unsigned char)p[q] >= 0x80) { fail(b, "non-ASCII byte in pattern"); return q; }
        if (!escaped) {
            char next = p[q + 1];
            if (next != ']' && q + 2 < ps) {
                char nextnext = p[q + 2];
                if (next == '-' && nextnext != ']') {
                    if ((unsigned char)nextnext >= 0x7f) { fail(b, "bracket range end out of range"); return q; }
                    for (int c = p[q]; c <= nextnext; c++) member[c] = 1;      /* Parser.cpp:26 */
                    q += 3;
                    continue;
                }
            }
        }
        escaped = !escaped && (p[q] == '\\');
        member[(unsigned char)p[q]] = 1;              /* Parser.cpp:33 */
        q++;
    }
    if (q == ps) { fail(b, "invalid expression!"); return q; }      /* Parser.cpp:35 */
    if (complement) for (int c = 0; c < 128; c++) member[c] = !member[c];   /* BitSet.cc:42-56 */
    return q;
}

/* ---- Parser.cpp:40-159  build_NFA ---- */
static void build(build_t *b, const char *p) {
    size_t ps = strlen(p);
    size_t cp = 0;
    int escaped = 0;
    push_op(b, OP_BRACKETS);
    do {
        if (b->failed) return;
        unsigned char ch = (unsigned char)p[cp];
        if (ch >= 0x80) { fail(b, "non-ASCII byte in pattern"); return; }
        if (!escaped && ch == '\\') { escaped = 1; continue; }
        switch (escaped ? 0 : ch) {
        case '[': {
            uint8_t member[128];
            uint32_t ni = next_initial(b);
            cp = bracket_expression(b, p, ps, cp, member);
            if (b->failed) return;
            push_nfa(b, atom_set(b, ni, member));
            push_op(b, OP_CONCATENATION);
            break;
        }
        case '(': push_op(b, OP_BRACKETS); break;
        case '|': push_op(b, OP_OR); break;
        case ')': clear_stack(b); break;
        case '.': {
            uint8_t member[128]; memset(member, 1, sizeof member);      /* Parser.cpp:106-109 */
            push_nfa(b, atom_set(b, next_initial(b), member));
            push_op(b, OP_CONCATENATION);
            break;
        }
        case '*':
            if (b->nn < 1) { fail(b, "invalid expression (nothing to repeat)"); return; }
            nfa_star(b, &b->nfas[b->nn - 1]);
            break;
        case '+':
            repeat(b);
            if (!b->failed) nfa_star(b, &b->nfas[b->nn - 1]);
            break;
        case '?': make_optional(b); break;
        case '{': {                                                     /* Parser.cpp:123-141 */
            char *cn1;
            cp++;
            long m = strtol(p + cp, &cn1, 10);
            if (m > (long)RRO_MAX_STATES) { fail(b, "too many states"); return; }
            for (long i = 0; i < m - 1 && !b->failed; i++) repeat(b);
            if (b->failed) return;
            if (*cn1 != '}') {
                if (*cn1 == '\0') { fail(b, "invalid expression (unterminated {)"); return; }
                char *cn2;
                long n = strtol(cn1 + 1, &cn2, 10);
                if (n > (long)RRO_MAX_STATES) { fail(b, "too many states"); return; }
                if (!n) {
                    repeat(b);
                    if (!b->failed) nfa_star(b, &b->nfas[b->nn - 1]);
                } else if (n > m) {
                    repeat(b);
                    if (!b->failed) make_optional(b);
                    for (long k = m + 1; k < n && !b->failed; k++) repeat(b);
                }
                cp = (size_t)(cn2 - p);
            } else cp = (size_t)(cn1 - p);
            break;
        }
        case '^':
        case '$':                                                       /* Parser.cpp:142-146 */
            push_nfa(b, atom_char(b, next_initial(b), 0));
            push_op(b, OP_CONCATENATION);
            break;
        default:                                                        /* Parser.cpp:147-150 */
            push_nfa(b, atom_char(b, next_initial(b), ch));
            push_op(b, OP_CONCATENATION);
            break;
        }
        escaped = 0;
    } while (++cp < ps);
    if (b->failed) return;
    clear_stack(b);
    if (b->failed) return;
    if (b->nn != 1) fail(b, "invalid expression");                      /* Parser.cpp:155 */
}

static void build_dense(rro_nfa *n) {
    int W = n->cls;
    n->dense = (uint64_t *)calloc((size_t)128 * n->states_n * W, sizeof(uint64_t));
    for (uint32_t s = 0; s < n->states_n; s++)
        for (unsigned c = 0; c < 128; c++) {
            const set_t *r = crow(n, s, c, 1);
            uint64_t *d = n->dense + ((size_t)c * n->states_n + s) * W;
            for (uint32_t j = 0; j < r->n; j++) d[r->v[j] >> 6] |= 1ULL << (r->v[j] & 63);
        }
    memset(n->dfinal, 0, sizeof n->dfinal);
    for (uint32_t j = 0; j < n->finals.n; j++) n->dfinal[n->finals.v[j] >> 6] |= 1ULL << (n->finals.v[j] & 63);
}

rro_nfa *rro_compile(const char *pattern, char *err, size_t errcap) {
    build_t b; memset(&b, 0, sizeof b);
    b.n = (rro_nfa *)calloc(1, sizeof(rro_nfa));
    build(&b, pattern);
    if (b.failed) {
        if (err && errcap) snprintf(err, errcap, "%s", b.err);
        for (int i = 0; i < b.nn; i++) set_free(&b.nfas[i].finals);
        free(b.nfas); free(b.ops);
        rro_free(b.n);
        return NULL;
    }
    rro_nfa *n = b.n;
    n->initial = b.nfas[0].initial;
    n->size = b.nfas[0].size;
    n->states_n = b.nfas[0].size;                                       /* Parser.cpp:163 */
    n->finals = b.nfas[0].finals;
    free(b.nfas); free(b.ops);
    (void)row(n, n->states_n ? n->states_n - 1 : 0, 0, 1);
    if (n->states_n > 256) n->cls = 0;                                  /* Parser.cpp:165-168 */
    else if (n->states_n > 128) n->cls = 4;
    else if (n->states_n > 64) n->cls = 2;
    else n->cls = 1;
    if (n->cls) build_dense(n);
    else n->mark = (uint8_t *)calloc(n->states_n, 1);
    if (err && errcap) err[0] = 0;
    return n;
}

void rro_free(rro_nfa *n) {
    if (!n) return;
    for (uint32_t s = 0; s < n->st_cap; s++) if (n->st[s]) {
        for (int r = 0; r < 256; r++) free(n->st[s]->r[r].v);
        free(n->st[s]);
    }
    free(n->st); set_free(&n->finals); free(n->dense); free(n->mark); free(n);
}
uint32_t rro_states_n(const rro_nfa *n) { return n->states_n; }
uint32_t rro_initial(const rro_nfa *n) { return n->initial; }
int rro_set_class(const rro_nfa *n) { return n->cls; }
int rro_is_final(const rro_nfa *n, uint32_t s) { return set_contains(&n->finals, s); }
uint32_t rro_row(const rro_nfa *n, uint32_t state, unsigned c, int fwd, uint32_t *out, uint32_t cap) {
    if (c >= 128) return 0;
    const set_t *r = crow(n, state, c, fwd);
    for (uint32_t i = 0; i < r->n && i < cap; i++) out[i] = r->v[i];
    return r->n;
}

/* ------------------------------------------------------------------ set primitives (BitSet.cc) ---- */
void rro_bs_or(int W, uint64_t *a, const uint64_t *b) { for (int i = 0; i < W; i++) a[i] |= b[i]; }      /* BitSet.cc:8-21 */
void rro_bs_and(int W, uint64_t *a, const uint64_t *b) { for (int i = 0; i < W; i++) a[i] &= b[i]; }     /* BitSet.cc:22-35 */
uint32_t rro_bs_cardinality(int W, const uint64_t *a) {                                                  /* BitSet.cc:36-41 */
    uint32_t r = 0; for (int i = 0; i < W; i++) r += (uint32_t)__builtin_popcountll(a[i]); return r;
}
uint32_t rro_bs_and_cardinality(int W, const uint64_t *a, const uint64_t *b) {                           /* BitSet.cc:104-109 */
    uint32_t r = 0; for (int i = 0; i < W; i++) r += (uint32_t)__builtin_popcountll(a[i] & b[i]); return r;
}
void rro_bs_add(int W, uint64_t *a, uint32_t t) { (void)W; a[t >> 6] |= 1ULL << (t & 63); }              /* BitSet.cc:98-103 */
int rro_bs_contains(int W, const uint64_t *a, uint32_t t) { (void)W; return (int)((a[t >> 6] >> (t & 63)) & 1); } /* BitSet.cc:110-115 */
void rro_bs_shl(int W, uint64_t *dst, const uint64_t *a, int32_t rotate) {                               /* BitSet.cc:116-180 */
    uint64_t tmp[4] = {0, 0, 0, 0};
    if (rotate >= 0 && rotate < 64 * W) {
        int wsh = rotate >> 6, bsh = rotate & 63;
        for (int i = W - 1; i >= wsh; i--) {
            uint64_t v = a[i - wsh] << bsh;
            if (bsh && i - wsh - 1 >= 0) v |= a[i - wsh - 1] >> (64 - bsh);
            tmp[i] = v;
        }
    }
    for (int i = 0; i < W; i++) dst[i] = tmp[i];
}
void rro_bs_complement(int W, uint64_t *a) { for (int i = 0; i < W; i++) a[i] = ~a[i]; }                  /* BitSet.cc:42-56 */
uint32_t rro_bs_iterate(int W, const uint64_t *a, int32_t *out, uint32_t cap) {                           /* BitSet.cc:57-97 */
    uint32_t k = 0;
    for (int i = 0; i < W; i++) {
        uint64_t w = a[i];
        while (w) {                                   /* BitSet.cc:78-80: ctz + clear lowest */
            if (k < cap) out[k] = i * 64 + __builtin_ctzll(w);
            k++;
            w &= w - 1;
        }
    }
    return k;
}

/* ------------------------------------------------------------------ hot path: NFA.cc:72-107 ---- */
static int accepts_w1(const rro_nfa *n, const uint8_t *s, size_t len) {
    const uint64_t *T = n->dense; const uint32_t N = n->states_n;
    uint64_t cur = 1ULL << n->initial;                          /* regex.h:135-143 */
    for (size_t i = 0; i < len; i++) {                          /* regex.h:157 */
        unsigned c = s[i];
        if (c == 0 || c >= 0x80) return 0;
        const uint64_t *col = T + (size_t)c * N;
        uint64_t nw = 0;
        for (uint64_t w = cur; w; w &= w - 1) nw |= col[__builtin_ctzll(w)];   /* NFA.cc:88-97 */
        cur = nw;                                               /* NFA.cc:99 */
    }
    return (cur & n->dfinal[0]) != 0;                           /* NFA.cc:103-107 */
}
static int accepts_wn(const rro_nfa *n, const uint8_t *s, size_t len) {
    const int W = n->cls; const uint64_t *T = n->dense; const uint32_t N = n->states_n;
    uint64_t cur[4] = {0, 0, 0, 0}, nw[4];
    cur[n->initial >> 6] = 1ULL << (n->initial & 63);
    for (size_t i = 0; i < len; i++) {
        unsigned c = s[i];
        if (c == 0 || c >= 0x80) return 0;
        const uint64_t *col = T + (size_t)c * N * W;
        nw[0] = nw[1] = nw[2] = nw[3] = 0;
        for (int k = 0; k < W; k++)
            for (uint64_t w = cur[k]; w; w &= w - 1) {
                const uint64_t *r = col + (size_t)(k * 64 + __builtin_ctzll(w)) * W;
                for (int j = 0; j < W; j++) nw[j] |= r[j];      /* BitSet.cc:8-21 */
            }
        for (int j = 0; j < W; j++) cur[j] = nw[j];
    }
    return rro_bs_and_cardinality(W, cur, n->dfinal) > 0;
}
/* The same loops with the reference's own vector ORs (BitSet.cc:8-21: W = 2 `_mm_or_si128`, W = 4 `_mm256_or_si256`, unaligned
 * loads) - what SURVEY.md 8(d)(ii) asks the CPU baseline to be.  Same results bit for bit (tests/test_oracle_golden.py runs the
 * golden corpora through both); taken when the CPU has AVX2 unless rro_set_simd(0) turned it off. */
#if defined(__x86_64__)
#include <immintrin.h>
__attribute__((target("avx2"))) static int accepts_w4_avx2(const rro_nfa *n, const uint8_t *s, size_t len) {
    const uint64_t *T = n->dense; const uint32_t N = n->states_n;
    uint64_t cur[4] = {0, 0, 0, 0};
    cur[n->initial >> 6] = 1ULL << (n->initial & 63);
    for (size_t i = 0; i < len; i++) {
        unsigned c = s[i];
        if (c == 0 || c >= 0x80) return 0;
        const uint64_t *col = T + (size_t)c * N * 4;
        __m256i nw = _mm256_setzero_si256();
        for (int k = 0; k < 4; k++)
            for (uint64_t w = cur[k]; w; w &= w - 1)
                nw = _mm256_or_si256(nw, _mm256_loadu_si256((const __m256i *)(col + (size_t)(k * 64 + __builtin_ctzll(w)) * 4)));
        _mm256_storeu_si256((__m256i *)cur, nw);
    }
    return rro_bs_and_cardinality(4, cur, n->dfinal) > 0;
}
__attribute__((target("sse2"))) static int accepts_w2_sse(const rro_nfa *n, const uint8_t *s, size_t len) {
    const uint64_t *T = n->dense; const uint32_t N = n->states_n;
    uint64_t cur[2] = {0, 0};
    cur[n->initial >> 6] = 1ULL << (n->initial & 63);
    for (size_t i = 0; i < len; i++) {
        unsigned c = s[i];
        if (c == 0 || c >= 0x80) return 0;
        const uint64_t *col = T + (size_t)c * N * 2;
        __m128i nw = _mm_setzero_si128();
        for (int k = 0; k < 2; k++)
            for (uint64_t w = cur[k]; w; w &= w - 1)
                nw = _mm_or_si128(nw, _mm_loadu_si128((const __m128i *)(col + (size_t)(k * 64 + __builtin_ctzll(w)) * 2)));
        _mm_storeu_si128((__m128i *)cur, nw);
    }
    return rro_bs_and_cardinality(2, cur, n->dfinal) > 0;
}
#endif
static int g_simd = -1;                     /* -1: not asked yet, 0: scalar loops, 1: the vector ORs */
void rro_set_simd(int on) { g_simd = on ? 1 : 0; }
int rro_simd(void) {
    if (g_simd < 0) {
#if defined(__x86_64__)
        g_simd = __builtin_cpu_supports("avx2") ? 1 : 0;
#else
        g_simd = 0;
#endif
    }
    return g_simd;
}
/* Sparse class: NFA.cc:77-85 (toUint32Array -> row pointers -> fastunion), full-width state index. */
static int accepts_sparse(const rro_nfa *n, const uint8_t *s, size_t len) {
    set_t cur = {0, 0, 0}, nxt = {0, 0, 0};
    set_add(&cur, n->initial);
    int ok = 1;
    for (size_t i = 0; i < len && ok; i++) {
        unsigned c = s[i];
        if (c == 0 || c >= 0x80) { ok = 0; break; }
        nxt.n = 0;
        uint32_t lo = UINT32_MAX, hi = 0;
        for (uint32_t k = 0; k < cur.n; k++) {
            const set_t *r = crow(n, cur.v[k], c, 1);
            for (uint32_t j = 0; j < r->n; j++) {
                uint32_t t = r->v[j];
                if (!n->mark[t]) { n->mark[t] = 1; if (t < lo) lo = t; if (t > hi) hi = t; }
            }
        }
        if (lo != UINT32_MAX)
            for (uint32_t t = lo; t <= hi; t++) if (n->mark[t]) { n->mark[t] = 0; set_reserve(&nxt, nxt.n + 1); nxt.v[nxt.n++] = t; }
        set_t tmp = cur; cur = nxt; nxt = tmp;
    }
    int acc = 0;
    if (ok) for (uint32_t k = 0; k < cur.n && !acc; k++) acc = set_contains(&n->finals, cur.v[k]);
    set_free(&cur); set_free(&nxt);
    return acc;
}

int rro_accepts(const rro_nfa *n, const uint8_t *s, size_t len) {
    if (n->cls == 1) return accepts_w1(n, s, len);
#if defined(__x86_64__)
    if (n->cls == 4 && rro_simd()) return accepts_w4_avx2(n, s, len);
    if (n->cls == 2 && rro_simd()) return accepts_w2_sse(n, s, len);
#endif
    if (n->cls) return accepts_wn(n, s, len);
    return accepts_sparse(n, s, len);
}

size_t rro_match_lines(const rro_nfa *n, const uint8_t *bytes, size_t nbytes, uint8_t *accept, size_t cap) {
    size_t line = 0, start = 0;
    for (size_t i = 0; i <= nbytes; i++) {
        if (i == nbytes) { if (start == nbytes) break; }
        else if (bytes[i] != '\n') continue;
        if (line < cap) accept[line] = (uint8_t)rro_accepts(n, bytes + start, i - start);
        line++;
        start = i + 1;
    }
    return line;
}

/* Search restated on top of whole-string acceptance (the reference has no search): the match of a
 * line is the substring [s, e) the reference ACCEPTS with the smallest e, and among those the smallest s.  Brute
 * force over (e, s): test infrastructure for short lines only. */
size_t rro_search_lines(const rro_nfa *n, const uint8_t *bytes, size_t nbytes, int32_t *start, int32_t *end, size_t cap) {
    size_t line = 0, ls = 0;
    for (size_t i = 0; i <= nbytes; i++) {
        if (i == nbytes) { if (ls == nbytes) break; }
        else if (bytes[i] != '\n') continue;
        if (line < cap) {
            const size_t len = i - ls;
            int32_t bs = -1, be = -1;
            for (size_t e = 0; e <= len && be < 0; e++)
                for (size_t s = 0; s <= e; s++)
                    if (rro_accepts(n, bytes + ls + s, e - s)) { bs = (int32_t)s; be = (int32_t)e; break; }
            start[line] = bs; end[line] = be;
        }
        line++;
        ls = i + 1;
    }
    return line;
}

/* All lazy matches of a line, left to right: the k-th match is the search above applied to line[p..) where p is the end
 * of the previous match (one byte further after an empty match).  Writes up to cap (start, end) pairs, flattened over
 * the lines in order; count[i] = matches of line i.  Returns the total number of matches. */
size_t rro_search_all(const rro_nfa *n, const uint8_t *bytes, size_t nbytes, uint32_t *count, size_t nlines_cap,
                      int32_t *start, int32_t *end, size_t cap) {
    size_t line = 0, ls = 0, total = 0;
    for (size_t i = 0; i <= nbytes; i++) {
        if (i == nbytes) { if (ls == nbytes) break; }
        else if (bytes[i] != '\n') continue;
        const size_t len = i - ls;
        uint32_t k = 0;
        size_t p = 0;
        while (p <= len) {
            int64_t bs = -1, be = -1;
            for (size_t e = p; e <= len && be < 0; e++)
                for (size_t s = p; s <= e; s++)
                    if (rro_accepts(n, bytes + ls + s, e - s)) { bs = (int64_t)s; be = (int64_t)e; break; }
            if (be < 0) break;
            if (total < cap) { start[total] = (int32_t)bs; end[total] = (int32_t)be; }
            total++; k++;
            p = be > bs ? (size_t)be : (size_t)be + 1;
        }
        if (line < nlines_cap) count[line] = k;
        line++;
        ls = i + 1;
    }
    return total;
}
