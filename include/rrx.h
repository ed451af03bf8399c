/*
 * rrx.h — C ABI of the MI355X-native RoaringRegex hot path (librrx.so).
 *
 * The reference has no FFI; its only interface is the C++ iterator facade of src/inc/regex.h.  This ABI is
 * what a binding for that facade binds for the hot path: each entry point names the reference interface it
 * replaces (paths relative to the reference's src/).  include/rregex.hpp rebuilds the reference's C++ names
 * (Regex::RRegex, get_acceptance_iter, IteratorWrapper, Match) on top of it; INTEGRATION.md shows the
 * reference-side stub.
 *
 * Conventions: plain pointers and sizes only; no exception crosses the boundary — every function returns
 * RRX_OK or an error code and rrx_last_error() describes the last failure on the calling thread
 * (the reference throws std::runtime_error, Parser.cpp:36,155).  "d_" parameters are DEVICE pointers
 * (hipMalloc'ed on `device`); `stream` is a hipStream_t (NULL = default stream); launches are asynchronous
 * on that stream.  There is no CPU matcher behind this ABI: matching requires a gfx950 device.
 */
#ifndef RRX_H
#define RRX_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct rrx_regex rrx_regex;    /* a compiled pattern: replaces Regex::RRegex, regex.h:212-228        */
typedef struct rrx_corpus rrx_corpus;  /* a device-resident batch of '\n'-delimited strings + its line index */

enum { RRX_OK = 0, RRX_ERR_PATTERN = 1, RRX_ERR_ARG = 2, RRX_ERR_HIP = 3, RRX_ERR_UNSUPPORTED = 4 };

/* engine selection for rrx_compile_ex */
enum { RRX_ENGINE_AUTO = 0, RRX_ENGINE_NFA = 1, RRX_ENGINE_DFA = 2, RRX_ENGINE_DFA_GLOBAL = 3 /* table kept in HBM/L2 */,
       RRX_ENGINE_NFA_WAVE = 4 /* state set spread over 8, 16 or 32 lanes of a wave, 2-8 words per lane: up to 8192 positions */,
       RRX_ENGINE_DFA2 = 5 /* table with one dependent lookup per two bytes (AUTO prefers it when it fits) */,
       RRX_ENGINE_NFA_BLOCK = 8 /* wave-resident NFA: one wave holds one state set of up to 65536 positions (32 words per lane),
                                   exception edges as sparse lists (the reference's Roaring class, Parser.cpp:165, at any
                                   size); dense: a byte costs the whole set */,
       RRX_ENGINE_NFA_SPARSE = 10 /* the same with the set kept as a LIST OF ITS NON-EMPTY BLOCKS of 2048 positions (a bit mask):
                                   a byte costs the live blocks only (README.md:18-21: sets are sparse) - three times a dense
                                   word per live block, so for texts on which few blocks are live; never chosen by AUTO */ };

/* ---- compile: RRegex::RRegex(const char*), Parser.cpp:161-170 (host only, no device needed) ---------- */
int rrx_compile(const char *pattern, rrx_regex **out);
int rrx_compile_ex(const char *pattern, int engine, rrx_regex **out);
void rrx_free(rrx_regex *re);
const char *rrx_last_error(void);

/* ---- what the reference would have built (for parity checks on the construction) -------------------- */
uint32_t rrx_num_states(const rrx_regex *re);      /* states_n, Parser.cpp:163                               */
int      rrx_set_class(const rrx_regex *re);       /* 1,2,4 = BitSet<W>, 0 = Roaring: Parser.cpp:165-168     */
uint32_t rrx_ref_initial(const rrx_regex *re);     /* initial_state, regex.h:81                              */
int      rrx_ref_is_final(const rrx_regex *re, uint32_t state);                 /* final_states, regex.h:177 */
/* forward row T[idx(state,c,true)] (NFA.cc:9-12) as ascending state numbers; returns its cardinality */
uint32_t rrx_ref_row(const rrx_regex *re, uint32_t state, unsigned c, uint32_t *out, uint32_t cap);

/* ---- what the device runs ---------------------------------------------------------------------------- */
int         rrx_engine(const rrx_regex *re);       /* RRX_ENGINE_NFA or RRX_ENGINE_DFA                       */
const char *rrx_engine_name(const rrx_regex *re);
uint32_t    rrx_useful_states(const rrx_regex *re);
uint32_t    rrx_byte_classes(const rrx_regex *re);
/* 1 if the stride-2 table of this regex is laid out in a PROFILED order, 2 while the search for one is running, else 0.  At a
 * regex' first rrx_match_corpus against a corpus of 64 MiB or more a background thread (tens of ms of host work, once per
 * regex; tables small enough to be replicated are left alone) orders the rows and columns of the table by a 64 KiB sample of
 * that corpus' text so that fewer of a half-wave's lookups fall into one LDS bank, uploads the table again in that order
 * (its own hipMalloc + copies, no lock held) and swaps the descriptors in under the regex' mutex, which launches take for the
 * length of a copy: launches before the swap run on the table as numbered.  rrx_set_option can forbid it.  *before / *after: the mean
 * number of distinct entries in the fullest bank per half-wave on that sample (either may be NULL).  Results never depend
 * on the order.                                                                                                        */
int         rrx_table_order(const rrx_regex *re, double *conflict_before, double *conflict_after);
/* The same ordering from a text sample the CALLER provides (host memory), before the regex' first match: `lanes` (a multiple of
 * 32) pieces of `bytes_per_lane` bytes each, lane-major - 32 consecutive pieces are stepped in lockstep the way a half-wave
 * steps 32 neighbouring stripes, so take them from 32 places of the text a stripe (some KiB) apart.  Host only.              */
int         rrx_order_table(rrx_regex *re, const void *sample, uint32_t lanes, uint32_t bytes_per_lane);
/* Per-regex options.  RRX_OPT_BACKGROUND_ORDER (default 1): 0 forbids the library to start a thread of its own and to
 * allocate device memory behind the caller's back for the profiled table order above - the table then stays as numbered
 * unless the caller orders it himself with rrx_order_table (which runs in the calling thread).  Set it before the regex'
 * first rrx_match_corpus; a search that is already running is not stopped.  RRX_ERR_ARG for an unknown option.          */
/* RRX_OPT_FLUSH_SLOTS (default 0 = automatic): the stride-2 batch kernel lets all lanes flush their result bits together every
 * `value` slots (16 bytes of a lane's text; 1, 2, 4, 8, 16 or 32); automatic: from the corpus' mean line length, about eight
 * line ends per period.  A tuning knob: results never depend on it.                                                       */
/* RRX_OPT_SAMPLED_TABLE (default 1): 0 keeps an automaton whose subset construction explodes on the NFA lane engine for every
 * launch (see rrx_learn_table).                                                                                          */
/* RRX_OPT_UNITS_PER_WORKGROUP (default 0 = off): the stride-2 batch kernel hands its stripes out in units of 64 inside the
 * workgroup, `value` (16 ... 65536) of them per workgroup of 16 waves, a wave taking its next unit from a counter in LDS.
 * Same results; measured no faster than one stripe per lane on any config (profiles/r04_unit_handout_ab.txt).           */
/* RRX_OPT_SEARCH_ANCHORED (default 1): the search kernels' forward table is the product of "any bytes, then the pattern" with
 * the pattern's own table, which tells the matches that start at the line start (no walk back to find the start); 0 builds the
 * forward table alone - fewer rows, every match start walked back to - which is also what a product beyond 65534 rows falls
 * back to.  Same results.  Set it before the regex' first search (RRX_ERR_ARG afterwards).                                  */
/* RRX_OPT_ITEMS_STRIDE2 (default 1): large batches of explicit items with a separator byte each (rrx_match_extents /
 * rrx_match_items, trim 1) are stepped two bytes per lookup by a stride-2 table of their own - every byte value an ordinary
 * symbol, '\n' too, the separator a 129th - where the regex has a stride-2 table; 0 keeps them on the byte-stride items kernel
 * (which serves trim 0 either way).  Same results.                                                                         */
enum { RRX_OPT_BACKGROUND_ORDER = 1, RRX_OPT_UNITS_PER_WORKGROUP = 2, RRX_OPT_SAMPLED_TABLE = 3, RRX_OPT_FLUSH_SLOTS = 4, RRX_OPT_SEARCH_ANCHORED = 5,
       RRX_OPT_ITEMS_STRIDE2 = 6 };
int         rrx_set_option(rrx_regex *re, int option, int64_t value);
/* The SAMPLED TABLE (an automaton that does not determinise - AUTO leaves it on the NFA lane engine - over text whose live sets
 * are few, README.md:18-21): the state sets a text sample reaches are interned into a table, every transition the sample and a
 * bounded closure leave open leads to an ESCAPE state.  rrx_match_corpus then runs the stride-2 table kernel (two result bits
 * per line: accepted, escaped) and lets the NFA engine decide the escaped lines: the result is exact for ANY text, text that
 * resembles the sample runs at table speed.  Built once per regex: by the first rrx_match_corpus against a corpus of 64 MiB or
 * more, from the corpus' own sample, in a background thread (RRX_OPT_BACKGROUND_ORDER 0: in the caller's) - or here, from `text`
 * (host memory, whole lines; the first line fragment is skipped), in the caller's thread.  A table that more than 2 % of its OWN
 * sample's lines leave is not installed (every escaped line is read twice: text whose live sets are not few - random a/b lines
 * under (a|b)*a(a|b){40} - stays on the NFA engine at its full rate).  RRX_ERR_UNSUPPORTED: the regex is not on the NFA lane
 * engine by AUTO's choice, no table fits, or the sample escapes from it.  rrx_sampled_table: 1 = in use, 2 = being built, 0 = none,
 * 3 = RETIRED: a launch saw more than 5 % of a corpus' lines escape (every launch leaves its count in pinned host memory, the
 * next one looks at it without waiting) - the regex is back on the NFA engine, and the first launch against a corpus of 64 MiB
 * or more after that LEARNS THE TABLE AGAIN from that corpus' sample (as the first build: beside the caller, or in the launch
 * itself with RRX_OPT_BACKGROUND_ORDER 0; at most three times per regex; a new table is under the same two rules; state 1 again
 * when it is in); *table_states, *open_transitions (entries that lead to ESCAPE) describe the table.  Host only.             */
int         rrx_learn_table(rrx_regex *re, const void *text, size_t nbytes);
int         rrx_sampled_table(const rrx_regex *re, uint32_t *table_states, uint32_t *open_transitions);
/* *lines = the number of lines the NFA engine had to decide in the regex' LAST sampled-table launch on `device` (synchronous:
 * waits for the device).  A table whose escapes stay above a few per cent of the lines was learnt from the wrong text.      */
int         rrx_sampled_escapes(const rrx_regex *re, int device, uint64_t *lines);
uint32_t    rrx_words_per_set(const rrx_regex *re); /* 32-bit words of the register-resident state set (NFA) */
int         rrx_accepts_empty(const rrx_regex *re); /* Processor::operator*() on the initial set, NFA.cc:103-107 */
/* Serialised device program as 32-bit words (layout: DESIGN.md "Device programs"); returns the word count
 * (call with cap = 0 to size the buffer).  kind = RRX_ENGINE_NFA / _DFA / _NFA_WAVE / _DFA2 / _NFA_BLOCK (NFA layout with
 * the exception rows replaced by the CSR arrays xoff[nbits+1], xtgt[]), or one of the two search tables below (DFA
 * layout); 0 if that form was not built.                                                                      */
#define RRX_PROGRAM_SEARCH_FWD 6   /* "any bytes, then the pattern": accepting where a match ends              */
#define RRX_PROGRAM_SEARCH_REV 7   /* the pattern right to left: accepting where a match starts                */
#define RRX_PROGRAM_DFA2_ORDER 11  /* [nstates, ncols, row_slot[nstates], col_slot[ncols]]: the stride-2 table's profiled order
                                      (0 words while the table is laid out as numbered)                                  */
#define RRX_PROGRAM_SEARCH_LINE 9  /* the forward table as the stripe-wise search kernel runs it: [nrows, ncols, start row,
                                      SKIP row, column of byte[256], entry[nrows][ncols]], entry = next row | '\n' << 16 |
                                      hit << 17 | match-starts-at-the-line-start << 18 (0 words: form not available)   */
#define RRX_PROGRAM_SEARCH_LINE2 14 /* its stride-2 form, what the kernel steps (two bytes per dependent lookup): [nrows, ncols, start row,
                                      SKIP row, layout (1: LDS, 2: HBM/L2, 0: the stripe-wise kernel is not used), column of the byte
                                      pair[128][128], first[nrows][ncols], all[nrows][ncols]], entry = next row | events << 24, events =
                                      flags of the first byte << 2 | of the second; flags 1 '\n', 2 hit, 3 hit that starts at the restart
                                      point; first: a hit leads to SKIP, all: back to the start row                         */
#define RRX_PROGRAM_SAMPLED_DFA 12  /* the sampled table (rrx_learn_table): the DFA layout, then escaped[nstates] (1: the ESCAPE state) */
#define RRX_PROGRAM_SAMPLED_DFA2 13 /* its stride-2 form as the kernel runs it: the DFA2 layout, byte 2 of an entry = RESULT BITS shifted
                                       in (two per line end: accepted, escaped), byte 3 = those bits                          */
#define RRX_PROGRAM_DFA2_ITEMS 15 /* the stride-2 table of explicit items with a separator each (rrx_match_extents / rrx_match_items, trim 1):
                                    [nstates, ncols, start, accepts_empty, 129, column of the code pair[129][129], next2[nstates][ncols]] - codes
                                    0 ... 127 the byte values ('\n' an ordinary byte), 128 END OF ITEM; entries as in the DFA2 layout.  0 words
                                    where the regex has no stride-2 table or this form does not fit beside the pair table              */
size_t rrx_program_words(const rrx_regex *re, int kind, uint32_t *out, size_t cap);

/* ---- batch of strings: the replacement for calling get_acceptance_iter(line)++ per string ------------ *
 * regex.h:225-227 + 156-162, for every '\n'-delimited string of a device-resident buffer.  A final fragment
 * without '\n' is a string too.  Bytes 0x00 and >= 0x80 reject their string (the reference cannot express
 * the former and has undefined behaviour on the latter, NFA.cc:10,97).                                       */
int    rrx_corpus_create(int device, const void *d_bytes, size_t nbytes, void *stream, rrx_corpus **out);
/* same with an explicit stripe (bytes per GPU lane: a power of two in [512, 16384]; 0 = chosen from nbytes and the mean line length:
 * 512 bytes up to 128 MiB - a lane steps its stripe as one chain of dependent lookups, short stripes are what makes a small corpus
 * fast -, 2 KiB at 1 GiB, 4 KiB at 8 GiB, longer for long lines) */
int    rrx_corpus_create_ex(int device, const void *d_bytes, size_t nbytes, uint32_t stripe_bytes, void *stream,
                            rrx_corpus **out);
uint32_t rrx_corpus_stripe_bytes(const rrx_corpus *c);
size_t rrx_corpus_num_lines(const rrx_corpus *c);
size_t rrx_corpus_num_bytes(const rrx_corpus *c);
void   rrx_corpus_free(rrx_corpus *c);
size_t rrx_corpus_bitmap_words(const rrx_corpus *c);   /* 32-bit words of the accept bitmap: ceil(lines / 32) */
/* THE HOT PATH.  Writes the accept BITMAP: bit (i & 31) of d_accept_bits[i >> 5] = 1 iff string i is accepted
 * (i.e. *it has a value, regex.h:160-162; its Match is then [start of string i, its terminator)).
 * d_accept_bits holds rrx_corpus_bitmap_words() words; it is zeroed and filled on `stream`.                  */
int rrx_match_corpus(const rrx_regex *re, const rrx_corpus *c, uint32_t *d_accept_bits, void *stream);
/* The same for a device-resident buffer that has no rrx_corpus yet, in ONE call: with the lane engines (tables, NFA) the
 * text is read once (the newline index is a by-product of the match: per-stripe counts, a scan, a compaction of the
 * lanes' verdict streams); the cooperative engines build the index first.  d_accept_bits holds cap_words words (zeroed and
 * filled on `stream`); *nlines = number of strings; RRX_ERR_ARG if the bitmap is too small.  Synchronous (the line count
 * comes back through pinned memory: one wait, no copies).  The regex handle keeps the scratch of its largest call until
 * rrx_free: 12 bytes per stripe + the lanes' verdict streams, worst case 1 bit per byte of text (one line per byte).   */
int rrx_match_device(const rrx_regex *re, int device, const void *d_bytes, size_t nbytes, uint32_t *d_accept_bits,
                     size_t cap_words, size_t *nlines, void *stream);
/* One byte per string (0/1) from the bitmap; d_accept holds nlines bytes, 16-byte aligned.                  */
int rrx_bitmap_to_bytes(int device, const uint32_t *d_accept_bits, size_t nlines, uint8_t *d_accept, void *stream);

/* Search (the reference's README promises match iterators, its code has acceptance only: SURVEY.md 8(f).1).  For
 * string i of the corpus: the substring [d_start[i], d_end[i]) (offsets relative to the start of the string) that the
 * pattern accepts as a whole string (regex.h:156-162) with the smallest end, and among those the smallest start;
 * 0xFFFFFFFF in both when no substring is accepted.  Bytes the pattern cannot match (including NUL and >= 0x80) are
 * ordinary text here.  One kernel (kernels_search.hip) for every pattern: its forward table in LDS when it fits, in HBM/L2
 * otherwise; RRX_ERR_UNSUPPORTED only when the REVERSE table does not fit 64 KiB of LDS, the forward table has more than 65534
 * rows, or its stride-2 form more than 256 MiB.  A pattern that accepts the empty string matches [0, 0) in every string.  */
int rrx_search_corpus(const rrx_regex *re, const rrx_corpus *corpus, uint32_t *d_start, uint32_t *d_end, void *stream);
/* ALL lazy matches of every string, left to right (what the reference's CLI is documented to print, README.md:30): the
 * k-th match of a string is the search above applied to the rest of the string after the previous match (one byte
 * further after an empty match).  Two passes: _count writes d_count[i] = matches of string i; the caller turns the
 * counts into the exclusive prefix d_first[i] (u64); _fill writes the matches of string i to the slots d_first[i] ...   */
int rrx_search_all_count(const rrx_regex *re, const rrx_corpus *corpus, uint32_t *d_count, void *stream);
int rrx_search_all_fill(const rrx_regex *re, const rrx_corpus *corpus, const uint64_t *d_first, uint32_t *d_start,
                        uint32_t *d_end, void *stream);

/* The same in ONE call and one pass over the text: d_first[i] (u64, nlines + 1 entries, d_first[nlines] = *total) =
 * slot of the first match of string i, the matches of string i at d_start/d_end[d_first[i] .. d_first[i + 1]).  The
 * match arrays hold `cap` entries: matches beyond are counted, not written - if *total > cap call again with arrays of
 * *total entries (d_first is complete either way).  Synchronous (returns *total).  Calls on one corpus must not
 * overlap (they share the corpus' scratch).                                                                        */
int rrx_search_all(const rrx_regex *re, const rrx_corpus *corpus, uint64_t *d_first, uint32_t *d_start, uint32_t *d_end,
                   size_t cap, size_t *total, void *stream);

/* explicit extents: item i = d_bytes[d_off[i] .. d_off[i+1] - trim); '\n' is an ordinary character here.  Asynchronous on
 * `stream`.  Large batches on a table engine (>= 65536 items) build an item index in a scratch buffer the regex handle keeps
 * until rrx_free (1 bit per byte + 8 bytes per stripe of the batch's extent).  The host does not know the extent; it takes
 * what is left of the allocation behind d_bytes (hipMemGetAddressRange) as its bound and nothing is read back - as long as
 * that bound is plausible for the batch, at most max(128 bytes per item, 16 MiB).  A batch inside a far larger allocation (a
 * memory pool, a caching allocator's block) and memory whose range the runtime does not report (pool, virtual or managed
 * memory) cost one synchronisation on `stream`: d_off[0] and d_off[nitems] are read back.  Calls with one regex on
 * different streams are ordered on the device by an event, not on the host.                                               */
int rrx_match_extents(const rrx_regex *re, int device, const void *d_bytes, const uint64_t *d_off, size_t nitems,
                      uint32_t trim, uint8_t *d_accept, void *stream);

/* A batch of items indexed ONCE and matched by many patterns (what rrx_corpus is for delimited text): the item-end
 * bitmap and the stripe index of section DESIGN.md 6.5 are built here (synchronous) and kept.  rrx_match_items fills
 * d_accept[i] (one byte per item, 16-byte aligned for the fast path) on `stream`; batches or patterns that do not admit the
 * stripe-wise kernel run as rrx_match_extents.  d_bytes and d_off must outlive the handle.  One match at a time per
 * handle (it owns the result scratch).                                                                              */
typedef struct rrx_items rrx_items;
int rrx_items_create(int device, const void *d_bytes, const uint64_t *d_off, size_t nitems, uint32_t trim, void *stream,
                     rrx_items **out);
size_t rrx_items_count(const rrx_items *items);
int rrx_items_stripe_wise(const rrx_items *items);         /* 1: the batch admits the stripe-wise kernel */
void rrx_items_free(rrx_items *items);
int rrx_match_items(const rrx_regex *re, const rrx_items *items, uint8_t *d_accept, void *stream);

/* ONE device-resident string of any length (regex.h:156-159: operator++ consumes the whole string; '\n' and every
 * other byte are ordinary, a NUL or a byte >= 0x80 rejects).  d_accept[0] = 1 iff accepted.  Strings of 32 KiB and
 * more are split into chunks that are stepped in parallel from every table state (automata with <= 254 table
 * states); synchronous with respect to `stream`.                                                                */
int rrx_match_string(const rrx_regex *re, int device, const void *d_bytes, size_t nbytes, uint8_t *d_accept, void *stream);

/* ---- host-buffer conveniences (PCIe inclusive; synchronous) ------------------------------------------ */
/* bytes/accept are HOST pointers; *nlines receives the number of strings; at most cap results are written */
int rrx_match_host(const rrx_regex *re, int device, const void *bytes, size_t nbytes, uint8_t *accept, size_t cap,
                   size_t *nlines);
/* One NUL-terminated host string: `auto it = r.get_acceptance_iter(text)++; *it` (test/main.cpp:25-27).
 * *accepted = has_value(); *len = strlen(text) = Match.end - Match.start.                                   */
int rrx_match_cstr(const rrx_regex *re, int device, const char *text, int *accepted, size_t *len);

#ifdef __cplusplus
}
#endif
#endif
