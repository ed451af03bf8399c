"""roaringregex_amd — MI355X-native engine for the RoaringRegex hot path.

Python host layer over the C ABI of librrx.so (include/rrx.h).  It mirrors the reference's interface for the
path (src/inc/regex.h:100-126, 212-228): RRegex(pattern), RRegex.get_acceptance_iter(text) ->
IteratorWrapper with advance() (the reference's operator++(int)) and value() (operator*, a Match or None),
plus the batch entry the reference lacks (Corpus / RRegex.match_corpus).  PyTorch is used only as plumbing
for device memory and streams.  There is no CPU matcher in this package: if librrx.so is missing the import
fails loudly, and matching without a gfx950 device raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("RRX_LIB") or os.path.join(_HERE, "librrx.so")   # RRX_LIB: A/B builds of the same ABI

ENGINE_AUTO, ENGINE_NFA, ENGINE_DFA, ENGINE_DFA_GLOBAL, ENGINE_NFA_WAVE, ENGINE_DFA2 = 0, 1, 2, 3, 4, 5
PROGRAM_SEARCH_FWD, PROGRAM_SEARCH_REV = 6, 7          # rrx_program_words kinds of the two search tables
ENGINE_NFA_BLOCK = 8
ENGINE_NFA_SPARSE = 10
PROGRAM_SEARCH_LINE = 9
PROGRAM_SEARCH_LINE2 = 14
PROGRAM_DFA2_ITEMS = 15
PROGRAM_DFA2_ORDER = 11
PROGRAM_SAMPLED_DFA = 12
PROGRAM_SAMPLED_DFA2 = 13
OPT_BACKGROUND_ORDER = 1
OPT_UNITS_PER_WORKGROUP = 2
OPT_SAMPLED_TABLE = 3
OPT_FLUSH_SLOTS = 4
OPT_SEARCH_ANCHORED = 5
OPT_ITEMS_STRIDE2 = 6

# every symbol include/rrx.h declares (tests check the library exports exactly these)
ABI_SYMBOLS = (
    "rrx_compile", "rrx_compile_ex", "rrx_free", "rrx_last_error",
    "rrx_num_states", "rrx_set_class", "rrx_ref_initial", "rrx_ref_is_final", "rrx_ref_row",
    "rrx_engine", "rrx_engine_name", "rrx_useful_states", "rrx_byte_classes", "rrx_table_order", "rrx_order_table", "rrx_set_option", "rrx_learn_table", "rrx_sampled_table", "rrx_sampled_escapes", "rrx_words_per_set", "rrx_accepts_empty",
    "rrx_program_words",
    "rrx_corpus_create", "rrx_corpus_create_ex", "rrx_corpus_stripe_bytes", "rrx_corpus_num_lines", "rrx_corpus_num_bytes", "rrx_corpus_free", "rrx_corpus_bitmap_words",
    "rrx_match_corpus", "rrx_match_device", "rrx_search_corpus", "rrx_search_all_count", "rrx_search_all_fill", "rrx_search_all", "rrx_bitmap_to_bytes",
    "rrx_match_extents", "rrx_items_create", "rrx_items_count", "rrx_items_stripe_wise", "rrx_items_free", "rrx_match_items",
    "rrx_match_string", "rrx_match_host", "rrx_match_cstr",
)


class RRegexError(RuntimeError):
    """The reference throws std::runtime_error (Parser.cpp:36,155); so do we."""


def _load():
    if not os.path.exists(_SO):
        raise ImportError(
            "roaringregex_amd: %s is missing — build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % _SO)
    # librrx.so needs libamdhip64.so.7.  PyTorch-ROCm ships its own copy under the same SONAME and the dynamic
    # loader keeps whichever is loaded first, so import torch first: one HIP runtime per process, the one that
    # owns the tensors we are handed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(_SO)
    vp, u32, sz, i32 = C.c_void_p, C.c_uint32, C.c_size_t, C.c_int
    sig = {
        "rrx_compile": (i32, [C.c_char_p, C.POINTER(vp)]),
        "rrx_compile_ex": (i32, [C.c_char_p, i32, C.POINTER(vp)]),
        "rrx_free": (None, [vp]),
        "rrx_last_error": (C.c_char_p, []),
        "rrx_num_states": (u32, [vp]),
        "rrx_set_class": (i32, [vp]),
        "rrx_ref_initial": (u32, [vp]),
        "rrx_ref_is_final": (i32, [vp, u32]),
        "rrx_ref_row": (u32, [vp, u32, C.c_uint, vp, u32]),
        "rrx_engine": (i32, [vp]),
        "rrx_engine_name": (C.c_char_p, [vp]),
        "rrx_useful_states": (u32, [vp]),
        "rrx_byte_classes": (u32, [vp]),
        "rrx_table_order": (i32, [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "rrx_set_option": (i32, [vp, i32, C.c_int64]),
        "rrx_learn_table": (i32, [vp, vp, C.c_size_t]),
        "rrx_sampled_table": (i32, [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
        "rrx_sampled_escapes": (i32, [vp, i32, C.POINTER(C.c_uint64)]),
        "rrx_order_table": (i32, [vp, vp, u32, u32]),
        "rrx_words_per_set": (u32, [vp]),
        "rrx_accepts_empty": (i32, [vp]),
        "rrx_program_words": (sz, [vp, i32, vp, sz]),
        "rrx_corpus_create": (i32, [i32, vp, sz, vp, C.POINTER(vp)]),
        "rrx_corpus_create_ex": (i32, [i32, vp, sz, u32, vp, C.POINTER(vp)]),
        "rrx_corpus_stripe_bytes": (u32, [vp]),
        "rrx_corpus_num_lines": (sz, [vp]),
        "rrx_corpus_num_bytes": (sz, [vp]),
        "rrx_corpus_free": (None, [vp]),
        "rrx_corpus_bitmap_words": (sz, [vp]),
        "rrx_match_corpus": (i32, [vp, vp, vp, vp]),
        "rrx_match_device": (i32, [vp, i32, vp, sz, vp, sz, C.POINTER(sz), vp]),
        "rrx_bitmap_to_bytes": (i32, [i32, vp, sz, vp, vp]),
        "rrx_match_extents": (i32, [vp, i32, vp, vp, sz, u32, vp, vp]),
        "rrx_items_create": (i32, [i32, vp, vp, sz, u32, vp, C.POINTER(vp)]),
        "rrx_items_count": (sz, [vp]),
        "rrx_items_stripe_wise": (i32, [vp]),
        "rrx_items_free": (None, [vp]),
        "rrx_match_items": (i32, [vp, vp, vp, vp]),
        "rrx_match_string": (i32, [vp, i32, vp, sz, vp, vp]),
        "rrx_search_corpus": (i32, [vp, vp, vp, vp, vp]),
        "rrx_search_all_count": (i32, [vp, vp, vp, vp]),
        "rrx_search_all_fill": (i32, [vp, vp, vp, vp, vp, vp]),
        "rrx_search_all": (i32, [vp, vp, vp, vp, vp, sz, C.POINTER(sz), vp]),
        "rrx_match_host": (i32, [vp, i32, vp, sz, vp, sz, C.POINTER(sz)]),
        "rrx_match_cstr": (i32, [vp, i32, C.c_char_p, C.POINTER(i32), C.POINTER(sz)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    return L


_L = _load()


def _check(rc):
    if rc != 0:
        raise RRegexError(_L.rrx_last_error().decode("latin-1"))


class _on:
    """Run a method body on `device` and, when the caller names a stream, with that stream as torch's CURRENT stream:
    the tensors the body allocates, torch ops such as cumsum/item and the native launches then all live on one stream
    (the native calls are handed the same stream by _stream_ptr)."""

    def __init__(self, device, stream):
        import torch
        self._dev = torch.cuda.device(device)
        self._st = torch.cuda.stream(stream) if stream is not None else None

    def __enter__(self):
        self._dev.__enter__()
        if self._st is not None:
            self._st.__enter__()

    def __exit__(self, *exc):
        if self._st is not None:
            self._st.__exit__(*exc)
        return self._dev.__exit__(*exc)


def _stream_ptr(stream):
    if stream is None:
        import torch
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


class Match:
    """regex.h:100-105: [start, end) into the caller's buffer; here as offsets plus the buffer."""

    def __init__(self, buf, start, end):
        self._buf, self.start, self.end = buf, start, end

    def str(self):
        return self._buf[self.start:self.end]


class IteratorWrapper:
    """regex.h:113-122 / 150-165.  advance() = operator++(int): consume the whole string (idempotent
    afterwards); value() = operator*: Match or None.  Before advance() the state set is {initial}."""

    def __init__(self, regex, text, device):
        self._re, self._text, self._device = regex, text, device
        self._consumed = False
        self._accepted = None

    def advance(self):
        if not self._consumed:
            acc, n = C.c_int(0), C.c_size_t(0)
            _check(_L.rrx_match_cstr(self._re._h, self._device, self._text, C.byref(acc), C.byref(n)))
            self._accepted, self._len = bool(acc.value), n.value
            self._consumed = True
        return self

    def value(self):
        if not self._consumed:
            # NFA.cc:103-107 on the initial set: only patterns whose initial state is final accept here
            return Match(self._text, 0, 0) if self._re.accepts_empty else None
        return Match(self._text, 0, self._len) if self._accepted else None

    def create_copy(self):
        c = IteratorWrapper(self._re, self._text, self._device)
        c.__dict__.update(self.__dict__)
        return c


class Corpus:
    """A device-resident batch of '\\n'-delimited strings plus its per-tile newline index (rrx_corpus)."""

    def __init__(self, data, device=None, stream=None, stripe=0):
        import torch
        if isinstance(data, (bytes, bytearray, memoryview)):
            data = torch.frombuffer(bytearray(data), dtype=torch.uint8) if len(data) else torch.empty(0, dtype=torch.uint8)
        if not data.is_cuda:
            dev = torch.device("cuda", 0 if device is None else device)
            data = data.to(dev)
        assert data.dtype == torch.uint8 and data.is_contiguous()
        self.data = data                       # keeps the bytes alive
        self.device = data.device.index
        self._h = C.c_void_p()
        with _on(self.device, stream):
            _check(_L.rrx_corpus_create_ex(self.device, C.c_void_p(data.data_ptr() if data.numel() else 0), data.numel(),
                                           stripe, _stream_ptr(stream), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _L is not None:      # (_L is None during interpreter shutdown)
            _L.rrx_corpus_free(self._h)
            self._h = None

    @property
    def num_lines(self):
        return _L.rrx_corpus_num_lines(self._h)

    @property
    def num_bytes(self):
        return _L.rrx_corpus_num_bytes(self._h)

    @property
    def stripe(self):
        return _L.rrx_corpus_stripe_bytes(self._h)


class Items:
    """A device-resident batch of explicit items - one byte buffer and an offsets array, item i = data[offsets[i] :
    offsets[i + 1] - trim] - indexed once (rrx_items) and matched by many patterns: RRegex.match_items."""

    def __init__(self, data, offsets, trim=0, stream=None):
        import torch
        assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
        assert offsets.is_cuda and offsets.dtype in (torch.int64, torch.uint64) and offsets.is_contiguous() and offsets.numel() >= 1
        self.data, self.offsets, self.trim = data, offsets, trim       # kept alive: the handle points into them
        self.device = data.device.index
        self._h = C.c_void_p()
        with _on(self.device, stream):
            _check(_L.rrx_items_create(self.device, C.c_void_p(data.data_ptr() if data.numel() else 0), C.c_void_p(offsets.data_ptr()),
                                       offsets.numel() - 1, trim, _stream_ptr(stream), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _L is not None:
            _L.rrx_items_free(self._h)
            self._h = None

    @property
    def num_items(self):
        return _L.rrx_items_count(self._h)

    @property
    def stripe_wise(self):
        """True if the batch admits the stripe-wise kernel (else every match runs lane per item)."""
        return bool(_L.rrx_items_stripe_wise(self._h))


class RRegex:
    """regex.h:212-228.  RRegex(pattern) compiles on the host (Parser.cpp:161-170)."""

    def __init__(self, pattern, engine=ENGINE_AUTO, device=0):
        if isinstance(pattern, str):
            pattern = pattern.encode("latin-1")
        self.pattern = pattern
        self.device = device
        self._h = C.c_void_p()
        _check(_L.rrx_compile_ex(pattern, engine, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None) and _L is not None:
            _L.rrx_free(self._h)
            self._h = None

    # ---- the reference's interface for this path
    def get_acceptance_iter(self, text):
        if isinstance(text, str):
            text = text.encode("latin-1")
        return IteratorWrapper(self, text, self.device)

    # ---- batch entry
    def match_corpus_bits(self, corpus, out=None, stream=None):
        """THE HOT PATH (rrx_match_corpus): the accept bitmap as an int32 tensor, bit (i & 31) of word i >> 5 =
        line i accepted.  Asynchronous on `stream`."""
        import torch
        nw = _L.rrx_corpus_bitmap_words(corpus._h)
        with _on(corpus.device, stream):
            if out is None:
                out = torch.empty(nw, dtype=torch.int32, device=corpus.data.device)
            assert out.is_cuda and out.dtype == torch.int32 and out.numel() >= nw
            _check(_L.rrx_match_corpus(self._h, corpus._h, C.c_void_p(out.data_ptr() if nw else 0), _stream_ptr(stream)))
        return out[:nw]

    def match_device_bits(self, data, cap_lines=None, out=None, stream=None):
        """One-shot (rrx_match_device): a device tensor nobody has indexed -> (accept bitmap as int32 words, number of
        strings).  With the stride-2 table engine the text is read once.  cap_lines bounds the number of strings the
        bitmap can hold (default: one per byte, the most a buffer can hold)."""
        import torch
        assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
        n = data.numel()
        if cap_lines is None:
            cap_lines = n + 1
        cap_words = (cap_lines + 31) // 32
        with _on(data.device.index, stream):
            if out is None:
                out = torch.empty(cap_words, dtype=torch.int32, device=data.device)
            assert out.is_cuda and out.dtype == torch.int32 and out.numel() >= cap_words
            nlines = C.c_size_t(0)
            _check(_L.rrx_match_device(self._h, data.device.index, C.c_void_p(data.data_ptr() if n else 0), n,
                                       C.c_void_p(out.data_ptr()), cap_words, C.byref(nlines), _stream_ptr(stream)))
        return out[:(nlines.value + 31) // 32], nlines.value

    def match_corpus(self, corpus, out=None, stream=None):
        """accept[i] = 1 iff line i of the corpus is accepted (one byte per line: bitmap + expansion)."""
        import torch
        n = corpus.num_lines
        with _on(corpus.device, stream):
            bits = self.match_corpus_bits(corpus, stream=stream)
            if out is None:
                out = torch.empty(n, dtype=torch.uint8, device=corpus.data.device)
            assert out.is_cuda and out.dtype == torch.uint8 and out.numel() >= n
            _check(_L.rrx_bitmap_to_bytes(corpus.device, C.c_void_p(bits.data_ptr() if n else 0), n,
                                          C.c_void_p(out.data_ptr() if n else 0), _stream_ptr(stream)))
        return out[:n]

    def search_corpus(self, corpus, stream=None):
        """Per string the accepted substring [start, end) with the smallest end, then the smallest start, as two int32
        tensors of offsets relative to the start of the string; (-1, -1) where nothing is accepted."""
        import torch
        n = corpus.num_lines
        with _on(corpus.device, stream):
            start = torch.empty(n, dtype=torch.int32, device=corpus.data.device)
            end = torch.empty(n, dtype=torch.int32, device=corpus.data.device)
            _check(_L.rrx_search_corpus(self._h, corpus._h, C.c_void_p(start.data_ptr() if n else 0), C.c_void_p(end.data_ptr() if n else 0),
                                        _stream_ptr(stream)))
        return start, end

    def search_all(self, corpus, stream=None):
        """ALL lazy matches of every string, left to right -> (count[n] int32, first[n] int64, start[total] int32,
        end[total] int32): the matches of string i are start/end[first[i] : first[i] + count[i]], relative to the string."""
        import torch
        n = corpus.num_lines
        dev = corpus.data.device
        with _on(corpus.device, stream):
            count = torch.zeros(n, dtype=torch.int32, device=dev)
            _check(_L.rrx_search_all_count(self._h, corpus._h, C.c_void_p(count.data_ptr() if n else 0), _stream_ptr(stream)))
            inclusive = torch.cumsum(count, dim=0, dtype=torch.int64)
            first = inclusive - count
            total = int(inclusive[-1].item()) if n else 0
            start = torch.empty(total, dtype=torch.int32, device=dev)
            end = torch.empty(total, dtype=torch.int32, device=dev)
            if total:
                _check(_L.rrx_search_all_fill(self._h, corpus._h, C.c_void_p(first.data_ptr()), C.c_void_p(start.data_ptr() if total else 0),
                                              C.c_void_p(end.data_ptr() if total else 0), _stream_ptr(stream)))
        return count, first, start, end

    def search_all_fused(self, corpus, cap=None, stream=None):
        """The same result through the one-call entry (rrx_search_all: one pass over the text) ->
        (first[n + 1] int64 CSR offsets, start[total] int32, end[total] int32).  cap: entries to provide for at first
        (default: two per string); the call is repeated with the exact size if there are more."""
        import torch
        n = corpus.num_lines
        dev = corpus.data.device
        with _on(corpus.device, stream):
            first = torch.empty(n + 1, dtype=torch.int64, device=dev)
            cap = int(cap) if cap is not None else 2 * n + 1024
            while True:
                start = torch.empty(cap, dtype=torch.int32, device=dev)
                end = torch.empty(cap, dtype=torch.int32, device=dev)
                total = C.c_size_t(0)
                _check(_L.rrx_search_all(self._h, corpus._h, C.c_void_p(first.data_ptr()), C.c_void_p(start.data_ptr() if cap else 0),
                                         C.c_void_p(end.data_ptr() if cap else 0), cap, C.byref(total), _stream_ptr(stream)))
                if total.value <= cap:
                    break
                cap = total.value
        return first, start[:total.value], end[:total.value]

    def match_items(self, items, out=None, stream=None):
        """One byte per item of an indexed batch (Items)."""
        import torch
        n = items.num_items
        with _on(items.device, stream):
            if out is None:
                out = torch.empty(n, dtype=torch.uint8, device=items.data.device)
            assert out.is_cuda and out.dtype == torch.uint8 and out.numel() >= n
            _check(_L.rrx_match_items(self._h, items._h, C.c_void_p(out.data_ptr() if n else 0), _stream_ptr(stream)))
        return out[:n]

    def match_extents(self, data, offsets, trim=0, out=None, stream=None):
        """item i = data[offsets[i] : offsets[i+1] - trim]; '\\n' is an ordinary character."""
        import torch
        n = offsets.numel() - 1
        assert data.is_cuda and offsets.is_cuda and offsets.dtype in (torch.int64, torch.uint64)
        with _on(data.device.index, stream):
            if out is None:
                out = torch.empty(n, dtype=torch.uint8, device=data.device)
            _check(_L.rrx_match_extents(self._h, data.device.index, C.c_void_p(data.data_ptr() if data.numel() else 0),
                                        C.c_void_p(offsets.data_ptr()), n, trim, C.c_void_p(out.data_ptr() if n else 0),
                                        _stream_ptr(stream)))
        return out[:n]

    def match_string(self, data, stream=None):
        """ONE device-resident string of any length (regex.h:156-159); '\n' is an ordinary character.  -> bool"""
        import torch
        assert data.is_cuda and data.dtype == torch.uint8
        with _on(data.device.index, stream):
            out = torch.zeros(1, dtype=torch.uint8, device=data.device)
            _check(_L.rrx_match_string(self._h, data.device.index, C.c_void_p(data.data_ptr() if data.numel() else 0), data.numel(),
                                       C.c_void_p(out.data_ptr()), _stream_ptr(stream)))
        return bool(out.item())

    def match_host(self, data):
        """Host bytes in, numpy accept vector out (upload + index + match + download; synchronous)."""
        import numpy as np
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        cap = len(a) + 1                       # a buffer of n bytes holds at most n + 1 strings; untouched pages are free
        out = np.empty(cap, dtype=np.uint8)
        n = C.c_size_t(0)
        _check(_L.rrx_match_host(self._h, self.device, C.c_void_p(a.ctypes.data if len(a) else 0), len(a),
                                 C.c_void_p(out.ctypes.data), cap, C.byref(n)))
        return out[:n.value]

    # ---- introspection (parity checks on the construction; see include/rrx.h)
    @property
    def states_n(self):
        return _L.rrx_num_states(self._h)

    @property
    def set_class(self):
        return _L.rrx_set_class(self._h)

    @property
    def initial(self):
        return _L.rrx_ref_initial(self._h)

    def finals(self):
        return [s for s in range(self.states_n) if _L.rrx_ref_is_final(self._h, s)]

    def row(self, state, c):
        n = self.states_n
        buf = (C.c_uint32 * max(n, 1))()
        k = _L.rrx_ref_row(self._h, state, c, buf, n)
        return list(buf[:k])

    @property
    def engine(self):
        return _L.rrx_engine(self._h)

    @property
    def engine_name(self):
        return _L.rrx_engine_name(self._h).decode()

    @property
    def useful_states(self):
        return _L.rrx_useful_states(self._h)

    def order_table(self, sample, lanes, bytes_per_lane):
        """Order the stride-2 table by a text sample of the caller's (numpy uint8, lanes x bytes_per_lane, lane-major) before
        the first match (rrx_order_table).  Host only."""
        import numpy as np
        a = np.ascontiguousarray(sample, dtype=np.uint8)
        assert a.size >= lanes * bytes_per_lane
        _check(_L.rrx_order_table(self._h, C.c_void_p(a.ctypes.data), lanes, bytes_per_lane))
        return self.table_order

    def set_background_order(self, enabled):
        """rrx_set_option(RRX_OPT_BACKGROUND_ORDER): False forbids the library's own thread and device allocations for the profiled
        table order (the table stays as numbered unless order_table is called)."""
        _check(_L.rrx_set_option(self._h, OPT_BACKGROUND_ORDER, 1 if enabled else 0))

    def learn_table(self, text):
        """rrx_learn_table: build the sampled table (an automaton AUTO leaves on the NFA lane engine) from a text sample - bytes or a
        numpy uint8 array of whole lines.  Returns (table states, open transitions)."""
        import numpy as np
        a = np.frombuffer(text, dtype=np.uint8) if isinstance(text, (bytes, bytearray)) else np.ascontiguousarray(text, dtype=np.uint8)
        _check(_L.rrx_learn_table(self._h, C.c_void_p(a.ctypes.data), a.size))
        return self.sampled_table

    def set_sampled_table(self, enabled):
        _check(_L.rrx_set_option(self._h, OPT_SAMPLED_TABLE, 1 if enabled else 0))

    @property
    def sampled_table(self):
        """None, or (table states, open transitions) of the sampled table in use."""
        n, o = C.c_uint32(0), C.c_uint32(0)
        return (n.value, o.value) if _L.rrx_sampled_table(self._h, C.byref(n), C.byref(o)) == 1 else None

    def sampled_escapes(self):
        """Lines the NFA engine had to decide in the last sampled-table launch on this regex' device (waits for the device)."""
        n = C.c_uint64(0)
        _check(_L.rrx_sampled_escapes(self._h, self.device, C.byref(n)))
        return n.value

    @property
    def sampled_table_pending(self):
        return _L.rrx_sampled_table(self._h, None, None) == 2

    @property
    def sampled_table_retired(self):
        """A corpus escaped from the sampled table (more than 5 % of its lines): the regex is back on the NFA engine."""
        return _L.rrx_sampled_table(self._h, None, None) == 3

    def set_flush_slots(self, slots):
        """rrx_set_option(RRX_OPT_FLUSH_SLOTS): 0 = automatic, or 1 ... 32 slots between two common flushes of the stride-2 kernel."""
        _check(_L.rrx_set_option(self._h, OPT_FLUSH_SLOTS, int(slots)))

    def set_search_anchored(self, enabled):
        """rrx_set_option(RRX_OPT_SEARCH_ANCHORED): False builds the search kernels' forward table without the product that tells
        the matches starting at the line start (fewer rows; every match start is walked back to).  Before the first search."""
        _check(_L.rrx_set_option(self._h, OPT_SEARCH_ANCHORED, 1 if enabled else 0))

    def set_items_stride2(self, enabled):
        """rrx_set_option(RRX_OPT_ITEMS_STRIDE2): False keeps large batches of explicit items with separators (trim 1) on the
        byte-stride items kernel instead of the stride-2 table of their own."""
        _check(_L.rrx_set_option(self._h, OPT_ITEMS_STRIDE2, 1 if enabled else 0))

    def set_units_per_workgroup(self, units):
        """rrx_set_option(RRX_OPT_UNITS_PER_WORKGROUP): the stride-2 batch kernel hands its stripes out in units of 64 inside
        the workgroup, `units` of them per workgroup (0: one stripe per lane and launch)."""
        _check(_L.rrx_set_option(self._h, OPT_UNITS_PER_WORKGROUP, int(units)))

    @property
    def table_order(self):
        """None, or (before, after): the stride-2 table is laid out in an order profiled on the first large corpus this regex
        met; the mean number of distinct entries in the fullest LDS bank per half-wave on the sample, as numbered / as ordered."""
        b, a = C.c_double(0), C.c_double(0)
        return (b.value, a.value) if _L.rrx_table_order(self._h, C.byref(b), C.byref(a)) == 1 else None

    @property
    def table_order_pending(self):
        """True while the background search for a profiled table order is running."""
        return _L.rrx_table_order(self._h, None, None) == 2

    @property
    def byte_classes(self):
        return _L.rrx_byte_classes(self._h)

    @property
    def words_per_set(self):
        return _L.rrx_words_per_set(self._h)

    def program(self, kind):
        """Serialised device program (numpy uint32), or None if that form was not built."""
        import numpy as np
        n = _L.rrx_program_words(self._h, kind, None, 0)
        if not n:
            return None
        out = np.zeros(n, dtype=np.uint32)
        _L.rrx_program_words(self._h, kind, C.c_void_p(out.ctypes.data), n)
        return out

    @property
    def accepts_empty(self):
        return bool(_L.rrx_accepts_empty(self._h))
