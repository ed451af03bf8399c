"""ctypes loader for the CPU oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (roaringregex_amd/) never does.  See oracle/rr_oracle.h for the parity-pin status.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    import fcntl
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "rr_oracle.c")
    ref_so = os.path.join(_HERE, "_ref", "libref_bitset.so")

    def stale():
        return force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)

    def ref_missing():
        return os.path.isdir("/root/reference/src") and (force or not os.path.exists(ref_so))

    if stale() or ref_missing():
        with open(os.path.join(_HERE, ".build.lock"), "w") as lk:      # several processes may get here at once
            fcntl.flock(lk, fcntl.LOCK_EX)
            if stale():
                subprocess.check_call(["make", "-C", _HERE, "liboracle.so"], stdout=subprocess.DEVNULL)
            if ref_missing():
                subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)


def lib():
    global _LIB
    if _LIB is None:
        build()
        L = C.CDLL(os.path.join(_HERE, "liboracle.so"))
        L.rro_compile.restype = C.c_void_p
        L.rro_compile.argtypes = [C.c_char_p, C.c_char_p, C.c_size_t]
        L.rro_free.argtypes = [C.c_void_p]
        for f in ("rro_states_n", "rro_initial"):
            getattr(L, f).restype = C.c_uint32
            getattr(L, f).argtypes = [C.c_void_p]
        L.rro_set_class.restype = C.c_int
        L.rro_set_class.argtypes = [C.c_void_p]
        L.rro_is_final.restype = C.c_int
        L.rro_is_final.argtypes = [C.c_void_p, C.c_uint32]
        L.rro_row.restype = C.c_uint32
        L.rro_row.argtypes = [C.c_void_p, C.c_uint32, C.c_uint, C.c_int, C.c_void_p, C.c_uint32]
        L.rro_accepts.restype = C.c_int
        L.rro_accepts.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.rro_match_lines.restype = C.c_size_t
        L.rro_match_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.rro_search_all.restype = C.c_size_t
        L.rro_search_all.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.rro_search_lines.restype = C.c_size_t
        L.rro_search_lines.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        L.rro_set_simd.argtypes = [C.c_int]
        L.rro_simd.restype = C.c_int
        _LIB = L
    return _LIB


def set_simd(on):
    """The BitSet<2> / BitSet<4> step with the reference's vector ORs (BitSet.cc:8-21) or with scalar words; same results."""
    lib().rro_set_simd(1 if on else 0)


def simd():
    return bool(lib().rro_simd())


def ref_bitset_lib():
    """The reference's own BitSet.cc (oracle/_ref/libref_bitset.so), or None when it was never built."""
    p = os.path.join(_HERE, "_ref", "libref_bitset.so")
    if not os.path.exists(p):
        try:
            build()
        except Exception:
            return None
    if not os.path.exists(p):
        return None
    return C.CDLL(p)


class OracleError(ValueError):
    pass


class OracleRegex:
    """Reference semantics of Regex::RRegex (regex.h:212-228) for whole-string acceptance."""

    def __init__(self, pattern):
        if isinstance(pattern, str):
            pattern = pattern.encode("latin-1")
        self.pattern = pattern
        err = C.create_string_buffer(256)
        self._h = lib().rro_compile(pattern, err, 256)
        if not self._h:
            raise OracleError(err.value.decode())

    def __del__(self):
        if getattr(self, "_h", None):
            lib().rro_free(self._h)
            self._h = None

    @property
    def states_n(self):
        return lib().rro_states_n(self._h)

    @property
    def initial(self):
        return lib().rro_initial(self._h)

    @property
    def set_class(self):
        return lib().rro_set_class(self._h)

    def finals(self):
        return [s for s in range(self.states_n) if lib().rro_is_final(self._h, s)]

    def row(self, state, c, fwd=True):
        n = self.states_n
        buf = (C.c_uint32 * max(n, 1))()
        k = lib().rro_row(self._h, state, c, 1 if fwd else 0, buf, n)
        return list(buf[:k])

    def accepts(self, s):
        if isinstance(s, str):
            s = s.encode("latin-1")
        return bool(lib().rro_accepts(self._h, s, len(s)))

    def match_lines(self, data):
        """data: bytes / numpy uint8 array.  Returns numpy uint8 accept vector, one entry per line."""
        import numpy as np
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = len(a)
        nl = int((a == 10).sum())
        nlines = nl + (1 if n and a[-1] != 10 else 0)
        out = np.zeros(max(nlines, 1), dtype=np.uint8)
        got = lib().rro_match_lines(self._h, a.ctypes.data if n else None, n, out.ctypes.data, nlines)
        assert got == nlines, (got, nlines)
        return out[:nlines]

    def search_lines(self, data):
        """Per line the accepted substring [start, end) with the smallest end, then the smallest start (-1, -1 if none).
        Brute force over substrings with the reference's whole-string acceptance: short lines only."""
        import numpy as np
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = len(a)
        nlines = int((a == 10).sum()) + (1 if n and a[-1] != 10 else 0)
        st = np.full(max(nlines, 1), -1, dtype=np.int32)
        en = np.full(max(nlines, 1), -1, dtype=np.int32)
        got = lib().rro_search_lines(self._h, a.ctypes.data if n else None, n, st.ctypes.data, en.ctypes.data, nlines)
        assert got == nlines, (got, nlines)
        return st[:nlines], en[:nlines]

    def search_all(self, data):
        """All lazy matches per line, left to right -> (count[nlines] uint32, start[total] int32, end[total] int32),
        offsets relative to the line.  Brute force: short lines only."""
        import numpy as np
        a = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
        n = len(a)
        nlines = int((a == 10).sum()) + (1 if n and a[-1] != 10 else 0)
        cnt = np.zeros(max(nlines, 1), dtype=np.uint32)
        cap = n + nlines + 1                                   # a line of n bytes has at most n + 1 matches
        st = np.full(cap, -1, dtype=np.int32)
        en = np.full(cap, -1, dtype=np.int32)
        total = lib().rro_search_all(self._h, a.ctypes.data if n else None, n, cnt.ctypes.data, nlines, st.ctypes.data, en.ctypes.data, cap)
        return cnt[:nlines], st[:total], en[:total]
