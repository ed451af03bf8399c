"""ctypes wrapper of tools/libsynth.so: deterministic synthetic corpora of the BASELINE configs."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

KIND = {"c1": 1, "email": 2, "url": 3, "arepeat": 4, "kwlines": 5, "kwlog": 6, "ablines": 7, "ablong": 8, "email_long": 9, "url_long": 10}


def _lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "libsynth.so")
        if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "synth.c")):
            # several ranks may get here at once: build under an exclusive lock, re-check inside it
            import fcntl
            with open(os.path.join(_HERE, ".build.lock"), "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(os.path.join(_HERE, "synth.c")):
                    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)
        L = C.CDLL(so)
        L.synth_fill.restype = C.c_int
        L.synth_fill.argtypes = [C.c_int, C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.synth_fill_window.restype = C.c_int
        L.synth_fill_window.argtypes = [C.c_int, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int]
        L.synth_fnv1a.restype = C.c_uint64
        L.synth_fnv1a.argtypes = [C.c_void_p, C.c_size_t]
        _LIB = L
    return _LIB


def fill(kind, seed, out, chunk=0, threads=None):
    """Fill a writable uint8 numpy array (or anything exposing ctypes data) in place."""
    if threads is None:
        threads = min(os.cpu_count() or 1, 64)
    rc = _lib().synth_fill(KIND[kind] if isinstance(kind, str) else kind, seed, C.c_void_p(out.ctypes.data), out.size, chunk, threads)
    if rc:
        raise RuntimeError("synth_fill failed: %d" % rc)
    return out


def fill_window(kind, seed, first_byte, out, chunk=0, threads=None):
    """out = bytes [first_byte, first_byte + out.size) of the corpus (kind, seed); first_byte: a multiple of the chunk (1 MiB).
    A window of a corpus of any size without generating the rest (chunks are independent and end with '\\n')."""
    step = chunk or (1 << 20)
    assert first_byte % step == 0
    if threads is None:
        threads = min(os.cpu_count() or 1, 64)
    rc = _lib().synth_fill_window(KIND[kind] if isinstance(kind, str) else kind, seed, first_byte // step, C.c_void_p(out.ctypes.data), out.size, chunk, threads)
    if rc:
        raise RuntimeError("synth_fill_window failed: %d" % rc)
    return out


def corpus(kind, seed, nbytes, chunk=0, threads=None):
    return fill(kind, seed, np.empty(nbytes, dtype=np.uint8), chunk, threads)


def fnv1a(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    return int(_lib().synth_fnv1a(C.c_void_p(a.ctypes.data), a.size))
