"""ctypes binding of oracle/libzkr_oracle.so -- CPU ORACLE, TEST INFRASTRUCTURE ONLY
(see oracle/zkr_oracle.c header).  Build with `make -C oracle`."""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


_path = None   # the library lib() loads; use_native() points it at a build for THIS host's CPU


def use_native():
    """CPU-baseline timing only (bench.py cpu_baseline): compile zkr_oracle.c with `-O3 -march=native` ON THIS HOST (the
    shipped libzkr_oracle.so is -march=x86-64-v2 so that it runs on every box: no BMI2 / ADX) and make lib() load that
    build.  Returns a short description of what will be loaded; falls back to the portable build when gcc or the compile
    fails.  Must be called before the first lib()."""
    global _path
    import subprocess
    import tempfile
    if _lib is not None:
        return "already loaded: " + os.path.basename(_path or "libzkr_oracle.so")
    src = os.path.join(_HERE, "zkr_oracle.c")
    flags = ["-O3", "-march=native", "-fPIC", "-fopenmp", "-shared"]
    for out_dir in (_HERE, tempfile.gettempdir()):
        out = os.path.join(out_dir, "libzkr_oracle_native.so")
        try:
            subprocess.check_call(["gcc"] + flags + ["-o", out, src], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
            ctypes.CDLL(out)   # an unloadable build (missing libgomp ...) must not replace the portable one
            _path = out
            return "gcc " + " ".join(flags[:2]) + " (built on this host)"
        except Exception:
            continue
    return "gcc -O3 -march=x86-64-v2 (portable build: the native compile failed on this host)"


def use_fastest(pk: bytes, witness: bytes):
    """CPU-baseline timing only: build the native library (use_native) and time ONE single-thread proof of the given small case
    with it and with the portable build; lib() then loads whichever was faster ON THIS HOST (gcc's -march=native is not always a
    win for this code: no ADX/MULX chains are generated, and on some hosts the wider vector unit only changes the scheduling).
    Returns a description with both times."""
    global _path
    import time
    if _lib is not None:
        return "already loaded: " + os.path.basename(_path or "libzkr_oracle.so")
    what = use_native()
    native, portable = _path, os.path.join(_HERE, "libzkr_oracle.so")
    if native is None:
        return what
    times = {}
    for name, path in (("native", native), ("x86-64-v2", portable)):
        L = ctypes.CDLL(path)
        L.zo_prove.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        L.zo_prove.restype = ctypes.c_int
        out, tm = ctypes.create_string_buffer(256), (ctypes.c_double * 3)()
        one = (1).to_bytes(32, "little")
        best = None
        for _ in range(2):
            t0 = time.perf_counter()
            if L.zo_prove(pk, len(pk), witness, len(witness) // 32, one, one, out, tm):
                raise RuntimeError("zo_prove failed")
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        times[name] = best
    pick = "native" if times["native"] <= times["x86-64-v2"] else "x86-64-v2"
    _path = native if pick == "native" else None
    return "gcc -O3 -march=%s (the faster on this host: native %.3f s, x86-64-v2 %.3f s on a small proof)" % (pick, times["native"], times["x86-64-v2"])


def lib():
    global _lib
    if _lib is None:
        path = _path or os.path.join(_HERE, "libzkr_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle/libzkr_oracle.so missing: run `make -C oracle` (or __graft_entry__.build())")
        L = ctypes.CDLL(path)
        L.zo_ntt.argtypes = [ctypes.c_char_p, ctypes.c_uint, ctypes.c_int]
        L.zo_ntt.restype = None
        for f in (L.zo_msm_g1, L.zo_msm_g2):
            f.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
            f.restype = ctypes.c_int
        L.zo_calc_h.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.zo_calc_h.restype = ctypes.c_int
        L.zo_prove.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p,
                               ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
        L.zo_prove.restype = ctypes.c_int
        L.zo_prove_mt.argtypes = L.zo_prove.argtypes + [ctypes.c_int]
        L.zo_prove_mt.restype = ctypes.c_int
        L.zo_max_threads.restype = ctypes.c_int
        L.zo_fr_dot.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p]
        L.zo_fr_dot.restype = None
        for f in (L.zo_fq_mul_std, L.zo_fr_mul_std):
            f.argtypes = [ctypes.c_char_p] * 3
            f.restype = None
        _lib = L
    return _lib


def ntt(data: bytes, inverse=False) -> bytes:
    n = len(data) // 32
    logn = n.bit_length() - 1
    assert 1 << logn == n
    buf = ctypes.create_string_buffer(bytes(data), len(data))
    lib().zo_ntt(buf, logn, 1 if inverse else 0)
    return buf.raw


def msm_g1(points: bytes, scalars: bytes):
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(64)
    inf = lib().zo_msm_g1(bytes(points), bytes(scalars), n, out)
    return None if inf else out.raw


def msm_g2(points: bytes, scalars: bytes):
    n = len(scalars) // 32
    out = ctypes.create_string_buffer(128)
    inf = lib().zo_msm_g2(bytes(points), bytes(scalars), n, out)
    return None if inf else out.raw


def calc_h(pk: bytes, witness: bytes) -> bytes:
    m = int.from_bytes(pk[8:12], "little")
    out = ctypes.create_string_buffer(32 * m)
    rc = lib().zo_calc_h(pk, len(pk), witness, len(witness) // 32, out)
    if rc:
        raise RuntimeError("zo_calc_h failed: %d" % rc)
    return out.raw


def prove(pk: bytes, witness: bytes, r: int, s: int, want_timings=False):
    out = ctypes.create_string_buffer(256)
    tm = (ctypes.c_double * 3)()
    rc = lib().zo_prove(pk, len(pk), witness, len(witness) // 32, r.to_bytes(32, "little"), s.to_bytes(32, "little"), out, tm)
    if rc:
        raise RuntimeError("zo_prove failed: %d" % rc)
    return (out.raw, list(tm)) if want_timings else out.raw


def prove_mt(pk: bytes, witness: bytes, r: int, s: int, threads=0, want_timings=False):
    """zo_prove on all host threads (OpenMP; threads = 0: every hardware thread): same bytes as prove().
    timings = [calc_h s, msm s, total s, threads used]."""
    out = ctypes.create_string_buffer(256)
    tm = (ctypes.c_double * 4)()
    rc = lib().zo_prove_mt(pk, len(pk), witness, len(witness) // 32, r.to_bytes(32, "little"), s.to_bytes(32, "little"), out, tm, threads)
    if rc:
        raise RuntimeError("zo_prove_mt failed: %d" % rc)
    return (out.raw, list(tm)) if want_timings else out.raw


def max_threads() -> int:
    return lib().zo_max_threads()


def fr_dot(a: bytes, b: bytes) -> int:
    """sum_i a_i b_i mod r over two arrays of 32-byte LE standard-form elements."""
    assert len(a) == len(b) and len(a) % 32 == 0
    out = ctypes.create_string_buffer(32)
    lib().zo_fr_dot(a, b, len(a) // 32, out)
    return int.from_bytes(out.raw, "little")
