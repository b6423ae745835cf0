"""oracle/ref_backend.py -- TEST INFRASTRUCTURE ONLY (never imported by the product package).

Drives the REAL reference CPU backend (oracle/_ref/libggml-ref.so, built from the sources under
/root/reference by oracle/Makefile.ref) through the same plug-in vtables and the same Python graph mirror
as the MI355X backend, so a graph can be executed on both and compared node for node -- what the
reference's tests/test-backend-ops.cpp does with ggml_backend_compare_graph_backend (ggml-backend.cpp:2006).

Used by: tests/ (parity checker) and bench.py's `cpu_baseline` leg ("kind": "reference").
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
REF_LIB = os.path.join(HERE, "_ref", "libggml-ref.so")


REF_LIB_V4 = os.path.join(HERE, "_ref", "v4", "libggml-ref.so")          # the same sources at -march=x86-64-v4 (oracle/Makefile.ref `v4`)
_V4_FLAGS = ("avx512f", "avx512bw", "avx512cd", "avx512dq", "avx512vl")


def ref_available():
    return os.path.exists(REF_LIB)


def ref_variant():
    """Which build ref_lib() loads.  The parity tests always use the x86-64-v3 build (the AVX2 code paths the golden vectors were checked on);
    ORACLE_REF_VARIANT=v4 -- set by bench.py's cpu_baseline leg only -- selects the AVX-512 build when it exists and this host CPU has the level."""
    if os.environ.get("ORACLE_REF_VARIANT") == "v4" and os.path.exists(REF_LIB_V4):
        try:
            flags = next(l for l in open("/proc/cpuinfo") if l.startswith("flags")).split()
            if all(f in flags for f in _V4_FLAGS):
                return "x86-64-v4", REF_LIB_V4
        except Exception:
            pass
    return "x86-64-v3", REF_LIB


_REF = None


def ref_lib():
    global _REF
    if _REF is None:
        if not ref_available():
            raise RuntimeError(f"{REF_LIB} missing: run `make -f oracle/Makefile.ref` where /root/reference exists")
        lib = C.CDLL(ref_variant()[1], mode=C.RTLD_GLOBAL)
        lib.ggml_backend_cpu_init.restype = C.c_void_p
        lib.ggml_backend_cpu_buffer_type.restype = C.c_void_p
        lib.ggml_backend_cpu_set_n_threads.argtypes = [C.c_void_p, C.c_int]
        lib.ggml_backend_buffer_free.argtypes = [C.c_void_p]
        lib.ggml_backend_free.argtypes = [C.c_void_p]
        lib.ggml_cpu_init()          # fills the f16->f32 lookup table the x86 vec_dot kernels read (ggml-cpu.c:3540-3560)
        _REF = lib
    return _REF


def make_ref_cpu_backend(pkg, n_threads=None):
    """pkg: the loaded llama.cpp-omni_amd package (for its ctypes mirror classes)."""
    lib = ref_lib()

    class RefCpuBackend(pkg.Backend):
        def __init__(self):
            self.lib = None
            self.ref = lib
            be = lib.ggml_backend_cpu_init()
            if n_threads:
                lib.ggml_backend_cpu_set_n_threads(be, int(n_threads))
            self._attach(be, lib.ggml_backend_cpu_buffer_type(), None)

        def set_n_threads(self, n):
            self.ref.ggml_backend_cpu_set_n_threads(self.be, int(n))

        def name(self):
            return "CPU(reference)"

        def free_buffer(self, b):
            self._buffers.remove(b)
            self.ref.ggml_backend_buffer_free(b)

        def set_option(self, key, value):
            return -1

        def close(self):
            if self.be:
                for b in list(self._buffers):
                    self.free_buffer(b)
                self.ref.ggml_backend_free(self.be)
                self.be = None

    return RefCpuBackend()
