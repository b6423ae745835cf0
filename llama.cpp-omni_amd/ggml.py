"""ctypes mirror of the ggml structs/vtables the MI355X backend is driven through, plus a minimal graph
builder that follows the reference's op constructors (ggml/src/ggml.c) one-for-one.

Struct layouts restate ``csrc/ggml_abi.h`` (checked against the reference headers by tests/test_abi.py).
"""
import ctypes as C
import os
import struct

import numpy as np

__all__ = [
    "lib_path", "load_library", "Backend", "backend", "Context", "Tensor",
    "GGML_TYPE_F32", "GGML_TYPE_F16", "GGML_TYPE_Q8_0", "GGML_TYPE_Q4_K", "GGML_TYPE_Q6_K", "GGML_TYPE_I32", "GGML_TYPE_I64",
    "GGML_BACKEND_BUFFER_USAGE_ANY", "GGML_BACKEND_BUFFER_USAGE_WEIGHTS", "GGML_BACKEND_BUFFER_USAGE_COMPUTE", "GGML_ROPE_TYPE_NORMAL", "GGML_ROPE_TYPE_NEOX", "type_traits", "row_size", "OP", "GLU", "UNARY", "ggml_tensor", "ggml_cgraph",
]

# ------------------------------------------------------------------------------------------------ constants
GGML_TYPE_F32, GGML_TYPE_F16, GGML_TYPE_Q8_0 = 0, 1, 8
GGML_TYPE_Q4_K, GGML_TYPE_Q6_K, GGML_TYPE_Q8_K = 12, 14, 15
GGML_TYPE_I32, GGML_TYPE_I64 = 26, 27
GGML_ROPE_TYPE_NORMAL, GGML_ROPE_TYPE_NEOX = 0, 2
GGML_PREC_F32 = 10
GGML_BACKEND_BUFFER_USAGE_ANY, GGML_BACKEND_BUFFER_USAGE_WEIGHTS, GGML_BACKEND_BUFFER_USAGE_COMPUTE = 0, 1, 2   # ggml-backend.h:49-53


class OP:
    NONE, DUP, ADD, SUB, MUL, DIV = 0, 1, 2, 6, 7, 8
    SQR, SQRT, LOG, SIN, COS, SUM_ROWS, REPEAT, CONCAT = 9, 10, 11, 12, 13, 15, 19, 21
    CLAMP, CONV_TRANSPOSE_1D, PAD, PAD_REFLECT_1D, ARANGE, TIMESTEP_EMBEDDING, LEAKY_RELU = 49, 50, 62, 63, 65, 66, 68
    NORM, RMS_NORM, MUL_MAT, SCALE, CPY, CONT, RESHAPE, VIEW, PERMUTE, TRANSPOSE = 23, 24, 28, 31, 33, 34, 35, 36, 37, 38
    IM2COL = 51
    POOL_1D = 58
    POOL_2D = 59
    GET_ROWS, SET_ROWS, SOFT_MAX, ROPE, FLASH_ATTN_EXT, UNARY, GLU = 39, 41, 45, 47, 69, 80, 89


class GLU:
    REGLU, GEGLU, SWIGLU, SWIGLU_OAI, GEGLU_ERF, GEGLU_QUICK = range(6)


class UNARY:
    ABS, SGN, NEG, STEP, TANH, ELU, RELU, SIGMOID, GELU, GELU_QUICK, SILU, HARDSWISH, HARDSIGMOID, EXP, GELU_ERF = range(15)


_TRAITS = {  # type -> (block elements, block bytes, numpy dtype or None)
    GGML_TYPE_F32: (1, 4, np.float32), GGML_TYPE_F16: (1, 2, np.float16), GGML_TYPE_Q8_0: (32, 34, None),
    GGML_TYPE_Q4_K: (256, 144, None), GGML_TYPE_Q6_K: (256, 210, None), GGML_TYPE_Q8_K: (256, 292, None),
    GGML_TYPE_I32: (1, 4, np.int32), GGML_TYPE_I64: (1, 8, np.int64),
    2: (32, 18, None), 3: (32, 20, None), 6: (32, 22, None), 7: (32, 24, None),      # Q4_0 Q4_1 Q5_0 Q5_1
    10: (256, 84, None), 11: (256, 110, None), 13: (256, 176, None),                # Q2_K Q3_K Q5_K
    30: (1, 2, np.uint16),                                                          # BF16 (as raw 16-bit words)
}


def type_traits(t):
    return _TRAITS[t]


def row_size(t, ne):
    blck, size, _ = _TRAITS[t]
    assert ne % blck == 0, "row length must be a multiple of the block size"
    return ne // blck * size


# ------------------------------------------------------------------------------------------------ structs
class ggml_tensor(C.Structure):
    pass


ggml_tensor._fields_ = [
    ("type", C.c_int), ("buffer", C.c_void_p), ("ne", C.c_int64 * 4), ("nb", C.c_size_t * 4), ("op", C.c_int),
    ("op_params", C.c_int32 * 16), ("flags", C.c_int32), ("src", C.POINTER(ggml_tensor) * 10),
    ("view_src", C.POINTER(ggml_tensor)), ("view_offs", C.c_size_t), ("data", C.c_void_p), ("name", C.c_char * 64),
    ("extra", C.c_void_p), ("padding", C.c_char * 8),
]
assert C.sizeof(ggml_tensor) == 336


class ggml_hash_set(C.Structure):
    _fields_ = [("size", C.c_size_t), ("used", C.c_void_p), ("keys", C.c_void_p)]


class ggml_cgraph(C.Structure):
    _fields_ = [("size", C.c_int), ("n_nodes", C.c_int), ("n_leafs", C.c_int), ("nodes", C.POINTER(C.POINTER(ggml_tensor))),
                ("grads", C.c_void_p), ("grad_accs", C.c_void_p), ("leafs", C.c_void_p), ("use_counts", C.c_void_p),
                ("visited_hash_set", ggml_hash_set), ("order", C.c_int)]


assert C.sizeof(ggml_cgraph) == 88
_vp, _sz = C.c_void_p, C.c_size_t
_TP = C.POINTER(ggml_tensor)


class buft_i(C.Structure):
    _fields_ = [("get_name", C.CFUNCTYPE(C.c_char_p, _vp)), ("alloc_buffer", C.CFUNCTYPE(_vp, _vp, _sz)),
                ("get_alignment", C.CFUNCTYPE(_sz, _vp)), ("get_max_size", _vp), ("get_alloc_size", _vp), ("is_host", C.CFUNCTYPE(C.c_bool, _vp))]


class buft_t(C.Structure):
    _fields_ = [("iface", buft_i), ("device", _vp), ("context", _vp)]


class buffer_i(C.Structure):
    _fields_ = [("free_buffer", C.CFUNCTYPE(None, _vp)), ("get_base", C.CFUNCTYPE(_vp, _vp)), ("init_tensor", C.CFUNCTYPE(C.c_int, _vp, _TP)),
                ("memset_tensor", C.CFUNCTYPE(None, _vp, _TP, C.c_uint8, _sz, _sz)), ("set_tensor", C.CFUNCTYPE(None, _vp, _TP, _vp, _sz, _sz)),
                ("get_tensor", C.CFUNCTYPE(None, _vp, _TP, _vp, _sz, _sz)), ("cpy_tensor", C.CFUNCTYPE(C.c_bool, _vp, _TP, _TP)),
                ("clear", C.CFUNCTYPE(None, _vp, C.c_uint8)), ("reset", _vp)]


class buffer_t(C.Structure):
    _fields_ = [("iface", buffer_i), ("buft", _vp), ("context", _vp), ("size", _sz), ("usage", C.c_int)]


class backend_i(C.Structure):
    _fields_ = [("get_name", C.CFUNCTYPE(C.c_char_p, _vp)), ("free", C.CFUNCTYPE(None, _vp)),
                ("set_tensor_async", C.CFUNCTYPE(None, _vp, _TP, _vp, _sz, _sz)), ("get_tensor_async", C.CFUNCTYPE(None, _vp, _TP, _vp, _sz, _sz)),
                ("cpy_tensor_async", C.CFUNCTYPE(C.c_bool, _vp, _vp, _TP, _TP)), ("synchronize", C.CFUNCTYPE(None, _vp)),
                ("graph_plan_create", _vp), ("graph_plan_free", _vp), ("graph_plan_update", _vp), ("graph_plan_compute", _vp),
                ("graph_compute", C.CFUNCTYPE(C.c_int, _vp, C.POINTER(ggml_cgraph))), ("event_record", C.CFUNCTYPE(None, _vp, _vp)),
                ("event_wait", C.CFUNCTYPE(None, _vp, _vp)), ("graph_optimize", _vp)]


class backend_t(C.Structure):
    _fields_ = [("guid", _vp), ("iface", backend_i), ("device", _vp), ("context", _vp)]


class dev_caps(C.Structure):
    _fields_ = [("async_", C.c_bool), ("host_buffer", C.c_bool), ("buffer_from_host_ptr", C.c_bool), ("events", C.c_bool)]


class dev_props(C.Structure):
    _fields_ = [("name", C.c_char_p), ("description", C.c_char_p), ("memory_free", _sz), ("memory_total", _sz), ("type", C.c_int),
                ("device_id", C.c_char_p), ("caps", dev_caps)]


class device_i(C.Structure):
    _fields_ = [("get_name", C.CFUNCTYPE(C.c_char_p, _vp)), ("get_description", C.CFUNCTYPE(C.c_char_p, _vp)),
                ("get_memory", C.CFUNCTYPE(None, _vp, C.POINTER(_sz), C.POINTER(_sz))), ("get_type", C.CFUNCTYPE(C.c_int, _vp)),
                ("get_props", C.CFUNCTYPE(None, _vp, C.POINTER(dev_props))), ("init_backend", C.CFUNCTYPE(_vp, _vp, C.c_char_p)),
                ("get_buffer_type", C.CFUNCTYPE(_vp, _vp)), ("get_host_buffer_type", C.CFUNCTYPE(_vp, _vp)), ("buffer_from_host_ptr", _vp),
                ("supports_op", C.CFUNCTYPE(C.c_bool, _vp, _TP)), ("supports_buft", C.CFUNCTYPE(C.c_bool, _vp, _vp)),
                ("offload_op", C.CFUNCTYPE(C.c_bool, _vp, _TP)), ("event_new", C.CFUNCTYPE(_vp, _vp)), ("event_free", C.CFUNCTYPE(None, _vp, _vp)),
                ("event_synchronize", C.CFUNCTYPE(None, _vp, _vp))]


class device_t(C.Structure):
    _fields_ = [("iface", device_i), ("reg", _vp), ("context", _vp)]


class reg_i(C.Structure):
    _fields_ = [("get_name", C.CFUNCTYPE(C.c_char_p, _vp)), ("get_device_count", C.CFUNCTYPE(_sz, _vp)),
                ("get_device", C.CFUNCTYPE(_vp, _vp, _sz)), ("get_proc_address", C.CFUNCTYPE(_vp, _vp, C.c_char_p))]


class reg_t(C.Structure):
    _fields_ = [("api_version", C.c_int), ("iface", reg_i), ("context", _vp)]


# ------------------------------------------------------------------------------------------------ library
def lib_path():
    if os.environ.get("MI355X_LIB"):                  # tuning builds (csrc/Makefile VARIANT=...)
        return os.environ["MI355X_LIB"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libggml-mi355x.so")


_LIB = None


def load_library():
    """dlopen libggml-mi355x.so.  Fails loudly when it has not been built: there is no fallback path."""
    global _LIB
    if _LIB is None:
        p = lib_path()
        if not os.path.exists(p):
            raise RuntimeError(f"{p} is missing: build it with `make -C llama.cpp-omni_amd/csrc` (or __graft_entry__.build()); "
                               "the MI355X backend has no CPU fallback")
        lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
        lib.ggml_backend_init.restype = C.POINTER(reg_t)
        lib.ggml_backend_mi355x_reg.restype = C.POINTER(reg_t)
        lib.ggml_backend_score.restype = C.c_int
        lib.mi355x_host_buffer_free.argtypes = [_vp]
        lib.mi355x_timed_event_new.restype = _vp
        lib.mi355x_timed_event_record.argtypes = [_vp, _vp]
        lib.mi355x_timed_event_elapsed_ms.argtypes = [_vp, _vp]
        lib.mi355x_timed_event_elapsed_ms.restype = C.c_float
        lib.mi355x_timed_event_free.argtypes = [_vp]
        lib.mi355x_set_option.argtypes = [_vp, C.c_char_p, C.c_long]
        lib.mi355x_get_stat.argtypes = [_vp, C.c_char_p]
        lib.mi355x_get_stat.restype = C.c_double
        lib.mi355x_debug_quantize.argtypes = [_vp, C.c_int, _vp, C.c_long, C.c_long, _vp]
        lib.mi355x_debug_quantize.restype = C.c_long
        lib.mi355x_handoff.argtypes = [_vp, _vp, _vp, _vp, C.c_size_t]
        lib.mi355x_handoff_tensor.argtypes = [_vp, _vp, _vp, _vp]
        lib.mi355x_handoff_count.argtypes = [C.c_int]
        lib.mi355x_handoff_count.restype = C.c_long
        lib.mi355x_module_device.argtypes = [C.c_char_p]
        _LIB = lib
    return _LIB


class Tensor:
    """Python handle on one ggml_tensor (kept alive by its Context)."""

    def __init__(self, ctx, t):
        self.ctx, self.t = ctx, t

    @property
    def ne(self):
        return tuple(self.t.ne)

    @property
    def nb(self):
        return tuple(self.t.nb)

    @property
    def type(self):
        return self.t.type

    @property
    def ptr(self):
        return C.pointer(self.t)

    def nbytes(self):
        blck, size, _ = _TRAITS[self.t.type]
        if any(n == 0 for n in self.t.ne):
            return 0
        if blck == 1:
            return size + sum((self.t.ne[i] - 1) * self.t.nb[i] for i in range(4))
        return self.t.ne[0] * self.t.nb[0] // blck + sum((self.t.ne[i] - 1) * self.t.nb[i] for i in range(1, 4))

    def nelements(self):
        return self.t.ne[0] * self.t.ne[1] * self.t.ne[2] * self.t.ne[3]

    def set_name(self, s):
        self.t.name = s.encode()[:63]
        return self


class Backend:
    """One device + one stream of the MI355X backend, driven purely through the plug-in vtables."""

    def __init__(self, device_index=0):
        self.lib = load_library()
        self.reg = self.lib.ggml_backend_mi355x_reg()       # (the direct-link entry: ggml_backend_init, the LOADER's entry, reports the devices once per process)
        if not self.reg or self.reg.contents.api_version != 2:
            raise RuntimeError("ggml_backend_mi355x_reg: bad registry / api_version")
        n = self.reg.contents.iface.get_device_count(self.reg)
        if n == 0:
            raise RuntimeError("libggml-mi355x.so: no gfx950 device visible (ggml_backend_score() == 0)")
        if device_index >= n:
            raise RuntimeError(f"device {device_index} requested but only {n} gfx950 device(s) visible")
        dev = self.reg.contents.iface.get_device(self.reg, device_index)
        dev_s = C.cast(dev, C.POINTER(device_t)).contents
        self._attach(dev_s.iface.init_backend(dev, None), dev_s.iface.get_buffer_type(dev), dev)

    def _attach(self, be, buft, dev=None):
        """Bind to an initialised ggml_backend_t + buffer type (any backend that speaks the plug-in ABI)."""
        self.be, self.buft, self.dev = be, buft, dev
        self.dev_s = C.cast(dev, C.POINTER(device_t)).contents if dev else None
        self.be_s = C.cast(self.be, C.POINTER(backend_t)).contents
        self.buft_s = C.cast(self.buft, C.POINTER(buft_t)).contents
        self.alignment = self.buft_s.iface.get_alignment(self.buft)
        self._buffers = []

    # -- device info
    def name(self):
        return self.dev_s.iface.get_name(self.dev).decode()

    def description(self):
        return self.dev_s.iface.get_description(self.dev).decode()

    def memory(self):
        f, t = _sz(), _sz()
        self.dev_s.iface.get_memory(self.dev, C.byref(f), C.byref(t))
        return f.value, t.value

    def supports_op(self, tensor):
        return bool(self.dev_s.iface.supports_op(self.dev, tensor.ptr))

    # -- buffers
    def alloc_buffer(self, size):
        b = self.buft_s.iface.alloc_buffer(self.buft, size)
        if not b:
            raise MemoryError(f"alloc_buffer({size}) failed")
        self._buffers.append(b)
        return b

    def free_buffer(self, b):
        self._buffers.remove(b)
        self.lib.mi355x_host_buffer_free(b)

    def tensor_set(self, tensor, data, offset=0):
        data = np.ascontiguousarray(data)
        buf = C.cast(tensor.t.buffer, C.POINTER(buffer_t)).contents
        assert offset + data.nbytes <= tensor.nbytes(), (offset, data.nbytes, tensor.nbytes())
        buf.iface.set_tensor(tensor.t.buffer, tensor.ptr, data.ctypes.data, offset, data.nbytes)

    def tensor_get(self, tensor, dtype=None, offset=0, nbytes=None):
        nbytes = tensor.nbytes() - offset if nbytes is None else nbytes
        out = np.empty(nbytes, dtype=np.uint8)
        buf = C.cast(tensor.t.buffer, C.POINTER(buffer_t)).contents
        self.synchronize()
        buf.iface.get_tensor(tensor.t.buffer, tensor.ptr, out.ctypes.data, offset, nbytes)
        if dtype is None:
            dtype = _TRAITS[tensor.t.type][2]
        return out if dtype is None else out.view(dtype)

    def host_array(self, nbytes):
        """uint8 numpy array over a pinned host buffer of the device's host buffer type (for async copies)."""
        hbt = self.dev_s.iface.get_host_buffer_type(self.dev)
        hbt_s = C.cast(hbt, C.POINTER(buft_t)).contents
        b = hbt_s.iface.alloc_buffer(hbt, nbytes)
        if not b:
            raise MemoryError("pinned host allocation failed")
        self._buffers.append(b)
        base = C.cast(b, C.POINTER(buffer_t)).contents.iface.get_base(b)
        return np.ctypeslib.as_array(C.cast(base, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def tensor_set_async(self, tensor, host, offset=0):
        """host: numpy array (ideally pinned, see host_array) that stays alive until synchronize()."""
        if self.be_s.iface.set_tensor_async:                          # optional in the ABI: synchronous fallback like ggml_backend_tensor_set_async
            self.be_s.iface.set_tensor_async(self.be, tensor.ptr, host.ctypes.data, offset, host.nbytes)
        else:
            buf = C.cast(tensor.t.buffer, C.POINTER(buffer_t)).contents
            buf.iface.set_tensor(tensor.t.buffer, tensor.ptr, host.ctypes.data, offset, host.nbytes)

    def tensor_get_async(self, tensor, host, offset=0):
        if self.be_s.iface.get_tensor_async:
            self.be_s.iface.get_tensor_async(self.be, tensor.ptr, host.ctypes.data, offset, host.nbytes)
        else:
            buf = C.cast(tensor.t.buffer, C.POINTER(buffer_t)).contents
            buf.iface.get_tensor(tensor.t.buffer, tensor.ptr, host.ctypes.data, offset, host.nbytes)

    # -- stream
    def graph_compute(self, graph):
        st = self.be_s.iface.graph_compute(self.be, C.byref(graph.g))
        if st != 0:
            raise RuntimeError(f"graph_compute returned status {st}")

    def synchronize(self):
        if self.be_s.iface.synchronize:                               # optional in the ABI (NULL for synchronous backends)
            self.be_s.iface.synchronize(self.be)

    def set_option(self, key, value):
        return self.lib.mi355x_set_option(self.be, key.encode(), int(value))

    def get_stat(self, key):
        return self.lib.mi355x_get_stat(self.be, key.encode())

    def handoff_tensor(self, src, dst_backend, dst):
        """device-to-device hand-off of a dense tensor to a tensor of another backend (RCCL send / recv; include/ggml-mi355x.h)"""
        return self.lib.mi355x_handoff_tensor(self.be, src.ptr, dst_backend.be, dst.ptr)

    def timed_event(self):
        return self.lib.mi355x_timed_event_new()

    def record(self, ev):
        self.lib.mi355x_timed_event_record(ev, self.be)

    def elapsed_ms(self, a, b):
        return self.lib.mi355x_timed_event_elapsed_ms(a, b)

    def close(self):
        if self.be:
            self.synchronize()
            for b in list(self._buffers):
                self.free_buffer(b)
            self.be_s.iface.free(self.be)
            self.be = None


_BACKENDS = {}


def backend(device_index=0):
    if device_index not in _BACKENDS:
        _BACKENDS[device_index] = Backend(device_index)
    return _BACKENDS[device_index]


class Graph:
    def __init__(self, nodes):
        self.nodes = nodes
        self.arr = (C.POINTER(ggml_tensor) * len(nodes))(*[n.ptr for n in nodes])
        self.g = ggml_cgraph()
        self.g.size = len(nodes)
        self.g.n_nodes = len(nodes)
        self.g.n_leafs = 0
        self.g.nodes = C.cast(self.arr, C.POINTER(C.POINTER(ggml_tensor)))


def _f32_bits(x):
    return struct.unpack("<i", struct.pack("<f", float(x)))[0]


class Context:
    """A bag of tensors + the op constructors of ggml.c, restricted to what the hot path uses.

    Every constructor mirrors its reference twin (same result shape/type, same op_params layout, same src slots):
    ggml_new_tensor :1620, ggml_view_* :3370-3470, ggml_reshape :3240, ggml_permute :3480, ggml_cont :3190,
    ggml_cpy :3150, ggml_mul_mat :3090, ggml_rms_norm :2960, ggml_rope_ext :3990, ggml_soft_max_ext :3830,
    ggml_get_rows :3640, ggml_set_rows :3690, ggml_flash_attn_ext :4890, ggml_glu :2770 (reference ggml/src/ggml.c).
    """

    def __init__(self, be):
        self.be = be
        self.tensors = []      # every tensor created, in creation order
        self.nodes = []        # op nodes in creation order == a valid topological order
        self.buffer = None

    # ---- creation
    def _new(self, type_, ne, view_src=None, view_offs=0):
        ne = list(ne) + [1] * (4 - len(ne))
        blck, size, _ = _TRAITS[type_]
        assert ne[0] % blck == 0
        t = ggml_tensor()
        t.type = type_
        for i in range(4):
            t.ne[i] = ne[i]
        t.nb[0] = size
        t.nb[1] = size * (ne[0] // blck)
        t.nb[2] = t.nb[1] * ne[1]
        t.nb[3] = t.nb[2] * ne[2]
        if view_src is not None:
            root = view_src.t.view_src.contents if view_src.t.view_src else view_src.t
            t.view_src = C.pointer(root)
            t.view_offs = view_offs + (view_src.t.view_offs if view_src.t.view_src else 0)
        T = Tensor(self, t)
        T._view_of = view_src
        self.tensors.append(T)
        return T

    def new_tensor(self, type_, *ne):
        return self._new(type_, ne)

    def _op(self, T, op, srcs, params=()):
        T.t.op = op
        for i, s in enumerate(srcs):
            if s is not None:
                T.t.src[i] = s.ptr
        for i, p in enumerate(params):
            T.t.op_params[i] = p
        T._srcs = srcs
        self.nodes.append(T)
        return T

    # ---- views (no data movement)
    def view_2d(self, a, ne0, ne1, nb1, offset):
        T = self._new(a.type, (ne0, ne1), view_src=a, view_offs=offset)
        T.t.nb[1] = nb1
        T.t.nb[2] = nb1 * ne1
        T.t.nb[3] = T.t.nb[2]
        return self._op(T, OP.VIEW, [a])

    def view_3d(self, a, ne0, ne1, ne2, nb1, nb2, offset):
        T = self._new(a.type, (ne0, ne1, ne2), view_src=a, view_offs=offset)
        T.t.nb[1], T.t.nb[2] = nb1, nb2
        T.t.nb[3] = nb2 * ne2
        return self._op(T, OP.VIEW, [a])

    def view_4d(self, a, ne0, ne1, ne2, ne3, nb1, nb2, nb3, offset):
        T = self._new(a.type, (ne0, ne1, ne2, ne3), view_src=a, view_offs=offset)
        T.t.nb[1], T.t.nb[2], T.t.nb[3] = nb1, nb2, nb3
        return self._op(T, OP.VIEW, [a])

    def reshape(self, a, *ne):
        assert int(np.prod(ne)) == a.nelements()
        T = self._new(a.type, ne, view_src=a)
        return self._op(T, OP.RESHAPE, [a])

    def permute(self, a, ax0, ax1, ax2, ax3):
        axes = (ax0, ax1, ax2, ax3)
        assert sorted(axes) == [0, 1, 2, 3]
        T = self._new(a.type, a.ne, view_src=a)
        for i, ax in enumerate(axes):
            T.t.ne[ax] = a.t.ne[i]
            T.t.nb[ax] = a.t.nb[i]
        return self._op(T, OP.PERMUTE, [a], axes)

    def transpose(self, a):
        T = self._new(a.type, a.ne, view_src=a)
        for i in range(4):                                            # (ggml_transpose: every stride is the source's, dims 0 / 1 swapped)
            T.t.nb[i] = a.t.nb[i]
        T.t.ne[0], T.t.ne[1] = a.t.ne[1], a.t.ne[0]
        T.t.nb[0], T.t.nb[1] = a.t.nb[1], a.t.nb[0]
        return self._op(T, OP.TRANSPOSE, [a])

    # ---- compute ops
    def cont(self, a, *ne):
        T = self._new(a.type, ne if ne else a.ne)
        return self._op(T, OP.CONT, [a])

    def cpy(self, a, b):
        assert a.nelements() == b.nelements()
        T = self._new(b.type, b.ne, view_src=b)
        for i in range(4):
            T.t.nb[i] = b.t.nb[i]
        return self._op(T, OP.CPY, [a, b])

    def cast(self, a, type_):
        T = self._new(type_, a.ne)
        return self._op(T, OP.CPY, [a, T])

    def _bin(self, op, a, b):
        T = self._new(a.type, a.ne)
        return self._op(T, op, [a, b])

    def add(self, a, b):
        return self._bin(OP.ADD, a, b)

    def sub(self, a, b):
        return self._bin(OP.SUB, a, b)

    def mul(self, a, b):
        return self._bin(OP.MUL, a, b)

    def div(self, a, b):
        return self._bin(OP.DIV, a, b)

    def scale(self, a, s, b=0.0):
        T = self._new(a.type, a.ne)
        return self._op(T, OP.SCALE, [a], (_f32_bits(s), _f32_bits(b)))

    # ---- ops of the Token2Wav graphs (constructors as in ggml.c: result shapes and op_params)
    def _math(self, op, a, params=()):
        T = self._new(a.type, a.ne)
        return self._op(T, op, [a], params)

    def sqr(self, a):
        return self._math(OP.SQR, a)

    def sqrt(self, a):
        return self._math(OP.SQRT, a)

    def log(self, a):
        return self._math(OP.LOG, a)

    def sin(self, a):
        return self._math(OP.SIN, a)

    def cos(self, a):
        return self._math(OP.COS, a)

    def clamp(self, a, lo, hi):
        """ggml_clamp is always in place: the result is a view of `a`"""
        T = self._new(a.type, a.ne, view_src=a)
        for i in range(4):
            T.t.nb[i] = a.t.nb[i]
        return self._op(T, OP.CLAMP, [a], (_f32_bits(lo), _f32_bits(hi)))

    def leaky_relu(self, a, slope):
        return self._math(OP.LEAKY_RELU, a, (_f32_bits(slope),))

    def sum_rows(self, a):
        T = self._new(a.type, (1, a.ne[1], a.ne[2], a.ne[3]))
        return self._op(T, OP.SUM_ROWS, [a])

    def repeat_4d(self, a, ne0, ne1, ne2, ne3):
        T = self._new(a.type, (ne0, ne1, ne2, ne3))
        return self._op(T, OP.REPEAT, [a])

    def repeat(self, a, b):
        return self.repeat_4d(a, *b.ne)

    def concat(self, a, b, dim):
        ne = list(a.ne)
        ne[dim] += b.ne[dim]
        T = self._new(a.type, ne)
        return self._op(T, OP.CONCAT, [a, b], (dim,))

    def pad_ext(self, a, lp0, rp0, lp1, rp1, lp2, rp2, lp3, rp3):
        T = self._new(a.type, (a.ne[0] + lp0 + rp0, a.ne[1] + lp1 + rp1, a.ne[2] + lp2 + rp2, a.ne[3] + lp3 + rp3))
        return self._op(T, OP.PAD, [a], (lp0, rp0, lp1, rp1, lp2, rp2, lp3, rp3))

    def pad_reflect_1d(self, a, p0, p1):
        T = self._new(a.type, (a.ne[0] + p0 + p1, a.ne[1], a.ne[2], a.ne[3]))
        return self._op(T, OP.PAD_REFLECT_1D, [a], (p0, p1))

    def arange(self, start, stop, step):
        import math
        T = self._new(GGML_TYPE_F32, (int(math.ceil((stop - start) / step)),))
        return self._op(T, OP.ARANGE, [], (_f32_bits(start), _f32_bits(stop), _f32_bits(step)))

    def timestep_embedding(self, ts, dim, max_period):
        T = self._new(GGML_TYPE_F32, (dim + (dim & 1), ts.ne[0]))
        return self._op(T, OP.TIMESTEP_EMBEDDING, [ts], (dim, max_period))

    def conv_transpose_1d(self, kernel, x, s0):
        """ggml_conv_transpose_1d(a, b, s0, p0 = 0, d0 = 1): kernel [K, Cout, Cin], x [L, Cin] -> [(L - 1) * s0 + K, Cout]"""
        T = self._new(GGML_TYPE_F32, ((x.ne[0] - 1) * s0 + kernel.ne[0], kernel.ne[1], 1, 1))
        return self._op(T, OP.CONV_TRANSPOSE_1D, [kernel, x], (s0, 0, 1))

    def unary(self, a, uop):
        T = self._new(a.type, a.ne)
        return self._op(T, OP.UNARY, [a], (uop,))

    def rms_norm(self, a, eps):
        T = self._new(a.type, a.ne)
        return self._op(T, OP.RMS_NORM, [a], (_f32_bits(eps),))

    def norm(self, a, eps):
        """ggml_norm (LayerNorm without the affine part), ggml.c"""
        T = self._new(a.type, a.ne)
        return self._op(T, OP.NORM, [a], (_f32_bits(eps),))

    def im2col(self, kernel, x, s0, s1, p0, p1, d0, d1, is_2d, dst_type):
        """ggml_im2col (ggml.c): kernel [KW, KH, IC, OC] (2-D) / [K, IC, OC] (1-D) gives only the window shape"""
        def out_size(ins, ks, s, p, d):
            return (ins + 2 * p - d * (ks - 1) - 1) // s + 1
        if is_2d:
            OH, OW = out_size(x.ne[1], kernel.ne[1], s1, p1, d1), out_size(x.ne[0], kernel.ne[0], s0, p0, d0)
            ne = (kernel.ne[2] * kernel.ne[1] * kernel.ne[0], OW, OH, x.ne[3])
        else:
            OW = out_size(x.ne[0], kernel.ne[0], s0, p0, d0)
            ne = (kernel.ne[1] * kernel.ne[0], OW, x.ne[2], 1)
        T = self._new(dst_type, ne)
        return self._op(T, OP.IM2COL, [kernel, x], (s0, s1, p0, p1, d0, d1, 1 if is_2d else 0))

    def pool_2d(self, x, op, k0, k1, s0, s1, p0, p1):
        """ggml_pool_2d (ggml.c): op 0 = max, 1 = avg; x [IW, IH, C, N] -> f32 [OW, OH, C, N]"""
        ne = ((x.ne[0] + 2 * p0 - k0) // s0 + 1, (x.ne[1] + 2 * p1 - k1) // s1 + 1, x.ne[2], x.ne[3])
        T = self._new(GGML_TYPE_F32, ne)
        return self._op(T, OP.POOL_2D, [x], (op, k0, k1, s0, s1, p0, p1))

    def pool_1d(self, x, op, k0, s0, p0):
        """ggml_pool_1d (ggml.c): windows along ne[0]"""
        ne = ((x.ne[0] + 2 * p0 - k0) // s0 + 1, x.ne[1], x.ne[2], x.ne[3])
        T = self._new(GGML_TYPE_F32, ne)
        return self._op(T, OP.POOL_1D, [x], (op, k0, s0, p0))

    def conv_1d(self, kernel, x, s0, p0, d0):
        """ggml_conv_1d (ggml.c): im2col in f16, then one MUL_MAT against the flattened f16 kernel -> [OL, OC, N]"""
        col = self.im2col(kernel, x, s0, 0, p0, 0, d0, 0, False, GGML_TYPE_F16)
        r = self.mul_mat(self.reshape(col, col.ne[0], col.ne[2] * col.ne[1]), self.reshape(kernel, kernel.ne[0] * kernel.ne[1], kernel.ne[2]))
        return self.reshape(r, col.ne[1], kernel.ne[2], col.ne[2])

    def mul_mat(self, a, b):
        assert a.ne[0] == b.ne[0] and b.ne[2] % a.ne[2] == 0 and b.ne[3] % a.ne[3] == 0
        T = self._new(GGML_TYPE_F32, (a.ne[1], b.ne[1], b.ne[2], b.ne[3]))
        return self._op(T, OP.MUL_MAT, [a, b])

    def glu_split(self, a, b, glu_op):
        T = self._new(a.type, a.ne)
        return self._op(T, OP.GLU, [a, b], (glu_op, 0))

    def swiglu_split(self, a, b):
        return self.glu_split(a, b, GLU.SWIGLU)

    def glu(self, a, glu_op, swapped=False):
        ne = list(a.ne)
        ne[0] //= 2
        T = self._new(a.type, ne)
        return self._op(T, OP.GLU, [a], (glu_op, int(swapped)))

    def rope_ext(self, a, pos, freq_factors, n_dims, mode, n_ctx_orig, freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow):
        T = self._new(a.type, a.ne)
        params = [0, n_dims, mode, 0, n_ctx_orig] + [_f32_bits(x) for x in (freq_base, freq_scale, ext_factor, attn_factor, beta_fast, beta_slow)] + [0, 0, 0, 0]
        return self._op(T, OP.ROPE, [a, pos, freq_factors], params)

    def soft_max_ext(self, a, mask, scale, max_bias=0.0):
        T = self._new(a.type, a.ne)
        return self._op(T, OP.SOFT_MAX, [a, mask], (_f32_bits(scale), _f32_bits(max_bias)))

    def get_rows(self, a, b):
        T = self._new(GGML_TYPE_I32 if a.type == GGML_TYPE_I32 else GGML_TYPE_F32, (a.ne[0], b.ne[0], b.ne[1], b.ne[2]))
        return self._op(T, OP.GET_ROWS, [a, b])

    def set_rows(self, a, b, c):
        """a: destination table, b: f32 rows, c: i64/i32 row indices -> view of a (ggml_set_rows)."""
        assert a.ne[0] == b.ne[0] and b.ne[1] == c.ne[0]
        T = self._new(a.type, a.ne, view_src=a)
        for i in range(4):
            T.t.nb[i] = a.t.nb[i]
        return self._op(T, OP.SET_ROWS, [b, c])

    def flash_attn_ext(self, q, k, v, mask, scale, max_bias=0.0, logit_softcap=0.0, sinks=None):
        T = self._new(GGML_TYPE_F32, (v.ne[0], q.ne[2], q.ne[1], q.ne[3]))
        self._op(T, OP.FLASH_ATTN_EXT, [q, k, v, mask, sinks], (_f32_bits(scale), _f32_bits(max_bias), _f32_bits(logit_softcap), GGML_PREC_F32))
        return T

    # ---- allocation: every non-view tensor gets its own aligned slot in one device buffer
    def alloc(self, usage=GGML_BACKEND_BUFFER_USAGE_ANY):
        """usage: ggml_backend_buffer_set_usage() of the reference (ggml-backend.cpp:178-186); libllama marks its model buffers
        GGML_BACKEND_BUFFER_USAGE_WEIGHTS (src/llama-model.cpp), which is what lets the backend keep F16 images of them"""
        al = self.be.alignment
        off = 0
        slots = []
        for T in self.tensors:
            if T.t.view_src:
                continue
            off = (off + al - 1) // al * al
            slots.append((T, off))
            off += max(T.nbytes(), 1)
        self.buffer = self.be.alloc_buffer(max(off, 1))
        buf = C.cast(self.buffer, C.POINTER(buffer_t)).contents
        buf.usage = usage
        base = buf.iface.get_base(self.buffer)
        for T, o in slots:
            T.t.buffer = self.buffer
            T.t.data = base + o
        for T in self.tensors:
            if T.t.view_src:
                root = T.t.view_src.contents
                T.t.buffer = root.buffer
                T.t.data = root.data + T.t.view_offs
        if buf.iface.init_tensor:                                     # optional in the ABI (NULL for host buffers)
            for T in self.tensors:
                st = buf.iface.init_tensor(self.buffer, T.ptr)
                assert st == 0
        return self

    def graph(self, nodes=None):
        if nodes is None and getattr(self, "roots", None):
            return self.graph_expand(self.roots)
        return Graph(list(self.nodes if nodes is None else nodes))

    def graph_expand(self, roots):
        """Node order of ggml_build_forward_expand() called on `roots` in turn (reference ggml.c ggml_visit_parents: depth-first
        over src[0..], a node is appended after its sources, visited once) -- the order libllama's graphs reach the backend in."""
        seen, order = set(), []
        by_addr = {C.addressof(T.t): T for T in self.tensors}

        def visit(T):
            key = C.addressof(T.t)
            if key in seen:
                return
            seen.add(key)
            for i in range(10):
                p = T.t.src[i]
                if p:
                    S = by_addr.get(C.addressof(p.contents))
                    if S is not None:
                        visit(S)
            if T.t.op != OP.NONE:
                order.append(T)

        import sys
        lim = sys.getrecursionlimit()
        sys.setrecursionlimit(max(lim, 20000))
        try:
            for r in roots:
                visit(r)
        finally:
            sys.setrecursionlimit(lim)
        return Graph(order)

    def free(self):
        if self.buffer:
            self.be.free_buffer(self.buffer)
            self.buffer = None
