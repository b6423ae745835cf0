"""llama.cpp-omni_amd -- MI355X-native ggml backend (libggml-mi355x.so) and its Python host mirror.

The product is the shared library under ``lib/`` (C++ host code + hand-written gfx950 HIP kernels, built by
``csrc/Makefile``).  It plugs into the reference unchanged through the ggml backend C-ABI
(``ggml_backend_init`` / vtables, reference ``ggml/src/ggml-backend-impl.h:17-210``).

This package is the *host-side mirror* used by the tests and by ``bench.py`` on a box that has no reference
checkout: it binds the very same C-ABI with ``ctypes`` and re-creates the small part of the ggml host API
that is needed to describe graphs (``ggml_new_tensor``, ``ggml_mul_mat``, ``ggml_rope_ext`` ... with the
reference's names, argument meaning and error behaviour, reference ``ggml/src/ggml.c``), so parity tests read
like the reference's own ``tests/test-backend-ops.cpp`` cases.  No arithmetic happens in Python and there is
no CPU fallback: if the HIP library is missing, importing ``backend()`` raises.

Because the directory name is not a valid Python identifier, import it with
``importlib`` (see ``tests/conftest.py``: ``load_pkg()``).
"""
from .ggml import *          # noqa: F401,F403
from .ggml import __all__    # noqa: F401
