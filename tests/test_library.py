"""CPU tests of the drop-in boundary: libggml-mi355x.so loads, exports every symbol include/ggml-mi355x.h declares,
registers through the reference's own loader, and refuses to pretend when there is no GPU (no compute here)."""
import ctypes as C
import os
import re
import subprocess

import pytest

from conftest import ROOT


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "ggml-mi355x.h")).read()
    return sorted(set(re.findall(r"^GGML_MI355X_API[^;(]*?\b(\w+)\s*\(", txt, flags=re.M)))


def test_header_symbols_are_exported(pkg):
    syms = declared_symbols()
    assert "ggml_backend_init" in syms and "ggml_backend_score" in syms and len(syms) >= 10
    out = subprocess.check_output(["nm", "-D", "--defined-only", pkg.lib_path()], text=True)
    exported = {l.split()[-1] for l in out.splitlines() if " T " in l}
    assert set(syms) <= exported, set(syms) - exported
    # nothing else leaks from the C++ side except HIP's own registration objects
    extra = {s for s in exported - set(syms)}
    assert not extra, extra


def test_only_weak_host_imports(pkg):
    """The product never hard-links ggml-base or anything under oracle/: the two host imports are weak."""
    out = subprocess.check_output(["nm", "-D", "--undefined-only", pkg.lib_path()], text=True)
    ggml = [l.split() for l in out.splitlines() if "ggml" in l]
    assert all(parts[0] == "w" for parts in ggml), ggml
    assert {p[-1] for p in ggml} == {"ggml_backend_buffer_init", "ggml_log_internal"}
    needed = subprocess.check_output(["readelf", "-d", pkg.lib_path()], text=True)
    assert "oracle" not in needed and "ggml-ref" not in needed


def test_registry_object_without_gpu(pkg):
    lib = pkg.load_library()
    reg = lib.ggml_backend_mi355x_reg()
    assert reg.contents.api_version == 2
    assert reg.contents.iface.get_name(reg) == b"MI355X"
    n = reg.contents.iface.get_device_count(reg)
    assert lib.ggml_backend_score() == (100 if n else 0)
    assert reg.contents.iface.get_proc_address(reg, b"mi355x_set_option")
    assert not reg.contents.iface.get_proc_address(reg, b"no_such_function")
    if n == 0:
        with pytest.raises(RuntimeError):
            pkg.Backend(0)                        # fail loudly: no CPU fallback exists


def test_missing_library_fails_loudly(pkg, tmp_path, monkeypatch):
    import sys
    g = sys.modules["llama_cpp_omni_amd.ggml"]
    monkeypatch.setattr(g, "_LIB", None)
    monkeypatch.setattr(g, "lib_path", lambda: str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        g.load_library()


def test_reference_loader_accepts_the_plugin():
    """The reference's own registry (oracle/_ref build of ggml-backend-reg.cpp) dlopens the plug-in via GGML_BACKEND_PATH."""
    tbo = os.path.join(ROOT, "oracle", "_ref", "test-backend-ops")
    if not os.path.exists(tbo):
        pytest.skip("oracle/_ref not built")
    env = dict(os.environ, GGML_BACKEND_PATH=os.path.join(ROOT, "llama.cpp-omni_amd", "lib", "libggml-mi355x.so"))
    r = subprocess.run([tbo, "support", "-o", "NONE"], env=env, capture_output=True, text=True, timeout=120)
    txt = r.stdout + r.stderr
    assert "failed to" not in txt.lower() or "not supported on this system" in txt
    assert ("MI355X" in txt) or ("not supported on this system" in txt), txt[-2000:]


def test_loader_entry_reports_its_devices_once_per_process(pkg):
    """ggml_backend_init is what ggml-backend-reg.cpp binds; a host that calls ggml_backend_load_all() repeatedly (token2wav-impl.cpp after omni_init) calls it again and
    appends the returned registry's devices each time.  The first call of a process returns the device-bearing registry, later calls a registry of the same name and proc
    addresses with no devices -- run in a process of its own, because the first call is global state."""
    import subprocess, sys, textwrap
    code = textwrap.dedent("""
        import sys
        sys.path.insert(0, %r)
        import conftest
        lib = conftest.load_pkg().load_library()
        full = lib.ggml_backend_mi355x_reg()
        a, b, c = lib.ggml_backend_init(), lib.ggml_backend_init(), lib.ggml_backend_init()
        import ctypes as C
        addr = lambda r: C.cast(r, C.c_void_p).value
        n = full.contents.iface.get_device_count(full)
        assert addr(a) == addr(full), "first call: the registry itself"
        assert addr(b) == addr(c) != addr(full)
        for r in (b, c):
            assert r.contents.api_version == 2 and r.contents.iface.get_name(r) == b"MI355X"
            assert r.contents.iface.get_device_count(r) == 0
            assert r.contents.iface.get_proc_address(r, b"mi355x_set_option")
        assert full.contents.iface.get_device_count(full) == n
        print("ok", n)
    """) % (__import__("os").path.dirname(__import__("os").path.abspath(__file__)),)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().startswith("ok"), (r.stdout, r.stderr[-2000:])
