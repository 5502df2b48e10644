"""The run-time compiled kernels' caches (cm_jit.hip) are fed by archive contents, so what an input can cause is
bounded and what is loaded is checked: a compile budget per process, cache files that carry their
source and are used only when it is the one asked for, cache directories only when nobody else can write to them.
hiprtc cross-compiles without a GPU, so all of this runs here."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import ctypes as C, sys
sys.path.insert(0, %r)
from zpaqfranz_amd import engine
L = engine.load()
L.zpq_cm_precompile.argtypes = [C.c_char_p, C.c_uint32]
src, args = engine.make_config(engine.expand_method("x4,3ci1", b""))
h = engine.compile_config(src, args)[0]
print("rc", L.zpq_cm_precompile(h, len(h)))
""" % ROOT


def run(cache, **env):
    e = dict(os.environ, ZPQ_JIT_CACHE=str(cache))
    e.pop("ZPQ_JIT_NOCACHE", None)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=300, env=e)
    assert r.returncode == 0, r.stderr[-1500:]
    return int(r.stdout.split("rc")[-1])


def test_budget_cache_integrity_and_directory_ownership(tmp_path):
    cache = tmp_path / "jit"
    # nothing cached, no compile allowed: refused (the caller then takes the interpreter-driven kernels)
    assert run(cache, ZPQ_JIT_MAX_COMPILES="0") != 0
    assert run(cache) == 0
    files = [f for f in os.listdir(cache) if f.endswith(".hsaco")]
    assert len(files) == 1 and (os.stat(cache).st_mode & 0o077) == 0
    path = os.path.join(cache, files[0])
    blob = open(path, "rb").read()
    assert blob[:8] == b"ZPQJIT2\n"
    # a valid cache file needs no compile
    assert run(cache, ZPQ_JIT_MAX_COMPILES="0") == 0
    # a file whose embedded source is not the one asked for is not used (same name: what a hash collision would look like)
    k = blob.index(b"#define ZN ")
    bad = blob[:k] + b"#define ZX " + blob[k + 11:]
    open(path, "wb").write(bad)
    assert run(cache, ZPQ_JIT_MAX_COMPILES="0") != 0
    open(path, "wb").write(blob)
    assert run(cache, ZPQ_JIT_MAX_COMPILES="0") == 0
    # a file or a directory somebody else could have written is not trusted
    os.chmod(path, 0o666)
    assert run(cache, ZPQ_JIT_MAX_COMPILES="0") != 0
    os.chmod(path, 0o644)
    os.chmod(cache, 0o777)
    assert run(cache, ZPQ_JIT_MAX_COMPILES="0") != 0
    os.chmod(cache, 0o700)
    assert run(cache, ZPQ_JIT_MAX_COMPILES="0") == 0
    # the limit on code objects kept per process
    assert run(cache, ZPQ_JIT_MAX_MODULES="0") != 0
