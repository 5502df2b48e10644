"""Builds libzpaqhip.so (the C-ABI engine, include/zpaqhip.h) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU; the resulting .so is git-ignored but ships to the GPU box with
the gpurun snapshot."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libzpaqhip.so")
SOURCES = ["ctx.hip", "sha.hip", "fragment.hip", "twins.hip", "dedup.hip", "lz77_enc.hip", "lz77_sa.hip", "lz77_dec.hip", "block.hip", "unblock.hip", "cm.hip", "cm_jit.hip", "config.hip", "e8e9.hip", "checksum.hip", "ibwt.hip"]


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, "zpq_internal.h"), os.path.join(CSRC, "cm_spec_src.inc"), os.path.join(CSRC, "lz77_waves.inc"), os.path.join(ROOT, "include", "zpaqhip.h"),
            os.path.join(HERE, "shim", "libzpaq_gpu.cpp"), os.path.join(HERE, "shim", "libzpaq_gpu.h"),
            os.path.join(HERE, "shim", "jidac_gpu.cpp"), os.path.join(HERE, "shim", "jidac_gpu.h"), os.path.join(HERE, "shim", "rccl_gather.cpp"),
            os.path.join(HERE, "shim", "rccl_gather.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return SO
    objs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    procs = []
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC,
               "-c", os.path.join(CSRC, s), "-o", o] + os.environ.get("ZPQ_EXTRA_FLAGS", "").split()
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode(errors="replace"))
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = ["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", SO] + objs + ["-lhiprtc", "-ldl"]
    subprocess.check_call(cmd)
    build_shim()
    if verbose:
        print("built", SO)
    return SO


SHIM_SO = os.path.join(HERE, "libzpaq_gpu.so")


def build_shim():
    """The libzpaq-shaped C++ host layer (zpaqfranz_amd/shim), linked against libzpaqhip.so."""
    # libzpaq_gpu.so leaves libzpaq::error to the application (as libzpaq does); the journaling engine has
    # no such hook and gets its own library so that any host can dlopen it
    for srcs, so in (([os.path.join(HERE, "shim", "libzpaq_gpu.cpp")], SHIM_SO),
                     ([os.path.join(HERE, "shim", "jidac_gpu.cpp")], os.path.join(HERE, "libzpaq_jidac.so"))):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"),
               "-I" + os.path.join(HERE, "shim")] + srcs + ["-L" + HERE, "-lzpaqhip", "-Wl,-rpath,$ORIGIN", "-o", so]
        subprocess.check_call(cmd)
    try:
        build_rccl()
    except (subprocess.CalledProcessError, OSError) as ex:
        # only the process-sharded add (zpqj_add_sharded over zpqr_allgatherv) needs it: a host without librccl still gets the
        # engine, the shim and the journaling library; RcclGather.lib() says what is missing when it is first asked for
        sys.stderr.write("zpaqfranz_amd.build: libzpaq_rccl.so not built (%s); the in-tree RCCL all-gather is unavailable\n" % ex)
    return SHIM_SO


RCCL_SO = os.path.join(HERE, "libzpaq_rccl.so")


def rocm_path():
    """ROCM_PATH, else what hipconfig says, else /opt/rocm"""
    p = os.environ.get("ROCM_PATH")
    if p:
        return p
    try:
        p = subprocess.run(["hipconfig", "--rocmpath"], capture_output=True, text=True, timeout=20).stdout.strip()
    except (OSError, subprocess.SubprocessError):
        p = ""
    return p or "/opt/rocm"


def build_rccl():
    """The all-gather of byte strings over RCCL (shim/rccl_gather.cpp: plain rccl.h, no torch) that zpqj_add_sharded takes as
    its one collective; a library of its own so that nothing else depends on librccl."""
    cmd = ["hipcc", "-O2", "-std=c++17", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(HERE, "shim"),
           os.path.join(HERE, "shim", "rccl_gather.cpp"), "-L" + HERE, "-lzpaqhip", "-L" + os.path.join(rocm_path(), "lib"), "-lrccl",
           "-Wl,-rpath,$ORIGIN", "-o", RCCL_SO]
    subprocess.check_call(cmd)
    return RCCL_SO


def build_shim_driver(out):
    """tests/cpp/shim_driver.cpp: a Jidac-style multi-threaded caller of the shim."""
    cmd = ["g++", "-O2", "-std=c++17", "-pthread", "-I" + os.path.join(HERE, "shim"), os.path.join(ROOT, "tests", "cpp", "shim_driver.cpp"),
           "-L" + HERE, "-lzpaq_gpu", "-lzpaqhip", "-Wl,-rpath," + HERE, "-o", out]
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
