"""Test infrastructure: libzpaqhip_emu.so -- the engine's own sources (zpaqfranz_amd/csrc/*.hip, host code AND kernels) compiled
for the host against the stand-in HIP runtime of tests/cpp/emu_rt (device memory = host memory, a kernel launch = every
workgroup on the fibre emulator tests/cpp/simt_emu.h, hiprtc = the host compiler).  The sources are taken as they are except
for AMD inline assembly, which no host assembler takes: `s_waitcnt vmcnt(0)` becomes the emulator's "every lane's memory
operations up to here" rendezvous, the empty register-pinning statement of the SHA-1 rounds is dropped, the four three-operand
instructions sha.hip spells out (v_add3, v_bitop3 x 2, v_bfi) become the C expressions they stand for.  Used by the CPU tests
that run the GPU test files on the emulator; never by the product (which loads libzpaqhip.so and fails without a GPU)."""
import hashlib
import os
import re
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zpaqfranz_amd", "csrc")
RT = os.path.join(ROOT, "tests", "cpp", "emu_rt")
ASAN = os.environ.get("ZPQ_EMU_ASAN") == "1"       # AddressSanitizer build (tools/emu/asan.sh): every device allocation gets red zones
COUNT = os.environ.get("ZPQ_EMU_COUNT") == "1"     # event counters of the LZ77 walk compiled in (tools/emu/lz_walk_counts.py)
OUT = os.path.join(ROOT, "tests", "_emu_asan" if ASAN else "_emu_count" if COUNT else "_emu")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SO = os.path.join(OUT, "libzpaqhip.so")        # (the product's file names, in tests/_emu: the shim libraries link by name)


def available():
    return os.path.exists(CLANG)


def activate():
    """points this process's loader (zpaqfranz_amd.engine / .build) at the emulated libraries"""
    build()
    sys.path.insert(0, ROOT)
    from zpaqfranz_amd import build as product
    from zpaqfranz_amd import engine
    engine._HERE = OUT
    product.HERE = OUT
    product.SHIM_SO = os.path.join(OUT, "libzpaq_gpu.so")
    os.environ.setdefault("ZPQ_JIT_NOCACHE", "1")


def _translate(text):
    text = text.replace('asm volatile("s_waitcnt vmcnt(0)" ::: "memory")', "EMU_WAIT_VMCNT0")
    text = re.sub(r'asm volatile\(""\s*:[^;]*\);', ";", text)
    # the four three-operand instructions sha.hip spells out, as the C they stand for
    text = re.sub(r'asm\("v_add3_u32 [^;]*;', "r = a + b + c;", text)
    text = re.sub(r'asm\("v_bitop3_b32 [^;]*bitop3:0x96[^;]*;', "r = a ^ b ^ c;", text)
    text = re.sub(r'asm\("v_bitop3_b32 [^;]*bitop3:0xe8[^;]*;', "r = (a & b) | (a & c) | (b & c);", text)
    text = re.sub(r'asm\("v_bfi_b32 [^;]*;', "r = (m & x) | (~m & y);", text)
    return text


def _flags():
    san = ["-fsanitize=address", "-shared-libasan", "-fno-omit-frame-pointer", "-g1", "-DEMU_ASAN=1"] if ASAN else ["-g0"]
    if COUNT:
        san = san + ["-DZPQ_LZ_COUNT=1"]
    return san + ["-O1", "-std=c++17", "-fPIC", "-w", "-x", "c++", "-I" + RT, "-I" + CSRC, "-I" + os.path.join(ROOT, "include"),
            '-DEMU_HOST_CXX="%s"' % CLANG, '-DEMU_RT_DIR="%s"' % RT]


def _stamp(paths):
    h = hashlib.sha1()
    for p in sorted(paths):
        h.update(p.encode())
        h.update(open(p, "rb").read())
    return h.hexdigest()


def build(verbose=False):
    """-> path of libzpaqhip_emu.so (rebuilt when a source, an include or the runtime changed)"""
    sys.path.insert(0, ROOT)
    from zpaqfranz_amd import build as product
    os.makedirs(OUT, exist_ok=True)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    deps += [os.path.join(dp, f) for dp, _, fs in os.walk(RT) for f in fs] + [os.path.join(ROOT, "tests", "cpp", "simt_emu.h"), os.path.join(ROOT, "include", "zpaqhip.h"), __file__]
    common = _stamp(deps)
    jobs = []
    objs = []
    for s in product.SOURCES:
        src = os.path.join(CSRC, s)
        key = hashlib.sha1((common + open(src).read()).encode()).hexdigest()[:16]
        obj = os.path.join(OUT, s + "." + key + ".o")
        objs.append(obj)
        if not os.path.exists(obj):
            jobs.append((s, src, obj))

    def one(job):
        s, src, obj = job
        for old in os.listdir(OUT):
            if old.startswith(s + ".") and old.endswith(".o"):
                os.unlink(os.path.join(OUT, old))
        tr = os.path.join(OUT, s + ".cpp")
        with open(tr, "w") as f:
            f.write('#line 1 "%s"\n' % src + _translate(open(src).read()))
        r = subprocess.run([CLANG] + _flags() + ["-c", tr, "-o", obj + ".tmp"], capture_output=True, text=True)
        if r.returncode:
            return s + ":\n" + r.stderr[-6000:]
        os.rename(obj + ".tmp", obj)
        return None

    with ThreadPoolExecutor(max_workers=os.cpu_count() or 2) as ex:
        errs = [e for e in ex.map(one, jobs) if e]
    if errs:
        raise RuntimeError("emulated build failed:\n" + "\n".join(errs))
    if jobs or not os.path.exists(SO):
        subprocess.check_call([CLANG, "-shared", "-o", SO + ".tmp"] + objs + ["-ldl", "-lpthread"] + (["-fsanitize=address", "-shared-libasan"] if ASAN else []))
        os.rename(SO + ".tmp", SO)
    # the host layers above the C ABI (zpaqfranz_amd/shim: plain C++), linked against the emulated engine
    shim = os.path.join(ROOT, "zpaqfranz_amd", "shim")
    if not os.path.exists(os.path.join(OUT, "shim")):
        os.symlink(shim, os.path.join(OUT, "shim"))
    for src, so in (("libzpaq_gpu.cpp", "libzpaq_gpu.so"), ("jidac_gpu.cpp", "libzpaq_jidac.so")):
        dst = os.path.join(OUT, so)
        newest = max(os.path.getmtime(os.path.join(shim, f)) for f in os.listdir(shim))
        if jobs or not os.path.exists(dst) or os.path.getmtime(dst) < newest:
            subprocess.check_call(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"), "-I" + shim,
                                   os.path.join(shim, src), "-L" + OUT, "-lzpaqhip", "-Wl,-rpath,$ORIGIN", "-o", dst + ".tmp"])
            os.rename(dst + ".tmp", dst)
    if verbose:
        print("built", SO, "(%d sources recompiled)" % len(jobs))
    return SO


if __name__ == "__main__":
    build(verbose=True)
