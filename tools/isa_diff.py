"""Which kernels of a source file changed?  Compiles zpaqfranz_amd/csrc/<file> at a git revision and in the working tree to
gfx950 assembly (hipcc -S --cuda-device-only), normalises labels and compares kernel by kernel (demangled names).
usage: python tools/isa_diff.py lz77_enc.hip [REV=HEAD]   -> lists identical / changed / new kernels"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm(csrc, name, out):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), "-I" + csrc,
                           os.path.join(csrc, name), "-o", out])
    kernels, cur, body = {}, None, []
    for ln in open(out):
        m = re.match(r"^(_Z\w+|\w+):\s*(;.*)?$", ln)
        if m and not ln.startswith(".L") and not ln.startswith("\t"):
            cur = m.group(1); body = []; kernels[cur] = body
            continue
        if ln.startswith("\t.end_amdhsa_kernel") or ln.startswith("\t.section") or ln.startswith(".Lfunc_end"):
            cur = None
        if cur is not None:
            t = ln.split(";")[0].strip()
            if not t or t.startswith("."):
                if not t.startswith(".LBB"):
                    continue
            t = re.sub(r"\.LBB\d+_(\d+)", r".LBB_\1", t)
            t = re.sub(r"_Z\w+", "SYM", t)
            body.append(t)
    return kernels


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    name = sys.argv[1]
    rev = sys.argv[2] if len(sys.argv) > 2 else "HEAD"
    with tempfile.TemporaryDirectory() as td:
        old = os.path.join(td, "old")
        os.makedirs(old)
        for f in subprocess.check_output(["git", "ls-tree", "--name-only", rev, "zpaqfranz_amd/csrc/"], cwd=ROOT, text=True).split():
            open(os.path.join(old, os.path.basename(f)), "wb").write(subprocess.check_output(["git", "show", rev + ":" + f], cwd=ROOT))
        a = asm(old, name, os.path.join(td, "a.s"))
        b = asm(os.path.join(ROOT, "zpaqfranz_amd", "csrc"), name, os.path.join(td, "b.s"))
    dm = demangle(sorted(set(a) | set(b)))
    same = [k for k in a if k in b and a[k] == b[k]]
    diff = [k for k in a if k in b and a[k] != b[k]]
    print("identical: %d, changed: %d, new: %d, gone: %d" % (len(same), len(diff), len(set(b) - set(a)), len(set(a) - set(b))))
    import difflib
    for k in diff:
        print("  changed:", dm[k][:160], "(%d -> %d instructions)" % (len(a[k]), len(b[k])))
        if os.environ.get("ISA_DIFF_SHOW") and os.environ["ISA_DIFF_SHOW"] in dm[k]:
            print("\n".join(list(difflib.unified_diff(a[k], b[k], lineterm="", n=1))[:60]))
    for k in sorted(set(b) - set(a)):
        print("  new:    ", dm[k][:160])
    for k in sorted(set(a) - set(b)):
        print("  gone:   ", dm[k][:160])


if __name__ == "__main__":
    main()
