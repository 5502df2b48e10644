"""Runs a script or a module with the engine of tools/_variants/<name> (tools/make_variant.sh) in place of the tree's own:
    python tools/run_variant.py <name> bench.py --no-cpu-baseline --workload dup8_m1
    python tools/run_variant.py <name> -m pytest tests/test_gpu_parity.py -m gpu -k lz77 -q
The variant's package directory goes to the front of sys.path; everything else (tests, oracle, bench) is the tree's."""
import os
import runpy
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
name = sys.argv[1]
vdir = os.path.join(root, "tools", "_variants", name)
if not os.path.exists(os.path.join(vdir, "zpaqfranz_amd", "libzpaqhip.so")):
    sys.exit("no such variant (tools/make_variant.sh %s ...): %s" % (name, vdir))
sys.path.insert(0, vdir)
import zpaqfranz_amd.engine as _e      # noqa: E402  (bound before anything of the tree's can be)
assert _e.lib_path().startswith(vdir), _e.lib_path()
print("[variant %s] %s" % (name, _e.lib_path()), file=sys.stderr)
if sys.argv[2] == "-m":
    sys.argv = sys.argv[3:]
    runpy.run_module(sys.argv[0], run_name="__main__", alter_sys=True)
else:
    sys.argv = sys.argv[2:]
    sys.path.insert(1, root)
    runpy.run_path(os.path.join(root, sys.argv[0]) if not os.path.isabs(sys.argv[0]) else sys.argv[0], run_name="__main__")
