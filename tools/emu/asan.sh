#!/bin/bash
# GPU tests against the emulated engine built with AddressSanitizer: reads and writes outside device allocations, which a GPU
# tolerates silently, stop the process here.  usage: tools/emu/asan.sh tests/test_gpu_twins.py [-k ...]
R=$(cd "$(dirname "$0")/../.." && pwd)
cd "$R"
export ZPQ_EMU_ASAN=1 ZPQ_TEST_EMU=1 PYTHONPATH="$R/tests/emu_site${PYTHONPATH:+:$PYTHONPATH}"
python tests/emu_build.py || exit 1
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
export ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0:halt_on_error=1:abort_on_error=0:allocator_may_return_null=1
LD_PRELOAD="$RT" python -m pytest "$@" -m gpu -q -x -p no:cacheprovider
