// Test infrastructure: the HIP device vocabulary and a stand-in HIP runtime over the fibre emulator (../simt_emu.h), as an
// include directory: with -I tests/cpp/emu_rt the engine's sources (zpaqfranz_amd/csrc/*.hip) find THESE <hip/hip_runtime.h>,
// <hip/hiprtc.h> and <rocprim/...> and compile for the host unchanged -- kernels and host code.  Device memory is host
// memory, a launch runs every workgroup on the emulator before it returns, streams and events are empty, hiprtc compiles
// the generated source with the host compiler into a shared object.  tests/emu_build.py builds libzpaqhip_emu.so from it.
#pragma once
#include <stdint.h>

#if defined(EMU_EXTERN_STATE)
#define EMU_VAR extern __attribute__((visibility("default")))
#else
#define EMU_VAR inline __attribute__((visibility("default")))
#endif
struct EmuIdx { unsigned x, y, z; };
EMU_VAR EmuIdx blockIdx, gridDim, blockDim;
EMU_VAR const char* g_emu_launch_error;
EMU_VAR unsigned long long g_emu_launches, g_emu_blocks;
