// Test infrastructure (see ../simt_state.h): <hip/hiprtc.h> for the host build.  "Compiling" a program = the source text, with
// its AMD inline assembly replaced by C (the one v_writelane macro of the context-mixing coder), compiled by the host compiler
// against this same include directory into a shared object; every `extern "C" __global__` kernel gets an entry point that
// takes the argument array of hipModuleLaunchKernel.  The "code" handed back is the object's path.
#pragma once
#include <sys/stat.h>
#include <unistd.h>

#include "hip_runtime.h"

#if defined(EMU_ASAN)
#define EMU_ASAN_FLAGS " -fsanitize=address -shared-libasan -fno-omit-frame-pointer -g1 -DEMU_ASAN=1"
#else
#define EMU_ASAN_FLAGS ""
#endif
typedef int hiprtcResult;
enum { HIPRTC_SUCCESS = 0, HIPRTC_ERROR_COMPILATION = 6 };
struct EmuRtcProgram { std::string src, log, path; };
typedef EmuRtcProgram* hiprtcProgram;
static inline const char* hiprtcGetErrorString(hiprtcResult r) { return r == HIPRTC_SUCCESS ? "success" : "host compilation of the generated source failed"; }
static inline hiprtcResult hiprtcCreateProgram(hiprtcProgram* p, const char* src, const char*, int, const char**, const char**) {
  *p = new EmuRtcProgram(); (*p)->src = src; return HIPRTC_SUCCESS;
}
static inline hiprtcResult hiprtcDestroyProgram(hiprtcProgram* p) { delete *p; *p = nullptr; return HIPRTC_SUCCESS; }
static inline void emu_replace_all(std::string& s, const std::string& a, const std::string& b) {
  for (size_t k = 0; (k = s.find(a, k)) != std::string::npos; k += b.size()) s.replace(k, a.size(), b);
}
// entry points: extern "C" void NAME__emu(void** a) { NAME(*(T0*)a[0], ...); } for every extern "C" __global__ ... void NAME(T0 x0, ...)
static inline std::string emu_kernel_entries(const std::string& src) {
  std::string out;
  const std::string key = "extern \"C\" __global__";
  for (size_t k = 0; (k = src.find(key, k)) != std::string::npos; k += key.size()) {
    const size_t v = src.find(" void ", k);
    const size_t open = src.find('(', v + 6);
    // (the parenthesis of __launch_bounds__ comes before " void ")
    const size_t close = src.find(')', open);
    if (v == std::string::npos || open == std::string::npos || close == std::string::npos) continue;
    std::string name = src.substr(v + 6, open - v - 6);
    while (!name.empty() && name.back() == ' ') name.pop_back();
    std::string call, params = src.substr(open + 1, close - open - 1);
    int idx = 0;
    for (size_t a = 0; a < params.size();) {
      size_t b = params.find(',', a);
      if (b == std::string::npos) b = params.size();
      std::string one = params.substr(a, b - a);
      while (!one.empty() && one.back() == ' ') one.pop_back();
      size_t e = one.size();
      while (e > 0 && (isalnum((unsigned char)one[e - 1]) || one[e - 1] == '_')) --e;       // strip the parameter's name
      const std::string type = one.substr(0, e);
      if (!type.empty()) { call += (idx ? ", *(" : "*(") + type + "*)a[" + std::to_string(idx) + "]"; ++idx; }
      a = b + 1;
    }
    out += "extern \"C\" __attribute__((visibility(\"default\"))) void " + name + "__emu(void** a) { " + name + "(" + call + "); }\n";
  }
  return out;
}
static inline unsigned long long emu_fnv64(const std::string& s) { unsigned long long h = 1469598103934665603ull; for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; } return h; }
static inline std::string emu_self_path() { Dl_info i; return dladdr((void*)&emu_fnv64, &i) && i.dli_fname ? i.dli_fname : ""; }
static inline hiprtcResult hiprtcCompileProgram(hiprtcProgram p, int, const char**) {
  std::string src = p->src;
  if (getenv("EMU_TRACE")) fprintf(stderr, "[emu] hiprtc: %.60s ...\n", src.substr(0, src.find("\n#define ZH_LDS") == std::string::npos ? 60 : src.find("\n#define ZH_LDS")).c_str());
  // the one piece of AMD assembly in generated source: v_writelane (cm_spec_src.inc ZWL) -> the same effect in C
  {
    const size_t k = src.find("#define ZWL(");
    if (k != std::string::npos) {
      const size_t e = src.find('\n', k);
      src.replace(k, e - k, "#define ZWL(v, L, p) { const int zwl_ = __builtin_amdgcn_readfirstlane((int)(v)); if ((int)(threadIdx.x & 63) == (L)) p = (decltype(p))zwl_; }");
    }
  }
  src += "\n" + emu_kernel_entries(src);
  const char* tmp = getenv("ZPQ_EMU_JIT_DIR");
  const std::string dir = tmp && *tmp ? tmp : "/tmp/zpq_emu_jit";
  mkdir(dir.c_str(), 0777);
  char name[64];
  const std::string self = emu_self_path();
  snprintf(name, sizeof name, "/k_%016llx", emu_fnv64(src + self));
  p->path = dir + name + ".so";
  if (access(p->path.c_str(), R_OK) == 0) return HIPRTC_SUCCESS;
  const std::string cpp = dir + name + "." + std::to_string((long)getpid()) + ".cpp", tmpso = p->path + "." + std::to_string((long)getpid());
  FILE* f = fopen(cpp.c_str(), "w");
  if (!f) { p->log = "cannot write " + cpp; return HIPRTC_ERROR_COMPILATION; }
  fputs(src.c_str(), f); fclose(f);
  const std::string cmd = std::string(EMU_HOST_CXX) + " -O1 -std=c++17 -fPIC -shared -w -DEMU_EXTERN_STATE" EMU_ASAN_FLAGS " -I" + EMU_RT_DIR + " -include hip/hip_runtime.h " + cpp +
                          " -o " + tmpso + " " + self + " -Wl,-rpath," + self.substr(0, self.rfind('/')) + " > " + cpp + ".log 2>&1";
  const int rc = system(cmd.c_str());
  if (rc != 0) {
    if (FILE* l = fopen((cpp + ".log").c_str(), "r")) { char buf[4096]; const size_t n = fread(buf, 1, sizeof buf - 1, l); buf[n] = 0; p->log = buf; fclose(l); }
    return HIPRTC_ERROR_COMPILATION;
  }
  rename(tmpso.c_str(), p->path.c_str());
  unlink(cpp.c_str()); unlink((cpp + ".log").c_str());
  return HIPRTC_SUCCESS;
}
static inline hiprtcResult hiprtcVersion(int* major, int* minor) { *major = 0; *minor = 0; return HIPRTC_SUCCESS; }
static inline hiprtcResult hiprtcGetProgramLogSize(hiprtcProgram p, size_t* n) { *n = p->log.size() + 1; return HIPRTC_SUCCESS; }
static inline hiprtcResult hiprtcGetProgramLog(hiprtcProgram p, char* out) { memcpy(out, p->log.c_str(), p->log.size() + 1); return HIPRTC_SUCCESS; }
static inline hiprtcResult hiprtcGetCodeSize(hiprtcProgram p, size_t* n) { *n = p->path.size() + 1; return HIPRTC_SUCCESS; }
static inline hiprtcResult hiprtcGetCode(hiprtcProgram p, char* out) { memcpy(out, p->path.c_str(), p->path.size() + 1); return HIPRTC_SUCCESS; }
