// Run-time specialisation of the context-mixing coder (rows a12, a14): for one block header -- component list +
// HCOMP bytecode -- generate HIP source (cm_spec_src.inc with a prelude of lane masks / X-macro lists and the
// HCOMP program translated to straight-line code), compile it with hiprtc for gfx950 and keep the module.
// This is the engine's counterpart of libzpaq's two x86 JITs (ZPAQL::assemble ZSFX/libzpaq.cpp:2709-3488,
// Predictor::assemble_p :3489-4261); like them it changes speed only: the semantics are the interpreters'
// (ZPAQL::run0 :1033-1254, Predictor::predict0/update0 :1846-2058), and tests compare with those.
// Code objects are cached per process (keyed by the generated source text itself) and on disk next to the library
// (zpaqfranz_amd/jit_cache, or $ZPQ_JIT_CACHE; a file carries its source and is used only if that is the one asked for,
// the directory only if nobody but its owner can write to it), so that a header this library generates itself is
// compiled once per installation.  What an archive brings -- foreign headers, post-processor programs -- is compiled
// within a per-process budget, never written to disk, and beyond the budget coded by the interpreter-driven kernels.
#include <dlfcn.h>
#include <hip/hiprtc.h>
#include <sys/stat.h>
#include <unistd.h>

#include <map>
#include <mutex>

#include "zpq_internal.h"

namespace {

const char* kSpecSrc =
#include "cm_spec_src.inc"
    ;

enum { NONE = 0, CONS, CM, ICM, MATCH, AVG, MIX2, MIX, ISSE, SSE };

std::string itos(long long v) { char b[32]; snprintf(b, sizeof b, "%lld", v); return b; }
std::string hex64(u64 v) { char b[32]; snprintf(b, sizeof b, "0x%llxull", (unsigned long long)v); return b; }

// ZPAQL bytecode -> straight-line device code.  Every jump target of ZPAQL is static (JT/JF/JMP are relative to the
// instruction, LJ is absolute), so each reachable instruction start becomes a label; a jump into the middle of another
// instruction is decoded from there like the interpreter would (the worklist follows every target).
std::string gen_zpaql(const std::vector<u8>& code, const char* fname, bool is_pcomp) {
  const u32 n = (u32)code.size();
  auto at = [&](u32 i) -> u32 { return i < n ? code[i] : 0u; };
  std::map<u32, std::string> stmt;      // pc -> statement(s), ending in a goto
  std::vector<u32> work;
  work.push_back(0);
  const std::string Mb = "M[b & ZMMASK]", Mc = "M[c & ZMMASK]", Hd = "H[d & ZHMASK]";
  auto go = [&](u32 from, u32 target) -> std::string {
    if (target >= n) return "goto Lerr;";
    work.push_back(target);
    // a loop needs a jump to a lower (or the same) address: those are counted, so that no program spins forever
    if (target <= from) return "{ if (++guard > zlim) goto Llim; goto L" + itos(target) + "; }";
    return "goto L" + itos(target) + ";";
  };
  while (!work.empty()) {
    const u32 pc = work.back(); work.pop_back();
    if (pc >= n || stmt.count(pc)) continue;
    const u32 op = code[pc];
    const u32 len = op == 255 ? 3 : (op & 7) == 7 ? 2 : 1;
    const u32 nxt = pc + len;
    const u32 N = at(pc + 1);
    std::string s;
    bool falls = true;
    if (op == 56) { s = "goto Lend;"; falls = false; }
    else if (op == 255) { s = go(pc, N + 256 * at(pc + 2)); falls = false; }
    else if (op == 39 || op == 47 || op == 63) {
      const u32 tgt = nxt + (((N + 128) & 255) - 128);          // may wrap below 0: then >= n, an error
      if (op == 63) { s = go(pc, tgt); falls = false; }
      else s = std::string(op == 39 ? "if (f) " : "if (!f) ") + go(pc, tgt);
    } else if (op >= 64 && op < 240 && (op < 120 || op >= 128)) {
      const u32 sel = op & 7, grp = op >> 3;
      static const char* srcs[7] = {"a", "b", "c", "d", nullptr, nullptr, nullptr};
      std::string v;
      if (sel < 4) v = srcs[sel];
      else if (sel == 4) v = "(u32)" + Mb;
      else if (sel == 5) v = "(u32)" + Mc;
      else if (sel == 6) v = Hd;
      else v = itos(N) + "u";
      switch (grp) {
        case 8: s = "a = " + v + ";"; break;
        case 9: s = "b = " + v + ";"; break;
        case 10: s = "c = " + v + ";"; break;
        case 11: s = "d = " + v + ";"; break;
        case 12: s = Mb + " = (u8)(" + v + ");"; break;
        case 13: s = Mc + " = (u8)(" + v + ");"; break;
        case 14: s = Hd + " = " + v + ";"; break;
        case 16: s = "a += " + v + ";"; break;
        case 17: s = "a -= " + v + ";"; break;
        case 18: s = "a *= " + v + ";"; break;
        case 19: s = "t = " + v + "; a = t ? a / t : 0u;"; break;
        case 20: s = "t = " + v + "; a = t ? a % t : 0u;"; break;
        case 21: s = "a &= " + v + ";"; break;
        case 22: s = "a &= ~(" + v + ");"; break;
        case 23: s = "a |= " + v + ";"; break;
        case 24: s = "a ^= " + v + ";"; break;
        case 25: s = "a <<= ((" + v + ") & 31);"; break;
        case 26: s = "a >>= ((" + v + ") & 31);"; break;
        case 27: s = "f = a == " + v + ";"; break;
        case 28: s = "f = a < " + v + ";"; break;
        case 29: s = "f = a > " + v + ";"; break;
        default: s = "goto Lerr;"; falls = false; break;
      }
    } else {
      auto reg = [&](u32 g) -> std::string { static const char* r[4] = {"a", "b", "c", "d"}; return r[g]; };
      const u32 g = op >> 3, k = op & 7;
      if (op < 56 && g < 4 && k >= 1 && k <= 4) {                 // X++ X-- X! X=0 on A B C D
        const std::string x = reg(g);
        s = k == 1 ? "++" + x + ";" : k == 2 ? "--" + x + ";" : k == 3 ? x + " = ~" + x + ";" : x + " = 0;";
      } else if (op < 56 && g >= 1 && g < 4 && k == 0) {          // X<>A
        const std::string x = reg(g);
        s = "t = a; a = " + x + "; " + x + " = t;";
      } else if (op < 32 && k == 7) {                             // X=R N
        s = reg(g) + " = R[" + itos(N) + "];";
      } else if (op == 32 || op == 40) {                          // *B<>A, *C<>A: the low byte of A only
        const std::string x = op == 32 ? Mb : Mc;
        s = "t = " + x + "; " + x + " = (u8)a; a = (a & 0xffffff00u) | t;";
      } else if ((g == 4 || g == 5) && k >= 1 && k <= 4) {
        const std::string x = g == 4 ? Mb : Mc;
        s = k == 1 ? "++" + x + ";" : k == 2 ? "--" + x + ";" : k == 3 ? x + " = (u8)~" + x + ";" : x + " = 0;";
      } else if (op == 48) s = "t = " + Hd + "; " + Hd + " = a; a = t;";
      else if (g == 6 && k >= 1 && k <= 4) s = k == 1 ? "++" + Hd + ";" : k == 2 ? "--" + Hd + ";" : k == 3 ? Hd + " = ~" + Hd + ";" : Hd + " = 0;";
      else if (op == 55) s = "R[" + itos(N) + "] = a;";
      else if (op == 57) s = is_pcomp ? "{ if (zop < zcap) zout[zop] = (u8)a; ++zop; }" : ";";   // OUT: HCOMP has no output stream
      else if (op == 59) s = "a = (a + (u32)" + Mb + " + 512u) * 773u;";
      else if (op == 60) s = Hd + " = (" + Hd + " + a + 512u) * 773u;";
      else { s = "goto Lerr;"; falls = false; }
    }
    if (falls) s += " " + go(pc, nxt);
    stmt[pc] = s;
  }
  std::string out = std::string("ZDEV void ") + fname + "(const u32 input, ZVm& z, g_u8* const M, g_u32* const R, const zh_ptr H" +
                    (is_pcomp ? ", g_u8* const zout, const u32 zcap, u32& zop" : "") + ") {\n";
  // Backward jumps are counted so that no program spins forever (the reference's ZPAQL::run0 has no limit at all,
  // ZSFX/libzpaq.cpp:1033-1254): a post-processor over the whole segment (z.g, limit z.lim); HCOMP per call (one byte) with
  // ZGUARD jumps free per call and, beyond those, a CREDIT for the whole block (z.credit, ZCREDIT at the start) -- a valid
  // program that initialises H or M in a long loop on its first byte draws on the credit once and is coded; a program that
  // never ends has used both up after ZGUARD + ZCREDIT jumps (about two seconds of one lane) and the block is refused with
  // err = 2 (ZPQ_ERR_LIMIT), not mistaken for a malformed one (err = 1, ZPQ_ERR_FORMAT).
  out += std::string("  u32 a = input, b = z.b, c = z.c, d = z.d, f = z.f, t = 0, guard = ") + (is_pcomp ? "z.g" : "0") + "; (void)t; (void)guard;\n";
  out += is_pcomp ? "  const u32 zlim = ZGUARD; (void)zlim;\n" : "  const u32 zlim = (u32)(ZGUARD) + z.credit; (void)zlim;\n";
  out += n ? "  goto L0;\n" : "  goto Lerr;\n";
  for (auto& kv : stmt) out += "L" + itos(kv.first) + ": " + kv.second + "\n";
  out += "Llim: z.err = 2; goto Lend;\nLerr: z.err = 1;\n";
  out += std::string("Lend: z.a = a; z.b = b; z.c = c; z.d = d; z.f = f;") +
         (is_pcomp ? " z.g = guard;" : " if (guard > (u32)(ZGUARD)) { const u32 used_ = guard - (u32)(ZGUARD); z.credit = used_ > z.credit ? 0u : z.credit - used_; }") + "\n}\n";
  return out;
}

u64 fnv64(const std::string& s, u64 h = 1469598103934665603ull) {
  for (unsigned char c : s) { h ^= c; h *= 1099511628211ull; }
  return h;
}

int env_int(const char* name, int def) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : def;
}

// A cache directory is used only if it belongs to this user (or root) and nobody else can write to it: a code object
// read from it is executed on the GPU.  No fallback under /tmp: without a safe directory there is no on-disk cache.
bool dir_is_safe(const std::string& dir, bool create) {
  struct stat st;
  if (lstat(dir.c_str(), &st) != 0) {
    if (!create || mkdir(dir.c_str(), 0700) != 0 || lstat(dir.c_str(), &st) != 0) return false;
  }
  if (!S_ISDIR(st.st_mode)) return false;
  if (st.st_uid != geteuid() && st.st_uid != 0) return false;
  return (st.st_mode & 022) == 0;
}

std::string cache_dir() {
  if (getenv("ZPQ_JIT_NOCACHE")) return "";
  std::string dir;
  const char* e = getenv("ZPQ_JIT_CACHE");
  if (e && *e) dir = e;
  else {
    Dl_info di;
    if (!dladdr((const void*)&env_int, &di) || !di.dli_fname) return "";
    const std::string p = di.dli_fname;
    const size_t k = p.rfind('/');
    dir = (k == std::string::npos ? std::string(".") : p.substr(0, k)) + "/jit_cache";
  }
  return dir_is_safe(dir, true) ? dir : "";
}

// Cache file = magic, length of the source, the source text itself, the code object: a file is used only if its source
// is byte for byte the one asked for (the name is a hash; names can collide, contents cannot), and only if it belongs to
// this user or root.
const char kMagic[8] = {'Z', 'P', 'Q', 'J', 'I', 'T', '2', '\n'};

bool read_cached(const std::string& path, const std::string& src, std::vector<char>& code) {
  struct stat st;
  if (lstat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode) || (st.st_uid != geteuid() && st.st_uid != 0) || (st.st_mode & 022)) return false;
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return false;
  bool ok = false;
  char magic[8];
  u64 n = 0;
  std::string have;
  if (fread(magic, 1, 8, f) == 8 && !memcmp(magic, kMagic, 8) && fread(&n, 8, 1, f) == 1 && n == src.size()) {
    have.resize(n);
    const u64 rest = (u64)st.st_size - 16 - n;
    if (fread(&have[0], 1, n, f) == n && have == src && (u64)st.st_size > 16 + n) {
      code.resize(rest);
      ok = fread(code.data(), 1, rest, f) == rest;
    }
  }
  fclose(f);
  return ok;
}

void write_cached(const std::string& path, const std::string& src, const std::vector<char>& code) {
  const std::string tmp = path + ".tmp" + itos((long long)getpid());
  FILE* f = fopen(tmp.c_str(), "wb");
  if (!f) return;
  const u64 n = src.size();
  const bool ok = fwrite(kMagic, 1, 8, f) == 8 && fwrite(&n, 8, 1, f) == 1 && fwrite(src.data(), 1, n, f) == n &&
                  fwrite(code.data(), 1, code.size(), f) == code.size();
  fclose(f);
  if (ok) { (void)chmod(tmp.c_str(), 0644); (void)rename(tmp.c_str(), path.c_str()); } else (void)unlink(tmp.c_str());
}

std::string cache_name(const std::string& src) {
  // what the file holds also depends on the compiler: its version is part of the name
  int maj = 0, min = 0;
  (void)hiprtcVersion(&maj, &min);
  char name[96];
  snprintf(name, sizeof name, "/cm_%016llx%016llx_%d_%d.hsaco", (unsigned long long)fnv64(src),
           (unsigned long long)fnv64(src, 0x9e3779b97f4a7c15ull ^ src.size()), maj, min);
  return name;
}

struct Compiled { std::vector<char> code; };
std::mutex g_mu;                                        // the maps below
std::mutex g_compile_mu;                                // one hiprtc compile at a time; g_mu is NOT held meanwhile
std::map<std::string, Compiled> g_code;                 // source text -> code object (any device: all gfx950)
std::map<std::pair<int, std::string>, zpq_cm_spec*> g_mods;    // (device, source text) -> loaded module
int g_fresh = 0;                                        // compiles this process has paid for (cache hits do not count)
size_t modules_loaded();

}  // namespace

struct zpq_cm_spec {
  hipModule_t mod;
  hipFunction_t enc, dec;
  u32 waves;          // ZW the module was compiled for (workgroup = waves * 64 threads at most)
  bool h_lds;
};

// waves per workgroup the kernels are compiled for: 16 (128 registers per lane); 8 where more registers are needed -- the
// one-bit-ahead values (ZPQ_CM_SPEC)
static u32 spec_waves() {
  if (const char* e = getenv("ZPQ_CM_WAVES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 8 || v == 16) return (u32)v; }
  return getenv("ZPQ_CM_SPEC") ? 8 : 16;
}

// The generated source for one header (also used by the build-time cache warmer and the tests).
int zpq_cm_spec_source(const zpq_cm_header& P, std::string* src, std::string* why) {
  if (P.n < 1 || P.n > 64) { *why = "more than 64 components"; return ZPQ_ERR_METHOD; }
  u64 mask[10] = {0};
  std::string mixes, sses, chain, far_inputs, mix2_updates;
  int nmix = 0, nsse = 0;
  // ISSEs fed by their left neighbour are evaluated as a group (one DPP shift per link); anything else that depends
  // on earlier components closes the group first
  u64 group = 0; u32 depth_of[64] = {0}, group_depth = 0;
  auto flush_group = [&]() {
    if (group) chain += "Z_ISSE_SYS(" + hex64(group) + "," + itos(group_depth) + ") ";
    group = 0; group_depth = 0;
  };
  for (u32 i = 0; i < P.n; ++i) {
    const std::vector<u8>& c = P.comps[i];
    const u32 t = c[0];
    if (t < 1 || t > 9) { *why = "invalid component"; return ZPQ_ERR_FORMAT; }
    mask[t] |= 1ull << i;
    const std::string I = itos(i);
    switch (t) {
      case ISSE:
        if (i > 0 && c[2] == i - 1) {
          depth_of[i] = (group >> (i - 1) & 1) ? depth_of[i - 1] + 1 : 1;
          group |= 1ull << i;
          if (depth_of[i] > group_depth) group_depth = depth_of[i];
        } else {
          flush_group();
          chain += "Z_ISSE(" + I + "," + itos(c[2]) + ") ";
          far_inputs += "{ const int q_ = rl(p, " + itos(c[2]) + "); ZWL(q_, " + I + ", pin); } ";
        }
        break;
      case AVG: flush_group(); chain += "Z_AVG(" + I + "," + itos(c[1]) + "," + itos(c[2]) + "," + itos(c[3]) + ") "; break;
      case MIX2:
        flush_group();
        chain += "Z_MIX2(" + I + "," + itos(c[2]) + "," + itos(c[3]) + ") ";
        mix2_updates += "Z_MIX2_UPD(" + I + "," + itos(c[2]) + "," + itos(c[3]) + "," + itos(c[4]) + ") ";
        break;
      case MIX:
        flush_group();
        mixes += "X(" + itos(nmix) + "," + I + "," + itos(c[2]) + "," + itos(c[3]) + "," + itos(c[4]) + ") ";
        chain += "Z_MIX(" + itos(nmix) + "," + I + "," + itos(c[2]) + "," + itos(c[3]) + ") ";
        ++nmix;
        break;
      case SSE:
        flush_group();
        sses += "X(" + itos(nsse) + "," + I + "," + itos(c[2]) + ") ";
        chain += "Z_SSE(" + itos(nsse) + "," + I + "," + itos(c[2]) + ") ";
        ++nsse;
        break;
      default: break;
    }
  }
  flush_group();
  if (nmix > 8 || nsse > 4) { *why = "more mixers / SSE stages than the wave coder keeps in registers"; return ZPQ_ERR_METHOD; }
  const bool h_lds = P.hh <= 10;
  // waves per workgroup: the tables (86 KiB) are shared, H[] is per block; one workgroup per compute unit
  u32 waves = spec_waves();
  while (waves > 1 && 88064u + (h_lds ? waves * (4u << P.hh) : 0u) > 160u * 1024u - 1024u) waves >>= 1;
  std::string s;
  s += "#define ZN " + itos(P.n) + "\n#define ZW " + itos(waves) + "\n";
  s += "#define ZH_LDS " + itos(h_lds ? 1 : 0) + "\n#define ZHMASK " + itos((1u << P.hh) - 1) + "u\n#define ZMMASK " + itos((1u << P.hm) - 1) + "u\n";
  {
    // backward jumps HCOMP may take per byte for free, and the block's one credit beyond them (gen_zpaql).  The environment
    // variables are for the tests (small limits reach both ends in milliseconds); they are part of the source text, hence of the
    // cache key.
    unsigned guard = 1u << 24, credit = 1u << 28;
    if (const char* e = getenv("ZPQ_JIT_HCOMP_GUARD")) { const long v = atol(e); if (v >= 1 && v <= (1l << 28)) guard = (unsigned)v; }
    if (const char* e = getenv("ZPQ_JIT_HCOMP_CREDIT")) { const long v = atol(e); if (v >= 0 && v <= (1l << 30)) credit = (unsigned)v; }
    s += "#define ZGUARD " + itos(guard) + "u\n#define ZCREDIT " + itos(credit) + "u\n";
  }

  if (getenv("ZPQ_CM_PROGRESS")) s += "#define ZPROGRESS 1\n";
  static const char* names[10] = {"", "CONS", "CM", "ICM", "MATCH", "AVG", "MIX2", "MIX", "ISSE", "SSE"};
  for (int t = 1; t <= 9; ++t) s += std::string("#define ZM_") + names[t] + " " + hex64(mask[t]) + "\n";
  s += "#define Z_FOR_MIX(X) " + mixes + "\n#define Z_FOR_SSE(X) " + sses + "\n#define Z_CHAIN " + chain + "\n";
  s += "#define Z_ISSE_FAR_INPUTS " + far_inputs + "\n#define Z_MIX2_UPDATES " + mix2_updates + "\n";
  if (getenv("ZPQ_CM_PROF")) s += "#define ZPROF 1\n";
  // requesting the next bit's entries one bit early (both outcomes): bit-exact, measured SLOWER (1.25 s against 1.03 s
  // for a 100 KB block, 448 against 400 ms for 2048 blocks): the ~80 instructions it adds per bit cost more than the
  // waits it removes -- most of a bit's waiting is LDS latency on the dependent chain, not the round of loads.  Kept
  // behind ZPQ_CM_SPEC=1 for tuning.
  s += std::string("#define ZSPEC ") + (getenv("ZPQ_CM_SPEC") ? "1" : "0") + "\n";
  // ZPQ_CM_PRE_LATE=1: the second nibble's bucket is fetched when the nibble is known (one line per component and
  // nibble) instead of both candidates one bit early (two lines): less memory traffic, one exposed round trip per byte
  s += std::string("#define ZPRE_LATE ") + (getenv("ZPQ_CM_PRE_LATE") ? "1" : "0") + "\n";
  std::string body = kSpecSrc;
  const std::string marker = "//@@HCOMP@@";
  const size_t k = body.find(marker);
  if (k == std::string::npos) { *why = "template marker missing"; return ZPQ_ERR_ARG; }
  body.replace(k, marker.size(), gen_zpaql(P.hcomp, "z_hcomp", false));
  *src = s + body;
  return ZPQ_OK;
}

// Compiles (or fetches from the caches) the code object for a source text.  Headers and post-processor programs come
// out of archives, so the work an input can cause is bounded: a process compiles at most ZPQ_JIT_MAX_COMPILES (64)
// sources it did not find in a cache and keeps at most ZPQ_JIT_MAX_MODULES (128) code objects / loaded modules; beyond
// that the caller gets an error and codes those blocks with the interpreter-driven kernels (cm.hip), which need no
// compiler.  `persist`: write the code object to the on-disk cache -- only for sources this library made from its OWN
// configurations (compress side, build-time warm-up); what a foreign archive brought is kept in memory only.
static int compile_source(zpq_ctx* ctx, const std::string& src, bool persist, const Compiled** out) {
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_code.find(src);
    if (it != g_code.end()) { *out = &it->second; return ZPQ_OK; }
    if (g_code.size() >= (size_t)env_int("ZPQ_JIT_MAX_MODULES", 128))
      return zpq_fail(ctx, ZPQ_ERR_METHOD, "run-time compiled kernels: limit of %d code objects reached (ZPQ_JIT_MAX_MODULES)", env_int("ZPQ_JIT_MAX_MODULES", 128));
  }
  std::lock_guard<std::mutex> ck(g_compile_mu);
  {
    std::lock_guard<std::mutex> lk(g_mu);              // another thread may have compiled it while this one waited
    auto it = g_code.find(src);
    if (it != g_code.end()) { *out = &it->second; return ZPQ_OK; }
  }
  Compiled c;
  const std::string dir = cache_dir();
  const std::string path = dir.empty() ? std::string() : dir + cache_name(src);
  if (!path.empty() && read_cached(path, src, c.code)) {
    std::lock_guard<std::mutex> lk(g_mu);
    *out = &(g_code[src] = std::move(c));
    return ZPQ_OK;
  }
  if (g_fresh >= env_int("ZPQ_JIT_MAX_COMPILES", 64))
    return zpq_fail(ctx, ZPQ_ERR_METHOD, "run-time compiled kernels: limit of %d compilations per process reached (ZPQ_JIT_MAX_COMPILES)", env_int("ZPQ_JIT_MAX_COMPILES", 64));
  ++g_fresh;
  hiprtcProgram prog;
  if (hiprtcCreateProgram(&prog, src.c_str(), "cm_spec.hip", 0, nullptr, nullptr) != HIPRTC_SUCCESS)
    return zpq_fail(ctx, ZPQ_ERR_HIP, "hiprtcCreateProgram failed");
  const char* opts[] = {"--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-label", "-Wno-unused-variable"};
  const hiprtcResult r = hiprtcCompileProgram(prog, 5, opts);
  if (r != HIPRTC_SUCCESS) {
    size_t ls = 0;
    std::string log;
    (void)hiprtcGetProgramLogSize(prog, &ls);
    if (ls > 1) { log.resize(ls); (void)hiprtcGetProgramLog(prog, &log[0]); }
    (void)hiprtcDestroyProgram(&prog);
    if (const char* d = getenv("ZPQ_JIT_DUMP")) { FILE* f = fopen(d, "w"); if (f) { fputs(src.c_str(), f); fclose(f); } }
    fprintf(stderr, "[zpaqhip] hiprtc failed on a generated kernel (%s):\n%.4000s\n", hiprtcGetErrorString(r), log.c_str());
    return zpq_fail(ctx, ZPQ_ERR_HIP, "hiprtc: %s: %.400s", hiprtcGetErrorString(r), log.c_str());
  }
  size_t cs = 0;
  (void)hiprtcGetCodeSize(prog, &cs);
  c.code.resize(cs);
  (void)hiprtcGetCode(prog, c.code.data());
  (void)hiprtcDestroyProgram(&prog);
  if (persist && !path.empty()) write_cached(path, src, c.code);
  std::lock_guard<std::mutex> lk(g_mu);
  *out = &(g_code[src] = std::move(c));
  return ZPQ_OK;
}

int zpq_cm_spec_get(zpq_ctx* ctx, const zpq_cm_header& P, bool own_config, zpq_cm_spec** out) {
  *out = nullptr;
  std::string src, why;
  int rc = zpq_cm_spec_source(P, &src, &why);
  if (rc) return zpq_fail(ctx, rc, "specialised coder: %s", why.c_str());
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_mods.find({ctx->device, src});
    if (it != g_mods.end()) { *out = it->second; return ZPQ_OK; }
    if (modules_loaded() >= (size_t)env_int("ZPQ_JIT_MAX_MODULES", 128))
      return zpq_fail(ctx, ZPQ_ERR_METHOD, "run-time compiled kernels: limit of loaded modules reached (ZPQ_JIT_MAX_MODULES)");
  }
  const Compiled* c = nullptr;
  rc = compile_source(ctx, src, own_config, &c);
  if (rc) return rc;
  zpq_cm_spec* k = new zpq_cm_spec();
  if (hipModuleLoadData(&k->mod, c->code.data()) != hipSuccess) { delete k; return zpq_fail(ctx, ZPQ_ERR_HIP, "hipModuleLoadData failed for the specialised coder"); }
  if (hipModuleGetFunction(&k->enc, k->mod, "cm_spec_encode") != hipSuccess || hipModuleGetFunction(&k->dec, k->mod, "cm_spec_decode") != hipSuccess) {
    (void)hipModuleUnload(k->mod); delete k;
    return zpq_fail(ctx, ZPQ_ERR_HIP, "specialised coder: kernels missing from the module");
  }
  k->h_lds = P.hh <= 10;
  k->waves = spec_waves();
  while (k->waves > 1 && 88064u + (k->h_lds ? k->waves * (4u << P.hh) : 0u) > 160u * 1024u - 1024u) k->waves >>= 1;
  std::lock_guard<std::mutex> lk(g_mu);
  auto ins = g_mods.insert({{ctx->device, src}, k});
  if (!ins.second) { (void)hipModuleUnload(k->mod); delete k; }
  *out = ins.first->second;
  return ZPQ_OK;
}

u32 zpq_cm_spec_waves(const zpq_cm_spec* k) { return k->waves; }

int zpq_cm_spec_launch(zpq_ctx* ctx, zpq_cm_spec* k, hipStream_t st, const void* d_jobs, u32 njobs, u32* d_counter, const void* d_tables,
                       int encode) {
  // one workgroup per compute unit (the tables fill more than half of its LDS); as many waves per workgroup as it
  // takes to seat every block, at most what the module was compiled for
  const u32 cus = (u32)ctx->cu_count;
  const u32 seats = njobs;                                    // one wave per block
  u32 w = (seats + cus - 1) / cus;
  if (w < 1) w = 1;
  if (w > k->waves) w = k->waves;
  u32 grid = (seats + w - 1) / w;
  if (grid > cus) grid = cus;
  void* args[] = {(void*)&d_jobs, (void*)&njobs, (void*)&d_counter, (void*)&d_tables};
  ZpqProfScope prof(ctx, encode ? "cm_spec_encode" : "cm_spec_decode", st);
  ZPQ_HIP(ctx, hipModuleLaunchKernel(encode ? k->enc : k->dec, grid, 1, 1, w * 64, 1, 1, 0, st, args, nullptr));
  return ZPQ_OK;
}

// ---- any post-processor program, translated (rows a14, a16) -------------------------------------------------------
// PostProcessor::write runs the PCOMP program once per decoded byte and once with 2^32-1 when the segment ends
// (ZSFX/libzpaq.cpp:2185-2226).  The interpreter in cm.hip does that at ~1000 cycles per ZPAQL instruction; the
// translated program runs at the speed of the machine code it became (one lane: a ZPAQL machine is one serial thread).
namespace {
const char* kPcompSrc = R"ZPQSRC(
typedef unsigned char u8; typedef unsigned short u16; typedef unsigned int u32; typedef unsigned long long u64;
#define ZGA __attribute__((address_space(1)))
typedef ZGA u32 g_u32; typedef ZGA u8 g_u8;
typedef g_u32* zh_ptr;
#define ZDEV __device__ inline __attribute__((always_inline))
struct ZVm { u32 a, b, c, d, f, err, g, lim; };      // err: 1 = malformed program, 2 = loop budget used up
//@@PCOMP@@
// seg (null: one segment of n bytes): u32[nseg] input bytes per segment of the block, then u32[nseg] (result) the output end
// of each -- the machine keeps its state from segment to segment and sees 2^32-1 at the end of each (PostProcessor::write)
extern "C" __global__ __launch_bounds__(64) void pcomp_spec(const u8* in, u32 n, u8* out, u32 cap, u32* H, u8* M, u32* R, u32* result, u32* seg, u32 nseg) {
  if (threadIdx.x) return;
  // budget of backward jumps for the WHOLE segment (a program that spins is a format error, not a hung GPU): what the
  // post-processors in use need is a small multiple of the bytes they read and write
  const u64 lim = (1ull << 22) + 64ull * ((u64)n + (u64)cap);
  ZVm z = {0, 0, 0, 0, 0, 0, 0, lim > 0x7fffffffull ? 0x7fffffffu : (u32)lim};
  u32 op = 0;
  const g_u8* gin = (const g_u8*)in;
  const u32 ns = seg ? nseg : 1u;
  u32 i = 0;
  for (u32 s = 0; s < ns && !z.err; ++s) {
    const u32 send = seg ? i + seg[s] : n;
    for (; i < send && i < n && !z.err; ++i) z_pcomp(gin[i], z, (g_u8*)M, (g_u32*)R, (zh_ptr)H, (g_u8*)out, cap, op);
    if (!z.err) z_pcomp(0xffffffffu, z, (g_u8*)M, (g_u32*)R, (zh_ptr)H, (g_u8*)out, cap, op);
    if (seg) seg[nseg + s] = op;
  }
  result[0] = op;
  result[1] = z.err == 2 ? (u32)-9 : z.err ? (u32)-6 : (op > cap ? (u32)-4 : 0u);
}
)ZPQSRC";

struct PcompMod { hipModule_t mod; hipFunction_t fn; };
std::map<std::pair<int, std::string>, PcompMod> g_pmods;
size_t modules_loaded() { return g_mods.size() + g_pmods.size(); }       // under g_mu
}  // namespace

// Runs pcomp[0..psize) over d_in[0..n) on the device; H, M, R are zeroed device arrays of 2^ph words, 2^pm bytes, 256 words.
int zpq_pcomp_spec_run(zpq_ctx* ctx, hipStream_t st, const u8* pcomp, u32 psize, u32 ph, u32 pm, const u8* d_in, u32 n, u8* d_out, u32 out_cap,
                       u32* d_H, u8* d_M, u32* d_R, u32* d_result, u32* d_seg, u32 nseg) {
  std::vector<u8> code(pcomp, pcomp + psize);
  std::string src = "#define ZHMASK " + itos((1u << ph) - 1) + "u\n#define ZMMASK " + itos((1u << pm) - 1) + "u\n#define ZGUARD z.lim\n";
  std::string body = kPcompSrc;
  const std::string marker = "//@@PCOMP@@";
  body.replace(body.find(marker), marker.size(), gen_zpaql(code, "z_pcomp", true));
  src += body;
  PcompMod pmod;
  bool have = false;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_pmods.find({ctx->device, src});
    if (it != g_pmods.end()) { pmod = it->second; have = true; }
    else if (modules_loaded() >= (size_t)env_int("ZPQ_JIT_MAX_MODULES", 128))
      return zpq_fail(ctx, ZPQ_ERR_METHOD, "run-time compiled kernels: limit of loaded modules reached (ZPQ_JIT_MAX_MODULES)");
  }
  if (!have) {
    const Compiled* c = nullptr;
    int rc = compile_source(ctx, src, false, &c);        // a post-processor out of an archive: never written to disk
    if (rc) return rc;
    if (hipModuleLoadData(&pmod.mod, c->code.data()) != hipSuccess) return zpq_fail(ctx, ZPQ_ERR_HIP, "hipModuleLoadData failed for a translated post-processor");
    if (hipModuleGetFunction(&pmod.fn, pmod.mod, "pcomp_spec") != hipSuccess) { (void)hipModuleUnload(pmod.mod); return zpq_fail(ctx, ZPQ_ERR_HIP, "translated post-processor: kernel missing"); }
    std::lock_guard<std::mutex> lk(g_mu);
    auto ins = g_pmods.insert({{ctx->device, src}, pmod});
    if (!ins.second) { (void)hipModuleUnload(pmod.mod); pmod = ins.first->second; }
  }
  void* args[] = {(void*)&d_in, (void*)&n, (void*)&d_out, (void*)&out_cap, (void*)&d_H, (void*)&d_M, (void*)&d_R, (void*)&d_result, (void*)&d_seg, (void*)&nseg};
  ZpqProfScope prof(ctx, "pcomp_spec", st);
  ZPQ_HIP(ctx, hipModuleLaunchKernel(pmod.fn, 1, 1, 1, 64, 1, 1, 0, st, args, nullptr));
  return ZPQ_OK;
}

// Host-only: generates and compiles the kernels for a header into the on-disk cache (no GPU needed: hiprtc
// cross-compiles).  Used by the build step so that a GPU box starts with the common models ready.
extern "C" int zpq_cm_precompile(const uint8_t* header, uint32_t header_len) {
  zpq_cm_header P;
  int rc = zpq_cm_parse_header(nullptr, header, header_len, P);
  if (rc) return rc;
  std::string src, why;
  rc = zpq_cm_spec_source(P, &src, &why);
  if (rc) return rc;
  const Compiled* c = nullptr;
  return compile_source(nullptr, src, true, &c);
}

// Diagnostic: the generated source text for a header (tests compile and inspect it).
extern "C" int zpq_cm_spec_source_text(const uint8_t* header, uint32_t header_len, char* out, size_t cap, size_t* len) {
  zpq_cm_header P;
  int rc = zpq_cm_parse_header(nullptr, header, header_len, P);
  if (rc) return rc;
  std::string src, why;
  rc = zpq_cm_spec_source(P, &src, &why);
  if (rc) return rc;
  if (len) *len = src.size();
  if (src.size() + 1 > cap) return ZPQ_ERR_CAPACITY;
  memcpy(out, src.c_str(), src.size() + 1);
  return ZPQ_OK;
}
