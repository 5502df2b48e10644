// Host side of block configuration: method string -> ZPAQ config
// source -> block header bytes and post-processor bytecode.  No device code in this file.
//
//  * zpaql_compile: the ZPAQL configuration compiler.  Replaces libzpaq::Compiler (grammar
//    ZSFX/libzpaq.h:596-760, translation ZSFX/libzpaq.cpp:2500-2706); pinned byte for byte against it
//    (tests/test_config_cpu.py drives both over every configuration this file can generate) and
//    against the headers stored in the reference's fixtures.
//  * make_config / expand_method: libzpaq 7.15's makeConfig() and the "0".."5" expansion at the top of
//    compressBlock().  NOT IN THE SNAPSHOT: ZSFX/libzpaq.cpp stops after LZBuffer (line 6552), so these
//    two are restated from the published libzpaq 7.15 (public domain) and anchored on what the
//    fixtures hold: level 1 (type 512) on every d/h/i block of AUTOTEST/sha256.zpaq, level 5 (type
//    512) on ZSFX/zsfx.zpaq and ZSFX/zsfx32.zpaq -- whose 255-byte headers exercise the c, i, a, m, t,
//    s and w component generators.  Pre-processor variants no fixture pins (byte-aligned LZ77, BWT,
//    E8E9) are refused with ZPQ_ERR_METHOD rather than approximated.
#include <ctype.h>
#include <stdlib.h>

#include <stdexcept>

#include "zpq_internal.h"

namespace {

struct ConfigError : std::runtime_error { using std::runtime_error::runtime_error; };

std::string itos(long long v) { return std::to_string(v); }
int lg2(unsigned x) { int r = 0; while (x) ++r, x >>= 1; return r; }          // bits needed for x
int popcount32(unsigned x) { int r = 0; while (x) r += x & 1, x >>= 1; return r; }

// ---------------------------------------------------------------------------------------------------
// ZPAQL compiler
// ---------------------------------------------------------------------------------------------------
class Zpaql {
 public:
  Zpaql(const char* src, const int* args) : p_(src), args_(args) {}

  // header: hsize[2] hh hm ph pm n COMP 0 HCOMP 0 ; pcomp: bytecode incl. the closing 0 (empty = none)
  void compile(std::vector<u8>& header, std::vector<u8>& pcomp, std::string& pcomp_cmd) {
    want("comp");
    header.assign(7, 0);
    header[2] = (u8)number(0, 255); header[3] = (u8)number(0, 255);
    header[4] = (u8)number(0, 255); header[5] = (u8)number(0, 255);
    const int n = number(0, 255);
    header[6] = (u8)n;
    for (int i = 0; i < n; ++i) {
      if (number(i, i) != i) fail("component index");
      component(header);
    }
    header.push_back(0);
    want("hcomp");
    std::vector<u8> code;
    const std::string closer = program(code);
    header.insert(header.end(), code.begin(), code.end());
    const size_t hsize = header.size() - 2;
    if (hsize > 65535) fail("header too large");
    header[0] = (u8)(hsize & 255); header[1] = (u8)(hsize >> 8);
    pcomp.clear(); pcomp_cmd.clear();
    if (closer == "pcomp") {
      // the rest of the line up to ';' names an external preprocessor (kept for callers, unused here)
      while (*p_ && *p_ != ';') pcomp_cmd += *p_++;
      if (*p_ != ';') fail("expected ;");
      ++p_;
      while (!pcomp_cmd.empty() && isspace((unsigned char)pcomp_cmd.back())) pcomp_cmd.pop_back();
      size_t b = 0; while (b < pcomp_cmd.size() && isspace((unsigned char)pcomp_cmd[b])) ++b;
      pcomp_cmd.erase(0, b);
      if (program(pcomp) != "end") fail("expected END");
    } else if (closer == "post") {      // old style "post 0 end": no post-processor
      number(0, 0);
      want("end");
    } else if (closer != "end") {
      fail("expected END or PCOMP");
    }
  }

 private:
  const char* p_;
  const int* args_;
  std::string tok_;

  [[noreturn]] void fail(const std::string& what) { throw ConfigError("config: " + what + (tok_.empty() ? "" : " at '" + tok_ + "'")); }

  // next token, lower-cased; comments are ( ... ) and nest
  bool next() {
    tok_.clear();
    int depth = 0;
    for (;; ++p_) {
      const char c = *p_;
      if (!c) { if (depth) fail("unbalanced ("); return false; }
      if (c == '(') ++depth;
      else if (c == ')') { if (--depth < 0) fail("unbalanced )"); }
      else if (depth == 0 && (unsigned char)c > ' ') break;
    }
    while ((unsigned char)*p_ > ' ' && *p_ != '(' && *p_ != ')') tok_ += (char)tolower((unsigned char)*p_++);
    return true;
  }
  void want(const char* w) { if (!next() || tok_ != w) fail(std::string("expected ") + w); }

  int number(int lo, int hi) {
    if (!next()) fail("unexpected end");
    long v = 0;
    const char* t = tok_.c_str();
    if (t[0] == '$' && t[1] >= '1' && t[1] <= '9') {
      if (t[2] == '+') v = atol(t + 3);
      if (args_) v += args_[t[1] - '1'];
    } else if (t[0] == '-' || isdigit((unsigned char)t[0])) {
      v = atol(t);
    } else {
      fail("expected a number");
    }
    if (v < lo || v > hi) fail("number out of range " + itos(lo) + ".." + itos(hi));
    return (int)v;
  }

  void component(std::vector<u8>& h) {
    static const struct { const char* name; int code, nargs; } K[] = {
        {"const", 1, 1}, {"cm", 2, 2}, {"icm", 3, 1}, {"match", 4, 2}, {"avg", 5, 3},
        {"mix2", 6, 5},  {"mix", 7, 5}, {"isse", 8, 2}, {"sse", 9, 4}};
    if (!next()) fail("component expected");
    for (const auto& k : K)
      if (tok_ == k.name) {
        h.push_back((u8)k.code);
        for (int i = 0; i < k.nargs; ++i) h.push_back((u8)number(0, 255));
        return;
      }
    fail("unknown component");
  }

  // opcode of an instruction token (-1 = none); *operand: 0 none, 1 byte, 2 word
  static int opcode(const std::string& t, int* operand) {
    static const char* R[7] = {"a", "b", "c", "d", "*b", "*c", "*d"};
    *operand = 0;
    if (t == "error") return 0;
    if (t == "halt") return 56;
    if (t == "out") return 57;
    if (t == "hash") return 59;
    if (t == "hashd") return 60;
    if (t == "jt") { *operand = 1; return 39; }
    if (t == "jf") { *operand = 1; return 47; }
    if (t == "jmp") { *operand = 1; return 63; }
    if (t == "lj") { *operand = 2; return 255; }
    if (t == "r=a") { *operand = 1; return 55; }
    for (int x = 0; x < 7; ++x) {
      const std::string r = R[x];
      if (t.compare(0, r.size(), r) != 0) continue;
      const std::string s = t.substr(r.size());
      if (s == "<>a" && x) return 8 * x;
      if (s == "++") return 8 * x + 1;
      if (s == "--") return 8 * x + 2;
      if (s == "!") return 8 * x + 3;
      if (s == "=0") return 8 * x + 4;
      if (s == "=r" && x < 4) { *operand = 1; return 8 * x + 7; }
      if (s == "=") { *operand = 1; return 64 + 8 * x + 7; }
      if (s.size() > 1 && s[0] == '=')
        for (int y = 0; y < 7; ++y)
          if (s.compare(1, std::string::npos, R[y]) == 0) return 64 + 8 * x + y;
      if (x == 0) {
        static const char* B[14] = {"+=", "-=", "*=", "/=", "%=", "&=", "&~", "|=", "^=", "<<=", ">>=", "==", "<", ">"};
        for (int k = 0; k < 14; ++k) {
          const std::string b = B[k];
          if (s.compare(0, b.size(), b) != 0) continue;
          const std::string y = s.substr(b.size());
          if (y.empty()) { *operand = 1; return 128 + 8 * k + 7; }
          for (int g = 0; g < 7; ++g)
            if (y == R[g]) return 128 + 8 * k + g;
        }
      }
    }
    return -1;
  }

  // Compiles one program up to its closing keyword (returned); appends the closing 0.
  std::string program(std::vector<u8>& c) {
    enum { JT = 39, JF = 47, JMP = 63, LJ = 255 };
    std::vector<size_t> ifs, dos;     // positions of pending forward-jump operands / loop heads
    auto patch_if = [&](size_t a, size_t target_rel, size_t target_abs) {
      if (c[a - 1] != LJ) {
        if (target_rel > 127) fail("IF too big, try IFL, IFNOTL");
        c[a] = (u8)target_rel;
      } else {
        if (target_abs > 65535) fail("program too long");
        c[a] = (u8)(target_abs & 255); c[a + 1] = (u8)(target_abs >> 8);
      }
    };
    for (;;) {
      if (!next()) fail("unexpected end of program");
      const std::string t = tok_;
      if (t == "end" || t == "pcomp" || t == "post") {
        c.push_back(0);      // (an IF or DO left open is not an error for the reference compiler either)
        return t;
      }
      if (t == "if" || t == "ifnot") {
        c.push_back(t == "if" ? JF : JT); c.push_back(0);
        ifs.push_back(c.size() - 1);
      } else if (t == "ifl" || t == "ifnotl") {
        c.push_back(t == "ifl" ? JT : JF); c.push_back(3);
        c.push_back(LJ); c.push_back(0); c.push_back(0);
        ifs.push_back(c.size() - 2);
      } else if (t == "else" || t == "elsel") {
        if (ifs.empty()) fail("ELSE without IF");
        const size_t a = ifs.back(); ifs.pop_back();
        const bool lng = t == "elsel";
        const size_t after = c.size() + (lng ? 3 : 2);      // first instruction of the ELSE part
        patch_if(a, after - a - 1, after);
        if (lng) { c.push_back(LJ); c.push_back(0); c.push_back(0); ifs.push_back(c.size() - 2); }
        else { c.push_back(JMP); c.push_back(0); ifs.push_back(c.size() - 1); }
      } else if (t == "endif") {
        if (ifs.empty()) fail("ENDIF without IF");
        const size_t a = ifs.back(); ifs.pop_back();
        patch_if(a, c.size() - a - 1, c.size());
      } else if (t == "do") {
        dos.push_back(c.size());
      } else if (t == "while" || t == "until" || t == "forever") {
        if (dos.empty()) fail("WHILE/UNTIL/FOREVER without DO");
        const size_t a = dos.back(); dos.pop_back();
        const long back = (long)a - (long)c.size() - 2;
        if (back >= -127) {
          c.push_back(t == "while" ? JT : t == "until" ? JF : JMP);
          c.push_back((u8)(back & 255));
        } else {
          if (a > 65535) fail("program too long");
          if (t == "while") { c.push_back(JF); c.push_back(3); }
          else if (t == "until") { c.push_back(JT); c.push_back(3); }
          c.push_back(LJ); c.push_back((u8)(a & 255)); c.push_back((u8)(a >> 8));
        }
      } else {
        int operand = 0;
        const int op = opcode(t, &operand);
        if (op < 0) fail("unknown instruction");
        c.push_back((u8)op);
        if (operand == 1) {
          if (op == JT || op == JF || op == JMP) c.push_back((u8)(number(-128, 127) & 255));
          else c.push_back((u8)number(0, 255));
        } else if (operand == 2) {
          const int v = number(0, 65535);
          c.push_back((u8)(v & 255)); c.push_back((u8)(v >> 8));
        }
      }
      if (c.size() > 65530) fail("program too long");
    }
  }
};

// ---------------------------------------------------------------------------------------------------
// makeConfig (libzpaq 7.15): "x|s N1,...,N9 components" -> config source; args[] = $1..$9
// ---------------------------------------------------------------------------------------------------

// The LZ77 level-1 ("lazy2") post-processor; rb low offset bits are sent raw for blocks above 16 MiB,
// and the E8E9 inverse is folded in when the pre-processor applied it.  The rb = 0 / no-E8E9 form is
// what every d/h/i block of the fixture carries (302 bytes, tests/golden).
std::string lazy2_source(int rb, bool doe8) {
  std::string p = "pcomp lazy2 3 ;\n";
  p += " (r1 = state\n  r2 = len - match or literal length\n  r3 = m - number of offset bits expected\n"
       "  r4 = ptr to buf\n  r5 = r - low bits of offset\n  c = bits - input buffer\n  d = n - number of bits in c)\n\n"
       "  a> 255 if\n";
  if (doe8)
    p += "    b=0 d=r 4 do (for b=0..d-1, d = end of buf)\n      a=b a==d ifnot\n        a+= 4 a<d if\n"
         "          a=*b a&= 254 a== 232 if (e8 or e9?)\n            c=b b++ b++ b++ b++ a=*b a++ a&= 254 a== 0 if (00 or ff)\n"
         "              b-- a=*b\n              b-- a<<= 8 a+=*b\n              b-- a<<= 8 a+=*b\n              a-=b a++\n"
         "              *b=a a>>= 8 b++\n              *b=a a>>= 8 b++\n              *b=a b++\n            endif\n"
         "            b=c\n          endif\n        endif\n        a=*b out b++\n      forever\n    endif\n\n";
  p += "    (reset state)\n    a=0 b=0 c=0 d=0 r=a 1 r=a 2 r=a 3 r=a 4\n    halt\n  endif\n\n"
       "  a<<=d a+=c c=a               (bits+=a<<n)\n  a= 8 a+=d d=a                (n+=8)\n\n"
       "  (if state==0 (expect new code))\n  a=r 1 a== 0 if (match code mm,mmm)\n    a= 1 r=a 2                 (len=1)\n"
       "    a=c a&= 3 a> 0 if          (if (bits&3))\n      a-- a<<= 3 r=a 3           (m=((bits&3)-1)*8)\n"
       "      a=c a>>= 2 c=a             (bits>>=2)\n      b=r 3 a&= 7 a+=b r=a 3     (m+=bits&7)\n"
       "      a=c a>>= 3 c=a             (bits>>=3)\n      a=d a-= 5 d=a              (n-=5)\n"
       "      a= 1 r=a 1                 (state=1)\n    else (literal, discard 00)\n"
       "      a=c a>>= 2 c=a             (bits>>=2)\n      d-- d--                    (n-=2)\n"
       "      a= 3 r=a 1                 (state=3)\n    endif\n  endif\n\n"
       "  (while state==1 && n>=3 (expect match length n*4+ll -> r2))\n  do a=r 1 a== 1 if a=d a> 2 if\n"
       "    a=c a&= 1 a== 1 if         (if bits&1)\n      a=c a>>= 1 c=a             (bits>>=1)\n"
       "      b=r 2 a=c a&= 1 a+=b a+=b r=a 2 (len+=len+(bits&1))\n      a=c a>>= 1 c=a             (bits>>=1)\n"
       "      d-- d--                    (n-=2)\n    else\n      a=c a>>= 1 c=a             (bits>>=1)\n"
       "      a=r 2 a<<= 2 b=a           (len<<=2)\n      a=c a&= 3 a+=b r=a 2       (len+=bits&3)\n"
       "      a=c a>>= 2 c=a             (bits>>=2)\n      d-- d-- d--                (n-=3)\n";
  p += rb ? "      a= 5 r=a 1                 (state=5)\n" : "      a= 2 r=a 1                 (state=2)\n";
  p += "    endif\n  forever endif endif\n\n";
  if (rb)
    p += "  (if state==5 && n>=8) (expect low bits of offset to put in r5)\n  a=r 1 a== 5 if a=d a> " + itos(rb - 1) + " if\n"
         "    a=c a&= " + itos((1 << rb) - 1) + " r=a 5            (save r in R5)\n    a=c a>>= " + itos(rb) + " c=a\n"
         "    a=d a-= " + itos(rb) + " d=a\n    a= 2 r=a 1                   (go to state 2)\n  endif endif\n\n";
  p += "  (if state==2 && n>=m) (expect m offset bits)\n  a=r 1 a== 2 if a=r 3 a>d ifnot\n"
       "    a=c r=a 6 a=d r=a 7          (save c=bits, d=n in r6,r7)\n    b=r 3 a= 1 a<<=b d=a         (d=1<<m)\n"
       "    a-- a&=c a+=d                (d=offset=bits&((1<<m)-1)|(1<<m))\n";
  if (rb) p += "    a<<= " + itos(rb) + " d=r 5 a+=d a-= " + itos((1 << rb) - 1) + "\n";
  p += "    d=a b=r 4 a=b a-=d c=a       (c=p=(b=ptr)-offset)\n\n    (while len-- (copy and output match d bytes from *c to *b))\n"
       "    d=r 2 do a=d a> 0 if d--\n      a=*c *b=a c++ b++          (buf[ptr++]-buf[p++])\n";
  if (!doe8) p += " out\n";
  p += "    forever endif\n    a=b r=a 4\n\n    a=r 6 b=r 3 a>>=b c=a        (bits>>=m)\n    a=r 7 a-=b d=a               (n-=m)\n"
       "    a=0 r=a 1                    (state=0)\n  endif endif\n\n"
       "  (while state==3 && n>=2 (expect literal length))\n  do a=r 1 a== 3 if a=d a> 1 if\n"
       "    a=c a&= 1 a== 1 if         (if bits&1)\n      a=c a>>= 1 c=a              (bits>>=1)\n"
       "      b=r 2 a&= 1 a+=b a+=b r=a 2 (len+=len+(bits&1))\n      a=c a>>= 1 c=a              (bits>>=1)\n"
       "      d-- d--                     (n-=2)\n    else\n      a=c a>>= 1 c=a              (bits>>=1)\n"
       "      d--                         (--n)\n      a= 4 r=a 1                  (state=4)\n    endif\n  forever endif endif\n\n"
       "  (if state==4 && n>=8 (expect len literals))\n  a=r 1 a== 4 if a=d a> 7 if\n    b=r 4 a=c *b=a\n";
  if (!doe8) p += " out\n";
  p += "    b++ a=b r=a 4                 (buf[ptr++]=bits)\n    a=c a>>= 8 c=a                (bits>>=8)\n"
       "    a=d a-= 8 d=a                 (n-=8)\n    a=r 2 a-- r=a 2 a== 0 if      (if --len<1)\n"
       "      a=0 r=a 1                     (state=0)\n    endif\n  endif endif\n  halt\nend\n";
  return p;
}

// E8E9 inverse over M[0..b) with output, as the post-processors of levels 2 and 3 run it when the segment ends
// (mirror of e8e9(), ZSFX/libzpaq.cpp:6117-6126)
// Only positions i <= n - 5 carry a transformed operand (e8e9() starts at n - 5): the test of byte i + 4 is made only where it
// lies inside the data (`a=b a<d`).  Until round 6 the program read M[i + 4] behind the data for an E8 / E9 among the last four
// bytes -- zeros or stale bytes there satisfy the test, and the reference's PostProcessor then "restored" an operand that had
// never been transformed (found by tests/test_pcomp_variants_cpu.py::test_e8_e9_among_the_last_bytes...).
std::string e8e9_inverse_source() {
  return "    d=b b=0 do\n      a=b a==d ifnot\n        a=*b a&= 254 a== 232 if\n"
         "          c=b b++ b++ b++ b++ a=b a<d if\n          a=*b a++ a&= 254 a== 0 if\n"
         "            b-- a=*b\n            b-- a<<= 8 a+=*b\n            b-- a<<= 8 a+=*b\n            a-=b a++\n"
         "            *b=a a>>= 8 b++\n            *b=a a>>= 8 b++\n            *b=a b++\n          endif endif\n          b=c\n        endif\n"
         "        a=*b out b++\n      forever\n    endif\n";
}

std::string make_config(const char* method, int args[9]) {
  const char kind = method[0];
  if (kind != 'x' && kind != 's' && kind != '0' && kind != 'i') throw ConfigError("method must begin with 0..5, x or s");
  for (int i = 0; i < 9; ++i) args[i] = 0;
  const char* m = method + 1;
  for (int i = 0; i < 9 && (isdigit((unsigned char)*m) || *m == ',' || *m == '.');) {
    if (isdigit((unsigned char)*m)) args[i] = args[i] * 10 + *m - '0';
    else if (++i < 9) args[i] = 0;
    ++m;
  }
  if (kind == '0') return "comp 0 0 0 0 0 hcomp end\n";

  const int level = args[1] & 3;
  const bool doe8 = args[1] >= 4 && args[1] <= 7;
  std::string hdr, pcomp;
  if (level == 1) {
    const int rb = args[0] > 4 ? args[0] - 4 : 0;
    hdr = "comp 9 16 0 $1+20 ";
    pcomp = lazy2_source(rb, doe8);
  } else if (level == 2) {
    // byte-aligned LZ77 (LZBuffer level 2, ZSFX/libzpaq.cpp:6221-6224, :6519-6547): 00xxxxxx = x+1 literals,
    // yyxxxxxx = match of x+$3 bytes, yy+1 offset bytes (offset-1, MSB first).  d = state, M = output, b = size.
    hdr = "comp 9 16 0 $1+20 ";
    pcomp = "pcomp lzpre c ;\n  a> 255 if\n";
    if (doe8) pcomp += e8e9_inverse_source();
    pcomp += "    b=0 c=0 d=0 a=0 r=a 1 r=a 2\n  halt\n  endif\n"
             "  c=a a=d a== 0 if\n    a=c a>>= 6 a++ d=a\n    a== 1 if\n      a+=c r=a 1 a=0 r=a 2\n    else\n"
             "      d++ a=c a&= 63 a+= $3 r=a 1 a=0 r=a 2\n    endif\n  else\n    a== 1 if\n      a=c *b=a b++\n";
    if (!doe8) pcomp += " out\n";
    pcomp += "      a=r 1 a-- a== 0 if d=0 endif r=a 1\n    else\n      a> 2 if\n        a=r 2 a<<= 8 a|=c r=a 2 d--\n      else\n"
             "        a=r 2 a<<= 8 a|=c c=a a=b a-=c a-- c=a\n        d=r 1\n        do\n          a=*c *b=a c++ b++\n";
    if (!doe8) pcomp += " out\n";
    pcomp += "        d-- a=d a> 0 while\n      endif\n    endif\n  endif\n  halt\nend\n";
  } else if (level == 3) {
    // BWT (LZBuffer level 3, ZSFX/libzpaq.cpp:6225-6226, :6317-6326): the transform with the end-of-string coded
    // as 255 and its position in the last 4 bytes, LSB first.  The program collects it in M, then inverts it through
    // a linked list in H.
    hdr = "comp 9 16 $1+20 $1+20 ";
    pcomp = "pcomp bwtrle c ;\n  a> 255 ifnot\n    *b=a b++\n  elsel\n"
            "    b-- a=*b\n    b-- a<<= 8 a+=*b\n    b-- a<<= 8 a+=*b\n    b-- a<<= 8 a+=*b c=a r=a 1\n"
            "    a=b r=a 2\n"
            "    do\n      a=b a> 0 if\n        b-- a=*b a++ a&= 255 d=a d! *d++\n      forever\n    endif\n"
            "    d=0 d! *d= 1 a=0\n    do\n      a+=*d *d=a d--\n    d<>a a! a> 255 a! d<>a until\n"
            "    b=0 do\n      a=c a>b if\n        d=*b d! *d++ d=*d d-- *d=b\n      b++ forever\n    endif\n"
            "    b=c b++ c=r 2 do\n      a=c a>b if\n        d=*b d! *d++ d=*d d-- *d=b\n      b++ forever\n    endif\n";
    if (args[0] <= 4) {
      pcomp += "    b=0 do\n      a=c a>b if\n        d=b a=*d a<<= 8 a+=*b *d=a\n      b++ forever\n    endif\n"
               "    d=r 1 b=0 do\n      a=d a== 0 ifnot\n        a=*d a>>= 8 d=a\n";
      pcomp += doe8 ? " *b=*d b++\n" : " a=*d out\n";
      pcomp += "      forever\n    endif\n";
      if (doe8) pcomp += e8e9_inverse_source();
      pcomp += "  endif\n  halt\nend\n";
    } else if (!doe8) {
      pcomp += "    d=r 1 do\n      a=d a== 0 ifnot\n        d=*d b=d a=*b out\n      forever\n    endif\n  endif\n  halt\nend\n";
    } else {
      // Above 16 MiB the links fill H's words and the bytes stay in M: there is no room to collect the output for the
      // whole-buffer E8E9 stage, so the inverse runs as a STREAM behind the walk (round 6; until then this combination was
      // refused): the last four bytes wait in R3..R6 (R7 = how many, R8 = the position of the oldest); a byte that arrives is
      // byte i + 4 of the oldest, which is exactly what e8e9()'s test reads (ZSFX/libzpaq.cpp:6117-6126: the operand bytes
      // i + 1 .. i + 3 are restored in the registers before they leave); what is left when the walk ends goes out untouched --
      // the last four positions never start an operand.
      pcomp += "    d=r 1 do\n      a=d a== 0 ifnot\n        d=*d b=d a=*b c=a\n"
               "        a=r 7 a== 4 if\n"
               "          a=r 3 a&= 254 a== 232 if\n            a=c a++ a&= 254 a== 0 if\n"
               "              a=r 6 a<<= 8 b=a a=r 5 a+=b a<<= 8 b=a a=r 4 a+=b\n              b=r 8 a-=b\n"
               "              b=a a&= 255 r=a 4 a=b a>>= 8 b=a a&= 255 r=a 5 a=b a>>= 8 a&= 255 r=a 6\n"
               "            endif\n          endif\n"
               "          a=r 3 out\n          a=r 8 a++ r=a 8\n"
               "        else\n          a++ r=a 7\n        endif\n"
               "        a=r 4 r=a 3 a=r 5 r=a 4 a=r 6 r=a 5 a=c r=a 6\n"
               "      forever\n    endif\n"
               "    a=r 7 a> 3 if a=r 3 out endif\n    a=r 7 a> 2 if a=r 4 out endif\n"
               "    a=r 7 a> 1 if a=r 5 out endif\n    a=r 7 a> 0 if a=r 6 out endif\n"
               "    a=0 r=a 7 r=a 8\n  endif\n  halt\nend\n";
    }
  } else if (doe8) {
    // E8E9 alone in front of a model (what level 4 picks for executable data).  libzpaq 7.15 undoes it with a
    // streaming 5-byte window; this program collects the segment in M (2^($1+20) bytes) and runs the same inverse as the
    // level 2 / 3 programs when the segment ends: any ZPAQ reader decodes it, the header bytes differ from 7.15's.
    hdr = "comp 9 16 0 $1+20 ";
    pcomp = "pcomp e8buf c ;\n  a> 255 ifnot\n    *b=a b++\n  else\n" + e8e9_inverse_source() + "    b=0 c=0 d=0\n  endif\n  halt\nend\n";
  } else {
    hdr = "comp 9 16 0 0 ";
    pcomp = "end\n";
  }

  // context model: H[0..254] contexts, H[255..511] position of the last occurrence of byte i-255,
  // M = the last 64 KiB filling backwards, C = pointer to the most recent byte
  int ncomp = 0;
  const int membits = args[0] + 20;
  int sb = 5;       // context bits of the last component
  std::string comp, hcomp = "hcomp\nc-- *c=a a+= 255 d=a *d=c\n";
  if (level == 2) {
    // the parse state of the byte-aligned LZ77 codes for the 256..511 context masks: R1 = 1 + bytes until the next
    // code (starting behind the post-processor preamble: 3 + 108 bytes, 56 more with the E8E9 stage as restated here), R2 = the code
    hcomp += "a=r 1 a== 0 if\n  a= " + itos(111 + 56 * (doe8 ? 1 : 0)) + "\nelse a== 1 if\n  a=*c r=a 2\n  a> 63 if a>>= 6 a++ a++\n"
             "  else a++ a++ endif\nelse\n  a--\nendif endif\nr=a 1\n";
  }
  while (*m && ncomp < 254) {
    std::vector<int> v;
    v.push_back((unsigned char)*m++);
    if (isdigit((unsigned char)*m)) {
      v.push_back(*m++ - '0');
      while (isdigit((unsigned char)*m) || *m == ',' || *m == '.') {
        if (isdigit((unsigned char)*m)) v.back() = v.back() * 10 + *m++ - '0';
        else { v.push_back(0); ++m; }
      }
    }
    const int c0 = v[0];
    if (c0 == 'c') {
      // N1%1000: 0 = ICM, 1..256 = CM with limit N1-1; N1/1000 halves memory; N2: 1..255 offset mod N2,
      // 1000..1255 distance to byte N2-1000; N3...: byte masks (+256: LZ77 state), 1000+: skip bytes
      while (v.size() < 3) v.push_back(0);
      comp += itos(ncomp) + " ";
      sb = 11;
      if (v[2] < 256) sb += lg2((unsigned)v[2]); else sb += 6;
      for (size_t i = 3; i < v.size(); ++i)
        if (v[i] < 512) sb += popcount32((unsigned)v[i]) * 3 / 4;
      if (sb > membits) sb = membits;
      if (v[1] % 1000 == 0) comp += "icm " + itos(sb - 6 - v[1] / 1000) + "\n";
      else comp += "cm " + itos(sb - 2 - v[1] / 1000) + " " + itos(v[1] % 1000 - 1) + "\n";
      hcomp += "d= " + itos(ncomp) + " *d=0\n";
      if (v[2] > 1 && v[2] <= 255) {
        if (lg2((unsigned)v[2]) != lg2((unsigned)v[2] - 1)) hcomp += "a=c a&= " + itos(v[2] - 1) + " hashd\n";
        else hcomp += "a=c a%= " + itos(v[2]) + " hashd\n";
      } else if (v[2] >= 1000 && v[2] <= 1255) {
        hcomp += "a= 255 a+= " + itos(v[2] - 1000) + " d=a a=*d a-=c a> 255 if a= 255 endif d= " + itos(ncomp) + " hashd\n";
      }
      for (size_t i = 3; i < v.size(); ++i) {
        if (i == 3) hcomp += "b=c ";
        if (v[i] == 255) hcomp += "a=*b hashd\n";
        else if (v[i] > 0 && v[i] < 255) hcomp += "a=*b a&= " + itos(v[i]) + " hashd\n";
        else if (v[i] >= 256 && v[i] < 512) {
          hcomp += "a=r 1 a> 1 if\n  a=r 2 a< 64 if\n    a=*b ";
          if (v[i] < 511) hcomp += "a&= " + itos(v[i] - 256);
          hcomp += " hashd\n  else\n    a>>= 6 hashd a=r 1 hashd\n  endif\nelse\n  a= 255 hashd a=r 2 hashd\nendif\n";
        } else if (v[i] >= 1256) {
          hcomp += "a= " + itos(((v[i] - 1000) >> 8) & 255) + " a<<= 8 a+= " + itos((v[i] - 1000) & 255) + " a+=b b=a\n";
        } else if (v[i] > 1000) {
          hcomp += "a= " + itos(v[i] - 1000) + " a+=b b=a\n";
        }
        if (v[i] < 512 && i < v.size() - 1) hcomp += "b++ ";
      }
      ++ncomp;
    }
    if ((c0 == 'm' || c0 == 't' || c0 == 's') && ncomp > (c0 == 't' ? 1 : 0)) {
      // m,8,24: MIX size rate; t,8,24: MIX2 size rate; s,8,32,255: SSE size start limit
      if (v.size() <= 1) v.push_back(8);
      if (v.size() <= 2) v.push_back(24 + 8 * (c0 == 's'));
      if (c0 == 's' && v.size() <= 3) v.push_back(255);
      comp += itos(ncomp);
      sb = 5 + v[1] * 3 / 4;
      if (c0 == 'm') comp += " mix " + itos(v[1]) + " 0 " + itos(ncomp) + " " + itos(v[2]) + " 255\n";
      else if (c0 == 't') comp += " mix2 " + itos(v[1]) + " " + itos(ncomp - 1) + " " + itos(ncomp - 2) + " " + itos(v[2]) + " 255\n";
      else comp += " sse " + itos(v[1]) + " " + itos(ncomp - 1) + " " + itos(v[2]) + " " + itos(v[3]) + "\n";
      if (v[1] > 8) {
        hcomp += "d= " + itos(ncomp) + " *d=0 b=c a=0\n";
        for (; v[1] >= 16; v[1] -= 8) {
          hcomp += "a<<= 8 a+=*b";
          if (v[1] > 16) hcomp += " b++";
          hcomp += "\n";
        }
        if (v[1] > 8) hcomp += "a<<= 8 a+=*b a>>= " + itos(16 - v[1]) + "\n";
        hcomp += "a<<= 8 *d=a\n";
      }
      ++ncomp;
    }
    if (c0 == 'i' && ncomp > 0) {
      // ISSE chain, context order growing by N1, N2, ...
      hcomp += "d= " + itos(ncomp - 1) + " b=c a=*d d++\n";
      for (size_t i = 1; i < v.size() && ncomp < 254; ++i) {
        for (int j = 0; j < v[i] % 10; ++j) {
          hcomp += "hash ";
          if (i < v.size() - 1 || j < v[i] % 10 - 1) hcomp += "b++ ";
          sb += 6;
        }
        hcomp += "*d=a";
        if (i < v.size() - 1) hcomp += " d++";
        hcomp += "\n";
        if (sb > membits) sb = membits;
        comp += itos(ncomp) + " isse " + itos(sb - 6 - v[i] / 10) + " " + itos(ncomp - 1) + "\n";
        ++ncomp;
      }
    }
    if (c0 == 'a') {
      // a24,0,0: MATCH; N1 = hash multiplier, N2/N3 halve the buffer / the table
      if (v.size() <= 1) v.push_back(24);
      while (v.size() < 4) v.push_back(0);
      comp += itos(ncomp) + " match " + itos(membits - v[3] - 2) + " " + itos(membits - v[2]) + "\n";
      hcomp += "d= " + itos(ncomp) + " a=*d a*= " + itos(v[1]) + " a+=*c a++ *d=a\n";
      sb = 5 + (membits - v[2]) * 3 / 4;
      ++ncomp;
    }
    if (c0 == 'w') {
      // w1,65,26,223,20,0: ICM-ISSE chain of length N1 over word contexts; a word is a run of bytes c
      // with (c & N4) in N2..N2+N3-1, hashed as hash*N5 + c + 1; N6 halves memory
      if (v.size() <= 1) v.push_back(1);
      if (v.size() <= 2) v.push_back(65);
      if (v.size() <= 3) v.push_back(26);
      if (v.size() <= 4) v.push_back(223);
      if (v.size() <= 5) v.push_back(20);
      if (v.size() <= 6) v.push_back(0);
      comp += itos(ncomp) + " icm " + itos(membits - 6 - v[6]) + "\n";
      for (int i = 1; i < v[1]; ++i) comp += itos(ncomp + i) + " isse " + itos(membits - 6 - v[6]) + " " + itos(ncomp + i - 1) + "\n";
      hcomp += "a=*c a&= " + itos(v[4]) + " a-= " + itos(v[2]) + " a&= 255 a< " + itos(v[3]) + " if\n";
      for (int i = 0; i < v[1]; ++i) {
        hcomp += i == 0 ? "  d= " + itos(ncomp) : std::string("  d++");
        hcomp += " a=*d a*= " + itos(v[5]) + " a+=*c a++ *d=a\n";
      }
      hcomp += "else\n";
      for (int i = v[1] - 1; i > 0; --i) hcomp += "  d= " + itos(ncomp + i - 1) + " a=*d d++ *d=a\n";
      hcomp += "  d= " + itos(ncomp) + " *d=0\nendif\n";
      ncomp += v[1] - 1;
      sb = membits - v[6];
      ++ncomp;
    }
  }
  return hdr + itos(ncomp) + "\n" + comp + hcomp + "halt\n" + pcomp;
}

// compressBlock()'s expansion of "LB[,R,t]" (L = level 0..5): returns the x/0 method it stands for.
// `data` (n bytes, host) is only read for level 5 (the search for periodic structure).
std::string expand_method(const std::string& method, const u8* data, u32 n) {
  if (method.empty()) throw ConfigError("empty method");
  if (!isdigit((unsigned char)method[0])) return method;
  const int arg0 = std::max(lg2(n + 4095) - 20, 0);
  int commas = 0, a[4] = {0, 0, 0, 0};
  for (size_t i = 1; i < method.size() && commas < 4; ++i) {
    if (method[i] == ',' || method[i] == '.') ++commas;
    else if (isdigit((unsigned char)method[i])) a[commas] = a[commas] * 10 + method[i] - '0';
  }
  const unsigned type = commas == 0 ? 512u : (unsigned)(a[1] * 4 + a[2]);
  const int level = method[0] - '0';
  const int doe8 = (int)(type & 2) * 2;
  std::string m = "x" + itos(arg0);
  const std::string htsz = "," + itos(19 + arg0 + (arg0 <= 6)), sasz = "," + itos(21 + arg0);
  if (level == 0) return "0" + itos(arg0) + ",0";
  if (level == 1) {
    if (type < 40) m += ",0";
    else {
      m += "," + itos(1 + doe8) + ",";
      if (type < 80) m += "4,0,1,15";
      else if (type < 128) m += "4,0,2,16";
      else if (type < 256) m += "4,0,2" + htsz;
      else if (type < 960) m += "5,0,3" + htsz;
      else m += "6,0,3" + htsz;
    }
  } else if (level == 2) {
    if (type < 32) m += ",0";
    else {
      m += "," + itos(1 + doe8) + ",";
      if (type < 64) m += "4,0,3" + htsz;
      else m += "4,0,7" + sasz + ",1";
    }
  } else if (level == 3) {
    if (type < 20) m += ",0";
    else if (type < 48) m += "," + itos(1 + doe8) + ",4,0,3" + htsz;
    else if (type >= 640 || (type & 1)) m += "," + itos(3 + doe8) + "ci1";
    else m += "," + itos(2 + doe8) + ",12,0,7" + sasz + ",1c0,0,511i2";
  } else if (level == 4) {
    if (type < 12) m += ",0";
    else if (type < 24) m += "," + itos(1 + doe8) + ",4,0,3" + htsz;
    else if (type < 48) m += "," + itos(2 + doe8) + ",5,0,7" + sasz + "1c0,0,511";
    else if (type < 900) {
      m += "," + itos(doe8) + "ci1,1,1,1,2a";
      if (type & 1) m += "w";
      m += "m";
    } else m += "," + itos(3 + doe8) + "ci1";
  } else {
    if (n && !data) throw ConfigError("levels 5..9 look at the data to choose periodic models: data pointer missing");
    m += "," + itos(doe8);
    if (type & 1) m += "w2c0,1010,255i1"; else m += "w1i1";
    m += "c256ci1,1,1,1,1,1,2a";
    // periodic models: histogram of the gaps between equal bytes
    const int NR = 1 << 12;
    std::vector<int> pt(256, 0), r(NR, 0);
    for (u32 i = 0; i < n; ++i) {
      const int k = (int)i - pt[data[i]];
      if (k > 0 && k < NR) ++r[k];
      pt[data[i]] = (int)i;
    }
    int n1 = (int)n - r[1] - r[2] - r[3];
    for (int i = 0; i < 2; ++i) {
      int period = 0; double score = 0; int t = 0;
      for (int j = 5; j < NR && t < n1; ++j) {
        const double s = r[j] / (256.0 + n1 - t);
        if (s > score) score = s, period = j;
        t += r[j];
      }
      if (period > 4 && score > 0.1) {
        m += "c0,0," + itos(999 + period) + ",255i1";
        if (period <= 255) m += "c0," + itos(period) + "i1";
        n1 -= r[period];
        r[period] = 0;
      } else break;
    }
    m += "c0,2,0,255i1c0,3,0,0,255i1c0,4,0,0,0,255i1mm16ts19t0";
  }
  return m;
}

int copy_out(zpq_ctx* ctx, const void* src, size_t len, void* dst, size_t cap, size_t* out_len, const char* what) {
  if (out_len) *out_len = len;
  if (!dst) return ZPQ_OK;
  if (len > cap) return ctx ? zpq_fail(ctx, ZPQ_ERR_CAPACITY, "%s needs %zu bytes", what, len) : ZPQ_ERR_CAPACITY;
  memcpy(dst, src, len);
  return ZPQ_OK;
}

}  // namespace

// internal entry points (block.hip)
int zpq_build_config(zpq_ctx* ctx, const char* method, const u8* host_data, u32 n, std::string* xmethod, int args[9],
                     std::vector<u8>* header, std::vector<u8>* pcomp) {
  try {
    *xmethod = expand_method(method ? method : "", host_data, n);
    const std::string src = make_config(xmethod->c_str(), args);
    std::string cmd;
    Zpaql(src.c_str(), args).compile(*header, *pcomp, cmd);
    return ZPQ_OK;
  } catch (const ConfigError& e) {
    return zpq_fail(ctx, ZPQ_ERR_METHOD, "%s", e.what());
  }
}

extern "C" {

int zpq_expand_method(zpq_ctx* ctx, const char* method, const uint8_t* data, size_t n, char* out, size_t cap) {
  try {
    const std::string m = expand_method(method ? method : "", data, (u32)n);
    return copy_out(ctx, m.c_str(), m.size() + 1, out, cap, nullptr, "method");
  } catch (const ConfigError& e) {
    return ctx ? zpq_fail(ctx, ZPQ_ERR_METHOD, "%s", e.what()) : ZPQ_ERR_METHOD;
  }
}

int zpq_make_config(zpq_ctx* ctx, const char* method, int32_t args[9], char* out, size_t cap, size_t* out_len) {
  try {
    int a[9];
    const std::string s = make_config(method ? method : "", a);
    for (int i = 0; i < 9; ++i) args[i] = a[i];
    return copy_out(ctx, s.c_str(), s.size() + 1, out, cap, out_len, "config");
  } catch (const ConfigError& e) {
    return ctx ? zpq_fail(ctx, ZPQ_ERR_METHOD, "%s", e.what()) : ZPQ_ERR_METHOD;
  }
}

// ---- the models behind Compressor::startBlock(int level) (ZSFX/libzpaq.h:1346) ------------------------------------------
// libzpaq keeps them as a byte array `models[]` in Compressor::startBlock(int), which is in the part of libzpaq.cpp the
// snapshot lacks.  They are min.cfg, mid.cfg and max.cfg of the ZPAQ distribution (public domain); kept here as config
// SOURCE and compiled by the same compiler as every other config, which gives libzpaq's 28 / 71 / 198 header bytes
// (tests/test_config_cpu.py holds the byte arrays and also runs the sources through the reference Compiler).
static const char* const kBuiltinModel[3] = {
    // level 1: min.cfg
    "comp 1 2 0 0 2 (hh hm ph pm n)\n"
    "  0 icm 16\n"
    "  1 isse 19 0\n"
    "hcomp\n"
    "  *b=a a=0 (save in rotating buffer M)\n"
    "  d=0 hash b-- hash *d=a (order 2 hash for icm)\n"
    "  d++ b-- hash b-- hash *d=a (order 4 for isse)\n"
    "  halt\n"
    "end\n",
    // level 2: mid.cfg
    "comp 3 3 0 0 8 (hh hm ph pm n)\n"
    "  0 icm 5\n"
    "  1 isse 13 0\n"
    "  2 isse 17 1\n"
    "  3 isse 18 2\n"
    "  4 isse 18 3\n"
    "  5 isse 19 4\n"
    "  6 match 22 24\n"
    "  7 mix 16 0 7 24 255\n"
    "hcomp\n"
    "  c++ *c=a b=c a=0 (save in rotating buffer M)\n"
    "  d= 1 hash *d=a   (orders 1...5 for isse)\n"
    "  b-- d++ hash *d=a\n"
    "  b-- d++ hash *d=a\n"
    "  b-- d++ hash *d=a\n"
    "  b-- d++ hash *d=a\n"
    "  b-- d++ hash b-- hash *d=a (order 7 for match)\n"
    "  d++ a=*c a<<= 8 *d=a (order 1 for mix)\n"
    "  halt\n"
    "end\n",
    // level 3: max.cfg
    "comp 5 9 0 0 22 (hh hm ph pm n)\n"
    "  0 const 160\n"
    "  1 icm 5  (orders 0-6)\n"
    "  2 isse 13 1 (sizebits j)\n"
    "  3 isse 16 2\n"
    "  4 isse 18 3\n"
    "  5 isse 19 4\n"
    "  6 isse 19 5\n"
    "  7 isse 20 6\n"
    "  8 match 22 24\n"
    "  9 icm 17 (order 0 word)\n"
    "  10 isse 19 9 (order 1 word)\n"
    "  11 icm 13 (sparse with gaps 1-3)\n"
    "  12 icm 13\n"
    "  13 icm 13\n"
    "  14 icm 14 (pic)\n"
    "  15 mix 16 0 15 24 255 (mix orders 1 and 0)\n"
    "  16 mix 8 0 16 10 255 (including last mixer)\n"
    "  17 mix2 0 15 16 24 0\n"
    "  18 sse 8 17 32 255 (order 0)\n"
    "  19 mix2 8 17 18 16 255\n"
    "  20 sse 16 19 32 255 (order 1)\n"
    "  21 mix2 0 19 20 16 0\n"
    "hcomp\n"
    "  c++ *c=a b=c a=0 (save in rotating buffer)\n"
    "  d= 2 hash *d=a b-- (orders 1,2,3,4,5,7)\n"
    "  d++ hash *d=a b--\n"
    "  d++ hash *d=a b--\n"
    "  d++ hash *d=a b--\n"
    "  d++ hash *d=a b--\n"
    "  d++ hash b-- hash *d=a b--\n"
    "  d++ hash *d=a b-- (match, order 8)\n"
    "  d++ a=*c a&~ 32 (lowercase words)\n"
    "  a> 64 if\n"
    "    a< 91 if (if a-z)\n"
    "      d++ hashd d-- (update order 1 word hash)\n"
    "      *d<>a a+=*d a*= 20 *d=a (order 0 word hash)\n"
    "      jmp 9\n"
    "    endif\n"
    "  endif\n"
    "  (else not a letter)\n"
    "    a=*d a== 0 ifnot (move word order 0 to 1)\n"
    "      d++ *d=a d--\n"
    "    endif\n"
    "    *d=0  (clear order 0 word hash)\n"
    "  (end else)\n"
    "  d++\n"
    "  d++ b=c b-- a=0 hash *d=a (sparse 2)\n"
    "  d++ b-- a=0 hash *d=a (sparse 3)\n"
    "  d++ b-- a=0 hash *d=a (sparse 4)\n"
    "  d++ a=b a-= 212 b=a a=0 hash\n"
    "    *d=a b<>a a-= 216 b<>a a=*b a&= 60 hashd (pic)\n"
    "  d++ a=*c a<<= 9 *d=a (mix)\n"
    "  d++\n"
    "  d++\n"
    "  d++ d++\n"
    "  d++ *d=a (sse)\n"
    "  halt\n"
    "end\n"};

const char* zpq_builtin_model_source(int level) { return level >= 1 && level <= 3 ? kBuiltinModel[level - 1] : nullptr; }

int zpq_builtin_model(int level, uint8_t* header, size_t header_cap, size_t* header_len) {
  if (level < 1 || level > 3) return ZPQ_ERR_ARG;
  uint8_t p[8];
  size_t pl = 0;
  return zpq_compile_config(nullptr, kBuiltinModel[level - 1], nullptr, header, header_cap, header_len, p, sizeof p, &pl);
}

int zpq_compile_config(zpq_ctx* ctx, const char* source, const int32_t* args, uint8_t* header, size_t header_cap,
                       size_t* header_len, uint8_t* pcomp, size_t pcomp_cap, size_t* pcomp_len) {
  try {
    int a[9] = {0};
    if (args) for (int i = 0; i < 9; ++i) a[i] = args[i];
    std::vector<u8> h, p;
    std::string cmd;
    Zpaql(source ? source : "", a).compile(h, p, cmd);
    int rc = copy_out(ctx, h.data(), h.size(), header, header_cap, header_len, "header");
    if (rc) return rc;
    return copy_out(ctx, p.data(), p.size(), pcomp, pcomp_cap, pcomp_len, "pcomp");
  } catch (const ConfigError& e) {
    return ctx ? zpq_fail(ctx, ZPQ_ERR_FORMAT, "%s", e.what()) : ZPQ_ERR_FORMAT;
  }
}

}  // extern "C"
