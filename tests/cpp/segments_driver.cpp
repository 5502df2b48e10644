// Test infrastructure: blocks of several segments through the shim's Compressor / Decompresser classes (ZSFX/libzpaq.h:1243-1264,
// 1340-1371), the way a streaming archiver drives them: one block, a segment per file.
//   segments_driver c <archive> <level 1..3 | @config-file> <pre> <file>...   pre: "none" or "delta" (the caller-side transform the
//                                                                             config's post-processor undoes: running differences)
//   segments_driver d <archive> <outdir>     every segment decoded to <outdir>/<block>.<segment>; one line each:
//                                            <block> <segment> <name> <bytes> <stored sha1 or -> <sha1 of the decoded bytes>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <stdexcept>
#include <string>
#include <vector>

#include "libzpaq_gpu.h"

void libzpaq::error(const char* msg) { throw std::runtime_error(msg); }

namespace {
struct FileReader : libzpaq::Reader {
  FILE* f;
  explicit FileReader(FILE* g) : f(g) {}
  int get() { return getc(f); }
  int read(char* buf, int n) { return (int)fread(buf, 1, n, f); }
};
struct FileWriter : libzpaq::Writer {
  FILE* f;
  explicit FileWriter(FILE* g) : f(g) {}
  void put(int c) { putc(c, f); }
  void write(const char* buf, int n) { fwrite(buf, 1, n, f); }
};
struct MemReader : libzpaq::Reader {
  const std::vector<unsigned char>& v; size_t i = 0;
  explicit MemReader(const std::vector<unsigned char>& w) : v(w) {}
  int get() { return i < v.size() ? v[i++] : -1; }
  int read(char* buf, int n) { size_t k = v.size() - i; if ((size_t)n < k) k = n; memcpy(buf, v.data() + i, k); i += k; return (int)k; }
};
std::vector<unsigned char> slurp(const char* path) {
  std::vector<unsigned char> v;
  FILE* f = fopen(path, "rb");
  if (!f) throw std::runtime_error(std::string("cannot read ") + path);
  unsigned char buf[65536]; size_t r;
  while ((r = fread(buf, 1, sizeof buf, f)) > 0) v.insert(v.end(), buf, buf + r);
  fclose(f);
  return v;
}
std::string hex(const char* p, int n) { std::string s; char b[3]; for (int i = 0; i < n; ++i) { snprintf(b, 3, "%02x", p[i] & 255); s += b; } return s; }
}  // namespace

int main(int argc, char** argv) {
  try {
    if (argc >= 6 && !strcmp(argv[1], "c")) {
      FILE* fo = fopen(argv[2], "wb");
      if (!fo) return 3;
      FileWriter out(fo);
      libzpaq::Compressor co;
      co.setOutput(&out);
      co.writeTag();
      if (argv[3][0] == '@') {
        const std::vector<unsigned char> cfg = slurp(argv[3] + 1);
        const std::string text(cfg.begin(), cfg.end());
        int args[9] = {0};
        co.startBlock(text.c_str(), args);
      } else co.startBlock(atoi(argv[3]));
      const bool delta = !strcmp(argv[4], "delta");
      unsigned prev = 0;                                      // the post-processor's `c` carries on from segment to segment
      for (int i = 5; i < argc; ++i) {
        const std::vector<unsigned char> data = slurp(argv[i]);
        libzpaq::SHA1 sha;
        if (!data.empty()) sha.write((const char*)data.data(), (int64_t)data.size());
        char digest[20];
        memcpy(digest, sha.result(), 20);
        std::vector<unsigned char> t(data);
        if (delta) for (size_t k = 0; k < t.size(); ++k) { const unsigned x = data[k]; t[k] = (unsigned char)(x - prev); prev = x; }
        const char* name = strrchr(argv[i], '/') ? strrchr(argv[i], '/') + 1 : argv[i];
        co.startSegment(name, std::to_string(data.size()).c_str());
        if (i == 5) co.postProcess();
        MemReader in(t);
        co.setInput(&in);
        while (co.compress(1 << 15)) {}
        co.endSegment(digest);
      }
      co.endBlock();
      fclose(fo);
      return 0;
    }
    if (argc == 4 && !strcmp(argv[1], "d")) {
      FILE* f = fopen(argv[2], "rb");
      if (!f) return 3;
      FileReader in(f);
      libzpaq::Decompresser d;
      d.setInput(&in);
      for (int blk = 0; d.findBlock(); ++blk) {
        libzpaq::StringBuffer name;
        for (int seg = 0; d.findFilename(&name); ++seg) {
          d.readComment();
          const std::string path = std::string(argv[3]) + "/" + std::to_string(blk) + "." + std::to_string(seg);
          FILE* fo = fopen(path.c_str(), "wb");
          if (!fo) return 4;
          FileWriter out(fo);
          libzpaq::SHA1 sha;
          d.setOutput(&out);
          d.setSHA1(&sha);
          while (d.decompress(1 << 14)) {}                    // in pieces, as an extractor with a progress display would
          const uint64_t bytes = sha.usize();
          char rec[21];
          d.readSegmentEnd(rec);
          fclose(fo);
          printf("%d %d %s %llu %s %s\n", blk, seg, std::string(name.c_str(), name.size()).c_str(), (unsigned long long)bytes,
                 rec[0] ? hex(rec + 1, 20).c_str() : "-", hex(sha.result(), 20).c_str());
          name.resize(0);
        }
      }
      fclose(f);
      return 0;
    }
  } catch (std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 2;
}
