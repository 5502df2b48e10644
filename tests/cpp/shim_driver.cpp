// Drives the libzpaq-shaped shim exactly the way Jidac's worker threads drive libzpaq
// (one StringBuffer + compressBlock per block per thread, ZSFX/zsfx.cpp:1783-1801 mirrors the
// extract side).  Usage: shim_driver <in_file> <method> <nthreads> <block_bytes> <out_prefix>
// Writes <out_prefix>.zpaq (blocks in input order), <out_prefix>.back (decompress of that) and
// prints the SHA-1/SHA-256 of the input as computed through libzpaq::SHA1 / SHA256.
#include <stdio.h>
#include <stdlib.h>

#include <string.h>

#include <algorithm>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "libzpaq_gpu.h"

void libzpaq::error(const char* msg) { throw std::runtime_error(msg); }

struct FileWriter : libzpaq::Writer {
  FILE* f;
  explicit FileWriter(FILE* g) : f(g) {}
  void put(int c) { putc(c, f); }
  void write(const char* buf, int n) { fwrite(buf, 1, n, f); }
};

struct FileReader : libzpaq::Reader {
  FILE* f;
  explicit FileReader(FILE* g) : f(g) {}
  int get() { return getc(f); }
  int read(char* buf, int n) { return (int)fread(buf, 1, n, f); }
};

// The extract side exactly as decompressThread drives it (ZSFX/zsfx.cpp:1783-1834): one Decompresser per block,
// findBlock / findFilename / readComment / decompress(1<<14) ... / readSegmentEnd, SHA-1 of the output through
// libzpaq::SHA1.  Blocks whose declared size exceeds max_usize are skipped with readSegmentEnd alone.
static int extract_mode(const char* path, size_t max_usize) {
  FILE* f = fopen(path, "rb");
  if (!f) return 3;
  FileReader in(f);
  try {
    for (;;) {
      libzpaq::Decompresser d;
      d.setInput(&in);
      double mem = 0;
      if (!d.findBlock(&mem)) break;
      libzpaq::StringBuffer name, comment, out;
      while (d.findFilename(&name)) {
        d.readComment(&comment);
        const size_t declared = strtoull(std::string(comment.c_str(), comment.size()).c_str(), 0, 10);
        libzpaq::SHA1 sha;
        char rec[21];
        bool skipped = declared > max_usize;
        if (!skipped) {
          d.setOutput(&out);
          d.setSHA1(&sha);
          while (d.decompress(1 << 14)) {}
        }
        d.readSegmentEnd(rec);
        printf("%s|%zu|%s|", std::string(name.c_str(), name.size()).c_str(), out.size(), skipped ? "skipped" : "decoded");
        const char* h = sha.result();
        for (int i = 0; i < 20; ++i) printf("%02x", skipped ? 0 : (unsigned char)h[i]);
        printf("|%d|", rec[0]);
        for (int i = 1; i <= 20; ++i) printf("%02x", (unsigned char)rec[i]);
        printf("|%.0f\n", mem);
      }
    }
  } catch (std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    fclose(f);
    return 5;
  }
  fclose(f);
  return 0;
}

// libzpaq::Compressor driven as ZSFX/libzpaq.h:426-531 documents: a block with the given config, `nseg` segments of equal
// shares of the input, SHA-1 of every segment.  Writes <out>.
static int compressor_mode(const char* in_path, const char* config_path, int nseg, const char* out_path) {
  std::vector<char> data, cfg;
  for (int k = 0; k < 2; ++k) {
    if (k && strncmp(config_path, "level:", 6) == 0) break;
    FILE* f = fopen(k ? config_path : in_path, "rb");
    if (!f) return 3;
    char buf[1 << 16]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) (k ? cfg : data).insert((k ? cfg : data).end(), buf, buf + r);
    fclose(f);
  }
  cfg.push_back(0);
  FILE* fo = fopen(out_path, "wb");
  if (!fo) return 3;
  FileWriter w(fo);
  try {
    libzpaq::Compressor co;
    co.setOutput(&w);
    co.writeTag();
    int args[9] = {0};
    if (strncmp(config_path, "level:", 6) == 0) co.startBlock(atoi(config_path + 6));     // libzpaq's built-in models 1..3
    else co.startBlock(cfg.data(), args);
    const size_t share = (data.size() + nseg - 1) / std::max(1, nseg);
    for (int k = 0; k < nseg; ++k) {
      const size_t lo = std::min(data.size(), k * share), hi = std::min(data.size(), lo + share);
      libzpaq::StringBuffer sb;
      sb.write(data.data() + lo, (int)(hi - lo));
      char fn[32]; snprintf(fn, sizeof fn, "seg%d", k);
      co.startSegment(fn, "c");
      co.setInput(&sb);
      if (k == 0) co.postProcess();
      while (co.compress(1 << 15)) {}
      libzpaq::SHA1 sha; sha.write(data.data() + lo, (int64_t)(hi - lo));
      co.endSegment(sha.result());
    }
    co.endBlock();
  } catch (std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    fclose(fo);
    return 5;
  }
  fclose(fo);
  return 0;
}

// N threads, each with its own archive copy and Decompresser, all decoding at the same time (the batchers coalesce them);
// every thread decodes every block of the archive with libzpaq::decompress and compares with the expectation.
static int parallel_extract_mode(const char* arc_path, const char* want_path, int nthreads) {
  std::vector<char> arc, want;
  for (int k = 0; k < 2; ++k) {
    FILE* f = fopen(k ? want_path : arc_path, "rb");
    if (!f) return 3;
    char buf[1 << 16]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) (k ? want : arc).insert((k ? want : arc).end(), buf, buf + r);
    fclose(f);
  }
  std::vector<std::string> errs(nthreads);
  std::vector<std::thread> th;
  for (int t = 0; t < nthreads; ++t)
    th.emplace_back([&, t] {
      try {
        libzpaq::StringBuffer in, out;
        in.write(arc.data(), (int)arc.size());
        libzpaq::decompress(&in, &out);
        if (out.size() != want.size() || memcmp(out.c_str(), want.data(), want.size()) != 0) errs[t] = "output differs";
      } catch (std::exception& e) { errs[t] = e.what(); }
    });
  for (auto& t : th) t.join();
  for (auto& e : errs) if (!e.empty()) { fprintf(stderr, "thread error: %s\n", e.c_str()); return 4; }
  printf("parallel extract ok\n");
  return 0;
}

int main(int argc, char** argv) {
  if (argc == 4 && std::string(argv[1]) == "--extract") return extract_mode(argv[2], strtoull(argv[3], 0, 10));
  if (argc == 6 && std::string(argv[1]) == "--compressor") return compressor_mode(argv[2], argv[3], atoi(argv[4]), argv[5]);
  if (argc == 5 && std::string(argv[1]) == "--parallel-extract") return parallel_extract_mode(argv[2], argv[3], atoi(argv[4]));
  if (argc < 6) return 2;
  std::vector<char> data;
  {
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    char buf[1 << 16]; size_t r;
    while ((r = fread(buf, 1, sizeof buf, f)) > 0) data.insert(data.end(), buf, buf + r);
    fclose(f);
  }
  const std::string method = argv[2];
  const int nthreads = atoi(argv[3]);
  const size_t block = strtoull(argv[4], 0, 10);
  const std::string prefix = argv[5];
  try {
    const size_t nblocks = data.empty() ? 1 : (data.size() + block - 1) / block;
    std::vector<libzpaq::StringBuffer> outs(nblocks);
    std::vector<std::thread> th;
    std::vector<std::string> errs(nthreads);
    for (int t = 0; t < nthreads; ++t)
      th.emplace_back([&, t] {
        try {
          for (size_t b = t; b < nblocks; b += nthreads) {
            libzpaq::StringBuffer sb;
            const size_t lo = b * block, hi = std::min(data.size(), lo + block);
            sb.write(data.data() + lo, (int)(hi - lo));
            char fn[64]; snprintf(fn, sizeof fn, "jDC20240101000000d%010zu", b + 1);
            libzpaq::compressBlock(&sb, &outs[b], method.c_str(), fn, "jDC\x01", true);
          }
        } catch (std::exception& e) { errs[t] = e.what(); }
      });
    for (auto& t : th) t.join();
    for (auto& e : errs) if (!e.empty()) { fprintf(stderr, "worker error: %s\n", e.c_str()); return 4; }
    FILE* fo = fopen((prefix + ".zpaq").c_str(), "wb");
    for (auto& o : outs) fwrite(o.c_str(), 1, o.size(), fo);
    fclose(fo);
    // round trip through libzpaq::decompress
    libzpaq::StringBuffer arc;
    for (auto& o : outs) arc.write(o.c_str(), (int)o.size());
    FILE* fb = fopen((prefix + ".back").c_str(), "wb");
    FileWriter w(fb);
    libzpaq::decompress(&arc, &w);
    fclose(fb);
    libzpaq::SHA1 s1; s1.write(data.data(), (int64_t)data.size());
    libzpaq::SHA256 s2; for (char c : data) s2.put(c);
    const char* a = s1.result(); const char* b = s2.result();
    for (int i = 0; i < 20; ++i) printf("%02x", (unsigned char)a[i]);
    printf(" ");
    for (int i = 0; i < 32; ++i) printf("%02x", (unsigned char)b[i]);
    printf("\n");
    // an unsupported method must surface through libzpaq::error, not be approximated
    try {
      libzpaq::StringBuffer sb, o; sb.write("hello", 5);
      libzpaq::compressBlock(&sb, &o, "x4,1,4,80,3,24", 0, 0, true);   // LZ77 with a secondary context of 80 bytes: beyond what the engine serves (64)
      printf("unsupported: NOT refused\n");
    } catch (std::exception& e) { printf("unsupported: refused (%s)\n", e.what()); }
  } catch (std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 5;
  }
  return 0;
}
