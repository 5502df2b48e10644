// Drives the libzpaq-shaped shim exactly the way Jidac's worker threads drive libzpaq
// (one StringBuffer + compressBlock per block per thread, ZSFX/zsfx.cpp:1783-1801 mirrors the
// extract side).  Usage: shim_driver <in_file> <method> <nthreads> <block_bytes> <out_prefix>
// Writes <out_prefix>.zpaq (blocks in input order), <out_prefix>.back (decompress of that) and
// prints the SHA-1/SHA-256 of the input as computed through libzpaq::SHA1 / SHA256.
#include <stdio.h>
#include <stdlib.h>

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

int main(int argc, char** argv) {
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
      libzpaq::compressBlock(&sb, &o, "3", 0, 0, true);   // level 3: byte-aligned LZ77 or BWT front end
      printf("unsupported: NOT refused\n");
    } catch (std::exception& e) { printf("unsupported: refused (%s)\n", e.what()); }
  } catch (std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 5;
  }
  return 0;
}
