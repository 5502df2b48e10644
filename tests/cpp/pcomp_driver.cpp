// Test infrastructure: libzpaq::Decompresser::pcomp() of the shim over an archive, without decoding a segment (no GPU needed for
// blocks without a context model; behind a model the head of the coded stream is decoded on the device):
// per block "<filename>|<hex of what pcomp() wrote, empty if it returned false>".
#include <stdio.h>
#include <stdlib.h>

#include <stdexcept>
#include <string>

#include "libzpaq_gpu.h"

void libzpaq::error(const char* msg) { throw std::runtime_error(msg); }

struct FileReader : libzpaq::Reader {
  FILE* f;
  explicit FileReader(FILE* g) : f(g) {}
  int get() { return getc(f); }
  int read(char* buf, int n) { return (int)fread(buf, 1, n, f); }
};

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  FILE* f = fopen(argv[1], "rb");
  if (!f) return 3;
  try {
    FileReader in(f);
    libzpaq::Decompresser d;
    d.setInput(&in);
    while (d.findBlock()) {
      libzpaq::StringBuffer name, pc;
      bool first = true;
      while (d.findFilename(&name)) {
        d.readComment();
        d.readSegmentEnd();                       // the segment is read, not decoded
        if (first) {
          fwrite(name.c_str(), 1, name.size(), stdout); printf("|");        // (a StringBuffer is not NUL-terminated)
          try {
            const bool have = d.pcomp(&pc);          // (behind a context model this decodes the head of the stream on the device)
            if (have) for (size_t i = 0; i < pc.size(); ++i) printf("%02x", pc.c_str()[i] & 255);
          } catch (std::exception& e) { printf("error: %s", e.what()); }
          printf("\n");
          first = false;
        }
        name.resize(0);
      }
    }
  } catch (std::exception& e) { fprintf(stderr, "error: %s\n", e.what()); return 1; }
  return 0;
}
