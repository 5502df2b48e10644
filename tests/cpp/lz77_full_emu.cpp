// Test infrastructure: zpq_lz77_encode_dev() -- host code and every kernel of zpaqfranz_amd/csrc/lz77_enc.hip -- on the CPU, over
// fake_hip.h.  The environment selects the path exactly as on the GPU (ZPQ_LZ_SEG, ZPQ_LZ_DIRECT); the switches are read once
// per process, so tests run one process per setting.
#include "fake_hip.h"

#define ZPQ_EMU_WALK_ONLY      // (skips the real headers)
#define ZPQ_EMU_FULL           // (... but keeps everything behind the parse kernels)
#include "lz77_enc.hip"

int zpq_lz77_sa_encode(zpq_ctx* ctx, zpq_lz77_job*, const size_t*, size_t) { return zpq_fail(ctx, ZPQ_ERR_METHOD, "suffix-array jobs are not part of this emulation"); }
int zpq_lz77_pack2_launch(zpq_ctx* ctx, const zpq_lzjob_dev*, const u32*, size_t, u32) { return zpq_fail(ctx, ZPQ_ERR_METHOD, "level-2 packing (lz77_sa.hip) is not part of this emulation"); }

// in: n bytes + 64 readable bytes; out: cap bytes.  Returns the length of the code stream or a negative status.
extern "C" long lz77_full_emu(const u8* in, u32 n, const int32_t args[9], u8* out, u32 cap, char* err, u32 err_cap) {
  zpq_ctx ctx;
  zpq_lz77_job j;
  memset(&j, 0, sizeof j);
  j.d_in = in; j.n = n;
  for (int k = 0; k < 9; ++k) j.args[k] = args[k];
  j.d_out = out; j.out_cap = cap;
  g_emu_launch_error = nullptr;
  const int rc = zpq_lz77_encode_dev(&ctx, &j, 1);
  if (rc != 0 || g_emu_launch_error) {
    if (err && err_cap) snprintf(err, err_cap, "rc %d: %s %s", rc, ctx.err.c_str(), g_emu_launch_error ? g_emu_launch_error : "");
    return rc ? rc : -100;
  }
  return (long)j.out_len;
}
