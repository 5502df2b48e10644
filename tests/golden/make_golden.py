#!/usr/bin/env python3
"""Regenerates tests/golden/* from the reference's own fixture AUTOTEST/sha256.zpaq, decoded
with the REAL reference decoder (oracle/_ref/libzpaqref.so = /root/reference/ZSFX/libzpaq.cpp
compiled in place).  Run in the build container (needs /root/reference):

    make -C oracle && python tests/golden/make_golden.py

Outputs (all derived data, no reference source):
  sha256.zpaq          the fixture archive itself (158 239 B; sha256 d90223fa...5400,
                       AUTOTEST/README.txt:37)
  dblock_plain.xz      plaintext of the d block (9 473 560 B = 256 files x 37 000 B + 1 560 B trailer)
  hblock_plain.bin     plaintext of the h block (bsize[4] + 388 x {sha1[20], usize[4]})
  iblock{1,2,3}.bin    plaintext of the three i blocks (file table)
  blocks.json          byte offsets / names / comments of the 6 blocks as parsed by the reference
"""
import hashlib, json, lzma, os, shutil, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import orc

SRC = "/root/reference/AUTOTEST/sha256.zpaq"
arc = open(SRC, "rb").read()
assert hashlib.sha256(arc).hexdigest() == "d90223faee2878d7854b9438864b4856a3c1f920c34efb8c136a8949b54e5400"
shutil.copyfile(SRC, os.path.join(HERE, "sha256.zpaq"))
off, blocks = 0, []
while off < len(arc):
    b = orc.ref_decompress_block(arc[off:], 20_000_000)
    assert b["sha1_ok"] == 1
    blocks.append(dict(offset=off, size=b["consumed"], filename=b["filename"].decode("latin1"),
                       comment=b["comment"].decode("latin1"), usize=len(b["data"]),
                       sha1=b["sha1"][1:].hex()))
    kind = b["filename"][17:18].decode()
    idx = int(b["filename"][18:])
    if kind == "d":
        open(os.path.join(HERE, "dblock_plain.xz"), "wb").write(lzma.compress(b["data"], preset=9 | lzma.PRESET_EXTREME))
    elif kind == "h":
        open(os.path.join(HERE, "hblock_plain.bin"), "wb").write(b["data"])
    elif kind == "i":
        open(os.path.join(HERE, "iblock%d.bin" % idx), "wb").write(b["data"])
    off += b["consumed"]
json.dump(blocks, open(os.path.join(HERE, "blocks.json"), "w"), indent=1)
print(json.dumps(blocks, indent=1))

# ZSFX/zsfx.zpaq, ZSFX/zsfx32.zpaq: one streaming block each, method "5" (23 components, no PCOMP).
# Golden vectors for makeConfig/compressBlock level 5 and the whole context-mixing encoder.
for name in ("zsfx.zpaq", "zsfx32.zpaq"):
    a = open("/root/reference/ZSFX/" + name, "rb").read()
    shutil.copyfile("/root/reference/ZSFX/" + name, os.path.join(HERE, name))
    b = orc.ref_decompress_block(a, 1 << 20)
    assert b["consumed"] == len(a) and b["sha1_ok"] == 1, (b["consumed"], len(a), b["sha1_ok"])
    open(os.path.join(HERE, name.replace(".zpaq", "_plain.xz")), "wb").write(lzma.compress(b["data"], preset=9 | lzma.PRESET_EXTREME))
    print(name, len(a), "->", len(b["data"]), b["filename"], b["comment"])
