import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import datagen, orc
from zpaqfranz_amd import Engine
eng = Engine(0)
def ofr(files):
    out=[]
    for fi,f in enumerate(files):
        off=0
        for ln in orc.chunk(f):
            out.append((fi,off,ln)); off+=ln
    return out
files = [b"", b"a", bytes(5000), bytes(3 << 20), datagen.text_like(70000, 1), b"", datagen.random_bytes(4096, 2),
         datagen.random_bytes(4097, 3), b"ab" * 400000, datagen.binary_like(1 << 20, 4), b"x" * 4095,
         datagen.mixed((1 << 20) + 12345, 5), datagen.mixed(3 * (1 << 20), 6), datagen.text_like((1 << 20) - 1, 7),
         datagen.text_like(1 << 20, 8), datagen.random_bytes((2 << 20) + 1, 9)]
for sub in (files, files[8:], files[12:], files[14:], files[3:4]+files[15:], files[:4]+files[15:]):
    a = eng.fragment_files(sub); b = ofr(sub)
    print(len(sub), len(a), len(b), a == b, [x for x in b if x not in a][:3], [x for x in a if x not in b][:3], flush=True)
