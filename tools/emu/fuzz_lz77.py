"""Random inputs and settings through the lz77_full_emu emulation (tests/cpp/lz77_full_emu.cpp: the engine's DEVICE source -- and for the
LZ77 encoder its host code too -- on the CPU fibre emulator) against the oracle.  usage: python fuzz_lz77.py <seed> <seconds>
(the LZ77 switches ZPQ_LZ_SEG / ZPQ_LZ_DIRECT select the path, as on the GPU)."""
import os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(tempfile.gettempdir(), "lz77_full_emu.so")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-Wno-unused", "-Wno-unused-value",
                       "-I" + os.path.join(ROOT, "zpaqfranz_amd", "csrc"), "-I" + os.path.join(ROOT, "tests", "cpp"), "-I" + os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tests", "cpp", "lz77_full_emu.cpp"), "-o", SO])
import ctypes as C, sys, time
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, datagen, orc
L=C.CDLL(SO)
L.lz77_full_emu.restype=C.c_long
L.lz77_full_emu.argtypes=[C.c_char_p,C.c_uint32,C.POINTER(C.c_int32),C.c_void_p,C.c_uint32,C.c_char_p,C.c_uint32]
rng=np.random.default_rng(int(sys.argv[1]))
def gen():
    kind=rng.integers(0,7)
    n=int(rng.integers(0,30000))
    if kind==0: return datagen.text_like(n,int(rng.integers(1,1000)))
    if kind==1: return datagen.binary_like(n,int(rng.integers(1,1000)))
    if kind==2: return datagen.mixed(n,int(rng.integers(1,1000)))
    if kind==3: return bytes(rng.integers(0,256,n,dtype=np.uint8))
    if kind==4:
        p=bytes(rng.integers(0,256,int(rng.integers(1,50)),dtype=np.uint8)); return (p*(n//len(p)+1))[:n]
    if kind==5:
        a=bytearray(datagen.text_like(n,3)); 
        for _ in range(int(rng.integers(0,20))):
            if n>100:
                s=int(rng.integers(0,n-50)); d=int(rng.integers(0,n-50)); l=int(rng.integers(1,50)); a[d:d+l]=a[s:s+l]
        return bytes(a)
    return bytes(int(rng.integers(0,3)) for _ in range(n))
bad=0; t0=time.time(); cases=0
while time.time()-t0 < float(sys.argv[2]):
    a0=int(rng.choice([0,1,2,3,4,5,6]))      # (7: blocks of 128 MiB are refused by design); mm=int(rng.integers(4,9)); lb=int(rng.integers(0,4)); ht=int(rng.integers(12,17))
    if ht-a0>=21: continue
    args=[a0,1,mm,0,lb,ht]
    b=gen(); n=len(b)
    try: want=orc.lz77_encode(b,args)
    except Exception as ex: continue
    cap=(n+n//8+1024+15)&~15; out=np.zeros(cap,dtype=np.uint8); err=C.create_string_buffer(400)
    r=L.lz77_full_emu(b+bytes(64),n,(C.c_int32*9)(*(args+[0]*9)[:9]),out.ctypes.data,cap,err,400)
    cases+=1
    if r<0 or bytes(out[:r])!=want:
        bad+=1; print("MISMATCH",args,n,r,err.value.decode()[:200]); open("/tmp/fuzz_fail_%d.bin"%cases,"wb").write(b)
print("env",{k:v for k,v in os.environ.items() if k.startswith("ZPQ_")},"cases",cases,"bad",bad)
