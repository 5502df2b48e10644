"""Random inputs and settings through the frag_emu emulation (tests/cpp/frag_emu.cpp: the engine's DEVICE source -- and for the
LZ77 encoder its host code too -- on the CPU fibre emulator) against the oracle.  usage: python fuzz_fragmenter.py <seed> <seconds>
(the LZ77 switches ZPQ_LZ_SEG / ZPQ_LZ_DIRECT select the path, as on the GPU)."""
import os, subprocess, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SO = os.path.join(tempfile.gettempdir(), "frag_emu.so")
subprocess.check_call(["/opt/rocm/lib/llvm/bin/clang++", "-O1", "-std=c++17", "-fPIC", "-shared", "-Wno-unknown-pragmas", "-Wno-unused", "-Wno-unused-value",
                       "-I" + os.path.join(ROOT, "zpaqfranz_amd", "csrc"), "-I" + os.path.join(ROOT, "tests", "cpp"), "-I" + os.path.join(ROOT, "include"),
                       os.path.join(ROOT, "tests", "cpp", "frag_emu.cpp"), "-o", SO])
import ctypes as C, sys, time
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, datagen, orc
L=C.CDLL(SO)
L.frag_emu.restype=C.c_long
L.frag_emu.argtypes=[C.c_char_p,C.POINTER(C.c_uint64),C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint32,C.c_uint64,C.c_uint32,C.c_uint64,C.c_void_p,C.c_void_p,C.c_void_p,C.c_void_p,C.c_uint64,C.c_char_p,C.c_uint32]
rng=np.random.default_rng(int(sys.argv[1]))
def gen(n):
    kind=rng.integers(0,6)
    if kind==0: return datagen.text_like(n,int(rng.integers(1,1000)))
    if kind==1: return datagen.binary_like(n,int(rng.integers(1,1000)))
    if kind==2: return datagen.mixed(n,int(rng.integers(1,1000)))
    if kind==3: return bytes(rng.integers(0,256,n,dtype=np.uint8))
    if kind==4:
        p=bytes(rng.integers(0,256,int(rng.integers(1,50)),dtype=np.uint8)); return (p*(n//len(p)+1))[:n]
    return bytes([int(rng.integers(0,256))])*n
t0=time.time(); cases=bad=0
while time.time()-t0<float(sys.argv[2]):
    frag=int(rng.integers(0,9)); minf=64<<frag; maxf=8128<<frag
    if rng.integers(0,3)==0: minf=int(rng.integers(64,5000)); maxf=int(rng.integers(minf, 20*minf))
    nfiles=int(rng.integers(1,7))
    files=[gen(int(rng.integers(0,300000))) if rng.integers(0,6) else b"" for _ in range(nfiles)]
    rep=None
    if nfiles>2 and rng.integers(0,2):
        files[-1]=files[0]; rep=list(range(nfiles)); rep[-1]=0
        if len(files[0])<minf: rep=None
    seg=int(rng.choice([16384,32768,65536,86016,262144])); seg=max(seg, 4*minf//4096*4096+4096)
    budget=int(rng.choice([1000,4096,70000,262144,1<<40])); waves=int(rng.integers(1,4))
    off=[0]
    for f in files: off.append(off[-1]+len(f))
    cap=sum(len(f)//minf+1 for f in files)+4
    fo=np.zeros(cap,dtype=np.uint64); fl=np.zeros(cap,dtype=np.uint32); ff=np.zeros(cap,dtype=np.uint32); err=C.create_string_buffer(256)
    reparr=np.array(rep,dtype=np.uint32) if rep is not None else None
    r=L.frag_emu(b"".join(files)+bytes(64),(C.c_uint64*len(off))(*off),nfiles,frag,minf,maxf,seg,waves,budget,reparr.ctypes.data if rep is not None else None,fo.ctypes.data,fl.ctypes.data,ff.ctypes.data,cap,err,256)
    want=[]
    for fi,f in enumerate(files):
        o=0
        for ln in orc.chunk(f,frag,minf,maxf): want.append((fi,o,ln)); o+=ln
    got=[(int(ff[i]),int(fo[i])-off[int(ff[i])],int(fl[i])) for i in range(max(r,0))]
    cases+=1
    if r<0 or got!=want:
        bad+=1; print("MISMATCH",frag,minf,maxf,seg,budget,waves,[len(f) for f in files],rep,r,err.value.decode())
print("cases",cases,"bad",bad)
