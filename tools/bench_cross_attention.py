#!/usr/bin/env python
"""Micro-benchmark of the decode cross-attention kernel (GPU box only): 256 rows x 12 heads x 197 keys, every row alive, the K/V
blocks of 12 layers in turn (cold: 12 x 2 x 116 MB), fp32 cache (mode 0) against the split mode's KV16 cache (mode 1: int16 rows +
one scale each, 132 bytes per row).  Round 3 with the 24-bit cache this one replaced: 53.8 us = 5.76 TB/s (fp32) against 43.3 us =
5.36 TB/s (192-byte rows); a double-buffered variant (chunks of 40 keys x 2) ran at 76 us and non-temporal loads at 42.8 - neither
kept.     python tools/bench_cross_attention.py [rows]
(rows < 256: what the launch would cost with the captions that are still open compacted to the front - round 5)"""
import ctypes as C, sys, os, torch
sys.path.insert(0, os.getcwd())
from embodied_captioning_amd import _native
lib=_native.load_library(); s=C.c_void_p(torch.cuda.current_stream().cuda_stream)
R=int(sys.argv[1]) if len(sys.argv)>1 else 256
H,NK=12,197
rows=R*H*NK
q=torch.randn(R,H*64,device='cuda')
nb=12   # distinct caches (12 layers) so data is cold: 12 x 2 x 116 MB
Ks=[torch.randn(rows,64,device='cuda') for _ in range(nb)]; Vs=[torch.randn(rows,64,device='cuda') for _ in range(nb)]
KB=(rows+31)//32*4224
Kp=[torch.zeros(KB,dtype=torch.uint8,device='cuda') for _ in range(nb)]; Vp=[torch.zeros(KB,dtype=torch.uint8,device='cuda') for _ in range(nb)]
for i in range(nb):
    lib.cap_op_pack_kv16(C.c_void_p(Ks[i].data_ptr()),C.c_void_p(Kp[i].data_ptr()),rows,s); lib.cap_op_pack_kv16(C.c_void_p(Vs[i].data_ptr()),C.c_void_p(Vp[i].data_ptr()),rows,s)
out=torch.zeros(R,H*64,device='cuda')
def run(i,mode):
    if mode==0:
        rc=lib.cap_op_decode_attention(2,C.c_void_p(q.data_ptr()),C.c_void_p(Ks[i].data_ptr()),C.c_void_p(Vs[i].data_ptr()),None,0,1,NK,NK,C.c_void_p(out.data_ptr()),R,H,0,s)
    else:
        rc=lib.cap_op_decode_attention(2,C.c_void_p(q.data_ptr()),C.c_void_p(Kp[i].data_ptr()),C.c_void_p(Vp[i].data_ptr()),None,0,1,NK,NK,C.c_void_p(out.data_ptr()),R,H,16,s)
    assert rc==0, lib.cap_last_error()
res={}
for rep in range(3):
  for mode in (0,1):
    for i in range(nb): run(i,mode)
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(5):
        for i in range(nb): run(i,mode)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)*1e3/(5*nb)
    res.setdefault(mode,[]).append(us)
byt={0:rows*64*4*2}; 
for m,v in res.items():
    b=rows*64*4*2 if m==0 else KB*2
    print('mode',m,'us',[round(x,1) for x in v],'TB/s',round(b/min(v)/1e6,2))
