import sys; sys.path.insert(0,'.')
import torch, numpy as np
from music2midi_amd import native, synth
from oracle.mx8 import mx_quant_dequant
lib=native.load()
def mx(a,b,e5=0):
    M,K=a.shape; N=b.shape[0]
    a_d,b_d=a.cuda().contiguous(),b.cuda().contiguous(); c=torch.empty((M,N),device='cuda')
    native.check(lib.m2m_mx8_matmul_f32(a_d.data_ptr(),b_d.data_ptr(),M,N,K,e5,c.data_ptr(),native.stream_handle()),"mx")
    return c.cpu()
K=128
for fmt,e5 in (("e4m3",0),("e5m2",1)):
    a=torch.from_numpy(synth.normal(3,"a",(64,K),1.0))
    a[5,40]=300.0; a[6,:32]*=1e-3; a[7,70]=3e4
    dq=mx(a,torch.eye(K),e5)
    want=mx_quant_dequant(a,fmt)
    diff=(dq-want).abs()
    bad=(diff>0).nonzero()
    print(fmt,"mismatching elements",len(bad),"of",a.numel(),"max rel",(diff/ (want.abs()+1e-30)).max().item())
    for (i,j) in bad[:12].tolist():
        blk=a[i,(j//32)*32:(j//32)*32+32]; 
        print(f"  a[{i},{j}]={a[i,j].item():.6g} blockamax={blk.abs().max().item():.4g} dev={dq[i,j].item():.6g} want={want[i,j].item():.6g}")
