import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np
from music2midi_amd import native
lib=native.load()
def mx(a,b,e5=0):
    M,K=a.shape; N=b.shape[0]
    a_d,b_d=a.cuda().contiguous(),b.cuda().contiguous(); c=torch.empty((M,N),device='cuda')
    native.check(lib.m2m_mx8_matmul_f32(a_d.data_ptr(),b_d.data_ptr(),M,N,K,e5,c.data_ptr(),native.stream_handle()),"mx")
    return c.cpu()
K=128
a=torch.ones(64,K); b=torch.ones(64,K)
c=mx(a,b); print("ones: expect",K,"got",c[0,0].item(), c.min().item(), c.max().item())
a=torch.full((64,K),2.0); c=mx(a,b); print("twos: expect",2*K, c[0,0].item())
a=torch.ones(64,K); a[:, :32]=4.0; c=mx(a,b); print("block0=4: expect", 4*32+96, c[0,0].item())
a=torch.ones(64,K); a[:, 32:64]=4.0; c=mx(a,b); print("block1=4: expect", 4*32+96, c[0,0].item())
a=torch.ones(64,K); a[:, 64:96]=4.0; c=mx(a,b); print("block2=4: expect", 224, c[0,0].item())
a=torch.zeros(64,K); a[:,5]=1.0; b2=torch.zeros(64,K); b2[:,5]=3.0; c=mx(a,b2); print("single k=5: expect 3", c[0,0].item())
for kk in (0,1,7,15,16,31,32,40,63,64,100,127):
    a=torch.zeros(64,K); a[:,kk]=1.0; b2=torch.ones(64,K); c=mx(a,b2); print("onehot k",kk,"->",c[0,0].item(), end=" | ")
print()
a=torch.zeros(64,K); a[3,:]=1.0; c=mx(a,torch.ones(64,K)); print("row 3 only: rows nonzero", c.abs().sum(1).nonzero().flatten().tolist())
b2=torch.zeros(64,K); b2[9,:]=1.0; c=mx(torch.ones(64,K),b2); print("col 9 only: cols nonzero", c.abs().sum(0).nonzero().flatten().tolist())
a=torch.ones(64,K)*0.5; c=mx(a,torch.ones(64,K)); print("halves: expect 64", c[0,0].item())
