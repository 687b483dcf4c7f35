import sys, copy; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch
from music2midi_amd.config import DEFAULT_CONFIG
from test_train_gpu import _setup
for mode in ("bf16","fp8"):
    model, tr, orc, params, geom, x, feats, cond, labels = _setup(copy.deepcopy(DEFAULT_CONFIG), mode, 4, 188, 48)
    loss,_ = tr.forward_backward(x.cuda(), cond.cuda(), labels.cuda())
    _,_,go = orc.loss_and_grads(feats, cond, labels)
    rows=[]
    for name,(off,shape) in tr.layout.items():
        g=tr.grads[off:off+int(np.prod(shape))].cpu().double(); r=go[name].reshape(-1).double()
        if r.norm()<1e-9: continue
        rows.append((float(torch.dot(g,r)/(g.norm()*r.norm())), float((g-r).norm()/r.norm()), name))
    rows.sort()
    cs=np.array([r[0] for r in rows])
    print(mode,"loss",loss.item(),"cos min/median/mean",cs.min(),np.median(cs),cs.mean())
    for r in rows[:8]: print("   %.4f %.3f %s"%r)
