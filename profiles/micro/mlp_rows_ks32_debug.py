import sys, os, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from flowcompare_amd import engine
import test_gpu_ops as T
for (rows,k0,k1,n_mid) in ((64,256,200,1),(64,256,200,2),(64,256,200,3),(300,250,150,3),(64,150,64,3),(64,256,8,3)):
    sd = T._mlp_state(k0+k1, n_mid, seed=rows+n_mid)
    x0, x1 = T._rand(rows,k0,seed=11,scale=2.0), T._rand(rows,k1,seed=12,scale=1.5)
    ref = T._mlp_ref(torch.cat((x0,x1),1), sd, n_mid)
    a = engine.op_mlp_hidden(x0.cuda(), x1.cuda(), sd, use_rows=True).cpu().double()
    b = engine.op_mlp_hidden(x0.cuda(), x1.cuda(), sd, use_rows=False).cpu().double()
    ea, eb = (a-ref).abs(), (b-ref).abs()
    print(rows,k0,k1,n_mid, "chain %.2e per-layer %.2e" % (ea.max().item(), eb.max().item()), "worst col", int(ea.max(0).values.argmax()), "worst row", int(ea.max(1).values.argmax()), "frac>1e-5: %.4f" % (ea>1e-5).double().mean().item())
