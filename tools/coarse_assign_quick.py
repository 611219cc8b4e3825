import torch, sys
sys.path.insert(0, "/root/repo")
from torchpq_amd import kernels as K
dev="cuda:0"
g=torch.Generator(device=dev); g.manual_seed(1)
for d,m,n in ((128,1000000,16384),(64,1000000,16384),(128,1000000,4096)):
    A=torch.randn(d,m,generator=g,device=dev); B=A[:,torch.randperm(m,generator=g,device=dev)[:n]].contiguous()
    op=K.CoarseAssignHip()
    lab=op(A,B); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): op(A,B)
    e1.record(); torch.cuda.synchronize()
    _,l32=K.MaxSimHip()(A[None],B[None],dim=2,mode="tn")
    print(d,m,n, round(e0.elapsed_time(e1)/5,3),"ms  rechecked",op.last_rechecked()/m, "equal", float((lab==l32[0]).double().mean()))
