import sys, torch
sys.path.insert(0, '/root/repo')
from transcodon_amd import _lib, ops
dev='cuda'
for (M,N,K) in [(131072,768,3072),(131072,3072,768),(131072,768,768)]:
    a=torch.randn(M,K,device=dev).bfloat16(); w=(torch.randn(N,K,device=dev)*0.03).bfloat16()
    out=torch.empty(M,N,device=dev)
    for _ in range(3): ops.gemm_train(a,w,_lib.EPI_PLAIN_F32,out=out)
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.gemm_train(a,w,_lib.EPI_PLAIN_F32,out=out)
    e1.record(); torch.cuda.synchronize()
    us=e0.elapsed_time(e1)/20*1e3
    print(M,N,K,round(us,1),'us',round(2*M*N*K/us/1e6,1),'TF/s')
