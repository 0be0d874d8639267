import os, sys, torch
os.environ["CCV_GEMM_TUNE"]="1"; os.environ["CCV_GEMM_RING"]="-1"
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(),"tools"))
from camc2v_amd import ops
import bench_kernels as bk
dev=torch.device("cuda:0")
for M,N,K,res in ((2048,1280,1280,True),(2048,3840,1280,False),(2048,1280,5120,True),(512,1280,1280,True),(512,3840,1280,False),(8192,640,640,True),(8192,1920,640,False),(8192,640,2560,True),(32768,320,320,True),(32768,960,320,False)):
    a=bk.rnd(M,K); w=bk.rnd(N,K,scale=0.05); bias=torch.zeros(N,device=dev); r=torch.zeros(M,N,device=dev)
    row=[]
    for t in ("0","44","24","42","22"):
        os.environ["CCV_GEMM_FAMTILE"]=t
        for sp in ("1","2"):
            os.environ["CCV_GEMM_SPLIT"]=sp
            fn=(lambda: ops.gemm(a,w,bias=bias,residual=r,out_f32=True)) if res else (lambda: ops.gemm(a,w))
            row.append(f"{t}/s{sp}:{bk.timeit(fn):6.1f}")
    print(M,N,K,"res" if res else "   "," ".join(row),flush=True)
