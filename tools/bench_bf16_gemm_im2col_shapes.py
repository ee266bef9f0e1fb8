import torch, torch.nn.functional as F, sys
sys.path.insert(0,'.')
def ms(fn,n=5,warm=2):
    for _ in range(warm): fn()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b)/n
for (M,K,N) in ((64*256*256,1152,128),(64*256*256,2304,128),(64*128*128,2304,256),(64*64*64,4608,512)):
    a=torch.randn(M,K,device='cuda').to(torch.bfloat16); w=torch.randn(N,K,device='cuda').to(torch.bfloat16)
    t=ms(lambda: F.linear(a,w)); fl=2.0*M*K*N
    print(f"vendor bf16 GEMM [{M},{K}]x[{K},{N}]: {t:.3f} ms {fl/t*1e-9:.0f} TF/s ({fl/t*1e-9/2500:.3f})", flush=True)
    del a,w
