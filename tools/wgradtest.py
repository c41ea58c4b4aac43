import torch, time
dev='cuda:0'
M=205312
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
x=torch.randn(M,256,device=dev,dtype=torch.bfloat16)
tot=0
for name,N,K in (("qkv",768,256),("gate",64,256),("out",256,256),("mlp_in",1536,256),("mlp_out",256,768)):
    dy=torch.randn(M,N,device=dev,dtype=torch.bfloat16); a=torch.randn(M,K,device=dev,dtype=torch.bfloat16)
    w=torch.randn(N,K,device=dev,dtype=torch.bfloat16)
    tw=bench(lambda: dy.t()@a); td=bench(lambda: dy@w); tb=bench(lambda: dy.sum(0)); tf=bench(lambda: torch.nn.functional.linear(a,w))
    gf=2*M*N*K/1e9
    print(f"{name:8s} N={N:5d} K={K:4d} fwd {tf:.3f} ms ({gf/tf:.0f} GF/ms) dgrad {td:.3f} wgrad {tw:.3f} ms ({gf/tw:.0f} GF/ms) bias-sum {tb:.3f}")
    tot+=tw+tb
print("wgrad+bias per block", tot, "ms; x8 =", 8*tot)
