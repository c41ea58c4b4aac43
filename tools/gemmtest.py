import torch, time
dev='cuda:0'
M=205312
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
x=torch.randn(M,256,device=dev,dtype=torch.bfloat16)
for N in (1364,1408,1536):
    w=torch.randn(N,256,device=dev,dtype=torch.bfloat16); b=torch.randn(N,device=dev,dtype=torch.bfloat16)
    print('in  N',N, '%.3f ms'%bench(lambda: torch.nn.functional.linear(x,w,b)))
for K in (682,704,768):
    a=torch.randn(M,K,device=dev,dtype=torch.bfloat16); w=torch.randn(256,K,device=dev,dtype=torch.bfloat16); b=torch.randn(256,device=dev,dtype=torch.bfloat16)
    print('out K',K, '%.3f ms'%bench(lambda: torch.nn.functional.linear(a,w,b)))
    g=torch.randn(M,256,device=dev,dtype=torch.bfloat16)
    print('  dgrad (g @ w) K',K, '%.3f ms'%bench(lambda: g@w), ' wgrad (g^T a)', '%.3f ms'%bench(lambda: g.t()@a))
w=torch.randn(768,256,device=dev,dtype=torch.bfloat16)
print('qkv', '%.3f ms'%bench(lambda: torch.nn.functional.linear(x,w)))
w=torch.randn(832,256,device=dev,dtype=torch.bfloat16)
print('qkv+gate 832', '%.3f ms'%bench(lambda: torch.nn.functional.linear(x,w)))
w=torch.randn(64,256,device=dev,dtype=torch.bfloat16)
print('gate 64', '%.3f ms'%bench(lambda: torch.nn.functional.linear(x,w)))
w=torch.randn(256,256,device=dev,dtype=torch.bfloat16)
print('out 256', '%.3f ms'%bench(lambda: torch.nn.functional.linear(x,w)))
q=torch.randn(512,4,401,64,device=dev,dtype=torch.bfloat16)
print('sdpa fwd 401', '%.3f ms'%bench(lambda: torch.nn.functional.scaled_dot_product_attention(q,q,q)))
