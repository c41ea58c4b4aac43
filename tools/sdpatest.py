import torch, time, os
dev='cuda:0'
def bench(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t=time.time()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.time()-t)/n*1e3
q=torch.randn(512,4,401,64,device=dev,dtype=torch.bfloat16,requires_grad=True)
k=torch.randn_like(q,requires_grad=True); v=torch.randn_like(q,requires_grad=True)
def fb():
    o=torch.nn.functional.scaled_dot_product_attention(q,k,v); o.backward(torch.ones_like(o))
print("default fwd %.3f fwd+bwd %.3f"%(bench(lambda: torch.nn.functional.scaled_dot_product_attention(q,k,v)), bench(fb)))
try:
    print("pref lib", torch.backends.cuda.preferred_rocm_fa_library())
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    print("ck fwd %.3f fwd+bwd %.3f"%(bench(lambda: torch.nn.functional.scaled_dot_product_attention(q,k,v)), bench(fb)))
    torch.backends.cuda.preferred_rocm_fa_library("aotriton")
except Exception as e: print("ck not available:", type(e).__name__, str(e)[:200])
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
    try:
        with sdpa_kernel(be):
            print(be, "fwd %.3f fwd+bwd %.3f"%(bench(lambda: torch.nn.functional.scaled_dot_product_attention(q,k,v)), bench(fb)))
    except Exception as e: print(be, "failed", str(e)[:100])
# padded seq 448/512 with mask? just time unmasked padded shapes to see sensitivity
for n in (384, 416, 448, 512):
    q2=torch.randn(512,4,n,64,device=dev,dtype=torch.bfloat16,requires_grad=True); k2=torch.randn_like(q2,requires_grad=True); v2=torch.randn_like(q2,requires_grad=True)
    def fb2():
        o=torch.nn.functional.scaled_dot_product_attention(q2,k2,v2); o.backward(torch.ones_like(o))
    print("seq",n,"fwd %.3f fwd+bwd %.3f"%(bench(lambda: torch.nn.functional.scaled_dot_product_attention(q2,k2,v2)), bench(fb2)))
