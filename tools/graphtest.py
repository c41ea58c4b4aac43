import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from bench import build_trainer
from viforsdes_amd.examples.sdes import ou_problem, lv_problem
wl = sys.argv[1] if len(sys.argv) > 1 else "ou"
problem = ou_problem() if wl == "ou" else lv_problem()
B = 128 if wl == "ou" else 512
dev = torch.device("cuda:0")
tr = build_trainer(problem, B, dev, True, seed=1234)
model, ctx = tr.ctx.model, tr.ctx
def step():
    r = tr._train_step(model); ctx.ema.update(); return r
for _ in range(3): step()
torch.cuda.synchronize(); t=time.time()
for _ in range(10): step()
torch.cuda.synchronize(); print("eager ms/step", (time.time()-t)*100)
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    res = step()
torch.cuda.synchronize()
vals=[]
for _ in range(3):
    g.replay(); vals.append(float(res.elbo_result.evidence_lower_bound))
print("elbo per replay", vals, "mean param", float(model.sde_parameter_posterior.mean.mean()))
torch.cuda.synchronize(); t=time.time()
for _ in range(20): g.replay()
torch.cuda.synchronize(); print("graph ms/step", (time.time()-t)*50)
