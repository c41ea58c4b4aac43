import sys, time; sys.path.insert(0, "/root/repo")
import torch, bench
from viforsdes_amd.examples.sdes import lv_problem
from viforsdes_amd.config import PretrainConfig
tr = bench.build_trainer(lv_problem(), 24, torch.device("cuda:0"), True, seed=1)
class C:  # print capture failures
    def __getattr__(self, n):
        return getattr(tr.console.__class__, n).__get__(tr.console)
orig = tr.console.config_panel
tr.console.config_panel = lambda *a, **k: print("PANEL:", a)
for n in (20, 300):
    t0 = time.time(); tr.pretrain_sde_parameters(PretrainConfig(n_iterations=n)); torch.cuda.synchronize()
    print("pretrain", n, "iters:", time.time() - t0, "s; graph:", hasattr(tr, "_pretrain_graph"))
