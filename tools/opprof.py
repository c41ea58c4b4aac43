"""torch.profiler view of one LV training step: which ATen ops launch the non-vsde kernels (GPU only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from viforsdes_amd.examples.sdes import lv_problem


def main():
    device = torch.device("cuda:0")
    tr = bench.build_trainer(lv_problem(), 512, device, True, seed=1234)
    model = tr.ctx.model

    def step():
        tr._train_step(model)
        tr.ctx.ema.update()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        for _ in range(2):
            step()
        torch.cuda.synchronize()
    print(prof.key_averages(group_by_input_shape=True).table(
        sort_by="self_cuda_time_total", row_limit=int(os.environ.get("OPPROF_ROWS", "50")), max_name_column_width=45, max_shapes_column_width=80))


if __name__ == "__main__":
    main()
