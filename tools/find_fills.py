"""Diagnostic (GPU): which Python call sites launch the small torch kernels (fills, copies) in one training step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from types import SimpleNamespace
from get_model_amd.config import build_model, region_model_yaml
from get_model_amd.ddp import GradBuckets
from get_model_amd import optim as GO
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_batch

dev = torch.device("cuda", 0)
model = build_model(region_model_yaml("finetune", num_layers=12)).to(dev).train()
buckets = GradBuckets(model)
opt = GO.configure_optimizer(model, SimpleNamespace(opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999]))
batch = synthetic_batch(64, 200, 1, dev)


def step():
    buckets.zero_grad()
    out = model(**model.get_input(batch))
    pred, obs = model.before_loss(out, batch)
    loss = model.after_loss(model.loss(pred, obs))
    loss.backward()
    buckets.finish()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
print(prof.key_averages(group_by_stack_n=6).table(sort_by="count", row_limit=40, max_name_column_width=60, max_src_column_width=110))
