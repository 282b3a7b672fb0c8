import sys, time, torch
sys.path.insert(0, '/root/repo')
sys.argv = ['x']
from types import SimpleNamespace
from get_model_amd.config import build_model, region_model_yaml
from get_model_amd.ddp import GradBuckets
from get_model_amd import optim as GO
import importlib.util
spec = importlib.util.spec_from_file_location('b', '/root/repo/bench.py'); bm = importlib.util.module_from_spec(spec); spec.loader.exec_module(bm)
dev = torch.device('cuda', 0)
model = build_model(region_model_yaml("finetune", num_layers=12)).to(dev); model.train()
buckets = GradBuckets(model)
opt = GO.configure_optimizer(model, SimpleNamespace(opt="adamw", lr=1e-4, weight_decay=0.05, opt_eps=1e-8, opt_betas=[0.9, 0.999]))
import os
BB = int(os.environ.get("BB", 256))
batch = bm.synthetic_batch(BB, 200, 1, dev)
def step():
    buckets.zero_grad()
    out = model(**model.get_input(batch)); pred, obs = model.before_loss(out, batch)
    loss = model.after_loss(model.loss(pred, obs)); loss.backward(); buckets.finish(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3*(t1-t0)/10:.2f} ms/step, total {1e3*(t2-t0)/10:.2f} ms/step")
