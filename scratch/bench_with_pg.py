"""default bench.py with a one-rank RCCL process group alive (does the group's mere existence change the overlap?)
PG_TOUCH=1: create (and use once) torch's 32 pooled streams before the group exists"""
import os, runpy, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29512")
torch.cuda.set_device(0)
if os.environ.get("PG_TOUCH"):
    ss = [torch.cuda.Stream() for _ in range(32)]
    for s in ss:
        with torch.cuda.stream(s):
            torch.zeros(4, device="cuda").add_(1)
    torch.cuda.synchronize()
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
if os.environ.get("PG_WARM"):
    t = torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path("bench.py", run_name="__main__")
