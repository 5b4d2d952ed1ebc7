"""run another script of this repo with a one-rank RCCL process group alive: python scratch/with_pg.py script.py [args]"""
import os, runpy, sys
import torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29513")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
t = torch.ones(8, device="cuda"); dist.all_reduce(t); torch.cuda.synchronize()
sys.argv = sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
