"""long-run stability: 500 captured steps of bench.py's schedule, then 40 eager prefetched steps; peak memory and loss"""
import subprocess, sys, json
r = subprocess.run([sys.executable, "bench.py", "--steps", "500", "--warmup", "5", "--no-cpu-baseline"], capture_output=True, text=True)
d = json.loads(r.stdout.strip().splitlines()[-1])
print("bench 500 steps:", d["ms_per_step"], "ms/step, loss", d["loss"])
import numpy as np, torch
sys.path.insert(0, ".")
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic, train
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device("cuda:0")
c = EasyConfig(); c.update(configs.model_cfg("S", dropout=0.5)); model = build_model_from_cfg(c).to(dev)
cc = EasyConfig(); cc.update(configs.criterion_cfg()); crit = build_criterion_from_cfg(cc).to(dev)
cfg = EasyConfig(); cfg.update({"num_classes": 13, "ignore_index": None, "ambiguity_args": configs.ambiguity_args("s3dis"), "feature_keys": "x,heights",
                                "use_amp": False, "step_per_update": 1, "grad_norm_clip": 10, "sched_on_epoch": True})
opt = torch.optim.AdamW(model.parameters(), lr=0.01)
nbs = [synthetic.make_batch(8, 24000, first_id=8 * k) for k in range(4)]
def loader(n):
    for k in range(n):
        nb = nbs[k % 4]
        yield {"pos": torch.from_numpy(nb["pos"]), "y": torch.from_numpy(nb["y"]), "x": torch.from_numpy(np.ascontiguousarray(nb["x"][:, :3].transpose(0, 2, 1))),
               "heights": torch.from_numpy(np.ascontiguousarray(nb["x"][:, 3:4].transpose(0, 2, 1)))}
import time
for ep in range(3):
    torch.cuda.reset_peak_memory_stats(); t = time.perf_counter()
    out = train.train_one_epoch(model, loader(40), crit, opt, None, None, ep, cfg)
    torch.cuda.synchronize()
    print(f"epoch {ep}: loss {out[0]:.3f} {(time.perf_counter()-t)/40*1e3:.1f} ms/step peak {torch.cuda.max_memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
