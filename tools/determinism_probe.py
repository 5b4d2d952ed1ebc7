"""Which gradients differ between two runs of the same eager train step (same weights, same batch)?  python tools/determinism_probe.py [width] [B] [N]"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
width = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
N = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = "cuda:0"
torch.manual_seed(0)
c = EasyConfig(); c.update(configs.model_cfg("S", dropout=0, width=width)); model = build_model_from_cfg(c).to(dev).train()
cc = EasyConfig(); cc.update(configs.criterion_cfg()); crit = build_criterion_from_cfg(cc).to(dev)
aa = EasyConfig(); aa.update(configs.ambiguity_args("s3dis"))
data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(B, N, first_id=300).items()}
state = {k: v.clone() for k, v in model.state_dict().items()}
runs = []
for r in range(6):
    model.load_state_dict(state)
    model.zero_grad(set_to_none=True)
    logits, stage = model(dict(data))
    loss = crit(logits, data["y"], stage, 13, None, aa)
    loss.backward()
    torch.cuda.synchronize()
    runs.append((logits.detach().clone(), float(loss), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}))
base = runs[0]
for r, (lg, ls, g) in enumerate(runs[1:], 1):
    bad = []
    for n in g:
        d = float((g[n] - base[2][n]).abs().max())
        s = float(base[2][n].abs().max())
        if d > 1e-4 * max(s, 1e-6):
            bad.append((n, d, s))
    print(f"run {r}: logits equal {torch.equal(lg, base[0])}, loss diff {abs(ls - base[1]):.2e}, tensors off by > 1e-4 of their range: {len(bad)}")
    for n, d, s in bad[:12]:
        print(f"     {n:60s} diff {d:.3e} range {s:.3e}")
