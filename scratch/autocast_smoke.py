import sys, torch
sys.path.insert(0, '.')
import amcontrast3d_amd
amcontrast3d_amd.activate()
from amcontrast3d_amd import configs, synthetic
from openpoints.loss import build_criterion_from_cfg
from openpoints.models import build_model_from_cfg
from openpoints.utils import EasyConfig
dev = torch.device('cuda:0')
torch.manual_seed(0)
c = EasyConfig(); c.update(configs.model_cfg('S', dropout=0))
model = build_model_from_cfg(c).to(dev).train()
cc = EasyConfig(); cc.update(configs.criterion_cfg()); crit = build_criterion_from_cfg(cc).to(dev)
aa = EasyConfig(); aa.update(configs.ambiguity_args('s3dis'))
data = {k: torch.from_numpy(v).to(dev) for k, v in synthetic.make_batch(2, 4096, first_id=5).items()}
logits32, stage = model(data); l32 = crit(logits32, data['y'], stage, 13, None, aa)
with torch.autocast('cuda', dtype=torch.bfloat16):
    logits, stage = model(data)
    loss = crit(logits.float(), data['y'], stage, 13, None, aa)
loss.backward()
print('fp32 loss', float(l32), 'bf16-autocast loss', float(loss), 'logits dtype', logits.dtype,
      'max |dlogit|', float((logits.float() - logits32).abs().max()), 'grad finite', all(torch.isfinite(p.grad).all() for p in model.parameters()))
