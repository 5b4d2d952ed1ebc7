import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from test_gpu_pipeline import _setup, _batches, _train
from amcontrast3d_amd.pipeline import GeometryPrefetcher
dev = torch.device('cuda:0')
model, criterion, aa = _setup(dev)
state = {k: v.clone() for k, v in model.state_dict().items()}
runs = []
for r in range(3):
    model.load_state_dict(state)
    src = _batches(dev, 4)
    if r == 2: src = GeometryPrefetcher(src, model, criterion.contrast_head, 13, None, aa, depth=1)
    runs.append(_train(model, criterion, aa, src, 4))
for name, a, b in (("plain vs plain", runs[0], runs[1]), ("plain vs prefetched", runs[0], runs[2])):
    print(name, [f"{float((x[0]-y[0]).abs().max()):.2e}" for x, y in zip(a, b)], [f"{abs(x[1]-y[1]):.2e}" for x, y in zip(a, b)])
