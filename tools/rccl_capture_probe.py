import os, torch, torch.distributed as dist, time
os.environ.setdefault("MASTER_ADDR","127.0.0.1"); os.environ.setdefault("MASTER_PORT","29533"); os.environ.setdefault("NCCL_DEBUG","WARN")
dev=torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
s=torch.cuda.Stream(); 
x=torch.ones(130, dtype=torch.float64, device=dev)
with torch.cuda.stream(s):
    for _ in range(3):
        dist.all_reduce(x)
torch.cuda.synchronize(); time.sleep(0.5)
g=torch.cuda.CUDAGraph()
y=torch.ones(130, dtype=torch.float64, device=dev)
with torch.cuda.stream(s):
    with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
        y.mul_(2.0)
        dist.all_reduce(y)
        y.add_(1.0)
torch.cuda.synchronize()
for i in range(5):
    y.fill_(1.0)
    with torch.cuda.stream(s):
        g.replay()
    torch.cuda.synchronize()
    print("replay", i, float(y[0]))
# timing: 68 collectives in one graph vs eager
g2=torch.cuda.CUDAGraph()
with torch.cuda.stream(s):
    with torch.cuda.graph(g2, stream=s, capture_error_mode="thread_local"):
        for _ in range(68):
            y.mul_(1.0)
            dist.all_reduce(y)
torch.cuda.synchronize()
with torch.cuda.stream(s):
    g2.replay(); torch.cuda.synchronize()
    t=time.perf_counter()
    for _ in range(10): g2.replay()
    torch.cuda.synchronize()
print("68 captured (mul + all_reduce) pairs: %.3f ms per replay" % ((time.perf_counter()-t)/10*1e3))
dist.destroy_process_group()
print("PROBE_OK")
