"""Is a cross-stream wait (event record + stream wait) free on the host when the producer stream is busy?
Producer: ~20 ms of queued kernels, eager or as one graph replay.  Prints the host time of wait_stream and of the
next launch on the waiting stream."""
import time, torch
dev = torch.device("cuda", 0)
x = torch.randn(8192, 8192, device=dev)
def work(n=12):
    y = x
    for _ in range(n):
        y = y @ x
        y = y * 1e-4
    return y
s1, s2, s3 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
with torch.cuda.stream(s1):
    work(2)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=s3):
    out = work()
torch.cuda.synchronize()
for mode in ("eager", "graph", "eager", "graph"):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(s1):
        if mode == "eager": work()
        else: g.replay()
    t1 = time.perf_counter()
    s2.wait_stream(s1)
    t2 = time.perf_counter()
    with torch.cuda.stream(s2):
        z = torch.zeros(16, device=dev) + 1
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"{mode:6s} launch {1e3*(t1-t0):7.3f} ms | wait_stream {1e3*(t2-t1):7.3f} ms | next launch {1e3*(t3-t2):7.3f} ms | drain {1e3*(t4-t3):7.3f} ms")
