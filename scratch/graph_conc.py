import torch, time
dev = torch.device("cuda:0")
# three independent chains of small-grid kernels (each chain ~ms, uses few CUs like the FPS chain)
xs = [torch.randn(8, 256, device=dev) for _ in range(3)]
def chain(x, n=300):
    for _ in range(n):
        x = x * 1.0001 + 0.1
    return x
streams = [torch.cuda.Stream() for _ in range(3)]
graphs, outs = [], []
for i in range(3):
    with torch.cuda.stream(streams[i]):
        chain(xs[i], 10)
torch.cuda.synchronize()
for i in range(3):
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=streams[i]):
        outs.append(chain(xs[i]))
    graphs.append(g)
def run(k):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(5):
        for i in range(k):
            with torch.cuda.stream(streams[i]):
                graphs[i].replay()
        torch.cuda.synchronize()
    return (time.perf_counter() - t) / 5 * 1e3
for k in (1, 2, 3):
    print(f"{k} graphs on {k} streams: {run(k):.2f} ms per round")
# same three chains as branches of ONE graph
g = torch.cuda.CUDAGraph()
s = streams
with torch.cuda.graph(g):
    cur = torch.cuda.current_stream()
    for i in range(3): s[i].wait_stream(cur)
    for i in range(3):
        with torch.cuda.stream(s[i]): outs.append(chain(xs[i]))
    for i in range(3): cur.wait_stream(s[i])
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(5): g.replay()
torch.cuda.synchronize(); print(f"one graph, 3 branches: {(time.perf_counter()-t)/5*1e3:.2f} ms")
