import sys, torch
dev = torch.device("cuda:0")
x = torch.ones(1 << 20, device=dev)
pat = sys.argv[1]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def body():
    cur = torch.cuda.current_stream()
    if pat == "A":   # nested fork, single wait each
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = x * 2
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                b = a + 1
            c = a * 3
            s1.wait_stream(s2)
            d = b + c
        cur.wait_stream(s1)
        return d
    if pat == "B":   # nested fork, s2 waits on s1 twice
        s1.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = x * 2
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                b = a + 1
            c = a * 3
            s2.wait_stream(s1)
            with torch.cuda.stream(s2):
                b2 = b + c
            e = c * 2
            s1.wait_stream(s2)
            d = b2 + e
        cur.wait_stream(s1)
        return d
    if pat == "C":   # both forked from origin; s2 later waits on s1
        s1.wait_stream(cur); s2.wait_stream(cur)
        with torch.cuda.stream(s1):
            a = x * 2
        with torch.cuda.stream(s2):
            b = x + 1
        s2.wait_stream(s1)
        with torch.cuda.stream(s2):
            b2 = b + a
        with torch.cuda.stream(s1):
            c = a * 3
        cur.wait_stream(s1); cur.wait_stream(s2)
        return b2 + c
def body_D():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    acc = None
    a = x
    for i in range(4):
        with torch.cuda.stream(s1):
            a = a * 2
        s2.wait_stream(s1)
        with torch.cuda.stream(s2):
            b = a + i
            acc = b if acc is None else acc + b
    cur.wait_stream(s1); cur.wait_stream(s2)
    return acc + a
if pat == "D":
    body = body_D
want = body(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    got = body()
g.replay(); torch.cuda.synchronize()
print(pat, "OK", torch.equal(got, want))
