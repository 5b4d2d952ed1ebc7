import numpy as np, sys
sys.path.insert(0,'.')
from amcontrast3d_amd.synthetic import make_scene
N, M = 24000, 6000
p = make_scene(0, N)['pos'].astype(np.float64)
# morton sort
q = ((p - p.min(0)) / (p.max(0) - p.min(0) + 1e-9) * 1023).astype(np.int64)
def part(x):
    x &= 0x3ff; x = (x | (x << 16)) & 0x30000ff; x = (x | (x << 8)) & 0x300f00f; x = (x | (x << 4)) & 0x30c30c3; x = (x | (x << 2)) & 0x9249249; return x
code = part(q[:,0].copy()) | (part(q[:,1].copy()) << 1) | (part(q[:,2].copy()) << 2)
order = np.argsort(code, kind='stable')
ps = p[order]
for gsize in (64, 256, 512, 1536):
    ng = (N + gsize - 1)//gsize
    gid = np.arange(N)//gsize
    cen = np.array([ps[gid==g].mean(0) for g in range(ng)])
    rad = np.array([np.sqrt(((ps[gid==g]-cen[g])**2).sum(1)).max() for g in range(ng)])
    temp = np.full(N, 1e10)
    gmax = np.full(ng, 1e10)
    cur = int(np.where(order==0)[0][0])
    scanned = []
    for it in range(1, M):
        c = ps[cur]
        D = np.sqrt(((cen - c)**2).sum(1))
        need = ~((D - rad >= np.sqrt(gmax)) & (D >= rad))
        scanned.append(need.sum())
        for g in np.nonzero(need)[0]:
            sl = slice(g*gsize, min((g+1)*gsize, N))
            d = ((ps[sl]-c)**2).sum(1)
            temp[sl] = np.minimum(temp[sl], d)
            gmax[g] = temp[sl].max()
        cur = int(np.argmax(temp))
    sc = np.array(scanned)
    print(f"group {gsize:5d}: groups {ng:4d}, mean scanned/iter {sc.mean():7.2f} ({sc.mean()/ng*100:5.1f}%), after it 500: {sc[500:].mean():6.2f}, after 3000: {sc[3000:].mean():6.2f}; points scanned/iter {sc.mean()*gsize:8.0f} of {N}")
