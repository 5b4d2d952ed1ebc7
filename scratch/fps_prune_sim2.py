import numpy as np, sys
sys.path.insert(0,'.')
from amcontrast3d_amd.synthetic import make_scene
N, M = 24000, 6000
p = make_scene(0, N)['pos'].astype(np.float64)
# the kernel's order: morton of a 16^3 grid, stable
lo, hi = p.min(0), p.max(0)
cc = np.clip(((p - lo) * (16.0/(hi-lo))).astype(np.int64), 0, 15)
def spread(v): return (v & 1) | ((v & 2) << 2) | ((v & 4) << 4) | ((v & 8) << 6)
code = spread(cc[:,0]) | (spread(cc[:,1]) << 1) | (spread(cc[:,2]) << 2)
order = np.argsort(code, kind='stable')
ps = p[order]
for gsize in (512, 256):
    ng = (N + gsize - 1)//gsize
    bounds = [(g*gsize, min((g+1)*gsize, N)) for g in range(ng)]
    cen = np.array([ps[a:b].mean(0) for a,b in bounds])
    rad = np.array([np.sqrt(((ps[a:b]-cen[g])**2).sum(1)).max() for g,(a,b) in enumerate(bounds)])
    blo = np.array([ps[a:b].min(0) for a,b in bounds]); bhi = np.array([ps[a:b].max(0) for a,b in bounds])
    res = {}
    for mode in ('sphere', 'box', 'both'):
        temp = np.full(N, 1e10); gmax = np.full(ng, 1e10)
        cur = int(np.where(order==0)[0][0]); scanned = []; maxwave = []
        for it in range(1, M):
            c = ps[cur]
            D = np.sqrt(((cen - c)**2).sum(1))
            dbox = np.sqrt((np.maximum(np.maximum(blo - c, c - bhi), 0)**2).sum(1))
            sk_s = (D - rad >= np.sqrt(gmax)); sk_b = dbox >= np.sqrt(gmax)
            need = ~(sk_s if mode=='sphere' else sk_b if mode=='box' else (sk_s | sk_b))
            scanned.append(need.sum())
            # per-wave load: group G -> wave G % 8
            maxwave.append(np.bincount(np.nonzero(need)[0] % 8, minlength=8).max())
            for g in np.nonzero(need)[0]:
                a,b = bounds[g]
                d = ((ps[a:b]-c)**2).sum(1)
                temp[a:b] = np.minimum(temp[a:b], d); gmax[g] = temp[a:b].max()
            cur = int(np.argmax(temp))
        sc = np.array(scanned); mw = np.array(maxwave)
        print(f"group {gsize} {mode:6s}: mean swept groups/iter {sc.mean():6.2f} ({sc.mean()/ng*100:4.1f}%), mean max-per-wave {mw.mean():5.2f}")
