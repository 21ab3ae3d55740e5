"""Debug-only NumPy mirror of k_project (fp64) + per-pixel evaluation, to classify Gaussians."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "rt-gaussian-splat-renderer_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from oracle import rtgs_ref as ref

def qmat(q):
    x,y,z,w = q; s = w*w-(x*x+y*y+z*z)
    return np.array([[s+2*x*x, 2*(x*y-w*z), 2*(x*z+w*y)],[2*(x*y+w*z), s+2*y*y, 2*(y*z-w*x)],[2*(x*z-w*y), 2*(y*z+w*x), s+2*z*z]])

def project(arrays, cam):
    n = arrays["position"].shape[0]
    pos = arrays["position"].astype(np.float64)
    M = ref.gaussian_cov_inv(arrays["rotation"].astype(np.float64), arrays["scale"].astype(np.float64))
    M = M.astype(np.float32).astype(np.float64)  # stored fp32
    o = np.array(cam["position"], np.float64)
    Rc = qmat(np.array(cam["rotation"], np.float64))
    P = np.diag([1/cam["fx"], 1/cam["fy"], 1.0])
    col = Rc @ P
    out = []
    for g in range(n):
        m = o - pos[g]; u = M[g] @ m; C = m @ u - 3
        rec = dict(C=C, type=-1)
        if C > 0:
            v = col.T @ u; Mc = col.T @ M[g] @ col
            Q = np.outer(v, v) - C*Mc
            det = Q[0,0]*Q[1,1]-Q[0,1]**2
            rec.update(v=v, Mc=Mc, Q=Q)
            general = True
            if Q[0,0] < 0 and det > 0:
                X0 = (Q[1,1]*Q[0,2]-Q[0,1]*Q[1,2])/det; Y0 = (Q[0,0]*Q[1,2]-Q[0,1]*Q[0,2])/det
                K = Q[2,2]-(Q[0,2]*X0+Q[1,2]*Y0)
                hx = np.sqrt(K*(-Q[1,1])/det) if K>0 else 0; hy = np.sqrt(K*(-Q[0,0])/det) if K>0 else 0
                rec.update(X0=X0, Y0=Y0, K=K, hx=hx, hy=hy)
                if K > 0 and max(abs(X0),abs(Y0),hx,hy) < 4e6:
                    general = False
                    b0 = v[0]*X0+v[1]*Y0-v[2]
                    rec["b0"] = b0
                    rec["type"] = 0 if b0 < 0 else -2
                elif not K > 0:
                    general = False; rec["type"] = -3
            if general: rec["type"] = 1
        out.append(rec)
    return out

if __name__ == "__main__":
    import scenes
    arrays, cam, depth = scenes.CASES["inside"]()
    recs = project(arrays, cam)
    import collections
    print(collections.Counter(r["type"] for r in recs))
    r = recs[739]
    print({k: (v if np.isscalar(v) else None) for k, v in r.items()})
    W,H = cam["width"], cam["height"]
    for (pi,pj) in [(11,69),(34,15)]:
        X = pi+0.5-W/2; Y = pj+0.5-H/2; E = np.array([X,Y,-1.0])
        osc = ref.OracleScene(**arrays)
        rr = ref.render_topk(osc, cam["position"], cam["rotation"], W, H, cam["fx"], cam["fy"], depth=depth, pixels=np.array([[pi,pj]]))
        want = set(int(i) for i in rr["idx"][0] if i >= 0)
        dlen = np.sqrt((X/cam["fx"])**2+(Y/cam["fy"])**2+1)
        hits = []
        for g, rec in enumerate(recs):
            if rec["type"] in (0,1):
                A = E@rec["Mc"]@E; b = rec["v"]@E; e = b*b-A*rec["C"]
                if e > 0 and b < 0:
                    hits.append(((-b-np.sqrt(e))/A*dlen, g, rec["type"]))
        hits.sort()
        mine = set(g for _,g,_ in hits[:depth])
        print((pi,pj), "oracle-only", want-mine, "mine-only", mine-want, [(round(t,5),g,ty) for t,g,ty in hits[:18]])
        for g in (want-mine):
            print("  missing", g, {k: (v if np.isscalar(v) else None) for k, v in recs[g].items()})
