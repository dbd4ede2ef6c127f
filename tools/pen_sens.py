import sys, numpy as np, torch, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
from smplifyx_amd import synthetic
from oracle import penetration as OP
m = synthetic.make_synthetic_model(0)
parts = synthetic.make_synthetic_parts(m)
ign = ["9,16", "9,17", "6,16", "6,17", "1,2", "12,22"]
v = np.asarray(m["v_template"], np.float64); f = np.asarray(m["f"]).astype(np.int64)
t=time.time()
pairs = OP.candidate_pairs(v, f, parts["segm"], parts["parents"], ign)
print('pairs', len(pairs), time.time()-t)
rng=np.random.RandomState(0)
def grad(vv, sigma):
    vt=torch.tensor(vv,dtype=torch.float64,requires_grad=True)
    l=OP.penetration_loss(vt,f,pairs,sigma); l.backward(); return float(l), vt.grad.numpy()
for sigma in (1e-2,1e-3,1e-4):
    l0,g0=grad(v,sigma)
    for eps in (1e-8,1e-7):
        l1,g1=grad(v+eps*rng.normal(size=v.shape),sigma)
        print('sigma',sigma,'eps',eps,'loss',l0,'rel dl',abs(l1-l0)/l0,'|g|',np.linalg.norm(g0),'rel dg',np.linalg.norm(g1-g0)/np.linalg.norm(g0))
    # which points dominate the gradient
    gn=np.linalg.norm(g0,axis=1); idx=np.argsort(-gn)[:5]; print('   top vertex grads',gn[idx], 'share of top 20', (np.sort(gn)[-20:]**2).sum()/(gn**2).sum())
