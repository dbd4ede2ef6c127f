import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tools')
import numpy as np, torch
import _frames as TF
from smplifyx_amd import engine, synthetic
cfg = TF.load_cfg("fit_smplx_combined_halpe.yaml", interpenetration=True)
model = synthetic.make_topology_model(0); parts = synthetic.topology_parts()
dm, jm = TF.device_model(model, cfg)
N=64
tr=[synthetic.make_frame_truth(i) for i in range(N)]
dev=torch.device("cuda"); t=lambda a: torch.tensor(np.asarray(a,np.float32),device=dev); z=lambda k: torch.zeros([N,k],device=dev)
rng=np.random.RandomState(3)
verts,_,_=dm.lbs_forward(t([x["global_orient"] for x in tr]),t([x["body_pose"] for x in tr]),t([x["betas"] for x in tr]),z(dm.num_expr),z(3),z(3),z(3),t(0.5*rng.normal(size=(N,dm.num_pca))),t(0.5*rng.normal(size=(N,dm.num_pca))))
verts=verts.contiguous(); faces=np.asarray(model["f"]).astype(np.int64)
for B in (1,64):
    pen=engine.Penetration(verts.shape[1],faces,parts["segm"],parts["parents"],cfg["ign_part_pairs"],max_collisions=128,max_batch=B)
    for _ in range(3): pen.eval(verts[:B],1e-4)
    torch.cuda.synchronize()
    pc=pen.phase_clocks(B)
    print("B=%d: end of A %.1f | staged %.1f near %.1f clusters %.1f wave0 %.1f all %.1f | records %.0f survivors %.0f surviving clusters %.0f"%(B,pc[:,0].mean(),pc[:,1].mean(),pc[:,2].mean(),pc[:,3].mean(),pc[:,4].mean(),pc[:,5].mean(),(pc[:,6]*100).mean(),(pc[:,7]*100).mean(),(pc[:,8]*100).mean()))
