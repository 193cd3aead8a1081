import os, sys, json, numpy as np
sys.path.insert(0,'/root/repo')
from grasptrajopt_amd import _capi, synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin
cfg=json.load(open('/root/repo/grasptrajopt_amd/data/panda_cfg.json'))
desc=load_builtin('panda_5k')
opts=_capi.default_opts()
h=_capi.SolverHandle(desc,cfg['link_ee'],cfg['link_gripper'],opts,device=0,n_gripper_points=100)
sc=syn.make_scene(0,n=128,res=0.0175)
h.set_scene(0,sc.c_all,sc.c_obs,sc.shape,sc.origin,sc.res)
B=int(os.environ.get("B","64"))
RT,qg=syn.make_goals(desc,h.eval_fk,cfg['link_ee'],B,seed=3)
qc=np.array(cfg['default_pose'])
Q0=np.stack([syn.make_seed(qc,qg[i],50,desc.param_index) for i in range(B)])
S=syn.standoff_pose(-0.1,'z')
for r in range(int(os.environ.get("REPS","2"))):
    Q,dQ,f,it,st=h.solve_batch(0,np.tile(qc,(B,1)),RT.reshape(B,1,16),1,S,[0,0,0],Q0)
print('iters mean',it.mean(),'max',it.max())
