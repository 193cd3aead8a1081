#!/usr/bin/env python3
"""Objective history of the slowest instances of the bench workload, from the CPU oracle (no GPU needed): the trial
objective per iteration and the relative decrease of the best value.  The stragglers of the 20-step call converge
linearly (accepted steps, decreases falling from 1e-5 to 1e-8): one iteration per round whatever is speculated.
usage: python tools/trace_stragglers.py"""
import os, sys, json, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle
oracle.build()
from grasptrajopt_amd import synthetic as syn
from grasptrajopt_amd.robot_desc import load_builtin
cfg=json.load(open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'grasptrajopt_amd', 'data', 'panda_cfg.json')))
desc=load_builtin('panda_5k')
o=oracle.Oracle(desc,cfg['link_ee'],cfg['link_gripper'],oracle.reference_opts(),n_gripper_points=100) if 'n_gripper_points' in oracle.Oracle.__init__.__code__.co_varnames else oracle.Oracle(desc,cfg['link_ee'],cfg['link_gripper'],oracle.reference_opts())
sc=syn.make_scene(0,n=128,res=0.0175)
o.set_scene(0,sc.c_all,sc.c_obs,sc.shape,sc.origin,sc.res)
B=640
RT,qg=syn.make_goals(desc,o.eval_fk,cfg['link_ee'],B,seed=3)
qc=np.array(cfg['default_pose'])
Q0=np.stack([syn.make_seed(qc,qg[i],50,desc.param_index) for i in range(B)])
S=syn.standoff_pose(-0.1,'z')
idx=[515,162,261,527,172,78, 0,1,2]
res=o.solve_batch(0,np.tile(qc,(len(idx),1)),RT[idx].reshape(len(idx),1,16),1,S,[0,0,0],Q0[idx],trace=True)
Q,dQ,f,it,st=res[:5]; tr=res[5]
print('iters',it.tolist(),'status',st.tolist())
for k,i in enumerate(idx):
    t=tr[k]; n=it[k]
    t=t[:n+1]
    print('inst',i,'iters',n)
    print('  f_try:', ' '.join(f'{x:.6g}' for x in t[::max(1,n//25)]))
    acc=np.minimum.accumulate(t)
    rel=(acc[:-1]-acc[1:])/(1+acc[1:])
    print('  rel decrease of best f per iteration (every 4th):', ' '.join(f'{x:.1e}' for x in rel[::4]))
