"""CPU probe (oracle only) of the reference's ball / cylinder scenes: runs the squeeze schedule on the oracle until the first simulation
warning and reports it.  SGO_DEBUG_PAIR=1 (set below) makes the oracle print every box-box pair it generates contacts for.
usage: python scripts/scene_probe.py softball softcylinder > profiles/r02_ball_cylinder_probe.txt 2>&1"""
import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
os.environ['SGO_DEBUG_PAIR']='1'
import numpy as np, softgrip_amd as sg
from helpers import oracle_sim, model_path
from softgrip_amd.create_dataset import episode_schedule
for scene in sys.argv[1:]:
    m=sg.load_model(model_path(scene))
    print(scene, [ (i,n) for i,n in enumerate(m.geom_names[:10])])
    s=oracle_sim(m,700.0); s.reset(); s.forward(); w=s.step()
    nsub=1
    for t,c in enumerate(episode_schedule()):
        if c is not None: s.ctrl[:]=c
        for _ in range(7):
            w=s.step(); nsub+=1
            if w: break
        if w: print(scene,'flag',w,'at env step',t,'substep',nsub, 'max|q fingers|', np.abs(s.qpos[:8]).max(), 'qvel', np.abs(s.qvel[:8]).max()); break
    else: print(scene,'episode ok')
