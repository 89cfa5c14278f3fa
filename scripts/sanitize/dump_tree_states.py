"""Dumps what scripts/sanitize/tree_msan_main.cpp replays: a model blob and, per physics substep of the oracle's run of the reference
schedule, qpos | qvel | act | qacc_warmstart | ctrl.  usage: dump_tree_states.py <scene | fuzz5> [env steps] [out dir]
(fuzz5: the random gripper with a free object and neighbour rows that exposed the r04 miscompile)"""
import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import softgrip_amd as sg
from helpers import random_gripper_xml, oracle_sim, model_path
from softgrip_amd.create_dataset import episode_schedule
which = sys.argv[1]
OUT = sys.argv[3] if len(sys.argv) > 3 else "/tmp/sg_msan"
os.makedirs(OUT, exist_ok=True)
if which == "fuzz5":
    free, neighbors = True, True
    rng = np.random.RandomState(40 + 2 * int(free) + int(neighbors))
    for i in range(6):
        xml = random_gripper_xml(rng, free)
        open(os.path.join(OUT, 'g%d.xml' % i), 'w').write(xml)
        m = sg.compile_mjcf(os.path.join(OUT, 'g%d.xml' % i), composite_neighbors=neighbors)
        nchain = int(np.flatnonzero(m.jnt_type != 3)[0])
        jids = [j for j in range(nchain, m.njnt) if m.jnt_type[j] == 2]
        ks = list(rng.uniform(300, 1400, 3))
    k = ks[0]; nsteps = 4
else:
    m = sg.load_model(model_path(which), "implicit")
    jids = list(range(65, 283)) if which.startswith("fourfinger") else list(range(9, 227)) if which.startswith("freeball") else list(range(11, 64))
    k = 700.0; nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
s = oracle_sim(m); s.jnt_stiffness[jids] = k; s.tendon_stiffness[0] = k; s.reset(); s.forward(); s.step()
blob = m.to_blob()
open(os.path.join(OUT, which + ".blob"), "wb").write(blob)
with open(os.path.join(OUT, which + ".states"), "wb") as f:
    f.write(struct.pack("dii", k, len(jids), 1)); f.write(np.array(jids, dtype=np.int32).tobytes()); f.write(np.array([0], dtype=np.int32).tobytes())
    recs = []
    for t, c in enumerate(episode_schedule()[:nsteps]):
        if c is not None: s.ctrl[:] = c
        for j in range(7):
            recs.append(np.concatenate([s.qpos, s.qvel, s.act, s.qacc_warmstart, s.ctrl]).astype(np.float64))
            s.step()
    f.write(struct.pack("i", len(recs)))
    for r in recs: f.write(r.tobytes())
print(which, "nq", m.nq, "nv", m.nv, "nu", m.nu, "records", len(recs))
