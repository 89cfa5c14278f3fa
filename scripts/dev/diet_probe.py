import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import softgrip_amd as sg
from softgrip_amd import native
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
m = sg.load_model(os.path.join(ROOT, "models", "fourfinger_softball_fix.sgmodel"), "implicit")
nm = native.NativeModel(m)
b = native.NativeBatch(nm, 2, 0)
b.set_stiffness(np.array([300.0, 1400.0]), list(range(65, 283)), [0])
flags = torch.zeros(2, dtype=torch.int32, device=b.device)
b.reset(1, flags=flags)
q = b.get_state()["qpos"].cpu().numpy()
print(os.environ.get("SOFTGRIP_LIB", "default"), "flags", flags.tolist(), "chain qpos[0:3]", q[0, :3], "(oracle: 1.580521e-04 -4.539690e-06 0)")
