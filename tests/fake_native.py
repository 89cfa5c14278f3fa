"""A fake native batch for host-logic tests (no GPU): records every physics call; sensordata = running substep count (like the
stub simulator of the harness fixture).  Test infrastructure only: tests/run_with_fake_native.py puts it in the library's place; no product file knows it."""
import numpy as np
import torch


class FakeModel:
    def __init__(self, model):
        self.model, self.nq, self.nu, self.nsensordata, self.ntendon, self.nelem = model, model.nv, model.nu, 12, 3, 110
        self.nboxes = 4
        self.nv = model.nv


class FakeBatch:
    """records every physics call; sensordata = running substep count (like the stub simulator of the fixture)"""
    log = []

    def __init__(self, nmodel, n_envs, device=0):
        self.n, self.nmodel, self.device = n_envs, nmodel, torch.device("cpu")
        self.ctrl = np.zeros(2)
        self.nsub = 0
        self.k = None

    def set_stiffness(self, k, jids, tids):
        self.k = np.array(k, copy=True)
        FakeBatch.log.append(("stiffness", list(jids), list(tids), self.k.copy()))

    def set_ctrl_broadcast(self, c):
        self.ctrl[:] = c

    def reset(self, sim_start, sens=None, flags=None, touch=None, mask=None):
        FakeBatch.log.append(("reset", sim_start))
        self.ctrl[:] = 0
        self.nsub = 0
        self._advance(sim_start, sens, flags, touch)

    def step(self, n, sens=None, sens_stride=0, flags=None, touch=None):
        self._advance(n, sens, flags, touch)

    def _advance(self, n, sens, flags, touch):
        for _ in range(n):
            self.nsub += 1
            FakeBatch.log.append(("substep", float(self.ctrl[0]), float(self.ctrl[1])))
        if sens is not None:
            sens[:] = self.nsub
        if flags is not None:
            flags.zero_()
        if touch is not None:
            touch.zero_()

    def solver_stats(self):
        return dict(ncon=torch.zeros(self.n, dtype=torch.int32))

    def profile_enable(self, on=True):
        pass

    def profile_read(self, reset=True):
        return 0.0, 0

    def profile_read_solver(self, reset=True):
        return 0.0, 0
