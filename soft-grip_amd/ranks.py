"""One process per GPU without a collective library (envs never interact -- the path shards by env / stiffness bin and
north_star asks for "no RCCL collectives"; BASELINE configs[3]).

Two pieces, used by bench.py and create_dataset.py:

* ``spawn_ranks(n)``   -- `--gpus N` without a launcher: the parent, which never touches a GPU, starts N copies of ITS OWN command line
  as child processes, rank r with RANK = LOCAL_RANK = r, WORLD_SIZE = N, MASTER_ADDR = 127.0.0.1 and a free MASTER_PORT (the variables
  a `torch.distributed.run` launch sets, so a rank does not care who started it), waits for them and returns the worst exit code;
  one failing rank ends the others.
* ``RankGroup``        -- what the ranks need from each other: a common start (``barrier``) and each other's timings (``gather``).
  Both go through a ``torch.distributed.TCPStore`` on MASTER_ADDR:MASTER_PORT -- a key-value socket server (rank 0's, or the launcher
  agent's when `torch.distributed.run` started the ranks), no process group, no NCCL / RCCL communicator, nothing on the GPUs.
"""
import datetime
import os
import socket
import subprocess
import sys
import time


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_as_rank():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def spawn_ranks(n, extra_env=None, poll=0.05):
    """start n copies of this process's own command line (sys.orig_argv) as ranks 0 .. n-1; -> exit code (0 iff all ranks returned 0)"""
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                    "MASTER_PORT": str(port)})
        env.setdefault("OMP_NUM_THREADS", "1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
        env.update(extra_env or {})
        procs.append(subprocess.Popen(list(sys.orig_argv), env=env))
    rc = 0
    live = list(procs)
    while live:
        time.sleep(poll)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:       # a rank that died would leave the others waiting at the next barrier: end them (exact PIDs)
                    q.terminate()
    return rc


class RankGroup:
    """barrier + gather over a TCPStore; a world of one needs no store at all"""

    def __init__(self, rank=None, world=None, timeout_s=1800):
        self.rank = int(os.environ.get("RANK", 0)) if rank is None else rank
        self.world = int(os.environ.get("WORLD_SIZE", 1)) if world is None else world
        self.store, self._n, self._hosts = None, 0, False
        if self.world > 1:
            from torch.distributed import TCPStore
            addr, port = os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ["MASTER_PORT"])
            to = datetime.timedelta(seconds=timeout_s)
            agent = os.environ.get("TORCHELASTIC_USE_AGENT_STORE", "False") == "True"   # torch.distributed.run: the agent serves the port
            if self.rank == 0 and not agent:
                try:
                    self.store = TCPStore(addr, port, self.world, True, to, wait_for_workers=False)
                    self._hosts = True
                except Exception:  # noqa: BLE001 -- the port is served already (a launcher's store): join it as a client
                    self.store = None
            if self.store is None:
                self.store = TCPStore(addr, port, self.world, False, to, wait_for_workers=False)

    def _key(self, tag, r):
        return "softgrip/%s/%d" % (tag, r)

    def barrier(self):
        if self.store is None:
            return
        self._n += 1
        tag = "barrier%d" % self._n
        self.store.set(self._key(tag, self.rank), "1")
        self.store.wait([self._key(tag, r) for r in range(self.world)])

    def gather(self, tag, value):
        """every rank's float, on every rank"""
        if self.store is None:
            return [float(value)]
        self.store.set(self._key(tag, self.rank), repr(float(value)))
        keys = [self._key(tag, r) for r in range(self.world)]
        self.store.wait(keys)
        return [float(self.store.get(k).decode()) for k in keys]

    def close(self):
        """the rank that serves the store leaves last"""
        if self.store is None:
            return
        if self._hosts:
            self.store.wait([self._key("done", r) for r in range(1, self.world)])
        else:
            self.store.set(self._key("done", self.rank), "1")
        self.store = None
