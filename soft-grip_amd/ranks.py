"""One process per GPU without a collective library (envs never interact -- the path shards by env / stiffness bin and
north_star asks for "no RCCL collectives"; BASELINE configs[3]).

Two pieces, used by bench.py and create_dataset.py:

* ``spawn_ranks(n, cmd)`` -- `--gpus N` without a launcher: the parent, which never touches a GPU, starts N child processes running
  ``cmd`` (default: ITS OWN command line, ``sys.orig_argv`` -- right for a script started from the shell, wrong for a function called
  with an explicit argv, which passes the command it means), rank r with RANK = LOCAL_RANK = r, WORLD_SIZE = N, MASTER_ADDR =
  127.0.0.1 and a free MASTER_PORT (the variables a `torch.distributed.run` launch sets, so a rank does not care who started it),
  waits for them and returns the worst exit code; one failing rank ends the others (SIGTERM to the exact PIDs, SIGKILL after a grace
  period), and so does anything that takes the parent out of its wait (KeyboardInterrupt, SIGTERM, a failed ``Popen``).
* ``RankGroup``           -- what the ranks need from each other: a common start (``barrier``) and each other's timings (``gather``).
  Both go through a ``torch.distributed.TCPStore`` on MASTER_ADDR:MASTER_PORT -- a key-value socket server (rank 0's, or the launcher
  agent's when `torch.distributed.run` started the ranks), no process group, no NCCL / RCCL communicator, nothing on the GPUs.
"""
import datetime
import errno
import os
import signal
import socket
import subprocess
import sys
import time


def free_port():
    """a port that was free a moment ago (nothing holds it between this call and rank 0's bind: RankGroup refuses to become a client of
    whatever else may have taken it unless that is a launcher's store)"""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launched_as_rank():
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def _end(procs, grace):
    """SIGTERM to every live child (exact PIDs), SIGKILL to what is still there after `grace` seconds; reaps them all"""
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            p.terminate()
        except OSError:
            pass
    deadline = time.monotonic() + grace
    for p in live:
        try:
            p.wait(max(0.0, deadline - time.monotonic()))
        except subprocess.TimeoutExpired:
            try:
                p.kill()
            except OSError:
                pass
            p.wait()


def spawn_ranks(n, extra_env=None, poll=0.05, cmd=None, timeout_s=None, grace_s=10.0):
    """start n copies of `cmd` (default: this process's own command line, sys.orig_argv) as ranks 0 .. n-1; -> exit code (0 iff all
    ranks returned 0; 124 when `timeout_s` ran out first).  No rank outlives this call."""
    cmd = list(sys.orig_argv) if cmd is None else list(cmd)
    port = free_port()
    procs = []
    rc = 0
    t_end = None if timeout_s is None else time.monotonic() + timeout_s
    old_term = None
    try:
        # SIGTERM to the parent must not orphan N GPU processes: turn it into an exception the finally clause sees
        if hasattr(signal, "SIGTERM"):
            try:
                old_term = signal.signal(signal.SIGTERM, lambda s, f: (_ for _ in ()).throw(KeyboardInterrupt()))
            except ValueError:      # not the main thread: the caller's handler stays
                old_term = None
        for r in range(n):
            env = dict(os.environ)
            env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n), "MASTER_ADDR": "127.0.0.1",
                        "MASTER_PORT": str(port)})
            env.setdefault("OMP_NUM_THREADS", "1")
            env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            env.pop("TORCHELASTIC_USE_AGENT_STORE", None)
            env.update(extra_env or {})
            procs.append(subprocess.Popen(cmd, env=env))
        live = list(procs)
        while live:
            time.sleep(poll)
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code < 0:
                    code = 128 - code     # killed by a signal: the shell's convention (SIGKILL -> 137), a valid exit status
                if code != 0 and rc == 0:
                    rc = code
                    _end(live, grace_s)   # a rank that died would leave the others waiting at the next barrier: end them
            if t_end is not None and live and time.monotonic() > t_end:
                rc = rc or 124
                _end(live, grace_s)
    finally:
        _end(procs, grace_s)
        if old_term is not None:
            signal.signal(signal.SIGTERM, old_term)
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
                except Exception as e:  # noqa: BLE001 -- torch raises RuntimeError / DistNetworkError with the errno in its text
                    if not _address_in_use(e):
                        raise
                    # the port is served already (a launcher's store that did not announce itself): join it as a client
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


def _address_in_use(exc):
    if isinstance(exc, OSError) and exc.errno == errno.EADDRINUSE:
        return True
    text = str(exc).lower()
    return "address already in use" in text or "eaddrinuse" in text or "errno: 98" in text
