"""A/B of two builds of the library: are their results the SAME BITS?  One whole episode of each scene at bench.py's batch (4096 envs, the
seeded stiffness draw) under each library -- every library in a process of its own, SOFTGRIP_LIB -- and the SHA-256 of the episode's whole
sensor block [envs][200][nsensordata] plus the flags.  For a change that re-orders work without changing any sum (r05: the delayed second
contact stream in place of the second pass, sg_rows.hip) the digests must be equal.
usage: python scripts/dev/ab_bits.py <libA.so>[@VAR=value] <libB.so>[@VAR=value] [scene ...]     (@VAR=value: an environment variable for that side)"""
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(scene, n):
    sys.path.insert(0, ROOT)
    import bench
    r = bench.Runner(scene, n, 0, 0, 1)
    for t in range(r.T):
        r.step(t)
    r.torch.cuda.synchronize()
    h = hashlib.sha256(r.out.cpu().numpy().tobytes())
    h.update(r.flags_or.cpu().numpy().tobytes())
    print(json.dumps({"scene": scene, "sha256": h.hexdigest(), "flagged": int((r.flags_or != 0).sum())}))


if __name__ == "__main__":
    if sys.argv[1] == "--one":
        one(sys.argv[2], int(sys.argv[3]))
        sys.exit(0)
    A, B = sys.argv[1], sys.argv[2]
    scenes = sys.argv[3:] or ["softbox", "softball", "softcylinder"]
    bad = 0
    for s in scenes:
        got = []
        for L in (A, B):
            L, _, kv = L.partition("@")
            env = dict(os.environ, SOFTGRIP_LIB=os.path.abspath(L))
            if kv:
                env[kv.split("=")[0]] = kv.split("=")[1]
            out = subprocess.run([sys.executable, os.path.abspath(__file__), "--one", s, os.environ.get("SG_AB_ENVS", "4096")],
                                 env=env, capture_output=True, text=True, timeout=600)
            if out.returncode != 0:
                print(out.stderr[-2000:])
                sys.exit(2)
            got.append(json.loads(out.stdout.strip().splitlines()[-1]))
        same = got[0] == got[1]
        bad += not same
        print(s, "SAME BITS" if same else "DIFFERENT", got[0]["sha256"][:16], got[1]["sha256"][:16], "flagged", got[0]["flagged"], got[1]["flagged"])
    sys.exit(1 if bad else 0)
