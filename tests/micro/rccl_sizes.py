"""GPU box helper: messages of whole-genome size to this rank itself through the real RCCL, in pieces of several sizes.
usage: rccl_sizes.py   (MUMEMTO_RCCL_CHUNK is read once per process: one subprocess per piece size)"""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import sys, json
sys.path.insert(0, %r)
import mumemto_amd
from mumemto_amd import synth
eng = mumemto_amd.Engine(0)
eng.set_docs(synth.pangenome(4, 20000, 0.01, seed=2))
eng.run(merge_metadata=True)
comm = mumemto_amd.Comm(eng, 0, 1, mumemto_amd.Comm.unique_id())
for n, w in ((3_050_000_001, 4), (1 << 30, 4), ((1 << 30) - 1, 4), (395_000_000, 8), (395_000_000, 1), (1_200_000_000, 8)):
    print(json.dumps(dict(elements=n, width=w, **comm.selftest(n, w))), flush=True)
comm.close()
''' % ROOT
for chunk in ("", "268435456", "536870912", "1073741824"):
    env = dict(os.environ)
    if chunk:
        env["MUMEMTO_RCCL_CHUNK"] = chunk
    print("== pieces of %s" % (chunk + " elements" if chunk else "2^29 bytes (the default)"), flush=True)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-800:] if r.returncode else "", flush=True)
