import os, sys, time
import numpy as np
ROOT='/root/repo'
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
p25 = ge.load_package(); p25.device_init(0)
P=0xFFFFFFFF00000001
inp, cfg = p25.p3_proof_from_json(open(os.path.join(ROOT,'tests','golden','proof_fibonacci.json')).read())
c = p25.Circuit.build_p3_verifier(cfg)
n = inp.size
t=time.time()
bad_positions=[]
counts={}
B=1024
for s in range(0, n, B):
    idx = np.arange(s, min(n, s+B))
    batch = np.tile(inp, (idx.size, 1))
    batch[np.arange(idx.size), idx] = (batch[np.arange(idx.size), idx] + np.uint64(1)) % np.uint64(P)
    proofs, st = c.prove(batch, seeds=np.arange(idx.size, dtype=np.uint64))
    for v in np.unique(st): counts[int(v)] = counts.get(int(v),0)+int((st==v).sum())
    bad_positions += [int(i) for i in idx[st==0]]
print('positions', n, 'status histogram', counts, 'accepted flips', len(bad_positions), bad_positions[:40], 'seconds', round(time.time()-t,1))
