import os, sys, time
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT="/root/repo"; sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import __graft_entry__ as ge

p25 = ge.load_package(); p25.device_init(0)
inputs, _ = p25.p3_proof_from_json(open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")).read())
c = p25.Circuit.build_p3_verifier(p25.P3Config.fib64())
info = c.info; c.digest()
B=256; pw=int(info.proof_words)
dev=torch.device("cuda",0)
d_in=torch.from_numpy(np.stack([inputs]*B).view(np.int64)).to(dev)
d_seeds=torch.arange(B,dtype=torch.int64,device=dev)
d_proofs=torch.zeros((B,pw),dtype=torch.int64,device=dev)
d_status=torch.zeros(B,dtype=torch.int32,device=dev)
for it in range(3):
    torch.cuda.synchronize()
    t0=time.perf_counter()
    c.prove_dev(d_in.data_ptr(), B, d_seeds.data_ptr(), d_proofs.data_ptr(), pw, d_status.data_ptr())
    t1=time.perf_counter()
    c.sync()
    t2=time.perf_counter()
    print(f"enqueue {t1-t0:.3f} s, total {t2-t0:.3f} s, {B/(t2-t0):.1f} proofs/s")
