"""What tests/_config4_worker.py's rank processes do, step by step, printing the GPU device nodes the process holds."""
import datetime, os, sys, time
def gpu_fds():
    out = []
    for fd in os.listdir("/proc/self/fd"):
        try:
            t = os.readlink(f"/proc/self/fd/{fd}")
        except OSError:
            continue
        if "kfd" in t or "/dev/dri" in t:
            out.append(t)
    return sorted(out)
R = os.environ.get("RANK", "?")
def say(what):
    print(f"rank {R} {what}: {gpu_fds()}", flush=True)
import numpy as np
import torch
import torch.distributed as dist
say("import torch")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge
dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
say("init gloo")
p25 = ge.load_package()
from plonky25_amd import aggregate as ag, dist as pd
say("load_package")
t = torch.tensor([1], dtype=torch.int32); dist.all_reduce(t, op=dist.ReduceOp.MIN)
say("all_reduce int32 MIN")
g = pd.ProofGatherer(8, 5, torch.device("cpu"))
blocks, sts = g.gather(torch.zeros((8 // dist.get_world_size(), 5), dtype=torch.int64), torch.zeros(8 // dist.get_world_size(), dtype=torch.int32))
say("ProofGatherer.gather")
recs = [None] * dist.get_world_size()
dist.all_gather_object(recs, {"a": 1})
say("all_gather_object")
t = torch.tensor([1], dtype=torch.int32); dist.all_reduce(t, op=dist.ReduceOp.MIN)
say("barrier")
dist.destroy_process_group()
say("destroy")
