"""Which step of a CPU-only rank process opens the GPU device nodes (/dev/kfd, /dev/dri/*)?  (the GPU box's process guard
counts processes that have the GPU open; tests/_config4_worker.py must keep its eight gloo ranks off the card)"""
import os, sys
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
print("start", gpu_fds(), {k: v for k, v in os.environ.items() if "VISIBLE" in k})
import numpy
print("numpy", gpu_fds())
import torch
print("import torch", gpu_fds())
import torch.distributed as dist
print("import dist", gpu_fds())
os.environ.update(RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533")
dist.init_process_group("gloo")
print("gloo init", gpu_fds())
t = torch.ones(4); dist.all_reduce(t)
print("all_reduce", gpu_fds())
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge
p25 = ge.load_package()
print("load_package", gpu_fds())
p25.lib()
print("lib()", gpu_fds())
inp, cfg = p25.p3_prove_fibonacci(3, 3, 4)
print("host call", gpu_fds())
print("device_count", torch.cuda.device_count(), gpu_fds())
