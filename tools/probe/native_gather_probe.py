"""Why is bench.py --dist-native slower than the torch.distributed gather on a world of one?  The bench's pipelined step loop with the
gather replaced by its parts, one variant per process run:  none | wait_torch | wait_comm | copy_comm | gather_comm | gather_comm_nowait
usage: native_gather_probe.py VARIANT [steps]"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import __graft_entry__ as ge
variant = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
p25 = ge.load_package(); p25.device_init(0)
from plonky25_amd import aggregate as pagg
dev = torch.device("cuda", 0)
inputs, cfg = p25.p3_proof_from_json(open(os.path.join(ROOT, "tests", "golden", "proof_fibonacci.json")).read())
circuit = p25.Circuit.build_p3_verifier(cfg); circuit.digest()
B, pw = 256, int(circuit.info.proof_words)
d_in = torch.from_numpy(np.stack([inputs] * B).view(np.int64)).to(dev)
d_seeds = torch.arange(B, dtype=torch.int64, device=dev)
d_proofs = [torch.zeros((B, pw), dtype=torch.int64, device=dev) for _ in range(2)]
d_all = [torch.zeros((B, pw), dtype=torch.int64, device=dev) for _ in range(2)]
d_st = torch.zeros((steps + 3, B), dtype=torch.int32, device=dev)
d_all_st = [torch.zeros(B, dtype=torch.int32, device=dev) for _ in range(2)]
comm = None
if "comm" in variant:
    comm = p25.Comm(p25.comm_unique_id(), 0, 1)
side = torch.cuda.Stream(device=dev)
MARK = pagg.TREE_MARK_SLOTS
host = {"prove": 0.0, "gather": 0.0}
def issue(k):
    buf = k & 1
    t = time.perf_counter()
    if variant == "wait_torch":
        circuit.stream_wait_mark(MARK + buf, side.cuda_stream)
    elif variant == "wait_comm":
        circuit.stream_wait_mark(MARK + buf, comm.stream)
    elif variant == "copy_comm":       # the wait, then a torch copy on the communicator's stream
        circuit.stream_wait_mark(MARK + buf, comm.stream)
        with torch.cuda.stream(torch.cuda.ExternalStream(comm.stream, device=dev)):
            d_all[buf].copy_(d_proofs[buf], non_blocking=True)
    elif variant == "gather_comm":
        comm.gather(circuit, MARK + buf, d_proofs[buf].data_ptr(), pw, d_st[k].data_ptr(), [B], 0, d_all[buf].data_ptr(), d_all_st[buf].data_ptr())
    elif variant == "gather_comm_nowait":      # no circuit: the gather's copies with no device-side wait in front
        comm.gather(None, -1, d_proofs[buf].data_ptr(), pw, d_st[k].data_ptr(), [B], 0, d_all[buf].data_ptr(), d_all_st[buf].data_ptr())
    host["gather"] += time.perf_counter() - t
issued = [0]
def step(k):
    buf = k & 1
    t = time.perf_counter()
    if k >= 2 and variant != "none":
        circuit.wait_stream(comm.stream if comm is not None else side.cuda_stream)
    circuit.prove_dev(d_in.data_ptr(), B, d_seeds.data_ptr(), d_proofs[buf].data_ptr(), pw, d_st[k].data_ptr())
    circuit.mark(MARK + buf)
    host["prove"] += time.perf_counter() - t
    while variant != "none" and issued[0] < k:
        issue(issued[0]); issued[0] += 1
for k in range(3):
    step(k)
circuit.sync(); torch.cuda.synchronize()
host = {"prove": 0.0, "gather": 0.0}
t0 = time.perf_counter()
for k in range(3, 3 + steps):
    step(k)
t_enq = time.perf_counter() - t0
circuit.sync()
if comm is not None: comm.sync()
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ok = bool((d_st.cpu().numpy() == 0).all())
print(f"{variant:20s} {B * steps / dt:7.2f} proofs/s   host: all steps enqueued after {t_enq * 1e3:7.1f} ms of {dt * 1e3:7.1f} ms, prove calls {host['prove'] * 1e3:6.1f} ms, gather calls {host['gather'] * 1e3:6.1f} ms  ok={ok}", flush=True)
