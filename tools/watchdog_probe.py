"""What the process group's flight recorder says about a collective over time (one rank over RCCL): the entry keys, and when `state` /
`retired` change after the device has finished -- the observable `parallel.drain_watchdog` waits on."""
import os
import pickle
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from importlib import import_module  # noqa: E402

par = import_module("qa-vit_amd.parallel")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
par.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import torch.distributed as dist  # noqa: E402
from torch._C import _distributed_c10d as c10d  # noqa: E402

x = torch.ones(1 << 20, device="cuda")
for _ in range(3):
    dist.all_reduce(x)
torch.cuda.synchronize()
t0 = time.monotonic()
first = True
for i in range(40):
    d = pickle.loads(c10d._dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=False))
    a = pickle.loads(c10d._dump_nccl_trace(includeCollectives=True, includeStackTraces=False, onlyActive=True))
    es = d.get("entries", [])
    if first and es:
        print("entry keys:", sorted(es[0].keys()))
        first = False
    print(f"t={time.monotonic() - t0:.3f}s all={len(es)} active={len(a.get('entries', []))} states={[e.get('state') for e in es]} retired={[e.get('retired') for e in es]}", flush=True)
    if es and all(e.get("retired") for e in es):
        break
    time.sleep(0.02)
print("drain says:", par.drain_watchdog())
dist.destroy_process_group()
