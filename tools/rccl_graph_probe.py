#!/usr/bin/env python3
"""Can RCCL collectives be captured into a hipGraph on this stack (PyTorch-ROCm, 1-rank communicator on one GPU)?  Every
variant runs in a child process (a crash must not take the probe down) and reports ok / the failure.
  python tools/rccl_graph_probe.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch, torch.distributed as dist
os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", "%d"
dev = torch.device("cuda:0"); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
variant = "%s"
x = torch.arange(1 << 20, dtype=torch.float32, device=dev).bfloat16()
y = torch.zeros_like(x)
def step():
    if variant == "a2a_sync":
        dist.all_to_all_single(y, x)
    elif variant == "a2a_async":
        h = dist.all_to_all_single(y, x, async_op=True); h.wait()
    elif variant == "allgather_sync":
        dist.all_gather_into_tensor(y, x)
    elif variant == "a2a_splits_async":
        h = dist.all_to_all_single(y, x, [x.numel()], [x.numel()], async_op=True); h.wait()
    y.mul_(2)
step(); torch.cuda.synchronize()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
y.zero_(); g.replay(); torch.cuda.synchronize()
print("RESULT ok" if torch.equal(y.float(), x.float() * 2) else "RESULT wrong")
dist.destroy_process_group()
'''
for i, v in enumerate(["a2a_sync", "a2a_async", "allgather_sync", "a2a_splits_async"]):
    r = subprocess.run([sys.executable, "-c", CHILD % (29871 + i, v)], capture_output=True, text=True, timeout=300,
                       env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")]
    print(v, line[0] if line else f"rc={r.returncode} {(r.stderr or r.stdout)[-300:]!r}", flush=True)
