"""The per-hop step with HOST buffers on either side (the C ABI takes device pointers; this is what a caller holding its
audio in host memory pays): pinned host input -> device, fe_step, device -> pinned host output, per step.
   serial:    copy in, step, copy out, synchronise - every step (a caller that needs hop t's output before hop t + 1 arrives)
   pipelined: the same three operations queued on one stream for all steps, one synchronise at the end
   overlapped: fe_step_host - the copies of the neighbouring hops under each kernel (three streams, events; enqueued in C++)
Prints frames/s next to the device-resident rate bench.py reports.  tools/pcie_inclusive.py [workload] [streams]"""
import json
import sys
import time

import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from fastenhancer_amd.config import FEConfig  # noqa: E402
from fastenhancer_amd.engine import Engine  # noqa: E402
from fastenhancer_amd.weights import default_state_dict  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "fe_b"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w = bench.WORKLOADS[wl]
cfg = FEConfig.from_model_kwargs(**bench.model_kwargs(w))
dev = torch.device("cuda:0")
eng = Engine(cfg, dev)
eng.load_state_dict(default_state_dict(cfg, torch.Generator().manual_seed(2)))
H = cfg.hop_size
steps = 400
x_host = (0.1 * torch.randn(steps, B, H)).pin_memory()
y_host = torch.empty(steps, B, H).pin_memory()
x_dev = torch.empty(B, H, device=dev)
y_dev = torch.empty(B, H, device=dev)
state = eng.new_state(B)
for t in range(50):
    eng.step(x_dev, state, y_dev)
torch.cuda.synchronize()


def run(mode):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        if mode != "resident":
            x_dev.copy_(x_host[t], non_blocking=True)
        eng.step(x_dev, state, y_dev)
        if mode != "resident":
            y_host[t].copy_(y_dev, non_blocking=True)
        if mode == "serial":
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


def run_overlapped(T):
    # fe_step_host: copy-in / kernel / copy-out of neighbouring hop blocks (T hops each) on three streams, enqueued by the library
    xh = x_host.permute(1, 0, 2).reshape(B, steps * H).contiguous().pin_memory()
    yh = torch.empty(B, steps * H).pin_memory()
    eng.step_host(xh[:, :8 * T * H], state, yh[:, :8 * T * H], T=T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.step_host(xh, state, yh, T=T)
    torch.cuda.synchronize()
    return B * steps / (time.perf_counter() - t0)


res = {m: run(m) for m in ("resident", "pipelined", "serial")}
res["overlapped"] = run_overlapped(1)
res["overlapped_4_hops_per_call"] = run_overlapped(4)
print(json.dumps({"workload": wl, "streams": B, "frames_per_s": {k: round(v) for k, v in res.items()},
                  "bytes_over_pcie_per_step": 2 * B * H * 4}))
