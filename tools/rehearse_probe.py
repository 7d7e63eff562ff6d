#!/usr/bin/env python3
"""Where does the time of bench.py's one-GPU rehearsal (RT_BENCH_REHEARSE=1: gloo, every rank on GPU 0) go?
Times gloo collectives on host tensors, on device tensors, and on device tensors behind a kernel, 4 MB each.

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/rehearse_probe.py
"""
import os
import sys
import time

import torch
import torch.distributed as dist

dist.init_process_group("gloo")
rank = dist.get_rank()
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)


def say(msg):
    if rank == 0:
        print(msg, file=sys.stderr, flush=True)


def timed(label, fn, n=4):
    dist.barrier()
    torch.cuda.synchronize()
    for i in range(n):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        say(f"{label} #{i}: {(time.perf_counter() - t) * 1e3:9.2f} ms")


host = torch.ones(1 << 20)
devt = torch.ones(1 << 20, device=dev)
a = torch.randn(2048, 2048, device=dev)
timed("host all_reduce 4 MB", lambda: dist.all_reduce(host))
timed("host reduce 4 MB", lambda: dist.reduce(host, dst=0))
timed("device reduce 4 MB", lambda: dist.reduce(devt, dst=0))


def behind_kernel():
    (a @ a).sum()
    dist.reduce(devt, dst=0)


timed("device reduce behind a matmul", behind_kernel)
s = torch.cuda.Stream(device=dev)


def on_stream():
    with torch.cuda.stream(s):
        (a @ a).sum()
        dist.reduce(devt, dst=0)
    s.synchronize()


timed("device reduce on a non-default stream", on_stream)
timed("staged by hand (device -> host, reduce, host -> device)",
      lambda: (host.copy_(devt), dist.reduce(host, dst=0), devt.copy_(host)))
dist.barrier()
dist.destroy_process_group()
