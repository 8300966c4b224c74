#!/usr/bin/env python3
"""Experiment: two resident batches (two ABI contexts, each with its own streams) stepped alternately on ONE GPU, against one
context stepped back to back.  Shows how much of the step is latency that a second, independent batch can fill."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
torch.cuda.set_device(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
s = [torch.cuda.Stream(), torch.cuda.Stream()]
p = [bench.Pipeline(B, 0, r, stream=s[r].cuda_stream) for r in range(2)]
for q in p:
    q.setup(); q.step(); q.ctx.synchronize()
def run(pipes, steps):
    for q in pipes: q.ctx.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        pipes[i % len(pipes)].step()
    for q in pipes: q.ctx.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3
for rep in range(2):
    print("one context : %.3f ms per step" % run(p[:1], 8))
    print("two contexts: %.3f ms per step" % run(p, 8))
