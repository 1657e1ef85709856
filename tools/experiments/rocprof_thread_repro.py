#!/usr/bin/env python3
"""A hypothesis that did NOT hold (profiles/r5_rocprof_suite_crash.txt): does rocprofv3's kernel trace survive a process that keeps
creating short-lived host threads which launch kernels?  It does — 6000 threads, no fault.  (What kills the profiled GPU suite is four or
more host threads submitting to one hardware queue: tools/experiments/rocprof_dist_bisect.sh.)
Usage (GPU box): rocprofv3 --kernel-trace --stats -d /tmp/x -- python3 tools/experiments/rocprof_thread_repro.py [threads]"""
import sys, threading
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
x = torch.zeros(1024, device="cuda")
def work(i):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        y = x + i
        y.sum().item()
for base in range(0, n, 8):
    th = [threading.Thread(target=work, args=(base + j,)) for j in range(8)]
    for t in th: t.start()
    for t in th: t.join()
    if base % 400 == 0:
        print("threads so far", base + 8, flush=True)
print("done", n)
