#!/usr/bin/env python3
"""Gate experiment: can pass-2 output stay on chip?  Producer (streamed input -> 128-byte lines scattered
inside a W-MiB window) followed by a dependent consumer streaming the window back, W = 16 ... 1024 MiB with the SAME window
reused every round (fits the 256 MiB Infinity Cache or not), against the same rounds walking a 16 GiB ring (every window comes
from / goes to HBM).  2^30 tuples (8 GiB key+payload) per repetition."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as g  # noqa: E402

pkg = g.load_package()
hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
n = 1 << 30
ik, ip = (torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(2))
hj.gen_unique(ik, n, 0, n, 1)
hj.fill_payload(ip, n, "rowid")
ring = 1 << 31
rk, rp = (torch.empty(ring, dtype=torch.int32, device="cuda") for _ in range(2))
hj.sync()
copy = hj.ubench("copy", ik, ip, rk, rp, n)
scat = hj.ubench("line_scatter", ik, ip, rk, rp, n)
print("same box: two-column stream copy %.0f GB/s, line scatter %.0f GB/s" % (copy, scat))
print("%-10s %-28s %10s %10s %8s" % ("window", "ring", "ms/8GiB", "GB/s", "rounds"))
for lg in (21, 22, 23, 24, 25, 26, 27):           # window tuples: 2^21 (16 MiB of key+payload) ... 2^27 (1 GiB)
    w = 1 << lg
    row = []
    for name, rg in (("same window (on chip?)", w), ("16 GiB ring (through HBM)", ring)):
        ms, gbs = hj.ubench_handoff(ik, ip, rk, rp, n, w, rg)
        row.append(ms)
        print("%-10s %-28s %10.3f %10.0f %8d" % ("%d MiB" % (w * 8 >> 20), name, ms, gbs, n // w))
    print("%-10s on-chip / through-HBM time ratio: %.3f" % ("", row[0] / row[1]))
