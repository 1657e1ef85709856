#!/usr/bin/env python3
"""tools/experiments/layout_gate.py [log2n] [rounds] [reps] — the layout gate of round 6: hj_ubench kinds 0-7 in ONE
process on the SAME 16 GiB (one allocation per side; kinds 0-3 see its halves as the two columns, kinds 4-7 see one array), the kinds
interleaved round-robin, median GB/s per kind.  Prints one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft  # noqa: E402

KINDS = ["copy", "line_scatter", "read", "write", "copy1", "pairs", "pairs_scatter", "soa_to_pairs_scatter"]


def main():
    import torch
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    pkg = graft.load_package()
    n = 1 << log2n
    dev = torch.device("cuda", 0)
    hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
    src = torch.empty(2 * n, dtype=torch.int32, device=dev)
    dst = torch.empty(2 * n, dtype=torch.int32, device=dev)
    src.fill_(1)
    dst.fill_(0)
    torch.cuda.synchronize()
    a, b, c, d = src[:n], src[n:], dst[:n], dst[n:]
    acc = {k: [] for k in KINDS}
    for _ in range(rounds):
        for k in KINDS:
            acc[k].append(hj.ubench(k, a, b, c, d, n, reps=reps))
    med = {k: round(sorted(v)[len(v) // 2], 1) for k, v in acc.items()}
    print(json.dumps({"log2n": log2n, "rounds": rounds, "reps": reps, "median_GBs": med,
                      "vs_copy": {k: round(med[k] / med["copy"], 4) for k in KINDS},
                      "all_GBs": {k: [round(x, 1) for x in v] for k, v in acc.items()},
                      "src_ptr": hex(src.data_ptr()), "dst_ptr": hex(dst.data_ptr())}))


if __name__ == "__main__":
    main()
