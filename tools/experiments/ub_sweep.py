import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import sys, os
sys.path.insert(0, %r)
import torch, __graft_entry__ as g
pkg = g.load_package()
dev = torch.device("cuda:0")
n = 1 << 30
hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
a, b, c, d = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
a.fill_(1); b.fill_(2); c.fill_(0); d.fill_(0)
torch.cuda.synchronize()
print(os.environ.get("HJ_UB_BLOCKS"), "nt", os.environ.get("HJ_UB_NT"), "copy %%.0f  line_scatter %%.0f GB/s" %% (hj.ubench("copy", a, b, c, d, n), hj.ubench("line_scatter", a, b, c, d, n)))
''' % ROOT
for blocks, nt in ((16384, 0), (16384, 240), (65536, 0), (65536, 240), (16384, 243), (16384, 241)):
    if True:
        env = dict(os.environ, HJ_UB_BLOCKS=str(blocks), HJ_UB_NT=str(nt))
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
        print(r.stdout.strip() or r.stderr[-500:])
