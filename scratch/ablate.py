# timing-only: scatter kernel time under ablation masks (results are wrong by construction)
import os, sys, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
code = r'''
import os, sys, torch
sys.path.insert(0, %r)
import __graft_entry__ as g
P = g.load_package()
n = 1 << 30
dev = torch.device("cuda:0")
hj = P.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
k = torch.empty(n, dtype=torch.int32, device=dev); p = torch.empty(n, dtype=torch.int32, device=dev)
hj.gen_unique(k, n, 0, n, 1); hj.fill_payload(p, n, "ones"); hj.sync()
hj.bind_device(0, k, p); hj.bind_device(1, k[:16], p[:16])
hj.configure(bits1=9, bits2=9)
for _ in range(2): hj.partition(0)
hj.timings_reset()
for _ in range(4): hj.partition(0)
t = hj.timings()
print("ABL", os.environ.get("HJ_WC_ABLATE", "0"), round(t["k_scatter_wc"]["total_ms"] / t["k_scatter_wc"]["launches"], 3), round(t["k_hist"]["total_ms"]/t["k_hist"]["launches"], 3))
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for abl in [0, 2, 6, 0]:
    env = dict(os.environ, HJ_WC_ABLATE=str(abl))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:])
