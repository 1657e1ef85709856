# two ranks on ONE GPU (if RCCL allows it): exercises the chunked P2P exchange + local join for real
import os, sys, json, faulthandler
faulthandler.dump_traceback_later(90, exit=True)
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
pkg = graft.load_package()
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
hj = pkg.HashJoin(0, stream=torch.cuda.current_stream().cuda_stream)
from importlib import import_module
dj = import_module(pkg.__name__ + ".dist").ShardedJoin(hj, pkg, dev)
dj.CHUNK = int(os.environ.get("HJ_DIST_CHUNK", dj.CHUNK))
n = 1 << int(sys.argv[1])
total = n * world
Rk = torch.empty(n, dtype=torch.int32, device=dev); Sk = torch.empty_like(Rk); Rp = torch.empty_like(Rk); Sp = torch.empty_like(Rk)
hj.gen_unique(Rk, n, rank * n, total, 1); hj.gen_unique(Sk, n, rank * n, total, 2)
hj.fill_payload(Rp, n, "ones"); hj.fill_payload(Sp, n, "ones"); hj.sync()
for i in range(3):
    m, agg = dj.join(Rk, Rp, Sk, Sp)
    assert m == total == agg, (m, total)
if rank == 0: print("P2P_OK world", world, "n", n, "matches", m)
dist.destroy_process_group()
