import faulthandler, os, sys, torch, torch.distributed as dist
faulthandler.dump_traceback_later(40, exit=True)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
n = 1 << int(sys.argv[1])
a = torch.arange(n, dtype=torch.int32, device="cuda"); b = torch.empty_like(a)
mode = sys.argv[2]
if mode == "single":
    dist.all_to_all_single(b, a, [n], [n])
elif mode == "even":
    dist.all_to_all_single(b, a)
elif mode == "copy":
    b.copy_(a)
torch.cuda.synchronize()
print(sys.argv[1:], "ok", bool((a == b).all()))
