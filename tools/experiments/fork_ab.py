import os,sys,time,torch
sys.path.insert(0,os.getcwd())
import __graft_entry__ as g
pkg=g.load_package()
for l in (24,26,28,29,30):
    n=1<<l
    Rk,Rp,Sk,Sp=(torch.empty(n,dtype=torch.int32,device="cuda") for _ in range(4))
    res={}
    for rep in range(3):
        for f in ("25","40"):
            os.environ["HJ_FORK_LOG2"]=f
            hj=pkg.HashJoin(0)
            hj.gen_unique(Rk,n,0,n,1); hj.gen_unique(Sk,n,0,n,2); hj.fill_payload(Rp,n,"ones"); hj.fill_payload(Sp,n,"ones"); hj.sync()
            hj.bind_device(0,Rk,Rp); hj.bind_device(1,Sk,Sp)
            for _ in range(3): assert hj.join()[0]==n
            torch.cuda.synchronize(); t0=time.perf_counter()
            steps=20 if l<30 else 10
            for _ in range(steps): hj.join()
            dt=(time.perf_counter()-t0)/steps
            res.setdefault(f,[]).append(dt*1e3)
            hj.close()
    print("2^%d  serial ms %s   forked ms %s" % (l, ["%.3f"%x for x in res["25"]], ["%.3f"%x for x in res["40"]]), flush=True)
    del Rk,Rp,Sk,Sp
