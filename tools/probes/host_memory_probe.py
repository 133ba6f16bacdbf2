import mmap, numpy as np, time, os
print("cpus", len(os.sched_getaffinity(0)))
os.system("free -g | head -2; cat /sys/kernel/mm/transparent_hugepage/enabled; nproc; ulimit -l")
sz=1<<30
t0=time.time(); a=np.empty(sz//8,dtype=np.uint64); a.fill(1); print("np.empty+fill 1GiB", time.time()-t0)
t0=time.time(); a.fill(2); print("second fill", time.time()-t0)
import torch
t0=time.time(); p=torch.empty((sz//8,),dtype=torch.int64,pin_memory=True); print("pinned alloc 1GiB", time.time()-t0)
t0=time.time(); p.fill_(1); print("pinned fill", time.time()-t0)
d=torch.empty_like(p,device="cuda:0")
torch.cuda.synchronize(); t0=time.time(); d.copy_(p); torch.cuda.synchronize(); print("h2d 1GiB pinned", time.time()-t0)
t=torch.from_numpy(a.view(np.int64))
torch.cuda.synchronize(); t0=time.time(); d.copy_(t); torch.cuda.synchronize(); print("h2d 1GiB pageable", time.time()-t0)
