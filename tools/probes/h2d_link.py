import torch, time
dev='cuda'
n=1<<30
hs=[torch.empty(n,dtype=torch.uint8).pin_memory() for _ in range(2)]
ds=[torch.empty(n,dtype=torch.uint8,device=dev) for _ in range(2)]
ss=[torch.cuda.Stream() for _ in range(2)]
def run(k, reps=4):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps):
        for i in range(k):
            with torch.cuda.stream(ss[i]): ds[i].copy_(hs[i], non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    return k*reps*n/dt/1e9
run(1,1); run(2,1)
print("h2d 1 stream GB/s", round(run(1),1)); print("h2d 2 streams GB/s", round(run(2),1))
# d2h concurrently with h2d
def both(reps=4):
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(ss[0]): ds[0].copy_(hs[0], non_blocking=True)
        with torch.cuda.stream(ss[1]): hs[1].copy_(ds[1], non_blocking=True)
    torch.cuda.synchronize(); dt=time.perf_counter()-t
    return reps*n/dt/1e9
print("h2d beside d2h GB/s each", round(both(),1))
