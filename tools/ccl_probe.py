# latency of the per-step header all-gather (1 rank is enough to see the software cost)
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1"); os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
s = torch.cuda.Stream(); torch.cuda.set_stream(s)
hd = torch.zeros((64, 8), dtype=torch.int64, device="cuda"); out = torch.zeros((64, 1, 8), dtype=torch.int64, device="cuda")
def run(n, async_op):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ws = []
    for k in range(n):
        w = dist.all_gather_into_tensor(out[k % 64], hd[k % 64:k % 64 + 1], async_op=async_op)
        if async_op: ws.append(w)
    t1 = time.perf_counter()
    for w in ws: w.wait()
    torch.cuda.synchronize(); t2 = time.perf_counter()
    return (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6
for a in (False, True, False, True):
    run(10, a); print("async" if a else "sync", "enqueue us %.1f total us %.1f" % run(50, a))
dist.destroy_process_group()
