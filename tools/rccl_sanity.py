"""RCCL sanity on a one-GPU box: a world-size-1 nccl group created the way bench.py creates its group
(device_id=...), a sum all-reduce of a Cornell-sized accumulator on torch's current stream, a barrier."""
import os, time, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
x = torch.ones((1024, 1024, 3), dtype=torch.float32, device="cuda")
for _ in range(3):
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    dist.all_reduce(x, op=dist.ReduceOp.SUM)
torch.cuda.synchronize()
print("nccl ok, sum", float(x.sum()), "all_reduce ms", (time.perf_counter() - t0) / 20 * 1e3, flush=True)
dist.destroy_process_group()
