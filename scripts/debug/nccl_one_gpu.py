import os, torch, torch.distributed as dist
rank = int(os.environ["RANK"])
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.full((4,), float(rank), device="cuda:0")
out = [torch.zeros(4, device="cuda:0") for _ in range(2)] if rank == 0 else None
try:
    dist.gather(x, out, dst=0)
    torch.cuda.synchronize()
    print("rank", rank, "gather ok", [o.tolist() for o in out] if out else None)
except Exception as e:
    print("rank", rank, "gather failed:", repr(e)[:300])
dist.destroy_process_group()
