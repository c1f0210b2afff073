import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
for op in (dist.ReduceOp.MAX, dist.ReduceOp.SUM, dist.ReduceOp.MIN):
    t = torch.tensor([1.5], dtype=torch.float64, device=dev); dist.all_reduce(t, op=op); assert float(t.item()) == 1.5
ft = torch.tensor([7], dtype=torch.int64, device=dev); dist.all_reduce(ft, op=dist.ReduceOp.SUM); assert int(ft.item()) == 7
dist.barrier(); torch.cuda.synchronize()
g = torch.empty((1024,), dtype=torch.uint8, device=dev); s = torch.ones((1024,), dtype=torch.uint8, device=dev)
dist.all_gather_into_tensor(g, s); torch.cuda.synchronize(); assert int(g.sum().item()) == 1024
dist.destroy_process_group(); print("rccl one-rank ops ok")
