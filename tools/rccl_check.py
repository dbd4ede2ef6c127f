"""Single-rank walk through the RCCL calls bench.py / dist.py make at N > 1 (init with device_id,
barrier, all_reduce MAX on a float64 device tensor, all_gather of device tensors): the 1-GPU box can
check the API usage against the real backend; the N > 1 control flow is covered by the gloo tests."""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
dist.barrier(); torch.cuda.synchronize()
t = torch.tensor([1.25], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
x = torch.arange(12, dtype=torch.float32, device=dev).reshape(3, 4)
outs = [torch.empty_like(x)]
dist.all_gather(outs, x)
assert float(t.item()) == 1.25 and torch.equal(outs[0], x)
dist.barrier(); dist.destroy_process_group()
print("rccl single-rank ok; backend nccl, torch", torch.__version__)
