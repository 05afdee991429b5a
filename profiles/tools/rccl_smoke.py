"""The collectives of traceweaver_amd/sharding.py and bench.py through RCCL (backend "nccl") on the GPU at hand, world size 1:
the box has one GPU, so this only shows that the backend initialises with device_id and carries the dtypes / reductions
the multi-GPU path uses (int64 / int32 all_gather, float64 SUM / MAX all_reduce, uint8 MAX); the N > 1 logic itself is
covered by the 2-rank gloo tests and by `bench.py --gpus N --backend gloo` with N ranks sharing the GPU.

    python profiles/tools/rccl_smoke.py
"""
import json
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29653")
    torch.cuda.set_device(0)
    t0 = time.perf_counter()
    dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    out = {"backend": dist.get_backend(), "init_s": None, "checks": {}}
    x = torch.arange(6, dtype=torch.int64, device="cuda").reshape(2, 3)
    g = [torch.zeros_like(x)]
    dist.all_gather(g, x)
    out["checks"]["all_gather_int64"] = bool(torch.equal(g[0], x))
    y = torch.arange(1000, dtype=torch.int32, device="cuda")
    g = [torch.zeros_like(y)]
    dist.all_gather(g, y)
    out["checks"]["all_gather_int32"] = bool(torch.equal(g[0], y))
    z = torch.tensor([1.0, float("nan"), -0.0, 3e300], dtype=torch.float64, device="cuda")   # gap rows carry NaN for dropped samples
    g = [torch.zeros_like(z)]
    dist.all_gather(g, z)
    out["checks"]["all_gather_f64_bits"] = bool(torch.equal(g[0].view(torch.int64), z.view(torch.int64)))
    f = torch.tensor([1.5, 2.5], dtype=torch.float64, device="cuda")
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    out["checks"]["all_reduce_f64_sum"] = f.tolist() == [1.5, 2.5]
    dist.all_reduce(f, op=dist.ReduceOp.MAX)
    out["checks"]["all_reduce_f64_max"] = f.tolist() == [1.5, 2.5]
    b = torch.tensor([0, 1, 0, 1], dtype=torch.uint8, device="cuda")
    dist.all_reduce(b, op=dist.ReduceOp.MAX)
    out["checks"]["all_reduce_u8_max"] = b.tolist() == [0, 1, 0, 1]
    dist.barrier()
    torch.cuda.synchronize()
    out["init_s"] = time.perf_counter() - t0
    dist.destroy_process_group()
    out["ok"] = all(out["checks"].values())
    print(json.dumps(out))


if __name__ == "__main__":
    main()
