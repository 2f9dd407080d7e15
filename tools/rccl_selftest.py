"""One-rank RCCL self-test (run under torchrun --nproc-per-node 1 on a GPU box): the builder's boxes have one GPU, so the N > 1 paths of
bench.py / tools/bench_train.py have only ever run over gloo.  This at least loads RCCL, creates the communicator the way
sr355.dist.init_from_env does (backend "nccl", device_id) and runs the collectives of those paths with their dtypes and sizes:
float64 [3] SUM (metric sums), float64 [1] MAX (elapsed time), float32 [17.6 M] SUM (the generator + discriminator gradient bucket), barrier."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import torch
import torch.distributed as dist

local = int(os.environ.get("LOCAL_RANK", "0"))
torch.cuda.set_device(local)
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29511")
dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
dev = torch.device("cuda", local)
sums = torch.tensor([31.5, 0.25, 2.0], dtype=torch.float64, device=dev)
dist.all_reduce(sums, op=dist.ReduceOp.SUM)
t = torch.tensor([1.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
bucket = torch.full((16930019 + 658305,), 0.5, dtype=torch.float32, device=dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
dist.all_reduce(bucket, op=dist.ReduceOp.SUM)
bucket /= world
torch.cuda.synchronize()
ms = 1e3 * (time.perf_counter() - t0)
dist.barrier()
ok = sums.tolist() == [31.5 * world, 0.25 * world, 2.0 * world] and float(t) == 1.25 and float(bucket[0]) == 0.5 and float(bucket[-1]) == 0.5
from sr355 import dist as D                      # the helpers themselves (they act for world > 1, and must be harmless no-ops at world 1)
D.allreduce_metric_sums(sums)
D.allreduce_mean_flat(bucket)
D.barrier()
print(f"rccl selftest {'ok' if ok else 'FAILED'}: backend {dist.get_backend()}, world {dist.get_world_size()}, 70 MB bucket all-reduce {ms:.2f} ms", flush=True)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
