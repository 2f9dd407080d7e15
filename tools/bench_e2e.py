"""BASELINE configs[4] row: streaming 1080p frames -> ESRGAN x4 (NB=23, G=32, both SelfAttention layers, bf16, reference patch mode
p=48 s=24: 3600 patches per frame) -> VGG16 defect vote on 14400 patches 96x96 of the SR frame, device resident
(sr355.pipeline.stream_sr_classify).  Prints one JSON line: frames/s, SR output MPix/s, per-frame stage times.

    python tools/bench_e2e.py [frames=4]
    python -m torch.distributed.run --nproc-per-node N ... tools/bench_e2e.py      # frames sharded over ranks
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "super-resolution-images-for-3d-printing-defect-detection_amd")]
import torch

from sr355 import Context
from sr355 import dist as D
from sr355.bench_rows import cfg4_streaming

n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 4
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
rank, world, local = D.init_from_env()
D.barrier()
row = cfg4_streaming(Context.get(local), n_frames, rank, world)
t = torch.tensor([row.pop("wall_s")], dtype=torch.float64, device="cuda")
D.allreduce_max(t)
if rank == 0:
    wall = float(t.item())
    row.update({"n_gpus": world, "frames_per_s": row["frames"] / wall, "sr_output_mpix_per_s": row["frames"] * 4320 * 7680 / 1e6 / wall,
                "ms_per_frame": 1e3 * wall / row["frames"] * world})
    print(json.dumps(row))
D.shutdown()
