"""One rank of a two-process render job on the GPU (started by tests/test_gpu_multirank.py under torch.distributed.run):
renders this rank's share of the job through v4ray_hip.distributed.render_split into a torch tensor bound as the
renderer's accumulator, reduces onto rank 0 and lets rank 0 save the image.  With V4R_BENCH_REHEARSE=1 every rank uses
GPU 0 and the gloo backend (RCCL refuses two ranks on one device), which is how a one-GPU box rehearses the N > 1 path."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--split", required=True, choices=["sample", "tile"])
    ap.add_argument("--out", required=True)
    ap.add_argument("--passes", type=int, default=6)
    ap.add_argument("--flags", type=int, default=0)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from v4ray_hip import Renderer, scenes
    from v4ray_hip.distributed import render_split
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rehearse = os.environ.get("V4R_BENCH_REHEARSE") == "1"
    device = 0 if rehearse else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(device)
    if rehearse:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    else:
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", device))
    scene = scenes.load("c3", 96, 80, flags=args.flags)
    r = Renderer(scene.renderer_param, scene.camera_param, scene, device=device)
    accum = torch.zeros(80, 96, 3, dtype=torch.float32, device=f"cuda:{device}")
    r.bind_accumulator(accum.data_ptr())
    stream = torch.cuda.ExternalStream(r.stream_ptr(), device=torch.device("cuda", device))

    def render(first, stride, n, rows=None):
        r.render_passes(n, first_pass=first, pass_stride=stride, rows=rows)
        if rehearse:
            r.synchronize()  # gloo stages through the host and knows nothing about streams

    with torch.cuda.stream(stream):
        n = render_split(render, accum, args.passes, dst=0, split=args.split)
    r.synchronize()
    torch.cuda.synchronize()
    if rank == 0:
        np.save(args.out, accum.cpu().numpy())
    print(f"rank {rank}: rendered {n} passes, backend {dist.get_backend()}", flush=True)
    r.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
