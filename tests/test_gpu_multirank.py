"""The N > 1 code path on the GPU: two fresh processes under torch.distributed.run, each rendering its share of one job
(by sample range, and by image tile) and reducing onto rank 0, against the same job rendered by one process.  A box
with one GPU rehearses it with both ranks on GPU 0 over gloo (V4R_BENCH_REHEARSE=1); with two or more GPUs visible the
ranks take a GPU each and reduce with RCCL."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from v4ray_hip import Renderer, _abi, scenes

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("split,flags,passes", [("sample", 0, 6), ("tile", _abi.V4R_FLAG_PLANE_LIGHT_PDF, 6), ("tile", 0, 256)])
def test_two_ranks_render_the_single_rank_image(split, flags, passes, tmp_path):
    import torch
    scene = scenes.load("c3", 96, 80, flags=flags)
    r = Renderer(scene.renderer_param, scene.camera_param, scene)
    r.render_passes(passes)
    single, n = r.read_accumulator()
    r.close()
    assert n == passes
    out = tmp_path / f"{split}.npy"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    if torch.cuda.device_count() < 2:
        env["V4R_BENCH_REHEARSE"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.join(HERE, "multirank_child.py"), "--split", split, "--out", str(out),
           "--passes", str(passes), "--flags", str(flags)]
    proc = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=240)
    assert proc.returncode == 0, proc.stdout[-3000:]
    both = np.load(out)
    if torch.cuda.device_count() >= 2:
        assert "backend nccl" in proc.stdout, proc.stdout[-2000:]  # two GPUs: the reduce went through RCCL
    if split == "tile" and flags == 0:
        # The default light pdf (packet-statistics rule) counts the rays that end on a shape within the band a rank
        # renders: a shape that straddles the bands gets smaller counts than in a whole pass, so the two-band image is
        # another estimate of the same picture -- same mean, and at 256 passes close per pixel -- not the same bits.
        a, b = both / passes, single / passes
        mse = float(((a - b) ** 2).mean())
        psnr = 10 * np.log10(float(b.max()) ** 2 / mse) if mse > 0 else np.inf
        print(f"default-rule tile split vs single rank, {passes} passes: PSNR {psnr:.1f} dB, means {a.mean():.6f} {b.mean():.6f}")
        assert abs(a.mean() - b.mean()) <= 0.01 * b.mean() and psnr >= 30.0
    elif split == "tile":
        # a pixel draws the streams of its place in the whole image: the bands are exactly the rows of the full render
        # (with a light pdf that is a function of the ray alone; the packet-statistics rule counts rays per band)
        assert np.array_equal(both, single)
    else:
        # pass p is the same wherever it runs: only the order of the final fp32 sum differs
        assert np.allclose(both, single, rtol=1e-5, atol=1e-6)
        assert abs(both.mean() - single.mean()) <= 1e-6 * single.mean()
