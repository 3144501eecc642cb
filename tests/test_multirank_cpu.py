"""The N > 1 path on CPU: two gloo ranks split a job by pass index and reduce their accumulators.
The renderer behind `render_split` is injected; here it is the CPU oracle (no GPU in this container)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle
from v4ray_hip import scenes
from v4ray_hip.distributed import rank_passes, rank_rows, render_split

TOTAL = 7   # odd on purpose: the ranks get 4 and 3 passes


def test_rank_passes_partition():
    for world in (1, 2, 3, 4, 8):
        for total in (0, 1, 5, 8, 17, 4096):
            seen = []
            for rank in range(world):
                first, stride, n = rank_passes(rank, world, total)
                seen += [first + k * stride for k in range(n)]
            assert sorted(seen) == list(range(total))
    with pytest.raises(ValueError):
        rank_passes(2, 2, 4)


def test_rank_rows_tile_the_image():
    for world in (1, 2, 3, 8):
        for height in (1, 7, 48, 1080):
            rows = [rank_rows(r, world, height) for r in range(world)]
            assert rows[0][0] == 0 and sum(n for _, n in rows) == height
            assert all(rows[k][0] + rows[k][1] == rows[k + 1][0] for k in range(world - 1))
            assert max(n for _, n in rows) - min(n for _, n in rows) <= 1


def _worker(rank, world, port, out_path, split="sample"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    s = scenes.load("c3", 48, 48)
    d = s.flatten(s.renderer_param, s.camera_param)
    accum = torch.zeros(48, 48, 3, dtype=torch.float32)

    def render(first, stride, n, rows=None):
        if rows is None:
            oracle.render(d, n, seed=0x5EED, first_pass=first, pass_stride=stride, threads=2, accum=accum.numpy())
        else:  # the oracle renders whole images: keep this rank's band of rows
            img, _ = oracle.render(d, n, seed=0x5EED, first_pass=first, pass_stride=stride, threads=2)
            accum.numpy()[rows[0]:rows[0] + rows[1]] += img[rows[0]:rows[0] + rows[1]]

    n = render_split(render, accum, TOTAL, dst=0, split=split)
    counts = torch.tensor([n])
    dist.all_reduce(counts)
    if rank == 0:
        np.savez(out_path, image=accum.numpy(), passes=int(counts.item()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reduce_to_the_single_rank_image(tmp_path):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = str(tmp_path / "reduced.npz")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["passes"]) == TOTAL
    s = scenes.load("c3", 48, 48)
    d = s.flatten(s.renderer_param, s.camera_param)
    whole, _ = oracle.render(d, TOTAL, seed=0x5EED)
    assert np.allclose(got["image"], whole, rtol=1e-5, atol=1e-6)
    assert not np.array_equal(got["image"], np.zeros_like(whole))


def test_two_ranks_tile_split_assembles_the_single_rank_image(tmp_path):
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    out = str(tmp_path / "tiles.npz")
    mp.spawn(_worker, args=(2, port, out, "tile"), nprocs=2, join=True)
    got = np.load(out)
    assert int(got["passes"]) == 2 * TOTAL          # every rank renders all the passes, of its own band
    s = scenes.load("c3", 48, 48)
    d = s.flatten(s.renderer_param, s.camera_param)
    whole, _ = oracle.render(d, TOTAL, seed=0x5EED)
    assert np.array_equal(got["image"], whole)      # disjoint bands: no reassociation at all
