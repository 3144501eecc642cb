"""Split of one render job over the ranks of a torch.distributed group: by sample range (default) or by image tile.

Passes are independent (the reference runs them concurrently and sums them, src/main.rs:46-63), so a job of
`total_passes` 1-spp passes is split by pass index: rank r renders passes r, r + N, r + 2N, ...  Pass p draws
the random streams keyed by (seed, p, ...) wherever it runs, so the split changes nothing but the order of
the final floating-point sum.  The only exchange step is one reduce of the (H, W, 3) fp32 accumulators onto
the destination rank -- RCCL over xGMI when the group's backend is "nccl".

The tile split gives rank r a band of image rows and all the passes: preferable when there are fewer passes
than ranks.  A pixel draws the random streams of its position in the whole image, so the bands are exactly
the rows of the single-rank image; the accumulators are zero outside a rank's band, and the same reduce
assembles the image.
"""
from typing import Callable, Tuple


def rank_passes(rank: int, world: int, total_passes: int) -> Tuple[int, int, int]:
    """(first_pass, pass_stride, n_passes) of `rank`: the passes {p : p mod world == rank, p < total_passes}."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    if total_passes < 0:
        raise ValueError("negative pass count")
    n = (total_passes - rank + world - 1) // world if total_passes > rank else 0
    return rank, world, n


def rank_rows(rank: int, world: int, height: int) -> Tuple[int, int]:
    """(row0, n_rows) of `rank`: bands of near-equal height that tile [0, height)."""
    if not 0 <= rank < world:
        raise ValueError("rank out of range")
    row0, row1 = height * rank // world, height * (rank + 1) // world
    return row0, row1 - row0


def render_split(render: Callable[..., None], accum, total_passes: int, dst: int = 0, group=None, split: str = "sample"):
    """Render this rank's share into `accum` (a torch tensor the renderer accumulates into), then sum all ranks'
    accumulators onto rank `dst`.  split="sample": `render(first_pass, pass_stride, n_passes)` with this rank's
    passes; split="tile": `render(0, 1, total_passes, rows=(row0, n_rows))` with this rank's band of rows
    (accum is (H, W, 3)).  Returns the number of passes this rank rendered."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    if split == "tile":
        row0, n_rows = rank_rows(rank, world, int(accum.shape[0]))
        n = total_passes if n_rows else 0
        if n:
            render(0, 1, n, rows=(row0, n_rows))
    elif split == "sample":
        first, stride, n = rank_passes(rank, world, total_passes)
        if n:
            render(first, stride, n)
    else:
        raise ValueError("split must be 'sample' or 'tile'")
    if world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return n
