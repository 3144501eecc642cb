"""Sample-range split of one render job over the ranks of a torch.distributed group.

Passes are independent (the reference runs them concurrently and sums them, src/main.rs:46-63), so a job of
`total_passes` 1-spp passes is split by pass index: rank r renders passes r, r + N, r + 2N, ...  Pass p draws
the random streams keyed by (seed, p, ...) wherever it runs, so the split changes nothing but the order of
the final floating-point sum.  The only exchange step is one reduce of the (H, W, 3) fp32 accumulators onto
the destination rank -- RCCL over xGMI when the group's backend is "nccl".
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


def render_split(render: Callable[[int, int, int], None], accum, total_passes: int, dst: int = 0, group=None):
    """Render this rank's share into `accum` (a torch tensor the renderer accumulates into) through
    `render(first_pass, pass_stride, n_passes)`, then sum all ranks' accumulators onto rank `dst`.
    Returns the number of passes this rank rendered."""
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    first, stride, n = rank_passes(rank, world, total_passes)
    if n:
        render(first, stride, n)
    if world > 1:
        dist.reduce(accum, dst=dst, op=dist.ReduceOp.SUM, group=group)
    return n
