"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed; RCCL on the GPU box,
gloo in the CPU tests).

The reference has no distributed code at all.  The path shards naturally:
  * within one AIME round every branch node's scene forward + prune/merge is independent
    (planners/mind/networks/network.py:318,497 loop per scene), so a round's scenes are block-
    distributed over the ranks; the ONE real exchange step per round is an all-gather of the kept
    children (small: ids + [a,100,7] histories), after which every rank holds the identical tree and
    takes the branching decisions redundantly (deterministic host code);
  * contingency solves are independent per scenario tree (planners/mind/planner.py:120-123): trees
    are dealt round-robin, results all-gathered.
Predictor results are bit-identical for any batch composition (the kernel's column-split rule depends
on the scene's own size only), so 1/2/4/8-rank runs build the same node sets.
"""
import numpy as np


class Shard:
    """Contiguous block sharding + object all-gather over a torch.distributed process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.active else 0
        self.world = dist.get_world_size(group) if self.active else 1

    def block(self, n):
        """[lo, hi) of this rank among n items (first ranks get the remainder)."""
        base, rem = divmod(n, self.world)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)

    def round_robin(self, n):
        return list(range(self.rank, n, self.world))

    def all_gather(self, obj):
        """list over ranks of `obj` (pickled; one collective)."""
        if not self.active or self.world == 1:
            return [obj]
        out = [None] * self.world
        self.dist.all_gather_object(out, obj, group=self.group)
        return out


def gather_blocks(shard, local_items):
    """Concatenate per-rank lists in rank order (= original order for block sharding)."""
    out = []
    for part in shard.all_gather(local_items):
        out.extend(part)
    return out


def gather_round_robin(shard, n, local_results):
    """Inverse of Shard.round_robin: local_results[i] belongs to item shard.rank + i * world."""
    parts = shard.all_gather(local_results)
    out = [None] * n
    for r, part in enumerate(parts):
        for i, v in enumerate(part):
            out[r + i * len(parts)] = v
    return out
