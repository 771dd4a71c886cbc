"""Multi-GPU sharding of the hot path (one process per GPU, torch.distributed; RCCL on the GPU box,
gloo in the CPU tests).

The reference has no distributed code at all.  The path shards naturally:
  * within one AIME round every branch node's scene forward + prune/merge is independent
    (planners/mind/networks/network.py:318,497 loop per scene), so a round's scenes are block-
    distributed over the ranks; the ONE real exchange step per round is an all-gather of the kept
    children -- a [P, 25] float32 header (scene index in the round, mode, path probability, the
    node's 11-point target window) and the [a, 60, 6] world-frame rows of the surviving modes, as
    packed tensors that never leave the device before the collective (RCCL ``all_gather_into_tensor``
    over xGMI; gloo moves the same tensors in the CPU tests).  Afterwards every rank holds the
    identical tree and takes the (cheap, deterministic) branching decisions itself; the re-basing of
    the next round's scenes (mind_aime_rebase) runs only on the rank that owns the scene;
  * contingency solves are independent per scenario tree (planners/mind/planner.py:120-123): trees
    are dealt round-robin, their [M, 8] float64 (xs | us) rows all-gathered.
Predictor results are bit-identical for any batch composition (the kernel's column-split rule depends
on the scene's own size only), so 1/2/4/8-rank runs build the same node sets.
"""
import numpy as np
import torch


class Shard:
    """Contiguous block / round-robin partitions + packed-tensor all-gather over a process group."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.active else 0
        self.world = dist.get_world_size(group) if self.active else 1
        self.backend = dist.get_backend(group) if self.active else None
        # RCCL moves device tensors; gloo (tests) moves host tensors
        self.device = torch.device("cuda", torch.cuda.current_device()) if self.backend == "nccl" else torch.device("cpu")
        self.n_collectives = 0
        self.bytes_gathered = 0
        # a one-rank group normally short-cuts every collective; MIND_FORCE_COLLECTIVES=1 (tests: a world-size-1 nccl group on the
        # one-GPU box) runs them for real, so the RCCL code path is executed even where a second device is missing
        import os
        self.force = os.environ.get("MIND_FORCE_COLLECTIVES", "0") == "1"
        self.native = False       # attach(): the runtime's mind_aime_plan is sharded through this group

    @property
    def sharded(self):
        """do the collectives run?  (more than one rank, or forced on a one-rank group)"""
        return self.active and (self.world > 1 or self.force)

    def block(self, n):
        """[lo, hi) of this rank among n items (first ranks get the remainder)."""
        base, rem = divmod(n, self.world)
        lo = self.rank * base + min(self.rank, rem)
        return lo, lo + base + (1 if self.rank < rem else 0)

    def round_robin(self, n):
        return list(range(self.rank, n, self.world))

    def all_gather_rows(self, *tensors):
        """Every tensor is [n_i, ...] on this rank with rank-dependent n_i.  Returns, per tensor, the rows of all
        ranks concatenated in rank order (on ``self.device``) -- one small collective for all the counts, one
        ``all_gather_into_tensor`` per tensor on buffers padded to the largest rank."""
        if not self.sharded:
            return [t for t in tensors]
        dist, W, dev = self.dist, self.world, self.device
        cnt = torch.tensor([int(t.shape[0]) for t in tensors], dtype=torch.int64, device=dev)
        cnts = torch.empty(W * len(tensors), dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(cnts, cnt, group=self.group)
        cnts = cnts.view(W, len(tensors)).tolist()
        self.n_collectives += 1
        out = []
        for i, t in enumerate(tensors):
            mx = max(c[i] for c in cnts)
            tail = tuple(t.shape[1:])
            if mx == 0:
                out.append(torch.empty((0,) + tail, dtype=t.dtype, device=dev))
                continue
            buf = torch.zeros((mx,) + tail, dtype=t.dtype, device=dev)
            if t.shape[0]:
                buf[:t.shape[0]] = t.to(dev)
            full = torch.empty((W * mx,) + tail, dtype=t.dtype, device=dev)
            dist.all_gather_into_tensor(full, buf, group=self.group)
            self.n_collectives += 1
            self.bytes_gathered += full.numel() * full.element_size()
            if all(c[i] == mx for c in cnts):
                out.append(full)
            else:
                out.append(torch.cat([full[r * mx:r * mx + cnts[r][i]] for r in range(W)]))
        return out

    # ---- transport of the sharded native AIME plan (mind_set_exchange, include/mind_hip.h) ---------------------------------
    def attach(self, rt):
        """Hand this group's collectives to the runtime's context: mind_aime_plan then block-distributes every round's scenes over the
        ranks and calls back here for its (two per round + one per plan) fixed-size exchanges on device buffers -- RCCL moves them where
        they are (`nccl`), gloo (tests) through host copies."""
        from . import _lib
        import ctypes as C

        def dev_bytes(ptr, n):
            """torch uint8 view of n bytes of device memory at ptr (no copy)"""
            class _P:
                __cuda_array_interface__ = {"shape": (int(n),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}
            return torch.as_tensor(_P(), device=torch.device("cuda", torch.cuda.current_device()))

        def cb(user, op, send, recv, nbytes):
            try:
                dist, W = self.dist, self.world
                if op == _lib.XCHG_ALLGATHER:
                    s, r = dev_bytes(send, nbytes), dev_bytes(recv, nbytes * W)
                    if self.backend == "nccl":
                        dist.all_gather_into_tensor(r, s, group=self.group)
                        torch.cuda.current_stream().synchronize()
                    else:
                        out = torch.empty(nbytes * W, dtype=torch.uint8)
                        dist.all_gather_into_tensor(out, s.cpu(), group=self.group)
                        r.copy_(out)
                        torch.cuda.current_stream().synchronize()
                    self.bytes_gathered += nbytes * W
                elif op == _lib.XCHG_ALLTOALLV:
                    # `nbytes` is the address of the library's table int64 [2][world]: bytes to / from every rank (include/mind_hip.h)
                    tab = (C.c_int64 * (2 * W)).from_address(nbytes)
                    sb, rb = [int(tab[r]) for r in range(W)], [int(tab[W + r]) for r in range(W)]
                    s, r = dev_bytes(send, max(sum(sb), 1))[:sum(sb)], dev_bytes(recv, max(sum(rb), 1))[:sum(rb)]
                    if self.backend == "nccl":
                        dist.all_to_all_single(r, s, output_split_sizes=rb, input_split_sizes=sb, group=self.group)
                        torch.cuda.current_stream().synchronize()
                    else:
                        out = torch.zeros(sum(rb), dtype=torch.uint8)
                        self._all_to_all_cpu(out, s.cpu(), rb, sb)
                        r.copy_(out)
                        torch.cuda.current_stream().synchronize()
                    self.bytes_gathered += sum(rb)
                else:
                    # the plan completes buffers whose entries every rank but the owner left zero: summed as INTEGERS the owner's bits arrive
                    # unchanged whatever the transport's arithmetic (a float sum turns an owner's -0.0 into +0.0)
                    t = dev_bytes(recv, nbytes).view(torch.int32)
                    if send != recv:
                        t.copy_(dev_bytes(send, nbytes).view(torch.int32))
                    if self.backend == "nccl":
                        dist.all_reduce(t, group=self.group)
                        torch.cuda.current_stream().synchronize()
                    else:
                        h = t.cpu()
                        dist.all_reduce(h, group=self.group)
                        t.copy_(h)
                        torch.cuda.current_stream().synchronize()
                    self.bytes_gathered += nbytes
                self.n_collectives += 1
                return 0
            except Exception as e:      # (an exception must not cross the C boundary)
                import traceback
                traceback.print_exc()
                return 1

        self._cb = _lib.EXCHANGE_FN(cb)         # kept alive with the shard
        self._rt = rt
        on = self.active and (self.world > 1 or self.force)
        rc = rt.lib.mind_set_exchange(rt.ctx, self.rank if on else 0, self.world if on else 1, self._cb if on else _lib.EXCHANGE_FN(0), None, 1 if self.force else 0)
        _lib.check(rt.lib, rt.ctx, rc, "mind_set_exchange")
        self.native = on
        rt._exchange_owner = self if on else None       # (the exchange belongs to the context: planners sharing it re-attach / detach, see detach)
        return self

    def _all_to_all_cpu(self, out, inp, rb, sb):
        """all_to_all_single on CPU tensors for the gloo tests: gloo's own all-to-all where this build has it, else one broadcast per
        (sender, receiver) pair's sender -- the same bytes end up in the same places"""
        dist = self.dist
        try:
            dist.all_to_all_single(out, inp, output_split_sizes=rb, input_split_sizes=sb, group=self.group)
            return
        except (RuntimeError, NotImplementedError):
            pass
        W, me = self.world, self.rank
        so, ro = [0], [0]
        for r in range(W):
            so.append(so[-1] + sb[r]); ro.append(ro[-1] + rb[r])
        # every rank tells the others how much it sends to whom (the receivers' sizes are known, the bystanders' are not)
        sizes = [None] * W
        dist.all_gather_object(sizes, sb, group=self.group)
        # a rank's segment for itself never travels (a forced one-rank group -- mind_set_exchange force != 0 -- puts real bytes there)
        if sb[me]:
            out[ro[me]:ro[me + 1]] = inp[so[me]:so[me + 1]]
        for src in range(W):
            for dst in range(W):
                n = sizes[src][dst]
                if n == 0 or src == dst:
                    continue
                buf = inp[so[dst]:so[dst + 1]].clone() if me == src else torch.empty(n, dtype=torch.uint8)
                dist.broadcast(buf, src=dist.get_global_rank(self.group, src) if self.group is not None else src, group=self.group)
                if me == dst:
                    out[ro[src]:ro[src + 1]] = buf

    @staticmethod
    def detach(rt):
        """Take the exchange off a context: the next mind_aime_plan on it plans alone again.  (A context outlives the planner that
        attached a group to it -- the per-thread runtime is shared by every planner of the thread -- so a planner without a shard makes
        sure nobody else's exchange is still on the context it plans with: ScenarioTreeGenerator._sync_exchange.)"""
        from . import _lib
        rc = rt.lib.mind_set_exchange(rt.ctx, 0, 1, _lib.EXCHANGE_FN(0), None, 0)
        _lib.check(rt.lib, rt.ctx, rc, "mind_set_exchange")
        rt._exchange_owner = None

    def broadcast(self, t, src=0):
        """In-place broadcast of a tensor on ``self.device`` from ``src``."""
        if self.sharded:
            self.dist.broadcast(t, src, group=self.group)
            self.n_collectives += 1
        return t


def gather_round_robin(shard, sizes, local_rows):
    """Inverse of ``Shard.round_robin`` for variable-size results: item i (dealt to rank i % world) is a
    [sizes[i], C] array; ``local_rows`` is this rank's items concatenated in its own order.  Returns the list of
    all n items (numpy) on every rank -- one packed all-gather."""
    n = len(sizes)
    t = torch.from_numpy(np.ascontiguousarray(local_rows))
    if not shard.sharded:
        flat = t
        order = list(range(n))
    else:
        flat = shard.all_gather_rows(t)[0].cpu()
        order = [i for r in range(shard.world) for i in range(r, n, shard.world)]
    flat = flat.numpy()
    out, o = [None] * n, 0
    for i in order:
        out[i] = flat[o:o + sizes[i]]
        o += sizes[i]
    return out
