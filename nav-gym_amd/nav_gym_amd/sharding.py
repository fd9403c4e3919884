"""Multi-GPU layout: arenas shard embarrassingly.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI on ROCm).  Rank r owns the
contiguous block of global arena indices shard_range(E_total, r, world); every arena's maps, spawn
tables and RNG streams are keyed by its GLOBAL index (navsim_config.env_index_base), so a sharded
run reproduces the single-GPU run bit for bit.  step() has no exchange; the only collective is the
optional all-gather of observations / rewards / dones for a centralised learner.
"""


def shard_range(n_total, rank, world_size):
    """Contiguous block [start, start + count) of global arena indices owned by `rank`."""
    base, rem = divmod(int(n_total), int(world_size))
    start = rank * base + min(rank, rem)
    return start, base + (1 if rank < rem else 0)


def gather_rows(local, group=None):
    """all_gather of per-arena rows (obs [E_local, D], reward [E_local], ...) in global arena order.
    Equal shard sizes use one all_gather_into_tensor (a single RCCL call); ragged shards fall back to
    all_gather of padded rows."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    n = torch.tensor([local.shape[0]], device=local.device, dtype=torch.int64)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n, group=group)
    counts = [int(c.item()) for c in counts]
    if len(set(counts)) == 1:
        out = torch.empty((world * counts[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    m = max(counts)
    pad = torch.zeros((m,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)


class RowGather(object):
    """The optional obs gather as a resident operation: shard sizes follow from (n_total, world_size) alone, so
    the output (and, for ragged shards, the padded per-rank parts) is allocated once and every call is ONE
    collective with no host synchronisation -- all_gather_into_tensor (RCCL: a single ncclAllGather) when the shards
    are equal, all_gather of rows padded to the largest shard otherwise.  `out` holds the rows in global arena
    order after run()."""

    def __init__(self, n_total, row_shape, dtype, device, rank, world_size, group=None):
        import torch
        self.group = group
        self.counts = [shard_range(n_total, r, world_size)[1] for r in range(world_size)]
        self.start = shard_range(n_total, rank, world_size)[0]
        self.count = self.counts[rank]
        self.equal = len(set(self.counts)) == 1
        row_shape = tuple(row_shape)
        self.out = torch.empty((int(n_total),) + row_shape, dtype=dtype, device=device)
        if not self.equal:
            m = max(self.counts)
            self.pad = torch.zeros((m,) + row_shape, dtype=dtype, device=device)
            self.parts = [torch.empty_like(self.pad) for _ in range(world_size)]

    def run(self, local):
        import torch.distributed as dist
        if local.shape[0] != self.count:
            raise ValueError("this rank owns %d rows, got %d" % (self.count, local.shape[0]))
        if self.equal:
            dist.all_gather_into_tensor(self.out, local.contiguous(), group=self.group)
            return self.out
        self.pad[: self.count].copy_(local)
        dist.all_gather(self.parts, self.pad, group=self.group)
        at = 0
        for part, c in zip(self.parts, self.counts):
            self.out[at:at + c].copy_(part[:c])
            at += c
        return self.out


def make_sharded_env(n_total, rank=None, world_size=None, **kwargs):
    """NavGymEnv over this rank's block of `n_total` global arenas on cuda:LOCAL_RANK."""
    import os
    from . import DEFAULT_KWARGS, NavGymEnv
    rank = int(os.environ.get("RANK", "0")) if rank is None else rank
    world_size = int(os.environ.get("WORLD_SIZE", "1")) if world_size is None else world_size
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    start, count = shard_range(n_total, rank, world_size)
    kw = dict(DEFAULT_KWARGS)
    kw.update(kwargs)
    kw.setdefault("device", "cuda:%d" % local_rank)
    return NavGymEnv(num_envs=count, env_index_base=start, **kw)
