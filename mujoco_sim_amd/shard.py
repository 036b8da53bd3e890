"""Env sharding across the GPUs of one node: contiguous env ranges per rank, no
data-path collective; one all-gather of the published state slice feeds the single ROS state topic
(publishers at mj_ros.cpp:554-564).  Works on any torch.distributed backend (RCCL on the GPU box,
gloo in the CPU tests)."""


def env_range(total, world, rank):
    """[lo, hi) of the envs owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def gather_state(local, total, world, rank):
    """all-gather the [n_local, stride] state slices into env order -> [total, stride]"""
    import torch
    import torch.distributed as dist

    sizes = [env_range(total, world, r)[1] - env_range(total, world, r)[0] for r in range(world)]
    stride = local.shape[1]
    if len(set(sizes)) == 1:
        out = torch.empty(total * stride, dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.reshape(-1).contiguous())
        return out.reshape(total, stride)
    # uneven shares (they differ by at most one env): every rank pads its slice to the largest share, one all-gather of equal
    # slots, then the padding rows are dropped — the same scheme as mjh_group_publish (csrc/group.hip)
    slot = max(sizes)
    padded = torch.zeros(slot, stride, dtype=local.dtype, device=local.device)
    padded[:local.shape[0]] = local
    out = torch.empty(world * slot * stride, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, padded.reshape(-1).contiguous())
    out = out.reshape(world, slot, stride)
    return torch.cat([out[r, :sizes[r]] for r in range(world)], 0)


def max_over_ranks(x):
    import torch
    import torch.distributed as dist

    t = torch.tensor([x], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
