"""Multi-GPU multi-try: one process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI on
MI355X, "gloo" in the CPU tests).  Multi-try chains are independent (the reference runs them back
to back, src/UserFunctions.cpp:302-325), so chain c goes to rank c % world_size with the full data
replicated, and the only communication is the final selection:
  1. all-gather of one (score, chain index) pair per rank,
  2. broadcast of the winning chain's arrays from the rank that owns it.
Ties go to the lowest chain index, which is what the reference's sequential `<` comparison keeps.
"""
import numpy as np


def select_winner(scores, chains):
    """scores / chains: per-rank best (score, chain index); ranks with no chain report -inf.
    Returns the index of the winning rank."""
    best = None
    for r, (s, c) in enumerate(zip(scores, chains)):
        if not np.isfinite(s) and s < 0:
            continue
        if best is None or s > scores[best] or (s == scores[best] and c < chains[best]):
            best = r
    if best is None:
        raise RuntimeError("no rank produced a chain")
    return best


def gather_select_broadcast(local, group):
    """`local` is this rank's best-chain result dict (or None when the rank had no chain to run).
    Returns the winning dict on every rank."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")
    mine = torch.tensor([local["best_score"] if local else -np.inf, local["best_chain"] if local else -1.0],
                        dtype=torch.float64, device=dev)
    allv = [torch.zeros(2, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(allv, mine, group=group)
    scores = [float(v[0]) for v in allv]
    chains = [int(v[1]) for v in allv]
    src = select_winner(scores, chains)
    # the winner publishes the names / shapes, then every array
    meta = [None]
    if rank == src:
        meta = [[(k, v.shape if hasattr(v, "shape") else None) for k, v in local.items() if k not in ("B", "B_obs")]]
    dist.broadcast_object_list(meta, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
    out = {}
    for key, shape in meta[0]:
        if shape is None:
            val = [local[key] if rank == src else 0.0]
            t = torch.tensor(val, dtype=torch.float64, device=dev)
            dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
            out[key] = float(t[0])
        else:
            if rank == src:
                t = torch.from_numpy(np.ascontiguousarray(local[key].reshape(-1, order="F"))).to(dev)
            else:
                t = torch.empty(int(np.prod(shape)), dtype=torch.float64, device=dev)
            dist.broadcast(t, src=dist.get_global_rank(group, src) if group is not None else src, group=group)
            out[key] = t.cpu().numpy().reshape(shape, order="F")
    if local is not None:
        for k in ("B", "B_obs"):
            if k in local:
                out[k] = local[k]
    return out


def multi_try(call, args, group):
    """Deals the chain indices 0..n_try round-robin over the ranks of `group`, runs this rank's share with `call`
    and returns the overall best chain on every rank."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    args.a.chain_offset, args.a.chain_stride = rank, world
    local = call() if rank <= args.a.n_try else None
    return gather_select_broadcast(local, group)
