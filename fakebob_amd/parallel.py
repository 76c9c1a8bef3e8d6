"""Multi-GPU sharding of independent attacks.

Each `(audio, target)` attack of the reference's driver loop is independent (attackMain.py:324-409
is a plain `for` loop sharing only the read-only model and, for OSI/SV, one pre-computed scalar
`threshold_estimated`), so the path shards embarrassingly: one process per GPU, model replicated,
utterances dealt round-robin, NO collective inside an attack.  RCCL (torch.distributed backend
"nccl" on ROCm; "gloo" in the CPU tests) is used only for
  * one broadcast of the estimated threshold (1 x float64) from rank 0 -- mirrors
    attackMain.py:356-357 / :393-394 --, one of the job's Philox key (1 x int64) when --seed is omitted, and
  * one all-reduce(sum) of {success_cnt, total_cnt, nes_iters, scored_utts} (4 x int64) at the end
    -- mirrors attackMain.py:312,335-336,411.
"""
import os


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n_items, rank, world):
    """Static round-robin: item i belongs to rank i % world (length-sorted callers get balance)."""
    return list(range(rank, n_items, world))


def init_process_group(backend=None):
    """Initialises torch.distributed when WORLD_SIZE > 1.  Returns the module or None."""
    rank, local_rank, world = dist_env()
    if world <= 1:
        return None
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        return dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        kw["device_id"] = torch.device("cuda", local_rank)
    dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return dist


def _device(dist):
    import torch
    if dist is not None and dist.get_backend() == "nccl":
        return torch.device("cuda", dist_env()[1])
    return torch.device("cpu")


def broadcast_threshold(value, dist=None, src=0):
    """Rank `src` estimated the threshold (FakeBob.estimate_threshold); everyone gets it."""
    if dist is None:
        return value
    import torch
    t = torch.tensor([float(value) if value is not None else 0.0], dtype=torch.float64, device=_device(dist))
    dist.broadcast(t, src=src)
    return float(t.item())


def broadcast_int(value, dist=None, src=0):
    """An integer (the job's Philox key) from rank `src` to everyone, as int64 -- not through a float."""
    if dist is None:
        return int(value)
    import torch
    t = torch.tensor([int(value)], dtype=torch.int64, device=_device(dist))
    dist.broadcast(t, src=src)
    return int(t.item())


def reduce_counters(counters, dist=None):
    """Sum a list of integer counters over all ranks (success_cnt, total_cnt, ...)."""
    if dist is None:
        return [int(c) for c in counters]
    import torch
    t = torch.tensor([int(c) for c in counters], dtype=torch.int64, device=_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [int(x) for x in t.tolist()]


def run_sharded(items, attack_fn, estimate_fn=None, dist=None):
    """Runs `attack_fn(item, threshold) -> (success_flag, n_iters, n_scored)` over this rank's shard.

    estimate_fn() -> threshold runs on rank 0 only and is broadcast (OSI / SV); None for CSI.
    Returns (global_success, global_total, global_iters, global_scored, local_results) where
    local_results = [(item_index, success_flag)] for this rank."""
    rank, _, world = dist_env() if dist is not None else (0, 0, 1)
    thr = None
    if estimate_fn is not None:
        thr = estimate_fn() if rank == 0 else None
        thr = broadcast_threshold(thr, dist)
    local = []
    succ = iters = scored = 0
    mine = shard_indices(len(items), rank, world)
    for i in mine:
        flag, n_it, n_sc = attack_fn(items[i], thr)
        local.append((i, flag))
        succ += 1 if flag == 1 else 0
        iters += n_it
        scored += n_sc
    g = reduce_counters([succ, len(mine), iters, scored], dist)
    return g[0], g[1], g[2], g[3], local
