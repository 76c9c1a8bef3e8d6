"""Multi-GPU sharding of independent attacks.

Each `(audio, target)` attack of the reference's driver loop is independent (attackMain.py:324-409
is a plain `for` loop sharing only the read-only model and, for OSI/SV, one pre-computed scalar
`threshold_estimated`), so the path shards embarrassingly: one process per GPU, model replicated,
NO collective inside an attack.  Attack cost varies by two orders of magnitude (early stop at
FAKEBOB.py:181-191 against max_iter = 1000, attackMain.sh:24), so the attacks are not dealt out in advance: a free
attack stream DRAWS the next global attack index from a ticket counter (`WorkQueue`) -- a lock-protected counter inside
one process, an atomic add on the job's own key-value store (a TCPStore) across ranks (no collective; nccl and gloo alike).
Results do not depend on who runs what: the Philox stream of an attack is its global index.  `schedule="static"`
keeps the round-robin deal of earlier rounds (A/B, bench.py's end_to_end line).  RCCL (torch.distributed backend
"nccl" on ROCm; "gloo" in the CPU tests) is used only for
  * one broadcast of the estimated threshold (1 x float64) from rank 0 -- mirrors
    attackMain.py:356-357 / :393-394 --, one of the job's Philox key (1 x int64) when --seed is omitted, and
  * one all-reduce(sum) of {success_cnt, total_cnt, nes_iters, scored_utts} (4 x int64) at the end
    -- mirrors attackMain.py:312,335-336,411.
"""
import os
import threading


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def shard_indices(n_items, rank, world):
    """Static round-robin: item i belongs to rank i % world (length-sorted callers get balance)."""
    return list(range(rank, n_items, world))


class WorkQueue(object):
    """Ticket dispenser over the global attack list: next() -> the next index nobody has drawn yet, or None when the
    list is exhausted.  Thread-safe (the K attack streams of a rank draw from the same object).

    world == 1 or schedule == "static": no communication -- static hands rank r the indices r, r + world, ... in
    order (what attack_main did before round 5, per stream as well when `streams` is given: stream k of K takes every
    K-th of them).  Dynamic with world > 1: `store.add(key, 1)` on the job's own store (`job_store()`: the TCPStore this
    module opened for the rendezvous, under its own prefix) is atomic across ranks and returns the new value; every rank
    constructs its STORE-BACKED queues in the same order, so queue number q uses the same key everywhere (or pass
    `name`)."""
    _made = 0   # store-backed queues constructed so far in this process (the default key's suffix)

    def __init__(self, n_items, dist=None, schedule="dynamic", rank=None, world=None, streams=None, name=None):
        if schedule not in ("dynamic", "static"):
            raise ValueError("schedule must be 'dynamic' or 'static'")
        r, _, w = dist_env() if dist is not None else (0, 0, 1)
        self.n = int(n_items)
        self.rank = r if rank is None else rank
        self.world = w if world is None else world
        self.schedule = schedule
        self._lock = threading.Lock()
        self._streams = streams
        self._store = None
        self._key = None
        if schedule == "dynamic" and self.world > 1:
            # Only a queue that really goes through the store takes a key number (round-5 advisor finding: local queues
            # -- bench.py's end_to_end builds seven per call -- advanced it too, so ranks that had built different numbers
            # of local queues disagreed on the key and each drew the whole list).  `name` pins the key explicitly.
            self._store = job_store()
            self._key = "next/%s" % (name if name is not None else WorkQueue._made)
            if name is None:
                WorkQueue._made += 1
        mine = list(range(self.rank, self.n, self.world))
        if schedule == "static" and streams:
            self._static = [mine[k::streams] for k in range(streams)]
        else:
            self._static = [mine]
        self._local_next = 0
        self.drawn = 0    # items this rank has drawn

    def next(self, stream=0):
        with self._lock:
            if self.schedule == "static":
                lst = self._static[stream if self._streams else 0]
                if not lst:
                    return None
                self.drawn += 1
                return lst.pop(0)
            if self._store is not None:
                i = int(self._store.add(self._key, 1)) - 1
            else:
                i = self._local_next
                self._local_next += 1
            if i >= self.n:
                return None
            self.drawn += 1
            return i


_STORE = None   # this job's own key-value store (the ticket counters of WorkQueue live in it)


def _make_store(rank, world, port=None):
    """A TCPStore of this job's own on MASTER_ADDR (rank 0 serves), wrapped in a PrefixStore: public torch.distributed
    API only (round-5 review: the queue used to borrow the process group's store through the private
    `distributed_c10d._get_default_store()`)."""
    import datetime
    import torch.distributed as td
    host = os.environ.get("MASTER_ADDR", "127.0.0.1")
    if port is None:
        port = int(os.environ.get("MASTER_PORT", "29533"))
    # (under torchrun the agent already serves a store on MASTER_PORT and says so: every worker is a client of it)
    agent = port == int(os.environ.get("MASTER_PORT", "-1")) and os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True"
    return td.TCPStore(host, port, world, is_master=(rank == 0 and not agent), timeout=datetime.timedelta(seconds=1800),
                       wait_for_workers=False)


def job_store():
    """The store WorkQueue draws tickets from.  Created by init_process_group(); when somebody else initialised
    torch.distributed (no handle here), a second TCPStore on FAKEBOB_STORE_PORT (default MASTER_PORT + 1)."""
    global _STORE
    if _STORE is None:
        import torch.distributed as td
        rank, _, world = dist_env()
        port = int(os.environ.get("FAKEBOB_STORE_PORT", str(int(os.environ.get("MASTER_PORT", "29533")) + 1)))
        _STORE = td.PrefixStore("fakebob", _make_store(rank, world, port))
    return _STORE


def init_process_group(backend=None):
    """Initialises torch.distributed when WORLD_SIZE > 1.  Returns the module or None.  The rendezvous store is created
    HERE (a TCPStore on MASTER_ADDR:MASTER_PORT) and handed to init_process_group, and the same handle -- under its own
    prefix -- serves WorkQueue's tickets."""
    global _STORE
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
    tcp = _make_store(rank, world)
    dist.init_process_group(backend, store=dist.PrefixStore("pg", tcp), rank=rank, world_size=world, **kw)
    _STORE = dist.PrefixStore("fakebob", tcp)
    return dist


def agree_on_failure(failed, dist=None):
    """True on EVERY rank when any rank reports a failure: one 1 x int64 all-reduce ahead of the result reduction, so a
    rank whose attack stream raised does not leave the others blocked in the final all-reduce (round-5 advisor finding)."""
    if dist is None:
        return bool(failed)
    return reduce_counters([1 if failed else 0], dist)[0] > 0


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


def run_sharded(items, attack_fn, estimate_fn=None, dist=None, schedule="dynamic", streams=1):
    """Runs `attack_fn(item, threshold) -> (success_flag, n_iters, n_scored)` over the items this rank draws
    (`streams` threads per rank; attack_fn must then be thread-safe -- one engine per stream in the drivers).

    estimate_fn() -> threshold runs on rank 0 only and is broadcast (OSI / SV); None for CSI.
    Returns (global_success, global_total, global_iters, global_scored, local_results) where
    local_results = [(item_index, success_flag)] for this rank, in the order they finished."""
    rank, _, world = dist_env() if dist is not None else (0, 0, 1)
    thr = None
    if estimate_fn is not None:
        thr = estimate_fn() if rank == 0 else None
        thr = broadcast_threshold(thr, dist)
    q = WorkQueue(len(items), dist, schedule, streams=streams if streams > 1 else None)
    local = []
    tot = [0, 0, 0]
    lock = threading.Lock()
    errors = []

    def worker(k):
        try:
            while True:
                i = q.next(k)
                if i is None:
                    return
                flag, n_it, n_sc = attack_fn(items[i], thr)
                with lock:
                    local.append((i, flag))
                    tot[0] += 1 if flag == 1 else 0
                    tot[1] += n_it
                    tot[2] += n_sc
        except BaseException as ex:  # noqa: BLE001  (re-raised on the calling thread)
            errors.append(ex)

    if streams <= 1:
        worker(0)
    else:
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(streams)]
        [t.start() for t in ths]
        [t.join() for t in ths]
    if agree_on_failure(bool(errors), dist):
        if errors:
            raise errors[0]
        raise RuntimeError("an attack stream of another rank failed: this rank's results are incomplete")
    g = reduce_counters([tot[0], len(local), tot[1], tot[2]], dist)
    if g[1] != len(items):
        raise RuntimeError("work queue handed out %d of %d attacks" % (g[1], len(items)))
    return g[0], g[1], g[2], g[3], local
