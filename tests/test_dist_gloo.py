"""N>1 path on CPU: world_size-2 gloo run of the sharding + broadcast + counter reduction that the
multi-GPU driver uses (the only collectives on the path)."""
import os
import subprocess
import sys
import textwrap

from fakebob_amd.parallel import WorkQueue, reduce_counters, run_sharded, shard_indices

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_indices_partition():
    for n in (0, 1, 7, 64, 256):
        for world in (1, 2, 4, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            flat = sorted(i for p in parts for i in p)
            assert flat == list(range(n))
            assert max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_single_process_path_needs_no_collective():
    items = list(range(5))
    out = run_sharded(items, lambda it, thr: (1 if it % 2 == 0 else -1, 10 + it, 51 * (10 + it)),
                      estimate_fn=lambda: 0.25)
    assert out[:4] == (3, 5, 60, 51 * 60)
    assert [i for i, _ in out[4]] == items
    assert reduce_counters([1, 2]) == [1, 2]


WORKER = textwrap.dedent('''
    import sys, json
    sys.path.insert(0, %r)
    from fakebob_amd import parallel as P
    dist = P.init_process_group("gloo")
    rank, _, world = P.dist_env()
    calls = []
    def estimate():
        calls.append("est")
        return 0.2277
    def attack(item, thr):
        assert abs(thr - 0.2277) < 1e-15            # every rank received rank 0's estimate
        return (1 if item %% 3 else -1), item + 1, 51 * (item + 1)
    key = P.broadcast_int((1 << 62) + 12345 if rank == 0 else 0, dist)   # beyond float64's 53 bits: an int64 path
    assert key == (1 << 62) + 12345
    g = P.run_sharded(list(range(11)), attack, estimate, dist)
    assert (len(calls) == 1) == (rank == 0)         # only rank 0 estimates
    with open(sys.argv[1] + "/rank%%d.json" %% rank, "w") as w:     # per-rank file: stdout of two ranks interleaves
        json.dump({"rank": rank, "g": g[:4], "local": g[4]}, w)
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29617", str(script), str(tmp_path)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:]
    import json
    rows = [json.load(open(str(tmp_path / ("rank%d.json" % k)))) for k in range(2)]
    items = list(range(11))
    want = [sum(1 for i in items if i % 3), 11, sum(i + 1 for i in items), 51 * sum(i + 1 for i in items)]
    for row in rows:
        assert row["g"] == want                    # identical global counters on both ranks
    locals_ = sorted(i for row in rows for i, _ in row["local"])
    assert locals_ == items                        # every utterance attacked exactly once


def test_work_queue_single_process_dynamic_and_static():
    for n in (0, 1, 7, 24):
        q = WorkQueue(n)                                   # dynamic, one process: a plain counter
        assert [q.next() for _ in range(n + 2)] == list(range(n)) + [None, None]
        for world in (1, 2, 4):
            for streams in (None, 3):
                got = []
                for r in range(world):
                    q = WorkQueue(n, None, "static", rank=r, world=world, streams=streams)
                    for k in range(streams or 1):
                        while True:
                            i = q.next(k)
                            if i is None:
                                break
                            got.append(i)
                            if streams:                    # stream k of rank r: every K-th of the rank's round-robin share
                                assert i % world == r and (i // world) % streams == k
                assert sorted(got) == list(range(n))
    # K threads on one queue: every item exactly once, and a stream that draws cheap items takes more of them
    import threading, time
    q = WorkQueue(60)
    took = {0: [], 1: [], 2: []}
    def run(k):
        while True:
            i = q.next(k)
            if i is None:
                return
            took[k].append(i)
            time.sleep(0.02 if k == 0 else 0.001)
    ths = [threading.Thread(target=run, args=(k,)) for k in took]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert sorted(took[0] + took[1] + took[2]) == list(range(60))
    assert len(took[0]) < len(took[1]) and len(took[0]) < len(took[2])


SKEW_WORKER = textwrap.dedent('''
    import sys, json, time
    sys.path.insert(0, %r)
    from fakebob_amd import parallel as P
    dist = P.init_process_group("gloo")
    rank, _, world = P.dist_env()
    schedule, streams = sys.argv[2], int(sys.argv[3])
    n = 40
    cost = [0.1 if i %% 2 == 0 else 0.005 for i in range(n)]    # 1 : 20, every expensive attack on an even index
    last = [0.0]
    def attack(item, thr):
        time.sleep(cost[item])
        last[0] = time.time()
        return 1, 1, 51
    dist.barrier()
    t0 = time.time()
    g = P.run_sharded(list(range(n)), attack, None, dist, schedule=schedule, streams=streams)
    with open(sys.argv[1] + "/%%s_rank%%d.json" %% (schedule, rank), "w") as w:
        json.dump({"rank": rank, "g": g[:4], "local": sorted(i for i, _ in g[4]), "busy_s": last[0] - t0}, w)
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_world_size_2_dynamic_queue_balances_skewed_attack_costs(tmp_path):
    """Attack cost varies by orders of magnitude (early stop against max_iter): with the ticket queue on the process
    group's store every attack runs exactly once and both ranks finish together, where the static round-robin deal
    leaves one rank with all the expensive ones.  Two streams per rank, as the drivers run several."""
    import json
    script = tmp_path / "skew_worker.py"
    script.write_text(SKEW_WORKER)
    env = dict(os.environ, OMP_NUM_THREADS="1")
    busy = {}
    for port, schedule in ((29619, "dynamic"), (29621, "static")):
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), str(script), str(tmp_path), schedule, "2"]
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:]
        rows = [json.load(open(str(tmp_path / ("%s_rank%d.json" % (schedule, k))))) for k in range(2)]
        assert sorted(rows[0]["local"] + rows[1]["local"]) == list(range(40))      # every attack exactly once
        assert rows[0]["g"] == rows[1]["g"] == [40, 40, 40, 51 * 40]
        busy[schedule] = [row["busy_s"] for row in rows]
    d, s = busy["dynamic"], busy["static"]
    # both ranks busy to the end: they finish within one expensive attack (0.1 s of ~0.55 s) of each other
    assert abs(d[0] - d[1]) <= 0.12 + 0.05 * max(d), busy
    assert max(s) >= 1.5 * max(d), busy                      # what the static deal costs on this list


HARDEN_WORKER = textwrap.dedent('''
    import sys, json, time
    sys.path.insert(0, %r)
    from fakebob_amd import parallel as P
    dist = P.init_process_group("gloo")
    rank, _, world = P.dist_env()
    # (1) ranks that built DIFFERENT numbers of local queues first still agree on the distributed queue's key
    for _ in range(3 if rank == 0 else 7):
        P.WorkQueue(5, None, "dynamic")
        P.WorkQueue(5, dist, "static")
    g = P.run_sharded(list(range(13)), lambda it, thr: (1, 1, 51), None, dist)
    assert g[:2] == (13, 13), g[:4]
    # (2) the job's own store (public API), not the process group's private default store
    assert P._STORE is not None and P.job_store() is P._STORE
    # (3) a stream that raises on ONE rank fails the job on EVERY rank instead of leaving the others in the all-reduce
    def attack(item, thr):
        if rank == 1 and item >= 0:
            raise ValueError("boom on rank 1")
        time.sleep(0.1)                             # (rank 1 gets to draw before rank 0 has emptied the queue)
        return 1, 1, 51
    dist.barrier()
    try:
        P.run_sharded(list(range(12)), attack, None, dist)
        outcome = "returned"
    except ValueError as ex:
        outcome = "ValueError"
    except RuntimeError as ex:
        outcome = "RuntimeError"
    with open(sys.argv[1] + "/h_rank%%d.json" %% rank, "w") as w:
        json.dump({"rank": rank, "outcome": outcome}, w)
    dist.barrier()
    dist.destroy_process_group()
''') % ROOT


def test_world_size_2_queue_keys_own_store_and_failure_propagation(tmp_path):
    """Round-5 advisor / review items on parallel.py: key numbering counts store-backed queues only, the store is this
    module's own TCPStore, and a failing rank is seen by every rank before the result reduction."""
    import json
    script = tmp_path / "harden_worker.py"
    script.write_text(HARDEN_WORKER)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
           "127.0.0.1", "--master-port", "29623", str(script), str(tmp_path)]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300,
                       env=dict(os.environ, OMP_NUM_THREADS="1"))
    assert r.returncode == 0, r.stdout[-3000:]
    rows = [json.load(open(str(tmp_path / ("h_rank%d.json" % k)))) for k in range(2)]
    assert rows[1]["outcome"] == "ValueError"       # the rank that failed re-raises its own error
    assert rows[0]["outcome"] == "RuntimeError"     # the other learns of it and raises too -- nobody blocks
