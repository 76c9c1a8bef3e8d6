"""N>1 path on CPU: world_size-2 gloo run of the sharding + broadcast + counter reduction that the
multi-GPU driver uses (the only collectives on the path)."""
import os
import subprocess
import sys
import textwrap

from fakebob_amd.parallel import reduce_counters, run_sharded, shard_indices

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
