"""More than one MI355X (attackMain.py:312, 324-411).  Skipped on a one-GPU box; on a node with
several devices it exercises what has never run on hardware otherwise: engines on two devices in one process (the
per-device dynamic-LDS opt-in, fb_kernels.h) and the driver under `torch.distributed.run` with the RCCL backend --
sharding, threshold / key broadcast, the final counter all-reduce -- whose results must not depend on the sharding."""
import os
import pickle
import subprocess
import sys

import numpy as np
import pytest
from scipy.io.wavfile import read

from fakebob_amd import _native as N

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _n_dev():
    try:
        return int(N.lib().fb_device_count())
    except Exception:  # noqa: BLE001
        return 0


needs2 = pytest.mark.skipif(_n_dev() < 2, reason="needs >= 2 visible GPUs")


@needs2
def test_engines_on_two_devices_in_one_process(full_system):
    from fakebob_amd.engine import Engine
    from fakebob_amd.models import synthetic_audio
    ubm, spk = full_system
    wavs = [(synthetic_audio(u, 20000 + 3000 * u) * 32768).astype(np.int16) for u in range(5)]
    raws = []
    for dev in (0, 1, 0):
        e = Engine(dev)
        try:
            e.load_gmm([ubm] + spk)
            assert e.gmm_kernel_variant == "fx2w/1"      # > 64 KB of dynamic LDS: needs the opt-in on each device
            raws.append(e.score_raw(wavs)[0])
        finally:
            e.close()
    assert np.array_equal(raws[0], raws[1]) and np.array_equal(raws[0], raws[2])


@pytest.mark.parametrize("backend", [pytest.param("nccl", marks=needs2), "gloo"])
def test_driver_under_torchrun_matches_a_single_process_run(tmp_path, backend):
    """backend nccl: one rank per GPU over RCCL (needs two devices).  backend gloo: the same two-rank run with both
    ranks on cuda:0 (FAKEBOB_DEVICE=0) -- everything but RCCL and the second device, on a one-GPU box."""
    from tests.test_gpu_driver import _site
    ids, ubm, spk = _site(tmp_path)
    common = ["-spk_id"] + ids + ["-archi", "gmm", "-task", "CSI", "-type", "targeted", "-max_iter", "12",
                                  "-samples", "10", "--streams", "2", "--seed", "11",
                                  "--model_dir", str(tmp_path / "model"), "--pre_model_dir", str(tmp_path / "pre-models"),
                                  "--test_dir", str(tmp_path / "data" / "test-set"),
                                  "--illegal_dir", str(tmp_path / "data" / "illegal-set")]
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    if backend == "gloo":
        env["FAKEBOB_DEVICE"] = "0"
    r1 = subprocess.run([sys.executable, "-m", "fakebob_amd.attack_main"] + common + ["--out_dir", str(tmp_path / "one")],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env, cwd=ROOT)
    assert r1.returncode == 0, r1.stdout[-3000:]
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                         "--master-addr", "127.0.0.1", "--master-port", "29641", "-m", "fakebob_amd.attack_main"] + common +
                        ["--out_dir", str(tmp_path / "two"), "--dist-backend", backend],
                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900, env=env, cwd=ROOT)
    assert r2.returncode == 0, r2.stdout[-3000:]

    def summary(out):
        rate = [l for l in out.splitlines() if "attack successful rate" in l]
        done = [l for l in out.splitlines() if "generate adversarial voices done" in l]
        # "... done: A attacks, I NES iterations, U utterances scored": U also counts the benign-decision filter, which
        # every rank runs on the whole list (attackMain.py:129,201,266), so only A and I are sharding-independent
        return rate, [l.split(" NES iterations")[0] for l in done]
    assert summary(r1.stdout) == summary(r2.stdout) and summary(r1.stdout)[1]   # the all-reduced counters, rank 0's print
    # every attack of the list exists in both runs, bit for bit: Philox stream = global attack index
    n = 0
    for root, _dirs, files in os.walk(str(tmp_path / "one" / "adversarial-audio")):
        for f in files:
            a = os.path.join(root, f)
            b = a.replace(str(tmp_path / "one"), str(tmp_path / "two"))
            assert os.path.exists(b), b
            assert np.array_equal(read(a)[1], read(b)[1])
            n += 1
    assert n >= 3
    for root, _dirs, files in os.walk(str(tmp_path / "one" / "checkpoint")):
        for f in files:
            a = os.path.join(root, f)
            ta, tb = pickle.load(open(a, "rb")), pickle.load(open(a.replace(str(tmp_path / "one"), str(tmp_path / "two")), "rb"))
            assert len(ta) == len(tb)
            for ra, rb in zip(ta, tb):
                assert ra[0] == rb[0] and np.array_equal(ra[1], rb[1]) and np.array_equal(ra[2], rb[2])


def test_bench_self_launch_two_ranks_on_one_gpu_weak_scaling_arithmetic():
    """`python bench.py --gpus 2` outside torchrun re-executes itself under torch.distributed.run with one rank per
    GPU; with --same-device --dist-backend gloo both ranks share cuda:0, which runs the whole N > 1 path of the bench on
    a one-GPU box on every driver run: the self-launch, RANK / WORLD_SIZE handling, the barrier + max-over-ranks timing,
    the all-reduced step counters (asserted inside bench.py: world x steps x attacks) and rank 0's single JSON line.
    The weak-scaling arithmetic is checked against an N = 1 run of the same per-rank work: two ranks on one device do
    twice the steps, so `value` lies between the N = 1 value (perfect serialisation of the two ranks' kernels) and
    twice it (never: they share the chip), with room for launch-gap filling."""
    import json
    env = dict(os.environ, PYTHONPATH=ROOT, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    common = ["--steps", "30", "--warmup", "5", "--streams", "1", "--no-cpu-baseline", "--no-secondary", "--no-single",
              "--precondition", "20"]

    def run(extra):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + common + extra, stdout=subprocess.PIPE,
                           stderr=subprocess.PIPE, text=True, timeout=900, env=env, cwd=ROOT)
        assert r.returncode == 0, r.stderr[-3000:]
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, from rank 0 only
        return json.loads(lines[0])
    one = run([])
    two = run(["--gpus", "2", "--same-device", "--dist-backend", "gloo"])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["steps"] == 30 and two["config"]["attacks_in_flight_per_gpu"] == 1
    for d in (one, two):
        assert d["metric"].startswith("NES iterations/sec") and d["unit"] == "NES iterations/s"
        assert abs(d["scored_utts_per_s"] - 51 * d["value"]) <= 1e-6 * d["scored_utts_per_s"]
    # value = (world x steps x attacks) / max-over-ranks time
    assert abs(two["value"] - 2 * 30 / (two["ms_per_step"] * 30 * 1e-3)) <= 1e-6 * two["value"]
    assert abs(one["value"] - 1 * 30 / (one["ms_per_step"] * 30 * 1e-3)) <= 1e-6 * one["value"]
    print("bench N=1 %.0f it/s, N=2 ranks on one device %.0f it/s" % (one["value"], two["value"]))
    assert 0.8 * one["value"] <= two["value"] <= 2.05 * one["value"]
