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
