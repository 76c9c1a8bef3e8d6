"""The cross-workgroup exchanges of the kernels rest on gfx950 behaviour (agent-scope relaxed atomics, "ready" carried by
the data word, no fences: DESIGN.md section 6).  fakebob_amd/lib/libfakebob_hip_fenced.so is the same library with release /
acquire semantics on every one of those words (csrc/fb_device.h, -DFB_FENCED): what the HIP memory model asks for.  A race
in the fence-free protocol would show as a difference between the two builds -- 200 NES iterations of BASELINE.json
configs[1] (GMM-UBM OSI, 5 speakers + UBM, spd = 50, 3 s) and 50 of configs[2] (i-vector-PLDA SV), every launch chain, trace
and adversarial audio compared bit for bit."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import os, sys
import numpy as np
sys.path.insert(0, %(root)r)
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system, synthetic_ivector_system
out = {}
audio = synthetic_audio(3, 48000)
ubm, spk = synthetic_gmm_system(5, 2048, 72)
for fused in (True, False):
    e = Engine(0)
    e.set_frontend(mfcc_f32=1)
    e.load_gmm([ubm] + spk)
    e.set_system("OSI")
    e.set_fused_chain(fused)
    p = nes_params("OSI", "targeted", seed=42, stream=3, samples_per_draw=50, epsilon=0.002, sigma=0.001, max_lr=0.001, min_lr=1e-6,
                   momentum=0.9, plateau_length=5, plateau_drop=2.0, adver_thresh=0.0, max_iter=200, target=0, threshold=1.0e3)
    adv, flag, advf, trace = e.attack(p, audio)
    out["gmm_%%d_trace" %% fused] = np.asarray(trace)[:, :3 + 5].copy()
    out["gmm_%%d_adv" %% fused] = np.asarray(advf)
    e.close()
sy = synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=1)
sy = sy.with_enrolled(sy.enrolled, [-40.0], [10.0])
e = Engine(0)
e.set_frontend(mfcc_f32=1)
e.load_ivector(sy, "SV")
p = nes_params("SV", "targeted", seed=42, stream=1, samples_per_draw=50, epsilon=0.002, sigma=0.001, max_iter=50, threshold=1.0e3)
adv, flag, advf, trace = e.attack(p, audio)
out["iv_trace"] = np.asarray(trace)[:, :4].copy()
out["iv_adv"] = np.asarray(advf)
e.close()
np.savez(sys.argv[1], **out)
'''


def _run(lib, path):
    env = dict(os.environ, FAKEBOB_HIP_LIB=lib)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}, path], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    return np.load(path)


def test_fenced_build_gives_the_same_bits(tmp_path):
    from fakebob_amd import build
    lib, fenced = build.LIB, build.variant_path("fenced")
    if not os.path.exists(fenced):
        pytest.skip("libfakebob_hip_fenced.so not built (__graft_entry__.build() builds it)")
    a = _run(lib, str(tmp_path / "plain.npz"))
    b = _run(fenced, str(tmp_path / "fenced.npz"))
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        x, y = a[k], b[k]
        assert x.shape == y.shape, k
        assert np.array_equal(x.view(np.uint64) if x.dtype == np.float64 else x, y.view(np.uint64) if y.dtype == np.float64 else y), k
    assert a["gmm_1_trace"].shape[0] == 200 and a["iv_trace"].shape[0] == 50
