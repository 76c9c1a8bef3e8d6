"""round 6: k_gmm_fx2w's SKIP form on the headline workload -- the kernel solo (thresholds from the attack's last iteration)
and the NES iteration rate with one attack in flight, FB_FXW_SKIP=0 against the default, and how far the scores move."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.getcwd())
from fakebob_amd.engine import Engine, nes_params  # noqa: E402
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system  # noqa: E402

ubm, spk = synthetic_gmm_system(5, 2048, 72)
audio = synthetic_audio(0, 48000)
kw = dict(samples_per_draw=50, epsilon=0.002, sigma=0.001, max_lr=0.001, min_lr=1e-6, momentum=0.9, plateau_length=5,
          plateau_drop=2.0, adver_thresh=0.0, max_iter=1000, target=0, threshold=0.2277)
res = {}
MODES = (("0", "0"), ("1", "0"), ("0", "2"), ("1", "2"))   # (FB_GMM_IL, FB_FXW_SKIP)
for il, mode in MODES:
    os.environ["FB_FXW_SKIP"] = mode
    os.environ["FB_GMM_IL"] = il
    e = Engine(0)
    e.set_frontend(mfcc_f32=1)
    e.load_gmm([ubm] + spk)
    e.set_system("OSI")
    p = nes_params("OSI", "targeted", seed=42, stream=0, **kw)
    e.bench_nes(p, audio, 0, 40, time_gmm=0)
    ms, rows = e.bench_gmm_kernel(20)
    t0 = time.perf_counter()
    r = e.bench_nes(p, audio, -1, 100, time_gmm=1)
    dt = time.perf_counter() - t0
    ms2, _ = e.bench_gmm_kernel(20)
    print("FB_GMM_IL=%s FB_FXW_SKIP=%s: kernel solo %.1f / %.1f us (%d rows), 100 iterations %.3f ms/step, kernel in the loop %.1f us, skip launches / violations %s"
          % (il, mode, 1e3 * ms, 1e3 * ms2, rows, 1e3 * dt / 100, 1e3 * r[1] / 100, e.debug_fxw_skip()))
    e.close()
# trajectories: a short attack with the early stop off, the trace rows of both forms
tr = {}
for il, mode in MODES:
    os.environ["FB_FXW_SKIP"] = mode
    os.environ["FB_GMM_IL"] = il
    e = Engine(0)
    e.set_frontend(mfcc_f32=1)
    e.load_gmm([ubm] + spk)
    e.set_system("OSI")
    p = nes_params("OSI", "targeted", seed=42, stream=0, **dict(kw, max_iter=12, threshold=50.0))
    adv, flag, advf, trace = e.attack(p, audio)
    tr[(il, mode)] = (np.asarray(trace), np.asarray(advf), e.debug_fxw_skip())
    e.close()
a = tr[("0", "0")]
for k in MODES[1:]:
    b = tr[k]
    print("IL=%s SKIP=%s against IL=0 SKIP=0, 12 iterations: largest |score difference| %.3g, first-iteration scores identical %s, adversarial audio "
          "identical %s, skip launches / violations %s" % (k[0], k[1], np.abs(a[0][:, 3:] - b[0][:, 3:]).max(),
                                                           np.array_equal(a[0][0], b[0][0]), np.array_equal(a[1], b[1]), b[2]))
