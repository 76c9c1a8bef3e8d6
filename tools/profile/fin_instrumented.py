"""Driver of tools/profile/fin_instrumented.sh: the fused 4-launch chain of one attack (configs[1] shapes) on the instrumented library."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from fakebob_amd.engine import Engine, nes_params  # noqa: E402
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system  # noqa: E402

ubm, spk = synthetic_gmm_system(5, 2048, 72)
e = Engine(0)
e.set_frontend(mfcc_f32=1)
e.load_gmm([ubm] + spk)
e.set_system("OSI", None, None)
e.set_fused_chain(True)
kw = dict(samples_per_draw=50, epsilon=0.002, sigma=0.001, max_lr=0.001, min_lr=1e-6, momentum=0.9, plateau_length=5,
          plateau_drop=2.0, adver_thresh=0.0, max_iter=1000, target=0, threshold=0.2277)
p = nes_params("OSI", "targeted", seed=42, stream=0, **kw)
a = synthetic_audio(0, 48000)
e.bench_nes(p, a, 0, 40)
lib = C.CDLL(os.environ["FAKEBOB_HIP_LIB"])
out = np.zeros(1024 * 12, np.uint64)
lib.fb_debug_fin_stamps(out.ctypes.data_as(C.c_void_p))
t = out.astype(np.int64).reshape(1024, 12) / 100.0
nf = 51 * 6
nu = (48000 + 255) // 256
fin, upd = t[:nf, :6], t[nf:nf + nu, :6]
t0 = min(fin[:, 0].min(), upd[:, 0].min())
last = int(np.argmax(fin[:, 4]))          # the last arriver is the only one with stamps 4, 5 of this launch
print("k_gmm_finalize_loss_update, stamps relative to the earliest workgroup entry (us); %d finalising + %d update workgroups" % (nf, nu))
for k, nm in enumerate(["entry", "frame log-likelihoods", "sum, tree", "score stored, arrival counted"]):
    r = fin[:, k] - t0
    d = (fin[:, k] - fin[:, k - 1]) if k else r
    print("fin %-32s at %6.2f .. %6.2f (mean %6.2f)   phase: mean %5.2f  min %5.2f  max %5.2f" % (nm, r.min(), r.max(), r.mean(), d.mean(), d.min(), d.max()))
L = t[last] - t0
print("last arriver (workgroup %d): arrival %.2f, loss body starts %.2f; raw scores arrived %.2f, losses formed %.2f, barrier %.2f, mean of losses %.2f, decisions made %.2f, stores complete %.2f, published %.2f"
      % (last, L[3], L[4], L[11], L[6], L[7], L[8], L[9], L[10], L[5]))
for k, nm in enumerate(["entry", "normals staged, next drawn", "publication seen", "losses read", "phase 1 (step, clip)", "phase 2 (next batch)"]):
    r = upd[:, k] - t0
    d = (upd[:, k] - upd[:, k - 1]) if k else r
    print("upd %-32s at %6.2f .. %6.2f (mean %6.2f)   phase: mean %5.2f  min %5.2f  max %5.2f" % (nm, r.min(), r.max(), r.mean(), d.mean(), d.min(), d.max()))
e.close()
