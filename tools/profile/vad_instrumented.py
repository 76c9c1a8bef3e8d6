"""Driver of tools/profile/vad_instrumented.sh: a 51-utterance NES-sized batch (3 s each) on the instrumented library."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from fakebob_amd.engine import Engine  # noqa: E402
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system  # noqa: E402

wavs = [(synthetic_audio(u % 7, 48000) * 32768).astype(np.int16) for u in range(51)]
ubm, spk = synthetic_gmm_system(C=64, D=72, n_speakers=1)
e = Engine(0)
e.set_frontend(mfcc_f32=1)
e.load_gmm([ubm] + spk)
for _ in range(5):
    e.score_raw(wavs)
lib = C.CDLL(os.environ["FAKEBOB_HIP_LIB"])
out = np.zeros(256 * 12, np.uint64)
lib.fb_debug_vad_stamps(out.ctypes.data_as(C.c_void_p))
n = 51 * 4
t = out.astype(np.int64).reshape(256, 12)[:n, :10] / 100.0   # us (100 MHz)
t0 = t[:, 0].min()
names = ["entry", "ticket drawn, loads issued and stored", "barrier (+ restage)", "VAD mean", "votes, ranks", "deltas",
         "block sums stored", "row offsets polled", "other parts' sums polled, barrier", "rows written"]
print("k_vad_delta_cmvn_p, stamps relative to the earliest workgroup entry (us); %d workgroups" % n)
for k, nm in enumerate(names):
    r = t[:, k] - t0
    d = (t[:, k] - t[:, k - 1]) if k else r
    print("%-44s at %6.2f .. %6.2f (mean %6.2f)   phase: mean %5.2f  min %5.2f  max %5.2f" % (nm, r.min(), r.max(), r.mean(), d.mean(), d.min(), d.max()))
e.close()
