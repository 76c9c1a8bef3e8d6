"""Driver of tools/profile/mfcc_instrumented.sh: a 51-utterance NES-sized batch (3 s each) on the instrumented library."""
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
out = np.zeros(4 * 16 * 12, np.uint64)
lib.fb_debug_mfcc_stamps(out.ctypes.data_as(C.c_void_p))
t = out.astype(np.int64).reshape(4, 16, 12)[:, :, :10] / 100.0   # us (100 MHz)
t0 = t[:, :, 0].min()
names = ["entry", "loads issued, tables copied", "workgroup barrier", "moments, window (samples arrived)", "first dft16, twiddles, transpose",
         "second dft16", "unpack, power", "combine, log", "dct, store", "mel pieces"]
print("k_mfcc_f32, stamps relative to the earliest wave entry (us); 4 workgroups x 16 waves")
order = [0, 1, 2, 3, 4, 5, 6, 9, 7, 8]      # stamp 9 (mel pieces done) sits between 6 and 7
for j, k in enumerate(order):
    n = names[k]
    r = t[:, :, k] - t0
    d = (t[:, :, k] - t[:, :, order[j - 1]]) if j else r
    print("%-40s at %6.2f .. %6.2f (mean %6.2f)   phase: mean %5.2f  min %5.2f  max %5.2f" % (n, r.min(), r.max(), r.mean(), d.mean(), d.min(), d.max()))
for g in range(4):
    print("workgroup %3d: entry of its waves %s" % (64 * g, np.array2string(t[g, :, 0] - t0, precision=2)))
    print("               end   of its waves %s" % np.array2string(t[g, :, 8] - t0, precision=2))
e.close()
