"""Driver of tools/profile/gsel_instrumented.sh: one i-vector scoring batch of configs[2] size on a library whose
gmm_wide_kernel.hip was built with -DFB_FXW_STAMP; prints where a workgroup of k_gsel_w (the last launched: pass B) spends
its time."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from fakebob_amd.engine import Engine  # noqa: E402
from fakebob_amd.models import synthetic_audio, synthetic_ivector_system  # noqa: E402

wavs = [(synthetic_audio(u % 7, 48000) * 32768).astype(np.int16) for u in range(51)]
sy = synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=1)
e = Engine(0)
e.load_ivector(sy, "SV")
lib = C.CDLL(os.environ["FAKEBOB_HIP_LIB"])
for _ in range(3):
    e.score_raw(wavs)
out = np.zeros(16, np.uint64)
lib.fb_debug_fxw_stamps(out.ctypes.data_as(C.c_void_p))
t = out.astype(np.int64)[8:14]
names = ["frame fragments", "tau / counters", "first parameter group", "tile loop", "ids and counts"]
for i, n in enumerate(names):
    print("%-28s %6.2f us" % (n, (t[i + 1] - t[i]) / 100.0))
print("workgroup 8 of k_gsel_w<5, true>: %.2f us" % ((t[5] - t[0]) / 100.0))
e.close()
