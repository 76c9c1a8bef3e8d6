"""Driver of tools/profile/rw_instrumented.sh: a 51-utterance i-vector batch (C = 2048, R = 400) on the instrumented library."""
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
for _ in range(3):
    e.score_raw(wavs)
lib = C.CDLL(os.environ["FAKEBOB_HIP_LIB"])
out = np.zeros(16 * 8 + 8, np.uint64)
lib.fb_debug_rw_stamps(out.ctypes.data_as(C.c_void_p))
t = out.astype(np.int64).reshape(-1)[:16 * 8].reshape(16, 8) / 100.0   # us (100 MHz)
t0 = t[0, 0]
print("block row: start | last column: flag seen, block formed, inverse seen | diagonal formed | factored (all relative to row 0's start, us)")
prev = None
for rb in range(13):
    r = t[rb] - t0
    line = "row %2d: start %7.2f" % (rb, r[0])
    if rb > 0:
        line += "  flag %7.2f  formed %7.2f  inverse %7.2f" % (r[1], r[2], r[3])
    line += "  diag %7.2f  factored %7.2f" % (r[4], r[5])
    if prev is not None:
        line += "   (+%.2f since the previous row's factor)" % (r[5] - prev)
    prev = r[5]
    print(line)
print("back substitution: %.2f us, ends %.2f us after row 0's start" % (t[15, 1] - t[15, 0], t[15, 1] - t0))
e.close()
