"""Driver of tools/profile/fxw_instrumented.sh: the SURVEY.md 8(d) scoring batch on an instrumented library."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from fakebob_amd.engine import Engine  # noqa: E402
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system  # noqa: E402

what = sys.argv[1]
wavs = [(synthetic_audio(u % 7, 48000) * 32768).astype(np.int16) for u in range(51)]
ubm, spk = synthetic_gmm_system(5, 2048, 72)
e = Engine(0)
e.load_gmm([ubm] + spk)
e.set_system("OSI")
lib = C.CDLL(os.environ["FAKEBOB_HIP_LIB"])
if what == "COUNT":
    out = np.zeros(4, np.uint64)
    lib.fb_debug_fxw_counts(out.ctypes.data_as(C.c_void_p))
    e.score_raw(wavs)
    lib.fb_debug_fxw_counts(out.ctypes.data_as(C.c_void_p))
    print("%s, one scoring pass of 51 utterances: %d updates (per wave), %d rescues, %d reference moves" %
          ((e.gmm_kernel_variant,) + tuple(int(v) for v in out[:3])))
    print("slices (register pairs = 2 components x 64 frames) whose every value lies more than 25 log2 units below the "
          "frame's running sum: %d of %d = %.3f (what a wave-uniform skip of their exponentials could save)" %
          (int(out[3]), 16 * int(out[0]), float(out[3]) / max(1.0, 16.0 * float(out[0]))))
else:
    e.score_raw(wavs)
    for _ in range(2):
        ms, rows = e.bench_gmm_kernel(10)
    out = np.zeros(16, np.uint64)
    lib.fb_debug_fxw_stamps(out.ctypes.data_as(C.c_void_p))
    t = out.astype(np.int64)
    names = ["features requested, tables staged", "frame fragments", "anchors, reference, state", "first parameter group",
             "tile loop", "last two models' updates", "epilogue"]
    for i, n in enumerate(names):
        print("%-36s %6.2f us" % (n, (t[i + 1] - t[i]) / 100.0))
    print("workgroup 8: %.2f us; launch %.1f us (%d rows)" % ((t[7] - t[0]) / 100.0, 1e3 * ms, rows))
e.close()
