import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from fakebob_amd.engine import Engine
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system
ubm, spk = synthetic_gmm_system(5, 2048, 72)
e = Engine(0); e.load_gmm([ubm] + spk); e.set_system("OSI")
wavs = [(synthetic_audio(u % 7, 48000) * 32768).astype(np.int16) for u in range(51)]
try:
    e.score_raw(wavs)
except Exception as ex:
    pass
for _ in range(3):
    ms, rows = e.bench_gmm_kernel(30)
print("ABL=%s  k_gmm solo %.4f ms (%d rows)" % (os.environ.get("ABL", "-"), ms, rows))
