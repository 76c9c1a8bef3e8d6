import sys, numpy as np
sys.path.insert(0, '.')
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system
ubm, spk = synthetic_gmm_system(5, 2048, 72)
e = Engine(0); e.load_gmm([ubm] + spk); e.set_system("OSI")
p = nes_params("OSI", "targeted", samples_per_draw=50, target=0, threshold=0.2277)
try:
    e.get_grad(p, synthetic_audio(0, 48000), it=0, want_grad=False)
except Exception as ex:
    print("(get_grad:", str(ex)[:60], ")")
ms, rows = e.bench_gmm_kernel(50)
print("gmm_ms %.4f rows %d  alg TF/s %.1f" % (ms, rows, 6*2048*4*72*rows/ms/1e9))
