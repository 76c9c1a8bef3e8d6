# error of the two GMM kernels against the float64-accumulating oracle (utterance-level mean log-likelihood)
import os, sys, numpy as np
sys.path.insert(0, '.')
from fakebob_amd.engine import Engine
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system, stack_models
from oracle import oracle as O
ubm, spk = synthetic_gmm_system(5, 2048, 72)
wavs = [(synthetic_audio(u, n) * 32768.0).astype(np.int16) for u, n in [(0, 48000), (1, 48000), (2, 20000), (5, 30000), (7, 48000), (9, 64000)]]
gc, miv, iv = stack_models([ubm] + spk)
raw_o, _ = O.gmm_score_batch(O.default_cfg(), wavs, gc, miv, iv, nthreads=8)
for mode in ("f32", "bx3", "fx2"):
    os.environ["FB_GMM_MODE"] = mode
    e = Engine(0); e.load_gmm([ubm] + spk)
    raw_g, _ = e.score_raw(wavs); e.close()
    d = np.abs(raw_g - raw_o)
    print(mode, "max abs err %.3e  mean %.3e  (scores ~ %.1f)" % (d.max(), d.mean(), np.abs(raw_o).mean()))
