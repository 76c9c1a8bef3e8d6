import sys, numpy as np
sys.path.insert(0, '/root/repo')
from fakebob_amd.engine import Engine
from fakebob_amd.models import synthetic_audio, synthetic_ivector_system
from oracle import oracle as O
sy = synthetic_ivector_system(C=96, D=72, R=48, L=24, n_speakers=3, seed=11)
e = Engine(0); e.load_ivector(sy, "OSI")
ctx = O.IvSystemCtx(O.default_cfg(), sy, nthreads=8)
for B, n in [(4, 48000), (9, 16000), (9, 16000), (12, 20000), (4, 48000)]:
    wavs = [(synthetic_audio(u + B, n) * 32768.0).astype(np.int16) for u in range(B)]
    llr_g, tv_g = e.score_raw(wavs)
    llr_o, ivs_o, tv_o = ctx.score_batch(wavs)
    ivs_g = e.debug_ivectors(len(wavs), sy.R)
    print(B, n, "iv err", np.abs(ivs_g - ivs_o).max(), "llr err", np.abs(llr_g - llr_o).max(), "per-utt", np.abs(ivs_g - ivs_o).max(axis=1).round(8))
