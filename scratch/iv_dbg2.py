import sys, numpy as np
sys.path.insert(0, '/root/repo')
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_ivector_system
from oracle import oracle as O
sy = synthetic_ivector_system(C=96, D=72, R=48, L=24, n_speakers=3, seed=11)
sy = sy.with_enrolled(sy.enrolled, z_mean=[-30.0, -50.0, -20.0], z_std=[5.0, 8.0, 4.0])
e = Engine(0); e.load_ivector(sy, "OSI")
ctx = O.IvSystemCtx(O.default_cfg(), sy, nthreads=8)
audio = synthetic_audio(4, 16000)
kw = dict(target=1, threshold=0.5)
pg = nes_params("OSI", "targeted", samples_per_draw=8, seed=3, stream=1, **kw)
flg, gg, alg, scg = e.get_grad(pg, audio, it=2)
ivs_g = e.debug_ivectors(9, sy.R)
z = O.noise(3, 2, 1, 16000, 4).astype(np.float64)        # [half][N]
cols = [audio] + [audio + pg.sigma * z[j] for j in range(4)] + [audio - pg.sigma * z[j] for j in range(4)]
wavs = [O.quantize(c) for c in cols]
llr_o, ivs_o, tv_o = ctx.score_batch(wavs)
print("iv err per utt", np.abs(ivs_g - ivs_o).max(axis=1))
llr_g2, tv_g2 = e.score_raw(wavs)
ivs_g2 = e.debug_ivectors(9, sy.R)
print("score_raw path iv err per utt", np.abs(ivs_g2 - ivs_o).max(axis=1))
