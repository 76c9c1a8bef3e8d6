import sys, numpy as np
sys.path.insert(0,'.')
from fakebob_amd.engine import Engine
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system, stack_models
from oracle import oracle as O
e=Engine(0); ubm,spk=synthetic_gmm_system(n_speakers=2,C=96,D=72); e.load_gmm([ubm]+spk)
w=(synthetic_audio(3,1000)*32768).astype(np.int16)
try:
    raw=e.debug_mfcc(w); e.set_frontend(compress_feats=1); got=e.debug_mfcc(w)
    print("T", raw.shape, np.array_equal(got,O.compress_roundtrip(raw)), np.abs(got-raw).max())
except Exception as ex: print("short:", ex)
e.set_frontend(compress_feats=1)
wavs=[(synthetic_audio(u,n)*32768).astype(np.int16) for u,n in ((0,48000),(1,20000),(5,30000))]
rg,tv=e.score_raw(wavs); gc,miv,iv=stack_models([ubm]+spk)
ro,tvo=O.gmm_score_batch(O.default_cfg(compress_feats=1),wavs,gc,miv,iv,nthreads=4)
ru,_=O.gmm_score_batch(O.default_cfg(),wavs,gc,miv,iv,nthreads=4)
print("gpu-oracle(compressed)", np.abs(rg-ro).max(), "compressed-uncompressed", np.abs(ru-ro).max(), tv, tvo)
