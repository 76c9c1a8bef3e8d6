# k_mfcc_r4 on the headline batch (51 x 3 s) next to a tiny GMM (C=32) instead of the 2048-component one:
# same MFCC work, but the chip is not at its power limit -> what the kernel takes at full clock
import sys, numpy as np
import os; sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '.'))
from fakebob_amd.engine import Engine
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system
C = int(sys.argv[1]) if len(sys.argv) > 1 else 32
ubm, spk = synthetic_gmm_system(5, C, 72)
e = Engine(0); e.load_gmm([ubm] + spk)
wavs = [(synthetic_audio(u % 7, 48000) * 32768).astype(np.int16) for u in range(51)]
for _ in range(60):
    e.score_raw(wavs)
e.close()
