"""round 6: how far a frame's log-likelihood moves between two NES batches of the same audio (fresh noise of sigma = 0.001)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.getcwd())
from fakebob_amd.engine import Engine  # noqa: E402
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system  # noqa: E402

ubm, spk = synthetic_gmm_system(5, 2048, 72)
audio = synthetic_audio(0, 48000)
e = Engine(0)
e.set_frontend(mfcc_f32=1)
e.load_gmm([ubm] + spk)
e.set_system("OSI")
lls = []
for it in range(2):
    rng = np.random.default_rng(100 + it)
    wavs = [np.clip(np.round((audio + (0.001 * rng.normal(size=audio.size) if b else 0.0)) * 32768.0), -32768, 32767).astype(np.int16)
            for b in range(51)]
    fs = [e.debug_feats(w)[0] for w in wavs]
    tv = [f.shape[0] for f in fs]
    feats = np.concatenate(fs, axis=0)
    ll = e.debug_gmm_frames(feats)
    lls.append((ll, np.asarray(tv), feats))
(a, tva, fa), (b, tvb, fb) = lls
print("voiced counts equal in %d of 51 columns; rows %d / %d" % (int(np.sum(tva == tvb)), a.shape[1], b.shape[1]))
offa, offb = np.concatenate([[0], np.cumsum(tva)]), np.concatenate([[0], np.cumsum(tvb)])
d = []
for c in range(51):
    if tva[c] == tvb[c]:
        d.append(a[:, offa[c]:offa[c + 1]] - b[:, offb[c]:offb[c + 1]])
d = np.concatenate(d, axis=1)
print("ll differences between the batches (same column, same voiced rank): mean |d| %.3f, p99 %.3f, max %.3f nats" %
      (np.abs(d).mean(), np.percentile(np.abs(d), 99), np.abs(d).max()))
i = np.unravel_index(np.argmax(np.abs(d)), d.shape)
print("the largest: model %d, frame %d: %.2f vs %.2f" % (i[0], i[1], a[i[0], i[1]], b[i[0], i[1]]))
print("frame lls span %.1f .. %.1f" % (a.min(), a.max()))
e.close()
