"""Deterministic stand-ins shared by make_golden_driver.py (which drives the REFERENCE's attackMain.py with them)
and tests/test_driver_rules.py (which drives fakebob_amd.attack_main with the same ones).  Not part of the product.

The site: data/test-set/<spk>/*.wav and data/illegal-set/<spk>/*.wav (attackMain.py:34-35); the first sample of
every wav encodes the decision the stub model will return for it (code 0 -> -1 = rejected, code c -> speaker c-1)."""
import os

import numpy as np
from scipy.io.wavfile import write

SPK_IDS = ["1580", "2830", "61"]
FS = 16000

# (set, speaker dir, file name, decision code)
VOICES = [
    ("test-set", "1580", "1580-a.wav", 1), ("test-set", "1580", "1580-b.wav", 2), ("test-set", "1580", "1580-c.wav", 1),
    ("test-set", "2830", "2830-a.wav", 2), ("test-set", "2830", "2830-b.wav", 2), ("test-set", "2830", "2830-c.wav", 3),
    ("test-set", "61", "61-a.wav", 3), ("test-set", "61", "61-b.wav", 1), ("test-set", "61", "61-c.wav", 3),
    ("test-set", "61", "61-dd.wav", 3),
    ("illegal-set", "237", "237-a.wav", 0), ("illegal-set", "237", "237-b.wav", 2), ("illegal-set", "237", "237-c.wav", 0),
    ("illegal-set", "3575", "3575-a.wav", 0), ("illegal-set", "3575", "3575-b.wav", 1),
    ("illegal-set", "3575", "3575-cc.wav", 0),
    ("illegal-set", "8230", "8230-a.wav", 0), ("illegal-set", "8230", "8230-b.wav", 0), ("illegal-set", "8230", "8230-c.wav", 0),
]


def make_site(root):
    rng = np.random.RandomState(3)
    for sub, spk, name, code in VOICES:
        d = os.path.join(root, "data", sub, spk)
        os.makedirs(d, exist_ok=True)
        a = np.clip(np.round(rng.normal(size=800) * 2000), -30000, 30000).astype(np.int16)
        a[0] = code
        write(os.path.join(d, name), FS, a)


class StubModel(object):
    """The reference's plugin API with decisions read off the audio itself."""

    def __init__(self, task, threshold=0.0):
        self.task = task
        self.spk_ids = list(SPK_IDS[:1] if task == "SV" else SPK_IDS)
        self.threshold = threshold
        self.decision_calls = 0

    @staticmethod
    def _code(audio):
        return int(round(float(np.asarray(audio).reshape(-1)[0]) * 32768.0))

    def make_decisions(self, audios, fs=16000, bits_per_sample=16, n_jobs=1, debug=False):
        self.decision_calls += 1
        lst = audios if isinstance(audios, list) else [audios]
        dec = []
        for a in lst:
            c = self._code(a)
            if self.task == "SV":
                dec.append(1 if c > 0 else -1)
            elif self.task == "CSI":
                dec.append(max(c - 1, 0))
            else:
                dec.append(c - 1)
        sc = np.zeros((len(dec), len(self.spk_ids)))
        return (dec, sc) if len(dec) > 1 else (dec[0], sc[0])

    def score(self, audios, **kw):
        return np.zeros(len(self.spk_ids))


class StubBob(object):
    """FakeBob stand-in: records every call; an attack 'succeeds' when the checkpoint path's stem has even length."""
    log = None

    def __init__(self, task, attack_type, model, **hp):
        self.task, self.attack_type, self.model, self.hp = task, attack_type, model, hp

    def estimate_threshold(self, audio, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        StubBob.log.append(("estimate", StubModel._code(audio)))
        return 0.4375, 3, 0.0

    def attack(self, audio, checkpoint_path, threshold=0., true=None, target=None, fs=16000, bits_per_sample=16,
               n_jobs=10, debug=False):
        stem = os.path.basename(checkpoint_path).split(".")[0]
        flag = 1 if len(stem) % 2 == 0 else -1
        StubBob.log.append(("attack", checkpoint_path, float(threshold), None if true is None else int(true),
                            None if target is None else int(target), flag))
        n = np.asarray(audio).reshape(-1).size
        return np.zeros((n, 1), np.int16), flag
