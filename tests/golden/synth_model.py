"""Exactly reproducible stand-in for a speaker-recognition `model` (the reference's plugin API,
README.md:136): scores are integer arithmetic on the int16-quantised audio, scaled by powers of
two, so they are bit-identical on every machine.  Used (a) by make_golden.py to drive the
*reference's* FakeBob when capturing golden vectors and (b) by the tests to drive the oracle /
product with the same scorer.  Not part of the product."""
import numpy as np


class SynthModel(object):
    def __init__(self, task, n_spk, n_samples, seed=0, threshold=0.0, lin_shift=20, quad_shift=37):
        rng = np.random.RandomState(seed)
        self.task = task
        self.S = 1 if task == "SV" else n_spk
        self.N = n_samples
        self.W = rng.randint(-8, 9, size=(self.S, n_samples)).astype(np.int64)
        self.Dq = rng.randint(0, 2, size=(self.S, n_samples)).astype(np.int64)
        self.bias = rng.randint(-2 ** 17, 2 ** 17, size=self.S).astype(np.int64)
        self.lin_shift, self.quad_shift = lin_shift, quad_shift
        self.threshold = threshold
        self.spk_ids = ["spk%02d" % i for i in range(self.S)]
        self.n_calls = 0
        self.n_scored = 0

    # same input normalisation + int16 cast as the reference wrappers (gmm_ubm_OSI.py:70-85)
    @staticmethod
    def to_int16_list(audios, bits_per_sample=16):
        if isinstance(audios, np.ndarray):
            if audios.ndim == 1 or (audios.ndim == 2 and (audios.shape[0] == 1 or audios.shape[1] == 1)):
                lst = [audios.reshape(-1)]
            else:
                lst = [audios[:, i] for i in range(audios.shape[1])]
        else:
            lst = [np.array(a) for a in audios]
        out = []
        for a in lst:
            if a.dtype != np.int16:
                a = (a * (2 ** (bits_per_sample - 1))).astype(np.int16)
            out.append(a)
        return out

    def raw(self, q_list):
        sc = np.empty((len(q_list), self.S), np.float64)
        for b, q in enumerate(q_list):
            q = q.astype(np.int64)
            lin = self.W @ q + self.bias            # exact int64
            quad = self.Dq @ (q * q)                # exact int64
            sc[b] = lin.astype(np.float64) * 2.0 ** -self.lin_shift - quad.astype(np.float64) * 2.0 ** -self.quad_shift
        return sc

    def score(self, audios, fs=16000, bits_per_sample=16, n_jobs=1, debug=False):
        q = self.to_int16_list(audios, bits_per_sample)
        self.n_calls += 1
        self.n_scored += len(q)
        sc = self.raw(q)
        if self.task == "SV":
            sc = sc[:, 0]
            return sc if sc.shape[0] > 1 else sc[0]
        return sc if sc.shape[0] > 1 else sc[0]

    def make_decisions(self, audios, fs=16000, bits_per_sample=16, n_jobs=1, debug=False):
        score = self.score(audios, fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        if self.task == "SV":
            if isinstance(score, np.ndarray):
                return [1 if s >= self.threshold else -1 for s in score], score
            return (1 if score >= self.threshold else -1), score
        if score.ndim == 1:
            score = score[np.newaxis, :]
        dec = list(np.argmax(score, axis=1))
        if self.task == "OSI":
            for i, v in enumerate(np.max(score, axis=1)):
                if v < self.threshold:
                    dec[i] = -1
        if score.shape[0] == 1:
            return dec[0], score.flatten()
        return dec, score


def synth_audio(n, seed, amp=3000, offset=0.0):
    rng = np.random.RandomState(seed)
    q = np.clip(np.round(rng.normal(size=n) * amp), -32768, 32767)
    return q / 32768.0 + offset
