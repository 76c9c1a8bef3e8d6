"""FakeBob: the reference's attack surface (FAKEBOB.py:19-299) on top of the GPU engine.

Same constructor, `attack`, `estimate_threshold`, `get_grad` and `loss_fn` signatures and return
shapes as the reference, so attackMain.py drives it unmodified (see fakebob_amd/dropin/).  The
whole NES loop -- noise, 51-utterance scoring, loss, gradient estimate, momentum sign step --
runs in libfakebob_hip.so; this file only marshals arguments and writes the trace pickle.

The reference never seeds its RNG (FAKEBOB.py:234).  Here the noise is a Philox4x32-10 stream
keyed by `seed` (constructor keyword, default drawn from numpy's global RNG so that
`np.random.seed(...)` still makes a run reproducible) and counted by (iteration, attack index).
"""
import pickle
import time

import numpy as np

from .engine import nes_params

UNTARGETED = "untargeted"


def _col(audio):
    audio = np.asarray(audio)
    if audio.ndim == 1:
        return audio[:, np.newaxis]
    if audio.shape[0] == 1:
        return audio.T
    return audio


class FakeBob(object):

    def __init__(self, task, attack_type, model, adver_thresh=0., epsilon=0.002, max_iter=1000,
                 max_lr=0.001, min_lr=1e-6, samples_per_draw=50, sigma=0.001, momentum=0.9,
                 plateau_length=5, plateau_drop=2., seed=None, verbose=True):
        self.task = task
        self.attack_type = attack_type
        self.model = model
        self.adver_thresh = adver_thresh
        self.epsilon = epsilon
        self.max_iter = max_iter
        self.max_lr = max_lr
        self.min_lr = min_lr
        self.samples_per_draw = samples_per_draw
        self.sigma = sigma
        self.momentum = momentum
        self.plateau_length = plateau_length
        self.plateau_drop = plateau_drop
        self.threshold = 0.
        self.true = None
        self.target = None
        self.seed = int(np.random.randint(0, 2 ** 31 - 1)) if seed is None else int(seed)
        self.verbose = verbose
        self._stream = 0  # Philox stream id: one per attack()/estimate_threshold() call
        if not hasattr(model, "engine"):
            raise TypeError(
                "FakeBob needs a fakebob_amd system (gmm_OSI/gmm_CSI/gmm_SV/iv_*): the NES loop is "
                "fused with scoring on the GPU and there is no CPU fallback")
        if getattr(model, "task", task) != task:
            raise ValueError("model implements task %s, attack asked for %s" % (model.task, task))

    # ------------------------------------------------------------------ helpers
    def _params(self, attack_type=None, max_iter=None, stream=None):
        return nes_params(self.task, attack_type or self.attack_type, adver_thresh=self.adver_thresh,
                          epsilon=self.epsilon, max_iter=self.max_iter if max_iter is None else max_iter,
                          max_lr=self.max_lr, min_lr=self.min_lr, samples_per_draw=self.samples_per_draw,
                          sigma=self.sigma, momentum=self.momentum, plateau_length=self.plateau_length,
                          plateau_drop=self.plateau_drop, threshold=self.threshold, target=self.target,
                          true=self.true, seed=self.seed, stream=self._stream if stream is None else stream)

    def _score_shape(self, sc):
        S = self.model.engine.n_speakers
        return sc[0] if self.task == "SV" else sc[:S].copy()

    # -------------------------------------------------------- estimate_threshold
    def estimate_threshold(self, audio, fs=16000, bits_per_sample=16, n_jobs=10, debug=False,
                           max_total_iters=1000000):
        """FAKEBOB.py:39-137.  Returns (score, n_iters, seconds), or None for CSI."""
        if self.task == "CSI":
            print("--- Warning: no need to estimate threshold for CSI, quitting ---")
            return
        audio = _col(audio)
        t0 = time.time()
        p = self._params(attack_type=UNTARGETED)
        self._stream += 1
        score, n_iters, n_outer, thr, _adv = self.model.engine.estimate_threshold(
            p, float(self.model.threshold), audio[:, 0], max_total_iters=max_total_iters)
        self.threshold = thr
        self.delta = None
        times = time.time() - t0
        if self.verbose:
            print("--- return at iter_outer:%d, return thresh:%f ---" % (n_outer, score))
            print("cost %d iters, %fs time" % (n_iters, times))
        return score, n_iters, times

    # -------------------------------------------------------------------- attack
    def attack(self, audio, checkpoint_path, threshold=0., true=None, target=None, fs=16000,
               bits_per_sample=16, n_jobs=10, debug=False):
        """FAKEBOB.py:139-221.  Returns (int16 adversarial audio (N,1), success_flag +-1) and
        writes the per-iteration trace [distance, adver_loss, score, used_time] to
        checkpoint_path (pickle protocol -1), like the reference."""
        audio = _col(audio)
        self.threshold = threshold
        self.true = true
        self.target = target
        p = self._params()
        self._stream += 1
        t0 = time.time()
        adv, flag, _advf, trace = self.model.engine.attack(p, audio[:, 0])
        dt = time.time() - t0
        n = trace.shape[0]
        per_iter = dt / max(n, 1)
        cp_global = []
        for r in range(n):
            used = 0. if (r == n - 1 and trace[r, 1] < 0) else per_iter  # the early-stop row stores 0. (:187)
            sc = trace[r, 3:]
            cp_global.append([trace[r, 0], np.array([trace[r, 1]]), sc[0] if self.task == "SV" else sc.copy(), used])
        if checkpoint_path:
            with open(checkpoint_path, "wb") as writer:
                pickle.dump(cp_global, writer, protocol=-1)
        if self.verbose:
            print("--- %d iters, distance:%f, loss:%f, %.1f iters/s ---" %
                  (n, trace[-1, 0], trace[-1, 1], n / dt if dt > 0 else 0.0))
        return adv[:, np.newaxis], flag

    # ------------------------------------------------------------------ get_grad
    def get_grad(self, audio, fs=16000, bits_per_sample=16, n_jobs=10, debug=False, iteration=0, noise_pos=None):
        """FAKEBOB.py:223-246 -> (final_loss, grad (N,1), adver_loss (1,), score)."""
        audio = _col(audio)
        p = self._params()
        fl, grad, al, sc = self.model.engine.get_grad(p, audio[:, 0], it=iteration, noise_pos=noise_pos)
        return fl, grad[:, np.newaxis], np.array([al]), self._score_shape(sc)

    # ------------------------------------------------------------------- loss_fn
    def loss_fn(self, audios, fs=16000, bits_per_sample=16, n_jobs=10, debug=False):
        """FAKEBOB.py:248-299 (host mirror; inside attack() the same formulas run on the GPU)."""
        score = self.model.score(audios, fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        score = np.asarray(score, np.float64)
        if self.task in ("OSI", "CSI"):
            if score.ndim == 1:
                score = score[np.newaxis, :]
        elif score.ndim == 0:
            score = score[np.newaxis]
        if self.task == "OSI":
            if self.attack_type == "targeted":
                other = np.max(np.delete(score, self.target, axis=1), axis=1, keepdims=True)
                loss = np.maximum(other, self.threshold) + self.adver_thresh - score[:, self.target:self.target + 1]
            else:
                loss = self.threshold + self.adver_thresh - np.max(score, axis=1, keepdims=True)
        elif self.task == "CSI":
            if self.attack_type == "targeted":
                other = np.max(np.delete(score, self.target, axis=1), axis=1, keepdims=True)
                loss = other + self.adver_thresh - score[:, self.target:self.target + 1]
            else:
                other = np.max(np.delete(score, self.true, axis=1), axis=1, keepdims=True)
                loss = score[:, self.true:self.true + 1] + self.adver_thresh - other
        else:
            loss = self.threshold + self.adver_thresh - score[:, np.newaxis]
        return loss, score
