"""FakeBob: the reference's attack surface (FAKEBOB.py:19-299) on top of the GPU engine.

Same constructor, `attack`, `estimate_threshold`, `get_grad` and `loss_fn` signatures and return
shapes as the reference, so attackMain.py drives it unmodified (see fakebob_amd/dropin/).  The
whole NES loop -- noise, 51-utterance scoring, loss, gradient estimate, momentum sign step --
runs in libfakebob_hip.so; this file only marshals arguments and writes the trace pickle.

A `model` that is not one of this package's systems -- the reference's plugin API (README.md:136: "just wrap
your system by providing score and make_decisions") -- is driven through `fb_attack_ext` / `fb_get_grad_ext`: its
`score` is called once per NES iteration on the (N, samples_per_draw + 1) float64 batch, exactly as FAKEBOB.py:250
does, while noise, perturbation, loss, gradient estimate, momentum sign step and clipping stay on the GPU.

The reference never seeds its RNG (FAKEBOB.py:234).  Here the noise is a Philox4x32-10 stream
keyed by `seed` (constructor keyword, default drawn from numpy's global RNG so that
`np.random.seed(...)` still makes a run reproducible) and counted by (iteration, attack index).
"""
import pickle
import time

import numpy as np

from .engine import nes_params

UNTARGETED = "untargeted"


def _check_bits(bits_per_sample):
    """bits_per_sample scales the int16 casts of the NES path exactly as in the reference -- every sample of the
    batch before scoring (gmm_ubm_OSI.py:85) and the returned adversarial audio (FAKEBOB.py:220) use
    2^(bits_per_sample-1); the container stays int16, so 2 .. 16."""
    b = int(bits_per_sample)
    if b < 2 or b > 16:
        raise ValueError("bits_per_sample=%r: the int16 casts take 2 .. 16" % (bits_per_sample,))
    return b


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
        if task not in ("OSI", "CSI", "SV"):
            raise ValueError("task must be OSI, CSI or SV")
        # a fakebob_amd system carries the engine its models live in: the whole NES iteration is fused with scoring
        # on the GPU.  Any other object is the reference's plugin API: it needs score (+ make_decisions for
        # estimate_threshold); only the NES arithmetic around it runs on the GPU.
        self._native = hasattr(model, "engine")
        self._own_engine = None
        if not self._native and not callable(getattr(model, "score", None)):
            raise TypeError("model must provide score(audios, fs=, bits_per_sample=, n_jobs=, debug=) "
                            "(and make_decisions for estimate_threshold), README.md:136 of the reference")
        if self._native and getattr(model, "task", task) != task:
            raise ValueError("model implements task %s, attack asked for %s" % (model.task, task))
        self._n_spk = None

    # ------------------------------------------------------------------ helpers
    def _params(self, attack_type=None, max_iter=None, stream=None, bits_per_sample=16):
        return nes_params(self.task, attack_type or self.attack_type, adver_thresh=self.adver_thresh,
                          epsilon=self.epsilon, max_iter=self.max_iter if max_iter is None else max_iter,
                          max_lr=self.max_lr, min_lr=self.min_lr, samples_per_draw=self.samples_per_draw,
                          sigma=self.sigma, momentum=self.momentum, plateau_length=self.plateau_length,
                          plateau_drop=self.plateau_drop, threshold=self.threshold, target=self.target,
                          true=self.true, seed=self.seed, stream=self._stream if stream is None else stream,
                          bits_per_sample=bits_per_sample)

    def _score_shape(self, sc):
        S = self.model.engine.n_speakers if self._native else self._speakers()
        return sc[0] if self.task == "SV" else sc[:S].copy()

    # ---------------------------------------------------- foreign models (plugin API)
    def _engine(self):
        """The engine that runs the NES kernels: the model's own, or (foreign model) one without any model loaded."""
        if self._native:
            return self.model.engine
        if self._own_engine is None:
            from .engine import Engine
            from .systems import default_device
            self._own_engine = Engine(default_device())
        return self._own_engine

    def _speakers(self, probe_audio=None, **score_kw):
        """Number of score columns of a foreign model: 1 for SV, len(model.spk_ids) when the model has the
        attribute the reference's drivers read (attackMain.py:95), else the width of ONE extra score call made the
        way the reference calls score (same keywords); a model that counts its queries should carry spk_ids."""
        if self._n_spk is None:
            if self.task == "SV":
                self._n_spk = 1
            elif hasattr(self.model, "spk_ids"):
                self._n_spk = len(self.model.spk_ids)
            elif probe_audio is not None:
                self._n_spk = int(np.asarray(self.model.score(probe_audio, **score_kw)).size)
            else:
                raise ValueError("cannot tell how many speakers the model scores: give it a spk_ids attribute")
        return self._n_spk

    def _score_fn(self, fs, bits_per_sample, n_jobs, debug):
        def fn(audios):  # (N, B) float64, the batch FAKEBOB.py:250 hands to model.score
            return self.model.score(audios, fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        return fn

    def _estimate_threshold_foreign(self, audio, fs, bits_per_sample, n_jobs, debug, max_total_iters, noise_all):
        """FAKEBOB.py:39-137 around a black-box model: its make_decisions decides, its score feeds the device NES
        gradient (fb_get_grad_ext); the momentum / plateau / sign-step bookkeeping is a few length-N numpy lines."""
        kw = dict(fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        eng = self._engine()
        init = np.asarray(self.model.score(audio, **kw), np.float64)   # FAKEBOB.py:53
        if self._n_spk is None and self.task != "SV" and not hasattr(self.model, "spk_ids"):
            self._n_spk = int(init.size)                               # its width tells S: no extra query
        S = self._speakers(audio, **kw)
        init = float(np.max(init)) if self.task == "OSI" else float(init.reshape(-1)[0])
        delta = abs(init / 10)
        self.delta = delta
        self.threshold = init + delta
        lo = np.clip(audio - self.epsilon, -1., 1.)
        hi = np.clip(audio + self.epsilon, -1., 1.)
        adver = np.array(audio, np.float64, copy=True)
        velocity = 0.
        n_iters = n_outer = 0
        spent = 0.
        fn = self._score_fn(fs, bits_per_sample, n_jobs, debug)
        while True:
            lr = self.max_lr
            history = []
            while True:
                t0 = time.time()
                decision, score = self.model.make_decisions(adver, **kw)
                score = float(np.max(score)) if self.task == "OSI" else float(np.asarray(score).reshape(-1)[0])
                if decision != -1:
                    if self.verbose:
                        print("--- return at iter_outer:%d, return thresh:%f ---" % (n_outer, score))
                        print("cost %d iters, %fs time" % (n_iters, spent))
                    return score, n_iters, spent
                if score >= self.threshold:
                    break
                if n_iters >= max_total_iters:
                    raise RuntimeError("estimate_threshold: max_total_iters %d reached" % max_total_iters)
                p = self._params(attack_type=UNTARGETED, bits_per_sample=bits_per_sample)
                noise = None if noise_all is None else noise_all[n_iters]
                loss, g, _, _ = eng.get_grad_ext(p, S, fn, adver[:, 0], it=n_iters, noise_pos=noise)
                velocity = self.momentum * velocity + (1.0 - self.momentum) * g[:, np.newaxis]
                history = (history + [loss])[-self.plateau_length:]
                if len(history) == self.plateau_length and history[-1] > history[0]:
                    if lr > self.min_lr:
                        lr = max(lr / self.plateau_drop, self.min_lr)
                    history = []
                adver = np.clip(adver - lr * np.sign(velocity), lo, hi)
                n_iters += 1
                spent += time.time() - t0
            self.threshold += delta
            n_outer += 1

    # -------------------------------------------------------- estimate_threshold
    def estimate_threshold(self, audio, fs=16000, bits_per_sample=16, n_jobs=10, debug=False,
                           max_total_iters=1000000, noise_all=None):
        """FAKEBOB.py:39-137.  Returns (score, n_iters, seconds), or None for CSI.
        noise_all (extension): the (iterations, N, samples_per_draw//2) normals np.random.normal would have
        returned, to replay a NumPy run; default: the device Philox stream."""
        if self.task == "CSI":
            print("--- Warning: no need to estimate threshold for CSI, quitting ---")
            return
        audio = _col(audio)
        bits = _check_bits(bits_per_sample)
        if not self._native:
            self._stream += 1
            return self._estimate_threshold_foreign(audio, fs, bits_per_sample, n_jobs, debug, max_total_iters, noise_all)
        t0 = time.time()
        p = self._params(attack_type=UNTARGETED, bits_per_sample=bits)
        self._stream += 1
        score, n_iters, n_outer, thr, _adv = self.model.engine.estimate_threshold(
            p, float(self.model.threshold), audio[:, 0], noise_all=noise_all, max_total_iters=max_total_iters)
        self.threshold = thr
        self.delta = None
        times = time.time() - t0
        if self.verbose:
            print("--- return at iter_outer:%d, return thresh:%f ---" % (n_outer, score))
            print("cost %d iters, %fs time" % (n_iters, times))
        return score, n_iters, times

    # -------------------------------------------------------------------- attack
    def attack(self, audio, checkpoint_path, threshold=0., true=None, target=None, fs=16000,
               bits_per_sample=16, n_jobs=10, debug=False, noise_all=None):
        """FAKEBOB.py:139-221.  Returns (int16 adversarial audio (N,1), success_flag +-1) and
        writes the per-iteration trace [distance, adver_loss, score, used_time] to
        checkpoint_path (pickle protocol -1), like the reference.  used_time is each iteration's own time, read from
        the device clock where its loss is evaluated (fb_attack_iter_seconds; the loop runs inside the library), 0. on
        the early-stop row (:187).
        noise_all (extension): (max_iter, N, samples_per_draw//2) normals to replay a NumPy run."""
        audio = _col(audio)
        bits = _check_bits(bits_per_sample)
        self.threshold = threshold
        self.true = true
        self.target = target
        p = self._params(bits_per_sample=bits)
        self._stream += 1
        t0 = time.time()
        kw = dict(fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
        if self._native:
            eng = self.model.engine
            adv, flag, _advf, trace = eng.attack(p, audio[:, 0], noise_all=noise_all)
        else:
            eng = self._engine()
            adv, flag, _advf, trace = eng.attack_ext(
                p, self._speakers(audio, **kw), self._score_fn(fs, bits_per_sample, n_jobs, debug), audio[:, 0],
                noise_all=noise_all)
        dt = time.time() - t0
        n = trace.shape[0]
        used_time = eng.attack_iter_seconds(n)
        cp_global = []
        for r in range(n):
            used = 0. if (r == n - 1 and trace[r, 1] < 0) else float(used_time[r])  # the early-stop row stores 0. (:187)
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
        p = self._params(bits_per_sample=_check_bits(bits_per_sample))
        if self._native:
            fl, grad, al, sc = self.model.engine.get_grad(p, audio[:, 0], it=iteration, noise_pos=noise_pos)
        else:
            kw = dict(fs=fs, bits_per_sample=bits_per_sample, n_jobs=n_jobs, debug=debug)
            fl, grad, al, sc = self._engine().get_grad_ext(
                p, self._speakers(audio, **kw), self._score_fn(fs, bits_per_sample, n_jobs, debug), audio[:, 0],
                it=iteration, noise_pos=noise_pos)
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
