"""Engine: one GPU + one HIP stream worth of device state behind the C ABI.

Thin numpy <-> pointer marshalling only; all arithmetic of the hot path runs
in libfakebob_hip.so.
"""
import ctypes as C

import numpy as np

from . import _native as N
from .models import stack_models


def nes_params(task, attack_type, adver_thresh=0., epsilon=0.002, max_iter=1000, max_lr=0.001,
               min_lr=1e-6, samples_per_draw=50, sigma=0.001, momentum=0.9, plateau_length=5,
               plateau_drop=2., threshold=0., target=None, true=None, seed=42, stream=0, bits_per_sample=16):
    p = N.NesParams()
    p.task = N.TASK[task]
    p.attack_type = N.ATTACK[attack_type]
    p.adver_thresh = float(adver_thresh); p.epsilon = float(epsilon); p.max_iter = int(max_iter)
    p.max_lr = float(max_lr); p.min_lr = float(min_lr); p.samples_per_draw = int(samples_per_draw)
    p.sigma = float(sigma); p.momentum = float(momentum); p.plateau_length = int(plateau_length)
    p.plateau_drop = float(plateau_drop); p.threshold = float(threshold)
    p.target = 0 if target is None else int(target)
    p.true_label = 0 if true is None else int(true)
    p.seed = int(seed); p.stream = int(stream)
    p.bits_per_sample = int(bits_per_sample)
    return p


class Engine(object):
    def __init__(self, device=0):
        self._L = N.lib()
        self._h = C.c_void_p()
        N.check(self._L.fb_engine_create(C.c_int(device), C.byref(self._h)))
        self.device = device
        self.task = "OSI"
        self.n_models = 0
        self.cfg = N.FrontendCfg()
        self._L.fb_default_frontend(C.byref(self.cfg))

    def close(self):
        if self._h:
            self._L.fb_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration
    def set_frontend(self, **over):
        old = {}
        for k, v in over.items():
            if not hasattr(self.cfg, k):
                raise KeyError(k)
            old[k] = getattr(self.cfg, k)
            setattr(self.cfg, k, v)
        try:
            N.check(self._L.fb_set_frontend(self._h, C.byref(self.cfg)))
        except Exception:
            for k, v in old.items():      # the engine kept its previous configuration: so does the mirror
                setattr(self.cfg, k, v)
            raise

    @property
    def feat_dim(self):
        return self.cfg.num_ceps * (self.cfg.delta_order + 1)

    def load_gmm(self, models):
        self.load_gmm_arrays(*stack_models(models))

    def load_gmm_arrays(self, gc, miv, iv):
        """gconsts (M, C), means_invvars (M, C, D), inv_vars (M, C, D), float32 -- Kaldi's DiagGmm members."""
        gc, miv, iv = (np.ascontiguousarray(a, np.float32) for a in (gc, miv, iv))
        M, Cn, D = miv.shape
        N.check(self._L.fb_load_gmm(self._h, C.c_int(M), C.c_int(Cn), C.c_int(D), N.ptr(gc),
                                    N.ptr(miv), N.ptr(iv)))
        self.n_models = M
        self._gmm_shape = (Cn, D)
        self.task = "OSI"
        self.kind = "gmm"

    def load_ivector(self, system, task="OSI"):
        """system: models.IvectorSystem"""
        sy = N.IvectorSystem()
        sy.C, sy.D, sy.R, sy.L, sy.S = system.C, system.D, system.R, system.L, system.S
        sy.lda_cols = system.lda.shape[1]
        sy.num_gselect = system.num_gselect
        self._iv_nsel = system.num_gselect
        sy.min_post = system.min_post
        sy.prior_offset = system.prior_offset
        keep = []
        for name in ("fg_weights", "fg_means_invcovars", "fg_inv_covars", "ie_M", "ie_sigma_inv", "mean_vec",
                     "lda", "plda_mean", "plda_transform", "plda_psi", "enrolled", "z_mean", "z_std"):
            a = getattr(system, name)
            keep.append(a)
            setattr(sy, name, a.ctypes.data)
        N.check(self._L.fb_load_ivector(self._h, C.byref(sy), C.c_int(N.TASK[task])))
        self.n_models = system.S
        self.task = task
        self.kind = "iv"

    def gmm_acc_stats(self, wav):
        """UBM (loaded alone) posterior statistics of one utterance: occ[C], F[C,D] (float64), voiced frames."""
        wav = np.ascontiguousarray(wav).reshape(-1)
        if wav.dtype != np.int16:
            raise TypeError("gmm_acc_stats takes int16 samples")
        Cn, D = self._gmm_shape
        occ = np.empty(Cn, np.float64)
        F = np.empty((Cn, D), np.float64)
        tv = C.c_int()
        N.check(self._L.fb_gmm_acc_stats(self._h, N.ptr(wav), C.c_int64(wav.size), N.ptr(occ), N.ptr(F), C.byref(tv)))
        return occ, F, tv.value

    def last_ivectors(self, B, R):
        """i-vectors (B, R) of the batch scored last with an i-vector system."""
        out = np.empty((B, R), np.float64)
        N.check(self._L.fb_last_ivectors(self._h, C.c_int(B), N.ptr(out)))
        return out

    debug_ivectors = last_ivectors   # name the parity tests use

    @property
    def gmm_kernel(self):
        """'fx2' | 'bx3': the diagonal-GMM arithmetic the loaded model runs on (fb_gmm_kernel_mode)."""
        rc = self._L.fb_gmm_kernel_mode(self._h)
        if rc < 0:
            N.check(rc)
        return {1: "bx3", 2: "fx2"}[rc]

    @property
    def gmm_kernel_variant(self):
        """The kernel the loaded GMM system is scored with: 'bx3', 'fx2', or 'fx2w/P' -- the one-wave-per-SIMD
        kernel with the speaker models as deltas from model 0, P = 1 .. 3 partial products per delta item
        (fb_gmm_kernel_variant)."""
        rc = self._L.fb_gmm_kernel_variant(self._h, None)
        if rc < 0:
            N.check(rc)
        return {1: "bx3", 2: "fx2"}.get(rc) or "fx2w/%d" % (rc - 10)

    @property
    def gmm_delta_tiles(self):
        """(tiles with 1, 2, 3 partial products per K chunk in their delta items): k_gmm_fx2w's per-tile choice for the
        loaded model (fb_gmm_delta_tiles); (0, 0, 0) when another kernel scores it."""
        a, b, c = C.c_int(), C.c_int(), C.c_int()
        rc = self._L.fb_gmm_delta_tiles(self._h, C.byref(a), C.byref(b), C.byref(c))
        if rc < 0:
            N.check(rc)
        return (a.value, b.value, c.value)

    @property
    def gmm_delta_tiles_f6(self):
        """Tiles of the F6 class (corrections as block-scaled fp6 products, fb_gmm_delta_tiles_f6)."""
        rc = self._L.fb_gmm_delta_tiles_f6(self._h)
        if rc < 0:
            N.check(rc)
        return rc

    @property
    def gmm_shift_rms(self):
        """fb_load_gmm's measure of how far the models were adapted from model 0 (what P is chosen from)."""
        v = C.c_double()
        rc = self._L.fb_gmm_kernel_variant(self._h, C.byref(v))
        if rc < 0:
            N.check(rc)
        return v.value

    def debug_iv_active(self):
        n = C.c_int()
        N.check(self._L.fb_debug_iv_active(self._h, C.byref(n)))
        return n.value

    def debug_iv_gselect(self, rows_cap=None):
        """(sel[rows][num_gselect], info dict) of the last i-vector batch -- fb_debug_iv_gselect."""
        info = (C.c_int64 * 5)()
        N.check(self._L.fb_debug_iv_gselect(self._h, None, C.c_int64(0), info))
        rows, nsel = int(info[4]), int(self._iv_nsel)
        sel = np.empty((rows, nsel), np.int32)
        N.check(self._L.fb_debug_iv_gselect(self._h, N.ptr(sel), C.c_int64(sel.size), info))
        return sel, dict(threshold_path=bool(info[0]), path=int(info[0]), overflow=int(info[1]), max_list=int(info[2]),
                         survivors=int(info[3]), rows=rows)

    def set_system(self, task, z_mean=None, z_std=None):
        zm = None if z_mean is None else np.ascontiguousarray(z_mean, np.float64)
        zs = None if z_std is None else np.ascontiguousarray(z_std, np.float64)
        N.check(self._L.fb_set_system(self._h, C.c_int(N.TASK[task]),
                                      None if zm is None else N.ptr(zm),
                                      None if zs is None else N.ptr(zs)))
        self.task = task

    @property
    def n_speakers(self):
        return self._L.fb_num_speakers(self._h)

    # ---- scoring
    def score_raw(self, audio_list, bits_per_sample=16):
        """list of 1-D arrays (int16, or float in [-1,1]) -> raw[B,M], tv[B]."""
        B = len(audio_list)
        off = np.zeros(B + 1, np.int64)
        off[1:] = np.cumsum([a.size for a in audio_list])
        raw = np.empty((B, self.n_models), np.float64)
        tv = np.empty(B, np.int32)
        if all(a.dtype == np.int16 for a in audio_list):
            cat = np.ascontiguousarray(np.concatenate([a.reshape(-1) for a in audio_list]))
            N.check(self._L.fb_score_i16(self._h, N.ptr(cat), N.ptr(off), C.c_int(B), N.ptr(raw), N.ptr(tv)))
        else:
            # mixed / float input: int16 entries are exact in float64 after /2^(bits-1)
            scale = float(2 ** (bits_per_sample - 1))
            parts = [a.reshape(-1).astype(np.float64) / scale if a.dtype == np.int16
                     else a.reshape(-1).astype(np.float64) for a in audio_list]
            cat = np.ascontiguousarray(np.concatenate(parts))
            N.check(self._L.fb_score_f64(self._h, N.ptr(cat), N.ptr(off), C.c_int(B),
                                         C.c_int(bits_per_sample), N.ptr(raw), N.ptr(tv)))
        return raw, tv

    def system_scores(self, raw):
        raw = np.ascontiguousarray(raw, np.float64)
        B = raw.shape[0]
        out = np.empty((B, self.n_speakers), np.float64)
        N.check(self._L.fb_system_scores(self._h, N.ptr(raw), C.c_int(B), N.ptr(out)))
        return out

    # ---- NES
    def get_grad(self, params, audio, it=0, noise_pos=None, want_grad=True):
        audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
        n = audio.size
        npz = None if noise_pos is None else np.ascontiguousarray(noise_pos, np.float64)
        if npz is not None and npz.shape != (n, params.samples_per_draw // 2):
            raise ValueError("noise_pos must be (N, samples_per_draw//2)")
        grad = np.empty(n, np.float64) if want_grad else None
        fl, al = C.c_double(), C.c_double()
        sc = np.empty(max(self.n_speakers, 1), np.float64)
        N.check(self._L.fb_get_grad(self._h, C.byref(params), N.ptr(audio), C.c_int64(n), C.c_uint32(it),
                                    None if npz is None else N.ptr(npz), C.byref(fl),
                                    None if grad is None else N.ptr(grad), C.byref(al), N.ptr(sc)))
        return fl.value, grad, al.value, sc

    def attack(self, params, audio, noise_all=None):
        audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
        n = audio.size
        na = None if noise_all is None else np.ascontiguousarray(noise_all, np.float64)
        S = self.n_speakers
        adv = np.empty(n, np.int16)
        adv_f = np.empty(n, np.float64)
        trace = np.zeros((max(params.max_iter, 1), 3 + S), np.float64)
        nt, flag = C.c_int(), C.c_int()
        N.check(self._L.fb_attack(self._h, C.byref(params), N.ptr(audio), C.c_int64(n),
                                  None if na is None else N.ptr(na), N.ptr(adv), N.ptr(adv_f),
                                  N.ptr(trace), C.byref(nt), C.byref(flag)))
        return adv, flag.value, adv_f, trace[:nt.value]

    def attack_iter_seconds(self, n):
        """Seconds per iteration of the last attack / attack_ext (device clock, fb_attack_iter_seconds)."""
        out = np.zeros(max(int(n), 0), np.float64)
        if n > 0:
            N.check(self._L.fb_attack_iter_seconds(self._h, N.ptr(out), C.c_int(int(n))))
        return out

    # ---- NES with a foreign model (the reference's plugin API): scores come from `score_fn`
    @staticmethod
    def _score_cb(score_fn, S, err):
        """score_fn(audios (N, B) float64) -> (B, S) scores, wrapped as an fb_score_cb.  The model gets its OWN
        C-contiguous (N, B) array, as the reference hands it one (FAKEBOB.py:237): the callback's buffer is the
        engine's pinned staging area, recycled on the next iteration, and a model may keep or edit its batch.  An
        exception raised by the model is kept in err[0] and re-raised by the caller (a ctypes callback cannot
        propagate it)."""
        def _cb(_ctx, aud, n, b, out):
            try:
                a = np.ascontiguousarray(np.ctypeslib.as_array(aud, shape=(b, n)).T)   # (N, B) copy, columns = utterances
                sc = np.asarray(score_fn(a), np.float64).reshape(b, S)
                np.ctypeslib.as_array(out, shape=(b, S))[...] = sc
                return 0
            except BaseException as ex:  # noqa: BLE001
                err[0] = ex
                return 1
        return N.SCORE_CB(_cb)

    @staticmethod
    def _raise_cb(err, ex):
        if getattr(ex, "code", None) == N.FB_E_CALLBACK and err[0] is not None:
            raise err[0]
        raise ex

    def get_grad_ext(self, params, S, score_fn, audio, it=0, noise_pos=None):
        """fb_get_grad_ext: one NES gradient estimate at `audio` scored by score_fn."""
        audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
        n = audio.size
        npz = None if noise_pos is None else np.ascontiguousarray(noise_pos, np.float64)
        if npz is not None and npz.shape != (n, params.samples_per_draw // 2):
            raise ValueError("noise_pos must be (N, samples_per_draw//2)")
        grad = np.empty(n, np.float64)
        fl, al = C.c_double(), C.c_double()
        sc = np.empty(S, np.float64)
        err = [None]
        cb = self._score_cb(score_fn, S, err)
        try:
            N.check(self._L.fb_get_grad_ext(self._h, C.byref(params), C.c_int(S), cb, None, N.ptr(audio),
                                            C.c_int64(n), C.c_uint32(it), None if npz is None else N.ptr(npz),
                                            C.byref(fl), N.ptr(grad), C.byref(al), N.ptr(sc)))
        except N.NativeError as ex:
            self._raise_cb(err, ex)
        return fl.value, grad, al.value, sc

    def attack_ext(self, params, S, score_fn, audio, noise_all=None):
        """fb_attack_ext: the whole attack loop with a foreign scorer -> (int16 adv, flag, float64 adv, trace)."""
        audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
        n = audio.size
        na = None if noise_all is None else np.ascontiguousarray(noise_all, np.float64)
        adv = np.empty(n, np.int16)
        adv_f = np.empty(n, np.float64)
        trace = np.zeros((max(params.max_iter, 1), 3 + S), np.float64)
        nt, flag = C.c_int(), C.c_int()
        err = [None]
        cb = self._score_cb(score_fn, S, err)
        try:
            N.check(self._L.fb_attack_ext(self._h, C.byref(params), C.c_int(S), cb, None, N.ptr(audio), C.c_int64(n),
                                          None if na is None else N.ptr(na), N.ptr(adv), N.ptr(adv_f), N.ptr(trace),
                                          C.byref(nt), C.byref(flag)))
        except N.NativeError as ex:
            self._raise_cb(err, ex)
        return adv, flag.value, adv_f, trace[:nt.value]

    def estimate_threshold(self, params, model_threshold, audio, noise_all=None, max_total_iters=100000):
        audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
        n = audio.size
        na = None if noise_all is None else np.ascontiguousarray(noise_all, np.float64)
        sc, tf = C.c_double(), C.c_double()
        ni, no = C.c_int(), C.c_int()
        adv_f = np.empty(n, np.float64)
        N.check(self._L.fb_estimate_threshold(self._h, C.byref(params), C.c_double(model_threshold),
                                              N.ptr(audio), C.c_int64(n), None if na is None else N.ptr(na),
                                              C.c_int(max_total_iters), C.byref(sc), C.byref(ni), C.byref(no),
                                              C.byref(tf), N.ptr(adv_f)))
        return sc.value, ni.value, no.value, tf.value, adv_f

    # ---- debug / bench hooks
    def debug_noise(self, seed, it, stream, n, half):
        z = np.empty((half, n), np.float32)
        N.check(self._L.fb_debug_noise(self._h, C.c_uint64(seed), C.c_uint32(it), C.c_uint32(stream),
                                       C.c_int64(n), C.c_int(half), N.ptr(z)))
        return z

    def debug_quantize(self, x, bits_per_sample=16):
        x = np.ascontiguousarray(x, np.float64).reshape(-1)
        q = np.empty(x.size, np.int16)
        N.check(self._L.fb_debug_quantize(self._h, N.ptr(x), C.c_int64(x.size), C.c_int(bits_per_sample), N.ptr(q)))
        return q

    def _num_frames(self, n):
        c = self.cfg
        if c.snip_edges:
            return 0 if n < c.frame_length else 1 + (n - c.frame_length) // c.frame_shift
        return (n + c.frame_shift // 2) // c.frame_shift

    def debug_mfcc(self, wav):
        wav = np.ascontiguousarray(wav, np.int16)
        T = self._num_frames(wav.size)
        out = np.empty((T, self.cfg.num_ceps), np.float32)
        To = C.c_int()
        N.check(self._L.fb_debug_mfcc(self._h, N.ptr(wav), C.c_int64(wav.size), N.ptr(out), C.byref(To)))
        return out[:To.value]

    def debug_feats(self, wav):
        wav = np.ascontiguousarray(wav, np.int16)
        T = self._num_frames(wav.size)
        out = np.empty((max(T, 1), self.feat_dim), np.float32)
        tv, To = C.c_int(), C.c_int()
        N.check(self._L.fb_debug_feats(self._h, N.ptr(wav), C.c_int64(wav.size), N.ptr(out), C.byref(tv),
                                       C.byref(To)))
        return out[:tv.value].copy(), To.value

    def debug_gmm_frames(self, feats):
        """Per-frame log-likelihoods [M, T] of `feats` [T, D] from the GMM kernel the engine scores with (test hook)."""
        feats = np.ascontiguousarray(feats, np.float32)
        T = feats.shape[0]
        out = np.empty((self.n_models, T), np.float64)
        N.check(self._L.fb_debug_gmm_frames(self._h, N.ptr(feats), C.c_int(T), N.ptr(out)))
        return out

    def stats(self):
        a, b, c, d = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        N.check(self._L.fb_stats(self._h, C.byref(a), C.byref(b), C.byref(c), C.byref(d)))
        return dict(scored_utts=a.value, scored_frames=b.value, voiced_frames=c.value, nes_iters=d.value)

    def bench_gmm_kernel(self, reps=20):
        ms, rows = C.c_double(), C.c_int64()
        N.check(self._L.fb_bench_gmm_kernel(self._h, C.c_int(reps), C.byref(ms), C.byref(rows)))
        return ms.value, rows.value

    def set_fused_chain(self, on):
        """True: 4 launches per NES iteration (best for one or two attacks per GPU); False: 6 launches, finalisation and loss on their own (better
        with >= 3 engines sharing a GPU); None: library default.  Same trajectories (fb_set_fused_chain)."""
        N.check(self._L.fb_set_fused_chain(self._h, C.c_int(-1 if on is None else (1 if on else 0))))

    def bench_nes(self, params, audio, warmup, iters, time_gmm=False):
        """Runs warmup+iters NES iterations (identical work to attack(), early stop disabled).
        Returns (ms over the timed iters [HIP events], summed GMM-kernel ms [HIP events around
        each launch on the engine stream], voiced rows of the last batch)."""
        audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
        ms, msg, rows = C.c_double(), C.c_double(), C.c_int64()
        N.check(self._L.fb_bench_nes(self._h, C.byref(params), N.ptr(audio), C.c_int64(audio.size),
                                     C.c_int(warmup), C.c_int(iters), C.c_int(int(time_gmm)),
                                     C.byref(ms), C.byref(msg), C.byref(rows)))
        return ms.value, msg.value, rows.value
