"""ctypes front door of the CPU oracle (oracle/fb_oracle.c).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, __graft_entry__.smoke() and the
``cpu_baseline`` leg of bench.py -- never by the product package fakebob_amd/.
See fb_oracle.h for the parity status (NES engine pinned by golden vectors
from the reference's Python; Kaldi-side arithmetic "parity unpinned").
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libfb_oracle.so")

TASK = {"OSI": 0, "CSI": 1, "SV": 2}
ATTACK = {"untargeted": 0, "targeted": 1}


class FrontendCfg(C.Structure):
    _fields_ = [
        ("sample_freq", C.c_double), ("frame_length", C.c_int), ("frame_shift", C.c_int),
        ("padded_length", C.c_int), ("num_mel_bins", C.c_int), ("num_ceps", C.c_int),
        ("low_freq", C.c_double), ("high_freq", C.c_double), ("preemph", C.c_double),
        ("cepstral_lifter", C.c_double), ("snip_edges", C.c_int), ("remove_dc", C.c_int),
        ("use_energy", C.c_int), ("raw_energy", C.c_int), ("energy_floor", C.c_double),
        ("vad_energy_threshold", C.c_double), ("vad_energy_mean_scale", C.c_double),
        ("vad_proportion_threshold", C.c_double), ("vad_frames_context", C.c_int),
        ("delta_window", C.c_int), ("delta_order", C.c_int), ("cmn_window", C.c_int), ("text_scores", C.c_int), ("compress_feats", C.c_int),
        ("mfcc_f32", C.c_int),
    ]


class NesParams(C.Structure):
    _fields_ = [
        ("task", C.c_int), ("attack_type", C.c_int), ("adver_thresh", C.c_double),
        ("epsilon", C.c_double), ("max_iter", C.c_int), ("max_lr", C.c_double),
        ("min_lr", C.c_double), ("samples_per_draw", C.c_int), ("sigma", C.c_double),
        ("momentum", C.c_double), ("plateau_length", C.c_int), ("plateau_drop", C.c_double),
        ("threshold", C.c_double), ("target", C.c_int), ("true_label", C.c_int),
        ("n_spk", C.c_int),
    ]


SCORE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int,
                       C.POINTER(C.c_double))


class GmmSystem(C.Structure):
    _fields_ = [
        ("cfg", FrontendCfg), ("task", C.c_int), ("M", C.c_int), ("C", C.c_int), ("D", C.c_int),
        ("gconsts", C.c_void_p), ("means_invvars", C.c_void_p), ("inv_vars", C.c_void_p),
        ("z_mean", C.c_void_p), ("z_std", C.c_void_p), ("nthreads", C.c_int),
        ("scored_utts", C.c_int64),
    ]


def build(force=False):
    """Compile libfb_oracle.so with gcc (oracle/Makefile)."""
    src = os.path.join(_HERE, "fb_oracle.c")
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < max(
            os.path.getmtime(src), os.path.getmtime(os.path.join(_HERE, "fb_oracle.h"))):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfb_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB)
        _lib.fbo_diag_gmm_loglikes.restype = C.c_double
        _lib.fbo_np_sum.restype = C.c_double
        _lib.fbo_plda_llr.restype = C.c_double
    return _lib


def _p(a, t=C.c_void_p):
    return a.ctypes.data_as(t)


def default_cfg(**over):
    cfg = FrontendCfg()
    lib().fbo_default_cfg(C.byref(cfg))
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


def philox(ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    k = (C.c_uint32 * 2)(*key)
    o = (C.c_uint32 * 4)()
    lib().fbo_philox4x32_10(c, k, o)
    return [int(x) for x in o]


def noise(seed, it, stream, N, half):
    z = np.empty((half, N), np.float32)
    lib().fbo_noise(C.c_uint64(seed), C.c_uint32(it), C.c_uint32(stream), C.c_int64(N),
                    C.c_int(half), _p(z))
    return z


def quantize(x, bits=16):
    x = np.ascontiguousarray(x, np.float64)
    q = np.empty(x.shape, np.int16)
    lib().fbo_quantize(_p(x), C.c_int64(x.size), C.c_int(bits), _p(q))
    return q


def num_frames(cfg, n):
    return lib().fbo_num_frames(C.byref(cfg), C.c_int64(n))


def mfcc(cfg, wav):
    wav = np.ascontiguousarray(wav, np.int16)
    T = num_frames(cfg, wav.size)
    out = np.empty((T, cfg.num_ceps), np.float32)
    got = lib().fbo_mfcc(C.byref(cfg), _p(wav), C.c_int64(wav.size), _p(out))
    if got != T:
        raise ValueError("fbo_mfcc computed %d of %d frames (a configuration the float32 path does not take?)" % (got, T))
    return out


def compress_roundtrip(mat):
    """Kaldi CompressedMatrix round trip (copy-feats --compress=true) of a (T, ncols) float32 matrix."""
    m = np.array(mat, np.float32, order="C", copy=True)
    lib().fbo_compress_roundtrip(_p(m), C.c_int(m.shape[0]), C.c_int(m.shape[1]))
    return m


def vad(cfg, mf):
    mf = np.ascontiguousarray(mf, np.float32)
    v = np.empty(mf.shape[0], np.uint8)
    lib().fbo_vad(C.byref(cfg), _p(mf), C.c_int(mf.shape[0]), _p(v))
    return v


def deltas(cfg, mf):
    mf = np.ascontiguousarray(mf, np.float32)
    out = np.empty((mf.shape[0], mf.shape[1] * (cfg.delta_order + 1)), np.float32)
    lib().fbo_deltas(C.byref(cfg), _p(mf), C.c_int(mf.shape[0]), _p(out))
    return out


def cmvn_sliding(cfg, feats):
    f = np.array(feats, np.float32, order="C", copy=True)
    lib().fbo_cmvn_sliding(C.byref(cfg), _p(f), C.c_int(f.shape[0]), C.c_int(f.shape[1]))
    return f


def frontend(cfg, wav):
    wav = np.ascontiguousarray(wav, np.int16)
    T = num_frames(cfg, wav.size)
    dim = lib().fbo_feat_dim(C.byref(cfg))
    f = np.empty((max(T, 1), dim), np.float32)
    To = C.c_int()
    tv = lib().fbo_frontend(C.byref(cfg), _p(wav), C.c_int64(wav.size), _p(f), C.byref(To))
    return f[:tv].copy(), To.value


def set_logsumexp(full_sum):
    """Process-wide: False (default) = Kaldi's VectorBase<float>::LogSumExp in the diagonal-GMM frame log-likelihood
    (components below max + log(FLT_EPSILON) are not summed), True = the full float64 sum.  Returns the old setting."""
    old = bool(lib().fbo_get_logsumexp())
    lib().fbo_set_logsumexp(C.c_int(1 if full_sum else 0))
    return old


def diag_gmm_loglikes(gconsts, miv, iv, feats):
    gconsts = np.ascontiguousarray(gconsts, np.float32)
    miv = np.ascontiguousarray(miv, np.float32)
    iv = np.ascontiguousarray(iv, np.float32)
    feats = np.ascontiguousarray(feats, np.float32)
    Cn, D = miv.shape
    ll = np.empty(feats.shape[0], np.float32)
    tot = lib().fbo_diag_gmm_loglikes(_p(gconsts), _p(miv), _p(iv), C.c_int(Cn), C.c_int(D),
                                      _p(feats), C.c_int(feats.shape[0]), _p(ll))
    return ll, tot


def gmm_score_batch(cfg, wavs, gconsts, miv, iv, nthreads=1):
    """wavs: list of int16 arrays.  models: gconsts[M,C], miv/iv [M,C,D] float32.
    returns raw[B,M] average log-likelihoods, tv[B]."""
    gconsts = np.ascontiguousarray(gconsts, np.float32)
    miv = np.ascontiguousarray(miv, np.float32)
    iv = np.ascontiguousarray(iv, np.float32)
    M, Cn, D = miv.shape
    B = len(wavs)
    off = np.zeros(B + 1, np.int64)
    off[1:] = np.cumsum([len(w) for w in wavs])
    cat = np.ascontiguousarray(np.concatenate([np.asarray(w, np.int16) for w in wavs]))
    raw = np.empty((B, M), np.float64)
    tv = np.empty(B, np.int32)
    rc = lib().fbo_gmm_score_batch(C.byref(cfg), _p(cat), _p(off), C.c_int(B), _p(gconsts),
                                   _p(miv), _p(iv), C.c_int(M), C.c_int(Cn), C.c_int(D),
                                   _p(raw), _p(tv), C.c_int(nthreads))
    if rc != 0:
        raise RuntimeError("oracle: utterance %d has no voiced frames" % (-rc - 1))
    return raw, tv


def gmm_acc_stats(cfg, wav, gconsts, miv, iv):
    """UBM posterior statistics of one utterance: (occ[C], F[C,D], voiced frames)."""
    gconsts = np.ascontiguousarray(gconsts, np.float32).reshape(-1)
    miv = np.ascontiguousarray(miv, np.float32)
    iv = np.ascontiguousarray(iv, np.float32)
    Cn, D = miv.shape[-2:]
    wav = np.ascontiguousarray(wav, np.int16)
    occ = np.empty(Cn, np.float64)
    F = np.empty((Cn, D), np.float64)
    lib().fbo_gmm_acc_stats.restype = C.c_int
    tv = lib().fbo_gmm_acc_stats(C.byref(cfg), _p(wav), C.c_int64(wav.size), _p(gconsts), _p(miv), _p(iv),
                                 C.c_int(Cn), C.c_int(D), _p(occ), _p(F))
    if tv <= 0:
        raise RuntimeError("oracle: no voiced frames")
    return occ, F, tv


def map_update_means(means, occ, F, tau=10.0):
    means = np.ascontiguousarray(means, np.float64)
    Cn, D = means.shape
    out = np.empty((Cn, D), np.float64)
    lib().fbo_map_update_means.restype = None
    lib().fbo_map_update_means(_p(means), _p(np.ascontiguousarray(occ, np.float64)),
                               _p(np.ascontiguousarray(F, np.float64)), C.c_int(Cn), C.c_int(D),
                               C.c_double(tau), _p(out))
    return out


def round6(x):
    lib().fbo_round6.restype = C.c_double
    return float(lib().fbo_round6(C.c_double(float(x))))


def np_sum(a):
    a = np.ascontiguousarray(a, np.float64)
    return lib().fbo_np_sum(_p(a), C.c_int64(a.size))


def loss(task, attack_type, score, threshold=0.0, adver_thresh=0.0, target=0, true=0):
    score = np.ascontiguousarray(score, np.float64)
    if score.ndim == 1:
        score = score[:, None]
    B, S = score.shape
    out = np.empty(B, np.float64)
    lib().fbo_loss(C.c_int(TASK[task]), C.c_int(ATTACK[attack_type]), _p(score), C.c_int(B),
                   C.c_int(S), C.c_double(threshold), C.c_double(adver_thresh),
                   C.c_int(target if target is not None else 0),
                   C.c_int(true if true is not None else 0), _p(out))
    return out


def nes_params(task, attack_type, n_spk, adver_thresh=0., epsilon=0.002, max_iter=1000,
               max_lr=0.001, min_lr=1e-6, samples_per_draw=50, sigma=0.001, momentum=0.9,
               plateau_length=5, plateau_drop=2., threshold=0., target=None, true=None):
    p = NesParams()
    p.task = TASK[task]; p.attack_type = ATTACK[attack_type]
    p.adver_thresh = adver_thresh; p.epsilon = epsilon; p.max_iter = max_iter
    p.max_lr = max_lr; p.min_lr = min_lr; p.samples_per_draw = samples_per_draw
    p.sigma = sigma; p.momentum = momentum; p.plateau_length = plateau_length
    p.plateau_drop = plateau_drop; p.threshold = threshold
    p.target = 0 if target is None else int(target)
    p.true_label = 0 if true is None else int(true)
    p.n_spk = n_spk
    return p


def py_score_fn(pyfn, S):
    """Wrap ``pyfn(audios[N,B] float64) -> scores[B,S]`` as an fbo_score_fn."""
    def _cb(ctx, aud, N, B, out):
        a = np.ctypeslib.as_array(aud, shape=(B, N)).T  # (N,B) view like the reference's
        s = np.asarray(pyfn(a), np.float64).reshape(B, S)
        np.ctypeslib.as_array(out, shape=(B, S))[...] = s
        return 0
    return SCORE_FN(_cb)


class GmmSystemCtx(object):
    """fbo_gmm_system wrapper: the oracle's own gmm_OSI/SV/CSI.score."""

    def __init__(self, cfg, task, gconsts, miv, iv, z_mean=None, z_std=None, nthreads=1):
        self.gc = np.ascontiguousarray(gconsts, np.float32)
        self.miv = np.ascontiguousarray(miv, np.float32)
        self.iv = np.ascontiguousarray(iv, np.float32)
        M, Cn, D = self.miv.shape
        self.zm = np.ascontiguousarray(z_mean if z_mean is not None else np.zeros(M), np.float64)
        self.zs = np.ascontiguousarray(z_std if z_std is not None else np.ones(M), np.float64)
        s = GmmSystem()
        s.cfg = cfg; s.task = TASK[task]; s.M = M; s.C = Cn; s.D = D
        s.gconsts = self.gc.ctypes.data; s.means_invvars = self.miv.ctypes.data
        s.inv_vars = self.iv.ctypes.data; s.z_mean = self.zm.ctypes.data
        s.z_std = self.zs.ctypes.data; s.nthreads = nthreads; s.scored_utts = 0
        self.s = s
        self.fn = C.cast(lib().fbo_gmm_system_score, SCORE_FN)
        self.ctx = C.cast(C.pointer(s), C.c_void_p)
        self.S = M if task == "CSI" else M - 1

    def score(self, audios):
        """audios (N,B) float64 -> (B,S)"""
        a = np.ascontiguousarray(np.asarray(audios, np.float64).T)
        B, N = a.shape
        out = np.empty((B, self.S), np.float64)
        rc = lib().fbo_gmm_system_score(self.ctx, _p(a), C.c_int64(N), C.c_int(B), _p(out))
        if rc:
            raise RuntimeError("oracle gmm score rc=%d" % rc)
        return out


def get_grad(p, fn, ctx, audio, noise_pos=None, seed=0, it=0, stream=0):
    audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
    N = audio.size
    npz = None if noise_pos is None else np.ascontiguousarray(noise_pos, np.float64)
    grad = np.empty(N, np.float64)
    fl = C.c_double(); al = C.c_double()
    sc = np.empty(p.n_spk, np.float64)
    rc = lib().fbo_get_grad(C.byref(p), fn, ctx, _p(audio), C.c_int64(N),
                            None if npz is None else _p(npz), C.c_uint64(seed), C.c_uint32(it),
                            C.c_uint32(stream), C.byref(fl), _p(grad), C.byref(al), _p(sc))
    if rc:
        raise RuntimeError("oracle get_grad rc=%d" % rc)
    return fl.value, grad, al.value, sc


def attack(p, fn, ctx, audio, noise_all=None, seed=0, stream=0):
    audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
    N = audio.size
    na = None if noise_all is None else np.ascontiguousarray(noise_all, np.float64)
    adv = np.empty(N, np.int16)
    adv_f = np.empty(N, np.float64)
    S = p.n_spk
    trace = np.zeros((max(p.max_iter, 1), 3 + S), np.float64)
    nt = C.c_int()
    flag = lib().fbo_attack(C.byref(p), fn, ctx, _p(audio), C.c_int64(N),
                            None if na is None else _p(na), C.c_uint64(seed), C.c_uint32(stream),
                            _p(adv), _p(adv_f), _p(trace), C.byref(nt))
    if flag == 0:
        raise RuntimeError("oracle attack failed")
    return adv, flag, adv_f, trace[:nt.value]


def estimate_threshold(p, model_threshold, fn, ctx, audio, noise_all=None, max_total_iters=10000,
                       seed=0, stream=0):
    audio = np.ascontiguousarray(audio, np.float64).reshape(-1)
    N = audio.size
    na = None if noise_all is None else np.ascontiguousarray(noise_all, np.float64)
    sc = C.c_double(); ni = C.c_int(); no = C.c_int(); tf = C.c_double()
    adv_f = np.empty(N, np.float64)
    rc = lib().fbo_estimate_threshold(C.byref(p), C.c_double(model_threshold), fn, ctx, _p(audio),
                                      C.c_int64(N), None if na is None else _p(na),
                                      C.c_int(max_total_iters), C.c_uint64(seed),
                                      C.c_uint32(stream), C.byref(sc), C.byref(ni), C.byref(no),
                                      C.byref(tf), _p(adv_f))
    if rc == 1:
        return None
    if rc:
        raise RuntimeError("oracle estimate_threshold rc=%d" % rc)
    return sc.value, ni.value, no.value, tf.value, adv_f


# ---------------------------------------------------------------- i-vector / PLDA
class IvSystem(C.Structure):
    """fbo_iv_system"""
    _fields_ = [
        ("C", C.c_int), ("D", C.c_int), ("R", C.c_int), ("L", C.c_int), ("S", C.c_int),
        ("num_gselect", C.c_int), ("min_post", C.c_double),
        ("dg_gconsts", C.c_void_p), ("dg_means_invvars", C.c_void_p), ("dg_inv_vars", C.c_void_p),
        ("fg_gconsts", C.c_void_p), ("fg_means_invcovars", C.c_void_p), ("fg_inv_covars", C.c_void_p),
        ("sigma_inv_m", C.c_void_p), ("u", C.c_void_p), ("prior_offset", C.c_double),
        ("mean_vec", C.c_void_p), ("lda", C.c_void_p), ("lda_cols", C.c_int),
        ("plda_mean", C.c_void_p), ("plda_transform", C.c_void_p), ("plda_psi", C.c_void_p),
        ("train", C.c_void_p), ("z_mean", C.c_void_p), ("z_std", C.c_void_p),
        ("cfg", FrontendCfg), ("nthreads", C.c_int),
    ]


class IvSystemCtx(object):
    """The oracle's own iv_OSI/iv_CSI/iv_SV.score.  Derived variables (fgmm-global-to-gmm,
    FullGmm gconsts, Sigma^-1 M, U, PLDA-space enrolled vectors) are computed here with numpy,
    independently of the engine's C++/HIP derivations."""

    def __init__(self, cfg, sysm, nthreads=1, share=None):
        """share: another IvSystemCtx over the same UBM / extractor / back-end whose derived variables are reused
        (only the enrolled speakers and z-norm statistics differ)."""
        f32, f64 = np.float32, np.float64
        Cn, D, R, L, S = sysm.C, sysm.D, sysm.R, sysm.L, sysm.S
        if share is not None:
            for name in ("dg_iv", "dg_miv", "dg_gc", "fg_gc", "fg_mic", "fg_P", "sim", "u", "mean_vec", "lda",
                         "plda_mean", "plda_tr", "plda_psi"):
                setattr(self, name, getattr(share, name))
            self._finish(cfg, sysm, nthreads)
            return
        r, c = np.tril_indices(D)
        P = np.zeros((Cn, D, D), f64)
        P[:, r, c] = sysm.fg_inv_covars.astype(f64)
        P[:, c, r] = sysm.fg_inv_covars.astype(f64)
        covar = np.linalg.inv(P)
        mic = sysm.fg_means_invcovars.astype(f64)
        mean = np.einsum("kde,ke->kd", covar, mic)
        var = np.einsum("kdd->kd", covar)
        w = sysm.fg_weights.astype(f64)
        self.dg_iv = np.ascontiguousarray((1.0 / var).astype(f32))
        self.dg_miv = np.ascontiguousarray((mean * (1.0 / var)).astype(f32))
        iv64, miv64 = self.dg_iv.astype(f64), self.dg_miv.astype(f64)
        self.dg_gc = np.ascontiguousarray((np.log(w) - 0.5 * D * np.log(2 * np.pi) + 0.5 * np.log(iv64).sum(1)
                                           - 0.5 * (miv64 * miv64 / iv64).sum(1)).astype(f32))
        _, logdet = np.linalg.slogdet(P)
        self.fg_gc = np.ascontiguousarray((np.log(w) - 0.5 * (D * np.log(2 * np.pi) - logdet
                                                               + np.einsum("kd,kd->k", mic, mean))).astype(f32))
        self.fg_mic = sysm.fg_means_invcovars
        self.fg_P = sysm.fg_inv_covars
        Sinv = np.zeros((Cn, D, D), f64)
        Sinv[:, r, c] = sysm.ie_sigma_inv
        Sinv[:, c, r] = sysm.ie_sigma_inv
        self.sim = np.ascontiguousarray(np.matmul(Sinv, sysm.ie_M))           # Sigma^-1 M   [C][D][R]
        rr, cc = np.tril_indices(R)
        self.u = np.empty((Cn, rr.size), f64)                                 # U_k = M_k^T Sigma_k^-1 M_k, packed lower
        for k0 in range(0, Cn, 128):                                          # (batched BLAS; 128 components at a time)
            U = np.matmul(sysm.ie_M[k0:k0 + 128].transpose(0, 2, 1), self.sim[k0:k0 + 128])
            self.u[k0:k0 + 128] = U[:, rr, cc]
        self.mean_vec = np.ascontiguousarray(sysm.mean_vec.astype(f64))
        self.lda = np.ascontiguousarray(sysm.lda.astype(f64))
        self.plda_mean, self.plda_tr, self.plda_psi = sysm.plda_mean, sysm.plda_transform, sysm.plda_psi
        self._finish(cfg, sysm, nthreads)

    def _finish(self, cfg, sysm, nthreads):
        f64 = np.float64
        Cn, D, R, L, S = sysm.C, sysm.D, sysm.R, sysm.L, sysm.S
        self.zm, self.zs = sysm.z_mean, sysm.z_std
        self.train = np.zeros((S, L), f64)
        s = IvSystem()
        s.C, s.D, s.R, s.L, s.S = Cn, D, R, L, S
        s.num_gselect, s.min_post = sysm.num_gselect, sysm.min_post
        s.dg_gconsts, s.dg_means_invvars, s.dg_inv_vars = self.dg_gc.ctypes.data, self.dg_miv.ctypes.data, self.dg_iv.ctypes.data
        s.fg_gconsts, s.fg_means_invcovars, s.fg_inv_covars = self.fg_gc.ctypes.data, self.fg_mic.ctypes.data, self.fg_P.ctypes.data
        s.sigma_inv_m, s.u, s.prior_offset = self.sim.ctypes.data, self.u.ctypes.data, sysm.prior_offset
        s.mean_vec, s.lda, s.lda_cols = self.mean_vec.ctypes.data, self.lda.ctypes.data, self.lda.shape[1]
        s.plda_mean, s.plda_transform, s.plda_psi = self.plda_mean.ctypes.data, self.plda_tr.ctypes.data, self.plda_psi.ctypes.data
        s.train, s.z_mean, s.z_std = self.train.ctypes.data, self.zm.ctypes.data, self.zs.ctypes.data
        s.cfg = cfg
        s.nthreads = nthreads
        self.s = s
        self.S, self.R, self.L = S, R, L
        for i in range(S):  # enrolled i-vectors through the same back-end (n = 1)
            self.train[i] = self.backend(sysm.enrolled[i].astype(f64))
        self.fn = C.cast(lib().fbo_iv_system_score, SCORE_FN)
        self.ctx = C.cast(C.pointer(s), C.c_void_p)

    def backend(self, ivec):
        ivec = np.ascontiguousarray(ivec, np.float64)
        y = np.empty(self.L, np.float64)
        lib().fbo_iv_backend(C.byref(self.s), _p(ivec), _p(y))
        return y

    def stats(self, feats):
        feats = np.ascontiguousarray(feats, np.float32)
        g = np.empty(self.s.C, np.float64)
        X = np.empty((self.s.C, self.s.D), np.float64)
        lib().fbo_iv_stats(C.byref(self.s), _p(feats), C.c_int(feats.shape[0]), _p(g), _p(X))
        return g, X

    def extract(self, gamma, X):
        iv = np.empty(self.R, np.float64)
        rc = lib().fbo_iv_extract(C.byref(self.s), _p(np.ascontiguousarray(gamma)), _p(np.ascontiguousarray(X)), _p(iv))
        if rc:
            raise RuntimeError("oracle: i-vector system not positive definite")
        return iv

    def score_batch(self, wavs):
        """list of int16 arrays -> (llr[B,S], ivectors[B,R], tv[B])"""
        B = len(wavs)
        off = np.zeros(B + 1, np.int64)
        off[1:] = np.cumsum([len(w) for w in wavs])
        cat = np.ascontiguousarray(np.concatenate([np.asarray(w, np.int16) for w in wavs]))
        llr = np.empty((B, self.S), np.float64)
        ivs = np.empty((B, self.R), np.float64)
        tv = np.empty(B, np.int32)
        rc = lib().fbo_iv_score_batch(C.byref(self.s), _p(cat), _p(off), C.c_int(B), _p(llr), _p(ivs), _p(tv))
        if rc:
            raise RuntimeError("oracle iv score rc=%d" % rc)
        return llr, ivs, tv

    def score(self, audios):
        a = np.ascontiguousarray(np.asarray(audios, np.float64).T)
        B, N = a.shape
        out = np.empty((B, self.S), np.float64)
        rc = lib().fbo_iv_system_score(self.ctx, _p(a), C.c_int64(N), C.c_int(B), _p(out))
        if rc:
            raise RuntimeError("oracle iv system score rc=%d" % rc)
        return out
