"""Enrolment on the device -- counterpart of the reference's build_spk_models.py.

build_spk_models.py turns `data/enrollment-set/<spk>-<utt>.wav` (one voice per speaker) and
`data/z-norm-set/*.wav` into `model/<spk>.iv`, `model/<spk>.gmm` (pickles `[spk_id, utt_id,
identity_location, z_mean, z_std]`, :152-156, :272-277) plus `model/<spk>-identity.gmm`.  It does so with
Kaldi programs; here the same steps run on the engine:

* GMM-UBM identity (:184-224): `gmm-global-acc-stats --update-flags=m` = posterior statistics of the UBM on
  the enrolment voice (`Engine.gmm_acc_stats`, GPU) followed by `gmm-global-est-map --update-flags=m`
  (gmm-global-est-map.cc:62-92: `MapDiagGmmUpdate`, means only, mean_tau = 10): an element-wise float64
  formula (`map_adapt_means`).  Variances and weights stay bit-identical to the UBM's, which is what lets
  the scoring kernel share the quadratic term between the UBM and every speaker model.
* GMM z-norm (:259-261): raw average log-likelihoods of the z-norm voices under every identity model,
  `np.mean` / `np.std` over the voices.
* i-vector identity (:104-150): the enrolment voice's i-vector; z-norm (:120-135): PLDA scores of the z-norm
  voices against that i-vector, `np.mean` / `np.std`.
"""
import os
import pickle

import numpy as np
from scipy.io.wavfile import read

from .kaldi_io import load_ivector_pre_models, read_diag_gmm, write_diag_gmm
from .models import DiagGmm, IvectorSystem

MEAN_TAU = 10.0   # MapDiagGmmOptions default (gmm-global-est-map.cc:31 registers the option, nothing overrides it)


def gmm_weights(gmm):
    """component weights recovered from the gconsts (for a DiagGmm that did not come from a Kaldi file)."""
    miv = gmm.means_invvars.astype(np.float64)
    iv = gmm.inv_vars.astype(np.float64)
    D = iv.shape[1]
    logw = gmm.gconsts.astype(np.float64) + 0.5 * D * np.log(2.0 * np.pi) - 0.5 * np.sum(np.log(iv), axis=1) \
        + 0.5 * np.sum(miv * miv / iv, axis=1)
    w = np.exp(logw)
    return w / w.sum()


def map_adapt_means(ubm, weights, occ, F, tau=MEAN_TAU):
    """MapDiagGmmUpdate with update-flags 'm': mean' = F/(occ+tau) + tau/(occ+tau) * mean for occ > 0, in the
    float64 'normal' form (means = means_invvars * (1/inv_vars)), written back as means_invvars = mean' *
    inv_vars; inv_vars are untouched and gconsts recomputed."""
    iv = ubm.inv_vars.astype(np.float64)
    means = ubm.means_invvars.astype(np.float64) * (1.0 / iv)
    occ = np.asarray(occ, np.float64)[:, None]
    new = np.where(occ > 0.0, np.asarray(F, np.float64) * (1.0 / (occ + tau)) + (tau / (occ + tau)) * means, means)
    return DiagGmm.from_internal(weights, (new * iv).astype(np.float32), ubm.inv_vars)


def _i16(a):
    a = np.asarray(a)
    return a if a.dtype == np.int16 else (a.astype(np.float64) * 32768.0).astype(np.int16)


def enroll_gmm(ubm, weights, enroll_audios, znorm_audios, device=0):
    """-> (identity models, z_mean[S], z_std[S]) for one enrolment voice per speaker."""
    from .engine import Engine
    e = Engine(device)
    try:
        e.load_gmm([ubm])
        ids = []
        for a in enroll_audios:
            occ, F, _ = e.gmm_acc_stats(_i16(a))
            ids.append(map_adapt_means(ubm, weights, occ, F))
        e.load_gmm(ids)
        raw, _ = e.score_raw([_i16(a) for a in znorm_audios])      # (n_znorm, S) average frame log-likelihoods
    finally:
        e.close()
    return ids, np.mean(raw, axis=0).flatten(), np.std(raw, axis=0).flatten()


def enroll_ivector(pre, enroll_audios, znorm_audios, device=0, num_gselect=20, min_post=0.025):
    """pre: dict from kaldi_io.load_ivector_pre_models (or the same arrays).  -> (i-vectors [S, R] float32,
    z_mean[S], z_std[S])."""
    from .engine import Engine
    R = pre["ie_M"].shape[-1]
    S = len(enroll_audios)
    base = IvectorSystem(enrolled=np.zeros((1, R), np.float32), z_mean=np.zeros(1), z_std=np.ones(1),
                         num_gselect=num_gselect, min_post=min_post, **pre)
    e = Engine(device)
    try:
        e.load_ivector(base, "CSI")
        e.score_raw([_i16(a) for a in enroll_audios])
        ivs = e.last_ivectors(S, R).astype(np.float32)           # Kaldi writes i-vectors as float32 text/binary
        sysm = base.with_enrolled(ivs, np.zeros(S), np.ones(S))
        e.load_ivector(sysm, "CSI")
        llr, _ = e.score_raw([_i16(a) for a in znorm_audios])     # (n_znorm, S) PLDA log-likelihood ratios
    finally:
        e.close()
    return ivs, np.mean(llr, axis=0).flatten(), np.std(llr, axis=0).flatten()


def _list_wavs(d):
    out = []
    for name in sorted(os.listdir(d)):
        utt = name.split(".")[0]
        out.append((utt.split("-")[0], utt, read(os.path.join(d, name))[1]))
    return out


def build_spk_models(enroll_dir="./data/enrollment-set", z_norm_dir="./data/z-norm-set", pre_model_dir="./pre-models",
                     model_dir="./model", architectures=("iv", "gmm"), device=0):
    """Writes the same artefacts as build_spk_models.py; returns {arch: [pickle lists]}."""
    os.makedirs(model_dir, exist_ok=True)
    enroll = _list_wavs(enroll_dir)
    znorm = [a for (_, _, a) in _list_wavs(z_norm_dir)]
    out = {}
    if "iv" in architectures:
        pre = load_ivector_pre_models(pre_model_dir)
        ivs, zm, zs = enroll_ivector(pre, [a for (_, _, a) in enroll], znorm, device)
        ark = os.path.abspath(os.path.join(model_dir, "ivector.ark"))
        models = []
        with open(ark, "wb") as w:
            for (spk, utt, _), v, m, s in zip(enroll, ivs, zm, zs):
                w.write((utt + " ").encode("ascii"))
                off = w.tell()
                w.write((" [ " + " ".join("%.9g" % x for x in v) + " ]\n").encode("ascii"))
                models.append([spk, utt, "%s:%d" % (ark, off), float(m), float(s)])
        for m in models:
            with open(os.path.join(model_dir, m[0] + ".iv"), "wb") as w:
                pickle.dump(m, w, protocol=-1)
        out["iv"] = models
    if "gmm" in architectures:
        ubm, weights = read_diag_gmm(os.path.join(pre_model_dir, "final.dubm"))
        ids, zm, zs = enroll_gmm(ubm, weights, [a for (_, _, a) in enroll], znorm, device)
        models = []
        for (spk, utt, _), g, m, s in zip(enroll, ids, zm, zs):
            loc = os.path.abspath(os.path.join(model_dir, spk + "-identity.gmm"))
            write_diag_gmm(loc, g, weights, binary=True)
            models.append([spk, utt, loc, float(m), float(s)])
            with open(os.path.join(model_dir, spk + ".gmm"), "wb") as w:
                pickle.dump(models[-1], w, protocol=-1)
        out["gmm"] = models
    return out


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--enroll_dir", default="./data/enrollment-set")
    ap.add_argument("--z_norm_dir", default="./data/z-norm-set")
    ap.add_argument("--pre_model_dir", default="./pre-models")
    ap.add_argument("--model_dir", default="./model")
    ap.add_argument("--architecture", "-archi", nargs="+", default=["iv", "gmm"], choices=["gmm", "iv"])
    a = ap.parse_args()
    for arch, ms in build_spk_models(a.enroll_dir, a.z_norm_dir, a.pre_model_dir, a.model_dir, a.architecture).items():
        for m in ms:
            print(m)
