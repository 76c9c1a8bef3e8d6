#!/usr/bin/env python
"""Captures golden vectors by IMPORTING the reference's own Python in this container
(/root/reference is read-only and does not exist on the GPU box; only the emitted data files
travel).  Run:  python tests/golden/make_golden.py   ->  tests/golden/*.npz / *.json

What can be imported: FAKEBOB.py end-to-end with a synthetic `model`, and the
six model wrappers with an empty pre-models/{utils,steps,sid} tree and a patched
`kaldi_helper.score`.  The Kaldi-side arithmetic cannot run here (parity unpinned there).

G1 loss_fn branches          G2 get_grad (patched np.random.normal)
G3 attack trajectories       G4 estimate_threshold trajectories
G5 wrapper int16 cast        G6 score post-processing (UBM subtract / z-norm / iv re-ordering)
G7 make_decisions            G8 helper text parsing + trial ordering
"""
import contextlib
import io
import json
import os
import pickle
import re
import shutil
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

from synth_model import SynthModel, synth_audio  # noqa: E402

import FAKEBOB as REF_FB  # noqa: E402  (the reference)


class PatchedNormal(object):
    """np.random.normal replacement drawing from a frozen legacy RandomState stream."""

    def __init__(self, seed):
        self.rs = np.random.RandomState(seed)
        self.calls = 0

    def __call__(self, loc=0.0, scale=1.0, size=None):
        self.calls += 1
        return self.rs.normal(loc, scale, size)


@contextlib.contextmanager
def patched_normal(seed):
    old = np.random.normal
    pn = PatchedNormal(seed)
    np.random.normal = pn
    try:
        yield pn
    finally:
        np.random.normal = old


def quiet(fn, *a, **k):
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        r = fn(*a, **k)
    return r, buf.getvalue()


# ----------------------------------------------------------------------- G1
def g1_loss():
    rng = np.random.RandomState(101)
    out = {}
    cases = []
    B, S = 7, 5
    for ci in range(6):
        score = rng.normal(size=(B, S)) * 2.0
        if ci == 1:
            score[2, :] = score[2, 0]          # all tied
            score[3, 1] = score[3, 4]          # pairwise tie
        if ci == 2:
            score = np.round(score * 4) / 4    # many ties
        for task, at, kw in [("OSI", "targeted", dict(target=0)), ("OSI", "targeted", dict(target=S - 1)),
                             ("OSI", "targeted", dict(target=2)), ("OSI", "untargeted", {}),
                             ("CSI", "targeted", dict(target=0)), ("CSI", "targeted", dict(target=S - 1)),
                             ("CSI", "untargeted", dict(true=1)), ("CSI", "untargeted", dict(true=S - 1)),
                             ("SV", "targeted", {})]:
            thr = float(rng.normal() * 0.5)
            kappa = float([0.0, 0.3, -0.2][ci % 3])

            class M(object):
                def score(self_, audios, **k):
                    return score[:, 0].copy() if task == "SV" else score.copy()
            fb = REF_FB.FakeBob(task, at, M(), adver_thresh=kappa)
            fb.threshold = thr
            fb.target = kw.get("target")
            fb.true = kw.get("true")
            loss, sc = fb.loss_fn(np.zeros((4, B)))
            cases.append(dict(task=task, attack=at, target=kw.get("target"), true=kw.get("true"), thr=thr,
                              kappa=kappa, idx=ci))
            out["score_%d" % ci] = score
            out["loss_%d" % (len(cases) - 1)] = loss
    np.savez_compressed(os.path.join(HERE, "g1_loss.npz"), **out)
    return cases


# ----------------------------------------------------------------------- G2
def g2_get_grad():
    out, meta = {}, []
    N = 1600
    for ci, (task, at, spd, kw) in enumerate([
            ("OSI", "targeted", 10, dict(target=1, thr=0.1)),
            ("OSI", "untargeted", 7, dict(thr=0.3)),            # odd spd -> 6 samples
            ("CSI", "targeted", 10, dict(target=3)),
            ("CSI", "untargeted", 16, dict(true=2)),
            ("SV", "targeted", 10, dict(thr=0.2)),
            ("OSI", "targeted", 200, dict(target=0, thr=0.0)),  # >128: numpy pairwise recursion
            ("OSI", "targeted", 2, dict(target=4, thr=0.0))]):
        model = SynthModel(task, 5, N, seed=300 + ci)
        audio = synth_audio(N, 400 + ci)
        fb = REF_FB.FakeBob(task, at, model, adver_thresh=0.05, samples_per_draw=spd, sigma=0.001)
        fb.threshold = kw.get("thr", 0.0)
        fb.target = kw.get("target")
        fb.true = kw.get("true")
        with patched_normal(500 + ci):
            final_loss, grad, adver_loss, score = fb.get_grad(audio)
        meta.append(dict(task=task, attack=at, spd=spd, target=kw.get("target"), true=kw.get("true"),
                         thr=kw.get("thr", 0.0), kappa=0.05, model_seed=300 + ci, audio_seed=400 + ci,
                         noise_seed=500 + ci, N=N))
        out["final_loss_%d" % ci] = np.float64(final_loss)
        out["grad_%d" % ci] = grad
        out["adver_loss_%d" % ci] = np.asarray(adver_loss, np.float64)
        out["score_%d" % ci] = np.asarray(score, np.float64)
    np.savez_compressed(os.path.join(HERE, "g2_get_grad.npz"), **out)
    return meta


# ----------------------------------------------------------------------- G3
def run_attack(tmp, task, at, model, audio, noise_seed, fbkw, atkw):
    fb = REF_FB.FakeBob(task, at, model, **fbkw)
    cp = os.path.join(tmp, "t.cp")
    with patched_normal(noise_seed) as pn:
        (adv, flag), log = quiet(fb.attack, audio, cp, **atkw)
    with open(cp, "rb") as r:
        trace = pickle.load(r)
    lrs = [float(x) for x in re.findall(r"lr:([0-9.]+)", log)]
    return adv, flag, trace, lrs, pn.calls


def g3_attack(tmp):
    out, meta = {}, []
    N = 1600

    def add(name, task, at, model_seed, audio_seed, noise_seed, fbkw, atkw, audio=None, model_thr=0.0):
        model = SynthModel(task, 5, N, seed=model_seed, threshold=model_thr)
        a = synth_audio(N, audio_seed) if audio is None else audio
        adv, flag, trace, lrs, calls = run_attack(tmp, task, at, model, a, noise_seed, fbkw, atkw)
        i = len(meta)
        S = model.S
        T = np.zeros((len(trace), 2 + S))
        for r, row in enumerate(trace):
            assert len(row) == 4                       # [distance, adver_loss, score, used_time]
            T[r, 0] = row[0]
            T[r, 1] = np.asarray(row[1]).reshape(-1)[0]
            T[r, 2:] = np.asarray(row[2], np.float64).reshape(-1)
        out["adv_%d" % i] = adv
        out["trace_%d" % i] = T
        out["lrs_%d" % i] = np.asarray(lrs)
        if audio is not None:
            out["audio_%d" % i] = a
        meta.append(dict(name=name, task=task, attack=at, model_seed=model_seed, audio_seed=audio_seed,
                         noise_seed=noise_seed, fbkw=fbkw, atkw=atkw, flag=int(flag), n_rows=len(trace),
                         n_get_grad=calls, N=N, adv_shape=list(adv.shape), adv_dtype=str(adv.dtype),
                         last_time_is_zero=bool(trace[-1][3] == 0.0), custom_audio=audio is not None))
        return meta[-1], T

    base = dict(epsilon=0.002, max_lr=0.001, min_lr=1e-6, samples_per_draw=10, sigma=0.001, momentum=0.9,
                plateau_length=5, plateau_drop=2.0)
    # (a) succeeds after a few iterations: target = the runner-up speaker of the benign audio
    s_b = SynthModel("OSI", 5, N, seed=600).score(synth_audio(N, 700))
    tgt = int(np.argsort(s_b)[-2])
    m, T = add("osi_targeted_success", "OSI", "targeted", 600, 700, 800,
               dict(base, max_iter=60, adver_thresh=0.0), dict(threshold=-1.0, target=tgt))
    k = m["n_rows"] - 1  # iteration index of the break
    assert m["flag"] == 1 and k >= 1, m
    # (d) the same run with max_iter = k+1: the break happens AT iter == max_iter-1 -> flag -1
    add("osi_success_at_last_iter_reports_failure", "OSI", "targeted", 600, 700, 800,
        dict(base, max_iter=k + 1, adver_thresh=0.0), dict(threshold=-1.0, target=tgt))
    # (b) success at iteration 0 (adver_thresh very negative)
    add("success_at_iter0", "OSI", "targeted", 600, 700, 801,
        dict(base, max_iter=5, adver_thresh=-50.0), dict(threshold=-1.0, target=1))
    # (c) never succeeds
    add("never_succeeds", "OSI", "targeted", 600, 700, 802,
        dict(base, max_iter=12, adver_thresh=50.0), dict(threshold=0.0, target=2))
    # (e) plateau LR halving down to min_lr (tiny epsilon ball: loss stalls quickly)
    add("plateau_to_min_lr", "OSI", "untargeted", 601, 701, 803,
        dict(base, max_iter=30, adver_thresh=50.0, epsilon=0.0002, plateau_length=2, min_lr=2.4e-4),
        dict(threshold=5.0))
    # (f) clipping at +-1 and at the epsilon ball
    hot = synth_audio(N, 702, amp=3000)
    hot[::3] = 32767 / 32768.0
    hot[1::3] = -1.0
    add("clip_at_full_scale", "CSI", "targeted", 602, 702, 804,
        dict(base, max_iter=8, adver_thresh=50.0, max_lr=0.004), dict(target=0), audio=hot)
    # CSI untargeted needs the true label = argmax of the benign scores
    model = SynthModel("CSI", 5, N, seed=603)
    true = int(np.argmax(model.score(synth_audio(N, 703))))
    add("csi_untargeted", "CSI", "untargeted", 603, 703, 805,
        dict(base, max_iter=25, adver_thresh=0.0), dict(true=true))
    add("sv", "SV", "targeted", 604, 704, 806,
        dict(base, max_iter=25, adver_thresh=0.0, samples_per_draw=7), dict(threshold=0.6))
    add("max_iter_1_never_success", "OSI", "targeted", 600, 700, 807,
        dict(base, max_iter=1, adver_thresh=-50.0), dict(threshold=-1.0, target=1))
    np.savez_compressed(os.path.join(HERE, "g3_attack.npz"), **out)
    return meta


# ----------------------------------------------------------------------- G4
def g4_estimate_threshold():
    out, meta = {}, []
    N = 1600
    for ci, (task, model_seed, audio_seed, noise_seed, margin, fbkw) in enumerate([
            ("OSI", 900, 1000, 1100, 0.02, dict(samples_per_draw=10)),
            ("OSI", 901, 1001, 1101, 0.30, dict(samples_per_draw=10, epsilon=0.004)),  # several outer iters
            ("SV", 902, 1002, 1102, 0.05, dict(samples_per_draw=8)),
            ("OSI", 903, 1003, 1103, -0.5, dict(samples_per_draw=10))]):            # accepted immediately
        model = SynthModel(task, 5, N, seed=model_seed)
        audio = synth_audio(N, audio_seed)
        s0 = model.score(audio)
        s0 = float(np.max(s0))
        model.threshold = s0 + margin
        fb = REF_FB.FakeBob(task, "targeted", model, **fbkw)
        with patched_normal(noise_seed) as pn:
            (res, log) = quiet(fb.estimate_threshold, audio)
        score, n_iters, _t = res
        n_outer = int(re.search(r"return at iter_outer:(\d+)", log).group(1))  # iter_outer at return
        meta.append(dict(task=task, model_seed=model_seed, audio_seed=audio_seed, noise_seed=noise_seed,
                         model_threshold=model.threshold, fbkw=fbkw, n_iters=int(n_iters), n_outer=int(n_outer),
                         score=float(score), final_threshold=float(fb.threshold), attack_type_after=fb.attack_type,
                         n_get_grad=pn.calls, N=N))
    # CSI returns None
    fb = REF_FB.FakeBob("CSI", "targeted", SynthModel("CSI", 5, N, seed=1))
    res, _ = quiet(fb.estimate_threshold, synth_audio(N, 1))
    assert res is None
    np.savez_compressed(os.path.join(HERE, "g4_estimate_threshold.npz"), **out) if out else None
    return meta


# --------------------------------------------------------------- G5,G6,G7,G8
def wrappers(tmp):
    """Runs the reference's six wrappers with a stub pre-models tree and a patched helper."""
    pre = os.path.join(tmp, "pre-models")
    for d in ("utils", "steps", "sid", "conf"):
        os.makedirs(os.path.join(pre, d), exist_ok=True)
    import gmm_ubm_OSI, gmm_ubm_CSI, gmm_ubm_SV, ivector_PLDA_OSI, ivector_PLDA_CSI, ivector_PLDA_SV  # noqa
    out, meta = {}, {}
    captured = {}
    rng = np.random.RandomState(7)
    spk_models = [["1580", "1580-utt", "/m/1580.gmm", -71.5, 2.5], ["61", "61-utt", "/m/61.gmm", -70.25, 3.0],
                  ["2830", "2830-utt", "/m/2830.gmm", -69.0, 1.5]]

    def mk(cls, *a, **k):
        cwd = os.getcwd()
        os.chdir(tmp)
        try:
            return cls(*a, **k)
        finally:
            os.chdir(cwd)

    osi = mk(gmm_ubm_OSI.gmm_OSI, os.path.join(tmp, "g-osi"), spk_models, "/m/final.dubm", pre_model_dir=pre, threshold=0.25)
    csi = mk(gmm_ubm_CSI.gmm_CSI, os.path.join(tmp, "g-csi"), spk_models, pre_model_dir=pre)
    sv = mk(gmm_ubm_SV.gmm_SV, os.path.join(tmp, "g-sv"), spk_models[0], "/m/final.dubm", pre_model_dir=pre, threshold=0.25)
    iosi = mk(ivector_PLDA_OSI.iv_OSI, os.path.join(tmp, "i-osi"), spk_models, pre_model_dir=pre, threshold=1.0)
    icsi = mk(ivector_PLDA_CSI.iv_CSI, os.path.join(tmp, "i-csi"), spk_models, pre_model_dir=pre)
    isv = mk(ivector_PLDA_SV.iv_SV, os.path.join(tmp, "i-sv"), spk_models[1], pre_model_dir=pre, threshold=1.0)
    meta["gmm_model_list"] = osi.model_list
    meta["gmm_csi_model_list"] = csi.model_list
    meta["gmm_spk_ids"] = osi.spk_ids
    meta["iv_spk_ids"] = iosi.spk_ids
    meta["iv_utt_ids"] = iosi.utt_ids
    meta["iv_z_means"] = [float(x) for x in iosi.z_norm_means]
    meta["iv_z_stds"] = [float(x) for x in iosi.z_norm_stds]
    with open(iosi.train_ivector_scp) as r:
        meta["iv_train_scp"] = r.read()
    meta["spk_models"] = spk_models

    # ---- G5 quantisation as seen by the helper
    vals = np.array([0.5, -0.5, 0.99999, 1.0, -1.0, 1.00004, 2.7e-5, -2.7e-5, 0.0, 1.5, -1.5, 3.0000305,
                     0.999984741, -0.99998, 1.0 - 2 ** -16, 123.456, -7.25])

    def capture_gmm(model_list, audio_list, **k):
        captured["audio"] = [np.array(a) for a in audio_list]
        captured["kw"] = sorted(k.keys())
        return captured["ret"]

    osi.kaldi_helper.score = capture_gmm
    captured["ret"] = np.zeros((1, 4))
    osi.score(vals)                                   # 1-D ndarray -> ONE utterance
    out["g5_vals"] = vals
    out["g5_q_1d"] = captured["audio"][0]
    meta["g5_helper_kwargs"] = captured["kw"]
    mat = np.stack([vals, -vals, vals * 0.5], axis=1)  # (N,3)
    captured["ret"] = np.zeros((3, 4))
    osi.score(mat)
    out["g5_mat"] = mat
    for i in range(3):
        out["g5_q_mat_%d" % i] = captured["audio"][i]
    lst = [vals[:5].copy(), (vals[5:] * 32768).astype(np.int16), vals[:9].copy()]  # ragged list, one already int16
    lst_copy = [a.copy() for a in lst]
    captured["ret"] = np.zeros((3, 4))
    osi.score(lst)
    for i in range(3):
        out["g5_list_in_%d" % i] = lst_copy[i]
        out["g5_q_list_%d" % i] = captured["audio"][i]
    meta["g5_list_input_unchanged"] = bool(all(np.array_equal(a, b) and a.dtype == b.dtype for a, b in zip(lst, lst_copy)))
    captured["ret"] = np.zeros((1, 4))
    osi.score(vals[:, None])                          # (N,1)
    out["g5_q_col"] = captured["audio"][0].reshape(-1)
    osi.score(vals[None, :])                          # (1,N)
    out["g5_q_row"] = captured["audio"][0].reshape(-1)
    osi.score(vals, bits_per_sample=8)
    out["g5_q_bits8"] = captured["audio"][0]

    # ---- G6/G7 post-processing + decisions
    raw3 = rng.normal(size=(4, 4)) * 0.5 - 70.0        # [ubm, s0, s1, s2]
    raw3[1, 1:] = raw3[1, 0] + np.array([0.25, 0.1, 0.25])   # max == threshold exactly (and tied)
    raw3[2, 1:] = raw3[2, 0] + np.array([0.2499999, 0.1, 0.0])
    out["g6_raw_osi"] = raw3
    captured["ret"] = raw3.copy()
    out["g6_osi_scores"] = osi.score(np.zeros((8, 4)))
    captured["ret"] = raw3.copy()
    dec, sc = osi.make_decisions(np.zeros((8, 4)))
    out["g7_osi_dec"] = np.array(dec)
    out["g7_osi_sc"] = sc
    captured["ret"] = raw3[:1].copy()
    out["g6_osi_scores_b1"] = osi.score(np.zeros(8))
    captured["ret"] = raw3[1:2].copy()
    dec, sc = osi.make_decisions(np.zeros(8))
    meta["g7_osi_b1_dec"] = int(dec)
    meta["g7_osi_b1_dec_type"] = type(dec).__name__
    out["g7_osi_b1_sc"] = sc

    sv.kaldi_helper.score = capture_gmm
    raw2 = raw3[:, :2].copy()
    captured["ret"] = raw2.copy()
    out["g6_raw_sv"] = raw2
    out["g6_sv_scores"] = sv.score(np.zeros((8, 4)))
    captured["ret"] = raw2.copy()
    dec, sc = sv.make_decisions(np.zeros((8, 4)))
    out["g7_sv_dec"] = np.array(dec)
    out["g7_sv_sc"] = sc
    captured["ret"] = raw2[1:2].copy()
    r = sv.score(np.zeros(8))
    meta["g6_sv_b1_type"] = type(r).__name__
    out["g6_sv_b1"] = np.float64(r)
    captured["ret"] = raw2[1:2].copy()
    dec, sc = sv.make_decisions(np.zeros(8))
    meta["g7_sv_b1_dec"] = int(dec)

    csi.kaldi_helper.score = capture_gmm
    rawc = rng.normal(size=(4, 3)) * 2.0 - 70.0
    rawc[3] = [-71.5 + 2.5, -70.25 + 3.0, -69.0 + 1.5]  # z-scores all exactly 1.0 -> argmax tie -> first
    captured["ret"] = rawc.copy()
    out["g6_raw_csi"] = rawc
    out["g6_csi_scores"] = csi.score(np.zeros((8, 4)))
    captured["ret"] = rawc.copy()
    dec, sc = csi.make_decisions(np.zeros((8, 4)))
    out["g7_csi_dec"] = np.array(dec)
    captured["ret"] = rawc[:1].copy()
    dec, sc = csi.make_decisions(np.zeros(8))
    meta["g7_csi_b1_dec"] = int(dec)
    out["g7_csi_b1_sc"] = sc

    # iv wrappers: helper returns scores already in the wrapper's (sorted) speaker order
    def capture_iv(audio_list, train_utt_id, **k):
        captured["audio"] = [np.array(a) for a in audio_list]
        captured["train"] = list(train_utt_id)
        captured["kw"] = dict((kk, (vv if isinstance(vv, (int, str)) else None)) for kk, vv in k.items())
        return captured["ret"]

    iosi.kaldi_helper.score = capture_iv
    rawi = rng.normal(size=(4, 3)) * 5.0
    captured["ret"] = rawi.copy()
    out["g6_raw_iv"] = rawi
    out["g6_iv_osi_scores"] = iosi.score(np.zeros((8, 4)))
    meta["iv_helper_train_utt_ids"] = captured["train"]
    meta["iv_helper_kwargs"] = captured["kw"]
    captured["ret"] = rawi.copy()
    dec, sc = iosi.make_decisions(np.zeros((8, 4)))
    out["g7_iv_osi_dec"] = np.array(dec)
    captured["ret"] = rawi[0].copy()                  # one audio: helper returns (S,)
    dec, sc = iosi.make_decisions(np.zeros(8))
    meta["g7_iv_osi_b1_dec"] = int(dec)
    out["g7_iv_osi_b1_sc"] = sc
    icsi.kaldi_helper.score = capture_iv
    captured["ret"] = rawi.copy()
    out["g6_iv_csi_scores"] = icsi.score(np.zeros((8, 4)))
    captured["ret"] = rawi.copy()
    dec, sc = icsi.make_decisions(np.zeros((8, 4)))
    out["g7_iv_csi_dec"] = np.array(dec)
    isv.kaldi_helper.score = capture_iv
    captured["ret"] = rawi[:, 0].copy()               # one speaker: helper returns (B,)
    out["g6_iv_sv_scores"] = isv.score(np.zeros((8, 4)))
    captured["ret"] = rawi[:, 0].copy()
    dec, sc = isv.make_decisions(np.zeros((8, 4)))
    out["g7_iv_sv_dec"] = np.array(dec)
    captured["ret"] = rawi[:1, 0].copy()
    r = isv.score(np.zeros(8))
    meta["g6_iv_sv_b1_type"] = type(r).__name__
    out["g6_iv_sv_b1"] = np.float64(r)
    meta["g7_iv_sv_value"] = [int(isv.make_decisions_value(v)) for v in (0.99, 1.0, 1.01)]

    # ---- G8 helper parsing / trial ordering (pure file functions of the reference helpers)
    from gmm_ubm_kaldiHelper import gmm_ubm_kaldiHelper
    from ivector_PLDA_kaldiHelper import ivector_PLDA_kaldiHelper
    gh = gmm_ubm_kaldiHelper(pre_model_dir=pre, audio_dir=os.path.join(tmp, "a"), mfcc_dir=os.path.join(tmp, "m"),
                             log_dir=os.path.join(tmp, "l"), score_dir=os.path.join(tmp, "s"))
    os.makedirs(gh.score_dir, exist_ok=True)
    texts = {}
    tab = rng.normal(size=(3, 2)) - 70.0               # 3 utts, 2 models
    for mi in range(2):
        txt = "".join("%05d-1 %.6g\n" % (u + 1, tab[u, mi]) for u in range(3))
        texts["%d.score" % (mi + 1)] = txt
        with open(os.path.join(gh.score_dir, "%d.score" % (mi + 1)), "w") as w:
            w.write(txt)
    out["g8_gmm_resolved"] = gh.resolce_scores(["a", "b"])
    with open(os.path.join(gh.score_dir, "1.score"), "w") as w:   # single utterance
        w.write("00001-1 -71.25\n")
    with open(os.path.join(gh.score_dir, "2.score"), "w") as w:
        w.write("00001-1 -70.5\n")
    out["g8_gmm_resolved_b1"] = gh.resolce_scores(["a", "b"])
    meta["g8_gmm_score_texts"] = texts
    ih = ivector_PLDA_kaldiHelper(pre_model_dir=pre, audio_dir=os.path.join(tmp, "ia"), mfcc_dir=os.path.join(tmp, "im"),
                                  log_dir=os.path.join(tmp, "il"), ivector_dir=os.path.join(tmp, "iv"))
    os.makedirs(ih.ivector_dir, exist_ok=True)
    tf = os.path.join(ih.ivector_dir, "trials")
    ih.write_trials(["1580-utt", "2830-utt", "61-utt"], ["00001-1", "00002-1"], trials_file=tf, flag=1)
    with open(tf) as r:
        meta["g8_trials_text"] = r.read()
    sc_txt = ""
    vals_s = rng.normal(size=6) * 5
    k = 0
    for tr in ["1580-utt", "2830-utt", "61-utt"]:
        for te in ["00001-1", "00002-1"]:
            sc_txt += "%s %s %.6g\n" % (tr, te, vals_s[k]); k += 1
    sf = os.path.join(ih.ivector_dir, "scores")
    with open(sf, "w") as w:
        w.write(sc_txt)
    meta["g8_iv_scores_text"] = sc_txt
    out["g8_iv_resolved"] = ih.resolve_score(sf)       # (2 utts, 3 spk)
    with open(sf, "w") as w:
        w.write("1580-utt 00001-1 1.5\n1580-utt 00002-1 -2.5\n")
    out["g8_iv_resolved_one_spk"] = ih.resolve_score(sf)
    with open(sf, "w") as w:
        w.write("1580-utt 00001-1 1.5\n2830-utt 00001-1 -2.5\n61-utt 00001-1 0.5\n")
    out["g8_iv_resolved_one_utt"] = ih.resolve_score(sf)
    meta["g8_utt_id_scheme"] = [("0000" + str(i + 1))[-5:] + "-1" for i in (0, 8, 9, 99, 12344)]
    np.savez_compressed(os.path.join(HERE, "g5678_wrappers.npz"), **out)
    return meta


def main():
    tmp = tempfile.mkdtemp(prefix="fb_golden_")
    try:
        meta = dict(numpy=np.__version__, reference="/root/reference @ v1",
                    g1=g1_loss(), g2=g2_get_grad(), g3=g3_attack(tmp), g4=g4_estimate_threshold(),
                    g5678=wrappers(tmp))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with open(os.path.join(HERE, "golden_meta.json"), "w") as w:
        json.dump(meta, w, indent=1, sort_keys=True)
    for f in sorted(os.listdir(HERE)):
        print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
