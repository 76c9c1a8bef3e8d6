"""Baseline evaluation of the six systems -- counterpart of the reference's test.py.

test.py is a script: it scores the test-set / illegal-set with every system one wrapper call at a time
and prints accuracy (CSI), threshold / FRR / FAR (SV) and threshold / FRR / IER / FAR (OSI).  Here the
same numbers come from functions; the scoring itself is the batched GPU path behind `make_decisions`.

    python -m fakebob_amd.evaluate -spk_id 1580 2830 4446 5142 61 [--model_dir model ...]

Formulas follow test.py: `set_threshold` (:46-71) scans the *target* scores as candidate thresholds and
keeps the first one minimising |FRR - FAR| (FRR: target < thr, FAR: impostor >= thr, both in percent);
OSI keeps only correctly identified target trials for the target scores (:262-266, :312-316) and takes the
maximum score of every impostor trial (:268-270); IER follows the reference's expression literally
(:273-275), including its broadcast over `target_scores[:, max_spk_index]`.
"""
import argparse
import os
import pickle

import numpy as np
from scipy.io.wavfile import read


def set_threshold(score_target, score_untarget):
    """(threshold, FRR %, FAR %) -- test.py:46-71."""
    st = np.asarray(score_target, np.float64).reshape(-1)
    su = np.asarray(score_untarget, np.float64).reshape(-1)
    best = (0.0, 0.0, 0.0)
    best_diff = np.inf
    for cand in st:
        frr = np.count_nonzero(st < cand) * 100 / st.size
        far = np.count_nonzero(su >= cand) * 100 / su.size
        diff = abs(frr - far)
        if diff < best_diff:            # strict: the first minimiser wins
            best, best_diff = (cand, frr, far), diff
    return best


def csi_accuracy(decisions, labels):
    """percent of test trials whose arg-max speaker is the true one (test.py:95-99, 127-131)."""
    d = np.asarray(decisions).reshape(-1)
    t = np.asarray(labels).reshape(-1)
    return np.count_nonzero(d == t) * 100 / d.size


def osi_metrics(target_scores, labels, untarget_scores):
    """(threshold, FRR %, IER %, FAR %) -- test.py:258-277 / 308-327."""
    ts = np.atleast_2d(np.asarray(target_scores, np.float64))
    us = np.atleast_2d(np.asarray(untarget_scores, np.float64))
    labels = np.asarray(labels).reshape(-1)
    max_idx = np.argmax(ts, axis=1)
    keep = np.flatnonzero(max_idx == labels)
    thr, frr, far = set_threshold(np.max(ts[keep], axis=1), np.max(us, axis=1))
    # literal: rows/cols of the (n, n) comparison flattened together, intersected with the mis-identified rows
    accepted = np.argwhere(ts[:, max_idx] >= thr).flatten()
    wrong = np.argwhere(max_idx != labels).flatten()
    ier = np.intersect1d(accepted, wrong).size * 100 / ts.shape[0]
    return thr, frr, ier, far


# ---------------------------------------------------------------------------------------- data + systems
def _read_dir(spk_dir):
    return [read(os.path.join(spk_dir, n))[1] for n in sorted(os.listdir(spk_dir))]


def load_trials(test_dir, spk_ids):
    """(audios, labels) of the enrolled speakers' test voices; label = index in spk_ids."""
    audios, labels = [], []
    spk_ids = list(spk_ids)
    for spk in sorted(os.listdir(test_dir)):
        if spk not in spk_ids:
            continue
        for a in _read_dir(os.path.join(test_dir, spk)):
            audios.append(a)
            labels.append(spk_ids.index(spk))
    return audios, np.asarray(labels)


def load_impostors(illegal_dir):
    out = []
    for spk in sorted(os.listdir(illegal_dir)):
        out += _read_dir(os.path.join(illegal_dir, spk))
    return out


def evaluate(architecture, model_list, pre_model_dir, test_dir, illegal_dir, group_prefix="test"):
    """Runs the three tasks of one architecture ('gmm' | 'iv'); returns a dict of the printed numbers."""
    from .systems import gmm_CSI, gmm_OSI, gmm_SV, iv_CSI, iv_OSI, iv_SV
    ubm = os.path.join(pre_model_dir, "final.dubm")
    res = {}
    if architecture == "iv":
        csi = iv_CSI(group_prefix + "-iv-CSI", model_list, pre_model_dir=pre_model_dir)
        osi = iv_OSI(group_prefix + "-iv-OSI", model_list, pre_model_dir=pre_model_dir)
        sv = lambda m: iv_SV(group_prefix + "-iv-SV-" + m[0], m, pre_model_dir=pre_model_dir)   # noqa: E731
    else:
        csi = gmm_CSI(group_prefix + "-gmm-CSI", model_list, pre_model_dir=pre_model_dir)
        osi = gmm_OSI(group_prefix + "-gmm-OSI", model_list, ubm, pre_model_dir=pre_model_dir)
        sv = lambda m: gmm_SV(group_prefix + "-gmm-SV-" + m[0], m, ubm, pre_model_dir=pre_model_dir)   # noqa: E731
    impostors = load_impostors(illegal_dir)
    # CSI
    audios, labels = load_trials(test_dir, csi.spk_ids)
    dec, _ = csi.make_decisions(audios)
    res["CSI"] = dict(accuracy=csi_accuracy(np.atleast_1d(dec), labels))
    # SV: every enrolled speaker against its own test voices and all impostors
    st, su = [], []
    for m in model_list:
        model = sv(m)
        own = _read_dir(os.path.join(test_dir, m[0]))
        st += list(np.atleast_1d(model.make_decisions(own)[1]))
        su += list(np.atleast_1d(model.make_decisions(impostors)[1]))
    thr, frr, far = set_threshold(st, su)
    res["SV"] = dict(threshold=thr, FRR=frr, FAR=far)
    # OSI
    audios, labels = load_trials(test_dir, osi.spk_ids)
    _, ts = osi.make_decisions(audios)
    _, us = osi.make_decisions(impostors)
    thr, frr, ier, far = osi_metrics(ts, labels, us)
    res["OSI"] = dict(threshold=thr, FRR=frr, IER=ier, FAR=far)
    return res


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--speaker_id", "-spk_id", nargs="+", required=True)
    ap.add_argument("--architecture", "-archi", nargs="+", default=["iv", "gmm"], choices=["gmm", "iv"])
    ap.add_argument("--model_dir", default="./model")
    ap.add_argument("--pre_model_dir", default="pre-models")
    ap.add_argument("--test_dir", default="./data/test-set")
    ap.add_argument("--illegal_dir", default="./data/illegal-set")
    args = ap.parse_args(argv)
    out = {}
    for archi in args.architecture:
        ext = ".iv" if archi == "iv" else ".gmm"
        models = []
        for spk in args.speaker_id:
            with open(os.path.join(args.model_dir, spk + ext), "rb") as r:
                models.append(pickle.load(r))
        r = evaluate(archi, models, args.pre_model_dir, args.test_dir, args.illegal_dir)
        name = "ivector-PLDA" if archi == "iv" else "gmm-ubm"
        print("----- Test of %s-based CSI, result ---> Accuracy:%f ----- " % (name, r["CSI"]["accuracy"]))
        print("----- Test of %s-based SV, result ---> threshold: %f FRR: %f, FAR: %f"
              % (name, r["SV"]["threshold"], r["SV"]["FRR"], r["SV"]["FAR"]))
        print("----- Test of %s-based OSI, result ---> threshold: %f, FRR: %f, IER: %f, FAR: %f -----"
              % (name, r["OSI"]["threshold"], r["OSI"]["FRR"], r["OSI"]["IER"], r["OSI"]["FAR"]))
        out[archi] = r
    return out


if __name__ == "__main__":
    main()
