"""`make_decisions` on the GPU path against decisions recomputed from ORACLE scores.

The decision rules themselves are pinned bit-exactly by the G7 goldens captured from the reference's wrappers
(tests/test_host_api.py, CPU).  Here the six systems run on the device and their decisions must equal the
reference rule applied to the oracle's scores of the same audio:
  OSI  argmax, -1 when max < threshold   (strict <:  gmm_ubm_OSI.py:101-106, ivector_PLDA_OSI.py:133-137)
  CSI  argmax                            (gmm_ubm_CSI.py:104-106, ivector_PLDA_CSI.py:128-131)
  SV   +1 when score >= threshold else -1 (gmm_ubm_SV.py:86-90, ivector_PLDA_SV.py:94-98)
including a threshold set EXACTLY to a returned score (the tie decides `<` against `<=`) and the
scalar-vs-list return shapes for one utterance."""
import numpy as np
import pytest

from fakebob_amd.models import stack_models, synthetic_audio, synthetic_ivector_system
from fakebob_amd.systems import gmm_CSI, gmm_OSI, gmm_SV, iv_CSI, iv_OSI, iv_SV

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-4
N_UTT = 7


def _batch():
    """(N, B) float audio in [-1, 1): B utterances of 1 s, different speakers / gains."""
    cols = [synthetic_audio(40 + u, 16000) * g for u, g in zip(range(N_UTT), (1.0, 0.5, 0.8, 1.0, 0.3, 0.9, 0.7))]
    return np.stack(cols, axis=1)


def _ref_osi(scores, thr):
    dec = list(np.argmax(scores, axis=1))
    for i, v in enumerate(np.max(scores, axis=1)):
        if v < thr:
            dec[i] = -1
    return [int(d) for d in dec]


def _ref_sv(scores, thr):
    return [1 if s >= thr else -1 for s in scores]


def _separating_threshold(values_a, values_b):
    """A threshold that splits the utterances about in half and is > 10 tolerances away from every score of
    both score sets, so that a 1e-4 score difference cannot move a decision."""
    v = np.sort(np.concatenate([values_a, values_b]))
    gaps = v[1:] - v[:-1]
    mid = len(v) // 2
    order = np.argsort(-gaps)
    for j in order:
        if gaps[j] > 20 * SCORE_TOL and abs(j - mid) <= len(v) // 3:
            return 0.5 * (v[j] + v[j + 1])
    j = int(order[0])
    assert gaps[j] > 20 * SCORE_TOL
    return 0.5 * (v[j] + v[j + 1])


@pytest.fixture(scope="module")
def gmm_setup(oracle, small_system, tmp_path_factory):
    ubm, spk = small_system
    d = tmp_path_factory.mktemp("dec")
    ml = [["spk%d" % i, "utt%d" % i, g, -60.0 - i, 2.0 + 0.5 * i] for i, g in enumerate(spk)]
    return ubm, spk, ml, str(d)


def test_gmm_osi_decisions_match_oracle(engine, oracle, gmm_setup):
    ubm, spk, ml, d = gmm_setup
    gc, miv, iv = stack_models([ubm] + spk)
    ctx = oracle.GmmSystemCtx(oracle.default_cfg(), "OSI", gc, miv, iv, nthreads=8)
    x = _batch()
    so = ctx.score(x)
    model = gmm_OSI(d + "/osi", ml, ubm, pre_model_dir=d, threshold=0.0, engine=engine)
    sg = model.score(x)
    assert np.abs(sg - so).max() <= SCORE_TOL
    model.threshold = _separating_threshold(sg.max(axis=1), so.max(axis=1))
    dec, sc = model.make_decisions(x)
    want = _ref_osi(so, model.threshold)
    assert [int(v) for v in dec] == want and -1 in want and any(w >= 0 for w in want)
    assert np.array_equal(sc, sg)
    # tie: threshold == the returned max score of utterance 2 -> accepted (strict <); one ulp above -> rejected
    i = 2
    model.threshold = float(sg[i].max())
    dec, _ = model.make_decisions(x)
    assert int(dec[i]) == int(np.argmax(so[i])) != -1
    model.threshold = float(np.nextafter(sg[i].max(), np.inf))
    dec, _ = model.make_decisions(x)
    assert int(dec[i]) == -1
    # one utterance: scalar decision, flat scores (gmm_ubm_OSI.py:108-110)
    model.threshold = float(sg[i].max())
    d1, s1 = model.make_decisions(x[:, i])
    assert np.ndim(d1) == 0 and int(d1) == int(np.argmax(so[i])) and np.array_equal(s1, sg[i])


def test_gmm_csi_decisions_match_oracle(engine, oracle, gmm_setup):
    ubm, spk, ml, d = gmm_setup
    gc, miv, iv = stack_models(spk)
    zm = np.array([m[3] for m in ml]); zs = np.array([m[4] for m in ml])
    ctx = oracle.GmmSystemCtx(oracle.default_cfg(), "CSI", gc, miv, iv, zm, zs, nthreads=8)
    x = _batch()
    so = ctx.score(x)
    model = gmm_CSI(d + "/csi", ml, pre_model_dir=d, engine=engine)
    dec, sc = model.make_decisions(x)
    assert np.abs(sc - so).max() <= SCORE_TOL
    top2 = np.sort(so, axis=1)[:, -2:]
    assert np.all(top2[:, 1] - top2[:, 0] > 20 * SCORE_TOL)      # no near-ties: the argmax is well defined
    assert [int(v) for v in dec] == [int(v) for v in np.argmax(so, axis=1)]
    d1, s1 = model.make_decisions(x[:, 0])
    assert np.ndim(d1) == 0 and int(d1) == int(np.argmax(so[0])) and s1.shape == (3,)


def test_gmm_sv_decisions_match_oracle(engine, oracle, gmm_setup):
    ubm, spk, ml, d = gmm_setup
    gc, miv, iv = stack_models([ubm, spk[0]])
    ctx = oracle.GmmSystemCtx(oracle.default_cfg(), "SV", gc, miv, iv, nthreads=8)
    x = _batch()
    so = ctx.score(x)[:, 0]
    model = gmm_SV(d + "/sv", ml[0], ubm, pre_model_dir=d, threshold=0.0, engine=engine)
    sg = model.score(x)
    assert sg.shape == (N_UTT,) and np.abs(sg - so).max() <= SCORE_TOL
    model.threshold = _separating_threshold(sg, so)
    dec, sc = model.make_decisions(x)
    want = _ref_sv(so, model.threshold)
    assert list(dec) == want and 1 in want and -1 in want
    i = 4
    model.threshold = float(sg[i])                                # tie: >= accepts (gmm_ubm_SV.py:87)
    assert model.make_decisions(x)[0][i] == 1
    d1, s1 = model.make_decisions(x[:, i])
    assert d1 == 1 and np.ndim(s1) == 0 and float(s1) == float(sg[i])
    model.threshold = float(np.nextafter(sg[i], np.inf))
    assert model.make_decisions(x)[0][i] == -1 and model.make_decisions(x[:, i])[0] == -1


@pytest.fixture(scope="module")
def iv_setup(oracle, tmp_path_factory):
    sy = synthetic_ivector_system(C=96, D=72, R=48, L=24, n_speakers=3, seed=11)
    # spk ids chosen so that the wrappers' sort-by-string re-ordering is NOT the identity
    ids = ["2830", "61", "1580"]
    zm = [-30.0, -50.0, -20.0]
    zs = [5.0, 8.0, 4.0]
    ml = [[ids[i], "utt%d" % i, sy.enrolled[i].copy(), zm[i], zs[i]] for i in range(3)]
    order = [ids.index(s) for s in sorted(ids)]                  # ivector_PLDA_OSI.py:65-82
    sy_sorted = sy.with_enrolled(sy.enrolled[order], [zm[i] for i in order], [zs[i] for i in order])
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy_sorted, nthreads=8)
    return sy, ml, order, ctx, str(tmp_path_factory.mktemp("ivdec"))


def test_iv_osi_and_csi_decisions_match_oracle(engine, oracle, iv_setup):
    sy, ml, order, ctx, d = iv_setup
    x = _batch()
    so = ctx.score(x)                                            # z-normed, sorted-speaker order
    model = iv_OSI(d + "/osi", ml, pre_model_dir=d, threshold=0.0, engine=engine, system=sy)
    assert model.spk_ids == sorted(m[0] for m in ml)
    sg = model.score(x)
    assert np.abs(sg - so).max() <= SCORE_TOL
    model.threshold = _separating_threshold(sg.max(axis=1), so.max(axis=1))
    dec, sc = model.make_decisions(x)
    want = _ref_osi(so, model.threshold)
    assert [int(v) for v in dec] == want and -1 in want and any(w >= 0 for w in want)
    i = 3
    model.threshold = float(sg[i].max())
    assert int(model.make_decisions(x)[0][i]) == int(np.argmax(so[i]))
    d1, s1 = model.make_decisions(x[:, i])
    assert np.ndim(d1) == 0 and int(d1) == int(np.argmax(so[i])) and np.array_equal(s1, sg[i])
    model.threshold = float(np.nextafter(sg[i].max(), np.inf))
    assert int(model.make_decisions(x)[0][i]) == -1
    csi = iv_CSI(d + "/csi", ml, pre_model_dir=d, engine=engine, system=sy)
    dec, sc = csi.make_decisions(x)
    assert np.abs(sc - so).max() <= SCORE_TOL
    top2 = np.sort(so, axis=1)[:, -2:]
    assert np.all(top2[:, 1] - top2[:, 0] > 20 * SCORE_TOL)
    assert [int(v) for v in dec] == [int(v) for v in np.argmax(so, axis=1)]


def test_iv_sv_decisions_match_oracle(engine, oracle, iv_setup):
    sy, ml, order, ctx_all, d = iv_setup
    m = ml[1]
    sv_sys = sy.with_enrolled(sy.enrolled[1:2], [m[3]], [m[4]])
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sv_sys, nthreads=8)
    x = _batch()
    so = ctx.score(x)[:, 0]
    model = iv_SV(d + "/sv", m, pre_model_dir=d, threshold=0.0, engine=engine, system=sy)
    sg = model.score(x)
    assert sg.shape == (N_UTT,) and np.abs(sg - so).max() <= SCORE_TOL
    model.threshold = _separating_threshold(sg, so)
    dec, _ = model.make_decisions(x)
    want = _ref_sv(so, model.threshold)
    assert list(dec) == want and 1 in want and -1 in want
    i = 5
    model.threshold = float(sg[i])                                # tie: >= accepts (ivector_PLDA_SV.py:95)
    assert model.make_decisions(x)[0][i] == 1 and model.make_decisions(x[:, i])[0] == 1
    assert model.make_decisions_value(float(sg[i])) == 1
    model.threshold = float(np.nextafter(sg[i], np.inf))
    assert model.make_decisions(x)[0][i] == -1 and model.make_decisions_value(float(sg[i])) == -1
