"""Enrolment on the device (fakebob_amd/enroll.py, counterpart of build_spk_models.py) against the CPU oracle's
restatement of `gmm-global-acc-stats` + `MapDiagGmmUpdate`, and end to end on a synthetic site."""
import os
import pickle

import numpy as np
import pytest
from scipy.io.wavfile import write

from fakebob_amd import enroll as EN
from fakebob_amd.engine import Engine
from fakebob_amd.models import stack_models, synthetic_audio, synthetic_gmm_system, synthetic_ivector_system, \
    synthetic_ubm_moments

pytestmark = pytest.mark.gpu


def _wav(utt, n):
    return (synthetic_audio(utt, n) * 32768.0).astype(np.int16)


def test_acc_stats_and_map_update_parity(oracle):
    C, D = 160, 72
    w, _, _ = synthetic_ubm_moments(C, D, 2001)
    ubm, _ = synthetic_gmm_system(1, C, D)
    gc, miv, iv = stack_models([ubm])
    cfg = oracle.default_cfg()
    e = Engine(0)
    try:
        e.load_gmm([ubm])
        for utt, n in [(21, 48000), (22, 160000)]:            # the second one is longer than the CMVN window
            wav = _wav(utt, n)
            occ_g, F_g, tv_g = e.gmm_acc_stats(wav)
            occ_o, F_o, tv_o = oracle.gmm_acc_stats(cfg, wav, gc[0], miv[0], iv[0])
            assert tv_g == tv_o
            assert abs(occ_g.sum() - tv_g) <= 1e-3               # posteriors of a frame sum to one
            assert np.abs(occ_g - occ_o).max() <= 1e-4 * max(1.0, occ_o.max())
            assert np.abs(F_g - F_o).max() <= 1e-4 * max(1.0, np.abs(F_o).max())
            new = EN.map_adapt_means(ubm, w, occ_g, F_g)
            means = ubm.means_invvars.astype(np.float64) * (1.0 / ubm.inv_vars.astype(np.float64))
            ref = oracle.map_update_means(means, occ_o, F_o, tau=10.0)
            got = new.means_invvars.astype(np.float64) / new.inv_vars.astype(np.float64)
            assert np.abs(got - ref).max() <= 1e-5 * max(1.0, np.abs(ref).max())
            assert np.array_equal(new.inv_vars.view(np.uint32), ubm.inv_vars.view(np.uint32))   # variances untouched
            moved = np.abs(got - means).max(axis=1)
            assert moved[occ_o > 1.0].min() > 0 and np.all(moved[occ_o == 0.0] == 0.0)
    finally:
        e.close()
    with pytest.raises(Exception):                                # statistics need the UBM loaded alone
        e2 = Engine(0)
        try:
            u, s = synthetic_gmm_system(2, 64, 72)
            e2.load_gmm([u] + s)
            e2.gmm_acc_stats(_wav(1, 16000))
        finally:
            e2.close()


def test_build_spk_models_gmm_site(tmp_path):
    from fakebob_amd.kaldi_io import read_diag_gmm, write_diag_gmm
    from fakebob_amd.systems import gmm_CSI
    C = 128
    w, _, _ = synthetic_ubm_moments(C, 72, 2001)
    ubm, _ = synthetic_gmm_system(1, C, 72)
    pre = tmp_path / "pre-models"
    (pre / "conf").mkdir(parents=True)
    (pre / "conf" / "mfcc.conf").write_text("--sample-frequency=16000\n--frame-length=25\n--low-freq=20\n--high-freq=7600\n"
                                            "--num-mel-bins=30\n--num-ceps=24\n--snip-edges=false\n")
    (pre / "conf" / "vad.conf").write_text("--vad-energy-threshold=5.5\n--vad-energy-mean-scale=0.5\n"
                                           "--vad-proportion-threshold=0.12\n--vad-frames-context=2\n")
    (pre / "delta_opts").write_text("--delta-window=3 --delta-order=2\n")
    write_diag_gmm(str(pre / "final.dubm"), ubm, w, binary=True)
    (tmp_path / "enroll").mkdir()
    (tmp_path / "znorm").mkdir()
    ids = ["1580", "2830", "61"]
    for i, sid in enumerate(ids):
        write(str(tmp_path / "enroll" / ("%s-enroll.wav" % sid)), 16000, _wav(100 + i, 64000))
    for j in range(6):
        write(str(tmp_path / "znorm" / ("z%d-utt.wav" % j)), 16000, _wav(200 + j, 24000))
    out = EN.build_spk_models(str(tmp_path / "enroll"), str(tmp_path / "znorm"), str(pre), str(tmp_path / "model"),
                              architectures=("gmm",))
    assert [m[0] for m in out["gmm"]] == sorted(ids)
    models = []
    for sid in ids:
        with open(str(tmp_path / "model" / (sid + ".gmm")), "rb") as r:
            m = pickle.load(r)
        assert m[0] == sid and m[1] == sid + "-enroll" and os.path.isabs(m[2]) and m[4] > 0
        g, w2 = read_diag_gmm(m[2])
        assert np.array_equal(g.inv_vars, ubm.inv_vars)            # mean-only adaptation
        assert np.abs(g.means_invvars - ubm.means_invvars).max() > 0
        models.append(m)
    # the z-norm statistics are exactly those of the CSI system's raw scores on the z-norm voices
    csi = gmm_CSI(str(tmp_path / "g"), models, pre_model_dir=str(pre))
    zn = [_wav(200 + j, 24000) for j in range(6)]
    sc = csi.score(zn)
    assert np.abs(sc.mean(axis=0)).max() < 1e-6 and np.abs(sc.std(axis=0) - 1.0).max() < 1e-6
    # MAP adaptation raises the likelihood of the adaptation data: own identity model > UBM
    e = Engine(0)
    try:
        for i, m in enumerate(models):
            e.load_gmm([ubm, read_diag_gmm(m[2])[0]])
            raw, _ = e.score_raw([_wav(100 + i, 64000)])
            assert raw[0, 1] > raw[0, 0]
    finally:
        e.close()


def test_enroll_ivector_identities_and_znorm():
    sy = synthetic_ivector_system(C=96, D=72, R=48, L=24, n_speakers=2, seed=11)
    pre = dict(fg_weights=sy.fg_weights, fg_means_invcovars=sy.fg_means_invcovars, fg_inv_covars=sy.fg_inv_covars,
               ie_M=sy.ie_M, ie_sigma_inv=sy.ie_sigma_inv, prior_offset=sy.prior_offset, mean_vec=sy.mean_vec,
               lda=sy.lda, plda_mean=sy.plda_mean, plda_transform=sy.plda_transform, plda_psi=sy.plda_psi)
    enroll = [_wav(300, 48000), _wav(301, 48000)]
    znorm = [_wav(310 + j, 24000) for j in range(5)]
    ivs, zm, zs = EN.enroll_ivector(pre, enroll, znorm)
    assert ivs.shape == (2, 48) and ivs.dtype == np.float32 and np.all(zs > 0)
    e = Engine(0)
    try:
        e.load_ivector(sy.with_enrolled(ivs, zm, zs), "CSI")
        llr, _ = e.score_raw(znorm)
        sc = e.system_scores(llr)
        assert np.abs(sc.mean(axis=0)).max() < 1e-9 and np.abs(sc.std(axis=0) - 1.0).max() < 1e-9
        llr_e, _ = e.score_raw(enroll)
        for i in range(2):                                            # a voice scores highest against its own i-vector
            assert llr_e[i, i] > llr[:, i].max() and llr_e[i, i] > llr_e[1 - i, i]
    finally:
        e.close()
