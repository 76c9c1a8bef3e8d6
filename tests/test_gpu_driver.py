"""End-to-end run of the driver counterpart (fakebob_amd/attack_main.py) on a synthetic site laid out
like the reference expects: Kaldi-format model files, pickled speaker models, pre-models/conf, wav
directories (attackMain.py:31-36, 87-272)."""
import os
import pickle

import numpy as np
import pytest
from scipy.io.wavfile import read, write

from fakebob_amd.models import synthetic_audio, synthetic_gmm_system, synthetic_ubm_moments

pytestmark = pytest.mark.gpu


def _site(tmp_path, n_spk=3, C=128):
    from fakebob_amd.kaldi_io import write_diag_gmm
    w, _, _ = synthetic_ubm_moments(C, 72, 2001)
    ubm, spk = synthetic_gmm_system(n_spk, C, 72)
    pre = tmp_path / "pre-models"
    (pre / "conf").mkdir(parents=True)
    (pre / "conf" / "mfcc.conf").write_text("--sample-frequency=16000\n--frame-length=25\n--low-freq=20\n--high-freq=7600\n"
                                            "--num-mel-bins=30\n--num-ceps=24\n--snip-edges=false\n")
    (pre / "conf" / "vad.conf").write_text("--vad-energy-threshold=5.5\n--vad-energy-mean-scale=0.5\n"
                                           "--vad-proportion-threshold=0.12\n--vad-frames-context=2\n")
    (pre / "delta_opts").write_text("--delta-window=3 --delta-order=2\n")
    write_diag_gmm(str(pre / "final.dubm"), ubm, w, binary=True)
    (tmp_path / "model").mkdir()
    ids = ["1580", "2830", "61"][:n_spk]
    for i, (sid, g) in enumerate(zip(ids, spk)):
        p = str(tmp_path / "model" / (sid + "-identity.gmm"))
        write_diag_gmm(p, g, w, binary=(i % 2 == 0))          # both Kaldi encodings
        with open(str(tmp_path / "model" / (sid + ".gmm")), "wb") as f:
            pickle.dump([sid, sid + "-enroll", p, -170.0 - i, 0.05 + 0.01 * i], f)
    for sub, spks in (("illegal-set", ["9001", "9002"]), ("test-set", ids)):
        for j, s in enumerate(spks):
            d = tmp_path / "data" / sub / s
            d.mkdir(parents=True)
            for u in range(2):
                a = (synthetic_audio(10 * j + u + (0 if sub == "illegal-set" else 50), 16000) * 32768).astype(np.int16)
                write(str(d / ("%s-utt%d.wav" % (s, u))), 16000, a)
    return ids, ubm, spk


def test_osi_targeted_site_run(tmp_path, capsys):
    from fakebob_amd import attack_main as AM
    from fakebob_amd.systems import gmm_OSI
    ids, ubm, spk = _site(tmp_path)
    ml = AM.load_spk_models(str(tmp_path / "model"), ids, "gmm")
    probe = gmm_OSI(str(tmp_path / "probe"), ml, str(tmp_path / "pre-models" / "final.dubm"),
                    pre_model_dir=str(tmp_path / "pre-models"), threshold=0.0)
    voices = AM.collect_voices(str(tmp_path / "data" / "illegal-set"))
    sc = probe.score([v[2] for v in voices])
    thr = float(sc.max()) + 0.002                       # every illegal voice is (just) rejected
    argv = ["-spk_id"] + ids + ["-archi", "gmm", "-task", "OSI", "-type", "targeted", "-thresh", str(thr),
            "-max_iter", "25", "-samples", "10", "--streams", "2", "--seed", "7",
            "--model_dir", str(tmp_path / "model"), "--pre_model_dir", str(tmp_path / "pre-models"),
            "--test_dir", str(tmp_path / "data" / "test-set"), "--illegal_dir", str(tmp_path / "data" / "illegal-set"),
            "--out_dir", str(tmp_path / "out")]
    np.random.seed(3)
    g, results, thr_est = AM.main(argv)
    out = capsys.readouterr().out
    assert "load data done, total num: 12" in out       # 4 rejected voices x 3 targets
    assert "attack successful rate" in out
    assert g[1] == 12 and len(results) == 12 and g[2] > 0 and g[3] >= 11 * g[2]
    assert thr_est >= float(sc.max())                   # the estimate lies at/above the benign scores
    base = tmp_path / "out" / "adversarial-audio" / "gmm-OSI-targeted"
    n = 0
    for (spk_id, name, audio) in voices:
        for t in range(3):
            wav = base / spk_id / (name.split(".")[0] + "_%d.wav" % t)
            cp = tmp_path / "out" / "checkpoint" / "gmm-OSI-targeted" / spk_id / (name.split(".")[0] + "_%d.cp" % t)
            assert wav.exists() and cp.exists()
            _, adv = read(str(wav))
            assert adv.dtype == np.int16 and adv.shape == (16000,)
            assert np.abs(adv / 32768.0 - audio).max() <= 0.002 + 1.0 / 32768
            with open(str(cp), "rb") as f:
                trace = pickle.load(f)
            assert 1 <= len(trace) <= 25 and len(trace[0]) == 4
            n += 1
    assert n == 12


def test_csi_site_filtering(tmp_path):
    """CSI keeps only correctly classified voices and expands the other speakers as targets."""
    from fakebob_amd import attack_main as AM
    ids, ubm, spk = _site(tmp_path)
    ml = AM.load_spk_models(str(tmp_path / "model"), ids, "gmm")
    model = AM.make_model("gmm", "CSI", ml, str(tmp_path / "pre-models"), 0.0, str(tmp_path / "g"))
    items = AM.build_attack_list("CSI", "targeted", model, str(tmp_path / "data" / "test-set"),
                                 str(tmp_path / "data" / "illegal-set"), str(tmp_path / "a"), str(tmp_path / "c"))
    voices = AM.collect_voices(str(tmp_path / "data" / "test-set"))
    dec, _ = model.make_decisions([v[2] for v in voices])
    n_ok = sum(1 for (s, _, _), d in zip(voices, dec) if ids.index(s) == d)
    assert len(items) == n_ok * 2
    assert all(it["target"] != it["true"] for it in items)
    un = AM.build_attack_list("CSI", "untargeted", model, str(tmp_path / "data" / "test-set"),
                              str(tmp_path / "data" / "illegal-set"), str(tmp_path / "a"), str(tmp_path / "c"))
    assert len(un) == n_ok and all(it["target"] is None for it in un)


def test_evaluate_counterpart_on_site(tmp_path):
    """fakebob_amd.evaluate (test.py counterpart): the three tasks of the GMM architecture on the synthetic site."""
    from fakebob_amd import attack_main as AM
    from fakebob_amd import evaluate as E
    ids, ubm, spk = _site(tmp_path)
    ml = AM.load_spk_models(str(tmp_path / "model"), ids, "gmm")
    r = E.evaluate("gmm", ml, str(tmp_path / "pre-models"), str(tmp_path / "data" / "test-set"),
                   str(tmp_path / "data" / "illegal-set"), group_prefix=str(tmp_path / "ev"))
    assert set(r) == {"CSI", "SV", "OSI"}
    assert 0.0 <= r["CSI"]["accuracy"] <= 100.0
    for k in ("FRR", "FAR"):
        assert 0.0 <= r["SV"][k] <= 100.0 and 0.0 <= r["OSI"][k] <= 100.0
    assert 0.0 <= r["OSI"]["IER"] <= 100.0 and np.isfinite(r["SV"]["threshold"]) and np.isfinite(r["OSI"]["threshold"])


def _oracle_site_scores(oracle, tmp_path, ids):
    """Oracle scores of the site's voices with the SAME model files the product reads (the format readers are not
    arithmetic): raw[b][m] per task -> the wrappers' post-processing in numpy."""
    from fakebob_amd import attack_main as AM
    from fakebob_amd import evaluate as E
    from fakebob_amd.kaldi_io import load_gmm_any
    from fakebob_amd.models import stack_models
    ml = AM.load_spk_models(str(tmp_path / "model"), ids, "gmm")
    ubm = load_gmm_any(str(tmp_path / "pre-models" / "final.dubm"))
    spk = [load_gmm_any(m[2]) for m in ml]
    cfg = oracle.default_cfg()

    def raw(models, wavs):
        gc, miv, iv = stack_models(models)
        return oracle.gmm_score_batch(cfg, [np.asarray(w, np.int16) for w in wavs], gc, miv, iv, nthreads=8)[0]
    imp = E.load_impostors(str(tmp_path / "data" / "illegal-set"))
    audios, labels = E.load_trials(str(tmp_path / "data" / "test-set"), ids)
    zm = np.array([m[3] for m in ml]); zs = np.array([m[4] for m in ml])
    return ml, ubm, spk, raw, imp, audios, labels, zm, zs


def test_evaluate_counterpart_equals_oracle_derived_numbers(tmp_path, oracle):
    """test.py counterpart on the GPU against the same (golden-pinned) formulas applied to ORACLE scores: accuracy,
    FRR, FAR, IER are counts and must be identical; thresholds are scores (<= 1e-4)."""
    from fakebob_amd import evaluate as E
    ids, _, _ = _site(tmp_path)
    ml, ubm, spk, raw, imp, audios, labels, zm, zs = _oracle_site_scores(oracle, tmp_path, ids)
    r = E.evaluate("gmm", ml, str(tmp_path / "pre-models"), str(tmp_path / "data" / "test-set"),
                   str(tmp_path / "data" / "illegal-set"), group_prefix=str(tmp_path / "ev"))
    csi = (raw(spk, audios) - zm) / zs
    assert r["CSI"]["accuracy"] == E.csi_accuracy(np.argmax(csi, axis=1), labels)
    st, su = [], []
    for i, sid in enumerate(ids):
        own = E._read_dir(str(tmp_path / "data" / "test-set" / sid))
        ro, ri = raw([ubm, spk[i]], own), raw([ubm, spk[i]], imp)
        st += list(ro[:, 1] - ro[:, 0])
        su += list(ri[:, 1] - ri[:, 0])
    thr, frr, far = E.set_threshold(st, su)
    assert r["SV"]["FRR"] == frr and r["SV"]["FAR"] == far and abs(r["SV"]["threshold"] - thr) <= 1e-4
    ro, ri = raw([ubm] + spk, audios), raw([ubm] + spk, imp)
    thr, frr, ier, far = E.osi_metrics(ro[:, 1:] - ro[:, :1], labels, ri[:, 1:] - ri[:, :1])
    assert (r["OSI"]["FRR"], r["OSI"]["IER"], r["OSI"]["FAR"]) == (frr, ier, far)
    assert abs(r["OSI"]["threshold"] - thr) <= 1e-4


def test_iv_architecture_site_run(tmp_path, capsys, oracle):
    """`-archi iv` end to end on a synthetic site: Kaldi-format final.ubm / final.ie / mean.vec / transform.mat /
    plda, `.iv` speaker pickles whose identity_location points into a text ark (build_spk_models.py:146-150), SV
    task with the threshold estimated on rank 0 (attackMain.py:393-394)."""
    from fakebob_amd import attack_main as AM
    from fakebob_amd.kaldi_io import write_ivector_pre_models
    from fakebob_amd.models import synthetic_ivector_system
    sy = synthetic_ivector_system(C=64, D=72, R=32, L=16, n_speakers=2, seed=4)
    pre = tmp_path / "pre-models"
    write_ivector_pre_models(str(pre), sy)
    (pre / "conf").mkdir()
    (pre / "conf" / "mfcc.conf").write_text("--sample-frequency=16000\n--frame-length=25\n--low-freq=20\n--high-freq=7600\n"
                                            "--num-mel-bins=30\n--num-ceps=24\n--snip-edges=false\n")
    (pre / "conf" / "vad.conf").write_text("--vad-energy-threshold=5.5\n--vad-energy-mean-scale=0.5\n"
                                           "--vad-proportion-threshold=0.12\n--vad-frames-context=2\n")
    (pre / "delta_opts").write_text("--delta-window=3 --delta-order=2\n")
    (tmp_path / "model").mkdir()
    ids = ["1580", "61"]
    ark = tmp_path / "model" / "ivector.1.ark"
    offs, txt = [], ""
    for i, sid in enumerate(ids):
        head = "%s-enroll  " % sid
        offs.append(len(txt) + len(head))
        txt += head + "[ " + " ".join("%.9g" % v for v in sy.enrolled[i]) + " ]\n"
    ark.write_text(txt)
    for i, sid in enumerate(ids):
        with open(str(tmp_path / "model" / (sid + ".iv")), "wb") as f:
            pickle.dump([sid, sid + "-enroll", "%s:%d" % (ark, offs[i]), -30.0 - i, 6.0 + i], f)
    for j, s in enumerate(["9001", "9002"]):
        d = tmp_path / "data" / "illegal-set" / s
        d.mkdir(parents=True)
        for u in range(2):
            write(str(d / ("%s-utt%d.wav" % (s, u))), 16000, (synthetic_audio(10 * j + u, 16000) * 32768).astype(np.int16))
    (tmp_path / "data" / "test-set").mkdir()
    # the benign scores of the illegal voices, from the ORACLE, place the system threshold just above them
    voices = AM.collect_voices(str(tmp_path / "data" / "illegal-set"))
    sv_sys = sy.with_enrolled(sy.enrolled[:1], [-30.0], [6.0])
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sv_sys, nthreads=8)
    so = ctx.score(np.stack([v[2] for v in voices], axis=1))[:, 0]
    thr = float(so.max()) + 0.01
    argv = ["-spk_id"] + ids + ["-archi", "iv", "-task", "SV", "-thresh", str(thr), "-max_iter", "12", "-samples", "10",
            "--streams", "2", "--seed", "11", "--model_dir", str(tmp_path / "model"), "--pre_model_dir", str(pre),
            "--test_dir", str(tmp_path / "data" / "test-set"), "--illegal_dir", str(tmp_path / "data" / "illegal-set"),
            "--out_dir", str(tmp_path / "out")]
    np.random.seed(5)
    g, results, thr_est = AM.main(argv)
    out = capsys.readouterr().out
    assert "load data done, total num: 4" in out and "attack successful rate" in out   # all 4 voices rejected -> attacked
    assert g[1] == 4 and len(results) == 4 and g[2] > 0
    assert thr_est >= thr - 1e-4                      # estimate_threshold returns an accepted score (FAKEBOB.py:96-103)
    base = tmp_path / "out" / "adversarial-audio" / "iv-SV-targeted" / "1580"
    for (spk_id, name, audio) in voices:
        _, adv = read(str(base / spk_id / name))
        assert adv.dtype == np.int16 and np.abs(adv / 32768.0 - audio).max() <= 0.002 + 1.0 / 32768
        assert (tmp_path / "out" / "checkpoint" / "iv-SV-targeted" / "1580" / spk_id / (name.split(".")[0] + ".cp")).exists()
