"""The headline configuration (BASELINE.json configs[1]: GMM-UBM OSI, UBM + 5 speakers, C = 2048, D = 72, spd = 50,
3 s @ 16 kHz) through the whole NES path against the CPU oracle at FULL size: one get_grad and a 5-iteration attack
(FAKEBOB.py:139-246) -- not only plain scoring.  Run twice: with the engine's defaults, and with the two file round
trips of the reference's real pipeline switched on (`copy-feats --compress=true` of steps/make_mfcc.sh,
gmm_ubm_kaldiHelper.py:138-140; 6-significant-digit score text, gmm_ubm_kaldiHelper.py:236-248)."""
import os

import numpy as np
import pytest

from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import stack_models, synthetic_audio

pytestmark = pytest.mark.gpu
NTHR = max(1, len(os.sched_getaffinity(0)))
KW = dict(samples_per_draw=50, epsilon=0.002, sigma=0.001, max_lr=0.001, min_lr=1e-6, momentum=0.9,
          plateau_length=5, plateau_drop=2.0, adver_thresh=0.0, target=0, threshold=0.2277)


def _pair(oracle, full_system, over):
    ubm, spk = full_system
    models = [ubm] + spk
    e = Engine(0)
    e.set_frontend(**over)
    e.load_gmm(models)
    e.set_system("OSI")
    gc, miv, iv = stack_models(models)
    ctx = oracle.GmmSystemCtx(oracle.default_cfg(**over), "OSI", gc, miv, iv, nthreads=NTHR)
    return e, ctx


@pytest.mark.parametrize("delta_p", ["", "3", "6"])
def test_config1_get_grad_and_attack_at_full_size(oracle, full_system, monkeypatch, delta_p):
    """What can and cannot be promised at N = 48 000, spd = 50.  Scores, losses and the gradient estimate agree with
    the float64-accumulating oracle to ~3e-6 / 1e-3 (gradient rms 0.22).  The update is sign(momentum gradient)
    (FAKEBOB.py:202): of 48 000 entries a dozen lie within that 1e-3 of zero, so ANY two float32 evaluations of the
    GMM (this kernel with 2 or 3 products, Kaldi's sgemv order, the oracle's float64 sums) step those samples in
    opposite directions, and the iteration amplifies it -- 12 samples after one update, ~500 after two, a quarter
    after five, measured identically for delta_p = 1, 2 and for the round-2 arithmetic (delta_p = 3).  Asserted here:
    the first iteration to score tolerance, sign flips only where the gradient is indistinguishable from zero, the
    same success flag and row count, later rows at the 1e-2 level, the perturbation inside the same epsilon ball.
    (Bit-identical trajectories are asserted where they are attainable: the 1 s / spd = 10 cases of
    test_gpu_parity.py and the injected-model goldens of test_gpu_plugin_api.py.)"""
    if delta_p:
        monkeypatch.setenv("FB_GMM_DELTA_P", delta_p)
    else:
        monkeypatch.delenv("FB_GMM_DELTA_P", raising=False)
    e, ctx = _pair(oracle, full_system, {})
    try:
        assert e.gmm_kernel_variant == "fx2w/%s" % (delta_p or "1")      # "fx2w/1": the kernel bench.py times
        audio = synthetic_audio(0, 48000)
        pg = nes_params("OSI", "targeted", seed=42, stream=0, max_iter=1000, **KW)
        po = oracle.nes_params("OSI", "targeted", ctx.S, max_iter=1000, **KW)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=0)
        flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=42, it=0, stream=0)
        assert abs(alg - alo) <= 1e-4 and abs(flg - flo) <= 1e-4
        assert np.abs(scg[:ctx.S] - sco).max() <= 1e-4
        rms = float(np.sqrt(np.mean(go * go)))
        flips = np.sign(gg) != np.sign(go)
        print("get_grad: score err %.2e, grad err %.2e (rms %.3f), %d sign flips, largest |g| among them %.2e" %
              (np.abs(scg[:ctx.S] - sco).max(), np.abs(gg - go).max(), rms, int(flips.sum()),
               np.abs(go[flips]).max() if flips.any() else 0.0))
        assert np.abs(gg - go).max() <= 0.02 * rms
        assert flips.mean() <= 1e-3 and (not flips.any() or np.abs(go[flips]).max() <= 0.02 * rms)
        pg.max_iter = po.max_iter = 5
        adv_g, flag_g, advf_g, tr_g = e.attack(pg, audio)
        adv_o, flag_o, advf_o, tr_o = oracle.attack(po, ctx.fn, ctx.ctx, audio, seed=42, stream=0)
        assert flag_g == flag_o and tr_g.shape == tr_o.shape == (5, 3 + ctx.S)
        rows = np.abs(tr_g - tr_o).max(axis=1)
        print("attack: |trace diff| per row", rows, "differing int16 samples", int(np.sum(adv_g != adv_o)))
        assert rows[0] <= 1e-4 and rows[1] <= 1e-3 and rows.max() <= 1e-2
        assert np.array_equal(tr_g[:, 2], tr_o[:, 2])                     # same learning-rate schedule
        assert np.abs(advf_g - audio).max() <= pg.epsilon + 1e-12 and np.abs(advf_g - advf_o).max() <= 2 * pg.epsilon + 1e-12
    finally:
        e.close()


@pytest.mark.parametrize("mfcc_f32", [0, 1])
def test_config1_succeeding_attack_decisions_at_full_size(oracle, full_system, mfcc_f32):
    """north_star: "reproduce attack-success decisions exactly on the same inputs" (FAKEBOB.py:181-189, 219) -- at
    configs[1] size with attacks that SUCCEED.  The target is the speaker the clean utterance already scores highest
    for, the system threshold is put 0.033 above that score (from the ORACLE's scores of the clean audio): the NES loop
    needs ~10-25 iterations to lift the target's score over it (measured: +0.035 after 15 iterations, +-0.005 from
    run to run).
    (1) The same decision on the same input: the adversarial audio either side stops with is scored by the OTHER side --
        the oracle must call the engine's result a success (loss < 0) and the engine the oracle's, and the iterate ONE
        step before was not yet one for either (the stop is at the same place of each trajectory).
    (2) The trajectories themselves: two float32 evaluations of the GMM step a dozen of the 48 000 samples differently
        in every update (the test above), so after a few iterations the runs are two runs of the same attack; their stop
        iterations differ by the run-to-run spread of the score trajectory around the threshold.  Asserted: both succeed
        (flag +1), stop iterations within +-12 of each other (measured on the MI355X box, engine / oracle: 15 / 10, 8 / 8,
        13 / 12, 13 / 13 with the float64 front-end, 9 / 15, 10 / 10, 14 / 20, 17 / 9 with the float32 one -- the spread of
        ONE implementation over seeds is the same 8 .. 17), both inside the same epsilon ball."""
    over = dict(mfcc_f32=mfcc_f32)
    e, ctx = _pair(oracle, full_system, over)
    kw = dict(KW)
    kw.pop("target"); kw.pop("threshold")
    stops = []
    try:
        for utt, seed in (((0, 42), (0, 43), (1, 42), (2, 43)) if not mfcc_f32 else ((0, 43), (1, 42))):
            audio = synthetic_audio(utt, 48000)
            s0 = ctx.score(audio[:, None])[0]
            tgt = int(np.argmax(s0))
            thr = float(s0[tgt] + 0.033)
            pg = nes_params("OSI", "targeted", seed=seed, stream=utt, max_iter=60, target=tgt, threshold=thr, **kw)
            po = oracle.nes_params("OSI", "targeted", ctx.S, max_iter=60, target=tgt, threshold=thr, **kw)
            adv_g, flag_g, advf_g, tr_g = e.attack(pg, audio)
            adv_o, flag_o, advf_o, tr_o = oracle.attack(po, ctx.fn, ctx.ctx, audio, seed=seed, stream=utt)
            it_g, it_o = tr_g.shape[0] - 1, tr_o.shape[0] - 1
            stops.append((utt, seed, it_g, it_o))
            assert flag_g == 1 and flag_o == 1, (utt, seed, flag_g, flag_o, it_g, it_o)
            assert tr_g[-1, 1] < 0.0 and tr_o[-1, 1] < 0.0 and (tr_g[:-1, 1] >= 0.0).all() and (tr_o[:-1, 1] >= 0.0).all()
            assert abs(it_g - it_o) <= 12, stops
            # (1) the other side's verdict on this side's final iterate (float64 adver -> its int16 cast -> scores -> loss)
            def loss_of(scores):
                others = np.delete(scores, tgt)
                return max(float(others.max()), thr) - float(scores[tgt])
            lg_on_o = loss_of(ctx.score(advf_g[:, None])[0])                     # engine's result, oracle's arithmetic
            lo_on_g = loss_of(e.system_scores(e.score_raw([advf_o])[0])[0])      # oracle's result, engine's arithmetic
            assert lg_on_o < 0.0 and abs(lg_on_o - tr_g[-1, 1]) <= 1e-4, (lg_on_o, tr_g[-1, 1])
            assert lo_on_g < 0.0 and abs(lo_on_g - tr_o[-1, 1]) <= 1e-4, (lo_on_g, tr_o[-1, 1])
            assert np.abs(advf_g - audio).max() <= pg.epsilon + 1e-12 and np.abs(advf_o - audio).max() <= pg.epsilon + 1e-12
            assert np.array_equal(adv_g[:, 0] if adv_g.ndim == 2 else adv_g, (advf_g * 32768).astype(np.int16))
        print("succeeding full-size attacks (mfcc_f32=%d): (utt, seed, stop iteration engine, oracle) %s" % (mfcc_f32, stops))
    finally:
        e.close()


def test_config1_with_the_reference_pipelines_file_round_trips(oracle, full_system):
    """compress_feats = 1 and text_scores = 1: what `attackMain.py` computes with a stock Kaldi recipe.  Both stages
    are bit-identical to the oracle's on the same input (tests/test_gpu_configs.py); end to end the 8-bit feature
    codes amplify the ~1e-6 differences of the two MFCC implementations wherever a value sits on a code boundary and
    the score text quantises to 1e-3 at |score| ~ 150, so the comparison is made at that resolution."""
    over = dict(compress_feats=1, text_scores=1)
    e, ctx = _pair(oracle, full_system, over)
    e0, ctx0 = _pair(oracle, full_system, {})
    try:
        audio = synthetic_audio(0, 48000)
        pg = nes_params("OSI", "targeted", seed=42, stream=0, max_iter=1000, **KW)
        po = oracle.nes_params("OSI", "targeted", ctx.S, max_iter=1000, **KW)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=3)
        flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=42, it=3, stream=0)
        fl0, _, al0, sc0 = e0.get_grad(pg, audio, it=3)
        print("faithful mode: score diff vs oracle %.2e (vs the engine's default mode %.2e)" %
              (np.abs(scg[:ctx.S] - sco).max(), np.abs(scg[:ctx.S] - sc0[:ctx.S]).max()))
        # a score is a difference of two 6-digit numbers of magnitude ~150: one text step is 1e-3
        assert np.abs(scg[:ctx.S] - sco).max() <= 2.5e-3
        assert abs(alg - alo) <= 2.5e-3 and abs(flg - flo) <= 2.5e-3
        for v in scg[:ctx.S]:
            assert abs(v * 1e3 - round(v * 1e3)) < 1e-6     # multiples of the text step
        # the round trips matter even for speaker-minus-UBM scores, where most of compression's ~4e-2 cancels
        assert np.abs(scg[:ctx.S] - sc0[:ctx.S]).max() > 1e-4
        # One NES estimate from the SAME input: a sample's loss moves by a whole text step (1e-3) when one of its
        # scores sits within the float32 error (~3e-6) of a rounding boundary -- about one of the 51 x 6 scores of a
        # batch does --, so the two estimates differ by a few steps' worth and the sign step in a handful of samples.
        rms = float(np.sqrt(np.mean(go * go)))
        flips = np.sign(gg) != np.sign(go)
        print("faithful mode: grad err %.2e (rms %.3f), %d of %d sign flips" % (np.abs(gg - go).max(), rms, int(flips.sum()), gg.size))
        assert np.abs(gg - go).max() <= 0.05 * rms and flips.mean() <= 5e-3
        pg.max_iter = po.max_iter = 5
        adv_g, flag_g, advf_g, tr_g = e.attack(pg, audio)
        adv_o, flag_o, advf_o, tr_o = oracle.attack(po, ctx.fn, ctx.ctx, audio, seed=42, stream=0)
        assert flag_g == flag_o and tr_g.shape == tr_o.shape
        frac = float(np.mean(adv_g != adv_o))
        rows = np.maximum(np.abs(tr_g[:, :2] - tr_o[:, :2]).max(axis=1), np.abs(tr_g[:, 3:] - tr_o[:, 3:]).max(axis=1))
        print("faithful mode: |trace diff| per row", rows, "differing int16 samples %.3f %%" % (100 * frac))
        # The trajectories coincide until the first such step lands differently; from then on they are two runs of the
        # same attack (sign steps of +-lr: the iterates of different runs differ in about half of the samples after a
        # few updates) -- same loss level, same epsilon ball.  Which iteration that is depends on the last bits of the
        # GMM arithmetic (round 3, delta form: none in these five; the log2-domain form: the third).
        assert rows[0] <= 2.5e-3 and rows[1] <= 5e-3 and rows.max() <= 2.5e-2
        assert np.abs(advf_g - advf_o).max() <= 2 * pg.max_lr * pg.max_iter
    finally:
        e.close()
        e0.close()


def test_config3_csi_untargeted_get_grad_at_full_size(oracle, full_system):
    """BASELINE configs[3]'s per-utterance work: GMM-UBM CSI (speaker models only: the first one is the base model of
    k_gmm_fx2w's delta form) untargeted, C = 2048, spd = 50, 3 s -- one get_grad against the oracle."""
    _, spk = full_system
    e = Engine(0)
    try:
        e.load_gmm(spk)
        e.set_system("CSI")
        assert e.gmm_kernel_variant.startswith("fx2w/")
        gc, miv, iv = stack_models(spk)
        ctx = oracle.GmmSystemCtx(oracle.default_cfg(), "CSI", gc, miv, iv, nthreads=NTHR)
        audio = synthetic_audio(4, 48000)
        kw = dict(KW, true=2)
        kw.pop("target"); kw.pop("threshold")
        pg = nes_params("CSI", "untargeted", seed=7, stream=1, **kw)
        po = oracle.nes_params("CSI", "untargeted", ctx.S, **kw)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=2)
        flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=7, it=2, stream=1)
        rms = float(np.sqrt(np.mean(go * go)))
        flips = np.sign(gg) != np.sign(go)
        print("full-size CSI get_grad (%s): score err %.2e, loss err %.2e, grad err %.2e (rms %.3f), %d sign flips" %
              (e.gmm_kernel_variant, np.abs(scg[:ctx.S] - sco).max(), abs(alg - alo), np.abs(gg - go).max(), rms, int(flips.sum())))
        assert abs(alg - alo) <= 1e-4 and abs(flg - flo) <= 1e-4
        assert np.abs(scg[:ctx.S] - sco).max() <= 1e-4
        assert np.abs(gg - go).max() <= 0.02 * rms
        assert flips.mean() <= 1e-3 and (not flips.any() or np.abs(go[flips]).max() <= 0.02 * rms)
    finally:
        e.close()
