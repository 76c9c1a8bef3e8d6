"""GPU parity for the non-default code paths: generic MFCC kernel (padded_length != 512), generic and
compile-time delta options, feature dims that need other GMM kernel instantiations, the plain f32-MFMA
GMM kernel, long utterances (sliding CMVN), and the attack's host-round-trip batch size."""
import numpy as np
import pytest

from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import stack_models, synthetic_audio, synthetic_gmm_system

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-4


def _wav(utt, n=48000):
    return (synthetic_audio(utt, n) * 32768.0).astype(np.int16)


FRONTENDS = [
    dict(),                                                          # recipe: P=512, deltas 2/3  (k_mfcc_r4, k_delta_cmvn<2,3>)
    dict(delta_window=2),                                            # k_delta_cmvn<2,2>
    dict(delta_order=1, delta_window=2),                             # generic delta path, D = 48
    dict(delta_order=0),                                             # no deltas, D = 24
    dict(sample_freq=8000.0, frame_length=200, frame_shift=80, padded_length=256, high_freq=3700.0),  # generic k_mfcc
    dict(num_ceps=20, num_mel_bins=23, delta_order=3, delta_window=2),  # D = 80
    dict(raw_energy=0),                                              # energy after pre-emphasis and windowing (k_mfcc_r16<12, false>)
    dict(raw_energy=0, frame_length=320, frame_shift=160),           # ... with a short window (k_mfcc_r16<0, false>)
    dict(frame_length=320, frame_shift=160),                         # k_mfcc_r16<0, true>
    dict(num_ceps=17, num_mel_bins=23, delta_order=3, delta_window=2),  # D = 68: k_gmm_fx2w with other K padding places
    dict(num_ceps=16, num_mel_bins=23, delta_order=3, delta_window=2),  # D = 64: ... and a whole spare half chunk
]


@pytest.mark.parametrize("over", FRONTENDS)
def test_frontend_and_scores_for_other_configs(oracle, over):
    cfg = oracle.default_cfg(**over)
    e = Engine(0)
    try:
        e.set_frontend(**over)
        D = e.feat_dim
        ubm, spk = synthetic_gmm_system(n_speakers=2, C=96, D=D)
        e.load_gmm([ubm] + spk)
        # the one-wave-per-SIMD kernel takes D = 64, 68, 72 (K padded to 80 with room for the constants and the reference)
        assert e.gmm_kernel_variant.startswith("fx2w/") == (D in (64, 68, 72))
        wavs = [_wav(0, 24000), _wav(1, 9000), _wav(2, 40000)]
        for w in wavs[:2]:
            fg, Tg = e.debug_feats(w)
            fo, To = oracle.frontend(cfg, w)
            assert Tg == To and fg.shape == fo.shape
            assert np.abs(fg.astype(np.float64) - fo).max() <= 2e-5
        raw_g, tv_g = e.score_raw(wavs)
        gc, miv, iv = stack_models([ubm] + spk)
        raw_o, tv_o = oracle.gmm_score_batch(cfg, wavs, gc, miv, iv, nthreads=4)
        assert np.array_equal(tv_g, tv_o)
        assert np.abs(raw_g - raw_o).max() <= SCORE_TOL
    finally:
        e.close()


@pytest.mark.parametrize("over", FRONTENDS[-2:] + [dict()])
def test_f6_class_with_other_feature_dimensions(oracle, monkeypatch, over):
    """The F6 items' block 2 holds the dimensions 64 .. D - 1 (eight of them at D = 72, four at 68, none at 64): heavily
    enrolled speakers (the corrections matter: one product alone is an order of magnitude off) with the class forced
    in every tile, against the oracle and against three f16 products."""
    from fakebob_amd.models import ENROL_REALISTIC
    cfg = oracle.default_cfg(**over)
    err = {}
    for p in ("6", "3", "1"):
        monkeypatch.setenv("FB_GMM_DELTA_P", p)
        e = Engine(0)
        try:
            e.set_frontend(**over)
            D = e.feat_dim
            ubm, spk = synthetic_gmm_system(n_speakers=3, C=256, D=D, **ENROL_REALISTIC)
            e.load_gmm([ubm] + spk)
            assert e.gmm_kernel_variant == "fx2w/" + p
            wavs = [_wav(0, 24000), _wav(1, 9000), _wav(2, 40000), _wav(3, 48000)]
            raw_g, tv_g = e.score_raw(wavs)
        finally:
            e.close()
        gc, miv, iv = stack_models([ubm] + spk)
        raw_o, tv_o = oracle.gmm_score_batch(cfg, wavs, gc, miv, iv, nthreads=4)
        assert np.array_equal(tv_g, tv_o)
        err[p] = float(np.abs(raw_g - raw_o).max())
    print("D = %d: max |err| F6 %.3g, three products %.3g, one product %.3g" % (D, err["6"], err["3"], err["1"]))
    assert D == {0: 72, 1: 72}.get(len(over), 68 if over.get("num_ceps") == 17 else 64)
    assert err["3"] <= SCORE_TOL and err["6"] <= 1e-5 + err["3"] and err["1"] > 3.0 * err["6"]


@pytest.mark.parametrize("mode,env", [("bx3", {"FB_GMM_MODE": "bx3"}), ("fx2", {"FB_GMM_NARROW": "1"})])
def test_other_gmm_kernels_still_match(oracle, monkeypatch, mode, env):
    """The default scoring kernel is k_gmm_fx2w; the general two-term kernel (k_gmm_fx2: any number of variance
    groups, partial tiles, the gselect dump) and the exact bf16x3 fallback stay selectable and correct."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    e = Engine(0)
    try:
        ubm, spk = synthetic_gmm_system(n_speakers=3, C=256, D=72)
        e.load_gmm([ubm] + spk)
        assert e.gmm_kernel == mode
        wavs = [_wav(3, 30000), _wav(4, 16000)]
        raw_g, _ = e.score_raw(wavs)
        gc, miv, iv = stack_models([ubm] + spk)
        raw_o, _ = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=4)
        assert np.abs(raw_g - raw_o).max() <= SCORE_TOL
    finally:
        e.close()


@pytest.mark.parametrize("batch", ["1", "3", "16"])
def test_attack_is_independent_of_the_queue_depth(monkeypatch, batch):
    """Early stop, plateau schedule and trace live on the device; the number of iterations queued per host
    round trip must not change anything."""
    ubm, spk = synthetic_gmm_system(n_speakers=3, C=128, D=72)
    audio = synthetic_audio(6, 16000)

    def run():
        e = Engine(0)
        try:
            e.load_gmm([ubm] + spk)
            e.set_system("OSI")
            raw, _ = e.score_raw([(audio * 32768.0).astype(np.int16)])
            sc = raw[0, 1:] - raw[0, 0]
            target = int(np.argsort(sc)[-2])
            p = nes_params("OSI", "targeted", samples_per_draw=10, max_iter=40, target=target,
                           threshold=float(sc.max()) - 0.05, epsilon=0.004, max_lr=0.002, seed=11, stream=2)
            return e.attack(p, audio)
        finally:
            e.close()

    monkeypatch.setenv("FB_ATTACK_BATCH", "4")
    ref = run()
    monkeypatch.setenv("FB_ATTACK_BATCH", batch)
    got = run()
    assert got[1] == ref[1]                          # success flag
    assert np.array_equal(got[0], ref[0])            # int16 adversarial audio
    assert np.array_equal(got[2], ref[2])            # float64 adversarial audio (bit-identical)
    assert np.array_equal(got[3], ref[3])            # trace rows (bit-identical)


@pytest.mark.parametrize("batch", ["4", "2", "3"])
def test_back_to_back_early_stopping_attacks_on_one_engine(oracle, monkeypatch, batch):
    """attackMain.py:331,370 calls fake_bob.attack back to back on one model object.  An attack that stops at
    iteration 0 leaves FB_ATTACK_BATCH - 1 queued launches behind that find the stop flag raised and do nothing;
    k_vad_delta_cmvn_p's two exchange-slot sets alternate per launch, so after an odd number of such launches the
    next attack once polled slots that still held the previous attack's CMVN block sums (round-4 advisor finding).
    The second attack must be the same on a used engine as on a fresh one, bit for bit, and equal the oracle's."""
    from oracle import oracle as O
    monkeypatch.setenv("FB_ATTACK_BATCH", batch)
    ubm, spk = synthetic_gmm_system(n_speakers=3, C=128, D=72)
    models = [ubm] + spk
    a1, a2 = synthetic_audio(6, 16000), synthetic_audio(9, 16000)
    gc, miv, iv = stack_models(models)
    ctx = O.GmmSystemCtx(O.default_cfg(), "OSI", gc, miv, iv, nthreads=4)
    s1 = ctx.score(a1[:, None])[0]
    tgt = int(np.argmax(s1))
    kw2 = dict(samples_per_draw=10, max_iter=6, target=0, threshold=-1.0, epsilon=0.002)

    def first(e):  # loss[0] < 0 at iteration 0: the launches queued behind it do nothing
        p = nes_params("OSI", "targeted", samples_per_draw=10, max_iter=8, target=tgt, threshold=float(s1.min() - 5.0),
                       adver_thresh=-1.0, seed=1)
        adv, flag, _, tr = e.attack(p, a1)
        assert flag == 1 and tr.shape[0] == 1

    def second(e):
        return e.attack(nes_params("OSI", "targeted", seed=11, stream=0, **kw2), a2)

    fresh = Engine(0)
    used = Engine(0)
    try:
        for e in (fresh, used):
            e.load_gmm(models)
            e.set_system("OSI")
        ref = second(fresh)
        first(used)
        got = second(used)
        again = second(used)   # and once more behind a full-length attack
    finally:
        fresh.close()
        used.close()
    for r in (got, again):
        assert r[1] == ref[1]
        assert np.array_equal(r[0], ref[0]) and np.array_equal(r[2], ref[2]) and np.array_equal(r[3], ref[3])
    po = O.nes_params("OSI", "targeted", ctx.S, **kw2)
    adv_o, flag_o, advf_o, tr_o = O.attack(po, ctx.fn, ctx.ctx, a2, seed=11, stream=0)
    assert flag_o == got[1] and tr_o.shape == got[3].shape
    assert np.abs(tr_o - got[3]).max() <= 1e-4
    assert int(np.sum(adv_o != got[0])) == 0


@pytest.mark.parametrize("spd,n", [(10, 16000), (50, 24000), (64, 9000)])
def test_update_in_the_finalising_launch_equals_the_separate_launch(monkeypatch, spd, n):
    """Round 5: on the fused chain the momentum sign step and the next iteration's perturbed batch ride in the launch
    that finalises the GMM scores and runs the loss body (k_gmm_finalize_loss_update): their workgroups draw the next
    iteration's normals while the scores are finalised and wait for the loss body's publication.  FB_FUSE_UPD=0 keeps
    k_update_perturb as a launch of its own: same arithmetic in the same order, so an attack that runs to max_iter, one
    that stops early and one right behind it on the same engine must be identical bit for bit (samples_per_draw = 64:
    the launch needs more than 64 KB of LDS per workgroup)."""
    ubm, spk = synthetic_gmm_system(n_speakers=3, C=128, D=72)
    audio = synthetic_audio(6, n)
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        e.set_system("OSI")
        e.set_fused_chain(True)
        raw, _ = e.score_raw([(audio * 32768.0).astype(np.int16)])
        sc = raw[0, 1:] - raw[0, 0]
        tgt = int(np.argmax(sc))
        p_stop = nes_params("OSI", "targeted", samples_per_draw=spd, max_iter=40, target=tgt, threshold=float(sc[tgt]) + 0.01,
                            epsilon=0.004, max_lr=0.002, seed=11, stream=2)
        p_full = nes_params("OSI", "targeted", samples_per_draw=spd, max_iter=7, target=tgt, threshold=float(sc.max()) + 50.0, seed=5)

        def run():
            return [e.attack(p_full, audio), e.attack(p_stop, audio), e.attack(p_stop, audio), e.get_grad(p_full, audio, it=3)]
        monkeypatch.setenv("FB_FUSE_UPD", "0")
        ref = run()
        monkeypatch.delenv("FB_FUSE_UPD", raising=False)
        got = run()
        # ... and with the arrival counter + last arriver instead of the exchange slots + last-indexed workgroup
        monkeypatch.setenv("FB_FIN_COUNTER", "1")
        got_counter = run()
        monkeypatch.delenv("FB_FIN_COUNTER", raising=False)
        got_again = run()     # (the slots were left clean by the launches before the counter ones)
        # ... and with the workgroups' roles drawn from an arrival ticket instead of blockIdx (round 6: not the default any more)
        monkeypatch.setenv("FB_FIN_TICKET", "1")
        got_ticket = run()
        monkeypatch.delenv("FB_FIN_TICKET", raising=False)
        got_last = run()      # (the ticket word was left at zero)
    finally:
        e.close()
    for other in (got, got_counter, got_again, got_ticket, got_last):
        for a, b in zip(ref[:3], other[:3]):
            assert a[1] == b[1] and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert ref[3][0] == other[3][0] and np.array_equal(ref[3][1], other[3][1])
    assert ref[0][3].shape[0] == 7 and 1 <= ref[1][3].shape[0] < 40 and ref[1][1] == 1


def test_attack_on_a_site_with_more_than_ten_models_equals_the_oracle(oracle, monkeypatch):
    """UBM + 12 speakers: k_gmm_fx2w scores them in two passes (round 4).  The whole NES loop on top of that -- get_grad,
    the loss over 12 scores, early stop, trace -- against the oracle on the 1 s / spd = 10 attack the bit-identical
    trajectories of tests/test_gpu_parity.py use."""
    for k in ("FB_GMM_NARROW", "FB_GMM_MODE", "FB_GMM_DELTA_P"):
        monkeypatch.delenv(k, raising=False)
    from fakebob_amd.models import stack_models
    ubm, spk = synthetic_gmm_system(n_speakers=12, C=256, D=72)
    models = [ubm] + spk
    gc, miv, iv = stack_models(models)
    e = Engine(0)
    try:
        e.load_gmm(models)
        e.set_system("OSI")
        assert e.gmm_kernel_variant.startswith("fx2w/")
        ctx = oracle.GmmSystemCtx(oracle.default_cfg(), "OSI", gc, miv, iv, nthreads=8)
        audio = synthetic_audio(6, 16000)
        kw = dict(samples_per_draw=10, max_iter=6, target=7, threshold=-1.0, epsilon=0.002)
        pg = nes_params("OSI", "targeted", seed=11, stream=0, **kw)
        po = oracle.nes_params("OSI", "targeted", ctx.S, **kw)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=0)
        flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=11, it=0, stream=0)
        assert scg[:ctx.S].shape == sco.shape == (12,)
        assert np.abs(scg[:ctx.S] - sco).max() <= 2e-5 and abs(flg - flo) <= 2e-5 and abs(alg - alo) <= 2e-5
        adv_g, flag_g, advf_g, tr_g = e.attack(pg, audio)
        adv_o, flag_o, advf_o, tr_o = oracle.attack(po, ctx.fn, ctx.ctx, audio, seed=11, stream=0)
        assert flag_g == flag_o and tr_g.shape == tr_o.shape
        # the first iteration to score tolerance; afterwards the gradient entries within float32 noise of zero step
        # their samples in opposite directions and the iteration amplifies it (tests/test_gpu_fullsize_gmm.py)
        rows = np.abs(tr_g - tr_o).max(axis=1)
        assert rows[0] <= 1e-4 and rows.max() <= 1e-2, rows
        assert np.array_equal(tr_g[:, 2], tr_o[:, 2])                         # same learning-rate schedule
        assert np.abs(advf_g - audio).max() <= pg.epsilon + 1e-12
    finally:
        e.close()


def test_text_scores_option_matches_the_oracle_and_quantises(oracle):
    """fb_frontend_cfg.text_scores: raw scores go through Kaldi's 6-significant-digit text output, as the
    reference's helpers read them; the device rounding is bit-identical to the oracle's."""
    cfg = oracle.default_cfg(text_scores=1)
    e = Engine(0)
    try:
        e.set_frontend(text_scores=1)
        ubm, spk = synthetic_gmm_system(n_speakers=2, C=96, D=72)
        e.load_gmm([ubm] + spk)
        wavs = [_wav(0, 24000), _wav(1, 16000), _wav(2, 40000)]
        raw_g, _ = e.score_raw(wavs)
        gc, miv, iv = stack_models([ubm] + spk)
        raw_o, _ = oracle.gmm_score_batch(cfg, wavs, gc, miv, iv, nthreads=4)
        for v in raw_g.reshape(-1):
            assert v == float("%.6g" % v)                       # 6 significant digits survive a text round trip
        # both sides round the same way; a score within float rounding of a 6-digit tie may land on either side
        assert np.mean(raw_g == raw_o) >= 0.8 and np.abs(raw_g - raw_o).max() <= 1.1e-3
        e.set_frontend(text_scores=0)
        e.load_gmm([ubm] + spk)
        raw_full, _ = e.score_raw(wavs)
        assert np.abs(raw_full - raw_g).max() <= 6e-4            # |scores| ~ 1e2 -> 3 decimals kept
        assert np.any(raw_full != raw_g)
    finally:
        e.close()


def test_model_outside_f16_range_runs_on_the_bf16_kernel(oracle):
    """k_gmm_fx2 needs |mu/sigma^2|, 1/(2 sigma^2), |gconst| < 2^15; a model with a variance of 1e-5 does not
    fit, is routed to k_gmm_bx3 at load time and still matches the oracle."""
    import copy
    e = Engine(0)
    try:
        ubm, spk = synthetic_gmm_system(n_speakers=2, C=96, D=72)
        e.load_gmm([ubm] + spk)
        assert e.gmm_kernel == "fx2"
        gc, miv, iv = [np.array(a, copy=True) for a in stack_models([ubm] + spk)]
        # one dimension of one component with variance 1e-5 in every model (variances stay shared)
        var_old = 1.0 / iv[:, 7, 3]
        mean = miv[:, 7, 3] * var_old
        iv[:, 7, 3] = np.float32(1.0e5)
        miv[:, 7, 3] = (mean * 1.0e5).astype(np.float32)
        gc[:, 7] += (0.5 * (np.log(1.0e5) + np.log(var_old)) - 0.5 * mean * mean * (1.0e5 - 1.0 / var_old)).astype(np.float32)
        e.load_gmm_arrays(gc, miv, iv)
        assert e.gmm_kernel == "bx3"
        wavs = [_wav(3, 30000), _wav(4, 16000)]
        raw_g, _ = e.score_raw(wavs)
        raw_o, _ = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=4)
        assert np.abs(raw_g - raw_o).max() <= SCORE_TOL * max(1.0, np.abs(raw_o).max() / 200.0)
    finally:
        e.close()


def test_compress_feats_option_is_bit_identical_to_the_oracle_round_trip(oracle):
    """fb_frontend_cfg.compress_feats: the MFCC matrix takes Kaldi's CompressedMatrix round trip (make_mfcc.sh's
    `copy-feats --compress=true`) on the device.  On the SAME matrix the device codes equal the oracle's bit for
    bit; downstream stages then see exactly what the oracle's do."""
    e = Engine(0)
    try:
        ubm, spk = synthetic_gmm_system(n_speakers=2, C=96, D=72)
        e.load_gmm([ubm] + spk)
        cfg0, cfg1 = oracle.default_cfg(), oracle.default_cfg(compress_feats=1)
        for utt, n in ((0, 48000), (1, 9000), (2, 1400), (3, 1000), (4, 100000), (5, 640000)):   # T = 300, 56, 9, 6 (two-byte), 625, 4000 (beyond the LDS-staged column)
            w = _wav(utt, n)
            e.set_frontend(compress_feats=0)
            raw = e.debug_mfcc(w)
            e.set_frontend(compress_feats=1)
            got = e.debug_mfcc(w)
            want = oracle.compress_roundtrip(raw)
            assert np.array_equal(got, want), (utt, n, np.abs(got - want).max())
            assert not np.array_equal(got, raw)
            fg, Tg = e.debug_feats(w)
            v = oracle.vad(cfg0, want).astype(bool)
            fo = oracle.cmvn_sliding(cfg0, oracle.deltas(cfg0, want))[v]
            assert fg.shape == fo.shape and np.abs(fg.astype(np.float64) - fo).max() <= 2e-5
        # whole path against the oracle's own compressed front-end: the 8-bit codes amplify the ~1e-6 differences
        # of the two MFCC implementations wherever a value sits on a code boundary, so a handful of codes may differ
        e.set_frontend(compress_feats=1)
        wavs = [_wav(0, 48000), _wav(1, 20000), _wav(5, 30000)]
        raw_g, tv_g = e.score_raw(wavs)
        gc, miv, iv = stack_models([ubm] + spk)
        raw_o, tv_o = oracle.gmm_score_batch(cfg1, wavs, gc, miv, iv, nthreads=4)
        raw_u, _ = oracle.gmm_score_batch(cfg0, wavs, gc, miv, iv, nthreads=4)
        assert np.abs(tv_g - tv_o).max() <= 2
        assert np.abs(raw_g - raw_o).max() <= 0.02 * np.abs(raw_u - raw_o).max() + 1e-4
        assert np.abs(raw_u - raw_o).max() > 1e-3          # the option matters: compression moves scores by far more
    finally:
        e.close()


def test_very_long_and_very_short_utterances_in_one_batch(oracle):
    """Ragged extremes in one call: a 20 s utterance (T = 2000 frames: beyond the fused delta+CMVN kernel's
    per-utterance limit, several sliding-CMVN windows), a 0.1 s one (T = 10) and ordinary ones."""
    e = Engine(0)
    try:
        ubm, spk = synthetic_gmm_system(n_speakers=2, C=96, D=72)
        e.load_gmm([ubm] + spk)
        wavs = [_wav(0, 320000), _wav(1, 1600), _wav(2, 48000), _wav(3, 24000)]
        raw_g, tv_g = e.score_raw(wavs)
        gc, miv, iv = stack_models([ubm] + spk)
        raw_o, tv_o = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=4)
        assert np.array_equal(tv_g, tv_o) and tv_o[0] > 1000 and tv_o[1] <= 10
        assert np.abs(raw_g - raw_o).max() <= SCORE_TOL
        # the same utterances one at a time (other launch shapes) agree with the batch
        for i, w in enumerate(wavs):
            r1, t1 = e.score_raw([w])
            assert t1[0] == tv_g[i] and np.abs(r1[0] - raw_g[i]).max() <= 2e-6
    finally:
        e.close()


def test_empty_and_degenerate_batches_are_errors_not_crashes():
    from fakebob_amd._native import NativeError
    e = Engine(0)
    try:
        ubm, spk = synthetic_gmm_system(n_speakers=1, C=64, D=72)
        e.load_gmm([ubm] + spk)
        with pytest.raises((NativeError, ValueError)):
            e.score_raw([])
        with pytest.raises((NativeError, ValueError)):
            e.score_raw([np.zeros(0, np.int16)])
        with pytest.raises(NativeError):
            e.score_raw([np.zeros(16000, np.int16)])          # digital silence: no voiced frame
        r, tv = e.score_raw([_wav(0, 16000)])                 # the engine is still usable afterwards
        assert tv[0] > 0 and np.isfinite(r).all()
    finally:
        e.close()


def test_generic_mfcc_kernel_matches_the_fast_one(oracle):
    """k_mfcc_r16 (four frames per wave, radix-16 passes in registers) handles 512-point frames with an even
    frame length; any other front-end runs on the generic k_mfcc.  A 401-sample frame (25.0625 ms) takes the generic
    kernel, a 400-sample frame the fast one: both against the oracle."""
    for flen in (400, 401):
        e = Engine(0)
        try:
            e.set_frontend(frame_length=flen)
            cfg = oracle.default_cfg(frame_length=flen)
            for utt, n in ((0, 48000), (1, 9000), (4, 100000)):
                w = _wav(utt, n)
                mo = oracle.mfcc(cfg, w)
                mg = e.debug_mfcc(w)
                assert mg.shape == mo.shape
                assert np.abs(mg.astype(np.float64) - mo).max() <= 2e-5
        finally:
            e.close()


def test_pipeline_round_trips_are_constructor_options(tmp_path, oracle):
    """text_scores / compress_feats (the two file round trips of the reference's pipeline) are constructor keywords of
    the systems, not only environment variables: gmm_SV(text_scores=True) returns scores that are differences of
    6-significant-digit values, equal to the oracle's with the same option."""
    from fakebob_amd.systems import gmm_SV
    ubm, spk = synthetic_gmm_system(n_speakers=1, C=96, D=72)
    ml = ["spk0", "utt0", spk[0], 0.0, 1.0]
    e = Engine(0)
    try:
        audio = synthetic_audio(2, 16000)
        plain = gmm_SV(str(tmp_path / "a"), ml, ubm, pre_model_dir=str(tmp_path), engine=e)
        s_plain = float(plain.score(audio))
        txt = gmm_SV(str(tmp_path / "b"), ml, ubm, pre_model_dir=str(tmp_path), engine=e, text_scores=True)
        s_txt = float(txt.score(audio))
        gc, miv, iv = stack_models([ubm, spk[0]])
        raw_o, _ = oracle.gmm_score_batch(oracle.default_cfg(text_scores=1), [_wav(2, 16000)], gc, miv, iv)
        assert s_txt == raw_o[0, 1] - raw_o[0, 0]
        assert s_txt != s_plain and abs(s_txt - s_plain) < 2e-3
    finally:
        e.set_frontend(text_scores=0)
        e.close()


@pytest.mark.parametrize("C", [32, 96])
@pytest.mark.parametrize("n_speakers", [1, 5])
def test_wide_scoring_kernel_single_tile_chunks(oracle, monkeypatch, n_speakers, C):
    """One to three component tiles in all: a chunk is a single tile, so the LDS-DMA schedule of k_gmm_fx2w only ever
    re-requests its last group (the clamp at the chunk's end) -- against the oracle."""
    monkeypatch.delenv("FB_GMM_NARROW", raising=False)
    monkeypatch.delenv("FB_GMM_MODE", raising=False)
    cfg = oracle.default_cfg()
    ubm, spk = synthetic_gmm_system(n_speakers=n_speakers, C=C, D=72)
    wavs = [_wav(u, 9000 + 3111 * u) for u in range(5)]
    gc, miv, iv = stack_models([ubm] + spk)
    raw_o, tv_o = oracle.gmm_score_batch(cfg, wavs, gc, miv, iv, nthreads=4)
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        assert e.gmm_kernel_variant.startswith("fx2w/")
        raw_g, tv_g = e.score_raw(wavs)
    finally:
        e.close()
    assert np.array_equal(tv_g, tv_o)
    assert np.abs(raw_g - raw_o).max() <= 2e-5


@pytest.mark.parametrize("delta_p", [1, 2, 3, 6])
@pytest.mark.parametrize("n_speakers", [1, 2, 3, 4, 5, 6, 7, 8, 9])
def test_wide_scoring_kernel_every_model_count(oracle, monkeypatch, n_speakers, delta_p):
    """k_gmm_fx2w is instantiated per model count M = 2 .. 10 (SV: UBM + 1; OSI: UBM + up to 9 speakers) and per number of
    partial products of the delta items (6: the F6 class, the corrections as block-scaled fp6 / fp4 products): each
    instantiation has its own LDS slot split, LDS-DMA piece schedule,
    accumulator rotation and update-slice schedule.  Several component chunks per strip (C = 1024 -> tiles streamed
    through both slots many times), a ragged last strip, against the oracle and against the general kernel on the
    same inputs."""
    monkeypatch.delenv("FB_GMM_NARROW", raising=False)
    monkeypatch.delenv("FB_GMM_MODE", raising=False)
    monkeypatch.setenv("FB_GMM_DELTA_P", str(delta_p))
    cfg = oracle.default_cfg()
    ubm, spk = synthetic_gmm_system(n_speakers=n_speakers, C=1024, D=72)
    wavs = [_wav(u, 16000 + 1234 * u) for u in range(7)]
    gc, miv, iv = stack_models([ubm] + spk)
    raw_o, tv_o = oracle.gmm_score_batch(cfg, wavs, gc, miv, iv, nthreads=8)
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        assert e.gmm_kernel_variant == "fx2w/%d" % delta_p
        raw_w, tv_w = e.score_raw(wavs)
    finally:
        e.close()
    monkeypatch.setenv("FB_GMM_NARROW", "1")
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        assert e.gmm_kernel_variant == "fx2"
        raw_n, _ = e.score_raw(wavs)
    finally:
        e.close()
    assert np.array_equal(tv_w, tv_o)
    assert np.abs(raw_w - raw_o).max() <= 2e-5, np.abs(raw_w - raw_o).max()
    assert np.abs(raw_w - raw_n).max() <= 2e-5


def test_wide_scoring_kernel_csi_base_is_the_first_speaker(oracle, monkeypatch):
    """CSI has no UBM in its model list (gmm_ubm_CSI.py:49): k_gmm_fx2w then scores speakers 1 .. as deltas from
    speaker 0 (two independent adaptations apart, so the shift statistic is ~sqrt 2 of the OSI one)."""
    for k in ("FB_GMM_NARROW", "FB_GMM_MODE", "FB_GMM_DELTA_P"):
        monkeypatch.delenv(k, raising=False)
    cfg = oracle.default_cfg()
    ubm, spk = synthetic_gmm_system(n_speakers=5, C=2048, D=72)
    wavs = [_wav(u, 20000 + 4321 * u) for u in range(4)]
    gc, miv, iv = stack_models(spk)
    raw_o, _ = oracle.gmm_score_batch(cfg, wavs, gc, miv, iv, nthreads=8)
    e = Engine(0)
    try:
        e.load_gmm(spk)
        e.set_system("CSI", np.zeros(5), np.ones(5))
        assert e.gmm_kernel_variant in ("fx2w/1", "fx2w/2", "fx2w/3"), (e.gmm_kernel_variant, e.gmm_shift_rms)
        raw_g, _ = e.score_raw(wavs)
    finally:
        e.close()
    assert np.abs(raw_g - raw_o).max() <= 2e-5


@pytest.mark.parametrize("n_speakers,delta_p", [(10, ""), (10, "6"), (12, ""), (18, "3"), (19, "6"), (27, ""), (28, "")])
def test_more_than_ten_models_run_in_several_passes(oracle, monkeypatch, n_speakers, delta_p):
    """One launch of k_gmm_fx2w holds a tile of 1 + M items and the logsumexp state of M models in the LDS of a CU: M <= 10.
    Larger sites -- UBM + 10 speakers is already one too many -- are scored in up to three PASSES of the same kernel, each
    with the base model and its share of the others (round 4; before, they fell to the general kernel at about twice
    the time per model); beyond 1 + 9 x 3 = 28 models the general kernel takes over.  Same results either way."""
    for k in ("FB_GMM_NARROW", "FB_GMM_MODE", "FB_GMM_DELTA_P"):
        monkeypatch.delenv(k, raising=False)
    if delta_p:
        monkeypatch.setenv("FB_GMM_DELTA_P", delta_p)
    ubm, spk = synthetic_gmm_system(n_speakers=n_speakers, C=256, D=72, enrol_frames=2000.0)
    wavs = [_wav(u, 16000 + 777 * u) for u in range(4)]
    gc, miv, iv = stack_models([ubm] + spk)
    raw_o, tv_o = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=8)
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        if n_speakers + 1 <= 28:
            assert e.gmm_kernel_variant.startswith("fx2w/"), e.gmm_kernel_variant
            if delta_p:
                assert e.gmm_kernel_variant == "fx2w/" + delta_p
        else:
            assert e.gmm_kernel_variant == "fx2"
        raw_g, tv_g = e.score_raw(wavs)
        raw_g2, _ = e.score_raw(wavs[::-1])
    finally:
        e.close()
    assert np.array_equal(tv_g, tv_o)
    assert np.abs(raw_g - raw_o).max() <= 2e-5, np.abs(raw_g - raw_o).max()
    assert np.array_equal(raw_g2[::-1], raw_g)            # every model's column, whichever pass wrote it
    monkeypatch.setenv("FB_GMM_NARROW", "1")
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        assert e.gmm_kernel_variant == "fx2"
        raw_n, _ = e.score_raw(wavs)
    finally:
        e.close()
    assert np.abs(raw_g - raw_n).max() <= 2e-5


def _frame_lls(models, x):
    """float64 reference of gmm-global-get-frame-likes: [M, T] from [T, D] features."""
    x = np.asarray(x, np.float64)
    out = []
    for m in models:
        ll = (m.gconsts.astype(np.float64)[None, :] + x @ m.means_invvars.astype(np.float64).T
              - 0.5 * (x * x) @ m.inv_vars.astype(np.float64).T)
        mx = ll.max(axis=1)
        out.append(mx + np.log(np.exp(ll - mx[:, None]).sum(axis=1)))
    return np.stack(out)


def test_wide_kernel_reference_rescue_and_range_paths(monkeypatch):
    """k_gmm_fx2w keeps its logsumexp relative to a per-frame reference R that starts from a lower bound (two anchor
    components) and is repaired by a cold path when a tile's values run more than 2^100 above it; frames whose balanced
    values leave f16's range take a shifted slow path.  The front-end never produces such frames, fb_debug_gmm_frames
    does: rows drawn from the model, rows tens of sigmas away from every component (the rescue), rows of magnitude
    1e3 .. 1e4 (the range shift), all-zero rows -- each against the float64 formula, for every delta variant P."""
    D, C = 72, 512
    ubm, spk = synthetic_gmm_system(n_speakers=5, C=C, D=D)
    models = [ubm] + spk
    rng = np.random.default_rng(7)
    var = 1.0 / ubm.inv_vars.astype(np.float64)
    mu = ubm.means_invvars.astype(np.float64) * var
    ks = rng.integers(0, C, 700)
    near = mu[ks] + np.sqrt(var[ks]) * rng.standard_normal((700, D))
    sd = np.sqrt(var.mean(axis=0) + mu.var(axis=0))
    far = mu[ks[:300]] + 25.0 * sd * rng.standard_normal((300, D))          # hundreds of nats below any anchor bound
    onefar = near[:200].copy()
    onefar[:, 3] += 60.0 * sd[3] * np.sign(rng.standard_normal(200))        # one dimension out: a narrow component may still win
    huge = rng.standard_normal((130, D)) * np.logspace(3, 4, 130)[:, None]  # |x| sd^-1 > 181: the range shift
    rows = np.concatenate([near, far, np.zeros((7, D)), onefar, huge, near[:50]]).astype(np.float32)
    want = _frame_lls(models, rows)
    for P in ("1", "2", "3", "6"):
        monkeypatch.setenv("FB_GMM_DELTA_P", P)
        e = Engine(0)
        try:
            e.load_gmm(models)
            assert e.gmm_kernel_variant == "fx2w/" + P
            got = e.debug_gmm_frames(rows)
        finally:
            e.close()
        assert np.isfinite(got).all()
        rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
        print("P =", P, "max relative error %.2e (|ll| up to %.1e)" % (rel.max(), np.abs(want).max()))
        # float32 accumulation at these magnitudes; with P < 3 a FRAME's speaker values carry the dropped products'
        # ~1e-4 (random from frame to frame: the scores average hundreds of frames)
        assert rel[0].max() <= 2e-6 and rel.max() <= (2e-6 if P == "3" else 1e-5)
        sysd = np.abs((got[1:] - got[:1]) - (want[1:] - want[:1]))[:, :700]
        assert sysd.max() <= 2e-3                      # per FRAME (the scores average hundreds of them)


def test_wide_kernel_with_far_models_keeps_every_models_sum(monkeypatch):
    """The reference's lower bound covers every model through a Cauchy-Schwarz slack on the deltas: models far from the
    base one (unrelated means) must neither overflow nor lose their whole sum to underflow."""
    D, C = 72, 256
    ubm, _ = synthetic_gmm_system(n_speakers=1, C=C, D=D)
    other, _ = synthetic_gmm_system(n_speakers=1, C=C, D=D, seed_ubm=99)
    from fakebob_amd.models import DiagGmm
    var = 1.0 / ubm.inv_vars.astype(np.float64)
    w = np.full(C, 1.0 / C)
    mu_far = (other.means_invvars.astype(np.float64) / other.inv_vars.astype(np.float64)) * 3.0 + 5.0
    far = DiagGmm.from_internal(w, (mu_far / var).astype(np.float32), ubm.inv_vars)
    base = DiagGmm.from_internal(w, ubm.means_invvars, ubm.inv_vars)
    models = [base, far]
    rng = np.random.default_rng(3)
    mu = ubm.means_invvars.astype(np.float64) * var
    ks = rng.integers(0, C, 400)
    rows = np.concatenate([mu[ks[:200]] + np.sqrt(var[ks[:200]]) * rng.standard_normal((200, D)),
                           mu_far[ks[200:]] + np.sqrt(var[ks[200:]]) * rng.standard_normal((200, D))]).astype(np.float32)
    want = _frame_lls(models, rows)
    e = Engine(0)
    try:
        e.load_gmm(models)
        assert e.gmm_kernel_variant.startswith("fx2w/")
        got = e.debug_gmm_frames(rows)
    finally:
        e.close()
    assert np.isfinite(got).all()
    # the far model's value is the base model's plus a delta of the same size: float32 accumulation rounds at the
    # larger of the two magnitudes (a frame near the far model is thousands of nats from the base one)
    rel = np.abs(got - want) / np.maximum(1.0, np.abs(want).max(axis=0, keepdims=True))
    print("far models: max error %.2e of the larger |ll| (up to %.1e)" % (rel.max(), np.abs(want).max()))
    assert rel.max() <= 5e-6
