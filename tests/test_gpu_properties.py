"""Size-independent properties at BASELINE.json's full sizes (C = 2048, D = 72, 3 s @ 16 kHz, spd = 50), where
the CPU oracle is too slow to be the checker: determinism, independence from batch composition and order,
agreement between the scoring entry points, and invariants of the NES step."""
import numpy as np
import pytest

from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import stack_models, synthetic_audio, synthetic_gmm_system, synthetic_ivector_system

pytestmark = pytest.mark.gpu


def _wav(utt, n=48000):
    return (synthetic_audio(utt, n) * 32768.0).astype(np.int16)


def test_gmm_scores_do_not_depend_on_batch_composition(engine, full_system):
    ubm, spk = full_system
    engine.load_gmm([ubm] + spk)
    wavs = [_wav(0), _wav(1, 30000), _wav(2), _wav(3, 70000), _wav(4, 16000)]
    raw_all, tv_all = engine.score_raw(wavs)
    perm = [3, 0, 4, 2, 1]
    raw_p, tv_p = engine.score_raw([wavs[i] for i in perm])
    assert np.array_equal(raw_p, raw_all[perm]) and np.array_equal(tv_p, tv_all[perm])   # bit-identical
    for i in (1, 3):
        # a different batch size changes how the component range is chunked over the chip, i.e. the order of the
        # float32 log-sum-exp partials: same numbers up to float32 rounding of a frame log-likelihood
        raw_1, tv_1 = engine.score_raw([wavs[i]])
        assert np.abs(raw_1[0] - raw_all[i]).max() <= 2e-6 and tv_1[0] == tv_all[i]
    # float input in [-1, 1) is the same utterance as its int16 samples (gmm_ubm_OSI.py:83-85)
    raw_f, _ = engine.score_raw([w.astype(np.float64) / 32768.0 for w in wavs[:2]])
    raw_i, _ = engine.score_raw(wavs[:2])                       # same batch shape -> same chunking -> same bits
    assert np.array_equal(raw_f, raw_i)


def test_nes_iteration_is_deterministic_and_consistent_with_scoring(full_system):
    ubm, spk = full_system
    audio = synthetic_audio(7, 48000)
    outs = []
    for rep in range(2):
        e = Engine(0)
        try:
            e.load_gmm([ubm] + spk)
            e.set_system("OSI")
            p = nes_params("OSI", "targeted", samples_per_draw=50, target=2, threshold=0.1, seed=42, stream=3)
            outs.append(e.get_grad(p, audio, it=5))
            if rep == 0:
                raw, _ = e.score_raw([(audio * 32768.0).astype(np.int16)])
                score0 = raw[0, 1:] - raw[0, 0]
        finally:
            e.close()
    (fl0, g0, al0, sc0), (fl1, g1, al1, sc1) = outs
    assert fl0 == fl1 and al0 == al1 and np.array_equal(g0, g1) and np.array_equal(sc0, sc1)   # bit-identical reruns
    assert np.abs(sc0[:5] - score0).max() <= 2e-6              # column 0 of the NES batch is the clean utterance
    others = np.delete(sc0[:5], 2)
    assert al0 == max(others.max(), 0.1) + 0.0 - sc0[2]         # OSI targeted loss of FAKEBOB.py:262 on those scores
    assert np.all(np.isfinite(g0)) and np.abs(g0).max() > 0


def test_attack_respects_the_linf_ball_and_reports_consistently(full_system):
    ubm, spk = full_system
    audio = synthetic_audio(9, 48000)
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        e.set_system("OSI")
        raw, _ = e.score_raw([(audio * 32768.0).astype(np.int16)])
        sc = raw[0, 1:] - raw[0, 0]
        target = int(np.argsort(sc)[-2])
        p = nes_params("OSI", "targeted", samples_per_draw=50, max_iter=30, target=target, epsilon=0.002,
                       threshold=float(sc.max()) - 0.02, seed=1, stream=0)
        adv, flag, adv_f, trace = e.attack(p, audio)
        adv2, flag2, adv_f2, trace2 = e.attack(p, audio)
    finally:
        e.close()
    assert flag == flag2 and np.array_equal(adv, adv2) and np.array_equal(trace, trace2)      # deterministic
    assert np.abs(adv_f - audio).max() <= 0.002 + 1e-15 and np.abs(adv_f).max() <= 1.0       # clip (:202-203)
    assert np.array_equal(adv, np.trunc(adv_f * 32768.0).astype(np.int64).astype(np.int16))  # final int16 cast (:220)
    n = trace.shape[0]
    assert 1 <= n <= 30 and (flag == 1) == (n < 30 or trace[-1, 1] < 0 and n - 1 < 29)
    assert np.all(np.diff(trace[:, 2]) <= 0)                    # the learning rate never grows
    assert np.all(trace[:, 0] <= 0.002 + 1e-12)                 # reported distance stays inside the ball
    if flag == 1:
        assert trace[-1, 1] < 0 and np.all(trace[:-1, 1] >= 0)  # stops at the first iteration with loss[0] < 0


def test_ivector_scores_do_not_depend_on_batch_composition():
    sy = synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=2)
    e = Engine(0)
    try:
        e.load_ivector(sy, "OSI")
        wavs = [_wav(0), _wav(1, 30000), _wav(2), _wav(3, 20000)]
        llr_all, tv_all = e.score_raw(wavs)
        perm = [2, 0, 3, 1]
        llr_p, tv_p = e.score_raw([wavs[i] for i in perm])
        assert np.array_equal(tv_p, tv_all[perm])
        assert np.array_equal(llr_p, llr_all[perm])             # no atomics anywhere: bit-identical
        llr_1, _ = e.score_raw([wavs[1]])
        assert np.abs(llr_1[0] - llr_all[1]).max() <= 1e-6      # (other chunking of the diagonal pre-selection)
    finally:
        e.close()


@pytest.mark.parametrize("pipeline", [(0, 0), (1, 0), (0, 1), (1, 1)], ids=["plain", "compress", "text", "compress+text"])
@pytest.mark.parametrize("how", ["env", "api"])
def test_fused_and_unfused_launch_chains_are_bit_identical(monkeypatch, pipeline, how):
    """The 5-launch NES chain (k_update_perturb, k_vad_delta_cmvn, k_gmm_finalize_loss) and the 8-launch chain
    (FB_NO_FUSE=1, or Engine.set_fused_chain(False) -- what bench.py and attack_main do with three or more attacks in
    flight) run the same arithmetic in the same orders: adversarial audio, float64 state and the whole trace must be
    bit-identical, early stop included -- also with the reference pipeline's round trips on (the drop-in modules'
    default), where the 8-launch chain runs the CompressedMatrix round trip as the stand-alone k_feat_compress and the
    fused chain as a phase of k_vad_delta_cmvn.  Repeated runs of the 8-launch chain must agree with each other too
    (k_feat_compress's header once depended on the block schedule)."""
    compress, text = pipeline
    ubm, spk = synthetic_gmm_system(n_speakers=3, C=128, D=72)
    audio = synthetic_audio(9, 16000)
    outs = []
    monkeypatch.delenv("FB_NO_FUSE", raising=False)
    for no_fuse in (False, True, True):
        if how == "env":
            if no_fuse:
                monkeypatch.setenv("FB_NO_FUSE", "1")
            else:
                monkeypatch.delenv("FB_NO_FUSE", raising=False)
        e = Engine(0)
        try:
            e.set_frontend(compress_feats=compress, text_scores=text)
            if how == "api":
                e.set_fused_chain(not no_fuse)
            e.load_gmm([ubm] + spk)
            e.set_system("OSI")
            s0 = e.system_scores(e.score_raw([(audio * 32768).astype(np.int16)])[0])[0]
            for kw in (dict(max_iter=9, threshold=float(s0.max()) + 0.5, target=int(np.argmin(s0))),     # runs to max_iter
                       dict(max_iter=40, threshold=float(s0.min()) - 1.0, target=int(np.argsort(s0)[1]))):  # stops early
                p = nes_params("OSI", "targeted", samples_per_draw=12, seed=17, stream=2, **kw)
                outs.append(e.attack(p, audio))
        finally:
            e.close()
    for k in (2, 4):
        for a, b in zip(outs[:2], outs[k:k + 2]):
            assert a[1] == b[1] and a[3].shape == b[3].shape
            assert np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    assert outs[1][3].shape[0] < 40                      # the second attack really stopped early


def test_feat_compress_kernel_is_deterministic_and_matches_the_fused_phase(oracle):
    """k_feat_compress (stand-alone, one workgroup per (utterance, 4 columns)) against the phase of k_vad_delta_cmvn
    (one workgroup per utterance) and against the oracle's fbo_compress_roundtrip: the same MFCC matrix bit for bit,
    many times over -- the header every workgroup reduces must come from the uncompressed matrix only."""
    ubm, spk = synthetic_gmm_system(n_speakers=1, C=64, D=72)
    wav = (synthetic_audio(4, 40000) * 32768).astype(np.int16)
    mats = []
    for fused in (True, False):
        e = Engine(0)
        try:
            e.set_frontend(compress_feats=1)
            e.set_fused_chain(fused)
            e.load_gmm([ubm] + spk)
            for _ in range(6 if not fused else 1):
                mats.append(e.debug_mfcc(wav))
        finally:
            e.close()
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        plain = e.debug_mfcc(wav)
    finally:
        e.close()
    want = oracle.compress_roundtrip(plain)
    for m in mats:
        assert np.array_equal(m.view(np.uint32), want.view(np.uint32))


def test_split_and_whole_utterance_vad_delta_cmvn_kernels_are_bit_identical(monkeypatch, oracle):
    """k_vad_delta_cmvn_p (an utterance over four workgroups that exchange their blocks of the CMVN column sums) against
    the one-workgroup kernel (FB_VAD_WHOLE=1) and the three-launch chain: the same features and scores bit for bit, on a
    ragged batch with utterances of 1, 2, 3, 5 frames (parts without frames) and of the full CMVN window."""
    ubm, spk = synthetic_gmm_system(n_speakers=2, C=64, D=72)
    lens = [160, 320, 480, 800, 1600, 4000, 16000, 30001, 48000, 47999, 7777]
    wavs = [(synthetic_audio(u, n) * 32768).astype(np.int16) for u, n in enumerate(lens)]
    got = {}
    for name, env, fused in (("split", {}, True), ("whole", {"FB_VAD_WHOLE": "1"}, True), ("three", {}, False)):
        monkeypatch.delenv("FB_VAD_WHOLE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        e = Engine(0)
        try:
            e.set_fused_chain(fused)
            e.load_gmm([ubm] + spk)
            raws = [e.score_raw(wavs) for _ in range(3)]                      # repeated: the exchange flags carry epochs
            feats = [e.debug_feats(w)[0] for w in (wavs[0], wavs[3], wavs[8])]
            got[name] = (raws, feats)
        finally:
            e.close()
    monkeypatch.delenv("FB_VAD_WHOLE", raising=False)
    for name in ("whole", "three"):
        for (ra, ta), (rb, tb) in zip(got["split"][0], got[name][0]):
            assert np.array_equal(ta, tb) and np.array_equal(ra, rb), name
        for fa, fb in zip(got["split"][1], got[name][1]):
            assert np.array_equal(fa.view(np.uint32), fb.view(np.uint32)), name
    gc, miv, iv = stack_models([ubm] + spk)
    raw_o, tv_o = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=8)
    assert np.array_equal(got["split"][0][0][1], tv_o) and np.abs(got["split"][0][0][0] - raw_o).max() <= 1e-4


def test_more_utterances_than_compute_units(oracle):
    """700 short utterances in one scoring batch: k_vad_delta_cmvn's workgroups (one per utterance, indices from a
    ticket, row offsets from the published counts of all earlier utterances) cannot all be resident at once."""
    ubm, spk = synthetic_gmm_system(n_speakers=2, C=64, D=72)
    e = Engine(0)
    try:
        e.load_gmm([ubm] + spk)
        wavs = [(synthetic_audio(u % 11, 4000 + 160 * (u % 13)) * (0.4 + 0.05 * (u % 9)) * 32768).astype(np.int16)
                for u in range(700)]
        for _ in range(2):                                # twice: the epoch tags of the first launch are stale now
            raw_g, tv_g = e.score_raw(wavs)
        gc, miv, iv = stack_models([ubm] + spk)
        raw_o, tv_o = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=16)
        assert np.array_equal(tv_g, tv_o)
        assert np.abs(raw_g - raw_o).max() <= 1e-4
    finally:
        e.close()
