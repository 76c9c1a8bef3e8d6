"""GPU parity of the i-vector / PLDA path (K8-K12) against the CPU oracle."""
import numpy as np
import pytest

from fakebob_amd.engine import nes_params
from fakebob_amd.models import synthetic_audio, synthetic_ivector_system

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-4


def _wav(utt, n=48000):
    return (synthetic_audio(utt, n) * 32768.0).astype(np.int16)


@pytest.fixture(scope="module")
def small_iv():
    sy = synthetic_ivector_system(C=96, D=72, R=48, L=24, n_speakers=3, seed=11)   # C not a multiple of 64
    return sy.with_enrolled(sy.enrolled, z_mean=[-30.0, -50.0, -20.0], z_std=[5.0, 8.0, 4.0])


def test_ivector_extraction_and_plda_parity(engine, oracle, small_iv):
    engine.load_ivector(small_iv, "OSI")
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), small_iv)
    wavs = [_wav(0, 16000), _wav(1, 24000), _wav(2, 9000), _wav(3, 60000)]
    llr_g, tv_g = engine.score_raw(wavs)
    llr_o, ivs_o, tv_o = ctx.score_batch(wavs)
    assert np.array_equal(tv_g, tv_o)
    ivs_g = engine.debug_ivectors(len(wavs), small_iv.R)
    assert np.abs(ivs_g - ivs_o).max() <= 1e-6 * max(1.0, np.abs(ivs_o).max())
    assert np.abs(llr_g - llr_o).max() <= SCORE_TOL
    sc = engine.system_scores(llr_g)
    assert np.allclose(sc, (llr_g - small_iv.z_mean) / small_iv.z_std, rtol=0, atol=1e-12)


def test_ivector_larger_system(engine, oracle):
    sy = synthetic_ivector_system(C=256, D=72, R=100, L=50, n_speakers=2, seed=5)
    engine.load_ivector(sy, "CSI")
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=8)
    wavs = [_wav(u) for u in range(4)]
    llr_g, tv_g = engine.score_raw(wavs)
    llr_o, ivs_o, tv_o = ctx.score_batch(wavs)
    ivs_g = engine.debug_ivectors(len(wavs), sy.R)
    assert np.array_equal(tv_g, tv_o)
    assert np.abs(ivs_g - ivs_o).max() <= 1e-6 * max(1.0, np.abs(ivs_o).max())
    assert np.abs(llr_g - llr_o).max() <= SCORE_TOL


def test_full_covariance_kernels_agree(engine, oracle, monkeypatch):
    """k_iv_fullcov_mfma (round 4: a bucket's pairs as ONE product with the Cholesky factor of the precision matrix on the
    float64 matrix cores, then row sums of squares) against k_iv_fullcov_lds (the triangle form, thread = pair,
    FB_IV_FULLCOV=lds) and the oracle: the same posteriors up to float64 rounding, so the same i-vectors to 1e-9 --
    with buckets of every fill (C = 256 components over 4 utterances of different lengths: empty ones, tails of a few
    pairs, full 128-pair chunks)."""
    sy = synthetic_ivector_system(C=256, D=72, R=100, L=50, n_speakers=2, seed=5)
    engine.load_ivector(sy, "CSI")
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=8)
    wavs = [_wav(0), _wav(1, 9000), _wav(2, 30000), _wav(3, 1700)]
    llr_o, ivs_o, tv_o = ctx.score_batch(wavs)
    out = {}
    for name in ("mfma", "lds", "mfma"):
        if name == "lds":
            monkeypatch.setenv("FB_IV_FULLCOV", "lds")
        else:
            monkeypatch.delenv("FB_IV_FULLCOV", raising=False)
        llr, tv = engine.score_raw(wavs)
        out.setdefault(name, []).append((llr, engine.debug_ivectors(len(wavs), sy.R)))
        assert np.array_equal(tv, tv_o)
    monkeypatch.delenv("FB_IV_FULLCOV", raising=False)
    scale = max(1.0, np.abs(ivs_o).max())
    for name, runs in out.items():
        for llr, ivs in runs:
            assert np.abs(ivs - ivs_o).max() <= 1e-9 * scale, name
            assert np.abs(llr - llr_o).max() <= 1e-7, name
    assert np.array_equal(out["mfma"][0][1], out["mfma"][1][1])          # deterministic
    d = np.abs(out["mfma"][0][1] - out["lds"][0][1]).max()
    print("matrix-core against triangle-form full-covariance kernel: max |i-vector difference| %.3g" % d)
    assert d <= 1e-9 * scale


@pytest.mark.parametrize("task,attack,kw", [
    ("OSI", "targeted", dict(target=1, threshold=0.5)),
    ("SV", "targeted", dict(threshold=0.1)),
    ("CSI", "untargeted", dict(true=2)),
])
def test_ivector_get_grad_parity(engine, oracle, small_iv, task, attack, kw):
    sy = small_iv if task != "SV" else small_iv.with_enrolled(small_iv.enrolled[:1], [-30.0], [5.0])
    engine.load_ivector(sy, task)
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=8)
    audio = synthetic_audio(4, 16000)
    pg = nes_params(task, attack, samples_per_draw=8, seed=3, stream=1, **kw)
    po = oracle.nes_params(task, attack, ctx.S, samples_per_draw=8, **kw)
    flg, gg, alg, scg = engine.get_grad(pg, audio, it=2)
    flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=3, it=2, stream=1)
    assert abs(alg - alo) <= SCORE_TOL and abs(flg - flo) <= SCORE_TOL
    assert np.abs(scg[:ctx.S] - sco).max() <= SCORE_TOL
    assert np.abs(gg - go).max() <= SCORE_TOL * 6.0 / pg.sigma


def test_ivector_attack_trajectory(engine, oracle, small_iv):
    engine.load_ivector(small_iv, "OSI")
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), small_iv, nthreads=8)
    audio = synthetic_audio(6, 16000)
    kw = dict(samples_per_draw=8, max_iter=4, target=0, threshold=-10.0)
    pg = nes_params("OSI", "targeted", seed=11, stream=0, **kw)
    po = oracle.nes_params("OSI", "targeted", ctx.S, **kw)
    adv_g, flag_g, advf_g, tr_g = engine.attack(pg, audio)
    adv_o, flag_o, advf_o, tr_o = oracle.attack(po, ctx.fn, ctx.ctx, audio, seed=11, stream=0)
    assert flag_g == flag_o and tr_g.shape == tr_o.shape
    assert np.abs(tr_g - tr_o).max() <= SCORE_TOL
    # observed on MI355X: 0 differing samples (the update is sign(momentum gradient); a flip needs a gradient
    # entry within the 1e-4-scale score error of zero) -- asserted exactly, not as a rate
    assert int(np.sum(adv_g != adv_o)) == 0


@pytest.mark.parametrize("solve", ["ll", "rw"])
def test_solve_tail_equals_the_separate_backend_and_loss_launches(engine, oracle, monkeypatch, solve):
    """Round 5: the back-end of an utterance runs in the tail of the solve kernel's workgroup that holds its solution,
    and inside the NES loop the last of those workgroups to arrive runs the loss / loop-control body (fb_iv_tail.h).
    FB_IV_TAIL=split keeps the separate k_iv_backend / k_loss launches: the same operations in the same order, so scores,
    gradient estimates, attack traces and adversarial audio must be identical bit for bit -- for both solve kernels, a
    ragged scoring batch, get_grad and an attack that stops early (its queued iterations find the stop flag raised and
    the tail's arrival counter must come back to zero for the next attack)."""
    sy = synthetic_ivector_system(C=256, D=72, R=100, L=50, n_speakers=3, seed=5)
    sy = sy.with_enrolled(sy.enrolled, z_mean=[-30.0, -50.0, -20.0], z_std=[5.0, 8.0, 4.0])
    monkeypatch.setenv("FB_IV_SOLVE", solve)
    wavs = [_wav(0), _wav(1, 9000), _wav(2, 30000), _wav(3, 1700)]
    audio = synthetic_audio(4, 16000)
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=8)

    def run():
        engine.load_ivector(sy, "OSI")
        llr, tv = engine.score_raw(wavs)
        ivs = engine.debug_ivectors(len(wavs), sy.R)
        p = nes_params("OSI", "targeted", samples_per_draw=8, seed=3, stream=1, target=1, threshold=0.5)
        gg = engine.get_grad(p, audio, it=2)
        s0 = engine.system_scores(engine.score_raw([(audio * 32768).astype(np.int16)])[0])[0]
        # an attack that needs a few iterations: the target is the best speaker, the threshold just above its score
        tgt = int(np.argmax(s0))
        pa = nes_params("OSI", "targeted", samples_per_draw=8, seed=11, stream=0, max_iter=30, target=tgt,
                        threshold=float(s0[tgt]) + 0.02, epsilon=0.004, max_lr=0.002)
        a1 = engine.attack(pa, audio)
        a2 = engine.attack(pa, audio)          # back to back: the counters of the first are clean again
        return llr, ivs, gg, a1, a2

    monkeypatch.setenv("FB_IV_TAIL", "split")
    ref = run()
    monkeypatch.delenv("FB_IV_TAIL", raising=False)
    got = run()
    assert np.array_equal(ref[0], got[0]) and np.array_equal(ref[1], got[1])
    assert ref[2][0] == got[2][0] and ref[2][2] == got[2][2]
    assert np.array_equal(ref[2][1], got[2][1]) and np.array_equal(ref[2][3], got[2][3])
    for a in (got[3], got[4]):
        assert a[1] == ref[3][1] and np.array_equal(a[0], ref[3][0]) and np.array_equal(a[2], ref[3][2])
        assert np.array_equal(a[3], ref[3][3])
    print("attack of the tail test: flag %d after %d iterations" % (got[3][1], got[3][3].shape[0]))
    assert 1 < got[3][3].shape[0] <= 30
    llr_o, ivs_o, _ = ctx.score_batch(wavs)
    assert np.abs(got[0] - llr_o).max() <= 1e-7
    assert np.abs(got[1] - ivs_o).max() <= 1e-9 * max(1.0, np.abs(ivs_o).max())


def test_four_wave_fill_equals_the_one_wave_fill(engine, oracle, monkeypatch):
    """k_iv_bucket_fill4 (round 5: a partition block's 64 frames filled by four waves, 16 frames each, behind a count of
    the four sub-blocks) against the one-wave fill (FB_IV_FILL1=1): the partition of the (frame, slot) pairs by component
    is the same stable one, so the pairs -- and with them posteriors, statistics, i-vectors and scores -- are identical
    bit for bit.  Batches of 1 ... 40 utterances: 2 ... 90 blocks, ragged last blocks, sub-blocks without a frame."""
    sy = synthetic_ivector_system(C=256, D=72, R=100, L=50, n_speakers=2, seed=5)
    engine.load_ivector(sy, "CSI")
    batches = [[_wav(0)] + [_wav(u, 9000 + 700 * u) for u in range(1, 6)],
               [_wav(u % 7, 16000 + 1300 * (u % 5)) for u in range(40)],
               [_wav(3, 1700)]]
    for wavs in batches:
        monkeypatch.setenv("FB_IV_FILL1", "1")
        llr_a, tv_a = engine.score_raw(wavs)
        iv_a = engine.debug_ivectors(len(wavs), sy.R)
        monkeypatch.delenv("FB_IV_FILL1", raising=False)
        llr_b, tv_b = engine.score_raw(wavs)
        iv_b = engine.debug_ivectors(len(wavs), sy.R)
        assert np.array_equal(tv_a, tv_b) and np.array_equal(iv_a, iv_b) and np.array_equal(llr_a, llr_b)
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=8)
    llr_o, ivs_o, _ = ctx.score_batch(batches[0])
    llr_g, _ = engine.score_raw(batches[0])
    assert np.abs(llr_g - llr_o).max() <= 1e-7
