"""fb_frontend_cfg.mfcc_f32 = 1: compute-mfcc-feats in Kaldi's own precision (BaseFloat = float32, SURVEY.md A.2 /
A.11) -- k_mfcc_f32 against the oracle's float32 twin (fbo_mfcc with cfg.mfcc_f32: the same operations in the same
order, so the MFCC matrix must be BIT-identical) and against the float64 oracle (what the mode may cost: nothing the
decisions see -- C0, the VAD's only input, comes from the exact integer energy in both)."""
import numpy as np
import pytest

from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import stack_models, synthetic_audio

pytestmark = pytest.mark.gpu


def _wav(utt, n=48000):
    return (synthetic_audio(utt, n) * 32768.0).astype(np.int16)


@pytest.fixture()
def engine32():
    e = Engine(0)
    e.set_frontend(mfcc_f32=1)
    yield e
    e.close()


@pytest.mark.parametrize("path", ["default", "FB_MFCC_HALFWORDS", "FB_MFCC_RECORDS"])
def test_mfcc_f32_matrix_is_bit_identical_to_the_oracle_twin(engine32, oracle, path, monkeypatch):
    """A lone utterance takes the kernel's fast paths: frame records computed from the common utterance length, and one
    32-bit load per sample pair (every interior frame starts on an even sample).  FB_MFCC_HALFWORDS: the 16-bit loads
    that frames at an odd sample offset of a packed batch take; FB_MFCC_RECORDS: the per-frame records a batch of
    different lengths loads."""
    if path != "default":
        monkeypatch.setenv(path, "1")
    cfg32, cfg64 = oracle.default_cfg(mfcc_f32=1), oracle.default_cfg()
    rng = np.random.default_rng(5)
    wavs = [_wav(0), _wav(1, 16000), (rng.normal(size=30000) * 4000).astype(np.int16), np.zeros(8000, np.int16),
            rng.integers(-32768, 32767, size=12345).astype(np.int16), np.full(3000, -32768, np.int16),
            _wav(3, 400), _wav(4, 161), _wav(5, 90), _wav(6, 100000)]       # shorter than a frame: reflections of reflections
    wavs += [_wav(10 + k, int(n)) for k, n in enumerate(rng.integers(200, 60000, size=12))]
    for w in wavs:
        mg = engine32.debug_mfcc(w)
        mo = oracle.mfcc(cfg32, w)
        assert mg.shape == mo.shape
        same = mg.view(np.uint32) == mo.view(np.uint32)
        zero = (mg == 0.0) & (mo == 0.0)                                      # (+0 / -0 of an exact zero)
        assert np.all(same | zero), (w.size, int((~(same | zero)).sum()), np.abs(mg - mo).max())
        # against the float64-between-storage-points oracle: C0 (the VAD's input) identical, the cepstra within
        # float32 arithmetic noise of values up to ~50
        m64 = oracle.mfcc(cfg64, w)
        assert np.array_equal(mg[:, 0].view(np.uint32), m64[:, 0].view(np.uint32))
        assert np.abs(mg.astype(np.float64) - m64).max() <= 1e-5 * max(10.0, np.abs(m64).max()) + 2e-4


def test_mfcc_f32_batch_of_ragged_utterances_and_every_wave_slot(engine32, oracle, small_system):
    """A batch whose frame count is not a multiple of the 4 frames a wave takes, utterance edges inside groups."""
    ubm, spk = small_system
    engine32.load_gmm([ubm] + spk)
    cfg32 = oracle.default_cfg(mfcc_f32=1)
    wavs = [_wav(u, n) for u, n in enumerate([48000, 1610, 16000, 7777, 48000, 4000, 30001])]
    raw_g, tv_g = engine32.score_raw(wavs)
    gc, miv, iv = stack_models([ubm] + spk)
    raw_o, tv_o = oracle.gmm_score_batch(cfg32, wavs, gc, miv, iv, nthreads=8)
    assert np.array_equal(tv_g, tv_o)
    assert np.abs(raw_g - raw_o).max() <= 2e-5
    raw_64, tv_64 = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=8)
    assert np.array_equal(tv_g, tv_64)                 # the same frames are voiced in either precision
    # against the float64 front-end: what float32 MFCC arithmetic itself moves (the two oracles differ by as much) --
    # ~1e-5 on 3 s utterances, up to ~1e-4 on the 10-frame one, whose average has nothing to average over
    d_oracles = np.abs(raw_o - raw_64)
    assert np.abs(raw_g - raw_64).max() <= d_oracles.max() + 2e-5 and d_oracles.max() <= 3e-4
    long_ones = [i for i, w in enumerate(wavs) if w.size >= 16000]
    assert np.abs(raw_g - raw_64)[long_ones].max() <= 1e-4


def test_mfcc_f32_full_size_scores_and_get_grad(oracle, full_system):
    """configs[1] size (C = 2048, 5 speakers + UBM, spd = 50, 3 s) in the float32 mode: scores and the NES estimate
    against the oracle twin, and the scores against the float64 oracle within north_star's 1e-4."""
    ubm, spk = full_system
    models = [ubm] + spk
    gc, miv, iv = stack_models(models)
    e = Engine(0)
    try:
        e.set_frontend(mfcc_f32=1)
        e.load_gmm(models)
        e.set_system("OSI")
        wavs = [_wav(0), _wav(1), _wav(2, 20000), _wav(5, 30000)]
        raw_g, tv_g = e.score_raw(wavs)
        raw_o, tv_o = oracle.gmm_score_batch(oracle.default_cfg(mfcc_f32=1), wavs, gc, miv, iv, nthreads=8)
        raw_64, tv_64 = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv, nthreads=8)
        assert np.array_equal(tv_g, tv_o) and np.array_equal(tv_g, tv_64)
        assert np.abs(raw_g - raw_o).max() <= 2e-5, np.abs(raw_g - raw_o).max()
        assert np.abs(raw_g - raw_64).max() <= 1e-4, np.abs(raw_g - raw_64).max()
        print("float32 front-end, full size: max |err| vs the float32 twin %.3g, vs the float64 oracle %.3g (speaker - UBM %.3g)"
              % (np.abs(raw_g - raw_o).max(), np.abs(raw_g - raw_64).max(),
                 np.abs((raw_g[:, 1:] - raw_g[:, :1]) - (raw_64[:, 1:] - raw_64[:, :1])).max()))
        ctx = oracle.GmmSystemCtx(oracle.default_cfg(mfcc_f32=1), "OSI", gc, miv, iv, nthreads=32)
        audio = synthetic_audio(0, 48000)
        kw = dict(samples_per_draw=50, target=0, threshold=0.2277)
        pg = nes_params("OSI", "targeted", seed=42, stream=0, **kw)
        po = oracle.nes_params("OSI", "targeted", ctx.S, **kw)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=0)
        flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=42, it=0, stream=0)
        assert np.abs(scg[:ctx.S] - sco).max() <= 2e-5 and abs(alg - alo) <= 2e-5 and abs(flg - flo) <= 2e-5
        rms = float(np.sqrt(np.mean(go ** 2)))
        assert np.abs(gg - go).max() <= 2e-2 * rms
    finally:
        e.close()


def test_mfcc_f32_attack_trajectory_equals_the_oracle_twin(oracle, small_system):
    """The 1 s / spd 10 attack of tests/test_gpu_parity.py in the float32 mode: same success flag, iteration count and
    0 differing int16 samples against the oracle running its float32 twin."""
    ubm, spk = small_system
    models = [ubm] + spk
    gc, miv, iv = stack_models(models)
    e = Engine(0)
    try:
        e.set_frontend(mfcc_f32=1)
        e.load_gmm(models)
        e.set_system("OSI")
        ctx = oracle.GmmSystemCtx(oracle.default_cfg(mfcc_f32=1), "OSI", gc, miv, iv, nthreads=8)
        audio = synthetic_audio(6, 16000)
        kw = dict(samples_per_draw=10, max_iter=6, target=0, threshold=-1.0, epsilon=0.002)
        pg = nes_params("OSI", "targeted", seed=11, stream=0, **kw)
        po = oracle.nes_params("OSI", "targeted", ctx.S, **kw)
        adv_g, flag_g, advf_g, tr_g = e.attack(pg, audio)
        adv_o, flag_o, advf_o, tr_o = oracle.attack(po, ctx.fn, ctx.ctx, audio, seed=11, stream=0)
        assert flag_g == flag_o and tr_g.shape == tr_o.shape
        assert np.abs(tr_g - tr_o).max() <= 1e-4
        assert int(np.sum(adv_g != adv_o)) == 0
    finally:
        e.close()


def test_mfcc_f32_is_refused_for_shapes_the_kernel_does_not_take():
    from fakebob_amd import _native as N
    e = Engine(0)
    try:
        with pytest.raises(N.NativeError):
            e.set_frontend(mfcc_f32=1, padded_length=256, frame_length=200, frame_shift=80)
        e.set_frontend(mfcc_f32=0, padded_length=512, frame_length=400, frame_shift=160)
        with pytest.raises(N.NativeError):
            e.set_frontend(mfcc_f32=1, raw_energy=0)
    finally:
        e.close()


@pytest.mark.parametrize("bins,ceps", [(23, 13), (16, 13), (20, 20), (26, 24)])
def test_mfcc_f32_other_mel_bin_counts_are_bit_identical(oracle, bins, ceps):
    """Round-5 advisor finding: with fewer mel bins the filters are wider than 48 FFT bins (Kaldi's default 23 bins at
    16 kHz: 52 = five pieces of 12 weights; 20 bins: 58; 16 bins: 67 = six) and the kernel added four piece sums per
    filter at most.  Every piece is summed now; the matrix must equal the oracle twin's bit for bit."""
    e = Engine(0)
    try:
        e.set_frontend(mfcc_f32=1, num_mel_bins=bins, num_ceps=ceps)
        cfg32 = oracle.default_cfg(mfcc_f32=1, num_mel_bins=bins, num_ceps=ceps)
        rng = np.random.default_rng(bins)
        for w in [_wav(0), _wav(1, 16000), (rng.normal(size=30000) * 4000).astype(np.int16),
                  rng.integers(-32768, 32767, size=12345).astype(np.int16), _wav(3, 400)]:
            mg = e.debug_mfcc(w)
            mo = oracle.mfcc(cfg32, w)
            assert mg.shape == mo.shape
            same = mg.view(np.uint32) == mo.view(np.uint32)
            zero = (mg == 0.0) & (mo == 0.0)
            assert np.all(same | zero), (bins, w.size, int((~(same | zero)).sum()), np.abs(mg - mo).max())
    finally:
        e.close()


def test_mfcc_f32_refuses_filters_wider_than_it_sums():
    """8 mel bins over 20 - 8000 Hz: filters of more than 72 FFT bins (> 6 pieces) -- the float32 mode must be refused
    (the drop-in classes then fall back to the float64 kernel), never computed with pieces dropped."""
    from fakebob_amd._native import NativeError
    e = Engine(0)
    try:
        with pytest.raises(NativeError):
            e.set_frontend(mfcc_f32=1, num_mel_bins=8, num_ceps=8)
        assert e.cfg.mfcc_f32 == 0 and e.cfg.num_mel_bins != 8      # the mirror kept the previous configuration
        e.set_frontend(num_mel_bins=8, num_ceps=8)                  # the float64 kernel takes it
    finally:
        e.close()
