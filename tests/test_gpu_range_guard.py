"""k_gmm_fx2 range guard (gmm_ubm_kaldiHelper.py:202-221 evaluates the same log-likelihoods in float32 with no
range limit): the two-term f16 split moves its operands by powers of two chosen from the MODEL at load time
(x * 2^4, x^2 * 2^kx2 with kx2 up to 4), so features with |x| >= 64 would overflow f16 and turn the scores
into NaN.  The kernel now rescales such frames per wave; these cases feed it features far outside the range
ordinary speech produces and compare with the oracle."""
import numpy as np
import pytest

from fakebob_amd.engine import Engine
from fakebob_amd.models import DiagGmm, stack_models, synthetic_gmm_system

pytestmark = pytest.mark.gpu


def _gated_tone(seed, n=16000, f0=3000.0, amp=0.9):
    """full-scale tone switched on and off every quarter second over a quiet noise floor: liftered cepstra of the
    tone frames sit tens of units away from the utterance mean"""
    t = np.arange(n)
    x = amp * np.sin(2 * np.pi * f0 * t / 16000.0) * ((t // 4000) % 2 == 0)
    x = x + 0.001 * np.random.default_rng(seed).normal(size=n)
    return (x * 32768.0).astype(np.int16)


def _small_variance_system(C):
    """UBM + 2 mean-adapted speakers (shared variances -> the shared quadratic item) in which ONE variance is tiny:
    max 1/(2 sigma^2) lands in [1024, 2048), the regime where fb_load_gmm picks x^2 * 2^4 (|x| < 64 without the
    guard)."""
    ubm, spk = synthetic_gmm_system(n_speakers=2, C=C, D=72, seed_ubm=3, seed_spk=30)
    var = ubm.variances()
    var[5, 30] = 3.5e-4
    out = []
    for g in [ubm] + spk:
        mu = g.means()
        mu[5, 30] = 0.1
        w = np.full(C, 1.0 / C)
        out.append(DiagGmm.from_moments(w, mu, var))
    half_inv = 0.5 * out[0].inv_vars.max()
    assert 1024.0 <= half_inv < 2048.0 and np.abs(out[0].means_invvars).max() < 1024.0
    return out


@pytest.mark.parametrize("C", [96, 80])      # 80: the last 32-component tile carries 16 padding components
def test_features_beyond_the_f16_range_score_like_the_oracle(oracle, C):
    models = _small_variance_system(C)
    e = Engine(0)
    try:
        e.set_frontend(cepstral_lifter=2000.0)            # lifter up to 37x instead of 12x
        e.load_gmm(models)
        assert e.gmm_kernel == "fx2"
        cfg = oracle.default_cfg(cepstral_lifter=2000.0)
        wavs = [_gated_tone(0), _gated_tone(1, f0=5200.0), _gated_tone(2, n=24000, f0=900.0)]
        feats = [oracle.frontend(cfg, w) for w in wavs]
        feats = [f[0] if isinstance(f, tuple) else f for f in feats]
        assert max(np.abs(f).max() for f in feats) > 64.0  # x^2 * 2^4 > 65504: inf without the guard
        raw_g, tv_g = e.score_raw(wavs)
        gc, miv, iv = stack_models(models)
        raw_o, tv_o = oracle.gmm_score_batch(cfg, wavs, gc, miv, iv)
        assert np.array_equal(tv_g, tv_o)
        assert np.all(np.isfinite(raw_g))
        # log-likelihoods here are O(-1e4 .. -1e5): float32 arithmetic carries ~1e-7 relative
        rel = np.abs(raw_g - raw_o) / np.abs(raw_o)
        print("range guard C=%d: raw in [%.4g, %.4g], max rel err %.3g" % (C, raw_o.min(), raw_o.max(), rel.max()))
        assert rel.max() <= 1e-6
        # the speaker-vs-UBM score differences survive at the same relative accuracy of the raw values
        assert np.abs((raw_g[:, 1:] - raw_g[:, :1]) - (raw_o[:, 1:] - raw_o[:, :1])).max() <= 2e-6 * np.abs(raw_o).max()
    finally:
        e.close()


def test_ordinary_features_are_unaffected_by_the_guard(engine, oracle, small_system):
    """sh = 0 for speech-like features: same scores as before, <= 1e-4 from the oracle (the guard only costs a
    wave-wide maximum in the prologue)."""
    from fakebob_amd.models import synthetic_audio
    ubm, spk = small_system
    engine.load_gmm([ubm] + spk)
    wavs = [(synthetic_audio(u, 16000) * 32768).astype(np.int16) for u in range(3)]
    raw_g, _ = engine.score_raw(wavs)
    gc, miv, iv = stack_models([ubm] + spk)
    raw_o, _ = oracle.gmm_score_batch(oracle.default_cfg(), wavs, gc, miv, iv)
    assert np.abs(raw_g - raw_o).max() <= 1e-4
