"""Surface details of FakeBob.attack the reference has and round 2 refused or approximated: bits_per_sample in the
int16 casts (gmm_ubm_OSI.py:85, FAKEBOB.py:220) and the per-iteration times of the trace pickle (FAKEBOB.py:205-212)."""
import pickle
import time

import numpy as np
import pytest

from fakebob_amd.engine import nes_params
from fakebob_amd.models import synthetic_audio

pytestmark = pytest.mark.gpu


def test_bits_per_sample_scales_every_cast_of_the_nes_path(engine, small_system):
    """attack(bits_per_sample=b): every NES sample is cast with 2^(b-1) before scoring and so is the returned audio.
    The scoring entry point's cast is pinned to the reference's goldens (G5, incl. b = 8); the NES path must be the
    same function: column 0 of a get_grad batch scores exactly like score_raw on the same audio with the same b."""
    ubm, spk = small_system
    engine.load_gmm([ubm] + spk)
    engine.set_system("OSI")
    audio = synthetic_audio(3, 16000)
    seen = {}
    for bits in (16, 12, 8):
        p = nes_params("OSI", "targeted", samples_per_draw=6, target=1, threshold=0.0, seed=5, max_iter=3,
                       bits_per_sample=bits)
        fl, grad, al, sc = engine.get_grad(p, audio, it=0)
        raw, _ = engine.score_raw([audio], bits_per_sample=bits)
        assert np.array_equal(sc[:3], engine.system_scores(raw)[0])
        adv, flag, advf, tr = engine.attack(p, audio)
        assert np.array_equal(adv, engine.debug_quantize(advf, bits))              # FAKEBOB.py:220 with 2^(b-1)
        assert np.abs(adv).max() <= 2 ** (bits - 1)
        seen[bits] = sc[:3].copy()
    assert not np.array_equal(seen[16], seen[8])                                    # 8-bit audio is different audio
    bad = nes_params("OSI", "targeted", samples_per_draw=6, target=1, bits_per_sample=24)
    with pytest.raises(Exception):
        engine.get_grad(bad, audio, it=0)


def test_trace_pickle_carries_each_iterations_own_time(engine, small_system, tmp_path):
    """The reference times every loop body (FAKEBOB.py:205-212).  The loop runs on the device here, several
    iterations per host round trip, so the times come from the device clock where each iteration's loss is
    evaluated: positive, individually plausible, and together no longer than the call."""
    from fakebob_amd.attack import FakeBob
    from fakebob_amd.systems import gmm_OSI
    ubm, spk = small_system
    engine.load_gmm([ubm] + spk)
    engine.set_system("OSI")
    audio = synthetic_audio(2, 16000)
    p = nes_params("OSI", "targeted", samples_per_draw=10, target=0, threshold=5.0, seed=3, max_iter=9)  # unreachable
    t0 = time.time()
    adv, flag, advf, tr = engine.attack(p, audio)
    wall = time.time() - t0
    secs = engine.attack_iter_seconds(tr.shape[0])
    assert tr.shape[0] == 9 and secs.shape == (9,)
    assert np.all(secs > 0.0) and secs.sum() <= wall and secs.max() < 0.5 * wall + 0.05
    assert secs[1:].max() <= 20.0 * secs[1:].min() + 1e-3                 # iterations of one attack cost alike
    with pytest.raises(Exception):
        engine.attack_iter_seconds(10)
