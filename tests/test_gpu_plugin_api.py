"""The reference's plugin API on the GPU path: FakeBob around a FOREIGN model (README.md:136 -- any object with
score / make_decisions; FAKEBOB.py:53,89,250), i.e. fb_get_grad_ext / fb_attack_ext with the scores coming back
from a Python callback and everything else of the NES iteration on the device.

These are DIRECT comparisons with golden vectors captured from the reference's own FAKEBOB.py
(tests/golden/make_golden.py): the same exactly reproducible SynthModel, the same injected np.random.normal
tensors -- no oracle in between.  Float64 / integer arithmetic throughout, so the bar is bit-exact equality."""
import json
import os
import pickle

import numpy as np
import pytest

from fakebob_amd.attack import FakeBob
from tests.golden.synth_model import SynthModel, synth_audio

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(G, "golden_meta.json")) as r:
        return json.load(r)


def _noise_stream(seed, N, half, count):
    rs = np.random.RandomState(seed)
    return [rs.normal(size=(N, half)) for _ in range(count)]


def test_g2_get_grad_through_the_plugin_api(meta):
    z = np.load(os.path.join(G, "g2_get_grad.npz"))
    for i, c in enumerate(meta["g2"]):
        model = SynthModel(c["task"], 5, c["N"], seed=c["model_seed"])
        audio = synth_audio(c["N"], c["audio_seed"])
        half = c["spd"] // 2
        noise = _noise_stream(c["noise_seed"], c["N"], half, 1)[0]
        fb = FakeBob(c["task"], c["attack"], model, adver_thresh=c["kappa"], samples_per_draw=c["spd"],
                     sigma=0.001, seed=1, verbose=False)
        fb.threshold, fb.target, fb.true = c["thr"], c["target"], c["true"]
        fl, grad, al, sc = fb.get_grad(audio, noise_pos=noise)
        assert fl == float(z["final_loss_%d" % i]), i
        assert al.shape == (1,) and al[0] == float(z["adver_loss_%d" % i].reshape(-1)[0])
        assert np.array_equal(np.asarray(sc).reshape(-1), z["score_%d" % i].reshape(-1))
        assert grad.shape == (c["N"], 1) and np.array_equal(grad[:, 0], z["grad_%d" % i].reshape(-1)), i
        assert model.n_calls == 1 and model.n_scored == 2 * half + 1      # ONE model.score call per get_grad (:250)


def test_g3_attack_trajectories_through_the_plugin_api(meta, tmp_path):
    z = np.load(os.path.join(G, "g3_attack.npz"))
    for i, c in enumerate(meta["g3"]):
        fbkw, at = c["fbkw"], c["atkw"]
        model = SynthModel(c["task"], 5, c["N"], seed=c["model_seed"])
        audio = z["audio_%d" % i] if c["custom_audio"] else synth_audio(c["N"], c["audio_seed"])
        half = fbkw["samples_per_draw"] // 2
        noise = np.stack(_noise_stream(c["noise_seed"], c["N"], half, fbkw["max_iter"]))
        fb = FakeBob(c["task"], c["attack"], model, seed=1, verbose=False, **fbkw)
        cp = str(tmp_path / ("cp_%d" % i))
        adv, flag = fb.attack(audio, cp, noise_all=noise, **at)
        want = z["trace_%d" % i]
        assert flag == c["flag"], c["name"]
        assert adv.dtype == np.int16 and adv.shape == tuple(c["adv_shape"])
        assert np.array_equal(adv, z["adv_%d" % i].reshape(adv.shape)), c["name"]
        assert model.n_calls == c["n_get_grad"], c["name"]
        with open(cp, "rb") as r:
            rows = pickle.load(r)                       # [distance, adver_loss (1,), score, used_time] per iteration
        assert len(rows) == c["n_rows"] == want.shape[0]
        assert [row[0] for row in rows] == list(want[:, 0]), c["name"]
        assert [float(row[1][0]) for row in rows] == list(want[:, 1]), c["name"]
        got_sc = np.array([np.asarray(row[2]).reshape(-1) for row in rows])
        assert np.array_equal(got_sc, want[:, 2:]), c["name"]
        assert (rows[-1][3] == 0.0) == bool(c["last_time_is_zero"])


def test_g4_estimate_threshold_through_the_plugin_api(meta):
    for c in meta["g4"]:
        fbkw = c["fbkw"]
        model = SynthModel(c["task"], 5, c["N"], seed=c["model_seed"], threshold=c["model_threshold"])
        audio = synth_audio(c["N"], c["audio_seed"])
        half = fbkw["samples_per_draw"] // 2
        noise = np.stack(_noise_stream(c["noise_seed"], c["N"], half, max(c["n_get_grad"], 1)))
        fb = FakeBob(c["task"], "targeted", model, seed=1, verbose=False, **fbkw)
        score, n_iters, _secs = fb.estimate_threshold(audio, noise_all=noise)
        assert n_iters == c["n_iters"] and score == c["score"] and fb.threshold == c["final_threshold"]
        assert fb.attack_type == c["attack_type_after"]
    assert FakeBob("CSI", "targeted", SynthModel("CSI", 5, 1600, seed=1), verbose=False).estimate_threshold(
        synth_audio(1600, 1)) is None


def test_philox_path_and_error_propagation():
    """Without injected noise the device Philox stream is used: same result as feeding those normals explicitly;
    an exception raised inside model.score surfaces unchanged."""
    from fakebob_amd.engine import Engine
    N, spd = 1601, 12
    model = SynthModel("OSI", 4, N, seed=5)
    audio = synth_audio(N, 6)
    fb = FakeBob("OSI", "targeted", model, samples_per_draw=spd, seed=77, verbose=False)
    fb.threshold, fb.target = 0.1, 2
    a = fb.get_grad(audio, iteration=9)
    e = Engine(0)
    zn = e.debug_noise(77, 9, 0, N, spd // 2).astype(np.float64).T.copy()
    e.close()
    b = fb.get_grad(audio, iteration=9, noise_pos=zn)
    assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
    adv, flag = fb.attack(audio, None, threshold=0.1, target=2)
    assert adv.shape == (N, 1) and flag in (1, -1)
    assert np.abs(adv[:, 0] / 32768.0 - audio).max() <= fb.epsilon + 1 / 32768.0

    class Broken(SynthModel):
        def score(self, audios, **kw):
            raise KeyError("model exploded")
    fbx = FakeBob("OSI", "targeted", Broken("OSI", 4, N, seed=5), samples_per_draw=4, verbose=False)
    fbx.target = 0
    with pytest.raises(KeyError):
        fbx.get_grad(audio)
