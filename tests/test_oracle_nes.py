"""Pins the oracle's NES engine (oracle/fb_oracle.c) against golden vectors captured from the
reference's own FAKEBOB.py (tests/golden/make_golden.py).  Everything here is float64 /
integer arithmetic, so the bar is bit-exact equality."""
import json
import os

import numpy as np
import pytest

from tests.golden.synth_model import SynthModel, synth_audio

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(G, "golden_meta.json")) as r:
        return json.load(r)


def _noise_stream(seed, N, half, count):
    """The tensors np.random.normal(size=(N, half)) returned to the reference, call by call."""
    rs = np.random.RandomState(seed)
    return [rs.normal(size=(N, half)) for _ in range(count)]


def test_np_sum_matches_numpy(oracle):
    rng = np.random.default_rng(0)
    for n in [0, 1, 3, 7, 8, 9, 49, 50, 51, 127, 128, 129, 200, 201, 1000, 4094]:
        for _ in range(20):
            a = rng.normal(size=n) * 10.0 ** rng.uniform(-3, 3, size=n)
            assert oracle.np_sum(a) == float(np.add.reduce(a)) if n else oracle.np_sum(a) == 0.0


def test_g1_loss_fn(oracle, meta):
    z = np.load(os.path.join(G, "g1_loss.npz"))
    for i, c in enumerate(meta["g1"]):
        score = z["score_%d" % c["idx"]]
        s = score[:, :1] if c["task"] == "SV" else score
        got = oracle.loss(c["task"], c["attack"], s, c["thr"], c["kappa"], c["target"], c["true"])
        want = z["loss_%d" % i].reshape(-1)
        assert np.array_equal(got, want), (i, c)


def test_g2_get_grad(oracle, meta):
    z = np.load(os.path.join(G, "g2_get_grad.npz"))
    for i, c in enumerate(meta["g2"]):
        model = SynthModel(c["task"], 5, c["N"], seed=c["model_seed"])
        audio = synth_audio(c["N"], c["audio_seed"])
        half = c["spd"] // 2
        noise = _noise_stream(c["noise_seed"], c["N"], half, 1)[0]
        p = oracle.nes_params(c["task"], c["attack"], model.S, adver_thresh=c["kappa"],
                              samples_per_draw=c["spd"], sigma=0.001, threshold=c["thr"],
                              target=c["target"], true=c["true"])
        fn = oracle.py_score_fn(model.score, model.S)
        fl, grad, al, sc = oracle.get_grad(p, fn, None, audio, noise_pos=noise)
        assert fl == float(z["final_loss_%d" % i])
        assert al == float(z["adver_loss_%d" % i].reshape(-1)[0])
        assert np.array_equal(sc, z["score_%d" % i].reshape(-1))
        assert np.array_equal(grad, z["grad_%d" % i].reshape(-1)), i
        assert model.n_scored == 2 * half + 1  # odd spd: only 2*(spd//2) noisy samples (:234-235)


def test_g3_attack_trajectories(oracle, meta):
    z = np.load(os.path.join(G, "g3_attack.npz"))
    for i, c in enumerate(meta["g3"]):
        fb, at = c["fbkw"], c["atkw"]
        model = SynthModel(c["task"], 5, c["N"], seed=c["model_seed"])
        audio = z["audio_%d" % i] if c["custom_audio"] else synth_audio(c["N"], c["audio_seed"])
        half = fb["samples_per_draw"] // 2
        noise = np.stack(_noise_stream(c["noise_seed"], c["N"], half, fb["max_iter"]))
        p = oracle.nes_params(c["task"], c["attack"], model.S, adver_thresh=fb["adver_thresh"],
                              epsilon=fb["epsilon"], max_iter=fb["max_iter"], max_lr=fb["max_lr"],
                              min_lr=fb["min_lr"], samples_per_draw=fb["samples_per_draw"], sigma=fb["sigma"],
                              momentum=fb["momentum"], plateau_length=fb["plateau_length"],
                              plateau_drop=fb["plateau_drop"], threshold=at.get("threshold", 0.0),
                              target=at.get("target"), true=at.get("true"))
        fn = oracle.py_score_fn(model.score, model.S)
        adv, flag, adv_f, trace = oracle.attack(p, fn, None, audio, noise_all=noise)
        want = z["trace_%d" % i]
        assert flag == c["flag"], c["name"]
        assert trace.shape[0] == c["n_rows"] == want.shape[0]
        assert model.n_calls == c["n_get_grad"]
        assert np.array_equal(trace[:, 0], want[:, 0]), c["name"]      # distance
        assert np.array_equal(trace[:, 1], want[:, 1]), c["name"]      # adver_loss
        assert np.array_equal(trace[:, 3:], want[:, 2:]), c["name"]    # score of the clean sample
        assert adv.dtype == np.int16 and np.array_equal(adv, z["adv_%d" % i].reshape(-1)), c["name"]
        lrs = z["lrs_%d" % i]                                          # printed with %f
        n_upd = lrs.shape[0]
        assert n_upd == c["n_rows"] - (1 if c["last_time_is_zero"] else 0)
        assert np.abs(trace[:n_upd, 2] - lrs).max() <= 5.1e-7 if n_upd else True
        if c["name"] == "plateau_to_min_lr":
            assert trace[-1, 2] == fb["min_lr"]


def test_g4_estimate_threshold(oracle, meta):
    for c in meta["g4"]:
        fb = c["fbkw"]
        model = SynthModel(c["task"], 5, c["N"], seed=c["model_seed"])
        audio = synth_audio(c["N"], c["audio_seed"])
        half = fb["samples_per_draw"] // 2
        noise = np.stack(_noise_stream(c["noise_seed"], c["N"], half, max(c["n_get_grad"], 1)))
        p = oracle.nes_params(c["task"], "targeted", model.S, samples_per_draw=fb["samples_per_draw"],
                              epsilon=fb.get("epsilon", 0.002))
        fn = oracle.py_score_fn(model.score, model.S)
        r = oracle.estimate_threshold(p, c["model_threshold"], fn, None, audio, noise_all=noise,
                                      max_total_iters=max(c["n_get_grad"], 1))
        score, n_iters, n_outer, thr_final, _ = r
        assert n_iters == c["n_iters"] and n_outer == c["n_outer"]
        assert score == c["score"] and thr_final == c["final_threshold"]
    # CSI: nothing to estimate (FAKEBOB.py:41-43 returns None)
    model = SynthModel("CSI", 5, 1600, seed=1)
    p = oracle.nes_params("CSI", "targeted", 5)
    assert oracle.estimate_threshold(p, 0.0, oracle.py_score_fn(model.score, 5), None, synth_audio(1600, 1)) is None


def test_philox_contract_replay_equals_explicit_noise(oracle):
    """noise_pos=None (Philox) must equal feeding the same float32 normals explicitly."""
    N, spd = 1601, 12
    model = SynthModel("OSI", 4, N, seed=5)
    audio = synth_audio(N, 6)
    p = oracle.nes_params("OSI", "targeted", 4, samples_per_draw=spd, target=2, threshold=0.1)
    fn = oracle.py_score_fn(model.score, 4)
    a = oracle.get_grad(p, fn, None, audio, seed=77, it=9, stream=4)
    z = oracle.noise(77, 9, 4, N, spd // 2).astype(np.float64).T.copy()
    b = oracle.get_grad(p, fn, None, audio, noise_pos=z)
    assert a[0] == b[0] and a[2] == b[2] and np.array_equal(a[1], b[1]) and np.array_equal(a[3], b[3])
