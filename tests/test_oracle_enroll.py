"""Oracle restatement of enrolment (gmm-global-acc-stats posteriors + means-only MAP) against independent
numpy / scipy math."""
import numpy as np
from scipy.special import logsumexp

from fakebob_amd.models import stack_models, synthetic_audio, synthetic_gmm_system


def test_acc_stats_match_numpy(oracle):
    ubm, _ = synthetic_gmm_system(1, 48, 72)
    gc, miv, iv = stack_models([ubm])
    cfg = oracle.default_cfg()
    wav = (synthetic_audio(5, 32000) * 32768.0).astype(np.int16)
    occ, F, tv = oracle.gmm_acc_stats(cfg, wav, gc[0], miv[0], iv[0])
    feats, T = oracle.frontend(cfg, wav)
    x = feats.astype(np.float64)
    ll = gc[0].astype(np.float64)[None, :] + x @ miv[0].astype(np.float64).T \
        - 0.5 * (x * x) @ iv[0].astype(np.float64).T
    post = np.exp(ll - logsumexp(ll, axis=1, keepdims=True))
    assert tv == x.shape[0]
    assert np.abs(occ - post.sum(axis=0)).max() <= 1e-4 * max(1.0, occ.max())
    assert np.abs(F - post.T @ x).max() <= 1e-4 * max(1.0, np.abs(F).max())
    assert abs(occ.sum() - tv) <= 1e-3


def test_map_update_formula(oracle):
    rng = np.random.default_rng(0)
    means = rng.normal(size=(6, 5))
    occ = np.array([0.0, 1e-3, 1.0, 10.0, 100.0, 0.0])
    xbar = rng.normal(size=(6, 5))
    F = occ[:, None] * xbar
    new = oracle.map_update_means(means, occ, F, tau=10.0)
    alpha = (occ / (occ + 10.0))[:, None]
    assert np.allclose(new, alpha * xbar + (1 - alpha) * means, rtol=0, atol=1e-14)
    assert np.array_equal(new[[0, 5]], means[[0, 5]])          # unoccupied components do not move
