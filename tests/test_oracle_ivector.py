"""Validates the oracle's i-vector / PLDA restatement (parity unpinned at the Kaldi boundary) against
independent numpy / scipy formulas: Gaussian densities from moments, softmax, normal equations,
two-covariance PLDA likelihood ratio."""
import numpy as np
import pytest
from scipy.special import logsumexp
from scipy.stats import multivariate_normal

from fakebob_amd.models import synthetic_audio, synthetic_ivector_system, tri_unpack


@pytest.fixture(scope="module")
def sysm():
    sy = synthetic_ivector_system(C=48, D=72, R=30, L=12, n_speakers=3, seed=21)
    return sy.with_enrolled(sy.enrolled, [-10.0, -20.0, -5.0], [3.0, 4.0, 5.0])


@pytest.fixture(scope="module")
def ctx(oracle, sysm):
    return oracle.IvSystemCtx(oracle.default_cfg(), sysm)


def _feats(oracle, utt=0, n=16000):
    return oracle.frontend(oracle.default_cfg(), (synthetic_audio(utt, n) * 32768).astype(np.int16))[0]


def test_posteriors_and_stats_vs_numpy(oracle, sysm, ctx):
    feats = _feats(oracle)
    gamma, X = ctx.stats(feats)
    x = feats.astype(np.float64)
    C, D = sysm.C, sysm.D
    P = tri_unpack(sysm.fg_inv_covars.astype(np.float64), D)
    covar = np.linalg.inv(P)
    mean = np.einsum("kde,ke->kd", covar, sysm.fg_means_invcovars.astype(np.float64))
    w = sysm.fg_weights.astype(np.float64)
    # gmm-gselect on the diagonalised UBM (diag of the covariance, fgmm-global-to-gmm)
    var = np.einsum("kdd->kd", covar)
    dll = np.log(w)[None] - 0.5 * (D * np.log(2 * np.pi) + np.log(var).sum(1))[None] \
        - 0.5 * (((x[:, None, :] - mean[None]) ** 2) / var[None]).sum(2)
    g2 = np.zeros(C)
    X2 = np.zeros((C, D))
    for t in range(x.shape[0]):
        top = np.argsort(-dll[t])[:sysm.num_gselect]
        fl = np.array([np.log(w[k]) + multivariate_normal(mean[k], covar[k]).logpdf(x[t]) for k in top])
        post = np.exp(fl - logsumexp(fl)).astype(np.float32)
        jmax = int(np.argmax(post))
        post[post < np.float32(sysm.min_post)] = 0
        post = post / post.sum() if post.sum() > 0 else np.eye(len(top), dtype=np.float32)[jmax]
        g2[top] += post
        X2[top] += post[:, None] * x[t][None]
    assert abs(gamma.sum() - x.shape[0]) < 1e-4        # posteriors of every frame sum to one
    assert np.abs(gamma - g2).max() <= 2e-4 and np.abs(X - X2).max() <= 5e-3   # float32 parameter storage
    assert (gamma > 0).sum() < C


def test_extraction_is_the_normal_equation_solution(sysm, ctx, oracle):
    feats = _feats(oracle, 1)
    gamma, X = ctx.stats(feats)
    iv = ctx.extract(gamma, X)
    R = sysm.R
    D = sysm.D
    Sinv = tri_unpack(sysm.ie_sigma_inv, D)
    M = sysm.ie_M
    lin = np.zeros(R)
    Q = np.eye(R)
    for k in range(sysm.C):
        lin += M[k].T @ Sinv[k] @ X[k]
        Q += gamma[k] * (M[k].T @ Sinv[k] @ M[k])
    lin[0] += sysm.prior_offset
    want = np.linalg.solve(Q, lin)
    want[0] -= sysm.prior_offset
    assert np.abs(iv - want).max() <= 1e-9 * max(1.0, np.abs(want).max())
    # MAP point estimate: gradient of the auxiliary function vanishes
    assert np.abs(Q @ (iv + np.eye(R)[0] * sysm.prior_offset) - lin).max() < 1e-8


def test_backend_and_plda_llr_vs_closed_form(sysm, ctx):
    rng = np.random.default_rng(0)
    iv = rng.normal(size=sysm.R)
    y = ctx.backend(iv)
    L = sysm.L
    z = sysm.lda.astype(np.float64) @ (iv.astype(np.float32).astype(np.float64) - sysm.mean_vec)
    z *= np.sqrt(L) / np.linalg.norm(z)
    y2 = sysm.plda_transform @ (z - sysm.plda_mean)
    y2 *= np.sqrt(L / np.sum(y2 ** 2 / (sysm.plda_psi + 1.0)))
    assert np.abs(y - y2).max() < 1e-12
    assert abs(np.sum(y ** 2 / (sysm.plda_psi + 1.0)) - L) < 1e-9      # Kaldi's length normalisation
    import ctypes as C
    psi = sysm.plda_psi
    for s in range(sysm.S):
        tr = ctx.train[s]
        got = ctx_llr(ctx, tr, y)
        given = multivariate_normal(psi / (psi + 1.0) * tr, np.diag(1.0 + psi / (psi + 1.0))).logpdf(y)
        without = multivariate_normal(np.zeros(L), np.diag(1.0 + psi)).logpdf(y)
        assert abs(got - (given - without)) < 1e-9


def ctx_llr(ctx, tr, y):
    import ctypes as C
    from oracle import oracle as O
    tr = np.ascontiguousarray(tr)
    y = np.ascontiguousarray(y)
    return O.lib().fbo_plda_llr(C.byref(ctx.s), O._p(tr), O._p(y))


def test_score_batch_and_system_score(oracle, sysm, ctx):
    wavs = [(synthetic_audio(u, n) * 32768).astype(np.int16) for u, n in [(0, 16000), (1, 9000)]]
    llr, ivs, tv = ctx.score_batch(wavs)
    assert llr.shape == (2, 3) and ivs.shape == (2, sysm.R)
    for b, wv in enumerate(wavs):
        feats = oracle.frontend(oracle.default_cfg(), wv)[0]
        assert tv[b] == feats.shape[0]
        iv = ctx.extract(*ctx.stats(feats))
        assert np.array_equal(iv, ivs[b])
        y = ctx.backend(iv)
        assert np.allclose(llr[b], [ctx_llr(ctx, ctx.train[s], y) for s in range(3)], rtol=0, atol=1e-12)
    aud = np.stack([synthetic_audio(0, 16000), synthetic_audio(3, 16000)], axis=1)
    sc = ctx.score(aud)
    assert np.allclose(sc[0], (llr[0] - sysm.z_mean) / sysm.z_std, rtol=0, atol=1e-12)
