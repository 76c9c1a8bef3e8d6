"""Model containers and the synthetic workload of SURVEY.md section 8(d).

The reference's speaker models are Kaldi files (`final.dubm`,
`*-identity.gmm`, build_spk_models.py:166-224) that are not available here, so
benchmarks and tests run on seeded synthetic models of the same shape.  GMMs
are handed to the engine in Kaldi's DiagGmm *internal* form (gconsts,
means_invvars, inv_vars as float32), exactly what `gmm-global-get-frame-likes`
evaluates ([EXT] SURVEY.md A.7).
"""
import numpy as np


class DiagGmm(object):
    """Kaldi DiagGmm internal form (float32)."""

    def __init__(self, gconsts, means_invvars, inv_vars):
        self.gconsts = np.ascontiguousarray(gconsts, np.float32)
        self.means_invvars = np.ascontiguousarray(means_invvars, np.float32)
        self.inv_vars = np.ascontiguousarray(inv_vars, np.float32)
        assert self.means_invvars.shape == self.inv_vars.shape
        assert self.gconsts.shape == (self.inv_vars.shape[0],)

    @property
    def num_gauss(self):
        return self.inv_vars.shape[0]

    @property
    def dim(self):
        return self.inv_vars.shape[1]

    @staticmethod
    def from_moments(weights, means, variances):
        """DiagGmm::ComputeGconsts ([EXT] A.7): float64 math, float32 storage."""
        w = np.asarray(weights, np.float64)
        mu = np.asarray(means, np.float64)
        var = np.asarray(variances, np.float64)
        iv = (1.0 / var).astype(np.float32)
        miv = (mu / var).astype(np.float32)
        return DiagGmm.from_internal(w, miv, iv)

    @staticmethod
    def from_internal(weights, means_invvars, inv_vars):
        """gconst_k = log w_k - 0.5*(D log 2pi + sum log var + sum mu^2/var), evaluated from the
        stored float32 (means_invvars, inv_vars) like Kaldi does after reading a model."""
        w = np.asarray(weights, np.float64)
        miv = np.asarray(means_invvars, np.float32).astype(np.float64)
        iv = np.asarray(inv_vars, np.float32).astype(np.float64)
        D = iv.shape[1]
        gc = np.log(w) - 0.5 * D * np.log(2.0 * np.pi) + 0.5 * np.sum(np.log(iv), axis=1) \
            - 0.5 * np.sum(miv * miv / iv, axis=1)
        return DiagGmm(gc.astype(np.float32), miv.astype(np.float32), iv.astype(np.float32))

    def means(self):
        return self.means_invvars.astype(np.float64) / self.inv_vars.astype(np.float64)

    def variances(self):
        return 1.0 / self.inv_vars.astype(np.float64)


def stack_models(models):
    """[DiagGmm] -> (gconsts[M,C], means_invvars[M,C,D], inv_vars[M,C,D]) float32."""
    gc = np.ascontiguousarray(np.stack([m.gconsts for m in models]), np.float32)
    miv = np.ascontiguousarray(np.stack([m.means_invvars for m in models]), np.float32)
    iv = np.ascontiguousarray(np.stack([m.inv_vars for m in models]), np.float32)
    return gc, miv, iv


# ---------------------------------------------------------------- synthetic
def synthetic_audio(utt=0, n_samples=48000, seed=1234, fs=16000):
    """harmonic voiced-like signal with a 1/3 low-energy duty cycle so the VAD
    is exercised; int16-exact float64 in [-1, 1)."""
    rng = np.random.default_rng(seed + utt)
    phi = rng.uniform(0.0, 2.0 * np.pi, size=8)
    xi = rng.normal(size=n_samples)
    n = np.arange(n_samples, dtype=np.float64)
    x = np.zeros(n_samples)
    for h in range(1, 9):
        x += np.sin(2.0 * np.pi * 110.0 * h * n / fs + phi[h - 1]) / h
    env = np.where((np.arange(n_samples) // 8000) % 3 != 2, 1.0, 0.02)
    x = 0.25 * env * x + 0.002 * xi
    return np.round(x * 32768.0) / 32768.0


def _dim_scale(D):
    s = np.full(D, 0.5)
    s[:min(24, D)] = 3.0
    s[24:min(48, D)] = 1.0
    return s


def synthetic_ubm_moments(C=2048, D=72, seed=2001):
    rng = np.random.default_rng(seed)
    s = _dim_scale(D)
    logits = rng.normal(0.0, 0.5, size=C)
    w = np.exp(logits - logits.max())
    w /= w.sum()
    mu = rng.normal(size=(C, D)) * s
    var = (s ** 2) * np.exp(rng.normal(0.0, 0.3, size=(C, D)))
    return w, mu, var


def synthetic_speaker_means(w, mu, spk=0, seed=2100, tau=10.0, frames=200.0, occupancy_power=1.0):
    """Mean-only MAP adaptation (build_spk_models.py:170 update_flags 'm'; [EXT] A.8) from a
    synthetic enrolment of `frames` voiced frames (200) with uneven occupancy
    n_k ~ w_k u_k^occupancy_power, u ~ Exp(1): alpha_k = n_k / (n_k + tau) of the way to the enrolment mean."""
    rng = np.random.default_rng(seed + spk)
    C, D = mu.shape
    s = _dim_scale(D)
    delta = rng.normal(size=(C, D)) * (0.3 * s)
    u = rng.exponential(1.0, size=C) ** occupancy_power
    n = frames * w * u / np.sum(w * u)
    alpha = n / (n + tau)
    return mu + alpha[:, None] * delta


# enrolment sizes of the synthetic speakers: SURVEY.md 8(d)'s single 200-frame utterance (alpha ~ 0.01: the headline
# workload), and what build_spk_models.py:184-224 does with a speaker's whole enrolment set -- tens of thousands of
# frames, alpha ~ 0.2 - 0.4 where the data fell: the models the scoring kernel meets in a real deployment
ENROL_SURVEY = dict(enrol_frames=200.0, tau=10.0, occupancy_power=1.0)
ENROL_REALISTIC = dict(enrol_frames=20000.0, tau=10.0, occupancy_power=2.0)


def synthetic_gmm_system(n_speakers=5, C=2048, D=72, seed_ubm=2001, seed_spk=2100, enrol_frames=200.0, tau=10.0,
                         occupancy_power=1.0):
    """Returns (ubm, [speaker models]) as DiagGmm; speakers share weights and variances with
    the UBM (so the engine's shared-quadratic path applies, as for real MAP-adapted models).
    Defaults = ENROL_SURVEY; synthetic_gmm_system(**ENROL_REALISTIC) is the heavily enrolled variant."""
    w, mu, var = synthetic_ubm_moments(C, D, seed_ubm)
    ubm = DiagGmm.from_moments(w, mu, var)
    spk = []
    for s in range(n_speakers):
        mu_s = synthetic_speaker_means(w, mu, s, seed_spk, tau=tau, frames=enrol_frames, occupancy_power=occupancy_power)
        m = DiagGmm.from_internal(w, (mu_s / var).astype(np.float32), ubm.inv_vars)
        spk.append(m)
    return ubm, spk


# ------------------------------------------------------------ i-vector / PLDA
def tri_pack(A):
    """(..., n, n) symmetric -> Kaldi SpMatrix packed lower-triangular (..., n(n+1)/2)."""
    n = A.shape[-1]
    r, c = np.tril_indices(n)
    return np.ascontiguousarray(A[..., r, c])


def tri_unpack(p, n):
    r, c = np.tril_indices(n)
    A = np.zeros(p.shape[:-1] + (n, n), p.dtype)
    A[..., r, c] = p
    A[..., c, r] = p
    return A


class IvectorSystem(object):
    """Everything sid/extract_ivectors.sh + ivector-plda-scoring read from pre-models/ plus the
    enrolled i-vectors (ivector_PLDA_kaldiHelper.py:197-213, 251-280), in Kaldi's internal forms:
      full UBM     weights [C], means_invcovars [C,D], inv_covars packed [C, D(D+1)/2]   (float32)
      extractor    M [C,D,R], Sigma_inv packed [C, D(D+1)/2], prior_offset               (float64)
      back-end     mean_vec [R], lda [L, R or R+1] (float32); plda mean [L], transform [L,L], psi [L]
      enrolled     raw i-vectors [S,R] (float32), z-norm mean/std [S]
    """

    def __init__(self, fg_weights, fg_means_invcovars, fg_inv_covars, ie_M, ie_sigma_inv, prior_offset,
                 mean_vec, lda, plda_mean, plda_transform, plda_psi, enrolled, z_mean=None, z_std=None,
                 num_gselect=20, min_post=0.025):
        f32, f64 = np.float32, np.float64
        self.fg_weights = np.ascontiguousarray(fg_weights, f32)
        self.fg_means_invcovars = np.ascontiguousarray(fg_means_invcovars, f32)
        self.fg_inv_covars = np.ascontiguousarray(fg_inv_covars, f32)
        self.ie_M = np.ascontiguousarray(ie_M, f64)
        self.ie_sigma_inv = np.ascontiguousarray(ie_sigma_inv, f64)
        self.prior_offset = float(prior_offset)
        self.mean_vec = np.ascontiguousarray(mean_vec, f32)
        self.lda = np.ascontiguousarray(lda, f32)
        self.plda_mean = np.ascontiguousarray(plda_mean, f64)
        self.plda_transform = np.ascontiguousarray(plda_transform, f64)
        self.plda_psi = np.ascontiguousarray(plda_psi, f64)
        self.enrolled = np.ascontiguousarray(np.atleast_2d(enrolled), f32)
        S = self.enrolled.shape[0]
        self.z_mean = np.ascontiguousarray(np.zeros(S) if z_mean is None else z_mean, f64)
        self.z_std = np.ascontiguousarray(np.ones(S) if z_std is None else z_std, f64)
        self.num_gselect = int(num_gselect)
        self.min_post = float(min_post)
        self.C, self.D, self.R = self.ie_M.shape
        self.L = self.lda.shape[0]
        self.S = S
        assert self.fg_inv_covars.shape == (self.C, self.D * (self.D + 1) // 2)
        assert self.ie_sigma_inv.shape == self.fg_inv_covars.shape
        assert self.lda.shape[1] in (self.R, self.R + 1) and self.enrolled.shape[1] == self.R

    def with_enrolled(self, enrolled, z_mean=None, z_std=None):
        return IvectorSystem(self.fg_weights, self.fg_means_invcovars, self.fg_inv_covars, self.ie_M,
                             self.ie_sigma_inv, self.prior_offset, self.mean_vec, self.lda, self.plda_mean,
                             self.plda_transform, self.plda_psi, enrolled, z_mean, z_std, self.num_gselect,
                             self.min_post)


def synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=1, seed=3001, prior_offset=10.0):
    """Sigma_k = A A^T / D + diag(sigma_k^2), A ~ N(0, 0.3^2); M_k ~ N(0, 0.05^2);
    prior offset 10; mean.vec = 0; LDA = first L rows of a random orthogonal R x R; PLDA
    psi_i = 8 * 0.97^i, mean 0, transform I.  The UBM weights / means / diagonal variances are the
    synthetic diagonal UBM's.  Enrolled i-vectors ~ N(0, I) (replace them with engine-extracted
    ones via with_enrolled())."""
    w, mu, var = synthetic_ubm_moments(C, D, 2001)
    rng = np.random.default_rng(seed)
    inv_covars = np.empty((C, D * (D + 1) // 2), np.float64)
    mic = np.empty((C, D), np.float64)
    for k in range(C):
        A = rng.normal(0.0, 0.3, size=(D, D))
        Sig = A @ A.T / D + np.diag(var[k])
        P = np.linalg.inv(Sig)
        P = 0.5 * (P + P.T)
        inv_covars[k] = tri_pack(P)
        mic[k] = P @ mu[k]
    M = rng.normal(0.0, 0.05, size=(C, D, R))
    Q, _ = np.linalg.qr(rng.normal(size=(R, R)))
    lda = Q[:L]
    psi = 8.0 * 0.97 ** np.arange(L)
    enrolled = rng.normal(size=(n_speakers, R))
    f32 = inv_covars.astype(np.float32)
    return IvectorSystem(w, mic, f32, M, f32.astype(np.float64), prior_offset, np.zeros(R), lda, np.zeros(L),
                         np.eye(L), psi, enrolled)
