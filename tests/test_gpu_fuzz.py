"""Randomised GPU-vs-oracle sweeps over model, batch and front-end shapes (fixed seeds; the longer versions of these
loops live in scratch/fuzz.py and scratch/fuzz_iv.py: 2 000 trials without a mismatch on the round-1 build)."""
import os

import numpy as np
import pytest

from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import stack_models, synthetic_audio, synthetic_gmm_system, synthetic_ivector_system

pytestmark = pytest.mark.gpu
TOL = 1e-4
FRONTENDS = [dict(), dict(), dict(delta_window=2), dict(delta_order=1, delta_window=2), dict(delta_order=0),
             dict(num_ceps=20, num_mel_bins=23, delta_order=3, delta_window=2), dict(num_ceps=13, num_mel_bins=23)]


def _wav(rng, n):
    return (synthetic_audio(int(rng.integers(0, 50)), n) * 32768.0).astype(np.int16)


@pytest.mark.parametrize("seed", [3, 4, 5])
def test_gmm_path_random_shapes(oracle, monkeypatch, seed):
    rng = np.random.default_rng(seed)
    for _ in range(8):
        over = FRONTENDS[rng.integers(len(FRONTENDS))]
        C = int(rng.choice([17, 33, 64, 100, 256]))
        S = int(rng.integers(2, 8))
        B = int(rng.choice([1, 3, 7, 16]))
        mode = str(rng.choice(["fx2", "fx2", "fx2", "bx3"]))
        monkeypatch.setenv("FB_GMM_MODE", mode)
        cfg = oracle.default_cfg(**over)
        e = Engine(0)
        try:
            e.set_frontend(**over)
            ubm, spk = synthetic_gmm_system(n_speakers=S, C=C, D=e.feat_dim)
            if rng.random() < 0.3:                      # own variances for some speakers: separate quadratic groups
                for m in spk[::2]:
                    m.inv_vars[:] = m.inv_vars * np.exp(rng.normal(scale=0.1, size=m.inv_vars.shape)).astype(np.float32)
            models = [ubm] + spk
            e.load_gmm(models)
            assert e.gmm_kernel == mode
            wavs = [_wav(rng, int(rng.choice([1600, 4000, 16000, 48000, 70000]))) for _ in range(B)]
            raw_g, tv_g = e.score_raw(wavs)
            raw_o, tv_o = oracle.gmm_score_batch(cfg, wavs, *stack_models(models), nthreads=8)
            info = dict(over=over, C=C, S=S, B=B, mode=mode)
            assert np.array_equal(tv_g, tv_o), info
            assert np.abs(raw_g - raw_o).max() <= TOL, info
            task = str(rng.choice(["OSI", "CSI", "SV"]))
            msel = models[:2] if task == "SV" else (spk if task == "CSI" else models)
            zm = rng.normal(-60, 2, size=len(spk)) if task == "CSI" else None
            zs = rng.uniform(1, 3, size=len(spk)) if task == "CSI" else None
            e.load_gmm(msel)
            e.set_system(task, zm, zs)
            ctx = oracle.GmmSystemCtx(cfg, task, *stack_models(msel), zm, zs, nthreads=8)
            kw = dict(samples_per_draw=int(rng.choice([2, 3, 8, 130])))
            if task != "CSI":
                kw["threshold"] = float(rng.normal(0, 0.1))
            if task != "SV":
                kw["target"] = int(rng.integers(0, ctx.S))
            audio = synthetic_audio(int(rng.integers(0, 50)), int(rng.choice([8000, 16000])))
            seed_n, it, st = int(rng.integers(1, 1000)), int(rng.integers(0, 9)), int(rng.integers(0, 5))
            pg = nes_params(task, "targeted", seed=seed_n, stream=st, **kw)
            po = oracle.nes_params(task, "targeted", ctx.S, **kw)
            flg, gg, alg, scg = e.get_grad(pg, audio, it=it)
            flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=seed_n, it=it, stream=st)
            info.update(task=task, **kw)
            assert abs(alg - alo) <= TOL and abs(flg - flo) <= TOL, info
            assert np.abs(scg[:ctx.S] - sco).max() <= TOL, info
            assert np.abs(gg - go).max() <= TOL * 6.0 / pg.sigma, info
        finally:
            e.close()


@pytest.mark.parametrize("seed", [6, 7])
def test_ivector_path_random_shapes(oracle, seed):
    rng = np.random.default_rng(seed)
    for _ in range(5):
        C = int(rng.choice([33, 64, 96, 130]))
        R = int(rng.choice([16, 40, 64, 100, 130]))            # 130: odd packed size -> register-staged contraction
        L = int(rng.choice([r for r in (8, 16, 24, 50) if r <= R]))
        S = int(rng.integers(1, 11))
        B = int(rng.choice([1, 2, 5, 9]))
        task = str(rng.choice(["OSI", "CSI", "SV"])) if S >= 2 else "SV"
        info = dict(C=C, R=R, L=L, S=S, B=B, task=task)
        e = Engine(0)
        try:
            sy = synthetic_ivector_system(C=C, D=72, R=R, L=L, n_speakers=S, seed=int(rng.integers(1, 1000)))
            zm, zs = list(rng.normal(-30, 5, size=S)), list(rng.uniform(2, 9, size=S))
            sy = sy.with_enrolled(sy.enrolled, z_mean=zm, z_std=zs)
            if task == "SV":
                sy = sy.with_enrolled(sy.enrolled[:1], zm[:1], zs[:1])
            e.load_ivector(sy, task)
            ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=8)
            wavs = [_wav(rng, int(rng.choice([9000, 16000, 30000, 48000]))) for _ in range(B)]
            llr_g, tv_g = e.score_raw(wavs)
            llr_o, ivs_o, tv_o = ctx.score_batch(wavs)
            ivs_g = e.debug_ivectors(len(wavs), sy.R)
            assert np.array_equal(tv_g, tv_o), info
            assert np.abs(ivs_g - ivs_o).max() <= 1e-6 * max(1.0, np.abs(ivs_o).max()), info
            assert np.abs(llr_g - llr_o).max() <= TOL, info
        finally:
            e.close()
