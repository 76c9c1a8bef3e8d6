"""i-vector / PLDA path against the CPU oracle AT THE SIZE IT IS BENCHMARKED AT (BASELINE.json configs[2]:
C = 2048, D = 72, R = 400, LDA 200; ivector_PLDA_kaldiHelper.py:197-213, 251-280).

The kernels that only take their large-shape paths here -- the LDS-DMA T-matrix contraction with clamped
ragged column tiles (627 tiles at R = 400), the packed in-place blocked Cholesky with look-ahead (13 panels),
the component-bucketed full-covariance posteriors over all 2048 Gaussians -- are compared with the oracle on
ragged batches, and one complete NES gradient estimate (spd = 50, 3 s of audio, SV) with oracle.get_grad."""
import os

import numpy as np
import pytest

from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_ivector_system

pytestmark = pytest.mark.gpu
SCORE_TOL = 1e-4
IVEC_RTOL = 1e-6


def _wav(utt, n):
    return (synthetic_audio(utt, n) * 32768.0).astype(np.int16)


def _threads():
    try:
        return max(1, min(32, len(os.sched_getaffinity(0))))
    except AttributeError:
        return 8


@pytest.fixture(scope="module")
def full_iv():
    sy = synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=3)
    return sy.with_enrolled(sy.enrolled, z_mean=[-40.0, -35.0, -45.0], z_std=[10.0, 8.0, 12.0])


@pytest.fixture(scope="module")
def full_ctx(oracle, full_iv):
    return oracle.IvSystemCtx(oracle.default_cfg(), full_iv, nthreads=_threads())


@pytest.fixture(scope="module")
def full_engine():
    e = Engine(0)
    yield e
    e.close()


def test_full_size_ivectors_and_llr_ragged_batch(full_engine, full_ctx, full_iv):
    """6 utterances of 0.7 .. 4.5 s (T <= and > the 300-frame CMN window) -> i-vectors <= 1e-6 relative,
    PLDA LLRs <= 1e-4, identical voiced-frame counts."""
    full_engine.load_ivector(full_iv, "OSI")
    wavs = [_wav(0, 48000), _wav(1, 31000), _wav(2, 11200), _wav(3, 72000), _wav(4, 16000), _wav(5, 52345)]
    llr_g, tv_g = full_engine.score_raw(wavs)
    llr_o, ivs_o, tv_o = full_ctx.score_batch(wavs)
    ivs_g = full_engine.debug_ivectors(len(wavs), full_iv.R)
    assert np.array_equal(tv_g, tv_o)
    assert llr_g.shape == (6, 3)
    err_iv = np.abs(ivs_g - ivs_o).max() / max(1.0, np.abs(ivs_o).max())
    err_llr = np.abs(llr_g - llr_o).max()
    print("full-size i-vector: max rel err %.3g, LLR max abs err %.3g, active components %d"
          % (err_iv, err_llr, full_engine.debug_iv_active()))
    assert err_iv <= IVEC_RTOL
    assert err_llr <= SCORE_TOL
    # a batch of one takes the same kernels with 63 padding rows in the MFMA tiles
    llr_1, _ = full_engine.score_raw(wavs[3:4])
    assert np.abs(llr_1[0] - llr_o[3]).max() <= SCORE_TOL


def test_full_size_batch_of_65_crosses_the_utterance_group_boundary(full_engine, full_ctx, full_iv):
    """65 short utterances: two 64-utterance groups in the contraction (the second with one useful row)."""
    full_engine.load_ivector(full_iv, "OSI")
    wavs = [_wav(100 + u, 8000 + 160 * (u % 7)) for u in range(65)]
    llr_g, tv_g = full_engine.score_raw(wavs)
    llr_o, ivs_o, tv_o = full_ctx.score_batch(wavs)
    ivs_g = full_engine.debug_ivectors(len(wavs), full_iv.R)
    assert np.array_equal(tv_g, tv_o)
    assert np.abs(ivs_g - ivs_o).max() / max(1.0, np.abs(ivs_o).max()) <= IVEC_RTOL
    assert np.abs(llr_g - llr_o).max() <= SCORE_TOL


def test_full_size_sv_get_grad_spd50(oracle, full_engine, full_ctx, full_iv):
    """BASELINE configs[2] itself: i-vector-PLDA SV targeted, samples_per_draw = 50, 3 s @ 16 kHz -- one
    FakeBob.get_grad (FAKEBOB.py:223-246) against oracle.get_grad with the same Philox stream."""
    sv = full_iv.with_enrolled(full_iv.enrolled[:1], [-40.0], [10.0])
    full_engine.load_ivector(sv, "SV")
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sv, nthreads=_threads(), share=full_ctx)
    audio = synthetic_audio(7, 48000)
    kw = dict(samples_per_draw=50, threshold=1.0)
    pg = nes_params("SV", "targeted", seed=42, stream=3, **kw)
    po = oracle.nes_params("SV", "targeted", ctx.S, **kw)
    flg, gg, alg, scg = full_engine.get_grad(pg, audio, it=4)
    flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=42, it=4, stream=3)
    print("full-size SV get_grad: |final_loss err| %.3g, |adver_loss err| %.3g, |score err| %.3g"
          % (abs(flg - flo), abs(alg - alo), np.abs(scg[:1] - sco).max()))
    assert abs(alg - alo) <= SCORE_TOL and abs(flg - flo) <= SCORE_TOL
    assert np.abs(scg[:1] - sco).max() <= SCORE_TOL
    assert np.abs(gg - go).max() <= SCORE_TOL * 6.0 / pg.sigma
    big = np.abs(go) > 10 * SCORE_TOL / pg.sigma
    assert np.all(np.sign(gg[big]) == np.sign(go[big]))


def test_full_size_osi_get_grad_spd200_b201(oracle):
    """BASELINE configs[4]'s single-GPU share: i-vector-PLDA OSI, 10 enrolled speakers, samples_per_draw = 200 -- a NES
    batch of 201 utterances (four 64-row utterance groups through the T-matrix contraction, 201 posterior systems) --
    one FakeBob.get_grad against oracle.get_grad with the same Philox stream."""
    sy = synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=10)
    sy = sy.with_enrolled(sy.enrolled, z_mean=list(np.linspace(-45.0, -35.0, 10)), z_std=list(np.linspace(8.0, 12.0, 10)))
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=_threads())
    e = Engine(0)
    try:
        e.load_ivector(sy, "OSI")
        audio = synthetic_audio(3, 48000)
        kw = dict(samples_per_draw=200, target=7, threshold=-1.0)
        pg = nes_params("OSI", "targeted", seed=11, stream=2, **kw)
        po = oracle.nes_params("OSI", "targeted", ctx.S, **kw)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=1)
        flo, go, alo, sco = oracle.get_grad(po, ctx.fn, ctx.ctx, audio, seed=11, it=1, stream=2)
        print("full-size OSI B = 201 get_grad: |final_loss err| %.3g, |adver_loss err| %.3g, |score err| %.3g"
              % (abs(flg - flo), abs(alg - alo), np.abs(scg[:ctx.S] - sco).max()))
        assert abs(alg - alo) <= SCORE_TOL and abs(flg - flo) <= SCORE_TOL
        assert np.abs(scg[:ctx.S] - sco).max() <= SCORE_TOL
        assert np.abs(gg - go).max() <= SCORE_TOL * 6.0 / pg.sigma
        big = np.abs(go) > 10 * SCORE_TOL / pg.sigma
        assert np.all(np.sign(gg[big]) == np.sign(go[big]))
    finally:
        e.close()


@pytest.mark.parametrize("groups", ["5", "3", "2"])
def test_row_wise_solve_kernel_equals_the_one_workgroup_kernel(oracle, monkeypatch, groups):
    """k_iv_solve_rw (round 4: the block rows of a posterior system dealt over five workgroups -- three or two when a
    larger batch would not leave all of them resident, forced here through FB_IV_RW_G --, cross-workgroup exchange
    through agent-scope stores, progress words and sentinel-polled inverse factors) against k_iv_solve_ll (one workgroup
    per matrix) and the oracle at the benchmarked size (C = 2048, R = 400), over REPEATED launches (the slot sets
    alternate with the launch epoch), a batch of 51 and one of 3, and a switch between the kernels on one engine (the slot
    buffer is plain scratch for k_iv_solve_ll and must be refilled)."""
    from fakebob_amd.models import synthetic_ivector_system
    sy = synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=3)
    ctx = oracle.IvSystemCtx(oracle.default_cfg(), sy, nthreads=_threads())
    wavs51 = [_wav(u % 7, 30000 + 997 * (u % 5)) for u in range(51)]
    wavs3 = wavs51[:3]
    llr_o, ivs_o, tv_o = ctx.score_batch(wavs51)
    monkeypatch.setenv("FB_IV_RW_G", groups)
    e = Engine(0)
    try:
        e.load_ivector(sy, "OSI")
        got = {}
        for mode in ("ll", "rw", "rw", "ll", "rw"):
            monkeypatch.setenv("FB_IV_SOLVE", mode)
            for name, wavs in (("b51", wavs51), ("b3", wavs3), ("b51", wavs51)):
                llr, tv = e.score_raw(wavs)
                ivs = e.debug_ivectors(len(wavs), sy.R)
                got.setdefault((mode, name), []).append((llr, ivs))
        for (mode, name), runs in got.items():
            n = 51 if name == "b51" else 3
            for llr, ivs in runs:
                assert np.abs(ivs - ivs_o[:n]).max() <= 1e-9 * max(1.0, np.abs(ivs_o).max()), (mode, name)
                assert np.abs(llr - llr_o[:n]).max() <= 1e-7, (mode, name)
            for llr, ivs in runs[1:]:
                assert np.array_equal(llr, runs[0][0]) and np.array_equal(ivs, runs[0][1]), (mode, name)   # deterministic
        d = np.abs(got[("rw", "b51")][0][1] - got[("ll", "b51")][0][1]).max()
        print("row-wise against one-workgroup solve: max |i-vector difference| %.3g" % d)
        assert d <= 1e-10
    finally:
        monkeypatch.delenv("FB_IV_SOLVE", raising=False)
        monkeypatch.delenv("FB_IV_RW_G", raising=False)
        e.close()
