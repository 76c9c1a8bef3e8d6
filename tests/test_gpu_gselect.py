"""gmm-gselect --n=20 (ivector_PLDA_kaldiHelper.py:197-213) without the dump (round 6): the threshold selection in the
matrix-core kernels -- the wide form (k_gsel_w pass A -> k_gsel_tau -> pass B -> k_gsel_final_w: records of the groups that
reach the threshold, no overflow, no rescue) and the general form (k_gmm_fx2_sel ... k_gsel_final: lists of survivors) --
against the path they replace (every log-likelihood dumped, k_iv_select): the SAME selection, slot for slot, and
bit-identical i-vectors; the general form's rescue (lists too small for the survivors -> the dump + k_iv_select launches
redo the batch); a model whose component count is not a multiple of the 32-component tile."""
import numpy as np
import pytest

from fakebob_amd.engine import Engine
from fakebob_amd.models import synthetic_audio, synthetic_ivector_system

pytestmark = pytest.mark.gpu


def _wav(utt, n):
    return (synthetic_audio(utt, n) * 32768.0).astype(np.int16)


@pytest.fixture(scope="module")
def full_iv():
    return synthetic_ivector_system(C=2048, D=72, R=400, L=200, n_speakers=2)


def _run(engine, system, wavs, monkeypatch, **env):
    for k in ("FB_IV_GSEL_DUMP", "FB_GSEL_CAP", "FB_GSEL_NARROW", "FB_GSEL_A_HALF"):
        monkeypatch.delenv(k, raising=False)
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    llr, tv = engine.score_raw(wavs)
    sel, info = engine.debug_iv_gselect()
    ivs = engine.debug_ivectors(len(wavs), system.R)
    return llr, tv, sel, info, ivs


def test_threshold_selection_equals_the_dump_selection_at_full_size(full_iv, monkeypatch):
    e = Engine(0)
    try:
        e.load_ivector(full_iv, "OSI")
        rng = np.random.default_rng(11)
        batches = [[_wav(0, 48000), _wav(1, 31000), _wav(2, 11200), _wav(3, 72000), _wav(4, 16000), _wav(5, 52345)],
                   [_wav(7, 4000)],                                                    # one strip, eight chunks
                   [(rng.normal(size=20000) * 6000).astype(np.int16), _wav(9, 1700)]]  # noise: flat posteriors
        for wavs in batches:
            llr_d, tv_d, sel_d, info_d, ivs_d = _run(e, full_iv, wavs, monkeypatch, FB_IV_GSEL_DUMP="1")
            llr_n, tv_n, sel_n, info_n, ivs_n = _run(e, full_iv, wavs, monkeypatch, FB_GSEL_NARROW="1")
            assert info_n["path"] == 1 and info_n["overflow"] == 0 and np.array_equal(sel_d, sel_n)
            assert np.array_equal(ivs_d.view(np.uint64), ivs_n.view(np.uint64))
            llr_t, tv_t, sel_t, info_t, ivs_t = _run(e, full_iv, wavs, monkeypatch)
            assert not info_d["threshold_path"] and info_t["path"] == 2
            assert info_t["overflow"] == 0
            assert info_t["rows"] == int(np.sum(tv_t)) == sel_t.shape[0]
            assert np.array_equal(sel_d, sel_t)                           # the same 20 components in the same order
            assert sel_t.min() >= 0 and sel_t.max() < full_iv.C
            assert np.array_equal(ivs_d.view(np.uint64), ivs_t.view(np.uint64))
            assert np.array_equal(llr_d.view(np.uint64), llr_t.view(np.uint64))
            _, _, sel_f, info_f, ivs_f = _run(e, full_iv, wavs, monkeypatch, FB_GSEL_A_HALF="1")   # pass A over half the tiles
            assert info_f["path"] == 2 and np.array_equal(sel_d, sel_f) and np.array_equal(ivs_d.view(np.uint64), ivs_f.view(np.uint64))
            assert 20.0 <= info_f["survivors"] / info_f["rows"] <= 64.0
            per_row = info_t["survivors"] / info_t["rows"]
            print("threshold gselect: %d rows, %.1f records (wide; pass A over half the tiles: %.1f) / %.1f survivors (general) per row, most "
                  "records of a (row, chunk) %d" % (info_t["rows"], per_row, info_f["survivors"] / info_f["rows"],
                                                    info_n["survivors"] / info_n["rows"], info_t["max_list"]))
            assert 20.0 <= per_row <= 21.0 and info_t["max_list"] <= 32
    finally:
        e.close()


def test_overflowing_lists_take_the_rescue_launches(full_iv, monkeypatch):
    """FB_GSEL_CAP=2: two entries per (row, chunk) list -- nearly every row overflows, the flag goes up and the gated dump +
    k_iv_select launches redo the batch: the same selection as the dump path, and the flag is down again after a batch
    that fits."""
    e = Engine(0)
    try:
        e.load_ivector(full_iv, "SV" if False else "OSI")
        wavs = [_wav(0, 30000), _wav(1, 9000)]
        _, _, sel_d, _, ivs_d = _run(e, full_iv, wavs, monkeypatch, FB_IV_GSEL_DUMP="1")
        _, _, sel_r, info_r, ivs_r = _run(e, full_iv, wavs, monkeypatch, FB_GSEL_CAP="2", FB_GSEL_NARROW="1")
        assert info_r["path"] == 1 and info_r["overflow"] == 1 and info_r["max_list"] > 2
        assert np.array_equal(sel_d, sel_r)
        assert np.array_equal(ivs_d.view(np.uint64), ivs_r.view(np.uint64))
        _, _, sel_t, info_t, _ = _run(e, full_iv, wavs, monkeypatch, FB_GSEL_NARROW="1")
        assert info_t["overflow"] == 0 and np.array_equal(sel_d, sel_t)
    finally:
        e.close()


@pytest.mark.parametrize("C", [1300, 2560])
def test_threshold_selection_with_a_padded_last_tile_and_other_group_counts(C, monkeypatch):
    """C = 1300: 41 tiles, the last with 20 real components (padding must never be selected), 82 groups (k_gsel_tau<8>) --
    the general form; C = 2560: 160 groups (k_gsel_tau<16>), the wide form with 20 tiles per chunk."""
    sy = synthetic_ivector_system(C=C, D=72, R=64, L=32, n_speakers=1)
    e = Engine(0)
    try:
        e.load_ivector(sy, "OSI")
        wavs = [_wav(20 + u, 16000 + 3000 * u) for u in range(5)]
        _, _, sel_d, info_d, ivs_d = _run(e, sy, wavs, monkeypatch, FB_IV_GSEL_DUMP="1")
        _, _, sel_t, info_t, ivs_t = _run(e, sy, wavs, monkeypatch)
        assert info_t["path"] == (2 if C == 2560 else 1) and info_t["overflow"] == 0 and not info_d["threshold_path"]
        assert np.array_equal(sel_d, sel_t) and sel_t.max() < C
        assert np.array_equal(ivs_d.view(np.uint64), ivs_t.view(np.uint64))
    finally:
        e.close()



def test_ties_by_the_hundred_take_the_final_kernels_record_path(monkeypatch):
    """A degenerate UBM: components 0 .. 99 are copies of one component, so wherever that component is among a frame's best,
    a hundred values tie at or above the threshold -- more than k_gsel_final_w ranks through LDS (64): it ranks them straight
    from the records.  Equal values are ordered by component index, descending (std::greater<pair<float,int>>, as gmm-gselect
    and k_iv_select do): the same selection as the dump path, slot for slot."""
    from fakebob_amd.models import IvectorSystem
    sy = synthetic_ivector_system(C=2048, D=72, R=64, L=32, n_speakers=1)
    w, mic, icv = sy.fg_weights.copy(), sy.fg_means_invcovars.copy(), sy.fg_inv_covars.copy()
    e = Engine(0)
    try:
        e.load_ivector(sy, "OSI")
        wavs = [_wav(30 + u, 24000) for u in range(4)]
        _, _, sel0, _, _ = _run(e, sy, wavs, monkeypatch, FB_IV_GSEL_DUMP="1")
        src = int(np.bincount(sel0.ravel(), minlength=2048).argmax())         # the component most frames select
        w[:100], mic[:100], icv[:100] = w[src], mic[src], icv[src]
        dup = IvectorSystem(w, mic, icv, sy.ie_M, sy.ie_sigma_inv, sy.prior_offset, sy.mean_vec, sy.lda, sy.plda_mean,
                            sy.plda_transform, sy.plda_psi, sy.enrolled, sy.z_mean, sy.z_std, sy.num_gselect, sy.min_post)
        e.load_ivector(dup, "OSI")
        _, _, sel_d, info_d, ivs_d = _run(e, dup, wavs, monkeypatch, FB_IV_GSEL_DUMP="1")
        _, _, sel_t, info_t, ivs_t = _run(e, dup, wavs, monkeypatch)
        assert info_t["path"] == 2 and not info_d["threshold_path"] and info_t["overflow"] == 0
        tied_rows = int(np.sum(np.sum(sel_d < 100, axis=1) >= 2))
        assert tied_rows > 0                                                   # rows whose best 20 contain copies: >= 100 values tie
        assert np.array_equal(sel_d, sel_t)
        assert np.array_equal(ivs_d.view(np.uint64), ivs_t.view(np.uint64))
        rows20 = sel_d[np.sum(sel_d < 100, axis=1) == 20]
        if rows20.size:                                                        # all twenty are copies: the highest indices, descending
            assert np.array_equal(rows20[0], np.arange(99, 79, -1))
        print("degenerate UBM: %d of %d rows with tied copies among their best 20" % (tied_rows, sel_d.shape[0]))
    finally:
        e.close()
