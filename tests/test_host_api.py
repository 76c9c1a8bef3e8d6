"""Host-side mirror of the reference interface (no GPU): C-ABI surface, wrapper input
normalisation / post-processing / decision rules against the golden vectors captured from the
reference's six wrappers (G5-G8), Kaldi file + conf readers."""
import json
import os
import re

import numpy as np
import pytest

from fakebob_amd import _native
from fakebob_amd.models import DiagGmm, synthetic_ubm_moments

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


# ------------------------------------------------------------------ C ABI
def test_library_loads_and_exports_every_declared_symbol():
    L = _native.lib()
    declared = set()
    for h in ("fakebob_hip.h", "fakebob_hip_test.h"):     # the drop-in boundary + the test / profiling hooks
        hdr = open(os.path.join(ROOT, "include", h)).read()
        hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
        names = set(re.findall(r"\b(fb_[a-z0-9_]+)\s*\(", hdr)) - {"fb_score_cb"}
        assert len(names) >= (18 if h == "fakebob_hip.h" else 8), (h, sorted(names))
        declared |= names
    for name in sorted(declared):
        assert hasattr(L, name), name
    assert set(_native.EXPORTS) == declared
    assert L.fb_version() >= 100


def test_no_gpu_means_a_loud_error_not_a_fallback():
    import ctypes as C
    L = _native.lib()
    if L.fb_device_count() > 0:
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = L.fb_engine_create(0, C.byref(h))
    assert rc != 0 and not h.value
    assert len(L.fb_last_error()) > 0
    from fakebob_amd.engine import Engine
    with pytest.raises(_native.NativeError):
        Engine(0)


def test_struct_layout_matches_header_order():
    hdr = open(os.path.join(ROOT, "include", "fakebob_hip.h")).read()
    body = hdr[hdr.index("typedef struct {", hdr.index("Kaldi front-end options")):hdr.index("} fb_frontend_cfg;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if not decl:
            continue
        typ, rest = decl.split(None, 1)
        names += [(typ, n.strip()) for n in rest.split(",")]
    assert [n for _, n in names] == [f[0] for f in _native.FrontendCfg._fields_]
    for (typ, n), (fn, ft) in zip(names, _native.FrontendCfg._fields_):
        assert {"double": "c_double", "int": "c_int"}[typ] == ft.__name__
    body = hdr[hdr.index("typedef struct {", hdr.index("FakeBob hyper-parameters")):hdr.index("} fb_nes_params;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for decl in body.split(";"):
        decl = decl.replace("typedef struct {", "").strip()
        if decl:
            names += [n.strip() for n in decl.split(None, 1)[1].split(",")]
    assert names == [f[0] for f in _native.NesParams._fields_]


# ----------------------------------------------------- wrappers vs goldens
class FakeEngine(object):
    """Captures what the wrappers hand to the engine and returns canned raw scores."""

    def __init__(self):
        self.ret = None
        self.audio = None
        self.bits = None

    def set_frontend(self, **kw):
        pass

    def load_gmm(self, models):
        self.n_models = len(models)

    def set_system(self, task, zm=None, zs=None):
        self.task, self.zm, self.zs = task, zm, zs

    def load_ivector(self, system, task):
        self.system, self.task, self.n_models = system, task, system.S

    @property
    def n_speakers(self):
        return self.n_models if self.task == "CSI" else self.n_models - 1

    def score_raw(self, lst, bits_per_sample=16):
        self.audio = [a.copy() for a in lst]
        self.bits = bits_per_sample
        return self.ret.copy(), np.ones(len(lst), np.int32)


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(G, "golden_meta.json")) as r:
        meta = json.load(r)["g5678"]
    return meta, np.load(os.path.join(G, "g5678_wrappers.npz"))


def _dummy_gmm(seed):
    w, mu, var = synthetic_ubm_moments(4, 72, seed=seed)
    return DiagGmm.from_moments(w, mu, var)


def _systems(meta, tmp_path):
    from fakebob_amd.systems import gmm_CSI, gmm_OSI, gmm_SV
    ml = [[m[0], m[1], _dummy_gmm(i), m[3], m[4]] for i, m in enumerate(meta["spk_models"])]
    ubm = _dummy_gmm(99)
    osi = gmm_OSI(str(tmp_path / "o"), ml, ubm, pre_model_dir=str(tmp_path), threshold=0.25, engine=FakeEngine())
    csi = gmm_CSI(str(tmp_path / "c"), ml, pre_model_dir=str(tmp_path), engine=FakeEngine())
    sv = gmm_SV(str(tmp_path / "s"), ml[0], ubm, pre_model_dir=str(tmp_path), threshold=0.25, engine=FakeEngine())
    return osi, csi, sv


def test_wrapper_attributes_and_model_order(gold, tmp_path):
    meta, z = gold
    osi, csi, sv = _systems(meta, tmp_path)
    assert osi.spk_ids == meta["gmm_spk_ids"] == csi.spk_ids      # CLI order, NOT sorted (unlike iv_*)
    assert len(osi.model_list) == len(meta["gmm_model_list"]) and osi.engine.n_models == 4
    assert len(csi.model_list) == len(meta["gmm_csi_model_list"]) and csi.engine.n_models == 3
    assert osi.engine.task == "OSI" and csi.engine.task == "CSI" and sv.engine.task == "SV"
    assert osi.threshold == 0.25 and osi.n_speakers == 3
    assert list(csi.engine.zm) == [m[3] for m in meta["spk_models"]]


def test_g5_input_normalisation(gold, tmp_path, oracle):
    """What reaches the scorer: one utterance for 1-D/(N,1)/(1,N), columns for (N,B), ragged lists
    with int16 passed through; the int16 cast itself (golden q arrays) is pinned on the oracle here
    and on the GPU kernel in test_gpu_parity.py."""
    meta, z = gold
    osi, _, _ = _systems(meta, tmp_path)
    e = osi.engine
    vals = z["g5_vals"]
    e.ret = np.zeros((1, 4))
    for arr in (vals, vals[:, None], vals[None, :]):
        osi.score(arr)
        assert len(e.audio) == 1 and np.array_equal(e.audio[0], vals)
    assert np.array_equal(oracle.quantize(vals), z["g5_q_1d"])
    assert np.array_equal(oracle.quantize(vals), z["g5_q_col"]) and np.array_equal(oracle.quantize(vals), z["g5_q_row"])
    assert np.array_equal(oracle.quantize(vals, bits=8), z["g5_q_bits8"])
    osi.score(vals, bits_per_sample=8)
    assert e.bits == 8
    mat = z["g5_mat"]
    e.ret = np.zeros((3, 4))
    osi.score(mat)
    assert len(e.audio) == 3
    for i in range(3):
        assert np.array_equal(e.audio[i], mat[:, i])
        assert np.array_equal(oracle.quantize(mat[:, i]), z["g5_q_mat_%d" % i])
    lst = [z["g5_list_in_%d" % i] for i in range(3)]
    keep = [a.copy() for a in lst]
    osi.score(lst)
    assert [a.dtype for a in e.audio] == [a.dtype for a in lst]           # int16 entry passed through
    for i in range(3):
        q = e.audio[i] if e.audio[i].dtype == np.int16 else oracle.quantize(e.audio[i])
        assert np.array_equal(q, z["g5_q_list_%d" % i])
        assert np.array_equal(lst[i], keep[i])                           # caller's arrays untouched
    assert meta["g5_list_input_unchanged"]


def test_g6_g7_gmm_postprocessing_and_decisions(gold, tmp_path):
    meta, z = gold
    osi, csi, sv = _systems(meta, tmp_path)
    x = np.zeros((8, 4))
    osi.engine.ret = z["g6_raw_osi"]
    assert np.array_equal(osi.score(x), z["g6_osi_scores"])
    dec, sc = osi.make_decisions(x)
    assert np.array_equal(np.array(dec), z["g7_osi_dec"]) and np.array_equal(sc, z["g7_osi_sc"])
    assert isinstance(dec, list)
    osi.engine.ret = z["g6_raw_osi"][:1]
    s1 = osi.score(np.zeros(8))
    assert s1.shape == (3,) and np.array_equal(s1, z["g6_osi_scores_b1"])
    osi.engine.ret = z["g6_raw_osi"][1:2]                 # max == threshold exactly: accepted (strict <)
    dec, sc = osi.make_decisions(np.zeros(8))
    assert dec == meta["g7_osi_b1_dec"] and np.array_equal(sc, z["g7_osi_b1_sc"]) and sc.ndim == 1
    sv.engine.ret = z["g6_raw_sv"]
    assert np.array_equal(sv.score(x), z["g6_sv_scores"])
    dec, sc = sv.make_decisions(x)
    assert np.array_equal(np.array(dec), z["g7_sv_dec"]) and np.array_equal(sc, z["g7_sv_sc"])
    sv.engine.ret = z["g6_raw_sv"][1:2]
    r = sv.score(np.zeros(8))
    assert np.ndim(r) == 0 and float(r) == float(z["g6_sv_b1"])
    dec, _ = sv.make_decisions(np.zeros(8))
    assert dec == meta["g7_sv_b1_dec"]
    csi.engine.ret = z["g6_raw_csi"]
    assert np.array_equal(csi.score(x), z["g6_csi_scores"])
    dec, sc = csi.make_decisions(x)
    assert np.array_equal(np.array(dec), z["g7_csi_dec"])
    csi.engine.ret = z["g6_raw_csi"][:1]
    dec, sc = csi.make_decisions(np.zeros(8))
    assert dec == meta["g7_csi_b1_dec"] and np.array_equal(sc, z["g7_csi_b1_sc"])


def test_iv_wrappers_reorder_speakers_and_match_goldens(gold, tmp_path):
    """ivector_PLDA_{OSI,CSI,SV}.py: speakers sorted by spk_id string, z-norm, decisions."""
    from fakebob_amd.models import synthetic_ivector_system
    from fakebob_amd.systems import iv_CSI, iv_OSI, iv_SV
    meta, z = gold
    base = synthetic_ivector_system(C=8, D=72, R=10, L=4, n_speakers=1)
    rng = np.random.default_rng(0)
    ml = [[m[0], m[1], rng.normal(size=10).astype(np.float32), m[3], m[4]] for m in meta["spk_models"]]
    osi = iv_OSI(str(tmp_path / "io"), ml, pre_model_dir=str(tmp_path), threshold=1.0, engine=FakeEngine(), system=base)
    csi = iv_CSI(str(tmp_path / "ic"), ml, pre_model_dir=str(tmp_path), engine=FakeEngine(), system=base)
    sv = iv_SV(str(tmp_path / "is"), ml[1], pre_model_dir=str(tmp_path), threshold=1.0, engine=FakeEngine(), system=base)
    assert osi.spk_ids == meta["iv_spk_ids"] and osi.utt_ids == meta["iv_utt_ids"]   # ['1580','2830','61']
    assert list(osi.z_norm_means) == meta["iv_z_means"] and list(osi.z_norm_stds) == meta["iv_z_stds"]
    # the engine received the enrolled vectors in the same (sorted) order
    want = np.stack([ml[i][2] for i in (0, 2, 1)])
    assert np.array_equal(osi.engine.system.enrolled, want)
    assert list(osi.engine.system.z_mean) == meta["iv_z_means"]
    x = np.zeros((8, 4))
    raw = z["g6_raw_iv"]
    osi.engine.ret = raw
    assert np.array_equal(osi.score(x), z["g6_iv_osi_scores"])
    dec, _ = osi.make_decisions(x)
    assert np.array_equal(np.array(dec), z["g7_iv_osi_dec"])
    osi.engine.ret = raw[:1]
    dec, sc = osi.make_decisions(np.zeros(8))
    assert dec == meta["g7_iv_osi_b1_dec"] and np.array_equal(sc, z["g7_iv_osi_b1_sc"])
    csi.engine.ret = raw
    assert np.array_equal(csi.score(x), z["g6_iv_csi_scores"])
    dec, _ = csi.make_decisions(x)
    assert np.array_equal(np.array(dec), z["g7_iv_csi_dec"])
    sv.engine.ret = raw[:, :1]
    assert np.array_equal(sv.score(x), z["g6_iv_sv_scores"])
    dec, _ = sv.make_decisions(x)
    assert np.array_equal(np.array(dec), z["g7_iv_sv_dec"])
    sv.engine.ret = raw[:1, :1]
    r = sv.score(np.zeros(8))
    assert np.ndim(r) == 0 and float(r) == float(z["g6_iv_sv_b1"])
    assert [sv.make_decisions_value(v) for v in (0.99, 1.0, 1.01)] == meta["g7_iv_sv_value"]


def test_kaldi_ivector_file_readers(tmp_path):
    """Round trip through hand-built Kaldi binary files ([EXT] formats; no real files offline)."""
    import struct
    from fakebob_amd import kaldi_io as K
    rng = np.random.default_rng(1)

    def tok(s):
        return s.encode() + b" "

    def i32(v):
        return b"\x04" + struct.pack("<i", v)

    def vec(v, d=False):
        return tok("DV" if d else "FV") + i32(v.size) + np.asarray(v, "<f8" if d else "<f4").tobytes()

    def mat(m, d=False):
        return tok("DM" if d else "FM") + i32(m.shape[0]) + i32(m.shape[1]) + np.ascontiguousarray(m, "<f8" if d else "<f4").tobytes()

    def sp(p, n, d=False):
        return tok("DP" if d else "FP") + i32(n) + np.asarray(p, "<f8" if d else "<f4").tobytes()
    C, D, R = 3, 4, 5
    tri = D * (D + 1) // 2
    w = rng.random(C).astype(np.float32)
    mic = rng.normal(size=(C, D)).astype(np.float32)
    covs = rng.normal(size=(C, tri)).astype(np.float32)
    ubm = b"\x00B" + tok("<FullGMM>") + tok("<GCONSTS>") + vec(np.zeros(C)) + tok("<WEIGHTS>") + vec(w) + \
        tok("<MEANS_INVCOVARS>") + mat(mic) + tok("<INV_COVARS>") + b"".join(sp(covs[k], D) for k in range(C)) + tok("</FullGMM>")
    (tmp_path / "final.ubm").write_bytes(ubm)
    M = rng.normal(size=(C, D, R))
    S = rng.normal(size=(C, tri))
    ie = b"\x00B" + tok("<IvectorExtractor>") + tok("<w>") + mat(np.zeros((0, 0)), True) + tok("<w_vec>") + vec(np.zeros(C), True) + \
        tok("<M>") + i32(C) + b"".join(mat(M[k], True) for k in range(C)) + tok("<SigmaInv>") + \
        b"".join(sp(S[k], D, True) for k in range(C)) + tok("<IvectorOffset>") + b"\x08" + struct.pack("<d", 12.5) + tok("</IvectorExtractor>")
    (tmp_path / "final.ie").write_bytes(ie)
    pm, pt, psi = rng.normal(size=3), rng.normal(size=(3, 3)), rng.random(3)
    (tmp_path / "plda").write_bytes(b"\x00B" + tok("<Plda>") + vec(pm, True) + mat(pt, True) + vec(psi, True) + tok("</Plda>"))
    (tmp_path / "mean.vec").write_bytes(b"\x00B" + vec(np.arange(R, dtype=np.float32)))
    (tmp_path / "transform.mat").write_text(" [\n  1 2 3 4 5 0.5\n  6 7 8 9 10 -0.5 ]\n")
    d = K.load_ivector_pre_models(str(tmp_path))
    assert np.array_equal(d["fg_weights"], w) and np.array_equal(d["fg_means_invcovars"], mic)
    assert np.array_equal(d["fg_inv_covars"], covs)
    assert np.array_equal(d["ie_M"], M) and np.array_equal(d["ie_sigma_inv"], S) and d["prior_offset"] == 12.5
    assert np.array_equal(d["plda_mean"], pm) and np.array_equal(d["plda_transform"], pt) and np.array_equal(d["plda_psi"], psi)
    assert np.array_equal(d["mean_vec"], np.arange(R)) and d["lda"].shape == (2, 6)
    # ivector.scp style `file:offset` into a text ark (build_spk_models.py:146-150)
    ark = tmp_path / "ivector.1.ark"
    ark.write_text("spk-utt  [ 0.5 -1.25 3 ]\nother  [ 1 2 ]\n")
    assert np.array_equal(K.read_ivector_location("%s:%d" % (ark, 8)), np.array([0.5, -1.25, 3], np.float32))
    assert np.array_equal(K.read_ivector_location("%s:%d" % (ark, 32)), np.array([1, 2], np.float32))
    with pytest.raises(FileNotFoundError):
        K.load_ivector_pre_models(str(tmp_path / "nope"))


# ------------------------------------------------------------- file readers
def test_kaldi_diag_gmm_round_trip(tmp_path):
    from fakebob_amd.kaldi_io import load_gmm_any, read_diag_gmm, write_diag_gmm
    w, mu, var = synthetic_ubm_moments(16, 72, seed=3)
    g = DiagGmm.from_moments(w, mu, var)
    for binary in (True, False):
        p = str(tmp_path / ("m_%d.gmm" % binary))
        write_diag_gmm(p, g, w, binary=binary)
        g2, w2 = read_diag_gmm(p)
        assert np.array_equal(g2.means_invvars, g.means_invvars) and np.array_equal(g2.inv_vars, g.inv_vars)
        assert np.allclose(w2, w, rtol=1e-6)
        assert np.abs(g2.gconsts - g.gconsts).max() < 1e-4
        assert np.array_equal(load_gmm_any(p).inv_vars, g.inv_vars)
    head = open(str(tmp_path / "m_1.gmm"), "rb").read(12)
    assert head.startswith(b"\x00B<DiagGMM> ")
    with pytest.raises(ValueError):
        read_diag_gmm(b"\x00B<FullGMM> ")


def test_kaldi_conf_parsing(tmp_path):
    from fakebob_amd.config import frontend_from_kaldi_conf, frontend_overrides
    mf = "--sample-frequency=16000\n--frame-length=25 # ms\n--low-freq=20\n--high-freq=7600\n" \
         "--num-mel-bins=30\n--num-ceps=24\n--snip-edges=false\n"
    vd = "--vad-energy-threshold=5.5\n--vad-energy-mean-scale=0.5\n--vad-proportion-threshold=0.12\n--vad-frames-context=2\n"
    o = frontend_overrides(mf, vd, "--delta-window=3 --delta-order=2\n")
    assert o["frame_length"] == 400 and o["padded_length"] == 512 and o["snip_edges"] == 0
    assert o["num_ceps"] == 24 and o["high_freq"] == 7600.0 and o["vad_frames_context"] == 2
    assert o["delta_window"] == 3 and o["delta_order"] == 2
    (tmp_path / "conf").mkdir()
    (tmp_path / "conf" / "mfcc.conf").write_text(mf)
    (tmp_path / "conf" / "vad.conf").write_text(vd)
    (tmp_path / "delta_opts").write_text("--delta-window=3 --delta-order=2\n")
    assert frontend_from_kaldi_conf(str(tmp_path)) == o
    with pytest.raises(ValueError):
        frontend_overrides("--window-type=hamming")
    for k in o:                                            # every override is a real struct field
        assert k in [f[0] for f in _native.FrontendCfg._fields_]


def test_fakebob_accepts_any_model_with_score_and_rejects_objects_without():
    """The reference's plugin API (README.md:136): any object with score / make_decisions is a model; it is
    driven through fb_attack_ext (tests/test_gpu_plugin_api.py).  Constructing needs no GPU."""
    from fakebob_amd.attack import FakeBob

    class Plain(object):
        spk_ids = ["a"]

        def score(self, a, **k):
            return 0.0
    fb = FakeBob("SV", "targeted", Plain(), verbose=False)
    assert fb._native is False and fb._speakers() == 1
    with pytest.raises(TypeError):
        FakeBob("SV", "targeted", object())
    with pytest.raises(ValueError):
        FakeBob("XYZ", "targeted", Plain())
    with pytest.raises(ValueError):                      # the container is int16: 2 .. 16 bits, nothing else
        fb.get_grad(np.zeros(1600), bits_per_sample=24)


def test_bench_roofline_object_and_traffic_file_follow_the_contract():
    """bench.py's roofline object: the keys the measurement contract names, algorithmic flops of SURVEY.md 8(d),
    and the committed PMC traffic file holds an entry for every GMM kernel variant and for the contraction."""
    import importlib.util
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("fb_bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    flops = (bench.S_SPK + 1) * bench.C_GAUSS * 4 * bench.D_FEAT * 15300
    assert flops == 6 * 2048 * 288 * 15300 and 54.0e9 < flops < 54.2e9      # 54.1 GFLOP per NES batch
    r = bench._gmm_roofline(flops_launch=flops, gmm_ms_avg=0.115, solo_ms=0.110, solo_rows=15300)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r
    # the peak is that of the pipe the kernel issues on (dense f16 / bf16 MFMA), so `frac` is a true fraction
    want_peak = 2500.0
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == want_peak
    assert abs(r["achieved"] - flops / 115e-6 / 1e12) < 1e-6
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0.0 < r["frac"] < 1.0
    assert r["frac"] < r["executed_frac"] < 1.0                           # utilisation of the pipe the MFMAs run on
    assert abs(r["solo_frac"] - flops / 110e-6 / 1e12 / want_peak) < 1e-9
    # per-tile products: the executed flops follow the tile histogram (P = 1 everywhere: (6 + 5) * 160 per (frame, component))
    r1 = bench._gmm_roofline(flops, 0.06, 0.06, 15300, "fx2w/1", (64, 0, 0))
    r3 = bench._gmm_roofline(flops, 0.06, 0.06, 15300, "fx2w/3", (0, 0, 64))
    rm = bench._gmm_roofline(flops, 0.06, 0.06, 15300, "fx2w/1", (32, 0, 32))
    assert abs(r1["executed_flops_per_launch"] / flops - 11 * 160 / 1728.0) < 1e-12
    assert abs(r3["executed_flops_per_launch"] / flops - 21 * 160 / 1728.0) < 1e-12
    assert abs(rm["executed_flops_per_launch"] - 0.5 * (r1["executed_flops_per_launch"] + r3["executed_flops_per_launch"])) < 1.0
    with open(os.path.join(root, "profiles", bench.TRAFFIC_FILE)) as f:
        tj = json.load(f)
    tr = tj["kernels"]
    for key in (bench.GMM_TRAFFIC_KEY, "k_iv_contract_dma<lin>+<quad>"):
        assert tr[key]["hbm_bytes_per_launch"] > 0
    # the committed PMC traffic is reported only for the kernel sources it was taken on (a hash of csrc/): a stale
    # profile gives None + traffic_stale, never another build's number
    assert len(tj["kernel_source_sha16"]) == 16
    val, prov = bench.committed_traffic(bench.GMM_TRAFFIC_KEY)
    if tj["kernel_source_sha16"] == bench.kernel_source_hash():
        assert val == tr[bench.GMM_TRAFFIC_KEY]["hbm_bytes_per_launch"] and prov["traffic_profiled_on"] == tj["kernel_source_sha16"]
    else:
        assert val is None and prov["traffic_stale"] is True


def test_kaldi_text_spmatrix_is_lower_triangular_and_writers_round_trip(tmp_path):
    """Kaldi writes a text SpMatrix as its lower triangle, row i holding i + 1 numbers (PackedMatrix::Write): a
    hand-written text final.ubm must load; and the binary writers round-trip a synthetic i-vector system through
    load_ivector_pre_models bit for bit."""
    from fakebob_amd import kaldi_io as K
    from fakebob_amd.models import synthetic_ivector_system
    txt = ("<FullGMM> <GCONSTS>  [ 0 0 ]\n<WEIGHTS>  [ 0.25 0.75 ]\n<MEANS_INVCOVARS>  [\n  1 2 3\n  4 5 6 ]\n"
           "<INV_COVARS>  [\n  1\n  0.5 2\n  -0.25 0.125 3 ]\n [\n  4\n  1 5\n  2 3 6 ]\n</FullGMM> ")
    w, mic, covs = K.read_full_gmm(txt.encode())
    assert list(w) == [0.25, 0.75] and mic.shape == (2, 3)
    assert np.array_equal(covs, np.array([[1, 0.5, 2, -0.25, 0.125, 3], [4, 1, 5, 2, 3, 6]], np.float64))
    # a full square text matrix is accepted too
    sq = "<FullGMM> <WEIGHTS>  [ 1 ]\n<MEANS_INVCOVARS>  [\n  1 2 ]\n<INV_COVARS>  [\n  2 0.5\n  0.5 3 ]\n</FullGMM> "
    assert np.array_equal(K.read_full_gmm(sq.encode())[2], np.array([[2, 0.5, 3]], np.float64))
    sy = synthetic_ivector_system(C=6, D=72, R=10, L=4, n_speakers=2, seed=2)
    K.write_ivector_pre_models(str(tmp_path / "pre"), sy)
    d = K.load_ivector_pre_models(str(tmp_path / "pre"))
    for k in ("fg_weights", "fg_means_invcovars", "fg_inv_covars", "ie_M", "ie_sigma_inv", "mean_vec", "lda",
              "plda_mean", "plda_transform", "plda_psi"):
        assert np.array_equal(np.asarray(d[k], getattr(sy, k).dtype), getattr(sy, k)), k
    assert d["prior_offset"] == sy.prior_offset


def test_kaldi_default_dither_is_reported():
    """Kaldi's default is --dither=1.0 and the stock voxceleb mfcc.conf does not set it: the loader must say that
    a real Kaldi run is random at the 1-LSB level (the engine always runs dither=0); --dither=0 is silent."""
    import warnings
    from fakebob_amd.config import frontend_overrides
    mf = "--sample-frequency=16000\n--frame-length=25\n--num-mel-bins=30\n--num-ceps=24\n--snip-edges=false\n"
    with pytest.warns(UserWarning, match="dither=1"):
        frontend_overrides(mf, "", "")
    with pytest.warns(UserWarning, match="dither=0.5"):
        frontend_overrides(mf + "--dither=0.5\n", "", "")
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        frontend_overrides(mf + "--dither=0\n", "", "")


def test_pipeline_option_precedence_and_dropin_defaults(monkeypatch):
    """text_scores / compress_feats: constructor keyword (True OR False) > FB_TEXT_SCORES / FB_COMPRESS_FEATS > the
    CLASS default -- None for the library classes (the flag is then not touched on the engine), both on for the
    subclasses the reference-named drop-in modules export (fakebob_amd/dropin/README.md).  Importing a drop-in module
    changes nothing for systems built from fakebob_amd.systems in the same process."""
    import subprocess
    import sys
    from fakebob_amd import systems
    for k in ("FB_TEXT_SCORES", "FB_COMPRESS_FEATS", "FB_MFCC_F32"):
        monkeypatch.delenv(k, raising=False)
    ref = systems.REFERENCE_PIPELINE
    assert systems._pipeline_options(None, None) == {}                        # nobody said anything: engine flags stay
    assert systems._pipeline_options(True, None) == {"text_scores": 1}
    assert systems._pipeline_options(None, False) == {"compress_feats": 0}   # an explicit False clears a shared engine's flag
    assert systems._pipeline_options(None, None, ref) == {"text_scores": 1, "compress_feats": 1, "mfcc_f32": 1}
    assert systems._pipeline_options(None, None, None, mfcc_f32=True) == {"mfcc_f32": 1}
    monkeypatch.setenv("FB_TEXT_SCORES", "1")
    monkeypatch.setenv("FB_COMPRESS_FEATS", "1")
    assert systems._pipeline_options(None, None) == {"text_scores": 1, "compress_feats": 1}
    assert systems._pipeline_options(False, False) == {"text_scores": 0, "compress_feats": 0}   # keyword wins, both ways
    monkeypatch.setenv("FB_TEXT_SCORES", "0")
    assert systems._pipeline_options(None, None, ref) == {"text_scores": 0, "compress_feats": 1, "mfcc_f32": 1}     # env wins over the class default
    sub = systems.reference_pipeline(systems.gmm_OSI)
    assert issubclass(sub, systems.gmm_OSI) and sub.__name__ == "gmm_OSI" and sub.PIPELINE == ref
    assert systems.gmm_OSI.PIPELINE is None and systems.iv_SV.PIPELINE is None
    # a fresh interpreter: the reference module names export subclasses; the library classes are untouched
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from fakebob_amd import systems\n"
            "import gmm_ubm_OSI, gmm_ubm_CSI, gmm_ubm_SV, ivector_PLDA_OSI, ivector_PLDA_CSI, ivector_PLDA_SV, FAKEBOB\n"
            "for mod, name in ((gmm_ubm_OSI, 'gmm_OSI'), (gmm_ubm_CSI, 'gmm_CSI'), (gmm_ubm_SV, 'gmm_SV'),\n"
            "                  (ivector_PLDA_OSI, 'iv_OSI'), (ivector_PLDA_CSI, 'iv_CSI'), (ivector_PLDA_SV, 'iv_SV')):\n"
            "    cls = getattr(mod, name)\n"
            "    assert issubclass(cls, getattr(systems, name)) and cls is not getattr(systems, name)\n"
            "    assert cls.PIPELINE == {'text_scores': True, 'compress_feats': True, 'mfcc_f32': True}\n"
            "    assert getattr(systems, name).PIPELINE is None\n"
            "assert systems._pipeline_options(None, None) == {}\n"
            "from fakebob_amd.attack import FakeBob\n"
            "assert FAKEBOB.FakeBob is FakeBob\n"
            % (os.path.join(root, "fakebob_amd", "dropin"), root))
    env = {k: v for k, v in os.environ.items() if k not in ("FB_TEXT_SCORES", "FB_COMPRESS_FEATS", "FB_MFCC_F32")}
    subprocess.run([sys.executable, "-c", code], check=True, env=env)


def test_class_default_float32_mfcc_falls_back_and_dropin_classes_pickle_by_reference(tmp_path, monkeypatch):
    """Round-4 advisor findings on the host side.  (1) The drop-in subclasses ask for Kaldi's float32 MFCC arithmetic by
    class default; a configuration outside the float32 kernel's shape (raw-energy=false, > 31 mel bins, ...) must then
    fall back to the float64 kernel with a warning -- it worked before the default existed --, while an explicit keyword
    or FB_MFCC_F32=1 keeps the error.  (2) A library system built on an engine whose pipeline flags an earlier drop-in
    system switched on warns that it inherits them.  (3) The drop-in classes carry their own module and qualified name,
    so they pickle by reference."""
    import pickle
    import sys
    import warnings
    from fakebob_amd import systems as S
    from fakebob_amd._native import NativeError

    class Cfg(object):
        text_scores = compress_feats = mfcc_f32 = 0

    class Eng(object):
        def __init__(self, f32_ok):
            self.cfg, self.calls, self.f32_ok = Cfg(), [], f32_ok

        def set_frontend(self, **kw):
            self.calls.append(dict(kw))
            if kw.get("mfcc_f32") and not self.f32_ok:
                raise NativeError(-1, "mfcc_f32 needs padded_length 512, raw_energy, ...")
            for k, v in kw.items():
                setattr(self.cfg, k, v)

    for k in ("FB_MFCC_F32", "FB_TEXT_SCORES", "FB_COMPRESS_FEATS"):
        monkeypatch.delenv(k, raising=False)
    e = Eng(f32_ok=False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        S._apply_frontend(e, {"raw_energy": 0}, None, None, None, S.REFERENCE_PIPELINE)
    assert len(e.calls) == 2 and e.calls[0]["mfcc_f32"] == 1 and e.calls[1]["mfcc_f32"] == 0 and e.calls[1]["raw_energy"] == 0
    assert e.cfg.mfcc_f32 == 0 and e.cfg.text_scores == 1 and any("float64 kernel" in str(x.message) for x in w)
    for kw, env in ((True, None), (None, "1")):            # asked for explicitly: the error stands
        e = Eng(f32_ok=False)
        if env:
            monkeypatch.setenv("FB_MFCC_F32", env)
        with pytest.raises(NativeError):
            S._apply_frontend(e, {"raw_energy": 0}, None, None, kw, S.REFERENCE_PIPELINE)
        monkeypatch.delenv("FB_MFCC_F32", raising=False)
        assert len(e.calls) == 1
    e = Eng(f32_ok=True)                                   # a shape the kernel takes: one call, no warning
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        S._apply_frontend(e, {}, None, None, None, S.REFERENCE_PIPELINE)
    assert len(e.calls) == 1 and e.cfg.mfcc_f32 == 1 and not w
    with warnings.catch_warnings(record=True) as w:       # a library class (no defaults) on that engine inherits the flags
        warnings.simplefilter("always")
        S._apply_frontend(e, {}, None, None, None, None)
    assert len(e.calls) == 1 and any("inherits" in str(x.message) for x in w)
    with warnings.catch_warnings(record=True) as w:       # ... and says what it wants when it names them
        warnings.simplefilter("always")
        S._apply_frontend(e, {}, False, False, False, None)
    assert e.cfg.mfcc_f32 == 0 and e.cfg.text_scores == 0 and not w

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fakebob_amd", "dropin"))
    try:
        import gmm_ubm_OSI as D1
        import ivector_PLDA_SV as D2
        for mod, name in ((D1, "gmm_OSI"), (D2, "iv_SV")):
            cls = getattr(mod, name)
            assert cls.__module__ == mod.__name__ and cls.__qualname__ == name and cls.PIPELINE == S.REFERENCE_PIPELINE
            assert pickle.loads(pickle.dumps(cls)) is cls
            assert issubclass(cls, getattr(S, name)) and getattr(S, name).PIPELINE is None
    finally:
        sys.path.pop(0)
        for m in ("gmm_ubm_OSI", "ivector_PLDA_SV"):
            sys.modules.pop(m, None)
