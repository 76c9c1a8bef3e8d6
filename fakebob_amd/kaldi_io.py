"""Readers/writers for the Kaldi on-disk objects the reference's pickled speaker models point at
(`final.dubm`, `*-identity.gmm`: attackMain.py:40-55, build_spk_models.py:146-224).

[EXT] The formats are Kaldi's (not in /root/reference): binary files start with "\\0B"; tokens
are space-terminated; int32 is written as a size byte (4) + little-endian value; a float vector
is "FV " + int32 dim + raw float32, a matrix "FM " + int32 rows + int32 cols + row-major data
("DV"/"DM" for float64).  DiagGmm: <DiagGMM> <GCONSTS> v <WEIGHTS> v <MEANS_INVVARS> m
<INV_VARS> m </DiagGMM>.  Real Kaldi files are unavailable offline, so these are validated by
round trip and against hand-written text fixtures only.
"""
import io
import os
import struct

import numpy as np

from .models import DiagGmm


class _Reader(object):
    def __init__(self, data):
        self.b = data
        self.i = 0
        self.binary = data[:2] == b"\x00B"
        if self.binary:
            self.i = 2

    def _skip_ws(self):
        while self.i < len(self.b) and self.b[self.i:self.i + 1] in b" \t\r\n":
            self.i += 1

    def token(self):
        self._skip_ws()
        j = self.i
        while j < len(self.b) and self.b[j:j + 1] not in b" \t\r\n":
            j += 1
        t = self.b[self.i:j].decode("ascii")
        self.i = j + 1 if self.binary else j
        return t

    def peek_token(self):
        i = self.i
        t = self.token()
        self.i = i
        return t

    def expect(self, tok):
        t = self.token()
        if t != tok:
            raise ValueError("kaldi_io: expected %s, got %r" % (tok, t))

    def int32(self):
        if self.binary:
            sz = self.b[self.i]
            if sz != 4:
                raise ValueError("kaldi_io: bad int size byte %d" % sz)
            v = struct.unpack_from("<i", self.b, self.i + 1)[0]
            self.i += 5
            return v
        return int(self.token())

    def vector(self):
        if self.binary:
            t = self.token()
            if t not in ("FV", "DV"):
                raise ValueError("kaldi_io: expected FV/DV, got %r" % t)
            dt = np.dtype("<f4") if t == "FV" else np.dtype("<f8")
            n = self.int32()
            v = np.frombuffer(self.b, dt, n, self.i).astype(np.float64)
            self.i += n * dt.itemsize
            return v
        self.expect("[")
        vals = []
        while True:
            t = self.token()
            if t == "]":
                break
            vals.append(float(t))
        return np.asarray(vals, np.float64)

    def matrix(self):
        if self.binary:
            t = self.token()
            if t not in ("FM", "DM"):
                raise ValueError("kaldi_io: expected FM/DM, got %r" % t)
            dt = np.dtype("<f4") if t == "FM" else np.dtype("<f8")
            r, c = self.int32(), self.int32()
            m = np.frombuffer(self.b, dt, r * c, self.i).astype(np.float64).reshape(r, c)
            self.i += r * c * dt.itemsize
            return m
        return np.asarray(self.text_rows(), np.float64)

    def text_rows(self):
        """Text-mode matrix body `[ r0 \n r1 ... ]` as a list of rows (rows may be ragged: a text SpMatrix)."""
        self.expect("[")
        rows = []
        while True:
            self._skip_ws()
            # rows are newline separated; the closing bracket ends the last row
            j = self.i
            while j < len(self.b) and self.b[j:j + 1] not in b"\n]":
                j += 1
            line = self.b[self.i:j].decode("ascii").split()
            if line:
                rows.append([float(x) for x in line])
            if j >= len(self.b):
                raise ValueError("kaldi_io: unterminated text matrix (no closing ']')")
            if self.b[j:j + 1] == b"]":
                self.i = j + 1
                break
            self.i = j + 1
        return rows


def read_diag_gmm(path_or_bytes):
    """Kaldi DiagGmm (binary or text) -> models.DiagGmm.  gconsts are recomputed from the stored
    weights / means_invvars / inv_vars exactly as DiagGmm::Read + ComputeGconsts does."""
    data = path_or_bytes
    if not isinstance(data, (bytes, bytearray)):
        with open(path_or_bytes, "rb") as r:
            data = r.read()
    rd = _Reader(bytes(data))
    t = rd.token()
    if t == "<DiagGMMBegin>":  # legacy alias
        t = "<DiagGMM>"
    if t != "<DiagGMM>":
        raise ValueError("kaldi_io: not a DiagGmm (first token %r)" % t)
    weights = miv = iv = None
    while True:
        t = rd.token()
        if t in ("</DiagGMM>", "<DiagGMMEnd>"):
            break
        if t == "<GCONSTS>":
            rd.vector()
        elif t == "<WEIGHTS>":
            weights = rd.vector()
        elif t == "<MEANS_INVVARS>":
            miv = rd.matrix()
        elif t == "<INV_VARS>":
            iv = rd.matrix()
        else:
            raise ValueError("kaldi_io: unexpected token %r in DiagGmm" % t)
    if weights is None or miv is None or iv is None:
        raise ValueError("kaldi_io: incomplete DiagGmm")
    return DiagGmm.from_internal(weights, miv.astype(np.float32), iv.astype(np.float32)), weights


def write_diag_gmm(path, gmm, weights, binary=True):
    """Inverse of read_diag_gmm (used for tests and to export synthetic systems)."""
    w = np.asarray(weights, np.float32)
    out = io.BytesIO()
    if binary:
        out.write(b"\x00B")

        def tok(s):
            out.write(s.encode("ascii") + b" ")

        def i32(v):
            out.write(b"\x04" + struct.pack("<i", int(v)))

        def vec(v):
            tok("FV"); i32(v.size); out.write(np.asarray(v, "<f4").tobytes())

        def mat(m):
            tok("FM"); i32(m.shape[0]); i32(m.shape[1]); out.write(np.ascontiguousarray(m, "<f4").tobytes())
        tok("<DiagGMM>"); tok("<GCONSTS>"); vec(gmm.gconsts); tok("<WEIGHTS>"); vec(w)
        tok("<MEANS_INVVARS>"); mat(gmm.means_invvars); tok("<INV_VARS>"); mat(gmm.inv_vars); tok("</DiagGMM>")
    else:
        def vec_t(v):
            return " [ " + " ".join("%.9g" % x for x in v) + " ]\n"

        def mat_t(m):
            return " [\n" + "\n".join("  " + " ".join("%.9g" % x for x in row) for row in m) + " ]\n"
        s = "<DiagGMM> \n<GCONSTS> " + vec_t(gmm.gconsts) + "<WEIGHTS> " + vec_t(w) + \
            "<MEANS_INVVARS> " + mat_t(gmm.means_invvars) + "<INV_VARS> " + mat_t(gmm.inv_vars) + "</DiagGMM> \n"
        out.write(s.encode("ascii"))
    with open(path, "wb") as wf:
        wf.write(out.getvalue())


# ------------------------------------------------------------ i-vector model files
def _read_file(path_or_bytes):
    if isinstance(path_or_bytes, (bytes, bytearray)):
        return bytes(path_or_bytes)
    with open(path_or_bytes, "rb") as r:
        return r.read()


def _sp(rd):
    """SpMatrix: binary "FP"/"DP" + int32 rows + packed lower-triangular data; text = the lower triangle, row i
    holding i + 1 numbers (PackedMatrix::Write) -- a full square text matrix is accepted as well."""
    if rd.binary:
        t = rd.token()
        if t not in ("FP", "DP"):
            raise ValueError("kaldi_io: expected FP/DP, got %r" % t)
        dt = np.dtype("<f4") if t == "FP" else np.dtype("<f8")
        n = rd.int32()
        cnt = n * (n + 1) // 2
        v = np.frombuffer(rd.b, dt, cnt, rd.i).astype(np.float64)
        rd.i += cnt * dt.itemsize
        return v, n
    rows = rd.text_rows()
    n = len(rows)
    if all(len(row) == i + 1 for i, row in enumerate(rows)):
        return np.array([x for row in rows for x in row], np.float64), n
    if all(len(row) == n for row in rows):
        return np.array([rows[i][j] for i in range(n) for j in range(i + 1)], np.float64), n
    raise ValueError("kaldi_io: text SpMatrix is neither lower-triangular nor square")


def _f64(rd):
    if rd.binary:
        sz = rd.b[rd.i]
        fmt = {4: "<f", 8: "<d"}[sz]
        v = struct.unpack_from(fmt, rd.b, rd.i + 1)[0]
        rd.i += 1 + sz
        return v
    return float(rd.token())


def read_vector(path):
    rd = _Reader(_read_file(path))
    return rd.vector()


def read_matrix(path):
    rd = _Reader(_read_file(path))
    return rd.matrix()


def read_full_gmm(path):
    """final.ubm -> (weights [C], means_invcovars [C,D], inv_covars packed [C, D(D+1)/2])."""
    rd = _Reader(_read_file(path))
    rd.expect("<FullGMM>")
    w = mic = None
    covs = []
    while True:
        t = rd.token()
        if t == "</FullGMM>":
            break
        if t == "<GCONSTS>":
            rd.vector()
        elif t == "<WEIGHTS>":
            w = rd.vector()
        elif t == "<MEANS_INVCOVARS>":
            mic = rd.matrix()
        elif t == "<INV_COVARS>":
            for _ in range(len(w)):
                covs.append(_sp(rd)[0])
        else:
            raise ValueError("kaldi_io: unexpected token %r in FullGmm" % t)
    return w, mic, np.stack(covs)


def read_ivector_extractor(path):
    """final.ie -> (M [C,D,R], Sigma_inv packed [C, D(D+1)/2], prior_offset).  Extractors with
    weight projections (<w> non-empty) are rejected: the voxceleb recipe trains without them."""
    rd = _Reader(_read_file(path))
    rd.expect("<IvectorExtractor>")
    rd.expect("<w>")
    wmat = rd.matrix()
    rd.expect("<w_vec>")
    rd.vector()
    if wmat.size:
        raise ValueError("kaldi_io: i-vector extractors with weight projections are unsupported")
    rd.expect("<M>")
    n = rd.int32()
    M = np.stack([rd.matrix() for _ in range(n)])
    rd.expect("<SigmaInv>")
    S = np.stack([_sp(rd)[0] for _ in range(n)])
    rd.expect("<IvectorOffset>")
    off = _f64(rd)
    rd.expect("</IvectorExtractor>")
    return M, S, off


def read_plda(path):
    rd = _Reader(_read_file(path))
    rd.expect("<Plda>")
    mean = rd.vector()
    tr = rd.matrix()
    psi = rd.vector()
    rd.expect("</Plda>")
    return mean, tr, psi


def read_ivector_location(loc):
    """identity_location of a speaker-model pickle: an array, or Kaldi's `file:offset` scp form
    pointing into a text/binary vector ark (build_spk_models.py:146-150)."""
    if isinstance(loc, np.ndarray):
        return loc.astype(np.float32)
    path, _, off = str(loc).rpartition(":")
    if not path or not off.isdigit():
        path, off = str(loc), "0"
    with open(path, "rb") as r:
        r.seek(int(off))
        data = r.read(1 << 20)
    if data[:2] == b"\x00B":
        return _Reader(data).vector().astype(np.float32)
    end = data.index(b"]")
    txt = data[:end].decode("ascii")
    txt = txt[txt.index("[") + 1:]
    return np.array(txt.split(), np.float32)


def load_ivector_pre_models(pre_model_dir):
    """pre-models/{final.ubm, final.ie, mean.vec, transform.mat, plda} -> dict of arrays."""
    def p(name):
        f = os.path.join(pre_model_dir, name)
        if not os.path.isfile(f):
            raise FileNotFoundError("i-vector system needs %s (README.md:71-81 of the reference)" % f)
        return f
    w, mic, covs = read_full_gmm(p("final.ubm"))
    M, S, off = read_ivector_extractor(p("final.ie"))
    mean, tr, psi = read_plda(p("plda"))
    return dict(fg_weights=w, fg_means_invcovars=mic, fg_inv_covars=covs, ie_M=M, ie_sigma_inv=S,
                prior_offset=off, mean_vec=read_vector(p("mean.vec")), lda=read_matrix(p("transform.mat")),
                plda_mean=mean, plda_transform=tr, plda_psi=psi)


def load_gmm_any(loc):
    """identity_location / ubm argument of the wrappers -> DiagGmm.  Accepts a DiagGmm, a Kaldi
    model file, or an .npz with gconsts / means_invvars / inv_vars."""
    if isinstance(loc, DiagGmm):
        return loc
    if isinstance(loc, (str, os.PathLike)):
        p = os.fspath(loc)
        if p.endswith(".npz"):
            z = np.load(p)
            return DiagGmm(z["gconsts"], z["means_invvars"], z["inv_vars"])
        return read_diag_gmm(p)[0]
    raise TypeError("cannot load a GMM from %r" % (loc,))


# ------------------------------------------------------------------ writers (binary Kaldi objects)
def _tok(t):
    return t.encode("ascii") + b" "


def _i32(v):
    return b"\x04" + struct.pack("<i", int(v))


def _vec_bytes(v, double=False):
    v = np.asarray(v).reshape(-1)
    return _tok("DV" if double else "FV") + _i32(v.size) + v.astype("<f8" if double else "<f4").tobytes()


def _mat_bytes(m, double=False):
    m = np.asarray(m)
    r, c = (m.shape if m.ndim == 2 else (0, 0))
    return _tok("DM" if double else "FM") + _i32(r) + _i32(c) + np.ascontiguousarray(m, "<f8" if double else "<f4").tobytes()


def _sp_bytes(packed, n, double=False):
    return _tok("DP" if double else "FP") + _i32(n) + np.asarray(packed).reshape(-1).astype("<f8" if double else "<f4").tobytes()


def write_vector(path, v, double=False):
    with open(path, "wb") as w:
        w.write(b"\x00B" + _vec_bytes(v, double))


def write_matrix(path, m, double=False):
    with open(path, "wb") as w:
        w.write(b"\x00B" + _mat_bytes(m, double))


def write_full_gmm(path, weights, means_invcovars, inv_covars):
    """final.ubm: FullGmm::Write (float32; gconsts are recomputed by every reader)."""
    C, D = np.asarray(means_invcovars).shape
    b = b"\x00B" + _tok("<FullGMM>") + _tok("<GCONSTS>") + _vec_bytes(np.zeros(C)) + _tok("<WEIGHTS>") + _vec_bytes(weights) \
        + _tok("<MEANS_INVCOVARS>") + _mat_bytes(means_invcovars) + _tok("<INV_COVARS>") \
        + b"".join(_sp_bytes(inv_covars[k], D) for k in range(C)) + _tok("</FullGMM>")
    with open(path, "wb") as w:
        w.write(b)


def write_ivector_extractor(path, M, sigma_inv, prior_offset):
    """final.ie: IvectorExtractor::Write without weight projections (float64)."""
    M = np.asarray(M, np.float64)
    C, D, _ = M.shape
    b = b"\x00B" + _tok("<IvectorExtractor>") + _tok("<w>") + _mat_bytes(np.zeros((0, 0)), True) + _tok("<w_vec>") \
        + _vec_bytes(np.zeros(C), True) + _tok("<M>") + _i32(C) + b"".join(_mat_bytes(M[k], True) for k in range(C)) \
        + _tok("<SigmaInv>") + b"".join(_sp_bytes(sigma_inv[k], D, True) for k in range(C)) \
        + _tok("<IvectorOffset>") + b"\x08" + struct.pack("<d", float(prior_offset)) + _tok("</IvectorExtractor>")
    with open(path, "wb") as w:
        w.write(b)


def write_plda(path, mean, transform, psi):
    with open(path, "wb") as w:
        w.write(b"\x00B" + _tok("<Plda>") + _vec_bytes(mean, True) + _mat_bytes(transform, True) + _vec_bytes(psi, True)
                + _tok("</Plda>"))


def write_ivector_pre_models(pre_model_dir, system):
    """Everything load_ivector_pre_models reads, from a models.IvectorSystem (synthetic sites, enrolment tools)."""
    os.makedirs(pre_model_dir, exist_ok=True)
    p = lambda n: os.path.join(pre_model_dir, n)   # noqa: E731
    write_full_gmm(p("final.ubm"), system.fg_weights, system.fg_means_invcovars, system.fg_inv_covars)
    write_ivector_extractor(p("final.ie"), system.ie_M, system.ie_sigma_inv, system.prior_offset)
    write_vector(p("mean.vec"), system.mean_vec)
    write_matrix(p("transform.mat"), system.lda)
    write_plda(p("plda"), system.plda_mean, system.plda_transform, system.plda_psi)
