"""float64 numpy model of k_gmm_fx2w's delta items: what an utterance average keeps of each class's rounding.

Classes: P = 1 / 2 / 3 (f16 partial products) and F6 -- the leading f16 product plus the two products P = 1 leaves out
(delta_2 . x_1, delta_1 . x_2) on block-scaled fp6 / fp4 operands (v_mfma_scale_f32_32x32x64_f8f6f4), with the block
layout and scale choice of fb_load_gmm / gmm_wide_kernel.hip: a lane's 32-value block = the K places 8 h .. 8 h + 7 of the
chunks 0 .. 3, one power-of-two scale per block, the smallest that brings the block's largest magnitude to <= 7.5 (e2m3)
resp. 6 (e2m1).  Variants of the class show where its error comes from (round 4: the parameters' rounding -- the same for
every frame -- is what an utterance average keeps; the frames' averages out):
  f6/1   delta_2 in one e2m3 term                      (the first form: 1.5e-5 on the realistic enrolment)
  f6     delta_2 in an e2m3 + an e2m1 term             (the kernel's)
  f6/66  delta_2 in two e2m3 terms                     (does not fit the 10 KB item)
The kernel itself was checked per frame against this model through fb_debug_gmm_frames: they agree to the float32
accumulation noise (2 - 3e-6 per frame).

  python tools/probes/f6_corr_emul.py [realistic|survey]
"""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
from fakebob_amd import models  # noqa: E402
from oracle import oracle  # noqa: E402

L2E = 1.4426950408889634


def f16(v):
    return np.asarray(v, np.float64).astype(np.float16).astype(np.float64)


def _quant(val, top, steps):
    """val (..., 32) -> nearest grid values with one power-of-two scale per row: the smallest e with max 2^-e <= top"""
    mx = np.abs(val).max(-1, keepdims=True)
    with np.errstate(divide="ignore"):
        ex = np.where(mx > 0, np.ceil(np.log2(np.where(mx > 0, mx, 1.0) / top)), -126.0)
    ex = np.where(mx * 2.0 ** -ex > top, ex + 1, ex)
    ex = np.where((mx > 0) & (mx * 2.0 ** -(ex - 1) <= top), ex - 1, ex)
    q = np.abs(val) * 2.0 ** -ex
    step = np.where(q < 2, steps[0], np.where(q < 4, steps[1], steps[2]))
    return np.sign(val) * np.minimum(np.rint(q / step) * step, top) * 2.0 ** ex


def q_e2m3(val):
    return _quant(val, 7.5, (0.125, 0.25, 0.5))


def q_e2m1(val):
    return _quant(val, 6.0, (0.5, 1.0, 2.0))


def q_frames(val):
    """the kernel's scale for the frames' blocks: exponent(max / 7.5) + 1 (one more where max / 7.5 is a power of two)"""
    mx = np.abs(val).max(-1, keepdims=True)
    t = (mx.astype(np.float32) * np.float32(0.13333334)).astype(np.float64)
    with np.errstate(divide="ignore"):
        ex = np.where(t > 0, np.floor(np.log2(np.where(t > 0, t, 1.0))) + 1, -126.0)
    q = np.abs(val) * 2.0 ** -ex
    step = np.where(q < 2, 0.125, np.where(q < 4, 0.25, 0.5))
    return np.sign(val) * np.minimum(np.rint(q / step) * step, 7.5) * 2.0 ** ex


def lse(a):
    mx = a.max(-1, keepdims=True)
    return (mx + np.log(np.exp(a - mx).sum(-1, keepdims=True)))[..., 0]


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "realistic"
    kw = models.ENROL_REALISTIC if kind == "realistic" else models.ENROL_SURVEY
    ubm, spk = models.synthetic_gmm_system(n_speakers=5, **kw)
    gc, miv, iv = models.stack_models([ubm] + spk)
    M, C, D = miv.shape
    iv0 = (iv[0] if iv.ndim == 3 else iv).astype(np.float64)
    cfg = oracle.default_cfg()
    feats = [np.asarray(oracle.frontend(cfg, (models.synthetic_audio(u, 48000) * 32768.0).astype(np.int16))[0], np.float64)
             for u in range(3)]
    off = np.cumsum([0] + [f.shape[0] for f in feats])
    X = np.concatenate(feats, 0)
    T = X.shape[0]
    var = 1.0 / iv0
    mu = miv[0].astype(np.float64) * var
    sd2 = var.mean(0) + np.maximum(0.0, (mu * mu).mean(0) - mu.mean(0) ** 2)
    kd = np.clip(np.rint(-0.5 * np.log2(sd2)), -24, 24)          # fb_load_gmm's balancing
    xs = X * 2.0 ** kd
    x1 = f16(xs)
    x2 = f16(xs - x1)
    mub = mu * 2.0 ** kd
    base = gc[0].astype(np.float64)[None, :] + X @ miv[0].astype(np.float64).T + (X * X) @ (-0.5 * iv0).T
    names = ["p3", "p2", "p1", "f6/1", "f6", "f6/66"]
    utt = {n: [] for n in names}
    frm = {n: [] for n in names}
    for m in range(1, M):
        dl = (miv[m] - miv[0]).astype(np.float64) * L2E * 2.0 ** -kd   # the float32 difference, as the host takes it
        dg = (gc[m] - gc[0]).astype(np.float64) * L2E
        d1 = f16(dl)
        d2 = dl - d1
        r = lse(base + (dg[None, :] + xs @ dl.T) / L2E)
        main_ = x1 @ d1.T
        val = {"p3": main_ + x1 @ d2.T + x2 @ d1.T, "p2": main_ + x1 @ d2.T,
               "p1": main_ + (d2 * mub).sum(1)[None, :]}
        corr = {n: np.zeros((T, C)) for n in ("f6/1", "f6", "f6/66")}
        for h in range(2):
            dims = [16 * (u // 8) + 8 * h + (u % 8) for u in range(32)]
            a = d2[:, dims]
            hi = q_e2m3(a)
            xq = q_frames(x1[:, dims])
            bq = q_frames(x2[:, dims] * 2.0 ** 12) @ q_e2m3(d1[:, dims] * 2.0 ** -12).T
            for n, lo in (("f6/1", 0.0 * a), ("f6", q_e2m1(a - hi)), ("f6/66", q_e2m3(a - hi))):
                corr[n] += xq @ (hi + lo).T + bq + ((a - hi - lo) * mub[:, dims]).sum(1)[None, :]
        # block 2 (lanes h = 0): the dimensions of chunk 4 -- {d2, d1 2^-12, what the first eight codes leave of d2, 0}
        c4 = [d for d in range(64, 72) if d < D]
        n4 = len(c4)
        a = np.zeros((C, 32))
        xb = np.zeros((T, 32))
        a[:, :n4], a[:, 8:8 + n4] = d2[:, c4], d1[:, c4] * 2.0 ** -12
        xb[:, :n4], xb[:, 8:8 + n4], xb[:, 16:16 + n4] = x1[:, c4], x2[:, c4] * 2.0 ** 12, x1[:, c4]
        qa1 = q_e2m3(a)
        a2 = a.copy()
        a2[:, 16:16 + n4] = d2[:, c4] - qa1[:, :n4]
        qa2 = q_e2m3(a2)
        xq = q_frames(xb)
        corr["f6/1"] += xq @ qa1.T + ((d2[:, c4] - qa1[:, :n4]) * mub[:, c4]).sum(1)[None, :]
        for n in ("f6", "f6/66"):
            corr[n] += xq @ qa2.T + ((d2[:, c4] - qa2[:, :n4] - qa2[:, 16:16 + n4]) * mub[:, c4]).sum(1)[None, :]
        for n in corr:
            val[n] = main_ + corr[n]
        for n in names:
            e = lse(base + (dg[None, :] + val[n]) / L2E) - r
            utt[n].append([e[off[u]:off[u + 1]].mean() for u in range(len(feats))])
            frm[n].append(np.sqrt((e ** 2).mean()))
    print("%s enrolment: %d frames of %d utterances, C = %d, %d speaker models" % (kind, T, len(feats), C, M - 1))
    for n in names:
        u = np.abs(np.array(utt[n]))
        print("%-6s utterance-average |err|: max %.3g rms %.3g   per-frame rms %.3g"
              % (n, u.max(), np.sqrt((u ** 2).mean()), np.sqrt((np.array(frm[n]) ** 2).mean())))


if __name__ == "__main__":
    main()
