# randomised GPU-vs-oracle sweep over model shapes, batch shapes and front-end options (run on the GPU box)
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system, synthetic_ivector_system, stack_models
from oracle import oracle as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
bad = 0
t0 = time.time()
for trial in range(N):
    over = {}
    if rng.random() < 0.3:
        over = [dict(delta_window=2), dict(delta_order=1, delta_window=2), dict(delta_order=0),
                dict(num_ceps=20, num_mel_bins=23, delta_order=3, delta_window=2), dict(num_ceps=13, num_mel_bins=23)][rng.integers(5)]
    if rng.random() < 0.2:
        over = dict(over, compress_feats=1)
    cfg = O.default_cfg(**over)
    C = int(rng.choice([17, 32, 33, 64, 96, 100, 200, 256, 300]))
    S = int(rng.integers(1, 9))
    B = int(rng.choice([1, 2, 3, 7, 16, 33]))
    lens = [int(rng.choice([1600, 4000, 9000, 16000, 24000, 48000, 70000])) for _ in range(B)]
    mode = str(rng.choice(["fx2", "fx2", "bx3", "f32"]))
    os.environ["FB_GMM_MODE"] = mode
    e = Engine(0)
    try:
        e.set_frontend(**over)
        D = e.feat_dim
        ubm, spk = synthetic_gmm_system(n_speakers=S, C=C, D=D)
        models = [ubm] + spk
        if rng.random() < 0.3:          # independent variances for some speakers (own quadratic groups)
            for m in spk[::2]:
                m.inv_vars[:] = m.inv_vars * np.exp(rng.normal(scale=0.1, size=m.inv_vars.shape)).astype(np.float32)
        e.load_gmm(models)
        wavs = [(synthetic_audio(int(rng.integers(0, 50)), n) * 32768.0).astype(np.int16) for n in lens]
        raw_g, tv_g = e.score_raw(wavs)
        gc, miv, iv = stack_models(models)
        raw_o, tv_o = O.gmm_score_batch(cfg, wavs, gc, miv, iv, nthreads=8)
        tol = 1e-4 if not over.get("compress_feats") else 5e-2
        ok = (np.array_equal(tv_g, tv_o) or over.get("compress_feats")) and np.abs(raw_g - raw_o).max() <= tol
        # one NES iteration on the first utterance
        task = str(rng.choice(["OSI", "CSI", "SV"])) if S >= 2 else "SV"
        if task == "SV":
            e.load_gmm(models[:2]); msel = models[:2]
        elif task == "CSI":
            e.load_gmm(spk); msel = spk
        else:
            msel = models
        zm = rng.normal(-60, 2, size=len(spk)) if task == "CSI" else None
        zs = rng.uniform(1, 3, size=len(spk)) if task == "CSI" else None
        e.set_system(task, zm, zs)
        gcs, mivs, ivs = stack_models(msel)
        ctx = O.GmmSystemCtx(cfg, task, gcs, mivs, ivs, zm, zs, nthreads=8)
        spd = int(rng.choice([2, 3, 5, 8, 11, 130]))   # spd < 2: the reference itself raises (B = 1 squeezes the scores, loss_fn fails)
        kw = dict(samples_per_draw=spd, threshold=float(rng.normal(0, 0.1)))
        attack = "targeted"
        if task == "CSI":
            kw = dict(samples_per_draw=spd, target=int(rng.integers(0, ctx.S)))
        elif task == "OSI":
            kw["target"] = int(rng.integers(0, ctx.S))
        audio = synthetic_audio(int(rng.integers(0, 50)), int(rng.choice([8000, 16000, 20000])))
        seed, it, st = int(rng.integers(1, 1000)), int(rng.integers(0, 9)), int(rng.integers(0, 5))
        pg = nes_params(task, attack, seed=seed, stream=st, **kw)
        po = O.nes_params(task, attack, ctx.S, **kw)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=it)
        flo, go, alo, sco = O.get_grad(po, ctx.fn, ctx.ctx, audio, seed=seed, it=it, stream=st)
        def close(a, b, t):          # NaN == NaN: samples_per_draw = 1 has no perturbed column, np.mean([]) is NaN (FAKEBOB.py:243)
            a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
            return np.array_equal(np.isnan(a), np.isnan(b)) and (np.nan_to_num(np.abs(a - b)).max() <= t)
        ok2 = close(alg, alo, tol) and close(flg, flo, tol) and close(scg[:ctx.S], sco, tol) \
            and close(gg, go, tol * 6.0 / pg.sigma)
        if spd == 1 and trial < 12:
            print("spd=1:", flg, flo, float(np.nanmax(np.abs(gg))) if not np.isnan(gg).all() else "all-nan", float(np.nanmax(np.abs(go))) if not np.isnan(go).all() else "all-nan")
        if not (ok and ok2):
            bad += 1
            print("MISMATCH trial", trial, dict(over=over, C=C, S=S, B=B, lens=lens, mode=mode, task=task, spd=spd),
                  "score err", float(np.abs(raw_g - raw_o).max()), "tv", tv_g.tolist(), tv_o.tolist(),
                  "grad:", abs(alg - alo), abs(flg - flo), float(np.abs(gg - go).max()))
    except Exception as ex:  # noqa: BLE001
        msg = str(ex)
        if "voiced" in msg:
            continue                      # an utterance without voiced frames: both sides refuse (checked elsewhere)
        bad += 1
        print("EXC trial", trial, dict(over=over, C=C, S=S, B=B, lens=lens, mode=mode), msg[:200])
    finally:
        e.close()
print("fuzz: %d trials, %d bad, %.0f s" % (N, bad, time.time() - t0))
