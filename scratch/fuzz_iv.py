# randomised GPU-vs-oracle sweep of the i-vector / PLDA path over system and batch shapes (run on the GPU box)
import os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_ivector_system
from oracle import oracle as O

rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
t0 = time.time()
for trial in range(N):
    C = int(rng.choice([33, 64, 96, 130, 256]))
    R = int(rng.choice([16, 40, 48, 64, 100, 130]))
    L = int(rng.choice([r for r in (8, 16, 24, 50, 100) if r <= R]))
    S = int(rng.integers(1, 11))
    B = int(rng.choice([1, 2, 5, 9, 20]))
    lens = [int(rng.choice([4000, 9000, 16000, 30000, 48000, 64000])) for _ in range(B)]
    task = str(rng.choice(["OSI", "CSI", "SV"])) if S >= 2 else "SV"
    info = dict(C=C, R=R, L=L, S=S, B=B, lens=lens, task=task)
    e = Engine(0)
    try:
        sy = synthetic_ivector_system(C=C, D=72, R=R, L=L, n_speakers=S, seed=int(rng.integers(1, 1000)))
        zm, zs = list(rng.normal(-30, 5, size=S)), list(rng.uniform(2, 9, size=S))
        sy = sy.with_enrolled(sy.enrolled, z_mean=zm, z_std=zs)
        if task == "SV":
            sy = sy.with_enrolled(sy.enrolled[:1], zm[:1], zs[:1])
        e.load_ivector(sy, task)
        ctx = O.IvSystemCtx(O.default_cfg(), sy, nthreads=8)
        wavs = [(synthetic_audio(int(rng.integers(0, 50)), n) * 32768.0).astype(np.int16) for n in lens]
        llr_g, tv_g = e.score_raw(wavs)
        llr_o, ivs_o, tv_o = ctx.score_batch(wavs)
        ivs_g = e.debug_ivectors(len(wavs), sy.R)
        ok = np.array_equal(tv_g, tv_o) and np.abs(llr_g - llr_o).max() <= 1e-4 \
            and np.abs(ivs_g - ivs_o).max() <= 1e-6 * max(1.0, np.abs(ivs_o).max())
        spd = int(rng.choice([2, 5, 8, 20]))
        kw = dict(samples_per_draw=spd)
        if task != "CSI":
            kw["threshold"] = float(rng.normal(0, 1))
        if task == "OSI" or task == "CSI":
            kw["target"] = int(rng.integers(0, ctx.S))
        audio = synthetic_audio(int(rng.integers(0, 50)), int(rng.choice([12000, 16000, 24000])))
        seed, it, st = int(rng.integers(1, 1000)), int(rng.integers(0, 9)), int(rng.integers(0, 5))
        pg = nes_params(task, "targeted", seed=seed, stream=st, **kw)
        po = O.nes_params(task, "targeted", ctx.S, **kw)
        flg, gg, alg, scg = e.get_grad(pg, audio, it=it)
        flo, go, alo, sco = O.get_grad(po, ctx.fn, ctx.ctx, audio, seed=seed, it=it, stream=st)
        ok2 = abs(alg - alo) <= 1e-4 and abs(flg - flo) <= 1e-4 and np.abs(scg[:ctx.S] - sco).max() <= 1e-4 \
            and np.abs(gg - go).max() <= 1e-4 * 6.0 / pg.sigma
        if not (ok and ok2):
            bad += 1
            print("MISMATCH", trial, info, "llr", float(np.abs(llr_g - llr_o).max()), "iv", float(np.abs(ivs_g - ivs_o).max()),
                  "tv", np.array_equal(tv_g, tv_o), "grad", abs(alg - alo), abs(flg - flo), float(np.abs(gg - go).max()))
    except Exception as ex:  # noqa: BLE001
        if "voiced" in str(ex):
            continue
        bad += 1
        print("EXC", trial, info, str(ex)[:300])
    finally:
        e.close()
print("fuzz_iv: %d trials, %d bad, %.0f s" % (N, bad, time.time() - t0))
