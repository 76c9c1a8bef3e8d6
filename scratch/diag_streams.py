import sys, time, os
sys.path.insert(0, os.getcwd())
import torch
import bench as B
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system
K = 3
ubm, spk = synthetic_gmm_system(5, 2048, 72)
models = [ubm] + spk
kw = dict(samples_per_draw=50, epsilon=0.002, sigma=0.001, max_iter=1000, target=0, threshold=0.2277)
engs = []; auds = []; prms = []
for k in range(K):
    e = Engine(0); e.load_gmm(models); e.set_system("OSI"); engs.append(e)
    auds.append(synthetic_audio(k, 48000)); prms.append(nes_params("OSI", "targeted", seed=42, stream=k, **kw))
res = [None]*K; win = [None]*K
def run(k, n, timed):
    t0 = time.perf_counter()
    res[k] = engs[k].bench_nes(prms[k], auds[k], 0, n, time_gmm=timed)
    win[k] = (t0, time.perf_counter())
w = B.Workers(K, run)
w.run(60, False)
for rep in range(12):
    n = 20
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    w.run(n, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("rep %d: %.0f it/s  dt=%.2f ms | " % (rep, K*n/dt, dt*1e3) + "  ".join("[%.2f..%.2f dev %.2f]" % ((a-t0)*1e3, (b-t0)*1e3, r[0]) for (a, b), r in zip(win, res)))
    if rep == 5: time.sleep(1.0)
w.close()
