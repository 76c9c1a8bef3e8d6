import sys, time, threading, numpy as np
sys.path.insert(0, '.')
from fakebob_amd.engine import Engine, nes_params
from fakebob_amd.models import synthetic_audio, synthetic_gmm_system
ubm, spk = synthetic_gmm_system(5, 2048, 72)
for K in (1, 2, 3, 4, 6):
    engs = []
    for k in range(K):
        e = Engine(0); e.load_gmm([ubm] + spk); e.set_system("OSI"); engs.append(e)
    p = [nes_params("OSI", "targeted", samples_per_draw=50, target=0, threshold=0.2277, stream=k) for k in range(K)]
    auds = [synthetic_audio(k, 48000) for k in range(K)]
    steps = 60
    def run(k, n):
        engs[k].bench_nes(p[k], auds[k], 0, n)
    ths = [threading.Thread(target=run, args=(k, 10)) for k in range(K)]
    [t.start() for t in ths]; [t.join() for t in ths]
    t0 = time.perf_counter()
    ths = [threading.Thread(target=run, args=(k, steps)) for k in range(K)]
    [t.start() for t in ths]; [t.join() for t in ths]
    dt = time.perf_counter() - t0
    print("K=%d streams: %.1f it/s aggregate (%.3f ms per iteration-equivalent)" % (K, K * steps / dt, 1e3 * dt / (K * steps)))
    [e.close() for e in engs]
