#!/usr/bin/env python
"""bench.py -- NES iterations/s of the FAKEBOB hot path on MI355X.

Workload (BASELINE.json configs[1]): GMM-UBM OSI targeted attack, 5 enrolled speakers + UBM
(C=2048 diagonal Gaussians, D=72), samples_per_draw=50 -> 51 utterances of 3 s @ 16 kHz scored
per NES iteration.  One "step" = one full NES iteration exactly as FakeBob.attack runs it
(FAKEBOB.py:168-214): Philox noise -> perturb -> int16 -> MFCC/VAD/deltas/CMVN -> GMM
log-likelihoods of all 6 models -> scores -> loss -> gradient estimate -> momentum sign step +
clip, with the early-stop *test* evaluated but not taken so that exactly K steps are timed.
Synthetic audio and seeded synthetic models: no dataset / Kaldi models exist
offline.

Attacks in flight: attacks on different utterances are independent (attackMain.py:324-409), so each
GPU runs --streams K of them concurrently, one engine (= HIP stream + device state) each, driven from K
host threads (ctypes releases the GIL).  Kernels of different attacks then overlap on the chip --
the float64 VALU front-end of one attack runs beside the MFMA GMM kernel of another -- which is what
fills the gaps a single launch chain leaves.  One "step" = one NES iteration of each of the K attacks
in flight; `value` counts all of them.

Multi-GPU: utterances are independent (attackMain.py:324-409), so each rank attacks its own
utterance ("weak" scaling, no data-path collective); RCCL is used only for the barrier, the
max-over-ranks time and the final counter reduction.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

S_SPK, C_GAUSS, D_FEAT, SPD, N_SAMPLES = 5, 2048, 72, 50, 48000
PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
README_GMM_ITS = 0.20         # reference README.md:113 (~5 s / iteration, unspecified CPU)


def cpu_baseline(audio, models, params_kw):
    """Times the CPU oracle (the build's C restatement of the identical computation; Kaldi itself
    cannot be installed offline) on one full NES iteration of the same workload, 1 thread --
    the reference's default n_jobs=1 (attackMain.sh:34)."""
    from oracle import oracle as O
    from fakebob_amd.models import stack_models
    gc, miv, iv = stack_models(models)
    ctx = O.GmmSystemCtx(O.default_cfg(), "OSI", gc, miv, iv, nthreads=1)
    po = O.nes_params("OSI", "targeted", ctx.S, **params_kw)
    t0 = time.perf_counter()
    O.get_grad(po, ctx.fn, ctx.ctx, audio, seed=42, it=0, stream=0)
    dt = time.perf_counter() - t0
    out = {"value": 1.0 / dt, "unit": "NES iterations/s", "cores": 1, "kind": "port",
           "sample": "1 full NES iteration (51 utterances x 3 s, 6 models) of the same workload, "
                     "CPU oracle single thread, %.1f s" % dt}
    # the same iteration with the oracle's utterance-parallel scoring on every host core (report
    # the single-thread and the all-cores restatement separately; the reference's own default is n_jobs=1)
    try:
        nthr = max(1, len(os.sched_getaffinity(0)))
        if nthr > 1:
            ctx = O.GmmSystemCtx(O.default_cfg(), "OSI", gc, miv, iv, nthreads=nthr)
            O.get_grad(po, ctx.fn, ctx.ctx, audio, seed=42, it=0, stream=0)        # warm the thread pool / caches
            t0 = time.perf_counter()
            n = 0
            while n < 3 and time.perf_counter() - t0 < 10.0:
                O.get_grad(po, ctx.fn, ctx.ctx, audio, seed=42, it=n + 1, stream=0)
                n += 1
            dta = (time.perf_counter() - t0) / n
            out["all_cores"] = {"value": 1.0 / dta, "unit": "NES iterations/s", "cores": nthr, "kind": "port",
                                "sample": "%d NES iterations, oracle with %d threads (utterances in parallel), "
                                          "%.2f s each" % (n, nthr, dta)}
    except Exception as ex:  # noqa: BLE001 -- the single-thread number above is the contract
        out["all_cores"] = {"error": str(ex)[:200]}
    return out


def bench_ivector(args, torch):
    """BASELINE.json configs[2]: i-vector-PLDA SV targeted attack, spd=50, C=2048, D=72, R=400, LDA 200
    (the T-matrix contraction path).  Not the headline metric: run with --arch iv."""
    from fakebob_amd.engine import Engine, nes_params
    from fakebob_amd.models import synthetic_audio, synthetic_ivector_system
    t0 = time.perf_counter()
    sy = synthetic_ivector_system(C=C_GAUSS, D=D_FEAT, R=400, L=200, n_speakers=1)
    sy = sy.with_enrolled(sy.enrolled, [-40.0], [10.0])
    import threading
    K = max(1, args.streams)
    engs = []
    for k in range(K):
        e = Engine(0)
        e.load_ivector(sy, "SV")
        engs.append(e)
    eng = engs[0]
    t_load = time.perf_counter() - t0
    kw = dict(samples_per_draw=SPD, epsilon=0.002, sigma=0.001, max_iter=1000, threshold=1.0)
    auds = [synthetic_audio(k, N_SAMPLES) for k in range(K)]
    prms = [nes_params("SV", "targeted", seed=42, stream=k, **kw) for k in range(K)]
    audio, p = auds[0], prms[0]
    res = [None] * K

    def run_all(n, timed):
        def run(k):
            res[k] = engs[k].bench_nes(prms[k], auds[k], 0, n, time_gmm=timed)
        ths = [threading.Thread(target=run, args=(k,)) for k in range(K)]
        [t.start() for t in ths]
        [t.join() for t in ths]

    run_all(max(1, args.warmup), False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_all(args.steps, True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms_dev, ms_con, rows = res[0]
    ms_con = sum(r[1] for r in res) / K
    its = K * args.steps / dt
    tri = 400 * 401 // 2
    bytes_stream = 8.0 * (C_GAUSS * D_FEAT * 400 + C_GAUSS * tri)       # Sigma^-1 M + U, float64, read once
    n_active = eng.debug_iv_active()
    n_bgroups = (SPD + 1 + 63) // 64                                   # utterance groups of 64 -> passes over the rows
    bytes_exec = 8.0 * n_active * (D_FEAT * 400 + tri) * n_bgroups
    # f64 MFMA work actually issued: 64-row tiles (51 useful), only the active rows
    flops_exec = 2.0 * 64 * n_bgroups * n_active * (D_FEAT * 400 + tri)
    con_ms = ms_con / args.steps
    out = {"metric": "NES iterations/sec (i-vector-PLDA SV, samples_per_draw=50, 3 s@16 kHz)", "value": its,
           "unit": "NES iterations/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f64 (extractor/PLDA), f32 MFMA (gselect)", "data": "synthetic",
           "vs_readme_nominal": its / 0.083,
           "config": {"workload": "i-vector-PLDA SV targeted, C=2048, D=72, R=400, LDA=200, spd=50, N=48000, "
                                  "%d attacks in flight" % K, "attacks_in_flight_per_gpu": K,
                      "voiced_rows_per_iter": rows, "model_load_s": t_load},
           "roofline": {"kernel": "k_iv_contract_dma<lin> + <quad> (T-matrix contraction: LDS-DMA ring, float64 MFMA "
                                  "v_mfma_f64_16x16x4, active rows only)", "bound": "hbm",
                        "achieved": bytes_stream / (con_ms * 1e-3) / 1e9, "peak": 8000.0, "unit": "GB/s",
                        "frac": bytes_stream / (con_ms * 1e-3) / 1e9 / 8000.0, "traffic": None,
                        "avg_launch_ms": con_ms, "algorithmic_bytes_per_launch": bytes_stream,
                        "active_components": n_active, "executed_bytes_per_launch": bytes_exec,
                        "executed_gbps": bytes_exec / (con_ms * 1e-3) / 1e9,
                        "note": "algorithmic = both matrices streamed once (SURVEY.md 8(d)); the kernels stream only "
                                "the rows of components with posterior mass, once per 64-utterance group",
                        "flops_per_launch": 2.0 * (SPD + 1) * (C_GAUSS * D_FEAT * 400 + C_GAUSS * tri),
                        "mfma_f64": {"executed_flops_per_launch": flops_exec,
                                     "executed_tflops": flops_exec / (con_ms * 1e-3) / 1e12, "peak_tflops": 78.6,
                                     "frac": flops_exec / (con_ms * 1e-3) / 1e12 / 78.6}}}
    try:  # HBM bytes per launch (both kernels) from the committed rocprofv3 PMC passes
        with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as r:
            out["roofline"]["traffic"] = json.load(r)["kernels"]["k_iv_contract_dma<lin>+<quad>"]["hbm_bytes_per_launch"]
        out["roofline"]["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 x2 fetch correction)"
    except Exception:
        pass
    if not args.no_cpu_baseline:
        from oracle import oracle as O
        import numpy as np
        ctx = O.IvSystemCtx(O.default_cfg(), sy, nthreads=1)
        n_s = SPD + 1                                        # one full NES batch (a few seconds of CPU work)
        wavs = [(synthetic_audio(u, N_SAMPLES) * 32768).astype(np.int16) for u in range(n_s)]
        t0 = time.perf_counter()
        ctx.score_batch(wavs)
        t8 = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": 1.0 / t8, "unit": "NES iterations/s", "cores": 1,
                               "kind": "port", "sample": "the %d utterances of one NES batch (3 s each) scored by the CPU "
                               "oracle, 1 thread, %.1f s" % (n_s, t8)}
    print(json.dumps(out))
    for e in engs:
        e.close()


GMM_MODE = os.environ.get("FB_GMM_MODE", "fx2") or "fx2"
GMM_TRAFFIC_KEY = {"fx2": "k_gmm_fx2<5, false>", "bx3": "k_gmm_bx3<5, false>", "f32": "k_gmm<36, false>"}[GMM_MODE]
PEAK_BF16_MFMA_TFLOPS = 2500.0     # dense (MI355X_MICROARCH.md)
# bf16 32x32x16 chain on random operands, this chip, scratch/bx_probe.hip: the clock drops to ~1.6 GHz
# under a saturated bf16 matrix pipe (DVFS), which bounds any real kernel below the 2.5 PF spec peak
BF16_MFMA_POWER_LIMITED_TFLOPS = 1660.0


def _gmm_roofline(achieved, flops_launch, gmm_ms_avg):
    M = S_SPK + 1
    shared = (1 + M) / (2.0 * M)
    r = {"bound": "mfma", "achieved": achieved, "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
         "frac": achieved / PEAK_F32_MFMA_TFLOPS, "traffic": None, "avg_launch_ms": gmm_ms_avg,
         "algorithmic_flops_per_launch": flops_launch}
    if GMM_MODE == "fx2":
        nk = (D_FEAT + 1 + 15) // 16
        ex = flops_launch * shared * 3 * (16.0 * nk / D_FEAT)
        ex_t = ex / (gmm_ms_avg * 1e-3) / 1e12 if gmm_ms_avg > 0 else 0.0
        r.update({"kernel": "k_gmm_fx2<5,false> (diag-GMM log-likelihood + logsumexp; f32 operands as a two-term "
                            "f16 split accurate to half an f32 ulp, 3 partial products on v_mfma_f32_32x32x16_f16, "
                            "f32 accumulate)",
                  "executed_flops_per_launch": ex, "executed_tflops": ex_t, "executed_pipe": "f16 MFMA",
                  "executed_frac": ex_t / PEAK_BF16_MFMA_TFLOPS,
                  "executed_frac_of_power_limited_ceiling": ex_t / BF16_MFMA_POWER_LIMITED_TFLOPS})
    elif GMM_MODE == "bx3":
        nk = (D_FEAT + 3 + 15) // 16
        ex = flops_launch * shared * 6 * (16.0 * nk / D_FEAT)
        ex_t = ex / (gmm_ms_avg * 1e-3) / 1e12 if gmm_ms_avg > 0 else 0.0
        r.update({"kernel": "k_gmm_bx3<5,false> (diag-GMM log-likelihood + logsumexp; f32 operands as an exact "
                            "3-way bf16 split, 6 partial products on v_mfma_f32_32x32x16_bf16, f32 accumulate)",
                  "executed_flops_per_launch": ex, "executed_tflops": ex_t, "executed_pipe": "bf16 MFMA",
                  "executed_frac": ex_t / PEAK_BF16_MFMA_TFLOPS,
                  "executed_frac_of_power_limited_ceiling": ex_t / BF16_MFMA_POWER_LIMITED_TFLOPS})
    else:
        r.update({"kernel": "k_gmm<36,false> (diag-GMM log-likelihood + logsumexp, f32 MFMA 32x32x2)",
                  "executed_flops_per_launch": flops_launch * shared, "executed_tflops": achieved * shared,
                  "executed_pipe": "f32 MFMA", "executed_frac": achieved * shared / PEAK_F32_MFMA_TFLOPS})
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--streams", type=int, default=3, help="attacks in flight per GPU (one engine/stream each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--arch", default="gmm", choices=["gmm", "iv"], help="gmm = headline (configs[1]); iv = configs[2]")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL over xGMI) | gloo (plumbing test)")
    ap.add_argument("--same-device", action="store_true",
                    help="plumbing test on a 1-GPU box: every rank uses cuda:0 (implies a non-RCCL backend)")
    args = ap.parse_args()

    import torch  # device sync + torch.distributed (RCCL); imported before the HIP library
    if args.arch == "iv":
        return bench_ivector(args, torch)
    from fakebob_amd import parallel
    rank, local_rank, world = parallel.dist_env()
    dev_index = 0 if args.same_device else local_rank
    dist = None
    if world > 1:
        if args.dist_backend == "nccl":
            torch.cuda.set_device(dev_index)
        dist = parallel.init_process_group(args.dist_backend)
    from fakebob_amd.engine import Engine, nes_params
    from fakebob_amd.models import synthetic_audio, synthetic_gmm_system

    import threading
    K = max(1, args.streams)
    ubm, spk = synthetic_gmm_system(S_SPK, C_GAUSS, D_FEAT)
    models = [ubm] + spk
    kw = dict(samples_per_draw=SPD, epsilon=0.002, sigma=0.001, max_lr=0.001, min_lr=1e-6, momentum=0.9,
              plateau_length=5, plateau_drop=2.0, adver_thresh=0.0, max_iter=1000, target=0, threshold=0.2277)
    engs, auds, prms = [], [], []
    for k in range(K):
        e = Engine(dev_index)
        e.load_gmm(models)
        e.set_system("OSI")
        engs.append(e)
        utt = rank * K + k                                  # a different utterance per attack
        auds.append(synthetic_audio(utt, N_SAMPLES))
        prms.append(nes_params("OSI", "targeted", seed=42, stream=utt, **kw))
    audio = auds[0]
    results = [None] * K

    def run(k, n, timed):
        results[k] = engs[k].bench_nes(prms[k], auds[k], 0, n, time_gmm=timed)

    def run_all(n, timed):
        ths = [threading.Thread(target=run, args=(k, n, timed)) for k in range(K)]
        [t.start() for t in ths]
        [t.join() for t in ths]

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run_all(args.warmup, False)
    barrier()
    t0 = time.perf_counter()
    run_all(args.steps, True)
    barrier()
    dt = time.perf_counter() - t0
    ms_dev = sum(r[0] for r in results) / K
    ms_gmm = sum(r[1] for r in results) / K                 # per attack: sum over its timed launches
    rows = int(sum(r[2] for r in results) / K)
    total_steps = args.steps * K
    if dist is not None:
        tdev = "cuda" if args.dist_backend == "nccl" else "cpu"
        t = torch.tensor([dt], dtype=torch.float64, device=tdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        # final counter reduction (mirrors success_cnt / total_cnt, attackMain.py:312,411): the only
        # data the ranks ever exchange
        total_steps, total_scored, _ = parallel.reduce_counters([args.steps * K, args.steps * K * (SPD + 1), rows], dist)
        assert total_steps == world * args.steps * K
    its = total_steps / dt
    out = None
    if rank == 0:
        gmm_ms_avg = ms_gmm / args.steps
        flops_launch = (S_SPK + 1) * C_GAUSS * 4 * D_FEAT * rows  # (S+1)*C*4D*F_voiced
        achieved = flops_launch / (gmm_ms_avg * 1e-3) / 1e12 if gmm_ms_avg > 0 else 0.0
        out = {
            "metric": "NES iterations/sec (and scored-utts/sec) at samples_per_draw=50, 3 s@16 kHz",
            "value": its, "unit": "NES iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": {"fx2": "f32 (GMM: two-term f16 split on MFMA, f32 accumulate, f32-equivalent; f64 front-end/NES)",
                                                                   "bx3": "f32 (GMM: exact bf16x3 split on MFMA, f32 accumulate; f64 front-end/NES)",
                                                                   "f32": "f32 (MFMA f32 GMM; f64 front-end/NES)"}[GMM_MODE],
            "data": "synthetic",
            "scored_utts_per_s": its * (SPD + 1),
            "vs_readme_nominal": its / README_GMM_ITS,
            "config": {"workload": "GMM-UBM OSI targeted, 5 speakers+UBM, C=2048, D=72, spd=50, "
                                   "N=48000 (3 s @ 16 kHz), %d attacks in flight per GPU "
                                   "(1 step = 1 NES iteration of each)" % K,
                       "attacks_in_flight_per_gpu": K,
                       "voiced_rows_per_iter": rows, "utterances_per_iter": SPD + 1,
                       "seeds": {"audio": 1234, "ubm": 2001, "speakers": 2100, "philox": 42}},
            # `achieved` uses the ALGORITHMIC flops of SURVEY.md 8(d): (S+1)*C*4D per voiced frame (two
            # length-D dot products per component per model, as Kaldi evaluates them in float32).  The
            # kernel (a) shares the quadratic term across the 6 models (mean-only MAP adaptation):
            # (1 + M)/(2M) = 7/12 of those products are executed, and (b) by default evaluates each
            # f32 product on the f16 matrix pipe as 3 partial products of a two-term f16 split
            # (k_gmm_fx2, f32-equivalent accuracy; FB_GMM_MODE=bx3 selects the exact 3-way bf16 split
            # with 6 partial products, FB_GMM_MODE=f32 the plain f32-MFMA kernel).
            # `peak` is the MFMA peak of the path's arithmetic type (f32: 157.3 TF) so `frac` can
            # exceed 1; `executed_*` give the honest utilisation of the pipe the instructions run on.
            "roofline": dict(_gmm_roofline(achieved, flops_launch, gmm_ms_avg),
                             gmm_share_of_stream_time=ms_gmm / ms_dev if ms_dev > 0 else None,
                             note="launch durations are HIP-event times on each attack's own stream; with "
                                  "several attacks in flight they include time shared with other attacks' "
                                  "kernels (solo launch: see solo_launch_ms)"),
        }
        try:  # HBM bytes per launch from the committed rocprofv3 PMC passes (not collectable in-process)
            with open(os.path.join(ROOT, "profiles", "r01_traffic.json")) as r:
                tr = json.load(r)["kernels"][GMM_TRAFFIC_KEY]
            out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            out["roofline"]["traffic_source"] = "profiles/r01_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, gfx950 x2 fetch correction)"
        except Exception:
            pass
        solo_ms, solo_rows = engs[0].bench_gmm_kernel(20)   # same kernel, same data, chip to itself
        out["roofline"]["solo_launch_ms"] = solo_ms
        out["roofline"]["solo_achieved"] = (S_SPK + 1) * C_GAUSS * 4 * D_FEAT * solo_rows / (solo_ms * 1e-3) / 1e12
        if world == 1 and not args.no_cpu_baseline:
            ckw = dict(kw)
            out["cpu_baseline"] = cpu_baseline(audio, models, ckw)
            out["gpu_over_cpu_port"] = its / out["cpu_baseline"]["value"]
        print(json.dumps(out))
        sys.stdout.flush()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for e in engs:
        e.close()


if __name__ == "__main__":
    main()
