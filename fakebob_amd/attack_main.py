#!/usr/bin/env python
"""Multi-GPU counterpart of the reference's driver (attackMain.py:274-475) -- SURVEY.md 8(f) row 1.

Same command-line flags, data layout (`data/test-set/<spk>/*.wav`, `data/illegal-set/...`), model
pickles (`model/<spk>.gmm|.iv` = [spk_id, utt_id, identity_location, z_mean, z_std]), benign-decision
filtering (CSI keeps correctly classified voices :129, OSI/SV keep rejected ones :201,:266), target
expansion (:141-162, :213-227), output naming (`<name>_<target>.wav` / `.cp`) and success-rate print
(`%d`, :411) as the reference.  What differs: every (audio, target) attack is independent, so the
list is sharded round-robin over the ranks of a `torch.distributed.run` launch (one process per
GPU) and, inside a rank, over `--streams` engines driven by host threads; the estimated threshold is
computed on rank 0 and broadcast, the success counters are all-reduced (fakebob_amd/parallel.py).

    python -m torch.distributed.run --nproc-per-node 8 -m fakebob_amd.attack_main -spk_id 1580 2830 61 ...

(The reference's own attackMain.py also runs unmodified on the engine: see fakebob_amd/dropin/.)
"""
import argparse
import os
import pickle
import threading

import numpy as np
from scipy.io.wavfile import read, write

from . import parallel
from .attack import FakeBob

bits_per_sample = 16
fs = 16000


def load_spk_models(model_dir, spk_id_list, architecture):
    ext = ".iv" if architecture == "iv" else ".gmm"
    out = []
    for spk_id in spk_id_list:
        with open(os.path.join(model_dir, spk_id + ext), "rb") as r:
            out.append(pickle.load(r))
    return out


def make_model(architecture, task, model_list, pre_model_dir, threshold, group_id):
    """attackMain.load_model (:38-85)."""
    from .systems import gmm_CSI, gmm_OSI, gmm_SV, iv_CSI, iv_OSI, iv_SV
    ubm = os.path.join(pre_model_dir, "final.dubm")
    if architecture == "iv":
        if task == "OSI":
            return iv_OSI(group_id, model_list, pre_model_dir=pre_model_dir, threshold=threshold)
        if task == "CSI":
            return iv_CSI(group_id, model_list, pre_model_dir=pre_model_dir)
        return iv_SV(group_id, model_list[0], pre_model_dir=pre_model_dir, threshold=threshold)
    if task == "OSI":
        return gmm_OSI(group_id, model_list, ubm, pre_model_dir=pre_model_dir, threshold=threshold)
    if task == "CSI":
        return gmm_CSI(group_id, model_list, pre_model_dir=pre_model_dir)
    return gmm_SV(group_id, model_list[0], ubm, pre_model_dir=pre_model_dir, threshold=threshold)


def collect_voices(data_dir):
    """[(spk_id, file_name, float audio in [-1,1))] in a deterministic (sorted) order."""
    out = []
    for spk_id in sorted(os.listdir(data_dir)):
        d = os.path.join(data_dir, spk_id)
        if not os.path.isdir(d):
            continue
        for name in sorted(os.listdir(d)):
            _, a = read(os.path.join(d, name))
            out.append((spk_id, name, a / (2 ** (bits_per_sample - 1))))
    return out


def build_attack_list(task, attack_type, model, test_dir, illegal_dir, out_audio_dir, out_cp_dir):
    """attackMain.loadData (:87-272) -> list of dicts {audio, true, target, wav_path, cp_path, name, spk}."""
    voices = collect_voices(test_dir if task == "CSI" else illegal_dir)
    if not voices:
        return []
    spk_ids = list(model.spk_ids) if hasattr(model, "spk_ids") else []
    decisions, _ = model.make_decisions([v[2] for v in voices], fs=fs, bits_per_sample=bits_per_sample)
    decisions = decisions if isinstance(decisions, list) else [decisions]
    items = []
    for (spk, name, audio), dec in zip(voices, decisions):
        stem = name.split(".")[0]
        base = dict(audio=audio, name=name, spk=spk, true=None, target=None,
                    wav_path=os.path.join(out_audio_dir, spk, name),
                    cp_path=os.path.join(out_cp_dir, spk, stem + ".cp"))
        if task == "CSI":
            true = spk_ids.index(spk)
            if int(dec) != true:          # skip those wrongly classified (:127-129)
                continue
            base["true"] = true
        elif int(dec) != -1:              # OSI / SV: keep voices the system rejects (:199-201, :264-266)
            continue
        if task != "SV" and attack_type == "targeted":
            for t in range(len(spk_ids)):
                if task == "CSI" and t == base["true"]:
                    continue
                it = dict(base, target=t)
                it["wav_path"] = os.path.join(out_audio_dir, spk, stem + "_" + str(t) + ".wav")
                it["cp_path"] = os.path.join(out_cp_dir, spk, stem + "_" + str(t) + ".cp")
                items.append(it)
        else:
            items.append(base)
    return items


def main(argv=None, model_factory=None, bob_factory=None):
    """model_factory(architecture, task, model_list, pre_model_dir, threshold, group_id) / bob_factory(task,
    attack_type, model, **hyper_parameters): injection points for the driver-rule tests (a stub model / stub FakeBob
    as in tests/golden/driver_site.py); default: the GPU systems and fakebob_amd.attack.FakeBob."""
    ap = argparse.ArgumentParser()
    ap.add_argument("--speaker_id", "-spk_id", nargs="+", type=str, required=True)
    ap.add_argument("--architecture", "-archi", default="gmm", choices=["gmm", "iv"])
    ap.add_argument("--task", "-task", default="OSI", choices=["OSI", "CSI", "SV"])
    ap.add_argument("--attack_type", "-type", default="targeted", choices=["untargeted", "targeted"])
    ap.add_argument("--adver_thresh", "-adver", default=0., type=float)
    ap.add_argument("--epsilon", "-epsilon", default=0.002, type=float)
    ap.add_argument("--max_iter", "-max_iter", default=1000, type=int)
    ap.add_argument("--max_lr", "-max_lr", default=0.001, type=float)
    ap.add_argument("--min_lr", "-min_lr", default=1e-6, type=float)
    ap.add_argument("--samples_per_draw", "-samples", default=50, type=int)
    ap.add_argument("--sigma", "-sigma", default=0.001, type=float)
    ap.add_argument("--momentum", "-momentum", default=0.9, type=float)
    ap.add_argument("--plateau_length", "-plateau_length", default=5, type=int)
    ap.add_argument("--plateau_drop", "-plateau_drop", default=2.0, type=float)
    ap.add_argument("--n_jobs", "-nj", default=1, type=int)                     # accepted, unused
    ap.add_argument("--debug", "-debug", default="f", choices=["t", "f"])       # accepted, unused
    ap.add_argument("--threshold", "-thresh", default=0., type=float)
    ap.add_argument("--streams", default=3, type=int, help="attacks in flight per GPU")
    ap.add_argument("--schedule", default="dynamic", choices=["dynamic", "static"],
                    help="dynamic: free attack streams draw the next attack (ticket counter); static: round-robin deal")
    ap.add_argument("--seed", default=None, type=int, help="Philox key (default: from numpy's global RNG)")
    ap.add_argument("--model_dir", default="./model")
    ap.add_argument("--pre_model_dir", default="pre-models")
    ap.add_argument("--test_dir", default="./data/test-set")
    ap.add_argument("--illegal_dir", default="./data/illegal-set")
    ap.add_argument("--out_dir", default=".")
    ap.add_argument("--dist-backend", default=None)
    args = ap.parse_args(argv)

    task, attack_type, spk_id_list = args.task, args.attack_type, args.speaker_id
    if task == "SV":                      # SV only supports one enrolled speaker (:449-451)
        attack_type, spk_id_list = "targeted", spk_id_list[0:1]
    ident = args.architecture + "-" + task + "-" + attack_type
    out_audio = os.path.join(args.out_dir, "adversarial-audio", ident)
    out_cp = os.path.join(args.out_dir, "checkpoint", ident)
    if task == "SV":
        out_audio, out_cp = os.path.join(out_audio, spk_id_list[0]), os.path.join(out_cp, spk_id_list[0])

    dist = parallel.init_process_group(args.dist_backend)
    rank, _local, world = parallel.dist_env()
    K = max(1, args.streams)
    if model_factory is None:
        model_list = load_spk_models(args.model_dir, spk_id_list, args.architecture)
        model_factory = make_model
    else:
        model_list = spk_id_list
    models = [model_factory(args.architecture, task, model_list, args.pre_model_dir, args.threshold,
                            os.path.join(args.out_dir, ident + ("-%d" % k))) for k in range(K)]
    if K >= 3:  # several engines share the GPU: the separate launches interleave better (fb_set_fused_chain)
        for m in models:
            if hasattr(m, "engine"):
                m.engine.set_fused_chain(False)
    # one Philox key for the whole job: drawn on rank 0 when --seed is omitted and broadcast, so that a multi-rank
    # run is reproducible and its results do not depend on the sharding
    seed = args.seed if args.seed is not None else (int(np.random.randint(0, 2 ** 31 - 1)) if rank == 0 else 0)
    seed = parallel.broadcast_int(seed, dist)
    hp = dict(adver_thresh=args.adver_thresh, epsilon=args.epsilon, max_iter=args.max_iter, max_lr=args.max_lr,
              min_lr=args.min_lr, samples_per_draw=args.samples_per_draw, sigma=args.sigma,
              momentum=args.momentum, plateau_length=args.plateau_length, plateau_drop=args.plateau_drop)
    if bob_factory is None:
        bobs = [FakeBob(task, attack_type, m, seed=seed, verbose=False, **hp) for m in models]
    else:
        bobs = [bob_factory(task, attack_type, m, **hp) for m in models]

    items = build_attack_list(task, attack_type, models[0], args.test_dir, args.illegal_dir, out_audio, out_cp)
    total = len(items)
    if rank == 0:
        print("------ load data done, total num: %d ------" % total)
    for it in items:
        os.makedirs(os.path.dirname(it["wav_path"]), exist_ok=True)
        os.makedirs(os.path.dirname(it["cp_path"]), exist_ok=True)

    threshold = 0.
    if task != "CSI" and total > 0:       # estimate the threshold on one random voice (:355-357, :393-394)
        thr = None
        if rank == 0:
            pick = items[int(np.random.choice(total, 1)[0])]["audio"]
            thr, _, _ = bobs[0].estimate_threshold(pick, fs=fs, bits_per_sample=bits_per_sample)
        threshold = parallel.broadcast_threshold(thr, dist)

    # every attack stream of every rank draws the next global attack index when it is free (attack cost ranges from one
    # iteration -- early stop, FAKEBOB.py:181-191 -- to max_iter); --schedule static: round-robin over ranks and streams
    queue = parallel.WorkQueue(total, dist, args.schedule, streams=K)
    results = {}
    lock = threading.Lock()
    errors = []

    def worker(k):
        try:
            run_stream(k)
        except BaseException as ex:  # noqa: BLE001  (re-raised below: a dead stream must not look like a short list)
            errors.append(ex)

    def run_stream(k):
        bob = bobs[k]
        while True:
            idx = queue.next(k)
            if idx is None:
                return
            it = items[idx]
            bob._stream = idx + 1         # Philox stream = global attack index: results do not depend on the sharding
            # the reference passes true= only for CSI untargeted (:346) and target= only for targeted attacks (:332,:367)
            true = it["true"] if (task == "CSI" and attack_type == "untargeted") else None
            adv, flag = bob.attack(it["audio"], it["cp_path"], threshold=threshold, true=true,
                                   target=it["target"], fs=fs, bits_per_sample=bits_per_sample)
            write(it["wav_path"], fs, adv)
            with lock:
                results[idx] = flag

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(K)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    if parallel.agree_on_failure(bool(errors), dist):   # every rank learns of it BEFORE the result reduction
        if errors:
            raise errors[0]
        raise RuntimeError("an attack stream of another rank failed: the job's results are incomplete")
    st = [m.engine.stats() if hasattr(m, "engine") else dict(nes_iters=0, scored_utts=0) for m in models]
    succ = sum(1 for f in results.values() if f == 1)
    g = parallel.reduce_counters([succ, len(results), sum(s["nes_iters"] for s in st), sum(s["scored_utts"] for s in st)], dist)
    if g[1] != total:
        raise RuntimeError("the work queue handed out %d of %d attacks" % (g[1], total))
    if rank == 0:
        if g[1] > 0:
            print('------ attack successful rate %d ------' % (g[0] * 100 / g[1]))
        print("----- generate adversarial voices done: %d attacks, %d NES iterations, %d utterances scored -----"
              % (g[1], g[2], g[3]))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return g, results, threshold


if __name__ == "__main__":
    main()
