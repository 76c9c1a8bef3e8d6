#!/usr/bin/env python
"""Pins the reference DRIVER's rules by importing /root/reference/attackMain.py in this
container and running its own loadData() and main() against a stub model / stub FakeBob on a synthetic site:

  * which voices survive the benign-decision filter -- CSI keeps the correctly classified ones (attackMain.py:126-130),
    OSI and SV keep the rejected ones (:198-202, :263-267);
  * target expansion (:152-162 CSI skips the true label, :221-227 OSI takes every enrolled speaker);
  * output naming: adversarial-audio/<archi>-<task>-<type>[/<spk> for SV]/<dir>/<stem>[_<target>].wav and
    checkpoint/.../<stem>[_<target>].cp (:118-119, :159-160, :276-285);
  * what attack() is called with (threshold = the estimate for OSI/SV, true= / target=) and the `%d` success-rate
    line (:411).

Output: tests/golden/g10_driver.json (data only).  Run: python tests/golden/make_golden_driver.py"""
import contextlib
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, REF)

import driver_site as DS  # noqa: E402


def main():
    tmp = tempfile.mkdtemp(prefix="fb_drv_")
    DS.make_site(tmp)
    pre = os.path.join(tmp, "pre-models")
    for d in ("utils", "steps", "sid", "conf"):
        os.makedirs(os.path.join(pre, d), exist_ok=True)
    cwd = os.getcwd()
    os.chdir(tmp)
    try:
        import attackMain as AM  # the reference driver (imports the six wrappers and FAKEBOB)
        out = {"cases": []}
        for task, at in (("CSI", "untargeted"), ("CSI", "targeted"), ("OSI", "untargeted"), ("OSI", "targeted"),
                         ("SV", "targeted")):
            spk_list = DS.SPK_IDS[:1] if task == "SV" else DS.SPK_IDS
            ident = "gmm-" + task + "-" + at
            # ---- loadData with the globals main() sets (:276-285)
            AM.adver_audio_dir = os.path.join("adversarial-audio", ident)
            AM.checkpoint_dir = os.path.join("checkpoint", ident)
            if task == "SV":
                AM.adver_audio_dir = os.path.join(AM.adver_audio_dir, spk_list[0])
                AM.checkpoint_dir = os.path.join(AM.checkpoint_dir, spk_list[0])
            model = DS.StubModel(task)
            audio_list, true_l, target_l, names, wav_paths, cp_paths = AM.loadData(task, at, model, spk_list)
            items = []
            for i in range(len(audio_list)):
                items.append({"name": names[i], "code": DS.StubModel._code(audio_list[i]),
                              "true": None if true_l is None else int(true_l[i]),
                              "target": None if target_l is None else int(target_l[i]),
                              "wav_path": wav_paths[i], "cp_path": cp_paths[i]})
            # ---- main() end to end with the model loader, FakeBob, wav writer and the random pick replaced
            DS.StubBob.log = []
            written = []
            AM.load_model = lambda spk_id_list, architecture, task_, threshold, id_, _t=task: DS.StubModel(_t, threshold)
            AM.FakeBob = DS.StubBob
            AM.write = lambda path, fs, audio: written.append(path)
            old_choice = np.random.choice
            np.random.choice = lambda n, k: np.array([n // 2])
            buf = io.StringIO()
            try:
                with contextlib.redirect_stdout(buf):
                    AM.main(spk_list, "gmm", task, 0.0, at, 0., 0.002, 1000, 0.001, 1e-6, 50, 0.001, 0.9, 5, 2.0, 1, False)
            finally:
                np.random.choice = old_choice
            lines = buf.getvalue().splitlines()
            rate = [ln for ln in lines if "attack successful rate" in ln]
            total = [ln for ln in lines if "load data done" in ln]
            out["cases"].append({"task": task, "attack_type": at, "items": items,
                                 "log": [list(x) for x in DS.StubBob.log], "written": written,
                                 "rate_line": rate[0], "total_line": total[0]})
        out["listdir_note"] = ("the reference walks os.listdir() order (filesystem dependent); compare as sets keyed by "
                               "wav_path, targets of one voice in ascending order")
        with open(os.path.join(HERE, "g10_driver.json"), "w") as w:
            json.dump(out, w, indent=1, sort_keys=True)
        print("wrote g10_driver.json:", [(c["task"], c["attack_type"], len(c["items"]), c["rate_line"]) for c in out["cases"]])
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()
