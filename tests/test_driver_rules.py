"""fakebob_amd.attack_main against the reference driver's own behaviour.

tests/golden/g10_driver.json was captured by IMPORTING /root/reference/attackMain.py and running its loadData()
and main() with a stub model / stub FakeBob on a synthetic site (tests/golden/make_golden_driver.py).  Here the same
stubs drive build_attack_list() and main(): the surviving voices (CSI keeps correctly classified, attackMain.py:126-130;
OSI / SV keep rejected, :198-202 / :263-267), the target expansion (:152-162, :221-227), every output path
(:118-119, :159-160), the arguments attack() receives and the `%d` success-rate line (:411) must be the reference's.
The reference walks os.listdir() order, which is filesystem dependent: items are compared keyed by path."""
import contextlib
import io
import json
import os

import pytest

from fakebob_amd import attack_main as AM
from tests.golden import driver_site as DS

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def gold():
    with open(os.path.join(G, "g10_driver.json")) as r:
        return {(c["task"], c["attack_type"]): c for c in json.load(r)["cases"]}


@pytest.fixture()
def site(tmp_path):
    DS.make_site(str(tmp_path))
    old = os.getcwd()
    os.chdir(str(tmp_path))       # the reference's paths are relative to the working directory
    yield str(tmp_path)
    os.chdir(old)


CASES = [("CSI", "untargeted"), ("CSI", "targeted"), ("OSI", "untargeted"), ("OSI", "targeted"), ("SV", "targeted")]


@pytest.mark.parametrize("task,at", CASES)
def test_build_attack_list_matches_reference_loadData(gold, site, task, at):
    c = gold[(task, at)]
    ident = "gmm-" + task + "-" + at
    out_a, out_c = os.path.join("adversarial-audio", ident), os.path.join("checkpoint", ident)
    if task == "SV":
        out_a, out_c = os.path.join(out_a, DS.SPK_IDS[0]), os.path.join(out_c, DS.SPK_IDS[0])
    model = DS.StubModel(task)
    items = AM.build_attack_list(task, at, model, "./data/test-set", "./data/illegal-set", out_a, out_c)
    assert model.decision_calls == 1                                  # ONE batched make_decisions call (:128,:200,:265)
    got = {os.path.normpath(it["wav_path"]): it for it in items}
    want = {os.path.normpath(it["wav_path"]): it for it in c["items"]}
    assert len(items) == len(c["items"]) and set(got) == set(want)
    for k, w in want.items():
        g = got[k]
        assert os.path.normpath(g["cp_path"]) == os.path.normpath(w["cp_path"])
        assert g["name"] == w["name"] and g["true"] == w["true"] and g["target"] == w["target"]
        assert DS.StubModel._code(g["audio"]) == w["code"]            # audio / 2^15 (:123-124)
    # targets of one voice come out in ascending order, like the reference's inner loop
    by_voice = {}
    for it in items:
        by_voice.setdefault((it["spk"], it["name"]), []).append(it["target"])
    for v in by_voice.values():
        assert v == sorted(v, key=lambda t: -1 if t is None else t)


@pytest.mark.parametrize("task,at", CASES)
def test_main_matches_reference_main(gold, site, task, at):
    c = gold[(task, at)]
    DS.StubBob.log = []
    spk = DS.SPK_IDS[:1] if task == "SV" else DS.SPK_IDS
    argv = ["-spk_id"] + spk + ["-task", task, "-type", at, "--streams", "1", "--seed", "5"]
    buf = io.StringIO()
    old_choice = AM.np.random.choice
    AM.np.random.choice = lambda n, k: AM.np.array([n // 2])
    try:
        with contextlib.redirect_stdout(buf):
            g, results, thr = AM.main(argv, model_factory=lambda archi, t, ml, pre, th, gid: DS.StubModel(t, th),
                                      bob_factory=DS.StubBob)
    finally:
        AM.np.random.choice = old_choice
    lines = buf.getvalue().splitlines()
    assert c["rate_line"] in lines                                    # '%d' of a float: truncation (:411)
    assert c["total_line"] in lines
    want_est = [x for x in c["log"] if x[0] == "estimate"]
    got_est = [x for x in DS.StubBob.log if x[0] == "estimate"]
    assert len(got_est) == len(want_est)                              # once for OSI / SV, never for CSI (:355-357,:393-394)
    want_att = {os.path.normpath(x[1]): x[2:] for x in c["log"] if x[0] == "attack"}
    got_att = {os.path.normpath(x[1]): list(x[2:]) for x in DS.StubBob.log if x[0] == "attack"}
    assert got_att == want_att                                        # threshold, true, target, flag per checkpoint path
    for p in c["written"]:
        assert os.path.isfile(p)                                      # scipy write(adver_audio_path, fs, adver_audio)
    assert g[0] == sum(1 for x in c["log"] if x[0] == "attack" and x[5] == 1) and g[1] == len(want_att)
