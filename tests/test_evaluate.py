"""fakebob_amd.evaluate against golden vectors produced by the reference's own `set_threshold` and metric
expressions (tests/golden/make_golden_eval.py)."""
import os

import numpy as np

from fakebob_amd import evaluate as E

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_evaluate.npz"))


def test_set_threshold_matches_reference():
    for i in range(int(G["n_cases"][0])):
        thr, frr, far = E.set_threshold(G["st%d" % i], G["su%d" % i])
        assert np.array_equal(np.array([thr, frr, far]), G["res%d" % i])


def test_osi_metrics_match_reference_expressions():
    for i in range(int(G["n_cases"][1])):
        got = E.osi_metrics(G["osi_ts%d" % i], G["osi_lab%d" % i], G["osi_us%d" % i])
        assert np.array_equal(np.array(got), G["osi_res%d" % i])


def test_csi_accuracy():
    assert E.csi_accuracy([0, 1, 2, 2], [0, 1, 1, 2]) == 75.0
