#!/usr/bin/env python
"""Golden vectors for fakebob_amd.evaluate: `set_threshold` and the OSI metric expressions of the
reference's test.py.  test.py is a script (it scores real data at import), so only the function
definition / the metric expressions are evaluated here: the `set_threshold` source is taken from the
reference file with `ast` at generation time and run on random score sets; nothing but the resulting
numbers is stored.   python tests/golden/make_golden_eval.py  ->  tests/golden/g9_evaluate.npz
"""
import ast
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/test.py"

src = open(REF).read()
tree = ast.parse(src)
fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "set_threshold"][0]
class _NP(object):   # numpy 2 dropped the `np.infty` alias the reference uses (written for numpy 1.15)
    infty = np.inf

    def __getattr__(self, k):
        return getattr(np, k)


ns = {"np": _NP()}
exec(compile(ast.Module(body=[fn], type_ignores=[]), REF, "exec"), ns)
ref_set_threshold = ns["set_threshold"]

rng = np.random.default_rng(99)
out = {}
cases = []
for i, (nt, nu, sep) in enumerate([(40, 200, 2.0), (7, 5, 0.3), (100, 100, 0.0), (1, 10, 1.0), (25, 60, 5.0)]):
    st = rng.normal(sep, 1.0, nt)
    su = rng.normal(0.0, 1.0, nu)
    if i == 2:                       # ties between the two sets and inside the target set
        st = np.round(st, 1)
        su = np.round(su, 1)
    thr, frr, far = ref_set_threshold(list(st), list(su))
    out["st%d" % i], out["su%d" % i] = st, su
    out["res%d" % i] = np.array([thr, frr, far], np.float64)
    cases.append(i)
# OSI expressions (test.py:258-277), evaluated literally on synthetic score matrices
for i, (n, S) in enumerate([(30, 5), (12, 3)]):
    target_scores = rng.normal(0, 1, (n, S))
    target_label_list = rng.integers(0, S, n)
    target_scores[np.arange(n), target_label_list] += 1.5
    untarget_scores = rng.normal(-0.5, 1, (2 * n, S))
    max_spk_index = np.argmax(target_scores, axis=1)
    keep_utt_index = np.argwhere(max_spk_index == target_label_list).flatten()
    keep_max_scores = np.max(target_scores[keep_utt_index], axis=1)
    max_scores = np.max(untarget_scores, axis=1)
    threshold, frr, far = ref_set_threshold([s for s in keep_max_scores], [s for s in max_scores])
    IER_cnt = np.intersect1d(np.argwhere(target_scores[:, max_spk_index] >= threshold).flatten(),
                             np.argwhere(max_spk_index != target_label_list).flatten()).size
    IER = IER_cnt * 100 / n
    out["osi_ts%d" % i], out["osi_lab%d" % i], out["osi_us%d" % i] = target_scores, target_label_list, untarget_scores
    out["osi_res%d" % i] = np.array([threshold, frr, IER, far], np.float64)
out["n_cases"] = np.array([len(cases), 2])
np.savez(os.path.join(HERE, "g9_evaluate.npz"), **out)
print("wrote g9_evaluate.npz")
