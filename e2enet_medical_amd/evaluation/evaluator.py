"""Label-map evaluation behind ``nnUNetTrainer_simple.validate``: the reference's ``aggregate_scores`` record
(e2enet/evaluation/evaluator.py:321-400) with its default confusion-matrix metrics (:37-51; formulas and empty-mask rules of
e2enet/evaluation/metrics.py:106-121, :601-790).

Host tooling, not the hot path: the inputs are the exported uint8 label volumes.  One joint histogram of (reference, test)
labels per case replaces the reference's thirteen boolean passes per label; the numbers are the same integers divided the same
way.  The surface-distance metrics (``default_advanced_metrics``) are not evaluated by ``aggregate_scores`` either
(``evaluate(advanced=False)``).
"""
import hashlib
import json
from collections import OrderedDict
from datetime import datetime

import numpy as np

DEFAULT_METRICS = ["False Positive Rate", "Dice", "Jaccard", "Precision", "Recall", "Accuracy", "False Omission Rate",
                   "Negative Predictive Value", "False Negative Rate", "True Negative Rate", "False Discovery Rate",
                   "Total Positives Test", "Total Positives Reference"]          # evaluator.py:37-51


def confusion_counts(test: np.ndarray, reference: np.ndarray, labels):
    """label -> (tp, fp, tn, fn) of the binary maps ``test == label`` / ``reference == label`` (metrics.py:67-80)."""
    assert test.shape == reference.shape, "Shape mismatch: {} and {}".format(test.shape, reference.shape)
    t = np.asarray(test).reshape(-1).astype(np.int64)
    r = np.asarray(reference).reshape(-1).astype(np.int64)
    lo = int(min(t.min(), r.min(), 0)) if t.size else 0
    n = int(max(t.max(), r.max(), max(int(l) for l in labels))) - lo + 1 if t.size else 1
    joint = np.bincount((r - lo) * n + (t - lo), minlength=n * n).reshape(n, n)      # [reference, test]
    size = int(t.size)
    out = OrderedDict()
    for l in labels:
        i = int(l) - lo
        tp = int(joint[i, i]) if 0 <= i < n else 0
        pos_t = int(joint[:, i].sum()) if 0 <= i < n else 0
        pos_r = int(joint[i, :].sum()) if 0 <= i < n else 0
        fp, fn = pos_t - tp, pos_r - tp
        out[l] = (tp, fp, size - tp - fp - fn, fn)
    return out


def metrics_from_counts(tp, fp, tn, fn, nan_for_nonexisting=True):
    """The thirteen default metrics of one binary pair, keyed and SORTED like the reference's result dicts
    (``self.metrics.sort()``, evaluator.py:166).  Empty / full rules: metrics.py:106-121, :601-790."""
    nan = float("NaN") if nan_for_nonexisting else 0.
    size = tp + fp + tn + fn
    test_empty, test_full = (tp + fp) == 0, (tp + fp) == size
    ref_empty, ref_full = (tp + fn) == 0, (tp + fn) == size
    dice = nan if (test_empty and ref_empty) else float(2. * tp / (2 * tp + fp + fn))
    jaccard = nan if (test_empty and ref_empty) else float(tp / (tp + fp + fn))
    precision = nan if test_empty else float(tp / (tp + fp))
    sensitivity = nan if ref_empty else float(tp / (tp + fn))
    specificity = nan if ref_full else float(tn / (tn + fp))
    fomr = nan if test_full else float(fn / (fn + tn))
    res = {"False Positive Rate": 1 - specificity, "Dice": dice, "Jaccard": jaccard, "Precision": precision,
           "Recall": sensitivity, "Accuracy": float((tp + tn) / (tp + fp + tn + fn)), "False Omission Rate": fomr,
           "Negative Predictive Value": 1 - fomr, "False Negative Rate": 1 - sensitivity, "True Negative Rate": specificity,
           "False Discovery Rate": 1 - precision, "Total Positives Test": tp + fp, "Total Positives Reference": tp + fn}
    return OrderedDict((k, res[k]) for k in sorted(res))


def evaluate_pair(test: np.ndarray, reference: np.ndarray, labels, nan_for_nonexisting=True):
    """label (str) -> metric dict: ``Evaluator.evaluate`` for a list of integer labels (evaluator.py:216-226)."""
    counts = confusion_counts(test, reference, labels)
    return OrderedDict((str(l), metrics_from_counts(*counts[l], nan_for_nonexisting)) for l in labels)


def aggregate_scores(cases, labels, nanmean=True, json_output_file=None, json_name="", json_description="",
                     json_author="Fabian", json_task=""):
    """``cases``: iterable of (test array, reference array, test name, reference name).  Returns the reference's
    ``all_scores`` ({"all": [per case], "mean": {label: {metric: mean}}}) and writes the reference's summary.json
    (evaluator.py:353-400) when ``json_output_file`` is given."""
    all_scores = OrderedDict()
    all_scores["all"] = []
    all_scores["mean"] = OrderedDict()
    for test, ref, test_name, ref_name in cases:
        res = evaluate_pair(test, ref, labels)
        if test_name is not None:
            res["test"] = test_name
        if ref_name is not None:
            res["reference"] = ref_name
        all_scores["all"].append(res)
        for label, score_dict in res.items():
            if label in ("test", "reference"):
                continue
            dst = all_scores["mean"].setdefault(label, OrderedDict())
            for score, value in score_dict.items():
                dst.setdefault(score, []).append(value)
    for label in all_scores["mean"]:
        for score in all_scores["mean"][label]:
            vals = all_scores["mean"][label][score]
            with np.errstate(all="ignore"):
                import warnings
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore", RuntimeWarning)          # (all-NaN slice: the reference prints the warning)
                    all_scores["mean"][label][score] = float(np.nanmean(vals)) if nanmean else float(np.mean(vals))
    if json_output_file is not None:
        json_dict = OrderedDict()
        json_dict["name"] = json_name
        json_dict["description"] = json_description
        json_dict["timestamp"] = str(datetime.today())
        json_dict["task"] = json_task
        json_dict["author"] = json_author
        json_dict["results"] = all_scores
        json_dict["id"] = hashlib.md5(json.dumps(json_dict).encode("utf-8")).hexdigest()[:12]
        with open(json_output_file, "w") as f:
            json.dump(json_dict, f, indent=4, sort_keys=True)            # (batchgenerators save_json: indent 4, sorted keys)
    return all_scores
