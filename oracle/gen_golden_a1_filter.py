#!/usr/bin/env python3
"""Golden vectors for the Quadrupedal action filter (quadrupedal/robots/action_filter.py), recorded from the UNMODIFIED
classes. TEST INFRASTRUCTURE; build container only (needs /root/reference and scipy):

    python oracle/gen_golden_a1_filter.py

`Minitaur._FilterAction` (robots/minitaur.py:1448-1457) low-passes the policy's motor commands with
`ActionFilterButter(sampling_rate = 1 / (time_step * action_repeat), num_joints = 12)` (:1438-1443): a 2nd-order
Butterworth low-pass at 4 Hz whose history is initialised with the current motor angles on the first step of an episode.
Cases: that default; a band-pass (per-joint cut-offs, order 2: 4-deep history); `ActionFilterExp`."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("METAGYM_REFERENCE", "/root/reference")
OUT = os.path.join(ROOT, "tests", "golden", "a1_filter.npz")


def main():
    if not os.path.isdir(REF):
        raise SystemExit("reference not mounted at %s — run in the build container" % REF)
    sys.path.insert(0, os.path.join(REF, "metagym", "quadrupedal", "robots"))
    import action_filter            # the module alone: it needs numpy + scipy only
    out = {"numpy_version": np.array(np.__version__)}
    import scipy
    out["scipy_version"] = np.array(scipy.__version__)
    cases = {
        "butter_default": action_filter.ActionFilterButter(sampling_rate=1 / (0.002 * 13), num_joints=12),
        "butter_bandpass": action_filter.ActionFilterButter(lowcut=[str(0.5 + 0.1 * i) for i in range(12)],
                                                            highcut=[str(3.0 + 0.2 * i) for i in range(12)],
                                                            sampling_rate=1 / 0.026, num_joints=12),
        "exp": action_filter.ActionFilterExp(alpha=[str(0.2 + 0.05 * i) for i in range(12)], num_joints=12),
    }
    for name, f in cases.items():
        rs = np.random.RandomState(len(name))
        out[name + "/a"], out[name + "/b"] = f.a.copy(), f.b.copy()
        xs, ys, kinds = [], [], []
        for episode in range(2):
            f.reset()
            if episode == 1:
                x0 = np.array([0, 0.9, -1.8] * 4) + rs.uniform(-0.1, 0.1, 12)
                f.init_history(x0)                         # minitaur.py:1452-1454
                xs.append(x0); ys.append(x0); kinds.append(2)
            else:
                xs.append(np.zeros(12)); ys.append(np.zeros(12)); kinds.append(1)
            for k in range(25):
                x = np.array([0, 0.9, -1.8] * 4) + rs.uniform(-0.5, 0.5, 12)
                y = f.filter(x)
                xs.append(x); ys.append(np.array(y)); kinds.append(0)
        out[name + "/x"], out[name + "/y"], out[name + "/kind"] = np.array(xs), np.array(ys), np.array(kinds)
    out["cases"] = np.array(list(cases))
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
