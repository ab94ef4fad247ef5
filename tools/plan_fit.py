#!/usr/bin/env python3
"""Fit the work-item planner's model of a CU (cu_finish_estimate() in srcnn_cpp_amd/csrc/srcnn_plan.cpp) to the stamped production
launch: tools/plan_fit_collect.sh writes, per CU, the rows of the first-dispatched workgroup and of the one that joined it and
their exit times under several row splits.  Model: while both run, the first takes `fast` us per row and the second `slow`; the
one left alone takes `alone`; each pays a fixed start.  Prints the least-squares rates (for SRCNN_DEBUG_RATES / the constants)
and the residuals of the current constants.      usage: python tools/plan_fit.py gpurun_out/plan_fit.txt"""
import sys

import numpy as np
from scipy.optimize import least_squares


def finish(p, rf, rs):
    fast, slow, alone, sf, ss = p[:5]
    alone_f = p[5] if len(p) > 5 else alone
    tf, ts = sf + rf * fast, ss + rs * slow
    f_first = tf <= ts
    fin_f = np.where(f_first, tf, ts + np.maximum(0.0, rf - (ts - sf) / fast) * alone_f)
    fin_s = np.where(f_first, tf + np.maximum(0.0, rs - (tf - ss) / slow) * alone, ts)
    return fin_f, fin_s


def main():
    d = np.loadtxt(sys.argv[1])
    cur = np.array([6.40, 8.40, 3.76, 3.63, 5.44])
    sizes = sorted({(int(w), int(h)) for w, h in d[:, :2]})
    for label, sel in [("all sizes", np.ones(len(d), bool))] + [(f"{w}x{h}", (d[:, 0] == w) & (d[:, 1] == h)) for w, h in sizes]:
        rf, rs, fin_f, fin_s = d[sel, 3], d[sel, 4], d[sel, 6], d[sel, 8]
        cu = np.maximum(fin_f, fin_s)

        def resid(p):
            a, b = finish(p, rf, rs)
            return np.maximum(a, b) - cu                       # the planner balances the CU's finish

        def resid_both(p):
            a, b = finish(p, rf, rs)
            return np.concatenate([a - fin_f, b - fin_s])

        r0 = resid(cur)
        print(f"{label}: {sel.sum()} CUs;  current constants: CU-finish residual mean {r0.mean():+.2f} us, sd {r0.std():.2f}, "
              f"slope vs fast rows {np.polyfit(rf, r0, 1)[0]:+.3f} us/row")
        for name, fn, p0 in (("5 rates, CU finish", resid, cur), ("5 rates, both exits", resid_both, cur),
                             ("6 rates (alone differs), both exits", resid_both, np.append(cur, cur[2]))):
            fit = least_squares(fn, p0)
            r = resid(fit.x)
            print(f"    {name}: {np.round(fit.x, 3).tolist()}  residual sd {r.std():.2f}, slope vs fast rows {np.polyfit(rf, r, 1)[0]:+.3f}")


if __name__ == "__main__":
    main()
