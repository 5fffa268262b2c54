#!/usr/bin/env python3
"""Every mantissa pair (2^23 x 2^23) through the regularize kernels' short division against the IEEE quotient, on the GPU
(dvo_selftest_division in slices of 2^17 values of b; a few minutes on an MI355X).  Prints a progress line per 2^20 values of b."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "direct-visual-odometry_amd"))
import dvo_amd as dvo

t0 = time.time()
total = bad_total = 0
for c in range(8):
    n, bad, first = dvo.selftest_division(c << 20, 1, 1 << 20)
    total += n; bad_total += bad
    print("b mantissas 0x%06x..0x%06x x all a: %d pairs, %d mismatches%s  (%.0f s)" % (
        c << 20, ((c + 1) << 20) - 1, n, bad, "" if not bad else "  first: mb=0x%06x ma=0x%06x" % (first >> 23, first & 0x7fffff), time.time() - t0), flush=True)
print("all %d mantissa pairs: %d mismatches" % (total, bad_total))
n, bad, first = dvo.selftest_sqrt()
print("sqrt: %d inputs in [2^-100, 2^100]: %d mismatches" % (n, bad))
sys.exit(1 if bad_total or bad else 0)
