"""CPU check of the polynomial sin / cos / atan kernels the DEVICE's SE(3) chain uses (dvo_math.h: ksin_d, kcos_d, katan_d): their
constants are read out of the header and the same formulas are evaluated in double here against libm.  Guards the constants against an
edit nobody notices until a pose is off in the tenth digit; the device-side comparison (all of the domain, FMA arithmetic) is
dvo_selftest_trig in the GPU suite."""
import math
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "direct-visual-odometry_amd", "csrc", "dvo_math.h")).read()
NUM = r"-?\d\.\d+e[+-]\d+"


def _body(name):
    i = SRC.index("double %s(double x)" % name)
    return SRC[i:SRC.index("\n}\n", i)]


def test_sin_cos_kernels():
    s = [float(v) for v in re.findall(NUM, _body("ksin_d"))]      # S6, S5, S4, S3, S2 (Horner order) then S1
    c = [float(v) for v in re.findall(NUM, _body("kcos_d"))]      # C6 .. C1
    assert len(s) == 6 and len(c) == 6
    S6, S5, S4, S3, S2, S1 = s
    C6, C5, C4, C3, C2, C1 = c
    assert abs(S1 + 1 / 6) < 1e-15 and abs(C1 - 1 / 24) < 1e-15
    x = np.random.RandomState(0).uniform(-math.pi / 4, math.pi / 4, 200000)
    z = x * x
    sin = x + z * x * (S1 + z * (S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)))))
    r = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))))
    hz = 0.5 * z
    w = 1.0 - hz
    cos = w + (((1.0 - w) - hz) + z * r)
    assert np.max(np.abs(sin - np.sin(x)) / np.maximum(np.abs(np.sin(x)), 1e-300)) < 4e-16
    assert np.max(np.abs(cos - np.cos(x)) / np.abs(np.cos(x))) < 4e-16


def test_atan_kernel_and_reduction_constants():
    b = _body("katan_d")
    nums = [float(v) for v in re.findall(NUM, b)]
    # four (hi, lo) pairs of the break points, then the odd / even polynomial coefficients (s1: aT10, 8, 6, 4, 2, 0; s2: aT9, 7, 5, 3, 1)
    assert len(nums) == 8 + 11
    hi = nums[0:8:2]; lo = nums[1:8:2]
    for h, l, ref in zip(hi, lo, (math.atan(0.5), math.atan(1.0), math.atan(1.5), math.pi / 2)):
        assert abs((h + l) - ref) < 1e-16 and abs(l) < 1e-16
    a10, a8, a6, a4, a2, a0, a9, a7, a5, a3, a1 = nums[8:]

    def fatan(x):
        if x < 0.4375:
            k = -1
        elif x < 0.6875:
            k = 0; x = (2.0 * x - 1.0) / (2.0 + x)
        elif x < 1.1875:
            k = 1; x = (x - 1.0) / (x + 1.0)
        elif x < 2.4375:
            k = 2; x = (x - 1.5) / (1.0 + 1.5 * x)
        else:
            k = 3; x = -1.0 / x
        z = x * x; w = z * z
        s1 = z * (a0 + w * (a2 + w * (a4 + w * (a6 + w * (a8 + w * a10)))))
        s2 = w * (a1 + w * (a3 + w * (a5 + w * (a7 + w * a9))))
        return x - x * (s1 + s2) if k < 0 else hi[k] - ((x * (s1 + s2) - lo[k]) - x)

    rs = np.random.RandomState(1)
    ts = np.concatenate([rs.uniform(0, 3, 100000), 10 ** rs.uniform(-8, 8, 50000), [0.4375, 0.6875, 1.1875, 2.4375]])
    assert max(abs(fatan(float(t)) - math.atan(t)) / math.atan(t) for t in ts) < 4e-16
    # the two-constant reduction of sincos_dev and the pi split of atan2_dev
    i = SRC.index("void sincos_dev(double x")
    red = [float(v) for v in re.findall(r"-?\d\.\d+(?:e[+-]\d+)?", SRC[i:SRC.index("\n}\n", i)])]
    assert any(abs(v - 2 / math.pi) < 1e-16 for v in red)
    assert any(abs(v - math.pi / 2) < 1e-16 for v in red) and any(abs(v - 6.123233995736766e-17) < 1e-30 for v in red)
    j = SRC.index("double atan2_dev(double y")
    a2 = [float(v) for v in re.findall(NUM, SRC[j:SRC.index("\n}\n", j)])]
    assert any(abs(v - math.pi) < 1e-15 for v in a2) and any(abs(v - 1.2246467991473532e-16) < 1e-30 for v in a2)
