#!/usr/bin/env python3
"""Run-to-run spread of the 320-step trajectory test (the weight gradients are summed with atomics, whose order varies):
the last epoch's mean PSNR against the reference trajectory's, N times.   python tools/microbench/psnr_spread.py 10"""
import contextlib
import io
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np   # noqa: E402
import torch         # noqa: E402
import test_gpu_train as T   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
worst = []
for i in range(n):
    buf = io.StringIO()
    ok = True
    try:
        with contextlib.redirect_stdout(buf):
            T.test_psnr_after_320_steps_matches_the_reference_trajectory(dev)
    except AssertionError:
        ok = False
    m = re.search(r"epoch psnr gpu \[(.*?)\] reference \[(.*?)\]", buf.getvalue().replace("\n", " "))
    a = np.array([float(v) for v in m.group(1).split()])
    b = np.array([float(v) for v in m.group(2).split()])
    worst.append(a[-1] - b[-1])
    print("run %d: last-epoch diff %+.4f  per epoch %s  %s" % (i, a[-1] - b[-1], np.round(a - b, 3), "ok" if ok else "FAILED"), flush=True)
worst = np.array(worst)
print("last-epoch signed diff: mean %+.4f std %.4f max |.| %.4f" % (worst.mean(), worst.std(), np.abs(worst).max()))
