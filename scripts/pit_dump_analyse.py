#!/usr/bin/env python3
"""DEBUG: what holds the deviation estimate of a tier-b sweep up - per eigen-direction and per segment - from QAMPY_HIP_PIT_DUMP files."""
import sys, glob, struct, re
import numpy as np
files = sorted(glob.glob(sys.argv[1] + "_c*_p*.bin"), key=lambda f: [int(x) for x in re.findall(r"_c(\d+)_p(\d+)", f)[0]])
only = sys.argv[2] if len(sys.argv) > 2 else None
for fn in files:
    if only and ("_c%s_" % only) not in fn:
        continue
    raw = open(fn, "rb").read()
    ntot, ncol, nsel, S = struct.unpack("4i", raw[:16])
    o = 16
    lam = np.frombuffer(raw, np.float64, ntot, o); o += 8 * ntot
    D = np.frombuffer(raw, np.complex64, ntot * ncol, o).reshape(ntot, S, nsel); o += 8 * ntot * ncol
    X = np.frombuffer(raw, np.complex64, ntot * ncol, o).reshape(ntot, S, nsel); o += 8 * ntot * ncol
    Y = np.frombuffer(raw, np.complex64, ntot * ncol, o).reshape(ntot, S, nsel); o += 8 * ntot * ncol
    th = np.frombuffer(raw, np.float64, 2 * ncol, o).reshape(S, nsel, 2); o += 16 * ncol
    th = th[..., 0] + 1j * th[..., 1]
    e = lam[:, None, None] * np.abs(D) ** 2
    tot = e.sum(0)
    s_w, j_w = np.unravel_index(tot.argmax(), tot.shape)
    print(fn.split("/")[-1], "S", S, "rms est", round(float(np.sqrt(e.sum() / (S * nsel))), 5), "worst seg", (int(s_w), int(j_w)), round(float(np.sqrt(tot.max())), 4))
    q = np.linspace(0, S, 9).astype(int)
    print("   by eighth:", [round(float(np.sqrt(tot[a:b].mean())), 5) for a, b in zip(q[:-1], q[1:])])
    # around the worst segment: correction size, frame, and how the (updated) start taps of s relate to the end taps of s - 1
    for s in range(max(1, s_w - 2), min(S, s_w + 4)):
        xs, yp = X[:, s, j_w] - D[:, s, j_w], Y[:, s - 1, j_w] * th[s - 1, j_w]          # start taps of THIS pass in frame 0 (X was updated by D since)
        c = np.vdot(yp * lam, xs) / np.vdot(yp * lam, yp)                  # x ~ c y (lambda-weighted)
        res = xs - c * yp
        print("      s=%d  |D|=%.4f  theta=%.3f rad  x[s] vs theta y[s-1]: scale %.4f phase %.4f rad, residual %.4f" % (
            s, float(np.sqrt(tot[s, j_w])), float(np.angle(th[s, j_w])), abs(c), float(np.angle(c)), float(np.sqrt((lam * np.abs(res) ** 2).sum() / (lam * np.abs(yp) ** 2).sum()))))
