#!/usr/bin/env python3
"""Parallel-in-time training (tier b) against the exact path (tier a): time, SER, tap / output deviation, device report.

    python scripts/pit_exp.py [--workload c3|ns|c2] [--nsym N] [--seeds 1000,1001] [--linewidth Hz] [--variants default,...]
"""
import argparse, json, os, sys, time
os.environ.setdefault("QAMPY_HIP_PIT_TIMING", "all")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from qampy_amd import synth, _lib
from qampy_amd.pipeline import ResidentReceiver
from qampy_amd.core import ber_functions as ber

VARIANTS = {
    "default": {},
    "tol1e-2": dict(tol=1e-2),
    "tol5e-2": dict(tol=5e-2),
    "noseed": dict(phase_seed=0),
    "gear4": dict(gear=4.),
    "gear16": dict(gear=16., acq_bound=0.16),
    "S256": dict(segments=256),
    "S1024": dict(segments=1024),
    "S2048": dict(segments=2048),
    "S4096": dict(segments=4096),
    "S2048/4096": (dict(segments=2048), dict(segments=4096)),
    "plateau.8": dict(acq_plateau=0.8),
    "plateau.95": dict(acq_plateau=0.95),
    "noacq": dict(acquire=0),
    "nocorr": dict(correction=0),
    "beta0": dict(corr_beta=0.),
    "beta1": dict(corr_beta=1.),
    "beta3": dict(corr_beta=3.),
    "beta6": dict(corr_beta=6.),
    "nocorr8": dict(correction=0, max_passes=8, tol=1e-3),
    "corr8": dict(max_passes=8, tol=1e-3),
    "tol1=.05": (dict(tol=0.05), {}),
    "tol1=.035": (dict(tol=0.035), {}),
    "tol1=.1": (dict(tol=0.1), {}),
    "S1=2816": (dict(segments=2816), {}),
    "S1=2304": (dict(segments=2304), {}),
    "b2=.3": ({}, dict(corr_beta=0.3)),
    "b2=.7": ({}, dict(corr_beta=0.7)),
    "b2=1.5": ({}, dict(corr_beta=1.5)),
    "S2=2944": ({}, dict(segments=2944)),
    "S2=1984": ({}, dict(segments=1984)),
    "acq6144": (dict(acq_chunk=3072, acq_max=6144), {}),
    "acq4096": (dict(acq_chunk=2048, acq_max=4096), {}),
    "acq5120": (dict(acq_chunk=2560, acq_max=5120), {}),
    "S1=3328": (dict(segments=3328), {}),
    "S1=3840": (dict(segments=3840), {}),
    "chunk2048": (dict(acq_chunk=2048), {}),
    "gear16c2048": (dict(gear=16., acq_bound=0.16, acq_chunk=2048), {}),
    "gear12c2048": (dict(gear=12., acq_bound=0.12, acq_chunk=2048), {}),
    "gear16c1024": (dict(gear=16., acq_bound=0.16, acq_chunk=1024), {}),
    "c1024": (dict(acq_chunk=1024), {}),
    "p.8c2048": (dict(acq_plateau=0.8, acq_chunk=2048), {}),
    "p.7c2048": (dict(acq_plateau=0.7, acq_chunk=2048), {}),
    "p.8": (dict(acq_plateau=0.8), {}),
    "tolF=.01": ({}, dict(tol=0.01)),
    "tolF=.005": ({}, dict(tol=0.005)),
    "tolF=.0025": ({}, dict(tol=0.0025)),
    "tol1=.02,F=.005": (dict(tol=0.02), dict(tol=0.005)),
    "tol1=.1,F=.005": (dict(tol=0.1), dict(tol=0.005)),
}

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c3")
ap.add_argument("--nsym", type=int, default=None)
ap.add_argument("--seeds", default="1000")
ap.add_argument("--linewidth", type=float, default=None)
ap.add_argument("--snr", type=float, default=None)
ap.add_argument("--variants", default="default")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--no-exact", action="store_true")
ap.add_argument("--mu", default=None, help="step sizes of the stages, comma separated")
args = ap.parse_args()
cfg = dict(bench.WORKLOADS[args.workload])
if args.linewidth is not None:
    cfg["linewidth"] = args.linewidth
if args.snr is not None:
    cfg["snr_db"] = args.snr
if args.mu:
    cfg["mu"] = tuple(float(x) for x in args.mu.split(","))
nsym = args.nsym or cfg["nsym"]
_lib.init(0)
out = []
for seed in [int(s) for s in args.seeds.split(",")]:
    d = synth.make_capture_dev(cfg["M"], nsym, nmodes=2, snr_db=cfg["snr_db"], theta=np.pi / 5.6, dgd=30e-12, linewidth=cfg["linewidth"], seed=seed)
    kw = dict(methods=cfg["methods"], Niter=cfg["niter"], Mtestangles=cfg["A"], Nbps=cfg["Nbps"], alphabet=d["alphabet_host"])

    def run(tier, pit=None):
        rx = ResidentReceiver(2, nsym * 2, 2, cfg["M"], cfg["ntaps"], cfg["mu"], tier=tier, pit=pit, **kw)
        rx.E.copy_from(d["E"])
        rx.run(); _lib.sync()
        ts = []
        for _ in range(args.reps):
            t0 = time.perf_counter(); rx.run(); _lib.sync(); ts.append(time.perf_counter() - t0)
        # per-stage times
        names = ["gram"] + ["train%d" % (s + 1) for s in range(rx.nstage)] + ["apply", "recover"]
        fns = [rx.build_gram] + [lambda s=s: rx.train(s) for s in range(rx.nstage)] + [rx.apply, rx.recover]
        ev = [_lib.Event() for _ in range(len(fns) + 1)]
        rx.reset(); ev[0].record()
        for j, f in enumerate(fns):
            f(); ev[j + 1].record()
        _lib.sync()
        st = {n: round(ev[j + 1].elapsed_ms(ev[j]), 3) for j, n in enumerate(names)}
        ser = ber.cal_ser_dev(rx.out, d["idx_tx"], rx.alphabet, 256, 8192, 2000)
        return rx, min(ts), st, ser

    ref = None
    if not args.no_exact:
        ref, t_ref, st_ref, ser_ref = run("a")
        w_ref, eq_ref = ref.wxy.to_host(), ref.eq.to_host()
        e_ref = [e.to_host() for e in ref.err]
        out.append(dict(seed=seed, tier="a", ms=round(t_ref * 1e3, 2), MSym_s=round(nsym / t_ref / 1e6, 2), stages_ms=st_ref,
                        errors=[r["errors"] for r in ser_ref], rot=[r["rotation"] for r in ser_ref]))
        print(json.dumps(out[-1]), flush=True)
    for v in args.variants.split(","):
        if v.startswith("g:"):          # g:S1:S2:tol1:tol2[:maxpass]  generic two-stage variant
            f = v.split(":")
            mp = int(f[5]) if len(f) > 5 else 0
            VARIANTS[v] = (dict(segments=int(f[1]), tol=float(f[3]), max_passes=mp), dict(segments=int(f[2]), tol=float(f[4]), max_passes=mp))
        if v.startswith("a:"):          # a:chunk:max  acquisition of the first stage
            f = v.split(":")
            VARIANTS[v] = dict(acq_chunk=int(f[1]), acq_max=int(f[2])) if len(cfg["methods"]) == 1 else (dict(acq_chunk=int(f[1]), acq_max=int(f[2])), {})
        if v.startswith("t:"):          # t:tol1:tol2[:start[:maxpass]]  automatic grid
            f = v.split(":")
            st = int(f[3]) if len(f) > 3 else 0
            mp = int(f[4]) if len(f) > 4 else 0
            VARIANTS[v] = (dict(tol=float(f[1]), start=st, max_passes=mp), dict(tol=float(f[2]), start=st, max_passes=mp))
        rx, t, st, ser = run("b", VARIANTS[v])
        ptm = [rx.pit_timing[s_] for s_ in range(rx.nstage)]
        rec = dict(seed=seed, tier="b", variant=v, pass_ms=[[round(x, 3) for x in p_[0]] for p_ in ptm], acq_ms=[round(p_[1], 3) for p_ in ptm], ms=round(t * 1e3, 2), MSym_s=round(nsym / t / 1e6, 2), stages_ms=st,
                   errors=[r["errors"] for r in ser], rot=[r["rotation"] for r in ser], report=rx.pit_reports())
        if ref is not None:
            w, eq = rx.wxy.to_host(), rx.eq.to_host()
            # deviation modulo a common quarter turn per output mode (the error functions' symmetry)
            dev_t, dev_o = [], []
            for m in range(w.shape[0]):
                c = np.vdot(w[m].ravel(), w_ref[m].ravel())
                g = 1j ** int(np.rint(np.angle(c) / (np.pi / 2)))
                dev_t.append(float(np.linalg.norm(w_ref[m] - g * w[m]) / np.linalg.norm(w_ref[m])))
                dev_o.append(float(np.sqrt(np.mean(np.abs(eq_ref[m] - g * eq[m]) ** 2))))
            # error traces against the exact path's (same start taps -> same trajectory up to the tolerance): rms over the sweep and
            # the worst of 256 blocks, per stage and mode
            err_dev = []
            for e_b, e_a in zip((x.to_host() for x in rx.err), e_ref):
                row = []
                for m in range(e_b.shape[0]):
                    c = np.vdot(e_b[m], e_a[m]); g = 1j ** int(np.rint(np.angle(c) / (np.pi / 2)))
                    d2 = np.abs(e_a[m] - g * e_b[m]) ** 2
                    nb_ = d2.size // 256
                    blk = np.sqrt(d2[:nb_ * 256].reshape(256, nb_).mean(axis=1))
                    row.append((float(np.sqrt(d2.mean())), float(blk.max()), float(blk[-1])))
                err_dev.append(row)
            rec["err_dev_rms_worstblock_lastblock"] = err_dev
            n8 = rx.err[0].shape[1] // 8
            rec.update(tap_dev_rel=dev_t, eq_rms_dev=dev_o,
                       err_pow_by_eighth=[[[round(float(np.mean(np.abs(e[m, i * n8:(i + 1) * n8]) ** 2)), 5) for i in range(8)] for m in range(2)]
                                          for e in (x.to_host() for x in rx.err)],
                       err_pow_exact=[[[round(float(np.mean(np.abs(e[m, i * n8:(i + 1) * n8]) ** 2)), 5) for i in range(8)] for m in range(2)] for e in e_ref])
        out.append(rec)
        print(json.dumps(rec), flush=True)
        print("##", v, seed, rec["ms"], st, rec["errors"], [(r["segments"], r["passes"], [round(x, 4) for x in r["defect"]], [round(x, 5) for x in r["deviation_taps"]], [round(x, 5) for x in r["deviation_taps_worst"]]) for r in rec["report"]], [[tuple(round(v, 5) for v in t) for t in row] for row in rec.get("err_dev_rms_worstblock_lastblock", [])],
              [round(x, 5) for x in rec.get("eq_rms_dev", [])], [round(x, 5) for x in rec.get("tap_dev_rel", [])], rec["pass_ms"], rec["acq_ms"], flush=True)
        del rx
    del ref
print(json.dumps(dict(what="exact (tier a) vs parallel-in-time (tier b)", workload=args.workload, nsym=nsym, results=out)))
