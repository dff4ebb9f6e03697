#!/usr/bin/env python3
"""
bench.py - equalised MSym/s of the adaptive-equaliser + carrier-recovery hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|ns] [--nsym S] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path (dual-mode tap training -> filter application -> blind phase search, unwrap and
de-rotation of every mode) over one synthetic dual-polarisation 2 SPS capture that is already resident in HBM.  With N
GPUs every rank processes its own independent channel (seed 1000 + rank, BASELINE.json config 4): weak scaling, no
collective on the data path; torch.distributed (RCCL) is used only for the barriers, the max-over-ranks of the elapsed
time and the sum of the symbol-error counters.

The JSON line carries, besides the driver's contract fields:
  roofline      for the dominant kernel (the stage with the largest share of the step): algorithmic bytes per launch /
                average launch duration measured with HIP events on the library stream inside the timed region
  cpu_baseline  the oracle's reference-flag OpenMP build ("port" of the pythran loops) timed on a bounded sample of the
                same workload on this box's host cores (rank 0, N = 1 only)
  stages_ms, ser, parity_vs_cpu  supporting numbers
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# SURVEY.md §8d / BASELINE.md configs.  Step sizes / linewidth of c3 and ns: MRDE is phase sensitive and false-locks when the
# CMA stage leaves a rotated constellation (its phase diffuses ~ mu^2 * N); mu = (2e-4, 2e-4) with a 100 Hz source
# converges on both modes over 2^22 symbols (verified with the CPU oracle, seeds 1000-1001).  c3 is the 2^22-symbol variant of the north-star configuration (64-QAM, 41 taps,
# CMA -> MRDE, 64-angle BPS) and the largest single-GPU configuration in BASELINE.json's `configs`.
WORKLOADS = {
    # configs[0]: the reference's own CPU-runnable plumbing case (Scripts/cma_equaliser.py): a parity-test case, not a bench line
    "c1": dict(M=4, nsym=2 ** 16, nmodes=1, ntaps=11, methods=("cma",), mu=(1e-3,), niter=(1,), adaptive=(False,), A=None, Nbps=0,
               snr_db=14, linewidth=0., label="QPSK 1-pol 2 SPS 2^16 sym, 11-tap CMA (no carrier recovery)"),
    "c2": dict(M=16, nsym=2 ** 20, ntaps=21, methods=("mcma",), mu=(1e-3,), niter=(1,), adaptive=(False,), A=32, Nbps=20,
               snr_db=25, linewidth=50e3, label="16-QAM 2-pol 2 SPS 2^20 sym, 21-tap MCMA + 32-angle BPS"),
    "c3": dict(M=64, nsym=2 ** 22, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), niter=(1, 1), adaptive=(False, False),
               A=64, Nbps=20, snr_db=30, linewidth=100.,
               label="64-QAM 2-pol 2 SPS 2^22 sym, 41-tap dual-mode CMA->MRDE + 64-angle BPS"),
    "ns": dict(M=64, nsym=10 ** 7, ntaps=41, methods=("cma", "mrde"), mu=(2e-4, 2e-4), niter=(1, 1), adaptive=(False, False),
               A=64, Nbps=20, snr_db=30, linewidth=100.,
               label="64-QAM 2-pol 2 SPS 10^7 sym, 41-tap dual-mode CMA->MRDE + 64-angle BPS (north star)"),
}
HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def make_input(cfg, nsym, seed, host=False):
    """The capture of this rank: synthesised on the GPU (csrc/synth.hip, ~1 ms) and copied to the host once for the CPU legs;
    --host-synth uses the numpy generator instead (same impairments, frequency-domain filters, ~10 s at 2^22 symbols)."""
    from qampy_amd import synth
    from qampy_amd.signals import SignalQAM
    nm = cfg.get("nmodes", 2)
    if not host:
        d = synth.make_capture_dev(cfg["M"], nsym, nmodes=nm, os=2, snr_db=cfg["snr_db"], theta=np.pi / 5.6 if nm == 2 else None, dgd=30e-12,
                                   linewidth=cfg["linewidth"], fb=20e9, beta=0.1, seed=seed)
        return SignalQAM(d["E"].to_host(), cfg["M"], fb=d["fb"], fs=d["fs"], symbols=d["symbols"].to_host(), coded_symbols=d["alphabet_host"])
    return synth.make_capture(cfg["M"], nsym, nmodes=nm, os=2, snr_db=cfg["snr_db"], theta=np.pi / 5.6 if nm == 2 else None, dgd=30e-12,
                              linewidth=cfg["linewidth"], fb=20e9, beta=0.1, seed=seed, dtype=np.complex64)


def make_receiver(cfg, sig, segments=0, prefix=0, prefix_mu=None):
    from qampy_amd.pipeline import ResidentReceiver
    return ResidentReceiver(sig.shape[0], sig.shape[1], 2, cfg["M"], cfg["ntaps"], cfg["mu"], methods=cfg["methods"], Niter=cfg["niter"],
                            adaptive_stepsize=cfg["adaptive"], TrSyms=(None,) * len(cfg["methods"]), Mtestangles=cfg["A"],
                            Nbps=cfg["Nbps"], dtype=np.complex64, alphabet=sig.coded_symbols, segments=segments, prefix=prefix,
                            prefix_mu=prefix_mu)


def timed_steps(rx, steps, warmup, barrier_sync):
    """W warm-up passes, then exactly K passes bracketed by barrier + device sync; HIP events between the stages."""
    from qampy_amd import _lib
    stage_fns = [rx.build_gram] + [lambda s=s: rx.train(s) for s in range(rx.nstage)] + [rx.apply] + ([rx.recover] if rx.Mtestangles else [])
    for _ in range(warmup):
        rx.run()
    ev = [[_lib.Event() for _ in range(len(stage_fns) + 1)] for _ in range(steps)]
    barrier_sync()
    t0 = time.perf_counter()
    for k in range(steps):
        rx.reset()
        ev[k][0].record()
        for j, fn in enumerate(stage_fns):
            fn()
            ev[k][j + 1].record()
    barrier_sync()
    elapsed = time.perf_counter() - t0
    stage_ms = [float(np.mean([ev[k][j + 1].elapsed_ms(ev[k][j]) for k in range(steps)])) for j in range(len(stage_fns))]
    return elapsed, stage_ms


def channel_bank_run(cfg, sig, nch, steps, barrier_sync, trainer="iterative"):
    """Informational: `nch` independent captures of the workload resident on ONE GPU, all stages for all channels per step.
    One exact training chain is one workgroup, so channels side by side are how the exact recurrence fills the chip (WDM
    receivers have them).  Every channel is an independent capture generated on the device (seed 2000 + c).  NOT the headline
    `value` (BASELINE configs are single captures)."""
    from qampy_amd import _lib
    from qampy_amd.core import ber_functions as ber
    from qampy_amd.pipeline import ChannelBank
    E = np.asarray(sig)
    bank = ChannelBank(nch, E.shape[0], E.shape[1], 2, cfg["M"], cfg["ntaps"], cfg["mu"], methods=cfg["methods"], Niter=cfg["niter"],
                       adaptive_stepsize=cfg["adaptive"], TrSyms=(None,) * len(cfg["methods"]), Mtestangles=cfg["A"], Nbps=cfg["Nbps"],
                       dtype=np.complex64, alphabet=sig.coded_symbols, trainer=trainer)
    # every channel is its own capture, synthesised in HBM (csrc/synth.hip) with the workload's impairments and its own seed
    from qampy_amd import synth
    nsym_c = E.shape[1] // 2
    idx_tx = []
    for c in range(nch):
        d = synth.make_capture_dev(cfg["M"], nsym_c, nmodes=E.shape[0], os=2, snr_db=cfg["snr_db"], theta=np.pi / 5.6 if E.shape[0] == 2 else None,
                                   dgd=30e-12, linewidth=cfg["linewidth"], fb=20e9, beta=0.1, seed=2000 + c, E=bank.E.row(c))
        idx_tx.append(d["idx_tx"])
    bank.run()
    ev0, ev1 = _lib.Event(), _lib.Event()
    barrier_sync()
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(steps):
        bank.run()
    ev1.record()
    barrier_sync()
    el = time.perf_counter() - t0
    nsym = E.shape[1] // 2
    sers = {}
    worst = 0.
    for c in range(nch):
        rows = ber.cal_ser_dev((bank.out if cfg["A"] else bank.eq).row(c), idx_tx[c], bank.rx.alphabet, 256, 8192, 2000) if cfg["A"] else []
        ser_c = [r["errors"] / max(r["compared"], 1) for r in rows]
        worst = max([worst] + ser_c)
        if c in (0, nch // 2, nch - 1):
            sers[str(c)] = ser_c
    return dict(channels=nch, trainer=trainer, value=round(nch * nsym * steps / el / 1e6, 3), unit="MSym/s", steps=steps, ms_per_step=round(el / steps * 1e3, 2),
                ms_per_step_events=round(ev1.elapsed_ms(ev0) / steps, 2), ser_of_channels=sers, worst_ser=worst,
                note="informational: %d independent captures of this workload processed together on one GPU (exact trainers: one "
                     "workgroup per channel and mode in a single launch per stage); not the headline value" % nch)


def symbol_errors(out, sig, trim=2000):
    """(errors, compared) per mode of a recovered signal; alignment on a prefix, decisions counted over the whole run."""
    from qampy_amd import synth
    from qampy_amd.core.equalisation import hip_equalisation as hk
    res = []
    npre = min(out.shape[1], 1 << 16)
    tx_idx = [hk.make_decision(np.ascontiguousarray(t), sig.coded_symbols)[2] for t in sig.symbols]
    for r in out:
        _, _, m, rot, lag = synth.count_symbol_errors(r[:npre], sig.symbols[:, :npre + 512], sig.coded_symbols, trim=min(trim, npre // 8))
        lag += min(trim, npre // 8)                                   # lag was measured on the trimmed prefix
        rx_idx = hk.make_decision(np.ascontiguousarray(r * np.complex64(np.exp(1j * rot * np.pi / 2))), sig.coded_symbols)[2]
        i0, i1 = trim, r.size - trim
        a = rx_idx[i0:i1]
        b = tx_idx[m][i0 - lag:i1 - lag]
        n = min(a.size, b.size)
        res.append((int(np.count_nonzero(a[:n] != b[:n])), int(n)))
    return res


def cpu_baseline(cfg, sig, sample_sym):
    """Oracle (reference-flag OpenMP build) on a bounded prefix of the same capture; returns timing + results."""
    from oracle import oracle
    from qampy_amd.core.equalisation import equalisation as host
    try:
        oracle.build(fast_native=True)       # -march=native for THIS host
    except Exception as e:                   # fall back to the prebuilt library
        print("cpu_baseline: native rebuild failed (%s), using the prebuilt oracle" % e, file=sys.stderr)
    E = np.ascontiguousarray(np.asarray(sig)[:, :2 * sample_sym])
    ntaps = cfg["ntaps"]
    w = host._init_taps(ntaps, E.shape[0], E.shape[0], np.complex64)
    tr = host._cal_training_symbol_len(2, ntaps, E.shape[1])
    nm = E.shape[0]
    syms = [host._reshape_symbols(sig.coded_symbols if m in host.DECISION_BASED else None, m, cfg["M"], np.complex64, nm)
            for m in cfg["methods"]]
    angles = np.linspace(-np.pi / 4, np.pi / 4, cfg["A"] or 1, endpoint=False, dtype=np.float32).reshape(1, -1)
    t0 = time.perf_counter()
    for s, m in enumerate(cfg["methods"]):
        _, w, _ = oracle.train_equaliser(E, tr, cfg["niter"][s], 2, np.float32(cfg["mu"][s]), w, None, cfg["adaptive"][s], syms[s], m, fast=True)
    t1 = time.perf_counter()
    eq = oracle.apply_filter_to_signal(E, 2, w, fast=True)
    t2 = time.perf_counter()
    N = cfg["Nbps"]
    if cfg["A"]:
        ph = np.array([oracle.select_angles(angles, oracle.bps(eq[m], angles, sig.coded_symbols, N, fast=True)) for m in range(nm)])
        ph[:, N:-N] = np.unwrap(ph[:, N:-N] * 4) / 4
        out = eq * np.exp(1j * ph)
    else:
        out = eq
    t3 = time.perf_counter()
    return dict(seconds=t3 - t0, train_s=t1 - t0, apply_s=t2 - t1, bps_s=t3 - t2, wxy=w, out=out.astype(np.complex64))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--nsym", type=int, default=None, help="override the number of symbol periods per capture")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=1 << 20, help="symbol periods of the capture the CPU baseline processes")
    ap.add_argument("--train-mode", default="exact", choices=["exact", "segmented"],
                    help="exact = the reference's sequential recurrence (default, parity tier A); segmented = opt-in "
                         "segment-parallel continuation (tier B, SER-equivalent, not tap-identical)")
    ap.add_argument("--segments", type=int, default=1024)
    ap.add_argument("--prefix", type=int, default=1 << 16, help="sequential convergence prefix (steps) of the segmented mode")
    ap.add_argument("--host-synth", action="store_true", help="generate the capture with the host (numpy) generator instead of on the GPU")
    ap.add_argument("--bank-trainer", default="iterative", choices=["auto", "iterative"],
                    help="trainer forms of the channel bank: auto = as for the single capture, iterative = block-iterative for every stage "
                         "(half the Gram table: more channels fit)")
    ap.add_argument("--bank", type=int, default=128, help="channels of the informational channel-bank run at N=1 (0 = skip): that many "
                    "independent captures of the same workload resident on the GPU and processed together")
    ap.add_argument("--tier-b", action="store_true", help="also time the opt-in segmented trainer on the same capture (informational)")
    args = ap.parse_args()

    from qampy_amd import sharding
    rank, local_rank, world = sharding.rank_info()
    cfg = dict(WORKLOADS[args.workload])
    nsym = args.nsym or cfg["nsym"]

    import torch                                     # plumbing only: barriers / reductions / device sync
    from qampy_amd import _lib
    ndev = max(_lib.device_count(), 1)
    dev = local_rank % ndev                          # a launcher may expose a single device per rank
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(dev)
        # RCCL ("nccl") on the GPUs; QAMPY_BENCH_BACKEND=gloo lets several ranks share ONE GPU (checking the multi-rank flow
        # on a single-GPU box), the reductions then go through host tensors
        backend = os.environ.get("QAMPY_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend=backend)
    red_dev = "cpu" if (world > 1 and os.environ.get("QAMPY_BENCH_BACKEND", "nccl") != "nccl") else "cuda"
    _lib.init(dev)

    def barrier_sync():
        _lib.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- independent channel per rank (seed 1000 + channel), synthesised on the GPU (or the host), then made resident
    sig = make_input(cfg, nsym, sharding.channel_seed(rank), host=args.host_synth)
    seg = dict(segments=args.segments, prefix=args.prefix) if args.train_mode == "segmented" else {}
    rx = make_receiver(cfg, sig, **seg)
    rx.load(sig)
    stage_names = ["gram"] + ["train%d:%s" % (s + 1, m) for s, m in enumerate(cfg["methods"])] + ["apply"] + (["bps_recover"] if cfg["A"] else [])

    # ---- timed region: exactly K steps, HIP events between the stages (same stream as the kernels)
    elapsed, stage_ms = timed_steps(rx, args.steps, args.warmup, barrier_sync)
    elapsed = sharding.reduce_max_time(elapsed, dist, device=red_dev)

    # ---- results of the last step: SER against the transmitted symbols
    # (on-device harness: alignment search + decisions + count in HBM, qh_ser_*_dev; nothing but 7 integers per row moves)
    ser_rows = rx.ser(sig.symbols, maxlag=256, window=8192, trim=2000)
    errs = [(d["errors"], d["compared"]) for d in ser_rows]
    res = dict(wxy=rx.wxy.to_host())
    counts_all = sharding.reduce_sum_counts([[e, n] for e, n in errs], dist, device=red_dev)

    if rank != 0:
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    value = sharding.aggregate_throughput(nsym, world, args.steps, elapsed)
    # ---- roofline of the dominant kernel
    bps_b = rx.bytes_per_symbol()
    stage_bytes = [rx.TrSyms[0] * (8 * 2 * rx.nmodes + 64 * 16)]     # gram: read the capture once, write 1 KiB per step
    for s in range(rx.nstage):
        stage_bytes.append(rx.Niter[s] * rx.TrSyms[s] * 8 * (rx.nmodes * 2 + rx.modes.size))
    stage_bytes += [rx.N * bps_b["apply"]] + ([rx.N * bps_b["bps"]] if cfg["A"] else [])
    dom = int(np.argmax(stage_ms))
    achieved = stage_bytes[dom] / (stage_ms[dom] * 1e-3) / 1e9
    # HBM bytes per launch of the dominant kernel from the committed PMC pass of the same workload (scripts/gpu_pmc.sh);
    # counters need their own rocprofv3 run, so they cannot be collected inside this process
    traffic = None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic_%s.json" % args.workload)))
        if args.train_mode == "exact" and args.nsym is None and 1 <= dom <= rx.nstage:
            mid = _lib.METHOD_ID[cfg["methods"][dom - 1]]
            kname = [k for k in pmc["kernels"] if k.startswith("qh::train_la_kernel<float, %d," % mid)
                     or k.startswith("qh::train_bi_kernel<float, %d," % mid)]
            traffic = pmc["kernels"][kname[0]]["hbm_bytes"] if kname else None
    except (OSError, KeyError, ValueError, IndexError):
        traffic = None
    roofline = dict(bound="hbm", kernel=stage_names[dom], achieved=round(achieved, 3), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 6), traffic=traffic, algorithmic_bytes=int(stage_bytes[dom]),
                    note=("exact sequential LMS recurrence: dependent-issue / barrier-latency bound, one workgroup per output mode (DESIGN.md 3.1)"
                          if args.train_mode == "exact" and 1 <= dom <= rx.nstage else "see DESIGN.md"))

    out = dict(metric="equalised MSym/s (2-pol, 2 SPS)", value=round(value, 4), unit="MSym/s", n_gpus=world, steps=args.steps,
               warmup=args.warmup, ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak",
               vs_baseline=None, dtype="f32", data="synthetic",
               config=dict(workload=cfg["label"], key=args.workload, nsym_per_channel=nsym, channels=world, ntaps=cfg["ntaps"],
                           methods=list(cfg["methods"]), niter=list(cfg["niter"]), test_angles=cfg["A"], bps_N=cfg["Nbps"],
                           complex_dtype="complex64", parallelism="1 independent channel per GPU", train_mode=args.train_mode,
                           **({"segments": args.segments, "prefix": args.prefix} if args.train_mode == "segmented" else {})),
               roofline=roofline,
               stages_ms={n: round(t, 3) for n, t in zip(stage_names, stage_ms)},
               stages_GBps={n: round(b / (t * 1e-3) / 1e9, 2) for n, b, t in zip(stage_names, stage_bytes, stage_ms)},
               # what actually bounds the exact trainers: shader cycles per recurrence step of the critical workgroup (2.4 GHz
               # clock; a lone wavefront issues one instruction per ~8.3 cycles)
               train_cycles_per_step={stage_names[1 + s2]: round(stage_ms[1 + s2] * 1e-3 * 2.4e9 / (rx.TrSyms[s2] * rx.Niter[s2]), 1)
                                      for s2 in range(rx.nstage)},
               ser=dict(per_mode_rank0=[e / max(n, 1) for e, n in errs], errors_all=int(counts_all[:, 0].sum()),
                        symbols_all=int(counts_all[:, 1].sum())),
               device=_lib.device_name())

    if world == 1 and not args.no_cpu_baseline:
        sample = min(nsym, args.cpu_sample)
        cb = cpu_baseline(cfg, sig, sample)
        # GPU on the identical sample -> SER / tap parity against the CPU path
        rx2 = make_receiver(cfg, sig[:, :2 * sample])
        rx2.load(np.asarray(sig)[:, :2 * sample])
        rx2.run()
        r2 = rx2.fetch()
        sig_s = sig.recreate_from_np_array(np.asarray(sig)[:, :2 * sample])
        sig_s._symbols = sig.symbols[:, :sample]
        e_gpu = symbol_errors(r2["out"] if cfg["A"] else r2["eq"], sig_s)
        e_cpu = symbol_errors(cb["out"], sig_s)
        out["cpu_baseline"] = dict(value=round(sample / cb["seconds"] / 1e6, 4), unit="MSym/s", cores=os.cpu_count(), kind="port",
                                   sample="first %d symbol periods of the same capture (all stages); oracle built with the "
                                          "reference's flags + OpenMP placement" % sample,
                                   stages_s=dict(train=round(cb["train_s"], 3), apply=round(cb["apply_s"], 3), bps=round(cb["bps_s"], 3)))
        out["parity_vs_cpu"] = dict(sample=sample, ser_gpu=[e / n for e, n in e_gpu], ser_cpu=[e / n for e, n in e_cpu],
                                    errors_gpu=[e for e, _ in e_gpu], errors_cpu=[e for e, _ in e_cpu],
                                    max_abs_tap_diff=float(np.max(np.abs(r2["wxy"] - cb["wxy"]))))
        out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 2)

    if world == 1 and args.train_mode == "exact" and args.bank > 1:
        try:
            out["channel_bank"] = channel_bank_run(cfg, sig, args.bank, max(1, min(args.steps, 2)), barrier_sync, args.bank_trainer)
        except Exception as e:                    # informational only: never take the headline line down with it
            out["channel_bank"] = dict(channels=args.bank, error="%s: %s" % (type(e).__name__, e))

    if world == 1 and args.train_mode == "exact" and args.tier_b:
        # informational: the opt-in segment-parallel training on the same capture (NOT the headline `value`)
        rxb = make_receiver(cfg, sig, segments=args.segments, prefix=args.prefix)
        rxb.load(sig)
        el_b, ms_b = timed_steps(rxb, args.steps, 1, barrier_sync)
        rb = rxb.fetch()
        e_b = symbol_errors(rb["out"], sig)
        out["tier_b_segmented"] = dict(value=round(nsym * args.steps / el_b / 1e6, 3), unit="MSym/s", segments=args.segments,
                                       prefix=args.prefix, stages_ms={n: round(t, 3) for n, t in zip(stage_names, ms_b)},
                                       ser=[e / max(n, 1) for e, n in e_b], errors=[e for e, _ in e_b],
                                       max_abs_tap_diff_vs_exact=float(np.max(np.abs(rb["wxy"] - res["wxy"]))),
                                       note="opt-in; same per-symbol work, different dependency structure; only meaningful when the "
                                            "taps converge within the sequential prefix (DESIGN.md tiers)")
        if "cpu_baseline" in out:
            out["tier_b_segmented"]["speedup_vs_cpu"] = round(out["tier_b_segmented"]["value"] / out["cpu_baseline"]["value"], 2)
    print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
