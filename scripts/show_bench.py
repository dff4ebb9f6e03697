#!/usr/bin/env python3
"""Compact view of a bench.py JSON line: python scripts/show_bench.py file"""
import json, sys
try:
    d = json.load(open(sys.argv[1]))                      # bench.py --detail-out (round 6: the full result, indented)
except ValueError:
    lines = [l for l in open(sys.argv[1]) if l.startswith("{")]
    d = json.loads(lines[-1])
for k in ("value", "ms_per_step", "headline_tier", "stages_ms", "speedup_vs_cpu", "note", "extra_shapes_error"):
    if d.get(k) is not None:
        print(k, d.get(k))
for k in ("one_receiver", "one_capture_at_a_time"):
    if d.get(k):
        print(k, d[k]["value"], d[k]["ms_per_step"], d[k].get("stages_ms"))
for key in ("two_in_flight", "three_in_flight"):
    if d.get(key):
        print(key, {k: v for k, v in d[key].items() if k != "note"})
tb = d.get("tier_b")
if tb and tb.get("in_flight"):
    print("in_flight", {k: v for k, v in tb["in_flight"].items() if k not in ("what", "one_receiver")})
if tb:
    print("tier_b", tb["value"], "certified", tb["certified"], tb.get("checks"))
    for st in tb["stages"]:
        print("  ", st["stage"], "S", st["S"], "P", st["P"], "pass_ms", st["pass_ms"], "acq", st["acquisition_ms"], "rms", st["est_deviation_rms"][-2:], "taps", st.get("est_deviation_taps", [])[-2:])
    print("  ", {k: tb.get(k) for k in ("out_rms_dev_vs_exact", "eq_rms_dev_vs_exact", "out_rms_dev_same_angle", "bps_angle_mismatch_fraction", "tap_rel_dev_vs_exact", "err_trace_rms_dev_vs_exact", "errors", "errors_exact")})
if d.get("tier_a"):
    print("tier_a", d["tier_a"]["value"], d["tier_a"].get("speedup_vs_cpu"))
if d.get("tier_b_loose"):
    l = d["tier_b_loose"]; print("loose", l["value"], l["passes"], l["out_rms_dev_vs_exact"], l["tap_rel_dev_vs_exact"])
if d.get("cert_24dB"):
    c = d["cert_24dB"]; print("cert24", c["errors_exact"], c["errors_tier_b"], c["within_3sigma"], c["passes"], c["out_rms_dev_vs_exact"])
for k in ("ns", "c2"):
    b = d.get(k)
    if b:
        print(k, b["tier_b"]["value"], "cert", b["tier_b"]["certified"], b["tier_b"]["checks"], [(s["S"], s["P"]) for s in b["tier_b"]["stages"]],
              b["tier_b"]["out_rms_dev_vs_exact"], b["tier_b"].get("eq_rms_dev_vs_exact"), b["tier_b"].get("out_rms_dev_same_angle"), b["tier_b"].get("bps_angle_mismatch_fraction"), b["tier_b"]["tap_rel_dev_vs_exact"], "a:", b["tier_a"]["value"], b["tier_b"]["errors"], b["tier_a"]["errors"])
r = d.get("roofline", {})
print("roofline", {k: v for k, v in r.items() if k not in ("note", "pipeline", "valu_instr_source")})
if d.get("cpu_baseline"):
    print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["one_thread"]["value"])
if d.get("channel_bank"):
    print("bank", d["channel_bank"].get("value"), d["channel_bank"].get("speedup_vs_cpu_bank"))
if d.get("adaptive_step"):
    for st in d["adaptive_step"]["stages"]:
        print("adaptive", st["stage"], st["ms_exact"], st["ms_tier_b"], st["speedup"], st["tap_rel_dev_vs_exact"], st["err_trace_rms_dev_vs_exact"], st["final_mu_rel_dev"], st["last_mode"])
