#!/usr/bin/env python3
"""Condenses the tools/ubench/ results (gpurun_out/<dir>/valu_rates.json, tcp_rates.json) into profiles/ubench_<tag>.json: the
measured ceilings bench.py's roofline uses.  Usage: python tools/summarize_ubench.py <gpurun_out subdir> <tag>"""
import json
import os
import re
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", sys.argv[1] if len(sys.argv) > 1 else "s2")
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
v = json.load(open(os.path.join(src, "valu_rates.json")))
valu = {}
for r in v["rows"]:
    if r["waves_per_simd"] == 8:
        ghz = r["memtime_ticks"] / r["ms"] / 1e6
        valu[r["inst"]] = {"ns_per_inst_per_simd": round(r["ns_per_inst_per_simd"], 4), "cycles": round(r["ns_per_inst_per_simd"] * ghz, 2),
                           "clock_ghz": round(ghz, 3)}
# tcp_rates may be cut short (the largest working sets fault on some boxes): parse row by row
txt = open(os.path.join(src, "tcp_rates.json")).read()
tcp = []
for m in re.finditer(r'\{"pattern": "([^"]+)", "working_set_kb": (\d+), "ms": ([\d.]+), "lane_requests": (\d+), '
                     r'"lane_requests_per_cu_per_ns": ([\d.]+), "bytes_per_cu_per_ns": ([\d.]+)\}', txt):
    tcp.append({"pattern": m.group(1), "working_set_kb": int(m.group(2)), "lane_requests_per_cu_per_ns": float(m.group(5)),
                "bytes_per_cu_per_ns": float(m.group(6))})


def rate(pat, kb):
    return [t["lane_requests_per_cu_per_ns"] for t in tcp if t["pattern"].startswith(pat) and t["working_set_kb"] == kb][0]


out = {
    "tag": tag, "device": v["device"], "cus": v["cus"],
    "collected_with": "tools/ubench/valu_rates.hip + tcp_rates.hip on MI355X (8 waves per SIMD / 20 waves per CU)",
    "valu_ns_per_inst_per_simd": {
        "node_step_mix": valu["node-step mix: cvt,fma,cvt,fma,max3,min3,cmp,cndmask"]["ns_per_inst_per_simd"],
        "v_fma_f32": valu["v_fma_f32"]["ns_per_inst_per_simd"], "v_mul_f32": valu["v_mul_f32"]["ns_per_inst_per_simd"],
        "v_cvt_f32_ubyte1": valu["v_cvt_f32_ubyte1"]["ns_per_inst_per_simd"],
        "v_pk_fma_f32": valu["v_pk_fma_f32"]["ns_per_inst_per_simd"]},
    "tcp_lane_requests_per_cu_per_ns": {"own64_l1_hit": rate("own64 (4 x dwordx4)", 16), "own64_l2_hit": rate("own64 (4 x dwordx4)", 2048),
                                        "own16_l2_hit": rate("own16", 2048)},
    # dense FMA throughput of the whole chip from the two rows above (wave64: 64 lanes x 2 flops per FMA, 1024 SIMDs), next to the
    # datasheet figure they have to be read against (MI355X_MICROARCH.md: 157.3 TFLOP/s vector FP32 = one wave64 FMA per 2 cycles
    # per SIMD at 2.4 GHz)
    "dense_fma_tflops": {
        "v_fma_f32": round(64 * 2 * 1024 / valu["v_fma_f32"]["ns_per_inst_per_simd"] / 1e3, 1),
        "v_pk_fma_f32": round(64 * 4 * 1024 / valu["v_pk_fma_f32"]["ns_per_inst_per_simd"] / 1e3, 1),
        "datasheet_vector_fp32": 157.3,
        "cycles_per_wave64_inst": {"v_fma_f32": valu["v_fma_f32"]["cycles"], "v_pk_fma_f32": valu["v_pk_fma_f32"]["cycles"],
                                   "datasheet_v_fma_f32": 2.0},
        "reading": "the datasheet's 2-cycle FMA is reached only by the PACKED form (two FMAs per lane in ~4.2 cycles); an unpacked "
                   "v_fma_f32 -- and every compare / select / convert / min / max the traversal consists of -- issues once per "
                   "~3.5-4.1 cycles.  bench.py therefore reports two VALU fractions: `frac` against the measured rate of the node "
                   "step's own instruction mix (what can actually bind) and `frac_vs_datasheet_issue` against 2 cycles per "
                   "instruction (the marketing peak, unreachable for unpackable code)"},
    "valu_all": valu, "tcp_all": tcp,
    "reading": "VALU: 'full' ops (fma, max, cvt, cmp, cndmask, bfe, dpp ...) issue at ~4.1 cycles per wave64 instruction, the simple "
               "ones (mul, add, and, shifts, mov) at ~2.3; an fma overlaps with a following cvt / cmp / cndmask / mul (pair ~5 "
               "cycles) but not with max3 / min3; v_pk_fma_f32 4.2 cycles for two fmas.  TCP: divergent gathers are byte-rate "
               "bound at ~55-60 B/CU/ns when they hit L1 (dwordx4 3.4-3.8 lane requests/CU/ns), ~1 line/CU/ns when they miss "
               "L1 and hit L2, ~0.25 lines/CU/ns when they miss L2 (32 MB set)"}
os.makedirs(os.path.join(R, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(R, "profiles", "ubench_%s.json" % tag), "w"), indent=1)
print(json.dumps({k: out[k] for k in ("valu_ns_per_inst_per_simd", "tcp_lane_requests_per_cu_per_ns")}))
