#!/usr/bin/env python3
"""ISA evidence for the frame kernels: per-kernel register / scratch / LDS / occupancy figures from hipcc's
-Rpass-analysis=kernel-resource-usage, static mnemonic histograms, and the instruction listing + histogram of the loop
that holds the node step (the innermost loop with the four global_load_dwordx4 of lv_node_step).

Usage: python tools/isa_report.py [out.txt]   (cross-compiles with the library's flags; no GPU needed)
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linevis_amd import build as lv_build  # noqa: E402

HOT = ["k_ao_rays", "k_ao_primary", "k_render_rt", "k_ppll_gather", "k_ppll_resolve", "k_render_rt_mlat", "k_ao_reduce",
       "k_ppll_raster_prism", "k_ppll_shade_prism", "k_ppll_cull_segments", "k_ppll_select_nearest"]


def demangle(names):
    out = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.split("\n")
    return [o.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0] for o in out[:len(names)]]


def compile_s(src, tmp):
    s_path = os.path.join(tmp, os.path.basename(src) + ".s")
    flags = [f for f in lv_build.FLAGS if f not in ("-fPIC",)]
    r = subprocess.run([lv_build.hipcc()] + flags + ["-S", "--cuda-device-only", src, "-o", s_path,
                                                     "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    if r.returncode != 0:
        raise SystemExit(r.stderr)
    return open(s_path).read(), r.stderr


def resources(rpass):
    res, cur = {}, None
    for line in rpass.split("\n"):
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|"
                      r"LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = m.group(2)
            res[cur] = {}
        elif cur:
            res[cur][m.group(1)] = m.group(2)
    return res


def functions(asm):
    """mangled name -> list of (label or None, instruction text) in program order"""
    fns, cur, name = {}, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            fns[name] = cur
            continue
        if cur is None:
            continue
        if line.startswith("\t.end_amdhsa_kernel") or re.match(r"^\.Lfunc_end", line):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur.append((m.group(1), None))
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        cur.append((None, t.split(";")[0].strip()))
    return fns


def mnemonic(t):
    return t.split()[0]


def classify(hist):
    groups = collections.OrderedDict((k, 0) for k in (
        "v_pk_*", "v_cvt_f32_ubyte*", "v_fma_f32 / v_fmac_f32", "v_mul_f32 / v_add_f32 / v_sub_f32", "v_max/min(3)_f32",
        "v_cmp* / v_cndmask", "v_div_* / v_rcp / v_sqrt / v_rsq", "other VALU", "global_load_dwordx4", "other global/flat/scratch",
        "ds_*", "s_* (SALU, waitcnt, branch)"))
    for m, n in hist.items():
        if m.startswith("v_pk_"): groups["v_pk_*"] += n
        elif m.startswith("v_cvt_f32_ubyte"): groups["v_cvt_f32_ubyte*"] += n
        elif m.startswith(("v_fma_f32", "v_fmac_f32")): groups["v_fma_f32 / v_fmac_f32"] += n
        elif m.startswith(("v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32")): groups["v_mul_f32 / v_add_f32 / v_sub_f32"] += n
        elif re.match(r"v_(max|min)3?_f32", m): groups["v_max/min(3)_f32"] += n
        elif m.startswith(("v_cmp", "v_cndmask")): groups["v_cmp* / v_cndmask"] += n
        elif m.startswith(("v_div_", "v_rcp", "v_sqrt", "v_rsq")): groups["v_div_* / v_rcp / v_sqrt / v_rsq"] += n
        elif m.startswith("v_"): groups["other VALU"] += n
        elif m.startswith("global_load_dwordx4"): groups["global_load_dwordx4"] += n
        elif m.startswith(("global_", "flat_", "scratch_", "buffer_")): groups["other global/flat/scratch"] += n
        elif m.startswith("ds_"): groups["ds_*"] += n
        elif m.startswith("s_"): groups["s_* (SALU, waitcnt, branch)"] += n
        else: groups["other VALU"] += n
    return groups


def node_step_loop(body):
    """Smallest backward-branch range [label .. branch] that contains >= 4 global_load_dwordx4."""
    pos = {lab: i for i, (lab, _) in enumerate(body) if lab}
    best = None
    for i, (lab, t) in enumerate(body):
        if t and t.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in pos and pos[tgt] < i:
                seg = body[pos[tgt]:i + 1]
                loads = sum(1 for _, x in seg if x and x.startswith("global_load_dwordx4"))
                if loads >= 4 and (best is None or len(seg) < len(best)):
                    best = seg
    return best


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "isa_r06.txt")
    lines = ["# ISA report of the frame kernels (hipcc %s, gfx950); generated by tools/isa_report.py" % " ".join(lv_build.FLAGS), ""]
    with tempfile.TemporaryDirectory() as tmp:
        for src in ("lv_render.hip", "lv_mlat.hip"):
            asm, rpass = compile_s(os.path.join(lv_build.CSRC, src), tmp)
            res = resources(rpass)
            fns = functions(asm)
            names = list(res)
            pretty = dict(zip(names, demangle(names)))
            lines.append("## %s -- resources per kernel (-Rpass-analysis=kernel-resource-usage)" % src)
            lines.append("%-52s %5s %5s %8s %8s %10s" % ("kernel", "VGPR", "SGPR", "scratch", "LDS B", "waves/SIMD"))
            for n in sorted(names, key=lambda x: pretty[x]):
                r = res[n]
                lines.append("%-52s %5s %5s %8s %8s %10s" % (pretty[n][:52], r.get("VGPRs"), r.get("TotalSGPRs"),
                                                             r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"),
                                                             r.get("Occupancy [waves/SIMD]")))
            lines.append("")
            for n in sorted(names, key=lambda x: pretty[x]):
                p = pretty[n]
                if not p.startswith(tuple(HOT)) or p.startswith(("k_ao_rays<true", "k_ao_primary<true", "k_render_rt<true",
                                                                  "k_ppll_gather<true", "k_render_rt_mlat<true", "k_ppll_raster_prism<true",
                                                                  "k_ppll_shade_prism<true")):
                    continue
                body = fns.get(n, [])
                hist = collections.Counter(mnemonic(t) for _, t in body if t)
                g = classify(hist)
                lines.append("### %s: %d instructions (static)" % (p, sum(hist.values())))
                lines.append("    " + ", ".join("%s %d" % kv for kv in g.items() if kv[1]))
                loop = node_step_loop(body)
                if loop and p.startswith(("k_ao_rays<false, false, 0, false, false>", "k_ao_rays<false, false, 1, false, false>")):
                    lh = collections.Counter(mnemonic(t) for _, t in loop if t)
                    lg = classify(lh)
                    lines.append("    descend loop (innermost loop holding the node fetch): %d instructions per iteration" % sum(lh.values()))
                    lines.append("      " + ", ".join("%s %d" % kv for kv in lg.items() if kv[1]))
                    lines.append("      mnemonics: " + ", ".join("%s %d" % kv for kv in sorted(lh.items(), key=lambda kv: -kv[1])))
                    lines.append("      listing:")
                    for lab, t in loop:
                        lines.append("        " + (lab + ":" if lab else "    " + t))
                lines.append("")
    open(out_path, "w").write("\n".join(lines) + "\n")
    print("wrote", out_path, len(lines), "lines")


if __name__ == "__main__":
    main()
