"""Instruction mix of the hot loops of the gfx950 kernels, by VALU issue class, from the compiler's assembly.

    python tools/isa_mix.py > profiles/r03_isa_mix.json          (compiles csrc/*.hip to assembly with build_hip.py's flags; no GPU needed)

Classes are the ones tools/micro/valu_rate2.hip measured on the MI355X (profiles/r03_valu_rate2.json): with >= 4 waves per SIMD a
wave64 instruction issues at
  full     ~1100 G wave-instructions/s chip-wide  -- v_add/sub/mul/fma/fmac/fmaak_f32, v_mov_b32, v_and/or/xor_b32, v_lshrrev_b32,
                                                      v_add/sub_u32, v_bitop3_b32; VGPR, inline-constant or literal operands only
  half     ~580 G/s                                -- the same opcodes with an SGPR source, and v_max/min/med3_f32, conversions,
                                                      v_floor/rndne/ldexp, every compare, v_cndmask, every DPP form, v_readlane /
                                                      v_readfirstlane, v_pk_*_f32, f64, v_lshlrev_b32, v_add3/lshl_add/and_or/bfe,
                                                      integer multiplies
  quarter  ~300 G/s                                -- v_exp/log/rcp/rsq/sqrt/sin/cos_f32, v_permlane32_swap
(at 1-2 waves per SIMD every class costs the same: a lone wave issues one instruction per 4-5 clocks).
For every kernel: each innermost loop (a backward branch to a label with no other loop inside) with its class counts, LDS / global /
scalar instruction counts; `hot` = the innermost loop with the most VALU instructions.  bench.py prices a launch's SQ_INSTS_VALU
with its hot loop's class shares."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "lidar-gs_amd"))
import build_hip  # noqa: E402

FULL = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_fmaak_f32", "v_fmamk_f32", "v_mov_b32", "v_and_b32", "v_or_b32",
        "v_xor_b32", "v_lshrrev_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_bitop3_b32", "v_not_b32", "v_mac_f32", "v_mad_f32"}
QUARTER = {"v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_permlane32_swap_b32", "v_permlane16_swap_b32",
           "v_rcp_iflag_f32"}
RATE = {"full": 1100.0, "half": 580.0, "quarter": 300.0}          # G wave-instructions / s, profiles/r03_valu_rate2.json (w4 columns)


def classify(mn, ops):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", mn)
    if not base.startswith("v_"):
        if base.startswith("ds_"):
            return "lds"
        if base.startswith(("global_", "buffer_", "flat_", "scratch_")):
            return "vmem"
        if base.startswith("s_waitcnt") or base in ("s_nop", "s_barrier"):
            return "wait"
        return "salu"
    if base.startswith("v_mfma") or base.startswith("v_accvgpr"):
        return "mfma"
    if base in QUARTER:
        return "quarter"
    if mn.endswith("_dpp") or "row_" in ops or "quad_perm" in ops:
        return "half"
    if base in FULL:
        # an SGPR (s12, s[4:5], vcc, exec, m0) among the SOURCE operands halves the rate
        srcs = ops.split(",")[1:]
        if any(re.match(r"^\s*-?\|?(s\d+|s\[\d+:\d+\]|vcc|exec|m0|ttmp)", s) for s in srcs):
            return "half"
        return "full"
    return "half"


def kernels_of(asm):
    """name -> list of (mnemonic, operands, label-or-None) lines."""
    out, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name, cur = m.group(1), []
            out[name] = cur
            continue
        if cur is None:
            continue
        if re.match(r"^\s*\.end_amdhsa_kernel|^\.Lfunc_end", line):
            cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur.append(("label", m.group(1), None))
            continue
        m = re.match(r"^\s+([a-z][a-z0-9_]+)\s*(.*?)(;.*)?$", line)
        if m and not m.group(1).startswith("."):
            cur.append((m.group(1), m.group(2).strip(), None))
    return out


def loops_of(lines):
    labels = {l[1]: i for i, l in enumerate(lines) if l[0] == "label"}
    loops = []
    for j, (mn, ops, _) in enumerate(lines):
        if mn.startswith(("s_cbranch", "s_branch")):
            tgt = ops.split()[-1] if ops else ""
            if tgt in labels and labels[tgt] < j:
                loops.append((labels[tgt], j))
    inner = [lp for lp in loops if not any(o != lp and lp[0] <= o[0] and o[1] <= lp[1] for o in loops)]
    return inner


def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except Exception:
        return n


def main():
    res = {"note": __doc__.split("\n\n")[0], "build_id": build_hip.build_id(), "class_rates_G_wave_inst_per_s": RATE, "rates_from": "profiles/r03_valu_rate2.json (4 waves per SIMD)", "kernels": {}}
    with tempfile.TemporaryDirectory() as tmp:
        for src, extra in build_hip.SOURCES.items():
            out = os.path.join(tmp, src + ".s")
            cmd = ["/opt/rocm/bin/hipcc"] + build_hip.COMMON + extra + ["-S", "--cuda-device-only", os.path.join(build_hip.CSRC, src), "-o", out]
            subprocess.check_call(cmd, stderr=subprocess.DEVNULL)
            for name, lines in kernels_of(open(out).read()).items():
                dn = re.sub(r"\(.*$", "", demangle(name)).replace("void ", "")
                if not dn.startswith("lg::"):
                    continue
                entry = {"source": src, "loops": []}
                for (i, j) in loops_of(lines):
                    body = [l for l in lines[i:j + 1] if l[0] != "label"]
                    cnt = {}
                    for mn, ops, _ in body:
                        c = classify(mn, ops)
                        cnt[c] = cnt.get(c, 0) + 1
                    valu = cnt.get("full", 0) + cnt.get("half", 0) + cnt.get("quarter", 0)
                    if valu == 0:
                        continue
                    ns = sum(cnt.get(c, 0) / RATE[c] for c in RATE)            # ns per G... relative: time units per wave-trip at the class roofs
                    entry["loops"].append({"lines": j - i + 1, "valu": valu, **{c: cnt.get(c, 0) for c in ("full", "half", "quarter", "lds", "vmem", "salu", "wait", "mfma")},
                                           "valu_mean_rate_G_per_s": round(valu / ns, 1)})
                whole = {}
                for mn, ops, _ in (l for l in lines if l[0] != "label"):
                    c = classify(mn, ops)
                    whole[c] = whole.get(c, 0) + 1
                wv = sum(whole.get(c, 0) for c in RATE)
                if wv:
                    entry["whole"] = {"valu": wv, **{c: whole.get(c, 0) for c in ("full", "half", "quarter", "lds", "vmem", "salu", "wait", "mfma")},
                                      "valu_mean_rate_G_per_s": round(wv / sum(whole.get(c, 0) / RATE[c] for c in RATE), 1)}
                if entry["loops"]:
                    entry["hot"] = max(entry["loops"], key=lambda l: l["valu"])
                # one-thread-per-element streams have no hot loop (their loops are short bisections / table fills): price them on the whole kernel
                # (the blends' walk loops -- the ones that read their entries from LDS -- are their hot loops whatever their share of the code)
                hot = entry.get("hot")
                entry["use"] = "hot" if hot and (hot["valu"] >= 0.3 * wv or (hot["valu"] >= 20 and hot["lds"] >= 4 and "render" in dn)) else "whole"
                if wv:
                    res["kernels"][dn] = entry
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
