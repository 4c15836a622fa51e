#!/usr/bin/env python3
"""Instruction-issue floor of the hot kernels.

For every kernel: the per-class VALU instruction histogram of the code a wave actually runs on the bench
workload -- taken from the device assembly (`make -C flac-codec_amd/csrc asm-<tu>`), restricted to the
representative path named below (which tap-count / fixed-order instantiation the bench data takes, every
small block once, loop bodies only for the streaming kernels) -- is weighted with the MEASURED sustained
issue cost of each instruction class (tools/ubench/issue_rate2.hip, four waves per SIMD:
profiles/r03_issue_rate_ubench.json).  The mix gives the average cost of one wave-instruction of that
kernel; multiplied by the kernel's measured dynamic instruction count (SQ_INSTS_VALU of the committed
counter pass) and divided by the chip's 1024 SIMDs it is `attainable_ms`: the time the kernel would take
if every SIMD issued back to back with nothing else in the way (no memory waits, no barriers, no launch
ramp).  bench.py prints it beside the measured launch time (`roofline.valu_issue.attainable_ms`).

usage: tools/issue_floor.py [--asm-dir flac-codec_amd/csrc] [--out profiles/r03_issue_floor.json]"""
import argparse
import json
import os
import re
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# opcode (without _e32/_e64/_dpp/_sdwa suffix) -> ubench entry
ALIAS = {
    "v_add_u32": "v_add_u32", "v_sub_u32": "v_sub_u32", "v_subrev_u32": "v_subrev_u32", "v_xor_b32": "v_xor_b32",
    "v_and_b32": "v_and_b32", "v_or_b32": "v_or_b32", "v_not_b32": "v_not_b32", "v_lshrrev_b32": "v_lshrrev_b32",
    "v_ashrrev_i32": "v_ashrrev_i32", "v_lshlrev_b32": "v_lshlrev_b32", "v_mov_b32": "v_mov_b32",
    "v_sad_u32": "v_sad_u32", "v_add3_u32": "v_add3_u32", "v_or3_b32": "v_or3_b32", "v_bitop3_b32": "v_bitop3_b32",
    "v_mad_i64_i32": "v_mad_i64_i32", "v_mad_u64_u32": "mad_i64_chain", "v_mad_i32_i24": "v_mad_i32_i24",
    "v_mad_u32_u24": "v_mad_i32_i24", "v_lshl_add_u64": "v_lshl_add_u64", "v_ashrrev_i64": "v_ashrrev_i64",
    "v_lshrrev_b64": "v_ashrrev_i64", "v_lshlrev_b64": "v_ashrrev_i64", "v_alignbit_b32": "v_alignbit_b32",
    "v_cndmask_b32": "v_cndmask_b32", "v_readlane_b32": "v_readlane_b32", "v_readfirstlane_b32": "v_readfirstlane_b32",
    "v_writelane_b32": "v_writelane_b32", "v_mul_f64": "v_mul_f64", "v_add_f64": "v_add_f64", "v_fma_f64": "v_fma_f64",
    "v_cvt_f64_i32": "v_cvt_f64_i32", "v_max_i32": "v_max_i32", "v_min_i32": "v_max_i32", "v_max_u32": "v_min_u32",
    "v_min_u32": "v_min_u32", "v_mul_lo_u32": "v_mul_lo_u32", "v_mul_hi_u32": "v_mul_lo_u32", "v_perm_b32": "v_perm_b32",
    "v_add_co_u32": "v_add_co_u32", "v_addc_co_u32": "v_addc_co_u32", "v_sub_co_u32": "v_sub_co_u32",
    "v_subb_co_u32": "v_addc_co_u32", "v_subrev_co_u32": "v_sub_co_u32", "v_mul_i32_i24": "v_mul_i32_i24",
    "v_mul_u32_u24": "v_mul_i32_i24", "v_bfe_u32": "v_bfe_u32", "v_bfe_i32": "v_bfe_u32", "v_lshl_add_u32": "v_lshl_add_u32",
    "v_and_or_b32": "v_and_or_b32", "v_lshl_or_b32": "v_and_or_b32", "v_mov_b64": "v_mov_b64", "v_pk_mov_b32": "v_mov_b64",
    "v_ldexp_f64": "v_ldexp_f64", "v_accvgpr_write_b32": "v_mov_b32", "v_accvgpr_read_b32": "v_mov_b32",
}
DEFAULT = "v_add3_u32"        # every other VALU opcode: the full-cost class


FAST = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_xor_b32", "v_and_b32", "v_or_b32", "v_not_b32", "v_lshrrev_b32",
        "v_ashrrev_i32", "v_mov_b32", "v_add_u32_lit")


def costs(path):
    """ns a SIMD spends per wave-instruction of each class at four waves per SIMD.  The simple 32-bit integer ops
    (FAST) issue at ~1.1 ns in a homogeneous stream, but NOT when they alternate with full-cost instructions: the
    measured mixes (v_add_u32 : v_sad_u32 = 1:1 -> 1.73 ns average, 2:1 -> 1.65, the order statistics' own 9:7 pattern
    -> 1.70) put a fast op inside a real instruction mix at ~1.6 ns.  The floor uses that mixed cost for them."""
    j = json.load(open(path))
    best = {}
    for r in j["results"]:
        if r["waves_per_simd"] == 4:
            best[r["op"]] = r["ns_per_inst"]
    if "mix_stats_9fast_7sad" in best:
        mixed = (16 * best["mix_stats_9fast_7sad"] - 7 * best["v_sad_u32"]) / 9
        best["fast_op_inside_a_mix"] = round(mixed, 4)
        for k in FAST:
            if k in best:
                best[k + "_homogeneous"] = best[k]
                best[k] = best["fast_op_inside_a_mix"]
    return best


def blocks_of(asm, key):
    s = open(asm).read()
    m = re.search(r"^(\S*" + re.escape(key) + r"\S*):", s, re.M)
    if not m:
        raise SystemExit(f"{key} not found in {asm}")
    a = m.start()
    b = s.index(".Lfunc_end", a)
    out, cur = [], ("entry", [], False)
    for line in s[a:b].split("\n"):
        mm = re.match(r"^(\.LBB\d+_\d+):(.*)", line)
        if mm:
            out.append(cur)
            cur = (mm.group(1), [], "Depth=2" in mm.group(2) or "Depth=3" in mm.group(2))
        else:
            t = line.split(";")[0].strip()
            if t and not t.startswith((".", "//")) and not t.endswith(":"):
                cur[1].append(t)
            elif "Loop" in line and ("Depth=2" in line or "Depth=3" in line) and not cur[1]:
                cur = (cur[0], cur[1], True)
    out.append(cur)
    return out


def opclass(ins):
    op = ins.split()[0]
    if "dpp" in op or "row_" in ins or "wave_sh" in ins or "quad_perm" in ins:
        return "v_add_u32_dpp" if not op.startswith("v_mov") else "v_mov_b32_dpp"
    base = re.sub(r"_(e32|e64|sdwa)$", "", op)
    if base == "v_add_u32" and re.search(r"\bs\d+\b|s\[\d+", ins):
        return "v_add_u32_e64_sgpr"
    if base.startswith("v_cmp"):
        return "v_cmp_vcc"
    return ALIAS.get(base, DEFAULT)


def mix(blocks):
    c = Counter()
    for _, ins, _ in blocks:
        for t in ins:
            if t.startswith("v_"):
                c[opclass(t)] += 1
    return c


def pick_cand(blocks, taps, fixed_subs):
    """representative path of a candidate wave: the order statistics, ONE exact FIXED pass (the t >> (k-1) form of
    the fixed order the data mostly takes: `fixed_subs` subtractions per sample), ONE FIR instantiation (`taps`
    64-bit multiply-adds per sample), the fold / exact pass of the LPC residual, and every small block once"""
    chosen = []
    for name, ins, inner in blocks:
        n = Counter(t.split()[0] for t in ins)
        mads = n["v_mad_i64_i32"]
        big = len(ins) >= 150
        if mads >= 64 and n["v_bitop3_b32"]:
            continue                                         # the overflow test of the checked re-run: not taken
        if mads:                                             # r05: a turn of fir64_dyn (the FIR is added below)
            continue
        if big and n["v_sad_u32"] >= 64:                    # order statistics
            chosen.append((name, ins, inner))
            continue
        subs = n["v_sub_u32_e32"] + n["v_sub_u32"]
        if big and n["v_xor_b32_e32"] >= 60 and n["v_lshrrev_b32_e32"] >= 60 and n["v_lshl_add_u64"] >= 16:
            chosen.append((name, ins, inner))                # fold of the LPC residual
            continue
        if big and n["v_lshrrev_b32_e32"] >= 60 and n["v_xor_b32_e32"] < 8 and subs < 8 and n["v_add3_u32"] >= 16:
            chosen.append((name, ins, inner))                # exact pass over the stored (folded) residual
            continue
        if big and n["v_xor_b32_e32"] >= 60 and n["v_ashrrev_i32_e32"] >= 60:   # an exact FIXED pass
            # K subtractions per sample (the first sample of a lane saves one), shift form without v_lshlrev (k >= 1)
            if abs(subs - 64 * fixed_subs) <= 2 and n["v_lshlrev_b32_e32"] < 8:
                chosen.append((name, ins, inner))
            continue
        if big and (n["v_or3_b32"] >= 16 or n["v_lshlrev_b32_e32"] >= 60):
            continue                                         # 31-bit fallback / k = 0 variants: not taken
        if not big:
            chosen.append((name, ins, inner))
    # r05 (fir64_dyn): the FIR is one region whose turns add tap pairs under scalar tests -- a wave with `taps` taps runs
    # taps mads, one 64-bit shift and one subtraction per sample
    if taps < 99:
        chosen.append(("fir64_dyn", ["v_mad_i64_i32 v[0:1], vcc, v0, s0, v[0:1]"] * (64 * taps) +
                       ["v_ashrrev_i64 v[0:1], s0, v[0:1]"] * 64 + ["v_sub_u32 v0, v0, v0"] * 64, False))
    return chosen


def pick_loops(blocks):
    """streaming kernels: the code a wave spends its time in is its innermost loop nest"""
    inner = [b for b in blocks if b[2]]
    return inner if inner else blocks


def floor(kernel, blocks, cost, dyn_insts, units=None):
    """units: work items (candidates) per launch, when the path is one item's: the large blocks then run once per item
    and the many small blocks (alternative branches of the partition tree, plan stores, ...) are weighted down to the
    rest of the measured per-item instruction count"""
    big = [b for b in blocks if len(b[1]) >= 120]
    small = [b for b in blocks if len(b[1]) < 120]
    cb, cs = mix(big), mix(small)
    nb, nsm = sum(cb.values()), sum(cs.values())
    w_small = 1.0
    if units and nsm:
        w_small = min(1.0, max(0.0, (dyn_insts / units - nb) / nsm))
    c = Counter()
    for k, v in cb.items():
        c[k] += v
    for k, v in cs.items():
        c[k] += v * w_small
    n = sum(c.values())
    ns = sum(cnt * cost.get(k, cost[DEFAULT]) for k, cnt in c.items())
    avg = ns / n
    return {"kernel_symbol": kernel, "static_valu_insts_of_path": round(n, 1), "large_blocks": nb, "small_blocks_weight": round(w_small, 3),
            "avg_ns_per_wave_inst_per_simd": round(avg, 4),
            "dynamic_valu_wave_insts": dyn_insts, "attainable_ms": round(dyn_insts * avg / 1024 * 1e-6, 4),
            "mix": {k: round(v, 1) for k, v in c.most_common(12)}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm-dir", default=os.path.join(ROOT, "flac-codec_amd", "csrc"))
    ap.add_argument("--out", default=os.path.join(ROOT, "profiles", "r05_issue_floor.json"))
    ap.add_argument("--tag", default="r05", help="round tag of the counter collections (profiles/<tag>_c_valu.json, <tag>_cfg2_valu.json ...)")
    ap.add_argument("--ubench", default=os.path.join(ROOT, "profiles", "r03_issue_rate_ubench.json"))
    a = ap.parse_args()
    cost = costs(a.ubench)
    prof = os.path.join(ROOT, "profiles")

    build_ids = set()

    def dyn(tag, kernel):
        v = json.load(open(os.path.join(prof, f"{tag}_valu.json")))
        build_ids.add(v.get("_build_id"))
        return v[kernel]["SQ_INSTS_VALU"]

    # config 3's collection: the newest one-letter tag of the round (r05_c, r05_d, ...), as bench.py picks it
    c3 = sorted(f for f in os.listdir(prof) if re.match(rf"^{a.tag}_[a-z]_valu\.json$", f))[-1][:-len("_valu.json")]
    A = lambda f: os.path.join(a.asm_dir, f)   # noqa: E731
    out = {"costs_ns_per_wave_inst_per_simd_at_4_waves": cost,
           "method": __doc__.split("usage:")[0].strip()}
    T = a.tag
    # config 3: the bench signal takes LPC order 2 (2 taps) and fixed order 2
    cd = blocks_of(A("cand_direct.gfx950.s"), "k_cand64pILi64ELi16ELb1ELb1ELb0")
    ac = blocks_of(A("autocorr.gfx950.s"), "k_autocorr4ILi13ELi4ELb1ELb1ELi0ELb1")
    fr = blocks_of(A("frame64_d.gfx950.s"), "k_frame64ILi128ELi64ELi16ELb1")
    out["config3"] = {
        "k_cand64": floor("k_cand64p<64,16,true,true>", pick_cand(cd, 2, 2), cost, dyn(c3, "k_cand64p"), 32768),
        "k_autocorr": floor("k_autocorr4<13,4,true,true,0,true>", pick_loops(ac), cost, dyn(c3, "k_autocorr4")),
        "k_pack": floor("k_frame64<128,64,16,true>", fr, cost, dyn(c3, "k_frame64")),
    }
    # config 3 on the high-order input: orders 8..12 win (10 taps as the representative instantiation)
    out["config3hi"] = {
        "k_cand64": floor("k_cand64p<64,16,true,true>", pick_cand(cd, 10, 2), cost, dyn(f"{T}_hi", "k_cand64p"), 32768),
        "k_autocorr": floor("k_autocorr4<13,4,true,true,0,true>", pick_loops(ac), cost, dyn(f"{T}_hi", "k_autocorr4")),
        "k_pack": floor("k_frame64<128,64,16,true>", fr, cost, dyn(f"{T}_hi", "k_frame64")),
    }
    # config 2: no LPC (SELF variant)
    c2 = blocks_of(A("cand_direct.gfx950.s"), "k_cand64pILi64ELi16ELb1ELb1ELb1")
    out["config2"] = {
        "k_cand64": floor("k_cand64p<64,16,true,true,SELF>", pick_cand(c2, 99, 2), cost, dyn(f"{T}_cfg2", "k_cand64p"), 32768),
        "k_pack": floor("k_frame64<128,64,16,true>", fr, cost, dyn(f"{T}_cfg2", "k_frame64")),
    }
    # config 5: order 32 (the bench signal: 2 taps; the high-order input: 28 taps as the representative instantiation)
    c5 = blocks_of(A("cand_direct.gfx950.s"), "k_cand64pILi64ELi32ELb1ELb1ELb0")
    out["config5"] = {
        "k_cand64": floor("k_cand64p<64,32,true,true>", pick_cand(c5, 2, 2), cost, dyn(f"{T}_cfg5", "k_cand64p"), 32768),
    }
    out["config5hi"] = {
        "k_cand64": floor("k_cand64p<64,32,true,true>", pick_cand(c5, 28, 2), cost, dyn(f"{T}_cfg5hi", "k_cand64p"), 32768),
    }
    out["_build_id"] = build_ids.pop() if len(build_ids) == 1 else None
    json.dump(out, open(a.out, "w"), indent=1)
    for cfg in ("config3", "config3hi", "config2", "config5", "config5hi"):
        for k, v in out[cfg].items():
            print(cfg, k, v["kernel_symbol"], "path", v["static_valu_insts_of_path"], "avg ns", v["avg_ns_per_wave_inst_per_simd"],
                  "dyn", f"{v['dynamic_valu_wave_insts'] / 1e6:.1f}M", "attainable_ms", v["attainable_ms"])


if __name__ == "__main__":
    main()
