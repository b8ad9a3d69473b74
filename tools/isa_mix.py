"""Instruction mix of the two blend kernels' hot loops, counted in the ISA the compiler actually emitted (gfx950 assembly of
ms-gs_amd/csrc/blend.hip) — the input of bench.py's issue model (roofline.valu_issue / roofline.useful_issue) instead of
hand-counted constants.  Run by `make` after the library is linked:

    hipcc --offload-arch=gfx950 <the library's flags> -S --cuda-device-only csrc/blend.hip -o build/blend.s
    python3 tools/isa_mix.py build/blend.s build/isa_mix.json

Classes (cycles per wave64 instruction per SIMD from tools/valu_calib.hip, profiles/r2_valu_calibration.txt):
    plain   fp32 / integer VALU                                         2.2
    half    v_cmp*, v_cndmask*, every DPP instruction, v_pk_*, lane-crossing moves (v_readlane, v_permlane*)   4.25
    trans   v_exp / v_rcp / v_log / v_sqrt / v_rsq / v_sin / v_cos       8.1
    salu    every scalar instruction incl. s_waitcnt and branches         2.0
    lds     ds_*                                                          4.0
    vmem    global_* / buffer_* / flat_* (issue slot only)                2.0

forward  (blend_forward_kernel<false>): the innermost loop that holds the most v_exp_f32 is the entry walk; one trip evaluates
         as many entries as it has v_exp_f32 (four).  Reported per (wave, entry).
backward (blend_backward_tile_kernel<false,false>): the innermost loop with >= 4 v_exp_f32 is the per-entry loop of the
         one-wave-per-tile kernel.  Its body is cut at the forward conditional branches the compiler left in place:
           quadrant   a conditionally skipped segment that holds one v_exp_f32 (one per 8x8 quadrant; averaged)
           reduction  the conditionally skipped tail that holds the DPP reduce-scatter and the atomic
           entry      everything that runs for every (tile, entry) visit
         bench.py weights the three with the counting replica's visits / quadrant evaluations / reductions.
"""
import json
import re
import sys

CYCLES = {"plain": 2.2, "half": 4.25, "trans": 8.1, "salu": 2.0, "lds": 4.0, "vmem": 2.0}


def classify(ins):
    op = ins.split()[0]
    if op.startswith("v_"):
        if op == "v_nop":
            return None
        if re.match(r"v_(exp|rcp|log|sqrt|rsq|sin|cos)_", op):
            return "trans"
        if op.startswith(("v_cmp", "v_cndmask", "v_pk_", "v_permlane", "v_readlane", "v_readfirstlane", "v_writelane")) \
                or "dpp" in ins or "row_" in ins or "quad_perm" in ins:
            return "half"
        return "plain"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    return None


def is_code(t):
    return bool(t) and not t.startswith((".", ";")) and not t.endswith(":")


def count(lines, a, b):
    c = {k: 0 for k in CYCLES}
    for i in range(a, b):
        t = lines[i].strip()
        if is_code(t):
            k = classify(t)
            if k:
                c[k] += 1
    return c


def kernel_range(lines, needle):
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) and needle in l:
            for j in range(i, len(lines)):
                if lines[j].strip().startswith(".Lfunc_end"):
                    return i, j
    raise SystemExit(f"isa_mix: kernel {needle} not found")


def loops_of(lines, a, b):
    lab = {}
    for i in range(a, b):
        m = re.match(r"^(\.LBB\d+_\d+):", lines[i])
        if m:
            lab[m.group(1)] = i
    out = []
    for i in range(a, b):
        m = re.search(r"\b(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(2) in lab and lab[m.group(2)] < i:
            out.append((lab[m.group(2)], i))
    return out, lab


def n_exp(lines, a, b):
    return sum("v_exp_f32" in lines[i] for i in range(a, b + 1))


def innermost(loops):
    return [(s, e) for (s, e) in loops if not any((s2 > s and e2 <= e) or (s2 >= s and e2 < e) for s2, e2 in loops)]


def cycles(c):
    return sum(c[k] * CYCLES[k] for k in c)


def valu(c):
    return c["plain"] + c["half"] + c["trans"]


def main():
    src, dst = sys.argv[1], sys.argv[2]
    lines = open(src).read().split("\n")
    out = {"source": "gfx950 ISA of ms-gs_amd/csrc/blend.hip (tools/isa_mix.py)", "cycles_per_class": CYCLES}

    a, b = kernel_range(lines, "blend_forward_kernelILb0E")
    loops, _ = loops_of(lines, a, b)
    s, e = max(innermost(loops), key=lambda se: (n_exp(lines, *se), se[1] - se[0]))
    trip = count(lines, s, e + 1)
    entries = n_exp(lines, s, e)
    fwd = {k: v / entries for k, v in trip.items()}
    out["blend_fwd"] = {"entries_per_trip": entries, "per_trip": trip, "per_wave_entry": fwd,
                        "valu_per_wave_entry": valu(fwd), "valu_cycles_per_wave_entry": sum(fwd[k] * CYCLES[k] for k in ("plain", "half", "trans")),
                        "cycles_per_wave_entry": cycles(fwd),
                        "valu_mix": [fwd["plain"] / valu(fwd), fwd["half"] / valu(fwd), fwd["trans"] / valu(fwd)]}

    a, b = kernel_range(lines, "blend_backward_tile_kernelILb0ELb0E")
    loops, lab = loops_of(lines, a, b)
    cand = [se for se in innermost(loops) if n_exp(lines, *se) >= 4]
    s, e = min(cand, key=lambda se: se[1] - se[0])
    # forward conditional branches inside the loop body -> conditionally skipped segments [branch + 1, target)
    segs = []
    i = s
    while i <= e:
        m = re.search(r"\bs_cbranch_\w+\s+(\.LBB\d+_\d+)", lines[i])
        if m and m.group(1) in lab and i < lab[m.group(1)] <= e + 1:
            segs.append((i + 1, lab[m.group(1)]))
            i = lab[m.group(1)]
            continue
        i += 1
    # the tail behind the last backward-targeting "skip the reduction" branch runs to the loop's back edge
    quad, red = [], None
    covered = set()
    for (p, q) in segs:
        c = count(lines, p, q)
        if n_exp(lines, p, q - 1) == 1:
            quad.append(c)
            covered.update(range(p, q))
    # reduction = the segment (anywhere in the loop) with the most DPP / bpermute / atomics
    def dpp_count(p, q):
        return sum(("dpp" in lines[k] or "row_" in lines[k] or "ds_bpermute" in lines[k] or "global_atomic" in lines[k]) for k in range(p, q))
    # find the branch that skips the reduction: a conditional branch to the loop header region followed by >= 10 DPP
    best = None
    for k in range(s, e + 1):
        if re.search(r"\bs_cbranch_\w+\s+\.LBB", lines[k]) and dpp_count(k + 1, e + 1) >= 10:
            best = k
    if best is not None:
        red = count(lines, best + 1, e + 1)
        covered.update(range(best + 1, e + 1))
    else:
        # the compiler turned "no contribution: skip the reduction" into the loop's own back edge: the reduction is then
        # the tail of the enclosing loop (same header, give or take the label lines), behind this loop's last instruction
        outer = [(s2, e2) for (s2, e2) in loops if s2 <= s and s - s2 <= 8 and e2 > e and dpp_count(e + 1, e2 + 1) >= 10]
        if outer:
            e2 = max(x[1] for x in outer)
            red = count(lines, e + 1, e2 + 1)
    entry = {k: 0 for k in CYCLES}
    for k in range(s, e + 1):
        t = lines[k].strip()
        if k not in covered and is_code(t):
            c = classify(t)
            if c:
                entry[c] += 1
    nq = max(len(quad), 1)
    quad_avg = {k: sum(c[k] for c in quad) / nq for k in CYCLES}
    out["blend_bwd"] = {"quadrant_segments": len(quad), "per_quadrant_step": quad_avg, "per_reduction": red, "per_entry_visit": entry,
                        "valu_per_quadrant_step": valu(quad_avg),
                        "valu_cycles_per_quadrant_step": sum(quad_avg[k] * CYCLES[k] for k in ("plain", "half", "trans")),
                        "cycles_per_quadrant_step": cycles(quad_avg), "cycles_per_reduction": cycles(red) if red else None,
                        "cycles_per_entry_visit": cycles(entry)}
    json.dump(out, open(dst, "w"), indent=1)
    print(f"isa_mix: forward {valu(fwd):.2f} VALU / (wave, entry) [{entries} entries per trip], backward "
          f"{valu(quad_avg):.1f} VALU / quadrant step x{len(quad)}, reduction {valu(red) if red else 0} VALU -> {dst}")


if __name__ == "__main__":
    main()
