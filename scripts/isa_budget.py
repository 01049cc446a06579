#!/usr/bin/env python3
"""Instruction budget of the brute-force triangle scans from the compiler's gfx950 assembly (no GPU needed).

    python scripts/isa_budget.py > profiles/r2_isa_budget.txt

Compiles raytracer_amd/csrc/render.hip to assembly with the library's flags, cuts out the inner loops of
k_closest_tri_brute<1,false> (the test as written) and k_closest_tri_brute<1,true> (staged) and counts instructions per
(wave, triangle) step.  Issue slots: a wave64 VALU instruction holds its SIMD for 2 cycles, v_rcp_f32 and the other
transcendentals for 8 (4 slots)."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "-std=c++17 -O3 --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize".split()


def assembly():
    out = os.path.join(tempfile.mkdtemp(), "render.s")
    subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "--cuda-device-only", "-S", "-o", out,
                    os.path.join(ROOT, "raytracer_amd", "csrc", "render.hip")], check=True, stderr=subprocess.DEVNULL)
    return open(out).read()


def kernel_body(text, mangled_part):
    m = re.search(r"^(_Z\S*%s\S*):.*?^\.Lfunc_end\d+:" % re.escape(mangled_part), text, flags=re.S | re.M)
    return m.group(0).splitlines()


def blocks(lines):
    """Basic blocks: label -> instructions (labels and directives dropped)."""
    out, cur = collections.OrderedDict(), "entry"
    out[cur] = []
    for line in lines:
        line = line.split(";")[0].rstrip()
        if not line.strip():
            continue
        if re.match(r"^\.LBB\S+:", line):
            cur = line.strip().rstrip(":")
            out[cur] = []
        elif line.startswith("\t") and not line.strip().startswith("."):
            out[cur].append(line.strip())
            if line.strip().startswith(("s_cbranch", "s_branch")):      # a fall-through block begins after a branch
                cur = "%s+%d" % (cur.split("+")[0], len(out))
                out[cur] = []
    return out


def classify(ins):
    op = ins.split()[0]
    if op.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        return "valu_transcendental"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "waitcnt"
    return "salu"


def summarise(name, instrs):
    c = collections.Counter(classify(i) for i in instrs)
    ops = collections.Counter(i.split()[0] for i in instrs if i.startswith("v_"))
    slots = c["valu"] + 4 * c["valu_transcendental"]
    print("%-58s VALU %3d (+%d transcendental) = %3d issue slots | LDS %2d  SALU %2d  waitcnt %2d" % (name, c["valu"], c["valu_transcendental"], slots, c["lds"], c["salu"], c["waitcnt"]))
    print("    " + ", ".join("%s x%d" % kv for kv in ops.most_common(12)))
    return slots


def main():
    text = assembly()
    print("Instruction budget per (wave, triangle) step of the brute-force closest-hit scan, one ray per lane (hipcc, gfx950).")
    print("54 algorithmic FLOP per pair; a SIMD issues one wave64 VALU instruction per 2 cycles, so 100 % of the FP32 peak is 27 slots per pair.\n")
    plain = blocks(kernel_body(text, "k_closest_tri_bruteILi1ELb0E"))
    loop = max(plain.items(), key=lambda kv: sum(1 for i in kv[1] if i.startswith("v_div_")))       # the block with the divisions: unrolled twice
    n_div = sum(1 for i in loop[1] if i.startswith("v_div_fixup"))
    print("== the test as written (RT_FLAG_PLAIN_TEST): inner loop block %s holds %d tests (%d IEEE divisions)" % (loop[0], n_div // 3, n_div))
    s = summarise("  whole block", loop[1])
    print("  per pair: %.1f issue slots; the three divisions: 3 x (2 v_div_scale + v_rcp + 5 fma/mul + v_div_fmas + v_div_fixup) = 3 x 13 = 39 of them\n" % (s / (n_div / 3.0)))

    staged = blocks(kernel_body(text, "k_closest_tri_bruteILi1ELb1E"))
    # the staged loops: blocks that read a record from LDS and compute d.n; the shared-origin form has no v_xor (no sign test)
    print("== staged (default): blocks of the two loops (rays with one origin / any rays), in program order")
    previous = ""
    totals = {}
    for label, ins in staged.items():
        v = [i for i in ins if i.startswith("v_")]
        if len(v) < 3:
            continue
        has_rcp = any(i.startswith("v_rcp") for i in ins)
        has_lds = any(i.startswith("ds_read") for i in ins)
        has_div = any(i.startswith("v_div_") for i in ins)
        has_cmp = any(i.startswith("v_cmp") for i in ins)
        shared = not any(i.startswith("v_xor") for i in ins)
        if has_div and has_lds:
            kind = "whole loop for waves with a ray out of range: the test as written"
        elif has_lds and has_cmp and not has_rcp and len(v) <= 16:
            kind = "front: d.n%s, parallel test" % (" ((a-o).n comes with the record: rays with one origin)" if shared else ", (a-o).n, sign test (any rays)")
            variant = "one origin" if shared else "any rays"
        elif has_rcp and not has_div and previous.startswith("front"):
            kind = "head: in-range division, hit point, alpha numerator, two comparisons"
        elif not has_rcp and not has_div and previous.startswith("head"):
            kind = "tail: beta numerator, sums, comparisons (only when a lane is still in play)"
        elif has_div and previous.startswith("tail"):
            kind = "fall-back: the test as written, for unsure pairs"
        else:
            continue
        slots = summarise("  %-12s %s" % (label, kind), ins)
        if kind.startswith(("front", "head")):
            totals[variant] = totals.get(variant, 0) + slots
        previous = kind
    print()
    for variant, slots in totals.items():
        print("common path (front + head), %s: %d issue slots per pair -> at most %.0f %% of the FP32 peak at full issue rate, %.0f %% at the measured 0.82 of the issue slots and 0.90 of the peak's clock" % (variant, slots, 2700.0 / slots, 0.82 * 0.90 * 2700.0 / slots))
    print("\nThe common path is front + head: every lane of the wave rejected after alpha.  A wave all of whose lanes are rejected by the front skips the head too;")
    print("the tail runs only for a wave with a lane still in play, the fall-back only for a wave with an unsure lane.  The PMC passes (profiles/r2*_profile.txt)")
    print("give the executed mean: SQ_INSTS_VALU per (wave, triangle) step.")


if __name__ == "__main__":
    main()
