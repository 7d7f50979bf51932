#!/usr/bin/env python3
"""Build-time ISA check for the lane kernels (juqbox.jl_amd/csrc/jq_lane_kernels.h).

gfx950 needs 2 wait states between a VALU write of a VGPR and a DPP instruction that reads that VGPR
through the DPP operand (src0); the hardware does not interlock (probes/dpp_hazard_probe.hip: stale data)
and LLVM's hazard recognizer cannot see the v_fmac_f64_dpp instructions because they are inline asm.
This script walks the final assembly (hipcc -save-temps) and fails if any v_*_dpp instruction reads, as src0,
a register written by a VALU instruction fewer than 2 wait states earlier.  Labels inside the window are
treated conservatively (unknown predecessor => violation).

usage: check_dpp_hazard.py file.s [file.s ...]
"""
import re
import sys

REG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")


def regs(tok):
    m = REG.fullmatch(tok.strip())
    if not m:
        return set()
    if m.group(3) is not None:
        return {int(m.group(3))}
    return set(range(int(m.group(1)), int(m.group(2)) + 1))


def valu_writes(op, args):
    """VGPRs written by a VALU instruction (first operand), empty for non-VALU or SGPR/VCC results."""
    if not op.startswith("v_"):
        return set()
    if op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    return regs(args[0]) if args else set()


def check(path):
    bad = 0
    ndpp = 0
    window = []          # (wait_states, written_regs | None for label)
    mwindow = []         # the same with a third field: wait states an MFMA result needs before a VALU read
    func = "?"
    for ln, raw in enumerate(open(path), 1):
        line = raw.split(";")[0].strip()
        if not line or line.startswith((".", "#", "//")) and not line.endswith(":"):
            continue
        if line.endswith(":"):
            if not line.startswith(".L"):
                func = line[:-1]
                window = []
                mwindow = []
            else:
                window.append((0, None))
                mwindow.append((0, None, 0))
            continue
        parts = line.split(None, 1)
        op = parts[0]
        args = [a.strip() for a in parts[1].split(",")] if len(parts) > 1 else []
        if op.endswith("_dpp") or " row_newbcast" in line or " row_ror" in line or " row_shr" in line or " quad_perm" in line:
            if op.startswith("v_") and len(args) >= 2:
                ndpp += 1
                # args[-1] may carry the dpp modifiers after the last register: strip them
                src0 = regs(args[1].split()[0])
                ws = 0
                for w, written in reversed(window):
                    if ws >= 2:
                        break
                    if written is None:
                        print("%s:%d: %s: DPP within 2 wait states of a label (unknown predecessor): %s" % (path, ln, func, line))
                        bad += 1
                        break
                    if written & src0:
                        print("%s:%d: %s: DPP src0 written by VALU %d wait state(s) earlier: %s" % (path, ln, func, ws, line))
                        bad += 1
                        break
                    ws += w
                # second hazard the compiler cannot see through inline asm: a double-precision MFMA result read (or, for
                # the tied accumulator, overwritten) by this VALU instruction needs 6 (4x4x4) / 11 (16x16x4) wait states
                # (LLVM GCNHazardRecognizer: DMFMA4x4/16x16WriteVgprVALUReadWaitStates) -- used by mm_t4 (jq_kernels.h)
                used = set()
                for a in args:
                    used |= regs(a.split()[0])
                ws = 0
                for w, written, mfma_need in reversed(mwindow):
                    if written is None:
                        break                     # label: the MFMA products never end a basic block right behind an MFMA
                    if mfma_need and (written & used) and ws < mfma_need:
                        print("%s:%d: %s: MFMA result read by inline-asm VALU %d wait state(s) later (needs %d): %s" % (path, ln, func, ws, mfma_need, line))
                        bad += 1
                        break
                    ws += w
                    if ws >= 11:
                        break
        mneed = 6 if op.startswith("v_mfma_f64_4x4x4") else 11 if op.startswith("v_mfma_f64_16x16x4") else 0
        if line.endswith(":"):
            pass
        elif op == "s_nop":
            mwindow.append((int(args[0], 0) + 1, set(), 0))
        else:
            mwindow.append((1, valu_writes(op, args), mneed))
        if len(mwindow) > 16:
            mwindow = mwindow[-16:]
        if op == "s_nop":
            window.append((int(args[0], 0) + 1, set()))
        else:
            window.append((1, valu_writes(op, args)))
        if len(window) > 8:
            window = window[-8:]
    return bad, ndpp


def main():
    total_bad = 0
    for p in sys.argv[1:]:
        bad, ndpp = check(p)
        print("%s: %d DPP instructions checked, %d violation(s)" % (p, ndpp, bad))
        total_bad += bad
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
