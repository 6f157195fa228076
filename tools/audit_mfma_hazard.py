#!/usr/bin/env python3
"""ISA audit: an MFMA result read by a NON-MFMA instruction too soon after the MFMA.

hipcc's hazard recogniser inserts the wait states the matrix pipe's write-back needs inside a basic block, but round 3 found a
kernel (moe_router_gate_kernel) where a loop ended in the MFMA and the ds_write of its accumulator sat right behind the loop
exit with none: three of four result registers reached LDS without the last k-step.  This walks the device assembly
(`hipcc -S --cuda-device-only`) and flags every non-MFMA reader (VALU, LDS, VMEM, export) of an MFMA's destination registers
that follows it within WAIT issue slots on the fall-through path, counting s_nop N as N + 1 slots and any other
instruction as one.  MFMA -> MFMA accumulator chaining is the hardware's own business and is not flagged.

usage: audit_mfma_hazard.py file.s [...]     exit status 1 when something is flagged
"""
import re
import sys

# issue slots an MFMA result must have aged before a non-MFMA instruction reads it.  hipcc's own schedules of the 16x16x32
# MFMAs (4 passes) never go below 5 by this tool's count (it counts an MFMA in between as one slot, like s_nop 0); the bug had 2.
WAIT = int(__import__("os").environ.get("MFMA_HAZARD_SLOTS", "4"))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(tok):
    out = set()
    for a, b, c in REG.findall(tok):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def audit(path):
    bad = []
    kernel = None
    pending = []   # [dst regs, age in slots, line no, text]
    for no, line in enumerate(open(path), 1):
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("//"):
            continue
        if t.endswith(":") and not t.startswith("."):
            kernel, pending = t[:-1], []
            continue
        if t.startswith(".") or t.endswith(":"):
            continue   # labels / directives: fall-through keeps the pending list (the conservative reading of a join)
        op = t.split()[0]
        if op.startswith(";;"):
            continue
        ops = t[len(op):]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            dst = regs(ops.split(",")[0])
            pending = [p for p in pending if not (p[0] & dst)]   # chained accumulation re-arms below
            pending.append([dst, 0, no, t])
            continue
        slots = int(ops.strip().split()[0]) + 1 if op == "s_nop" else 1
        if op not in ("s_nop",):
            # which operands are read?  everything but the first operand of value-producing instructions; stores read all
            parts = ops.split(",")
            reads = parts if (op.startswith(("ds_write", "global_store", "buffer_store", "flat_store", "scratch_store", "exp"))
                              or len(parts) == 1) else parts[1:]
            used = set()
            for r in reads:
                used |= regs(r)
            for p in pending:
                if p[1] < WAIT and (p[0] & used) and not op.startswith(("s_", "v_mfma")):
                    bad.append((kernel, no, p[1], p[3], t))
            written = regs(parts[0]) if parts and not op.startswith(("ds_write", "global_store", "buffer_store", "s_")) else set()
            pending = [p for p in pending if not (p[0] & written)]
        for p in pending:
            p[1] += slots
        pending = [p for p in pending if p[1] < WAIT]
        if op in ("s_endpgm", "s_branch", "s_setpc_b64"):
            pending = []
    return bad


if __name__ == "__main__":
    total = 0
    for f in sys.argv[1:]:
        for kernel, no, age, mf, rd in audit(f):
            total += 1
            print(f"{f}:{no}: [{kernel}] reads an MFMA result after {age} slot(s)\n    {mf}\n    {rd}")
    print(f"{total} suspicious read(s)")
    sys.exit(1 if total else 0)
