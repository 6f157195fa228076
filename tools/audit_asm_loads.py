#!/usr/bin/env python3
"""Static check of the hand-counted `s_waitcnt vmcnt(N)` code (gemv_stream_kernel.hpp).

The streaming GEMV issues its loads from inline asm and waits with literal vmcnt values; the compiler does
not know the destination registers are in flight, so nothing stops it from reading or copying one of them
between the load and the wait (it did: a loop-carried copy of the early activation registers, fixed by
peeling the first batch).  This tool disassembles the device code of the given objects and walks every
kernel in program order:

  * every VMEM load / store bumps the outstanding count (gfx9: one vmcnt for both);
  * `s_waitcnt vmcnt(N)` retires all but the newest N;
  * an instruction that READS a register of a load still in flight is reported (an ALU write to it re-defines it:
    the other arm of `valid ? load : 0`).

The walk is per basic block (llvm-objdump --symbolize-operands gives the block labels), each starting from an
empty in-flight list: exact for what a block issues itself, blind to what it inherits.  (A walk over the whole
control-flow graph was tried: the compiler encodes "real refill or dummy" in a scalar flag tested two blocks
later, so path-insensitive flow reports paths that cannot execute.)  A tripwire for the build, not a proof.

usage: audit_asm_loads.py obj/gemv_stream_inst_w8.o [...]      exit status 1 when anything is reported
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernels to skip (none today; the fused two-body launches of round 2, which the path-insensitive walk could not follow, are gone)
FUSED_KERNELS = ()
REG = re.compile(r"\b([va])(\d+)\b|\b([va])\[(\d+):(\d+)\]")
WAIT = re.compile(r"vmcnt\((\d+)\)")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1):
            out.add((m.group(1), int(m.group(2))))
        else:
            out.update((m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1))
    return out


def disassemble(obj):
    tmp = tempfile.mkdtemp(prefix="dihip_audit_")
    try:
        local = os.path.join(tmp, os.path.basename(obj))
        shutil.copy(obj, local)
        subprocess.run([OBJDUMP, "--offloading", local], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, check=True)
        code = [f for f in os.listdir(tmp) if "gfx950" in f]
        if not code:
            raise RuntimeError("no gfx950 code object in " + obj)
        return subprocess.run([OBJDUMP, "-d", "--symbolize-operands", os.path.join(tmp, code[0])], stdout=subprocess.PIPE, check=True, text=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


LOADS = ("global_load", "buffer_load", "flat_load", "scratch_load")
STORES = ("global_store", "buffer_store", "flat_store", "scratch_store")
ATOMICS = ("global_atomic", "buffer_atomic", "flat_atomic")

def step(state, mn, ops, where, findings):
    """one instruction applied to one in-flight list (tuple of (frozenset of dest regs, text), oldest first)"""
    if mn == "s_waitcnt":
        m = WAIT.search(ops)
        if m:
            keep = int(m.group(1))
            state = state[len(state) - keep:] if keep and keep < len(state) else (state if keep else ())
        return state
    is_load = mn.startswith(LOADS)
    no_dest = mn.startswith(STORES + ("ds_write", "ds_store", "s_", "v_cmp", "v_cmpx", "v_nop", "v_readfirstlane", "v_readlane"))
    parts = ops.split(",", 1)
    reads = regs_of(ops if no_dest else (parts[1] if len(parts) > 1 else ""))
    writes = frozenset() if (no_dest or is_load) else frozenset(regs_of(parts[0]))
    for dest, text in state:
        if dest & reads:
            findings.add("%s: `%s %s` reads a register of `%s` still in flight" % (where, mn, ops, text))
    if writes:
        # inside one block there is no other arm: the compiler took the destination for dead (a load whose result is
        # never read -- the tail's dummy loads) and gave the register to something else; the load lands on top of it
        for dest, text in state:
            if dest & writes:
                findings.add("%s: `%s %s` writes a register `%s` has yet to land in" % (where, mn, ops, text))
        state = tuple((d - writes, t) for d, t in state)
    if is_load:
        dest = frozenset() if "lds" in ops.split() else frozenset(regs_of(parts[0]))
        state = state + ((dest, mn + " " + ops),)
    elif mn.startswith(STORES) or mn.startswith(ATOMICS):
        returns = mn.startswith(ATOMICS) and (" sc0" in ops or " glc" in ops)
        state = state + ((frozenset(regs_of(parts[0])) if returns else frozenset(), mn + " " + ops),)
    return state


def successors(blocks, order):
    succ = {}
    for i, lab in enumerate(order):
        out, fall = [], True
        for mn, ops in blocks[lab]:
            if mn == "s_branch" or mn.startswith("s_cbranch"):
                out.append(ops.strip())
                fall = mn != "s_branch"
            elif mn in ("s_endpgm", "s_setpc_b64"):
                fall = False
        if fall and i + 1 < len(order):
            out.append(order[i + 1])
        succ[lab] = [o for o in out if o in blocks]
    return succ


def is_vmem(mn):
    return mn.startswith(LOADS) or mn.startswith(STORES) or mn.startswith(ATOMICS)


def audit_loads_forward(name, blocks, order, findings):
    """Every `global_load`, every path forward from it, until a wait that provably covers it -- `vmcnt(N)` with N <= the VMEM
    operations issued on the path since (the ring fill has two arms per slot, real chunk or dummy, with the same number of
    loads each, so the count does not depend on the path) -- or until a newer load takes the same destination:
      * nothing else may WRITE the destination: the compiler only does that when it takes the load's result for dead (a load
        into a scratch variable -- the tail's dummy loads were that) and has given the register to another value, which the
        data then lands on top of (it was the zero the accumulators are reset from);
      * the activation loads issued ahead of the ring fill (`global_load_dwordx4` without `nt`) may not be READ either.  For the
        ring loads a read ends the walk instead: the unrolled main loop has paths that cannot execute (the compiler keeps "real
        refill or dummy" in a scalar flag tested two blocks later), on which the counted waits look too weak; their
        consumption is straight-line `wait, consume, refill` code, which rule (1) checks exactly."""
    for lab in order:
        for n, (mn, ops) in enumerate(blocks[lab]):
            # the hand-issued loads all use the scalar-base form `vDst, vOff, s[base]`; a load the compiler emits with `off`
            # addressing is one it tracks (and waits for) itself
            if not mn.startswith("global_load") or "lds" in ops.split() or ", s[" not in ops:
                continue
            early = mn == "global_load_dwordx4" and " nt" not in " " + ops
            dest0 = frozenset(regs_of(ops.split(",")[0]))
            text = mn + " " + ops
            seen = set()
            work = [(lab, n + 1, 0, dest0)]
            while work:
                b, start, c, dest = work.pop()
                closed = False
                ins = blocks[b]
                targets = []  # branches BEHIND the starting point (a label-to-label block can hold one in its middle)
                for k in range(start, len(ins)):
                    m2, o2 = ins[k]
                    if m2 == "s_waitcnt":
                        w = WAIT.search(o2)
                        if w and int(w.group(1)) <= c:
                            closed = True
                            break
                        continue
                    if m2 == "s_branch" or m2.startswith("s_cbranch"):
                        targets.append(o2.strip())
                        if m2 == "s_branch":
                            break
                        continue
                    parts = o2.split(",", 1)
                    no_dest = m2.startswith(NO_DEST)
                    reads = regs_of(o2 if no_dest else (parts[1] if len(parts) > 1 else ""))
                    if dest & reads:
                        if early:
                            findings.add("%s %s+%d: `%s %s` reads a register of `%s` (issued in %s) before a wait covers it"
                                         % (name, b, k, m2, o2, text, lab))
                        else:
                            closed = True
                            break
                    if not no_dest:
                        hit = dest & frozenset(regs_of(parts[0]))
                        if hit and not m2.startswith(LOADS):
                            findings.add("%s %s+%d: `%s %s` writes a register `%s` (issued in %s) has yet to land in"
                                         % (name, b, k, m2, o2, text, lab))
                        dest = dest - hit
                        if not dest:
                            closed = True
                            break
                    if is_vmem(m2):
                        c += 1
                    if m2 in ("s_endpgm", "s_setpc_b64"):
                        closed = True
                        break
                if closed:
                    continue
                # every exit of the block leaves with the state at its END: the compiler keeps "real chunk or dummy" of a ring
                # slot in a scalar flag and branches around the arm not taken from the middle of a block -- counting only the
                # loads in front of such a branch would follow paths that cannot execute (both arms issue the same number)
                last = ins[-1][0] if ins else ""
                if last != "s_branch" and last not in ("s_endpgm", "s_setpc_b64"):
                    i = order.index(b)
                    if i + 1 < len(order):
                        targets.append(order[i + 1])
                for t in targets:
                    key = (t, c, dest)
                    if t in blocks and key not in seen and len(seen) < 20000:
                        seen.add(key)
                        work.append((t, 0, c, dest))


NO_DEST = STORES + ("ds_write", "ds_store", "s_", "v_cmp", "v_cmpx", "v_nop", "v_readfirstlane", "v_readlane")


def audit_kernel(name, blocks, order):
    """blocks: {label: [(mnemonic, operands)]}, order: labels in layout order -> set of findings.
    (1) Each basic block on its own, starting from an empty in-flight list: inside a block the vmcnt arithmetic is exact
    for the loads the block itself issued, so every finding is real; what is carried in from other blocks is not seen.
    (2) Across blocks: audit_loads_forward."""
    findings = set()
    for lab in order:
        state = ()
        for n, (mn, ops) in enumerate(blocks[lab]):
            state = step(state, mn, ops, "%s %s+%d" % (name, lab, n), findings)
    audit_loads_forward(name, blocks, order, findings)
    return findings


def audit(obj, only=None, skip=()):
    """skip: substrings of kernel names to leave out (kernels that join two audited bodies through a scalar flag)"""
    text = disassemble(obj)
    findings, kernels = [], 0
    head = re.compile(r"^[0-9a-f]+ <(.+)>:$")
    insn = re.compile(r"^\s+(\S+)\s*(.*?)\s*(//.*)?$")
    name, blocks, order, cur = None, {}, [], None

    def flush():
        nonlocal kernels
        if name and order and (only is None or only in name) and not any(x in name for x in skip):
            kernels += 1
            findings.extend(sorted(audit_kernel(name, blocks, order)))

    for ln in text.splitlines():
        h = head.match(ln)
        if h:
            lab = h.group(1)
            if re.fullmatch(r"L\d+", lab):  # basic-block label inside the current kernel
                cur = lab
                blocks[cur] = []
                order.append(cur)
            else:
                flush()
                name, blocks, order, cur = lab, {"entry": []}, ["entry"], "entry"
            continue
        m = insn.match(ln)
        if m and name and m.group(1) and not ln.lstrip().startswith("//"):
            blocks[cur].append((m.group(1), m.group(2)))
    flush()
    return kernels, findings


def main(argv):
    bad = 0
    for obj in argv:
        kernels, findings = audit(obj, skip=FUSED_KERNELS)
        print("%s: %d kernels, %d findings" % (os.path.basename(obj), kernels, len(findings)))
        for f in findings[:40]:
            print("  " + f)
        bad += len(findings)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
