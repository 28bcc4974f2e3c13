#!/usr/bin/env python3
"""Static check of the generated step text (npore_amd/csrc/gen_fill_asm.py), per wave role, over EVERY path through the
text (labels and branches are followed; a state is the program counter plus the memory operations still in flight per
counter).  The rules are the architected ones -- counters, not instruction distances:

  R1  no instruction reads or writes a VGPR that an LDS / global LOAD issued earlier is still going to fill (loads and
      LDS operations of one wave retire in order per counter; `s_waitcnt cnt(N)` retires all but the last N);
  R2  no instruction writes a VGPR that is a DATA register of an LDS write / global store of more than 64 bits which no
      wait on its counter has retired yet: such an instruction reads its data registers over several cycles after issue
      (ISA manual, "manually inserted wait states": VMEM stores of more than 64 bits; measured for ds_write_b128 in
      round 3, LABNOTES);
  R3  the text starts with `s_waitcnt vmcnt(0) lgkmcnt(0)` (the compiled code around it does not wait for loads into
      registers the statement declares clobbered) and reaches its end with nothing on the LGKM counter and no load and
      no wide store on the VM counter (the compiled code reuses the scratch registers at once); plain 32-bit traceback
      stores may stay in flight;
  R4  no VALU writes a LOOP-CARRIED register (an in/out operand of the statement, or LENST, the one scratch register
      whose value lives from step to step -- and MATV, the record's first register, where MAT.VAL lives) while exec is narrowed to some lanes (the hand-over's lane 0 / lane 63
      regions, the masked stores): the other lanes would keep a stale value.  The word queues (rqx / rqz / rqw), which
      are refilled under the mask of their valid lanes on purpose, are exempt.

`findings(role)` returns the violations (empty = clean); tests/test_host_logic.py fails on any.
    python scripts/check_asm_pending.py          # prints them
"""
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "npore_amd", "csrc"))
import gen_fill_asm as G   # noqa: E402


def vregs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset("v%d" % k for k in range(int(m.group(1)), int(m.group(2)) + 1))
    if re.fullmatch(r"v\d+", tok):
        return frozenset([tok])
    m = re.fullmatch(r"%\[(\w+)\]", tok)
    if m:
        return frozenset(["%" + m.group(1)])
    return frozenset()


def parse(line):
    op, _, rest = line.partition(" ")
    ops = [o.strip().split(" ")[0] for o in re.split(r",(?![^\[]*\])", rest)] if rest else []
    return op, ops


_WIDE = re.compile(r"ds_write_b(96|128)|ds_write2(st64)?_b64|global_store_dwordx[34]|buffer_store_dwordx[34]")
_LOAD = ("ds_read", "ds_bpermute", "ds_permute", "global_load", "buffer_load")
_STORE = ("ds_write", "global_store", "buffer_store")


def findings(role, lines=None):
    """list of (rule, position, instruction, registers) over all paths of one role's text"""
    lines = G.gen_role(role) if lines is None else lines
    labels, ins = {}, []
    for ln in lines:
        if ln.endswith(":"):
            labels[ln[:-1]] = len(ins)
        else:
            ins.append(ln)
    out = set()
    if not ins or ins[0] != "s_waitcnt vmcnt(0) lgkmcnt(0)":
        out.add(("R3", 0, ins[0] if ins else "", ("the text must begin with s_waitcnt vmcnt(0) lgkmcnt(0)",)))
    seen = set()
    carried = {"%" + n for n, c, _ in G.operands(role)[0] if c.startswith("+")} - {"%rqx", "%rqz", "%rqw"} | {G.LENST, G.MATV}
    # an operation in flight: (registers a load will fill, data registers of a wide write)
    stack = [(0, (), (), True)]
    while stack:
        pc, lg, vm, full = stack.pop()
        while True:
            if pc >= len(ins):          # the end of the text
                if lg:
                    out.add(("R3", pc, "<end>", ("lgkm operations in flight at the end of the text",)))
                if any(d or w for d, w in vm):
                    out.add(("R3", pc, "<end>", ("a load or a wide store in flight at the end of the text",)))
                break
            key = (pc, lg, vm, full)
            if key in seen:
                break
            seen.add(key)
            ln = ins[pc]
            op, ops = parse(ln)
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", ln)
                if m:
                    n = int(m.group(1))
                    lg = lg[len(lg) - n:] if n else ()
                m = re.search(r"vmcnt\((\d+)\)", ln)
                if m:
                    n = int(m.group(1))
                    vm = vm[len(vm) - n:] if n else ()
                pc += 1
                continue
            if op in ("s_mov_b64", "s_and_saveexec_b64", "s_or_saveexec_b64") and (ops[0] == "exec" or "saveexec" in op):
                full = op == "s_mov_b64" and ops[1] == "-1"
                pc += 1
                continue
            touched = frozenset().union(*[vregs(o) for o in ops]) if ops else frozenset()
            is_load = op.startswith(_LOAD)
            is_store = op.startswith(_STORE)
            writes = vregs(ops[0]) if ops and (op.startswith("v_") or is_load) and not op.startswith("v_cmp") else frozenset()
            filling = frozenset().union(*[d for d, _ in lg + vm]) if (lg or vm) else frozenset()
            if touched & filling:
                out.add(("R1", pc, ln, tuple(sorted(touched & filling))))
            if not full and op.startswith("v_") and not op.startswith("v_cmp") and writes & carried:
                out.add(("R4", pc, ln, tuple(sorted(writes & carried))))
            # a wide write is safe from the loads of its OWN counter that follow it (in order), not from anything else
            for queue, own in ((lg, op.startswith("ds_")), (vm, op.startswith(("global_", "buffer_")))):
                for _, wide in queue:
                    if writes & wide and not (is_load and own):
                        out.add(("R2", pc, ln, tuple(sorted(writes & wide))))
            if is_load or is_store:
                dest = vregs(ops[0]) if is_load else frozenset()
                wide = frozenset().union(*[vregs(o) for o in ops[1:]]) - vregs(ops[0]) if _WIDE.match(op) else frozenset()
                if op.startswith("ds_"):
                    lg = (lg + ((dest, wide),))[-12:]
                else:
                    vm = (vm + ((dest, wide),))[-8:]
            if op == "s_branch":
                pc = labels[ops[0]]
                continue
            if op.startswith("s_cbranch"):
                stack.append((labels[ops[0]], lg, vm, full))
            pc += 1
    return sorted(out)


def main():
    bad = 0
    for role in range(4):
        f = findings(role)
        bad += len(f)
        print("role", role, len(f), "findings")
        for rule, pc, ln, regs in f[:40]:
            print("   ", rule, pc, ln, regs)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
