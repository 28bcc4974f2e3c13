"""Static check of the generated step text: no instruction may read or write a VGPR that an LDS / global LOAD issued
earlier is still going to fill (loads return in order per counter; s_waitcnt retires all but the last N)."""
import os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "npore_amd", "csrc"))
import gen_fill_asm as G

def vregs(tok):
    tok = tok.strip()
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m: return {"v%d" % k for k in range(int(m.group(1)), int(m.group(2)) + 1)}
    if re.fullmatch(r"v\d+", tok): return {tok}
    m = re.fullmatch(r"%\[(\w+)\]", tok)
    if m: return {"%" + m.group(1)}
    return set()

def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":"): return None
    op, _, rest = line.partition(" ")
    ops = [o.strip() for o in re.split(r",(?![^\[]*\])", rest)] if rest else []
    # drop modifiers on last operand
    ops = [o.split(" ")[0] if not o.startswith("v[") else o.split(" ")[0] for o in ops]
    return op, ops

def analyse(role):
    lines = G.gen_role(role)
    labels = {}
    ins = []
    for ln in lines:
        if ln.endswith(":"):
            labels[ln[:-1]] = len(ins)
        else:
            ins.append(ln)
    problems = set()
    seen = set()
    stack = [(0, (), ())]     # pc, pending lgkm list of dest sets (in order), pending vm list
    while stack:
        pc, lg, vm = stack.pop()
        while pc < len(ins):
            key = (pc, lg, vm)
            if key in seen: break
            seen.add(key)
            ln = ins[pc]
            p = parse(ln)
            op, ops = p
            # registers touched
            touched = set()
            for o in ops: touched |= vregs(o)
            if op == "s_waitcnt":
                m = re.search(r"lgkmcnt\((\d+)\)", ln)
                if m:
                    n = int(m.group(1)); lg = lg[len(lg) - n:] if n else ()
                m = re.search(r"vmcnt\((\d+)\)", ln)
                if m:
                    n = int(m.group(1)); vm = vm[len(vm) - n:] if n else ()
                pc += 1; continue
            pend = set().union(*[set(d) for d in lg], *[set(d) for d in vm]) if (lg or vm) else set()
            hit = touched & pend
            if hit:
                problems.add((pc, ln, tuple(sorted(hit))))
            if op.startswith("ds_read") or op.startswith("ds_bpermute"):
                lg = lg + (tuple(sorted(vregs(ops[0]))),)
            elif op.startswith("ds_write"):
                lg = lg + ((),)
            elif op.startswith("global_load"):
                vm = vm + (tuple(sorted(vregs(ops[0]))),)
            elif op.startswith("global_store"):
                vm = vm + ((),)
            # cap list lengths (old ones are surely done? no: keep but bound for memo) 
            if len(vm) > 8: vm = vm[-8:]
            if len(lg) > 12: lg = lg[-12:]
            if op == "s_branch":
                tgt = ops[0]; pc = labels[tgt]; continue
            if op.startswith("s_cbranch"):
                tgt = ops[0]; stack.append((labels[tgt], lg, vm))
            pc += 1
    return sorted(problems)

def sources_overwritten_soon(role, window=8):
    """a VALU write to a register that a DS / VMEM instruction issued within the last `window` instructions READS
    (address or data): such an instruction reads its registers over a few cycles after issue (measured: a
    ds_write_b128 followed directly by a v_mov to its second data register stored the new value in some launches)"""
    lines = [l for l in G.gen_role(role)]
    out = []
    recent = []      # (index, text, set of source regs)
    for i, ln in enumerate(lines):
        if ln.endswith(":"):
            continue
        p = parse(ln)
        if not p:
            continue
        op, ops = p
        if op.startswith("v_") and ops:
            dst = vregs(ops[0])
            for (j, t, src) in recent:
                if i - j <= window and dst & src:
                    out.append((j, t, i, ln, tuple(sorted(dst & src))))
        if op.startswith(("ds_", "global_")):
            is_load = op.startswith(("ds_read", "ds_bpermute", "global_load"))
            src = set()
            for o in (ops[1:] if is_load else ops):
                src |= vregs(o)
            if is_load:
                src -= vregs(ops[0])      # (address == destination is the LDS unit's own business)
            recent.append((i, ln, src))
        recent = [r for r in recent if i - r[0] <= window]
    return out


for role in range(4):
    so = sources_overwritten_soon(role)
    print("role", role, len(so), "DS/VMEM sources overwritten within 8 instructions")
    for j, t, i, ln, regs in so[:20]:
        print("    %d: %s   <-  %d: %s   %s" % (j, t, i, ln, regs))
for role in range(4):
    pr = analyse(role)
    print("role", role, len(pr), "findings")
    for pc, ln, hit in pr[:40]:
        print("   ", pc, ln, hit)
