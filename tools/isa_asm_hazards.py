"""The shortlist kernels issue LDS reads and scalar loads from inline asm and wait for them in a LATER asm statement
(`s_waitcnt lgkmcnt(0)` tied to the destination registers).  The compiler cannot see that a destination is not ready in
between: a register copy, a spill or any other use it schedules between the load and the wait reads a stale value.
This lists every instruction that reads the destination of an asm load before the next asm `s_waitcnt lgkmcnt(0)`.
usage: python tools/isa_asm_hazards.py file.s [kernel-substring]"""
import re, sys


def regs(tok):
    """'v[4:7]' -> {'v4',..}, 's5' -> {'s5'}"""
    m = re.match(r'^([vsa])\[(\d+):(\d+)\]$', tok)
    if m: return {f'{m.group(1)}{i}' for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r'^([vsa])(\d+)$', tok)
    return {tok} if m else set()


def operands(line):
    line = line.split(';')[0].strip()
    if not line or line.startswith('.') or line.endswith(':'): return None, []
    parts = line.split(None, 1)
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    return parts[0], ops


def scan(path, pat='shortlist_kernel', out=print):
    """Returns (kernels seen, asm loads seen, hazards)."""
    src = open(path).read().split('\n')
    name = None; in_asm = False; pending = {}; n_bad = 0; n_kern = 0; n_loads = 0
    for i, l in enumerate(src):
        m = re.match(r'^(_Z\w+):', l)
        if m:
            name = m.group(1) if pat in m.group(1) else None
            pending = {}
            if name:
                n_kern += 1
                out('== ' + re.sub(r'EEvNS.*', '', name.split(pat)[-1]))
            continue
        if name is None: continue
        if 'ASMSTART' in l: in_asm = True; continue
        if 'ASMEND' in l: in_asm = False; continue
        op, ops = operands(l)
        if op is None: continue
        if in_asm:
            if op.startswith('s_waitcnt') and 'lgkmcnt(0)' in l: pending = {}
            elif op.startswith(('ds_read', 's_load')) and ops:
                n_loads += 1
                for r in regs(ops[0]): pending[r] = i + 1
            continue
        if op.startswith('s_waitcnt') and 'lgkmcnt(0)' in l: pending = {}; continue   # (a compiler wait covers them too)
        # sources: every operand but the first (the destination); stores / compares / writelane read all of theirs
        srcs = ops if op.startswith(('v_cmp', 'v_writelane', 'ds_write', 'global_store', 'scratch_store', 's_cmp', 'flat_store')) else ops[1:]
        used = set()
        for o in srcs:
            for t in re.findall(r'[vsa]\[\d+:\d+\]|[vsa]\d+', o): used |= regs(t)
        hit = used & set(pending)
        if hit:
            n_bad += 1
            out(f'   line {i+1}: {l.strip()[:70]}   <- reads {sorted(hit)} loaded by asm at line {[pending[r] for r in sorted(hit)]} before any wait')
        # a write to a pending register: the compiler reuses it while the load is in flight (WAW)
        if ops:
            for r in regs(ops[0]):
                if r in pending and not op.startswith(('v_cmp', 's_cmp')):
                    out(f'   line {i+1}: {l.strip()[:70]}   <- overwrites {r} while its asm load (line {pending[r]}) is in flight')
                    n_bad += 1
                    del pending[r]
    return n_kern, n_loads, n_bad


if __name__ == '__main__':
    k, n, bad = scan(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else 'shortlist_kernel')
    print('kernels:', k, ' asm loads:', n, ' hazards:', bad)
