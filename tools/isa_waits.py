"""Which s_waitcnt vmcnt(0) / scratch accesses sit inside the tile loop (loop depth >= 3) of the shortlist kernels?
A vmcnt(0) there drains the tile DMA queue.  usage: python tools/isa_waits.py file.s [kernel-substring]"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else 'shortlist_kernel'
name = None; depth = 0; label = ''
for i, l in enumerate(src):
    m = re.match(r'^(_Z\w+):', l)
    if m:
        name = m.group(1) if pat in m.group(1) else None
        depth = 0
        if name: print('==', re.sub(r'EEvNS.*', '', name.split('shortlist_kernel')[1]))
        continue
    if name is None: continue
    m = re.match(r'^(\.LBB\w+|; %bb\.\d+):\s*(;.*)?$', l)
    if m:
        label = m.group(1); depth = 0
        # depth annotations follow on this and the next comment lines
        j = i
        while True:
            d = re.search(r'Depth=(\d+)', src[j])
            if d: depth = max(depth, int(d.group(1)))
            j += 1
            if not src[j].strip().startswith(';'): break
        continue
    if depth >= int(sys.argv[3] if len(sys.argv) > 3 else 3) and re.search(r'vmcnt\(0\)|scratch_', l):
        print(f'   line {i+1} depth {depth} {label}: {l.strip()[:70]}')
