"""Static check of a gfx950 assembly listing: no instruction may touch the destination registers of an inline-asm LDS read
between the read and the `s_waitcnt lgkmcnt(N)` that covers it (LDS operations return in order: a wait for N leaves the N
youngest in flight).  Hand-written asm reads are invisible to the compiler's own wait insertion, and it may reuse or copy their
destination registers before the data has arrived.  Path-insensitive (linear scan; an unconditional branch clears the state).
Usage: check_async_lds.py file.s"""
import re, sys
def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()
def check(text):
    total = 0
    for f in re.split(r'\n(?=_Z\w+:)', '\n' + text)[1:]:
        name = f.split(':')[0]
        lines = [l.strip() for l in f.split('\n')]
        pending, bad, in_asm, queue = {}, [], False, []       # queue: every LDS operation in issue order (they return in order)
        for i, l in enumerate(lines):
            if l.startswith(';;#ASMSTART'): in_asm = True; continue
            if l.startswith(';;#ASMEND'): in_asm = False; continue
            if not l or l[0] in ';.' or l.endswith(':'): continue
            op = l.split()[0]
            if op == 's_waitcnt' and 'lgkmcnt(' in l:
                keep = int(re.search(r'lgkmcnt\((\d+)\)', l).group(1))
                queue = queue[len(queue) - keep:] if keep else []
                live = set()
                for q in queue: live |= q
                pending = {r: v for r, v in pending.items() if r in live}
                continue
            if op in ('s_branch', 's_endpgm'): pending, queue = {}, []; continue
            toks = [t.strip(',') for t in l.split()[1:]]
            if op.startswith('ds_'):
                dst = regs(toks[0]) if op.startswith('ds_read') and in_asm else set()   # compiler-issued reads get compiler-inserted waits
                queue.append(dst)
                for r in dst: pending[r] = i
                if dst: continue
            used = set()
            for t in toks: used |= regs(t)
            hit = used & set(pending)
            if hit: bad.append((i, l, min(pending[r] for r in hit)))
        if bad:
            print(name[:100], len(bad), "violations")
            for b in bad[:5]: print("    line %d: %s   (read issued at line %d)" % b)
        total += len(bad)
    return total
if __name__ == "__main__":
    n = check(open(sys.argv[1]).read())
    print("violations:", n)
    sys.exit(1 if n else 0)
