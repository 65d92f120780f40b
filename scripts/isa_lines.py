#!/usr/bin/env python3
"""Static instruction count per source line of one kernel: scripts/isa_lines.py <asm with .loc (hipcc -gline-tables-only -S)> <kernel symbol prefix> <source> [top]"""
import re, collections, sys
asm, sym, srcf = sys.argv[1:4]
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
lines = open(asm).read().split('\n')
start = [i for i, l in enumerate(lines) if l.startswith(sym) and ':' in l[:200] and not l.startswith('\t')][0]
end = [i for i, l in enumerate(lines) if i > start and l.startswith('.Lfunc_end')][0]
cur = None; cnt = collections.Counter(); kinds = collections.defaultdict(collections.Counter)
for l in lines[start:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = int(m.group(2)); continue
    m = re.match(r'\s+(v_|s_|ds_|global_|buffer_|flat_)(\w+)', l)
    if m and cur is not None:
        cnt[cur] += 1; kinds[cur][m.group(1)] += 1
src = open(srcf).read().split('\n')
print('total', sum(cnt.values()))
for ln, c in sorted(cnt.items(), key=lambda x: -x[1])[:top]:
    print(f"{ln:5d} {c:4d} {dict(kinds[ln])}  | {src[ln-1].strip()[:110]}")
