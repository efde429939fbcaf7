"""Annotate the device asm of one kernel of tk_flat.hip with source lines (needs gpurun_out/asm/flat_g.s built with
-gline-tables-only):  python tools/asm_annot.py [kernel-substring] > out.txt"""
import re, sys
want = sys.argv[1] if len(sys.argv) > 1 else "_Z14tk_flat_kernel"
lines = open('/root/repo/gpurun_out/asm/flat_g.s').read().split('\n')
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
    if m: files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
start = next(i for i, l in enumerate(lines) if l.startswith(want) and l.split(':')[0].startswith(want) and ':' in l)
end = next(i for i in range(start, len(lines)) if 'uses_flat_scratch' in lines[i])
cur = ''
for l in lines[start + 1:end]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = '%s:%d' % (files.get(int(m.group(1)), '?').replace('tk_', '').replace('.h', ''), int(m.group(2)))
        continue
    s = l.strip()
    if re.match(r'^\.LBB', l):
        print(l.split(';')[0]); continue
    if not s or s.startswith(';') or s.startswith('.'): continue
    print('   %-86s %s' % (s.split(';')[0].strip()[:86], cur))
