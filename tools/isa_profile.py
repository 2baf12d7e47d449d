#!/usr/bin/env python3
"""Dev aid: static VALU instruction count per source line of a HIP kernel.
   hipcc ... -gline-tables-only -S --cuda-device-only -o k.s k.hip ; tools/isa_profile.py k.s k.hip [top]"""
import re, sys, collections
asm, src = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
files = {}
cur = None
cnt = collections.Counter(); tot = 0
for ln in open(asm):
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', ln)
    if m:
        files[int(m.group(1))] = (m.group(3) or m.group(2)); continue
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', ln)
    if m:
        cur = (int(m.group(1)), int(m.group(2))); continue
    m = re.match(r'\s+(v_\w+)', ln)
    if m and cur:
        cnt[cur] += 1; tot += 1
lines = open(src).read().split('\n')
print("total VALU (static):", tot)
for (f, l), n in cnt.most_common(top):
    name = files.get(f, '?').split('/')[-1]
    text = lines[l - 1].strip()[:100] if name == src.split('/')[-1] and l <= len(lines) else ''
    print(f"{n:6d}  {name}:{l}  {text}")
