import sys, time, os
sys.path.insert(0, 'tests')
from vp8_testlib import *
P = load_package()
t0=time.time()
names = sys.argv[1:] or FIXTURES
allok = True
for name in names:
    try:
        got = P.decode_ivf_gpu(ivf_path(name))
    except Exception as e:
        print(name, 'EXC', e); allok=False; continue
    gold = golden_md5(name)
    bad = [i for i,(a,b) in enumerate(zip(got,gold)) if a!=b]
    print(f'{name:20s} shown={len(got)} gold={len(gold)} mismatches={len(bad)} first_bad={bad[:3]}')
    allok &= (got==gold)
print('ALL OK' if allok else 'FAIL', 'elapsed', time.time()-t0)
