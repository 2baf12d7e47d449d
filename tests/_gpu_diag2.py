import sys, time, os, ctypes
sys.path.insert(0, 'tests')
from vp8_testlib import *
P = load_package()
name = sys.argv[1]; fi_want = int(sys.argv[2])
w,h,frames = P.read_ivf(ivf_path(name))
parser = P.Parser(); ctx = P.Vp8Hip()
for fi,data in enumerate(frames[:fi_want+1]):
    hdr, changed, mbs, coef, mvs = P.parse_to_numpy(parser, data)
    if changed:
        g = P.geom(hdr.width, hdr.height); ctx.configure(hdr.width, hdr.height, 4, 1)
    r = parser.refs
    if fi == fi_want:
        ctx.fill_slot(0, hdr, mbs, coef, mvs)
        o = np.zeros(g.frame_size,np.uint8); o1 = np.zeros(g.frame_size,np.uint8)
        oracle_decode(hdr, mbs, coef, mvs, o1, (None,None,None), 1)
        oracle_decode(hdr, mbs, coef, mvs, o, (None,None,None), 3)
        ctx.decode([(0, 0, None)], 3)
        gb = ctx.download_full(0)
        def pl(b): return np.lib.stride_tricks.as_strided(b[g.y_off:], shape=(g.aligned_h,g.aligned_w), strides=(g.y_stride,1))
        G,O,O1 = pl(gb),pl(o),pl(o1)
        d = (G!=O)
        print('diff mask MB(0,0),(0,1):'); print(d[:16,:32].astype(int))
        print('recon'); print(O1[:16,:20]); print('oracle lf'); print(O[:16,:20]); print('gpu lf'); print(G[:16,:20])
        mbd = d.reshape(g.aligned_h//16,16,g.aligned_w//16,16).sum(axis=(1,3)); print(mbd[:6,:20])
        m = mbs[0]; print('mb0', list(m[:8]))
    parser.swap(hdr)
