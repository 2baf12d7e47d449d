import sys, time, os, ctypes
sys.path.insert(0, 'tests')
from vp8_testlib import *
P = load_package()
name = sys.argv[1]; maxf = int(sys.argv[2]) if len(sys.argv)>2 else 3
w,h,frames = P.read_ivf(ivf_path(name))
parser = P.Parser(); ctx = P.Vp8Hip()
obufs=None
for fi,data in enumerate(frames[:maxf]):
    hdr, changed, mbs, coef, mvs = P.parse_to_numpy(parser, data)
    if changed:
        g = P.geom(hdr.width, hdr.height); obufs=[np.zeros(g.frame_size,np.uint8) for _ in range(4)]
        ctx.configure(hdr.width, hdr.height, 4, 1)
    r = parser.refs
    refs_o = (obufs[r.lst_idx], obufs[r.gld_idx], obufs[r.alt_idx])
    # upload reference frames from oracle so errors don't accumulate
    for idx in set((r.lst_idx, r.gld_idx, r.alt_idx)):
        if idx != r.new_idx: ctx.upload_frame(idx, obufs[idx])
    ctx.fill_slot(0, hdr, mbs, coef, mvs)
    for stages,label in ((1,'recon'),(3,'recon+lf'),(7,'all')):
        o = np.zeros(g.frame_size,np.uint8)
        oracle_decode(hdr, mbs, coef, mvs, o, refs_o, stages)
        for rep in range(3):
            ctx.decode([(0, r.new_idx, (r.lst_idx, r.gld_idx, r.alt_idx))], stages)
            gbuf = ctx.download_full(r.new_idx)
            d = coded_area_equal(gbuf, o, g) if stages!=7 else bordered_area_equal(gbuf,o,g)
            print(f'frame {fi} type={hdr.frame_type} lf={hdr.filter_level} {label:9s} rep{rep}:', 'OK' if not d else d)
    obufs[r.new_idx][:] = o
    parser.swap(hdr)
st = ctx.stats(); print('stats', st.recon_ms, st.lf_ms, st.extend_ms, st.recon_waves, st.lf_waves)
