cd $GRAFT_REPO_ROOT
for f in kf_1920x1080 p_1920x1080 kf_640x360; do
  python3 - <<PY
import subprocess, sys, os
sys.path.insert(0, "tests")
src = open("tests/golden/$f.ivf", "rb").read()
# loop the stream 30x into one IVF (header + frames repeated; key-frame streams and K+P streams both restart cleanly at a key frame)
hdr, body = src[:32], src[32:]
open("/tmp/loop.ivf", "wb").write(hdr + body * 30)
PY
  ./libvpx.opencl_amd/bin/vpxdec --summary --noblit -o /dev/null /tmp/loop.ivf 2>&1 | tail -1 | sed "s/^/$f: /"
  ./oracle/_ref/vpxdec_ref --summary --noblit -o /dev/null /tmp/loop.ivf 2>&1 | tail -1 | sed "s/^/$f reference: /"
done
