"""CPU: the WebM reader of the command line tools (libvpx.opencl_amd/csrc/host/webm.c) hands out, frame for frame, what the
reference's encoder wrote -- the same encode stored as IVF is the witness (tests/golden/make_fixtures.py --webm) -- and
survives damaged files."""
import ctypes
import os

from vp8_testlib import GOLDEN, ivf_path, load_package


class Reader(ctypes.Structure):      # webm_reader, webm.h
    _fields_ = [("data", ctypes.c_void_p), ("size", ctypes.c_size_t), ("pos", ctypes.c_size_t), ("track", ctypes.c_uint),
                ("width", ctypes.c_uint), ("height", ctypes.c_uint), ("codec", ctypes.c_char * 32)]


def frames_of(H, path):
    r = Reader()
    rc = H.webm_open(ctypes.byref(r), path.encode())
    if rc:
        return rc, None, []
    d, n, out = ctypes.c_void_p(), ctypes.c_size_t(), []
    while True:
        rc = H.webm_next(ctypes.byref(r), ctypes.byref(d), ctypes.byref(n))
        if rc != 1:
            break
        out.append(ctypes.string_at(d.value, n.value))
    info = (r.track, r.width, r.height, r.codec)
    H.webm_close(ctypes.byref(r))
    return rc, info, out


def test_frames_equal_the_ivf_twin(pkg):
    H = pkg.load_host()
    rc, info, frames = frames_of(H, os.path.join(GOLDEN, "container_176x144.webm"))
    assert rc == 0 and info == (1, 176, 144, b"V_VP8")
    _, _, twin = pkg.read_ivf(os.path.join(GOLDEN, "container_176x144.ivf_twin"))
    assert frames == twin and len(frames) == 12


def test_other_files_are_refused_and_damage_is_survived(pkg, tmp_path):
    H = pkg.load_host()
    assert frames_of(H, ivf_path("kf_odd_67x45"))[0] == -2
    assert frames_of(H, str(tmp_path / "missing.webm"))[0] == -1
    data = open(os.path.join(GOLDEN, "container_176x144.webm"), "rb").read()
    whole = frames_of(H, os.path.join(GOLDEN, "container_176x144.webm"))[2]
    for cut in (0, 3, 40, 200, 4000, len(data) - 7):
        p = tmp_path / f"cut{cut}.webm"
        p.write_bytes(data[:cut])
        rc, info, frames = frames_of(H, str(p))
        assert rc in (0, -1, -2) and frames == whole[:len(frames)]
    for seed in range(40):                      # flipped bytes: no crash, no frame that reaches outside the file image
        import random
        rnd = random.Random(seed)
        b = bytearray(data)
        for _ in range(6):
            b[rnd.randrange(len(b))] = rnd.randrange(256)
        p = tmp_path / "flip.webm"
        p.write_bytes(bytes(b))
        rc, info, frames = frames_of(H, str(p))
        assert rc in (0, -1, -2) and sum(len(f) for f in frames) <= len(b)
