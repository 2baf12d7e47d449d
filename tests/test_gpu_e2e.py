"""GPU: the threaded host feeder + batched pixel path of tools/e2e.py reproduces the reference's
per-frame MD5s: frames parsed concurrently by several feeder threads into pinned IR slots, three slot sets in flight."""
import os
import sys

import pytest

from vp8_testlib import load_package

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fixture,nframes,batch,threads", [("kf_640x360", 75, 16, 6), ("kf_odd_67x45", 200, 64, 3),
                                                           ("kf_640x360", 10, 32, 2)])
def test_threaded_feeder_end_to_end(fixture, nframes, batch, threads):
    import e2e
    out = e2e.run(load_package(), 0, fixture=fixture, nframes=nframes, batch=batch, threads=threads)
    assert out["md5_mismatches"] == 0
    assert out["frames"] == nframes and out["Mpix_s"] > 0
