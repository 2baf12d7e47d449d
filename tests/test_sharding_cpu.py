"""CPU, world_size 2 and 8, gloo: the multi-GPU path's sharding + digest gather reproduce the 1-rank
decode_to_md5 listing, also where the frames do not divide by the ranks and where ranks are left without a frame.  The decode itself is done by the oracle here (there is no GPU); on the GPU
box the same helpers run over RCCL in bench.py / tests marked gpu."""
import os
import socket
import sys

import pytest
import torch.multiprocessing as mp

from vp8_testlib import ROOT, golden_md5


def _worker(rank, world, port, name, q):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    from vp8_testlib import load_package, ivf_path, oracle_decode
    import numpy as np
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = load_package()
    from libvpx_opencl_amd import sharding
    w, h, frames = P.read_ivf(ivf_path(name))
    g = P.geom(w, h)

    def decode_block(lo, hi):                       # all key frames: each shard is self-contained
        out = []
        for data in frames[lo:hi]:
            parser = P.Parser()
            hdr, _, mbs, coef, mvs = P.parse_to_numpy(parser, data)
            buf = np.zeros(g.frame_size, np.uint8)
            oracle_decode(hdr, mbs, coef, mvs, buf, (None, None, None))
            out.append(P.frame_md5(buf, g, w, h))
            parser.close()
        return out

    dist.barrier()
    full = sharding.sharded_listing(dist, len(frames), decode_block)      # the code path bench.py --gpus N runs over RCCL
    if rank == 0:
        q.put(full)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_ranges(pkg):
    from libvpx_opencl_amd import sharding
    for n in (1, 7, 10, 64):
        for world in (1, 2, 3, 8):
            got = [sharding.shard_range(n, world, r) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            assert max(b - a for a, b in got) - min(b - a for a, b in got) <= 1


@pytest.mark.parametrize("world,name", [(2, "kf_640x360"),          # ten frames, five a rank
                                        (8, "kf_640x360"),          # ten frames over the eight ranks of a node: two ranks take two, six take one
                                        (8, "kf_odd_67x45")])       # three frames over eight ranks: five ranks have nothing to decode
def test_gloo_listing_equals_single_rank(pkg, world, name):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q)) for r in range(world)]
    for p in procs:
        p.start()
    full = q.get(timeout=180)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert full == golden_md5(name)
