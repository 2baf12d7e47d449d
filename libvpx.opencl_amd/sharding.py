"""Frame sharding for the multi-GPU path: key frames (or whole GOPs) are independent
units, so rank r of N decodes a contiguous block of frames on its own GPU; the only collectives are a
start/stop barrier and the gather of per-frame MD5 digests -- bytes, never pixels.  Works with any
torch.distributed backend ("nccl" == RCCL over xGMI on the MI355X node, "gloo" in the CPU tests)."""


def shard_range(nframes, world, rank):
    """Contiguous block [lo, hi) of rank `rank`: the first nframes % world ranks get one extra frame."""
    base, extra = divmod(nframes, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_digests(dist, local_hex, nframes, device=None):
    """All-gather per-frame MD5 hex digests (16 bytes each) -> list for the whole stream, in frame order.
    Shards may differ in length by one frame, so every rank pads to ceil(nframes / world)."""
    import torch
    world = dist.get_world_size()
    per = -(-nframes // world)
    t = torch.zeros((per, 16), dtype=torch.uint8, device=device)
    for i, h in enumerate(local_hex):
        t[i] = torch.tensor(list(bytes.fromhex(h)), dtype=torch.uint8)
    bufs = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(bufs, t)
    out = []
    for r in range(world):
        lo, hi = shard_range(nframes, world, r)
        rows = bufs[r].cpu().numpy()
        out += [bytes(rows[i].tolist()).hex() for i in range(hi - lo)]
    return out


def sharded_listing(dist, nframes, decode_block, device=None):
    """The multi-GPU decode of one all-key-frame stream of `nframes` frames (every rank seeks the
    same IVF by frame index, vpxdec.c:266-318): this rank decodes its contiguous block -- `decode_block(lo, hi)`
    returns the MD5 hex digests of frames lo .. hi-1 in order -- and the digests are all-gathered into the
    ordered `decode_to_md5` listing of the whole stream, identical on every rank.  bench.py calls this over RCCL
    with the HIP pixel path behind `decode_block`, tests/test_sharding_cpu.py over gloo with the oracle."""
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(nframes, world, rank)
    local = decode_block(lo, hi)
    if len(local) != hi - lo:
        raise RuntimeError(f"rank {rank}: decode_block({lo}, {hi}) returned {len(local)} digests")
    return gather_digests(dist, local, nframes, device=device)
