"""GPU box helper: what ONE rank of a bucket-sharded suffix sort does (Engine.set_sort_shard), timed on the one GPU there is.
The other ranks' pieces of the suffix-array / BWT columns come from a full run kept on the host (instead of the exchange),
so that LCP, scan and output are those of the real run and can be compared.  usage: sort_share_probe.py <haps> <length> <world>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import mumemto_amd
from mumemto_amd import synth
from mumemto_amd.dist import DevicePointerView

haps, length, world = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
bases = np.empty(haps * length, np.uint8)
for h, b in synth.haplotypes_sparse(haps, length, 0.001, 11):
    bases[h * length:(h + 1) * length] = b
lens = np.full(haps, length, np.uint64)
eng = mumemto_amd.Engine(0)
eng.set_producer("guided", 14, 30)
for rep in range(2):
    t = time.perf_counter()
    eng.run_partitioned(None, flat=(bases, lens))
    full_s = time.perf_counter() - t
full_ms = eng.stage_ms()
want = eng.output_text()
assert not eng.is_wide()
sa, bwt = eng.sa().copy(), eng.bwt().copy()
print("one GPU, whole sort: %.3f s, stage ms %s" % (full_s, [round(x, 1) for x in full_ms]), flush=True)
spent = [0.0]


def others():
    t0 = time.perf_counter()
    lo, hi, bw = eng.columns_device()
    for r, (first, count) in enumerate(eng.sort_pieces()):
        if r == 0 or not count:
            continue
        torch.as_tensor(DevicePointerView(lo + 4 * first, (count,), "<i4"), device="cuda:0").copy_(
            torch.from_numpy(sa[first:first + count].view(np.int32)))
        torch.as_tensor(DevicePointerView(bw + first, (count,), "|u1"), device="cuda:0").copy_(torch.from_numpy(bwt[first:first + count]))
    torch.cuda.synchronize()
    spent[0] = time.perf_counter() - t0


eng.set_sort_shard(0, world, after_sort=others)
for rep in range(2):
    t = time.perf_counter()
    eng.run_partitioned(None, flat=(bases, lens))
    share_s = time.perf_counter() - t
ms = eng.stage_ms()
print("rank 0 of %d: pieces %s" % (world, eng.sort_pieces()[:3]), flush=True)
print("rank 0 of %d: %.3f s of which %.3f s stand in for the exchange (H2D of the other pieces); stage ms %s (sort stage without "
      "the stand-in: %.1f)" % (world, share_s, spent[0], [round(x, 1) for x in ms], ms[1] - spent[0] * 1e3), flush=True)
assert eng.output_text() == want
print("output identical to the whole run: True")
