"""Head-room: a rank's share of BASELINE configs[3] -- {anchor + 12} x 3.05 Gbp, 79.3 G text
characters, 228 GB of device heap at its peak -- on a device with 88 GB declared off limits (mmt_pool_set_reserve, the run-time
form of MUMEMTO_HEAP_RESERVE): the estimate refuses it as one suffix array, the engine runs it as anchor partitions inside the
rank + its own fold + re-sort (a sequence of partitions that runs out of memory is repeated with smaller ones: partitioned.cpp),
and the rows must be the whole run's, in the same order, up to the stream-end quirk.  The two runs happen in a process of their
own (tests/big_reserve.py) so that nothing an earlier test left on the device shapes the heap; this file sorts behind the
others for the same reason."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_a_share_of_configs3_on_a_device_with_200_gb():
    import torch
    import mumemto_amd
    mumemto_amd.load_library().mmt_pool_trim()           # what this process still holds mapped goes back first
    free, total = torch.cuda.mem_get_info(0)
    host_gb = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] / 2**20
    if free < 270 * 2**30 or host_gb < 120:
        pytest.skip("the device (%.0f GB free) or the host (%.0f GB) is not this test's alone" % (free / 2**30, host_gb))
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "big_reserve.py")], capture_output=True, text=True, timeout=900)
    tail = "\n".join(l for l in r.stdout.splitlines() if l.startswith("{") or l == "OK")
    print(tail)
    assert r.returncode == 0 and r.stdout.rstrip().endswith("OK"), r.stdout[-3000:] + r.stderr[-3000:]
    assert '"partitions_inside_the_rank": 1,' in r.stdout and '"rows_not_in_the_whole_run": 0' in r.stdout and '"same_order": true' in r.stdout
