"""Static check of the compiled kernels (no GPU needed: hipcc cross-compiles gfx950): no workgroup barrier publishes LDS-DMA data behind a wait
that lets younger register loads or stores stay in flight (tools/scan_dma_waits.py; the hardware finding behind it: tools/microbench/vmorder.hip)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="hipcc not available")
def test_no_barrier_publishes_lds_dma_data_behind_a_mixed_count():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_dma_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 suspicious barrier(s)" in r.stdout
