"""Static check of the compiled kernels (no GPU needed: hipcc cross-compiles gfx950): no workgroup barrier publishes LDS-DMA data behind a wait
that lets younger register loads or stores stay in flight (tools/scan_dma_waits.py; the hardware finding behind it: tools/microbench/vmorder.hip)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _hipcc():
    sys.path.insert(0, ROOT)
    from office_person_detection_vit_amd.csrc import build
    return build.hipcc_path()


@pytest.mark.skipif(not (os.path.exists(_hipcc()) or shutil.which(_hipcc())), reason="hipcc (as csrc/build.py resolves it) not available")
def test_no_barrier_publishes_lds_dma_data_behind_a_mixed_count():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "scan_dma_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "0 suspicious barrier(s)" in r.stdout and "none exempted" in r.stdout
    assert "[bf16]" in r.stdout and "kernels_attn.hip" in r.stdout   # both instantiations of the element-typed files, the per-file flags
