"""``python bench.py --gpus N`` from a plain shell (VERDICT r1 missing #1): the parent starts N fresh rank processes before it
imports torch or touches a GPU, relays rank 0's JSON line and fails when any rank fails.  Run here with ``--cpu-rehearsal``:
gloo rendezvous, the step loop's all-gather + rank-0 NMS tail, max-over-ranks timing, the final barrier — no device compute."""

import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR", "OPD_TEST_HOOKS")}
    return env


def test_self_launch_two_ranks_prints_one_json_line():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--cpu-rehearsal"],
                       env=_clean_env(), capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["config"]["global_batch"] == 16 and out["scaling"] == "weak"
    assert out["rehearsal"].startswith("cpu") and out["roofline"] is None and out["value"] > 0
    assert out["detections_last_step"] == 0   # both ranks' (empty) records reached rank 0 and went through the NMS tail


def test_self_launch_eight_ranks_rehearsal():
    """The shape the driver's scaling run has (N = 8) as far as a CPU can take it: eight fresh rank processes, gloo rendezvous on 127.0.0.1,
    the pipelined step loop with its all-gather, max-over-ranks timing, rank 0's NMS tail over 8 x 8 frame slots, the final barrier."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "7", "--warmup", "2", "--cpu-rehearsal"],
                       env=_clean_env(), capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["steps"] == 7 and out["config"]["global_batch"] == 64 and out["config"]["parallelism"] == "frame-sharded dp8"
    assert out["exchange"] == "torch.distributed" and out["detections_last_step"] == 0 and out["sustained"] is None


def test_launcher_fails_when_a_rank_fails(tmp_path):
    sys.path.insert(0, ROOT)
    import bench

    script = tmp_path / "child.py"
    script.write_text("import os, sys, time\n"
                      "if os.environ['RANK'] == '1':\n    sys.exit(3)\n"
                      "assert os.environ['WORLD_SIZE'] == '3' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
                      "time.sleep(60)\n")
    t0 = time.time()
    rc = bench.launch_ranks(3, [], env=_clean_env(), script=str(script))
    assert rc == 3 and time.time() - t0 < 30    # the healthy ranks were stopped, not waited for


def test_mismatched_world_size_is_refused():
    env = _clean_env()
    env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "2", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--cpu-rehearsal"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
