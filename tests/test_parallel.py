"""N > 1 path on CPU: world_size 2, gloo, 127.0.0.1 (the kernels themselves are covered by -m gpu)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_histogram_merge_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", "29731", os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "DIST_OK" in r.stdout


def test_bucket_ownership_is_contiguous_and_balanced():
    import numpy as np
    from bionumpy_amd import parallel
    for world in (1, 2, 3, 4, 8):
        owner = parallel.rank_of_bucket(world)
        assert owner[0] == 0 and owner[-1] == world - 1 and np.all(np.diff(owner) >= 0)
        sizes = np.bincount(owner, minlength=world)
        assert sizes.max() - sizes.min() <= 1
