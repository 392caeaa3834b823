"""bench.py's launch contract, the part that needs no GPU: it never reports fewer ranks than it was asked for."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, **env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(env_extra)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)


def test_more_gpus_than_the_node_has_is_an_error_not_a_one_gpu_line():
    """Round 5: `python bench.py --gpus 8` with no launcher ran ONE rank and printed n_gpus 1.  Now: N ranks or a non-zero exit."""
    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    out = _bench(["--gpus", str(have + 8), "--steps", "2", "--warmup", "1"])
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "--gpus %d" % (have + 8) in out.stderr and "HIP device" in out.stderr


def test_a_launcher_with_another_world_size_is_an_error():
    out = _bench(["--gpus", "1", "--steps", "2", "--warmup", "1"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", ESAC_BENCH_ONE_DEVICE="1")
    assert out.returncode != 0 and "WORLD_SIZE" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
