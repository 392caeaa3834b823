"""The multi-GPU score exchange from a plain C caller (no torch in the process).  Kept in a file of its own that sorts last: it maps
ROCm's 570 MB librccl.so.1 into a fresh process, the one test of the suite whose duration depends on the box's image cache."""

import pytest

from esac_amd import build


@pytest.mark.gpu
def test_c_program_runs_the_score_exchange_on_rccl_without_torch():
    """The multi-GPU exchange from the plain C caller: the library binds RCCL itself at the first esac_hip_comm_* call (ROCm's
    librccl.so.1, no torch in the process), a one-rank communicator, the all-reduce in place on the NULL stream; -13 before
    esac_hip_comm_init and after esac_hip_comm_destroy."""
    import subprocess
    from tests.native import build as nb
    exe = nb.build_abi_check()
    out = subprocess.run([exe, build.LIB_PATH, "gpu", "comm"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "comm ok" in out.stdout
