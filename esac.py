"""`import esac` -- drop-in module name of the reference extension (code/esac/setup.py:28-38,
esac.cpp:513-516).  The MI355X-native implementation lives in esac_amd/ (HIP kernels + C ABI);
this shim only re-exports the reference's two entry points plus the RNG/diagnostic helpers."""
from esac_amd.api import (backward, forward, forward_batch, get_rng_state, last_result, set_exact_sampling,  # noqa: F401
                          set_exact_scores, set_limits, set_seed)
