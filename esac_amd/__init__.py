"""esac_amd -- MI355X-native ESAC hypothesis/inlier hot path (HIP kernels behind a C ABI).

Layout: csrc/ (HIP kernels + C ABI, built to libesac_hip.so), api.py (host mirror of the
reference's `esac.forward` interface), distributed.py (hypothesis sharding + one RCCL
all-reduce), synthetic.py (synthetic frames: no datasets exist offline), build.py.
"""
__all__ = ["api", "build", "synthetic"]
