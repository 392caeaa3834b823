R=$GRAFT_REPO_ROOT
cd $R
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -DESAC_PROFILE_CYCLES esac_amd/csrc/esac_kernels.hip esac_amd/csrc/esac_score_tiled.hip esac_amd/csrc/esac_refine.hip esac_amd/csrc/esac_backward.hip esac_amd/csrc/esac_capi.hip -o /tmp/libesac_prof.so 2>&1 | tail -3
ESAC_HIP_LIB=/tmp/libesac_prof.so timeout 300 python scripts/dev/cyc.py 2>&1 | tail -20
