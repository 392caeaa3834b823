R=$GRAFT_REPO_ROOT
cd $R
python esac_amd/build.py /tmp/libesac_prof.so -DESAC_PROFILE_CYCLES 2>&1 | tail -3
ESAC_HIP_LIB=/tmp/libesac_prof.so timeout 300 python scripts/dev/cyc.py 2>&1 | tail -32
