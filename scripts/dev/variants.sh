# same-box comparison of builds of libesac_hip.so that differ in -D flags
# usage (GPU box): bash scripts/dev/variants.sh "<bench args>" "<name>=<flags>" ...
R=$GRAFT_REPO_ROOT
cd $R
ARGS=$1; shift
SRC="esac_amd/csrc/esac_kernels.hip esac_amd/csrc/esac_score_tiled.hip esac_amd/csrc/esac_refine.hip esac_amd/csrc/esac_backward.hip esac_amd/csrc/esac_capi.hip"
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $flags $SRC -o /tmp/lib_$name.so 2>/dev/null &
done
wait
for rep in 1 2; do
for v in "$@"; do
  name=${v%%=*}
  ESAC_HIP_LIB=/tmp/lib_$name.so timeout 600 python bench.py $ARGS --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('%-12s value %.0f ms/step %.4f' % ('$name', d['value'], d['ms_per_step']), {k: round(v,4) for k,v in d.get('phase_ms',{}).items() if k in ('sample_p3p','score','select_rescore','refine')}, [(k['stage'], round(k['avg_us'],1)) for k in d.get('kernels',[])])"
done
done
