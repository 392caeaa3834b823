# same-box comparison of builds of libesac_hip.so that differ in -D flags
# usage (GPU box): bash scripts/dev/variants.sh "<bench args>" "<name>=<flags>" ...
R=$GRAFT_REPO_ROOT
cd $R
ARGS=$1; shift
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  python esac_amd/build.py /tmp/lib_$name.so $flags > /dev/null 2>&1 &
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
