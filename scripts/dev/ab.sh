cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 600 --warmup 60 --no-cpu-baseline --no-training --batch 0 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value %.0f ms %.4f phases %s' % (d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items() if k in ('sample_p3p','score','select_rescore','refine')}))"; }
for rep in 1 2 3; do
echo "== default"; unset ESAC_HIP_LIB; run
echo "== variant $1"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/$1; run
done
