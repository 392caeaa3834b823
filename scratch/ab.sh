cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --steps 400 --warmup 40 --no-cpu-baseline --no-training --batch 64 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('value %.0f ms %.4f phases %s batched %.0f' % (d['value'], d['ms_per_step'], {k: round(v,4) for k,v in d['phase_ms'].items() if k in ('sample_p3p','score','select_rescore','refine')}, d['batched']['value']))"; }
echo "== default"; run; run
echo "== max-ilp"; export ESAC_HIP_LIB=$GRAFT_REPO_ROOT/scratch/libesac_ilp.so; run; run
