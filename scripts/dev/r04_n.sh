#!/bin/bash
# cfg5b: pending list of the screened chain in hypothesis order (0) against expert-major, dealt to the XCDs (1):
# k_sample_prescreen duration, FETCH_SIZE, L2 hit rate (each its own rocprofv3 run), then the cfg5b parity test with the switch on
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
O=$R/gpurun_out/r04n
mkdir -p $O
CMD="python $R/bench.py --config cfg5b --steps 4 --warmup 1 --no-cpu-baseline --no-extras --no-exact"
for m in 0 1; do
  export ESAC_PENDING_EXPERT_MAJOR=$m
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats$m -o t -- $CMD > $O/bench$m.json 2> $O/err$m.txt)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch$m -o t -- $CMD > /dev/null 2>&1)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc$m -o t -- $CMD > /dev/null 2>&1)
  python - $m $(find $O/stats$m -name "*.db" | head -1) $(find $O/fetch$m -name "*.db" | head -1) $(find $O/tcc$m -name "*.db" | head -1) <<'PY' | tee -a $O/r04_cfg5b_pending_order.txt
import sqlite3, sys, json
m, stats, fetch, tcc = sys.argv[1:5]
db = sqlite3.connect(stats)
rows = {r[0]: r for r in db.execute("select name,total_calls,total_duration,average from top_kernels")}
def cnt(path):
    d = sqlite3.connect(path)
    return {(k, c): v for k, c, v in d.execute("select kernel_name, counter_name, avg(value) from counters_collection group by kernel_name, counter_name")}
cf, ct = cnt(fetch), cnt(tcc)
print("ESAC_PENDING_EXPERT_MAJOR=%s" % m)
for name, r in sorted(rows.items(), key=lambda kv: -kv[1][2]):
    if "esac::k_sample" not in name and "k_pending" not in name and "k_bucket" not in name: continue
    f = cf.get((name, "FETCH_SIZE")); h, ms = ct.get((name, "TCC_HIT_sum")), ct.get((name, "TCC_MISS_sum"))
    print("  %-46s calls %4d avg %9.1f us  fetch %s  L2 hit %s" % (name.replace("void esac::","").replace("(esac::KArgs)","")[:46], r[1], r[3] / 1e3,
          ("%.2f GB (x2-corrected)" % (2 * 1024 * f / 1e9)) if f else "-", ("%.3f" % (h / (h + ms))) if h is not None and (h + ms) > 0 else "-"))
PY
  tail -1 $O/bench$m.json | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('  bench: ms_per_step %.3f' % d['ms_per_step'], {k['stage']: round(k['avg_us'],1) for k in d.get('kernels',[])})" | tee -a $O/r04_cfg5b_pending_order.txt
  rm -rf $O/stats$m $O/fetch$m $O/tcc$m
done
ESAC_PENDING_EXPERT_MAJOR=1 timeout 600 python -m pytest tests/test_gpu_parity_large.py -m gpu -q -k "config5b" 2>&1 | tail -3 | tee -a $O/r04_cfg5b_pending_order.txt
