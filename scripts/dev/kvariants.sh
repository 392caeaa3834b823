# same-box comparison of builds that differ in -D flags, per KERNEL (rocprofv3 kernel trace): usage like variants.sh plus a kernel-name filter
# bash scripts/dev/kvariants.sh "<bench args>" "<kernel substring>" "<name>=<flags>" ...
R=$GRAFT_REPO_ROOT
cd $R
ARGS=$1; shift
FILT=$1; shift
for v in "$@"; do
  name=${v%%=*}; flags=${v#*=}
  python esac_amd/build.py /tmp/lib_$name.so $flags > /dev/null 2>&1 &
done
wait
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  name=${v%%=*}
  rm -rf /tmp/kv_$name
  (cd /tmp && ESAC_HIP_LIB=/tmp/lib_$name.so timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kv_$name -o t -- python $R/bench.py $ARGS --no-cpu-baseline --no-extras > /dev/null 2>&1)
  python - $(find /tmp/kv_$name -name "*.db" | head -1) $name "$FILT" <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [r for r in db.execute("select name,total_calls,total_duration,average from top_kernels") if "esac::" in r[0]]
steps = max([r[1] for r in rows if "k_refine" in r[0]] + [1])
tot = sum(r[2] for r in rows) / steps
print("%-12s all kernels %.1f us/call; " % (sys.argv[2], tot / 1e3 if tot > 1e6 else tot) + "; ".join("%s %.1f" % (r[0].replace("void ","").replace("(esac::KArgs)","").replace("esac::","")[:28], r[2] / steps) for r in sorted(rows, key=lambda r: -r[2]) if sys.argv[3] in r[0]))
PY
done
done
