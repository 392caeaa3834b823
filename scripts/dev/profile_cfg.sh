# profile one bench.py workload: kernel trace + the PMC passes (each its own run) -> profiles/<round>_<cfg>_kernels.{json,txt}
# usage (on the GPU box): bash scripts/dev/profile_cfg.sh <cfg> <round> [extra bench args]
R=$GRAFT_REPO_ROOT
CFG=$1; RND=${2:-r04}; shift; shift
O=$R/gpurun_out/prof_${RND}_${CFG}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --config $CFG --no-cpu-baseline --no-extras --no-exact $*"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/stats -o t -- $CMD > $O/bench_under_rocprof.json 2> $O/err_stats.txt
i=0
for SET in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" \
           "SQ_INSTS_VALU SQ_INSTS_VALU_TRANS_F32 SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS"; do
  i=$((i+1))
  # (counter passes serialise the kernels: the speculative route's cross-stream hand-offs would time out in them and its polling
  # kernels would count 20 ms of spinning -- the counters are taken on the same kernels in stream order; durations come from the trace above)
  ESAC_SPECULATE=0 timeout 900 rocprofv3 --kernel-trace --pmc $SET -d $O/pmc$i -o t -- $CMD > /dev/null 2> $O/err_pmc$i.txt
done
cd $R
python scripts/profile_to_json.py --config $CFG --round $RND --stats $(find $O/stats -name "*.db" | head -1) \
   --pmc $(find $O/pmc* -name "*.db") --command "bench.py --config $CFG --no-cpu-baseline --no-extras --no-exact $*" --out-dir gpurun_out/profiles_$RND
tail -1 $O/bench_under_rocprof.json > gpurun_out/profiles_$RND/${RND}_${CFG}_bench_under_rocprof.json
rm -rf $O/stats $O/pmc*   # the rocpd databases are tens of MB each: only the summaries travel back
