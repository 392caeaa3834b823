#!/bin/bash
# everything a round commits as evidence, in one GPU call: the GPU test suite, the measurement set, the soaks, the rocprofv3 timelines
RND=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
P=gpurun_out/profiles_$RND
mkdir -p $P
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -i "passed\|failed\|error" | tail -5 > $P/${RND}_gpu_tests.txt
bash scripts/dev/measure_round.sh $RND > $P/${RND}_measure_log.txt 2>&1
bash scripts/dev/soak_round.sh $RND > $P/${RND}_soak_log.txt 2>&1
for c in cfg2 cfg3 cfg4; do scripts/dev/timeline.sh ${RND}_timeline_$c scripts/dev/fwd_loop.py $c > /dev/null 2>&1; cp gpurun_out/${RND}_timeline_$c.txt $P/; done
ESAC_SPECULATE=0 scripts/dev/timeline.sh ${RND}_timeline_cfg3_stream_order scripts/dev/fwd_loop.py cfg3 > /dev/null 2>&1; cp gpurun_out/${RND}_timeline_cfg3_stream_order.txt $P/
scripts/dev/timeline.sh ${RND}_timeline_backward scripts/dev/bwd_loop.py 100 > /dev/null 2>&1; cp gpurun_out/${RND}_timeline_backward.txt $P/
scripts/dev/spec_ab.sh ESAC_SPECULATE=0 ESAC_X=1 ESAC_SPECULATE=0 ESAC_X=1 > $P/${RND}_spec_ab_final.txt 2>&1
cat $P/${RND}_gpu_tests.txt; tail -30 $P/${RND}_measure_log.txt | cut -c1-400; cat $P/${RND}_soak_log.txt | tail -12; cat $P/${RND}_spec_ab_final.txt
