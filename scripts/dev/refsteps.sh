R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 0 1 2 100; do
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rs$m -o r -- python $R/scripts/dev/refsteps.py $m 2>&1 | grep "max_ref_steps"
python $R/scripts/summarize_rocprof.py $(find $R/gpurun_out/rs$m -name "*.db" | head -1) | grep k_refine | head -1
done
