#!/bin/bash
# refinement cycle sections (pinned) for one workgroup and for a team of 8
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r04c
mkdir -p $O
ESAC_REFINE_TEAM=0 bash scripts/dev/cyc.sh > $O/cyc_team0.txt 2>&1
ESAC_REFINE_TEAM=8 bash scripts/dev/cyc.sh > $O/cyc_team8.txt 2>&1
paste $O/cyc_team0.txt $O/cyc_team8.txt | cut -c1-200
