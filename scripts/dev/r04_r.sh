#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ESAC_REFINE_TEAM=8 bash scripts/dev/cyc.sh 2>&1 | grep "smp_\|total"
