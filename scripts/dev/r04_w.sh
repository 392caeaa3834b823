#!/bin/bash
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
ESAC_SLOT_TEAMS=0 timeout 600 python scripts/dev/bwd_diag.py 196 232 181 225 2>&1 | tail -40
