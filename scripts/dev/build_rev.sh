#!/bin/bash
# usage: build_rev.sh <git-rev> <out.so>  -- builds libesac_hip.so of another revision for same-box A/B runs
set -e
rev=$1; out=$2
d=/tmp/rev_$rev; rm -rf $d; mkdir -p $d
cd ${GRAFT_REPO_ROOT:-/root/repo}
git archive $rev esac_amd/csrc include | tar -x -C $d
cd $d/esac_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared $(ls *.hip) -ldl -L/opt/rocm/lib -lrccl -o $out  # -lrccl: revisions 8debdc1..b178b61 link it
