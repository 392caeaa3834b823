R=$GRAFT_REPO_ROOT
cd $R
for cfg in cfg3 cfg4; do
echo "== $cfg"
bash scripts/dev/variants.sh "--config $cfg --steps 60 --warmup 6" "ho64=-DX1" "ho128=-DESAC_HANDOVER=128" "lat4096=-DESAC_LATENCY_MAX=4096" "lat512=-DESAC_LATENCY_MAX=512" 2>&1 | tail -4
done
