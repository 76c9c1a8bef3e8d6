cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for cfg in "none 1" "7 0" "7 1"; do
  set -- $cfg
  if [ $1 = none ]; then unset FB_FUSE_PARTS; else export FB_FUSE_PARTS=$1; export FB_VADP_STACK=1; fi
  if [ $2 = 0 ]; then export FB_FUSE_UPD=0; else unset FB_FUSE_UPD; fi
  timeout 300 python bench.py --no-cpu-baseline --no-secondary --no-single --steps 200 --warmup 20 --chain unfused 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('parts=$1 fuse_upd=$2', round(d['value']))"
done; done
