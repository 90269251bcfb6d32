for lib in "" build/var_mg16/liblgr_hip.so build/var_mg4/liblgr_hip.so; do
  if [ -n "$lib" ]; then export LGR_HIP_LIB=$GRAFT_REPO_ROOT/$lib; else unset LGR_HIP_LIB; fi
  echo "== ${lib:-in-tree (8)}"; python3 tools/bench_configs.py ransac 2>/dev/null
  python3 bench.py --matching cluster --steps 10 --warmup 2 --no-cpu-baseline --no-matcher-extremes --no-stage-rooflines 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cluster', round(d['ms_per_step'],2), round(d['stage_ms']['ransac'],3))"
done
