set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
for c in c2 c3 c4 c5; do
  timeout -k 10 300 python bench.py --config $c --steps 20 --warmup 3 > gpurun_out/final/bench_$c.json 2> gpurun_out/final/bench_$c.err || echo "bench $c failed"
done
timeout -k 10 200 python tools/rank_share_time.py c4 > gpurun_out/final/rank_share_c4.log 2>&1 || true
timeout -k 10 200 python tools/rank_share_time.py c5 1 8 > gpurun_out/final/rank_share_c5.log 2>&1 || true
python - <<'PY' > gpurun_out/final/commit_ms.log 2>&1
import time, esctp1raytracer_amd as esc
r = esc.Renderer(0)
for cfg in ("c3", "c4", "c5"):
    sc = esc.Scene.synthetic(cfg)
    r.upload(sc); r.synchronize()
    t0 = time.perf_counter(); r.upload(sc); r.synchronize(); t1 = time.perf_counter()
    print(cfg, "upload (stage + commit incl. group build) ms", (t1 - t0) * 1e3)
PY
for v in "" "ESC_GROUPS=0"; do for c in "c3 3840 2160" "c4 3840 2160" "c5 7680 4320"; do env $v timeout -k 10 100 python tools/split_time.py $c >> gpurun_out/final/splits.log 2>&1; done; done
cat gpurun_out/final/splits.log gpurun_out/final/commit_ms.log gpurun_out/final/rank_share_c4.log
