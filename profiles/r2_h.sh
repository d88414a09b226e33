#!/bin/bash
set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -40 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_rgb.json 2> $O/bench_rgb.err
python bench.py --config images --steps 40 --warmup 5 > $O/bench_images.json 2> $O/bench_images.err
python bench.py --config images --perceptual --steps 20 --warmup 3 > $O/bench_images_perc.json 2> $O/bench_images_perc.err || true
python bench.py --config perceptual --steps 40 --warmup 5 --no-cpu-baseline > $O/bench_perc.json 2> $O/bench_perc.err
python profiles/shard_proxy.py > $O/shard_proxy.json 2> $O/shard_proxy.err
python - <<'PY'
import json
for f in ("bench_rgb","bench_images","bench_images_perc","bench_perc"):
    try:
        d=json.loads(open("gpurun_out/r2h/%s.json"%f).read().strip().splitlines()[-1])
        print(f, round(d["value"]), "ms/step %.3f"%d["ms_per_step"], d.get("reference_batch",{}).get("value"), d["config"].get("init_seconds"), d["config"].get("dropped_at_init"))
    except Exception as e: print(f, "ERR", e)
d=json.load(open("gpurun_out/r2h/shard_proxy.json"))
for r in d["rows"]: print(r)
PY
