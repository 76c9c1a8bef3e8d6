cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r2a/pytest.log
for i in 1 2 3 4 5 6 7 8; do
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-single > gpurun_out/r2a/b20_$i.json 2>> gpurun_out/r2a/bench_20.err
done
