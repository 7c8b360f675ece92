set -e
mkdir -p gpurun_out/r6
python -m pytest tests/test_gpu_graph.py -x -q -m gpu > gpurun_out/r6/graph_tests.log 2>&1 || { tail -40 gpurun_out/r6/graph_tests.log; exit 1; }
tail -2 gpurun_out/r6/graph_tests.log
python tools/bench_tiles.py --res 256 512 1024 --streams 1 2 --tiles 2000 --batch 2>&1 | tee gpurun_out/r6/tiles_graph.txt
