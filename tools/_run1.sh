python -m pytest tests/test_graph_gpu.py -m gpu -x -q 2>&1 | tail -2
for b in 1 128; do
echo "batch $b"; python bench.py --batch $b --steps 6 --warmup 1 --no-cpu-baseline --no-frontend 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])"
done
echo "batch 128 w6"; SSLAM_CHOL_TAIL_WIDTH=6 python bench.py --batch 128 --steps 6 --warmup 1 --no-cpu-baseline --no-frontend 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['kernel_ms'])"
