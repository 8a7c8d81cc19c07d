timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.txt 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.txt | tail -3
for s in "callspace 1500 777" "options 1000 778" "ordered" "frames 300" "large 300 779"; do timeout 1200 python tools/soak.py $s 2>&1 | tail -1; done
