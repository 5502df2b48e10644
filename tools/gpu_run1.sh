cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "sha or fragmenter or two_rank or journaling" > gpurun_out/t1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t1.log
timeout 200 python bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_serial.json 2> gpurun_out/b_serial.err
ZPQ_SHA_WAVES=4 timeout 200 python bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_serial_w4.json 2> gpurun_out/b_serial_w4.err
ZPQ_SHA_WAVES=3 timeout 200 python bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_serial_w3.json 2> gpurun_out/b_serial_w3.err
ZPQ_SHA_NO_ORDER=1 timeout 200 python bench.py --pipeline 1 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/b_serial_noorder.json 2> gpurun_out/b_serial_noorder.err
for d in 3 4 6; do timeout 200 python bench.py --pipeline $d --steps 12 --warmup 3 --no-cpu-baseline > gpurun_out/b_p$d.json 2> gpurun_out/b_p$d.err; done
