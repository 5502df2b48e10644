cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 500 python -m pytest tests/test_gpu_parity.py -x -q -k "two_rank" > gpurun_out/t5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/t5.log
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --steps 6 --warmup 2 --copies 64 --dist-backend gloo --same-device --no-cpu-baseline > gpurun_out/b_2rank_p.json 2> gpurun_out/b_2rank_p.err
timeout 200 python bench.py --no-cpu-baseline > gpurun_out/b_p3.json 2> gpurun_out/b_p3.err
