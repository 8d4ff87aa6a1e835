export TAG=r05_k
bash tools/gpu/run.sh tests tests/test_gpu_golden.py -k "NA_as_zero"
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29572 bench.py --gpus 1 --force-dist --workload c5 --scale 0.125 --no-cpu-baseline --steps 3 --warmup 1 2>gpurun_out/r05_k/c5_err.txt | grep "^{" | tail -1 | cut -c1-500 | tee gpurun_out/r05_k/force_dist_c5.txt
tail -3 gpurun_out/r05_k/c5_err.txt | cut -c1-300
