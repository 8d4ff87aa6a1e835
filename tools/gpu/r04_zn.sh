#!/bin/bash
# round 4, step zn: the --gpus N code path on one rank (sharded engine + RCCL collectives with world_size 1, launched as the driver launches it), both exchanges
R=$GRAFT_REPO_ROOT; O=gpurun_out/r04_zn; mkdir -p $R/$O; cd $R
for ag in collective p2p; do
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 1 --force-dist --steps 3 --warmup 1 --allgather $ag 2>$O/err_$ag.txt | tail -1 | cut -c1-900 | tee -a $O/lines.txt
done
