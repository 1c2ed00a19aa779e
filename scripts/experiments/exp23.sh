mkdir -p gpurun_out/exp23
O=gpurun_out/exp23
for N in 2 4; do
ACINO_DIST_BACKEND=gloo ACINO_FORCE_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29700+N)) bench.py --gpus $N --steps 10 --warmup 2 > $O/bench_gloo_$N.json 2> $O/bench_gloo_$N.err
echo "N=$N rc=$?"; cut -c1-1500 $O/bench_gloo_$N.json; tail -3 $O/bench_gloo_$N.err | cut -c1-300
done
