O=gpurun_out/r3t8; mkdir -p $O
timeout 600 python bench.py --gpus 2 --shared-gpu --steps 20 --warmup 5 > $O/bench_shared_gpu_2.json 2> $O/bench_shared_gpu_2.err
timeout 600 python bench.py --gpus 4 --shared-gpu --steps 20 --warmup 5 > $O/bench_shared_gpu_4.json 2> $O/bench_shared_gpu_4.err
for n in 2 4; do echo "== $n"; cut -c1-600 $O/bench_shared_gpu_$n.json; grep -v "amdgpu.ids\|Gloo\|socket" $O/bench_shared_gpu_$n.err | tail -8; done
