cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
for s in 0 1 2 3 0 1 2 3; do echo "strict=$s" >> gpurun_out/c6/ab.txt; LORA_HIP_STRICT_SYNC=$s LORA_HIP_DEBUG=1 timeout 300 python bench.py --steps 40 --no-cpu-baseline 2>gpurun_out/c6/err.txt | cut -c1-180 >> gpurun_out/c6/ab.txt; grep "per-job avg" gpurun_out/c6/err.txt | tail -1 >> gpurun_out/c6/ab.txt; done
cat gpurun_out/c6/ab.txt
