cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c38
export LORA_BENCH_CACHE=/dev/shm/lora_bench
{
echo "## cfg4 2s"; REPS=2 bash tools/ab.sh "--config 4 --seconds 2 --steps 40" ab/def.so ab/detk2.so
echo "## cfg4 8s"; REPS=1 bash tools/ab.sh "--config 4 --seconds 8 --steps 20" ab/def.so ab/detk2.so
echo "## cfg3 sf9"; REPS=1 bash tools/ab.sh "--config 3 --sf 9 --steps 30" ab/def.so ab/detk2.so
echo "## cfg3 sf11"; REPS=1 bash tools/ab.sh "--config 3 --sf 11 --steps 12" ab/def.so ab/detk2.so
echo "## cfg3 sf12 64 packets"; REPS=1 bash tools/ab.sh "--config 3 --sf 12 --packets 64 --steps 12" ab/def.so ab/detk2.so
} > gpurun_out/c38/ab.txt 2>&1
cat gpurun_out/c38/ab.txt
