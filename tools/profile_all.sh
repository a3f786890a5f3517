#!/bin/bash
# tools/profile_all.sh - the round's evidence set on ONE GPU box (gpurun -- 'bash tools/profile_all.sh'; locally delete gpurun_out/prof_* first): per SF kernel stats + PMC traffic + bench line (tools/profile_round.sh), then the default
# line, the reference-API path and config 4
cd $GRAFT_REPO_ROOT
export LORA_BENCH_CACHE=/dev/shm/lora_bench
mkdir -p gpurun_out
{
tools/profile_round.sh sf7
export PROFILE_LINE_FLAGS=--no-cpu-baseline   # (the CPU legs once, in the sf7 set and the default line; the gradient second line needs them on: see below)
tools/profile_round.sh sf8 --config 3 --sf 8 --packets 1024
tools/profile_round.sh sf7_256 --config 3 --sf 7 --packets 256    # config 3's own cell size (BASELINE: 256 packets per SF): one workgroup per CU
tools/profile_round.sh sf8_256 --config 3 --sf 8 --packets 256
PROFILE_STEPS=8 tools/profile_round.sh sf9 --config 3 --sf 9
PROFILE_STEPS=8 tools/profile_round.sh sf10 --config 3 --sf 10
PROFILE_STEPS=6 tools/profile_round.sh sf11 --config 3 --sf 11
PROFILE_STEPS=4 tools/profile_round.sh sf12 --config 3 --sf 12
PROFILE_STEPS=6 tools/profile_round.sh sf9_1024 --config 3 --sf 9 --packets 1024   # more jobs than CUs (four packets per workgroup)
# the reference's shipped demodulator (gradient, decoder_impl.cc:499) on the same workloads: its own kernels
tools/profile_round.sh sf7_grad --demod 0
PROFILE_STEPS=8 tools/profile_round.sh sf9_grad --config 3 --sf 9 --demod 0
PROFILE_STEPS=4 tools/profile_round.sh sf12_grad --config 3 --sf 12 --demod 0
python bench.py 2>/dev/null | tail -1 > gpurun_out/default_line.json
LORA_HIP_STRICT_SYNC=0 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/default_fast_sync_line.json   # the closed-form SYNC alone (LORA_HIP_FLAG_FAST_SYNC): what the exact re-evaluation costs
python bench.py --path work 2>/dev/null | tail -1 > gpurun_out/work_line.json
python bench.py --config 4 --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/cfg4_line.json
python bench.py --config 4 --seconds 8 --steps 30 --warmup 5 2>/dev/null | tail -1 > gpurun_out/cfg4_8s_line.json
python bench.py --config 4 --seconds 2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg4_2s_line.json                               # few jobs for the device: a decoupled pass (header-only jobs + payload pass)
LORA_HIP_DECOUPLED=0 python bench.py --config 4 --seconds 2 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg4_2s_ordinary_line.json   # ... and the same pass by the complete kernels
PROFILE_STEPS=20 tools/profile_round.sh cfg4_2s --config 4 --seconds 2   # kernel stats of the decoupled pass: walker3_kernel_sf9_skip, demod_symbols_wave_kernel<9, 2>, payload_chain_kernel
python bench.py --demod 0 2>/dev/null | tail -1 > gpurun_out/default_grad_line.json
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/torchrun1_line.json
python bench.py --streams 1 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/streams1_line.json
python bench.py --config 3 --sf 7 --packets 256 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/sf7_256_line.json   # config 3's own cell size at SF7 / SF8: no more jobs than CUs (the plan for one workgroup per CU; SF8: walker2_kernel_sf8_wide)
python bench.py --config 3 --sf 8 --packets 256 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/sf8_256_line.json
for sf in 7 8 9; do python bench.py --config 3 --sf $sf --packets 256 --lanes 2 --no-cpu-baseline --no-grad-line 2>/dev/null | tail -1 > gpurun_out/sf${sf}_256_lanes2_line.json; done   # two passes in flight on streams of their own: the job rate of a cell that leaves half of every CU idle
python bench.py --config 4 --seconds 2 --lanes 1 --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg4_2s_lanes1_line.json                          # config 4's default is six lanes (bench.py --lanes): this is the single pipeline of round 5
# decimation 2 / 4 (round 6: walker2's LD builds, lora_wave_decim.inc.hip): the config-3 cell at 500 / 250 ksps, FFT demodulator; one cell on the generic kernels beside them
for sf in 7 8 9; do for d in 4 2; do
  python bench.py --config 3 --sf $sf --packets $([ $sf = 9 ] && echo 256 || echo 1024) --samp-rate $([ $d = 4 ] && echo 5e5 || echo 2.5e5) --no-cpu-baseline --no-grad-line 2>/dev/null | tail -1 > gpurun_out/sf${sf}_d${d}_line.json
done; done
LORA_HIP_NO_FAST=1 python bench.py --config 3 --sf 8 --packets 1024 --samp-rate 5e5 --steps 3 --warmup 1 --min-seconds 0.1 --no-cpu-baseline --no-grad-line 2>/dev/null | tail -1 > gpurun_out/sf8_d4_generic_line.json
PROFILE_LINE_FLAGS="--no-cpu-baseline --no-grad-line" tools/profile_round.sh sf8_d4 --config 3 --sf 8 --packets 1024 --samp-rate 5e5
{ for d in 8 4; do python tools/sf6_bench.py $d 2>/dev/null; LORA_HIP_NO_FAST=1 python tools/sf6_bench.py $d 2>/dev/null; done; } > gpurun_out/sf6_walker.txt   # SF6 (implicit header; 512 streams = 512 jobs): walker2's LD builds against the generic kernels
# noise over the stream, idle gaps included (LORA_BENCH_NOISE_DB: in-band SNR; BASELINE's workloads are noiseless): the cuts whose speculative job sits one sample beside the true trajectory are
# repaired in the probe launch (lora_stitch.hpp, round 6); LORA_HIP_NO_REPAIR=1: the serial path for each of them, as before
for n in 60 40 35 30; do LORA_BENCH_NOISE_DB=$n python bench.py --no-cpu-baseline --no-grad-line 2>/dev/null | tail -1 > gpurun_out/noise${n}_line.json; done
LORA_HIP_NO_REPAIR=1 LORA_BENCH_NOISE_DB=60 python bench.py --no-cpu-baseline --no-grad-line --steps 3 --warmup 1 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/noise60_norepair_line.json
for sf in 9 12; do LORA_BENCH_NOISE_DB=50 python bench.py --config 3 --sf $sf --no-cpu-baseline --no-grad-line 2>/dev/null | tail -1 > gpurun_out/noise50_sf${sf}_line.json; done
for sec in 2 32; do LORA_BENCH_NOISE_DB=50 python bench.py --config 4 --seconds $sec --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/noise50_cfg4_${sec}s_line.json; done   # decoupled passes: explicit probes, repaired in a launch of their own
LORA_HIP_NO_REPAIR=1 LORA_BENCH_NOISE_DB=50 python bench.py --config 4 --seconds 2 --no-cpu-baseline --steps 5 --warmup 2 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/noise50_cfg4_2s_norepair_line.json
LORA_HIP_NO_REPAIR=1 LORA_BENCH_NOISE_DB=50 python bench.py --config 3 --sf 9 --no-cpu-baseline --no-grad-line --steps 3 --warmup 1 --min-seconds 0 2>/dev/null | tail -1 > gpurun_out/noise50_sf9_norepair_line.json
tools/pmc_walker.sh sq_sf7
tools/pmc_walker.sh sq_sf9 --config 3 --sf 9
tools/pmc_walker.sh sq_sf12 --config 3 --sf 12
for t in sq_sf7 sq_sf9 sq_sf12; do python tools/pmc_summary.py gpurun_out/$t > gpurun_out/$t.json; rm -rf gpurun_out/$t; done
python bench.py --path mux --config 4 --seconds 2 --steps 5 2>/dev/null | tail -1 > gpurun_out/mux_cfg4_2s_line.json
python bench.py --split --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/split1_line.json
} > gpurun_out/profile_all.log 2>&1
# keep what travels back small: the per-dispatch traces are not needed once summarised... (kernel_stats + counter_collection csv only)
find gpurun_out/prof_* -type f ! -name "*kernel_stats.csv" ! -name "*counter_collection.csv" ! -name "*.log" ! -name "line.json" -delete 2>/dev/null
du -sh gpurun_out | tail -1
tail -30 gpurun_out/profile_all.log
