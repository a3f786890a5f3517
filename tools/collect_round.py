#!/usr/bin/env python3
"""Collects what tools/final_round.sh (profile_all.sh) left under gpurun_out/ into profiles/<round>_* (after tools/profile_summary.py <tag> <round> for every
tag) and prints the table.  usage: tools/collect_round.py r04"""
import json, os, shutil, sys
RND = sys.argv[1] if len(sys.argv) > 1 else "r04"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
out={"command":"tools/pmc_walker.sh sq_sf7 ; tools/pmc_walker.sh sq_sf9 --config 3 --sf 9 ; tools/pmc_walker.sh sq_sf12 --config 3 --sf 12 (six rocprofv3 --kernel-trace --pmc passes each, bench.py --steps 3 --warmup 1 --no-cpu-baseline; tools/pmc_summary.py), part of tools/profile_all.sh on the round's final sources",
     "units":"means per dispatch over the device; SQ_*_CYCLES, SQ_ACTIVE_* and SQ_WAIT_* count quad-cycles (x4 = clocks); GRBM_GUI_ACTIVE is summed over the 8 XCDs","source_hash":bench.source_hash(),"kernels":{}}
for tag in ("sq_sf7","sq_sf9","sq_sf12"):
    d=json.load(open("gpurun_out/%s.json"%tag))
    for k,v in d.items():
        if "walker" not in k: continue
        c={n:x["mean"] for n,x in v.items()}
        g=c["GRBM_GUI_ACTIVE"]
        out["kernels"][k]={"counters":c,"derived":{
            "valu_busy_fraction (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE / 8))": round(c["SQ_ACTIVE_INST_VALU"]*4/(1024*g/8),3),
            "wavefront_waiting_fraction (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": round(c["SQ_WAIT_ANY"]/c["SQ_WAVE_CYCLES"],3),
            "waiting_for_an_instruction_to_issue (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": round(c["SQ_WAIT_INST_ANY"]/c["SQ_WAVE_CYCLES"],3),
            "valu_wave_instructions_per_pass": c["SQ_INSTS_VALU"],
            "lds_bank_conflict_fraction (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE)": round(c["SQ_LDS_BANK_CONFLICT"]/max(c["SQ_LDS_IDX_ACTIVE"],1),4),
            "mfma_ops": c.get("SQ_INSTS_VALU_MFMA_MOPS_F32",0)}}
json.dump(out,open("profiles/%s_sq_counters.json" % RND,"w"),indent=1)
for a,b in (("default_line","default_bench_line"),("default_grad_line","default_grad_bench_line"),("work_line","work_bench_line"),("cfg4_line","cfg4_bench_line"),("cfg4_8s_line","cfg4_8s_bench_line"),("cfg4_2s_line","cfg4_2s_bench_line"),("torchrun1_line","torchrun_world1_bench_line"),("streams1_line","streams1_bench_line"),("mux_cfg4_2s_line","mux_cfg4_2s_bench_line"),("split1_line","split_world1_bench_line"),("default_fast_sync_line","default_fast_sync_bench_line"),("cfg4_2s_ordinary_line","cfg4_2s_ordinary_bench_line"),("sf7_256_line","sf7_256_default_bench_line"),("sf8_256_line","sf8_256_default_bench_line"),("sf7_256_lanes2_line","sf7_256_lanes2_bench_line"),("sf8_256_lanes2_line","sf8_256_lanes2_bench_line"),("sf9_256_lanes2_line","sf9_256_lanes2_bench_line"),("cfg4_2s_lanes1_line","cfg4_2s_lanes1_bench_line"),("sf7_d4_line","sf7_d4_bench_line"),("sf7_d2_line","sf7_d2_bench_line"),("sf8_d4_line","sf8_d4_default_bench_line"),("sf8_d2_line","sf8_d2_bench_line"),("sf9_d4_line","sf9_d4_bench_line"),("sf9_d2_line","sf9_d2_bench_line"),("sf8_d4_generic_line","sf8_d4_generic_bench_line")):
    shutil.copy("gpurun_out/%s.json"%a,"profiles/%s_%s.json"%(RND,b))
for f in ("noise60","noise40","noise35","noise30","noise60_norepair","noise50_sf9","noise50_sf12","noise50_sf9_norepair","noise50_cfg4_2s","noise50_cfg4_32s","noise50_cfg4_2s_norepair"):
    if os.path.exists("gpurun_out/%s_line.json"%f):
        shutil.copy("gpurun_out/%s_line.json"%f,"profiles/%s_%s_bench_line.json"%(RND,f)); d=json.load(open("gpurun_out/%s_line.json"%f)); print(f, d["value"], d["ms_per_step"])
if os.path.exists("gpurun_out/sf6_walker.txt"): shutil.copy("gpurun_out/sf6_walker.txt","profiles/%s_sf6_walker.txt"%RND)
for tag in ["sf7","sf8","sf7_256","sf8_256","sf9","sf10","sf11","sf12","sf9_1024","sf7_grad","sf9_grad","sf12_grad","sf8_d4"]:
    line=json.load(open("profiles/%s_%s_bench_line.json"%(RND,tag))); pmc=json.load(open("profiles/%s_%s_pmc_traffic.json"%(RND,tag)))
    assert pmc["source_hash"]==bench.source_hash(), tag
    r=line["roofline"]; n=line["config"]["items_per_gpu"]; rp=pmc["rocprof_walker_avg_ms_per_pass"]
    print("%-10s %-26s value %7.1f Gs/s  kernel %.4f ms frac %.4f | rocprof %.4f ms frac %.4f | traffic %.2fx write %.3f GB | cpu %.1f (ref %.1f) | grad2 %s"%(tag, r["kernel"], line["value"]/1e3, r["kernel_ms_per_pass"], r["frac"], rp, 8*n/(rp*1e-3)/1e9/8000, pmc["traffic_over_algorithmic"], pmc["write_bytes_per_pass_raw"]/1e9, (line.get("cpu_baseline") or {}).get("value",0), ((line.get("cpu_baseline") or {}).get("reference_build") or {}).get("value",0), (line.get("reference_default_demodulator") or {}).get("frac")))
for f in ["default_line","default_fast_sync_line","default_grad_line","work_line","cfg4_line","cfg4_8s_line","cfg4_2s_line","torchrun1_line","streams1_line","mux_cfg4_2s_line","split1_line","sf7_256_line","sf8_256_line","sf7_256_lanes2_line","sf8_256_lanes2_line","sf9_256_lanes2_line","cfg4_2s_lanes1_line","sf7_d4_line","sf7_d2_line","sf8_d4_line","sf8_d2_line","sf9_d4_line","sf9_d2_line","sf8_d4_generic_line"]:
    d=json.load(open("gpurun_out/%s.json"%f)); print(f, d["value"], d.get("roofline",{}).get("frac"), d.get("roofline",{}).get("frac_rocprof"), d["config"].get("bit_exact_vs_expected"), d["config"].get("process_group"), (d.get("reference_default_demodulator") or {}).get("frac"), (d.get("one_handle_per_channel") or {}).get("value"))
for k,v in out["kernels"].items(): print(k, list(v["derived"].values())[:4])
