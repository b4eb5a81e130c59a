"""profiles/r03/bounds.json from one run of scripts/profile_r03.sh: per-frame counter totals of k_shadow (PMC passes over
bench.py --steps 1 --warmup 0, which renders the cold frame and one timed frame) + the unit rates calibrated on the
same box (scripts/calib.hip).  bench.py prices the kernel's live time against them.
The file names the build the counts were taken on (rt_build_info of the library the profiled run loaded, from the
run's own JSON line): bench.py prints "stale" instead of a fraction when the loaded library is another.
    python scripts/make_bounds.py gpurun_out/prof_<tag> profiles/r03 <tag>"""
import ast
import json
import os
import re
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def kernel_counters(path, kernel="k_shadow"):
    for line in open(path):
        m = re.match(rf"^{kernel}(?:<\w+>)? calls (\d+) (\{{.*\}})\s*$", line)
        if m:
            return int(m.group(1)), {k: float(v) for k, v in ast.literal_eval(m.group(2)).items()}
    raise SystemExit(f"no {kernel} line in {path}")


def bench_line(name):  # bench.py's JSON line among rocprofv3's own messages
    return json.loads(next(l for l in open(os.path.join(src, name)) if l.startswith('{"metric"')))


build = bench_line("bench_trace.log")["build"]
for name in ("bench_pmc_units.log", "bench_pmc_fetch.log", "bench_pmc_l2.log", "bench_pmc_mix.log"):
    other = bench_line(name)["build"]
    assert other == build, (name, other, build)
calls, u = kernel_counters(os.path.join(src, "pmc_units.txt"))
_, f = kernel_counters(os.path.join(src, "pmc_fetch.txt"))
_, l2 = kernel_counters(os.path.join(src, "pmc_l2.txt"))
_, mx = kernel_counters(os.path.join(src, "pmc_mix.txt"))
# scripts/calib_mix.hip on the same box: nominal cycles of a SIMD per wave64 instruction, one instruction kind per line
cyc = {}
for line in open(os.path.join(src, "calib_mix.txt")):
    m = re.match(r"^(\S+)\s+[\d.]+ ms\s+([\d.]+) nominal cycles", line)
    if m:
        cyc[m.group(1)] = float(m.group(2))
launches = 16                      # k_shadow launches per 1080p 64-spp frame (16 batches of 2^23 paths)
frames = calls / launches          # bench.py --steps 1 --warmup 0 renders the cold frame + one step
calib = json.load(open(os.path.join(src, "calib.json")))
clock_hz = calib["clock_mhz"] * 1e6
cus = calib["cus"]
l1_hit = 1024.0 / calib["gather_l1_share1"]["lane_bytes_per_cycle_per_cu"]   # cycles per wave-load with 64 distinct L1-resident records
l2_srv = 1024.0 / calib["gather_share1"]["lane_bytes_per_cycle_per_cu"]     # ... with 64 distinct L2-served records
out = {
    "workload": {"width": 1920, "height": 1080, "spp": 64, "config": "config3"},
    "kernel": "k_shadow", "launches_per_frame": launches, "frames_profiled": frames, "tag": tag,
    "build": build,
    "per_frame": {
        "valu_insts": u["SQ_INSTS_VALU"] / frames,
        "salu_insts": u["SQ_INSTS_SALU"] / frames,
        "vmem_read_insts": u["SQ_INSTS_VMEM_RD"] / frames,
        "lds_insts": u["SQ_INSTS_LDS"] / frames,
        "tcp_accesses": u["TCP_TOTAL_CACHE_ACCESSES_sum"] / frames,
        "tcp_l2_read_req": u["TCP_TCC_READ_REQ_sum"] / frames,
        "ta_busy_frac": u["TA_BUSY_avr"] / (u["GRBM_GUI_ACTIVE"] / 8.0),   # GRBM_GUI_ACTIVE is summed over the 8 XCDs
        "ta_busy_max_frac": u["TA_BUSY_max"] / (u["GRBM_GUI_ACTIVE"] / 8.0),
        "wave_cycles_waiting_frac": u["SQ_WAIT_ANY"] / u["SQ_WAVE_CYCLES"],
        "wave_cycles_issue_stalled_frac": u["SQ_WAIT_INST_ANY"] / u["SQ_WAVE_CYCLES"],
        "gpu_cycles_per_xcd": u["GRBM_GUI_ACTIVE"] / 8.0 / frames,
        "hbm_fetch_bytes": f["FETCH_SIZE"] * 1024.0 / frames,     # rocprofv3 reports KiB
        "hbm_write_bytes": l2["WRITE_SIZE"] * 1024.0 / frames,
        "l2_hit_rate": l2["TCC_HIT_sum"] / (l2["TCC_HIT_sum"] + l2["TCC_MISS_sum"]),
        # the VALU instructions by the classes the SQ counts; "other" = moves, logic, shifts, compares, min / max, selects, permutes
        "valu_mix": {
            "fma_f32": mx["SQ_INSTS_VALU_FMA_F32"] / frames, "add_f32": mx["SQ_INSTS_VALU_ADD_F32"] / frames,
            "mul_f32": mx["SQ_INSTS_VALU_MUL_F32"] / frames, "cvt": mx["SQ_INSTS_VALU_CVT"] / frames,
            "trans_f32": mx["SQ_INSTS_VALU_TRANS_F32"] / frames, "int32": mx["SQ_INSTS_VALU_INT32"] / frames,
            "other": (mx["SQ_INSTS_VALU"] - mx["SQ_INSTS_VALU_FMA_F32"] - mx["SQ_INSTS_VALU_ADD_F32"] - mx["SQ_INSTS_VALU_MUL_F32"]
                      - mx["SQ_INSTS_VALU_CVT"] - mx["SQ_INSTS_VALU_TRANS_F32"] - mx["SQ_INSTS_VALU_INT32"]) / frames,
        },
        "valu_active_lanes_frac": mx["SQ_THREAD_CYCLES_VALU"] / (64.0 * mx["SQ_INSTS_VALU"]),
    },
    "calib": {
        "clock_hz": clock_hz, "cus": cus, "simds": 4 * cus,
        "valu_cycles_per_wave_inst_per_simd": calib["valu_5waves_per_simd"]["cycles_per_inst_per_simd_by_event"],
        # cycles of a SIMD per wave64 instruction by class: [cheapest, dearest] instruction of the class in calib_mix.txt
        "valu_class_cycles": {
            "fma_f32": [cyc["v_fma_f32(asm)"], cyc["v_fmac_f32"]],
            "add_f32": [cyc["v_add_f32"], cyc["v_sub_f32"]],
            "mul_f32": [cyc["v_mul_f32"], cyc["v_mul_f32"]],
            "cvt": [cyc["v_cvt_f32_ubyte0"], cyc["v_cvt_f32_ubyte0"]],
            "trans_f32": [cyc["v_rcp_f32"], cyc["v_rcp_f32"]],
            "int32": [min(cyc["v_add_u32"], cyc["v_sub_u32"]), max(cyc["v_mul_lo_u32"], cyc["v_min_u32"], cyc["v_bfe_u32"])],
            "other": [min(cyc["v_lshrrev_b32"], cyc["v_mov_b32"], cyc["v_and_b32"], cyc["v_xor_b32"]),
                      max(cyc["v_perm_b32"], cyc["v_max3_f32"], cyc["v_cmp_le_f32"], cyc["v_cndmask(sgpr)"], cyc["v_and_or_b32"])],
        },
        "tcp_l1_hit_accesses_per_clk_per_cu": 64.0 / l1_hit,
        "tcp_l2_served_accesses_per_clk_per_cu": 64.0 / l2_srv,
        "note": "scripts/calib_mix.hip: one instruction kind per register-only stream at 6 waves per SIMD (the figure "
                "valu_cycles_per_wave_inst_per_simd of scripts/calib.hip is a stream the compiler paired into v_pk_fma_f32, so it is "
                "per HALF of a packed instruction and prices no instruction k_shadow executes; kept for the record).  "
                "scripts/calib.hip on the same box: register-only v_fma_f32 stream at 5 waves per SIMD; 16-byte-per-lane gathers of "
                "64 distinct 32-byte records per wave instruction from a 16 KiB (L1-resident) and an 8 MiB (L2-served) table; "
                "rocprofv3 counts 64 TCP_TOTAL_CACHE_ACCESSES for such an instruction (profiles/r02/calib_pmc.txt)",
    },
}
json.dump(out, open(os.path.join(dst, "bounds.json"), "w"), indent=1)
for name in ("kernel_stats.csv", "pmc_units.txt", "pmc_fetch.txt", "pmc_l2.txt", "pmc_mix.txt", "pmc_wr.txt", "calib.json", "calib_mix.txt"):
    if not os.path.exists(os.path.join(src, name)):
        continue
    shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
print(json.dumps(out["per_frame"], indent=1))
