"""Copies the round-6 profile set from gpurun_out/r06prof (tools/gpu_profiles_r06.sh) into profiles/r06_* and derives profiles/r06_traffic.json
-- HBM bytes per launch of every kernel from the FETCH_SIZE / WRITE_SIZE passes (corrected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE under-reports by 2x on gfx950), the brick kernel's SQ / LDS ratios -- which bench.py reads for `roofline.traffic`
and `roofline.valu_frac`."""
import json, os, re, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
S = os.path.join(ROOT, "gpurun_out", "r06prof"); D = os.path.join(ROOT, "profiles")
for f in ("seq_kernel_stats.csv", "seq_probe.txt", "host_input_probe.txt", "bench_default.json", "bench_driver_cmd.json", "bench_dry_rank3of8.json", "bench_c1.json", "bench_c3.json", "bench_c4.json", "bench_c4_wavefront.json", "bench_c3_1000.json", "c4w_kernel_stats.csv", "bench_kernel_stats.csv",
          "bench_driver_cmd_kernel_stats.csv", "c1_kernel_stats.csv", "c3_kernel_stats.csv", "c4_kernel_stats.csv", "merge_kernel_stats.csv", "pmc_summary.txt",
          "parity_vs_faithful.json", "box.txt"):
    if os.path.exists(os.path.join(S, f)):
        shutil.copy(os.path.join(S, f), os.path.join(D, "r06_" + f))
sec, cur = {}, None
for line in open(os.path.join(D, "r06_pmc_summary.txt")):
    if line.startswith("== "):
        cur = line[3:].split(":")[0]; sec[cur] = {"command": line.split(": ", 1)[1].strip(), "k": {}}
    elif cur and line.strip():
        name = line[:58].strip()
        sec[cur]["k"][name] = {m.group(1): (float(m.group(2)), int(m.group(3))) for m in re.finditer(r"(\w+)=([0-9.e+]+)\(n=(\d+)\)", line[58:])}


def hbm(fetch_sec, write_sec, kernel):
    f, w = sec[fetch_sec]["k"][kernel]["FETCH_SIZE"], sec[write_sec]["k"][kernel]["WRITE_SIZE"]
    return {"kernel": kernel, "fetch_size_kb": f[0], "write_size_kb": w[0], "launches": f[1], "hbm_bytes_per_launch": int((2 * f[0] + w[0]) * 1024),
            "command": sec[fetch_sec]["command"].replace("FETCH_SIZE", "<FETCH_SIZE | WRITE_SIZE>")}


bd = json.load(open(os.path.join(S, "bench_driver_cmd.json")))
ib = [k for k in sec["fetch"]["k"] if "k_integrate_batch" in k][0]
lib_hash = open(os.path.join(S, "lib_source_hash.txt")).read().strip() if os.path.exists(os.path.join(S, "lib_source_hash.txt")) else None
out = {"lib_source_hash": lib_hash, "_source": "rocprofv3 --pmc passes (one counter group per run, --kernel-trace only) over the benched configuration, averages per launch, MI355X; "
                  "tools/gpu_profiles_r06.sh, raw output profiles/r06_pmc_summary.txt.  hbm_bytes_per_launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: the x2 on FETCH_SIZE is the "
                  "gfx950 correction of MI355X_MICROARCH.md; profiles/r04_fetch_calibration.txt (tools/ubench/fetch_calib.hip) shows it holds for 1- and 4-byte gathers too: the counter is 64 B per distinct 128-byte line, and the lines move whole.",
       "command": sec["fetch"]["command"].replace("FETCH_SIZE", "<CTR>")}
t = hbm("fetch", "write", ib)
launches = t["launches"]
sq, lds, tcc = sec["sq"]["k"][ib], sec["lds"]["k"][ib], sec["tcc"]["k"][ib]
# frames the PMC command integrates: 5 warm-up + 7 bursts of 20 timed + 20 of the per-kernel breakdown pass + 2 x 300 of the host-input legs, over `launches` launches
frames = 5 + 7 * 20 + 20 + 600
t.update({"frames_per_launch": frames / launches, "valu_wave_insts_per_launch": sq["SQ_INSTS_VALU"][0],
          "sq": {"SQ_WAIT_ANY/SQ_WAVE_CYCLES": sq["SQ_WAIT_ANY"][0] / sq["SQ_WAVE_CYCLES"][0], "SQ_ACTIVE_INST_VALU/SQ_WAVE_CYCLES": sq["SQ_ACTIVE_INST_VALU"][0] / sq["SQ_WAVE_CYCLES"][0],
                 "SQ_WAVES": sq["SQ_WAVES"][0], "SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE": lds["SQ_LDS_BANK_CONFLICT"][0] / lds["SQ_LDS_IDX_ACTIVE"][0],
                 "SQ_LDS_IDX_ACTIVE/SQ_BUSY_CYCLES": lds["SQ_LDS_IDX_ACTIVE"][0] / sq["SQ_BUSY_CYCLES"][0]},
          "tcc_hit": tcc["TCC_HIT_sum"][0], "tcc_miss": tcc["TCC_MISS_sum"][0]})
u = bd["config"]["frame_stats"]["unique"]
t["algorithmic_bytes_per_launch"] = 9 * u * t["frames_per_launch"]
t["traffic_over_algorithmic"] = t["hbm_bytes_per_launch"] / t["algorithmic_bytes_per_launch"]
out["integrate"] = t
fs = bd["config"]["frame_stats"]
P, V, segs = fs["p_valid"], fs["v_pcl"], 0.3e6
models = {"k_voxelize_depth": ("2*P_used + 8*P (pixel payload) + 64*V (one slot line per sensor voxel)", 2 * fs["p_used"] + 8 * P + 64 * V),
          "k_segments": ("64*V (slot) + 8*P (payloads) + 16*V (ray record) + 8*16*V (private segment slots)", 64 * V + 8 * P + 16 * V + 128 * V),
          "k_scatter": ("8*16*V read + 8*segments written + 12*V (slots cleared)", 128 * V + 8 * segs + 12 * V), "k_apply_slab": ("16 B per voxel of every slab slot read + 9 B per voxel updated", None),
          "k_plan": ("~16 B per active brick", None)}
for short in ("k_voxelize_depth", "k_segments", "k_scatter", "k_plan", "k_apply_slab"):
    k = [x for x in sec["fetch"]["k"] if short in x]
    if k:
        e = hbm("fetch", "write", k[0]); e["frames_per_launch"] = t["frames_per_launch"]
        m = models[short]
        e["byte_model"] = m[0]
        if m[1]:
            e["algorithmic_bytes_per_launch"] = m[1] * t["frames_per_launch"]; e["traffic_over_algorithmic"] = e["hbm_bytes_per_launch"] / e["algorithmic_bytes_per_launch"]
        out[short[2:]] = e
for c, short in ((1, "k_marching_cubes_lds"), (3, "k_octo_depth"), (4, "k_esdf_round")):
    k = [x for x in sec[f"fetch_c{c}"]["k"] if short in x]
    if k:
        e = hbm(f"fetch_c{c}", f"write_c{c}", k[0])
        if c == 1:      # one mesh = the brick sign summary + the tile kernel
            ks = [x for x in sec["fetch_c1"]["k"] if "k_mc_summary" in x]
            if ks:
                e2 = hbm("fetch_c1", "write_c1", ks[0])
                e["kernels"] = {"k_marching_cubes_lds": e["hbm_bytes_per_launch"], "k_mc_summary": e2["hbm_bytes_per_launch"]}
                e["hbm_bytes_per_launch"] += e2["hbm_bytes_per_launch"]; e["note"] = "per mesh: k_mc_summary + k_marching_cubes_lds"
        if c == 4:      # one ESDF update = all round launches of the update: the bench line's unit
            upd = sec["fetch_c4"]["k"][[x for x in sec["fetch_c4"]["k"] if "k_esdf_collect" in x][0]]["FETCH_SIZE"][1]
            e["launches_per_update"] = e["launches"] / upd
            e["hbm_bytes_per_launch"] = int(e["hbm_bytes_per_launch"] * e["launches_per_update"]); e["note"] = "per ESDF update (all k_esdf_round launches of one update)"
            if "sq_c4" in sec and "lds_c4" in sec:
                q4, l4 = sec["sq_c4"]["k"][k[0]], sec["lds_c4"]["k"][k[0]]
                e["sq"] = {"SQ_WAIT_ANY/SQ_WAVE_CYCLES": q4["SQ_WAIT_ANY"][0] / q4["SQ_WAVE_CYCLES"][0], "SQ_ACTIVE_INST_VALU/SQ_WAVE_CYCLES": q4["SQ_ACTIVE_INST_VALU"][0] / q4["SQ_WAVE_CYCLES"][0],
                           "SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE": l4["SQ_LDS_BANK_CONFLICT"][0] / l4["SQ_LDS_IDX_ACTIVE"][0],
                           "SQ_LDS_IDX_ACTIVE/SQ_BUSY_CYCLES": l4["SQ_LDS_IDX_ACTIVE"][0] / q4["SQ_BUSY_CYCLES"][0]}
        if c == 3:      # eight queued frames per launch (round 6): the bench line's unit is a frame
            e["frames_per_launch"] = 45.0 / e["launches"]; e["note"] = "k_octo_depth_batch: per launch of up to eight queued frames (the PMC command integrates 25 device-resident frames and 20 host images)"
        out[f"config{c}"] = e
if "fetch_c4w" in sec:      # configs[3] with esdf_mode 1: one update = k_esdf_diff + all k_esdf_wave launches
    k = [x for x in sec["fetch_c4w"]["k"] if "k_esdf_wave" in x]
    if k:
        e = hbm("fetch_c4w", "write_c4w", k[0])
        upd = sec["fetch_c4w"]["k"][[x for x in sec["fetch_c4w"]["k"] if "k_esdf_collect" in x][0]]["FETCH_SIZE"][1]
        e["launches_per_update"] = e["launches"] / upd
        e["hbm_bytes_per_launch"] = int(e["hbm_bytes_per_launch"] * e["launches_per_update"]); e["note"] = "per ESDF update (all k_esdf_wave launches of one update), esdf_mode 1"
        out["config4_wavefront"] = e
if "fetch_merge" in sec:
    for short in ("k_fuse_splat", "k_merge_pack", "k_merge_finish"):
        k = [x for x in sec["fetch_merge"]["k"] if short in x]
        if k:
            out[short[2:]] = hbm("fetch_merge", "write_merge", k[0])
if "sq_seq" in sec:      # the sequential semantics: the replay and grouping kernels of a 72-frame stream
    sq = {}
    for short in ("k_seq_replay", "k_seq_group", "k_seq_split", "k_seq_classify"):
        k = [x for x in sec["sq_seq"]["k"] if short + "<" in x or x.endswith(short)]
        if not k:
            continue
        q = sec["sq_seq"]["k"][k[0]]
        e = {"launches": q["SQ_WAVES"][1], "SQ_WAVES": q["SQ_WAVES"][0], "valu_wave_insts_per_launch": q["SQ_INSTS_VALU"][0],
             "SQ_WAIT_ANY/SQ_WAVE_CYCLES": q["SQ_WAIT_ANY"][0] / q["SQ_WAVE_CYCLES"][0], "SQ_ACTIVE_INST_VALU/SQ_WAVE_CYCLES": q["SQ_ACTIVE_INST_VALU"][0] / q["SQ_WAVE_CYCLES"][0]}
        if "fetch_seq" in sec and k[0] in sec["fetch_seq"]["k"]:
            e.update({kk: vv for kk, vv in hbm("fetch_seq", "write_seq", k[0]).items() if kk in ("fetch_size_kb", "write_size_kb", "hbm_bytes_per_launch")})
        sq[short] = e
    out["sequential"] = sq
json.dump(out, open(os.path.join(D, "r06_traffic.json"), "w"), indent=1)
print(json.dumps({k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a in ("hbm_bytes_per_launch", "traffic_over_algorithmic", "frames_per_launch", "sq", "launches_per_update")}) for k, v in out.items() if k not in ("_source", "command")}, indent=1))
