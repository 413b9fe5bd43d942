"""Condense gpurun_out/<round>/ rocprofv3 CSVs into the tracked profiles/<round>/ summaries.

    python scripts/summarize_profiles.py r01

The two maze3d variants are traced in separate runs (`bench_maze.py --only discrete|continuous`) so each has
its own kernel-stats row. Writes profiles/<round>/pmc_summary.json (per-kernel averages of every collected counter, plus the
derived HBM traffic per launch: 2*FETCH_SIZE + WRITE_SIZE in KB — FETCH_SIZE on gfx950 reports half of
a coalesced stream's bytes, /opt/skills/guides/MI355X_MICROARCH.md §HBM) and copies the kernel-stats
CSVs and bench JSON lines. Also refreshes profiles/quadrotor_pmc.json, which bench.py reads for its
`roofline.traffic` field."""
import collections
import csv
import json
import os
import shutil
import sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = os.path.join(ROOT, "gpurun_out", rnd)
P = os.path.join(ROOT, "profiles", rnd)
os.makedirs(P, exist_ok=True)


def agg(path, key):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    if not os.path.exists(path):
        return {}, {}
    for r in csv.DictReader(open(path)):
        if key in r["Kernel_Name"]:
            out[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size")
                    if k in r}
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}, meta


summary = {}
for kernel, prefix, stem in (("quadrotor_step_kernel", "quad", "q"), ("maze3d_step_kernel", "maze", "m"),
                             ("walker_step_wave_kernel", "walker", "w")):
    merged, meta = {}, {}
    for d in ("pmc_sq", "pmc_fetch", "pmc_write"):
        a, m = agg(os.path.join(R, "%s_%s" % (prefix, d), "%s_counter_collection.csv" % stem), kernel)
        merged.update(a.get(kernel, {}))
        meta = m or meta
    if "FETCH_SIZE" in merged and "WRITE_SIZE" in merged:
        merged["hbm_bytes_per_launch"] = (2.0 * merged["FETCH_SIZE"] + merged["WRITE_SIZE"]) * 1024.0
        merged["hbm_read_bytes_per_launch"] = 2.0 * merged["FETCH_SIZE"] * 1024.0
        merged["hbm_write_bytes_per_launch"] = merged["WRITE_SIZE"] * 1024.0
    if "SQ_WAVES" in merged:
        w = merged["SQ_WAVES"]
        merged["valu_insts_per_wave"] = merged.get("SQ_INSTS_VALU", 0) / w
        merged["wave_cycles_per_wave"] = merged.get("SQ_WAVE_CYCLES", 0) * 4 / w      # counter is in quad-cycles
        merged["valu_active_frac"] = merged.get("SQ_ACTIVE_INST_VALU", 0) / max(merged.get("SQ_WAVE_CYCLES", 1), 1)
        merged["wait_any_frac"] = merged.get("SQ_WAIT_ANY", 0) / max(merged.get("SQ_WAVE_CYCLES", 1), 1)
        merged["wait_inst_frac"] = merged.get("SQ_WAIT_INST_ANY", 0) / max(merged.get("SQ_WAVE_CYCLES", 1), 1)
    merged["dispatch"] = meta
    summary[kernel] = merged
# the contact-rich walker batch (bench.py C4_grounded_*): its own counters next to the airborne batch's
g, gmeta = agg(os.path.join(R, "walker_grounded_pmc_sq", "w_counter_collection.csv"), "walker_step_wave_kernel")
g = g.get("walker_step_wave_kernel", {})
if "SQ_WAVES" in g:
    w = g["SQ_WAVES"]
    g["valu_insts_per_wave"] = g.get("SQ_INSTS_VALU", 0) / w
    g["wave_cycles_per_wave"] = g.get("SQ_WAVE_CYCLES", 0) * 4 / w
    g["valu_active_frac"] = g.get("SQ_ACTIVE_INST_VALU", 0) / max(g.get("SQ_WAVE_CYCLES", 1), 1)
    g["wait_any_frac"] = g.get("SQ_WAIT_ANY", 0) / max(g.get("SQ_WAVE_CYCLES", 1), 1)
    g["dispatch"] = gmeta
    summary["walker_step_wave_kernel (C4 grounded: every robot lying on the ground)"] = g
json.dump(summary, open(os.path.join(P, "pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
for src, dst in (("quad_trace/q_kernel_stats.csv", "quadrotor_bench_kernel_stats.csv"),
                 ("quad_graph_trace/q_kernel_stats.csv", "quadrotor_bench_graph_kernel_stats.csv"),
                 ("maze_trace/m_kernel_stats.csv", "maze_bench_kernel_stats.csv"),
                 ("maze_discrete_trace/m_kernel_stats.csv", "maze3d_discrete_kernel_stats.csv"),
                 ("maze_continuous_trace/m_kernel_stats.csv", "maze3d_continuous_kernel_stats.csv"),
                 ("bench_eager.json", "bench_eager.json"), ("bench_force_dist.json", "bench_force_dist.json"),
                 ("bench_mixed_force_dist.json", "bench_mixed_force_dist.json"),
                 ("walker_trace/w_kernel_stats.csv", "walker_bench_kernel_stats.csv"),
                 ("a1_trace/a_kernel_stats.csv", "bench_all_secondary_kernel_stats.csv"),
                 ("bench.json", "bench.json"), ("bench_maze.jsonl", "bench_maze.jsonl"),
                 ("bench_walker.jsonl", "bench_walker.jsonl"),
                 ("walker_grounded_trace/w_kernel_stats.csv", "walker_grounded_kernel_stats.csv"),
                 ("bench_walker_grounded.jsonl", "bench_walker_grounded.jsonl"), ("bench_20steps.json", "bench_20steps.json")):
    if os.path.exists(os.path.join(R, src)):
        shutil.copy(os.path.join(R, src), os.path.join(P, dst))
# north_star batch sizes (2^17 / 2^20 quadrotors on one GPU): kernel-stats CSV + the SQ counters of a separate --pmc pass
north = {}
for nn in (131072, 1048576):
    src = os.path.join(R, "quad_%d_trace" % nn, "q_kernel_stats.csv")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(P, "quadrotor_%denvs_kernel_stats.csv" % nn))
    a, meta = agg(os.path.join(R, "quad_%d_pmc_sq" % nn, "q_counter_collection.csv"), "quadrotor_step_kernel")
    m = a.get("quadrotor_step_kernel", {})
    if "SQ_WAVES" in m:
        w = m["SQ_WAVES"]
        m["valu_insts_per_wave"] = m.get("SQ_INSTS_VALU", 0) / w
        m["valu_active_frac"] = m.get("SQ_ACTIVE_INST_VALU", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
        m["wait_any_frac"] = m.get("SQ_WAIT_ANY", 0) / max(m.get("SQ_WAVE_CYCLES", 1), 1)
        m["dispatch"] = meta
        north[str(nn)] = m
if north:
    json.dump(north, open(os.path.join(P, "pmc_quadrotor_north_star.json"), "w"), indent=1)
    print(json.dumps(north, indent=1))
if "hbm_bytes_per_launch" in summary.get("quadrotor_step_kernel", {}):
    q = summary["quadrotor_step_kernel"]
    json.dump({"round": rnd, "kernel": "quadrotor_step_kernel", "envs": 65536,
               "hbm_bytes_per_launch": q["hbm_bytes_per_launch"],
               "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes; bytes = (2*FETCH_SIZE + "
                         "WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE counts half of a coalesced stream)"},
              open(os.path.join(ROOT, "profiles", "quadrotor_pmc.json"), "w"), indent=1)
