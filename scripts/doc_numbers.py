"""Print the figures DESIGN.md / README.md quote from profiles/<round>/ (so the prose can be synced with the files):
    python scripts/doc_numbers.py r03"""
import csv, json, sys
R = sys.argv[1] if len(sys.argv) > 1 else "r03"
D = "profiles/%s/" % R
last = lambda f: json.loads(open(D + f).read().strip().splitlines()[-1])
b, e = last("bench.json"), last("bench_eager.json")
s = b["secondary"]
print("headline value %.4g  wall ms/step graph %.5f eager %.5f  HIP events us graph %.2f eager %.2f  frac %.4f copy-frac %.4f  GB/s %.0f" % (
    b["value"], b["ms_per_step"], e["ms_per_step"], b["roofline"]["avg_launch_us"], e["roofline"]["avg_launch_us"], b["roofline"]["frac"],
    b["roofline"]["frac_of_measured_copy_ceiling"], b["roofline"]["achieved"]))
print("done frac", b["sanity"]["done_frac_timed_region"], " cpu_baseline %.4g" % b["cpu_baseline"]["value"], " gpu/cpu %.0f" % (b["value"] / b["cpu_baseline"]["value"]))
for k in ("north_star_quadrotor_hovering_131072envs_1gpu", "north_star_quadrotor_hovering_1048576envs_1gpu"):
    print(k, "us %.2f  %.4g env-steps/s  hbm %.3f  valu %.2f" % (s[k]["us_per_launch"], s[k]["env_steps_per_s"], s[k]["roofline"]["frac"], s[k]["valu_issue"]["frac"]))
for k in ("C3_maze3d_discrete_9x9_256x256_16384envs", "C3_maze3d_continuous_9x9_256x256_16384envs"):
    print(k, "ms %.3f  %.4g  frac %.3f  GB/s %.0f" % (s[k]["ms_per_launch"], s[k]["env_steps_per_s"], s[k]["roofline"]["frac"], s[k]["roofline"]["achieved"]))
print("C3 cpu %.4g" % s["C3_maze3d_discrete_9x9_256x256_16384envs"]["cpu_baseline"]["value"])
c1 = s["C1_maze2d_15x15_escape_1env"]; print("C1 1env eager us %.2f graph us %.2f ; 2^20: %.4g" % (c1["us_per_step_eager"], c1["us_per_step_hipgraph_100"], s["C1_maze2d_15x15_escape_1048576envs"]["env_steps_per_s"]))
c4 = s["C4_humanoid_8192envs_256variants"]
print("C4 ms %.4f  %.4g env-steps/s  TFLOP/s %.3f frac %.4f  cpu %.4g  gpu/cpu %.1f" % (c4["ms_per_launch"], c4["env_steps_per_s"], c4["roofline"]["achieved"], c4["roofline"]["frac"], c4["cpu_baseline"]["value"], c4["env_steps_per_s"] / c4["cpu_baseline"]["value"]))
c5 = s["C5_mixed_share_65536quad_plus_65536maze3d_64x64"]; print("C5 %.4g  overlap %.4f  maze alone ms %.3f" % (c5["env_steps_per_s_two_streams"], c5["overlap_gain"], c5["ms_maze3d_alone"]))
a = s["A1_actuation_substep_65536envs"]; print("A1 act us %.2f frac %.3f GB/s %.0f  rate %.3g" % (a["us_per_substep_pair"], a["roofline"]["frac"], a["roofline"]["achieved"], a["robot_substeps_per_s"]))
g = s["A1GymEnv_python_side_16384envs_null_physics"]; print("A1GymEnv null ms eager %.3f graph %.3f  %.3g" % (g["ms_per_env_step"], g["ms_per_env_step_hipgraph"], g["env_steps_per_s_hipgraph"]))
q = s["quadrupedal_v0_urdf_8192envs"]; print("quadrupedal ms eager %.3f graph %.3f  %.4g env-steps/s  %.3g substeps/s" % (q["ms_per_env_step"], q["ms_per_env_step_hipgraph"], q["env_steps_per_s_hipgraph"], q["physics_substeps_per_s_hipgraph"]))
for f in ("walker_bench_kernel_stats.csv", "bench_all_secondary_kernel_stats.csv", "quadrotor_bench_kernel_stats.csv", "quadrotor_bench_graph_kernel_stats.csv"):
    try:
        for r in csv.DictReader(open(D + f)):
            if "walker_step_wave" in r["Name"] or "quadrotor_step" in r["Name"]:
                print(f, r["Name"][:70].replace("(anonymous namespace)::", ""), r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3))
    except FileNotFoundError:
        pass
for f in ("bench_walker.jsonl",):
    for l in open(D + f):
        if l.startswith("{"):
            j = json.loads(l); print(j["workload"], "ms %.4f  %.4g" % (j["avg_step_ms"], j["env_steps_per_s"]))
for f in ("walker_knockouts.txt", "quad_a1_knockouts.txt"):
    print(open(D + f).read().strip())
