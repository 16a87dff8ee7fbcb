"""Fill the R4_* placeholders of DESIGN.md section 5 from profiles/r4_bench.json (run once after the final measurement)."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(ROOT, "profiles", "r4_bench.json")))
t, tl, f = d["tick_replay"], d["tick_replay_long"], d["frontend"]
ri = f.get("ransac_icp", {})
cpu1 = d["cpu_baseline"]["value"]
v = {
    "R4_HEADLINE": f"{d['value']:,.0f}", "R4_SINGLE_STREAM": f"{d['single_stream']['value']:,.0f}",
    "R4_FACTOR": f"{d['roofline_factor']['full_batch']['ms_per_launch']:.2f}", "R4_SOLVE": f"{d['roofline_factor']['full_batch']['backward_solve_ms']:.2f}",
    "R4_JACFRAC": f"{100 * d['roofline_jacobian_build']['full_batch']['frac']:.0f}", "R4_JAC": f"{d['roofline_jacobian_build']['full_batch']['ms_per_launch']:.2f}",
    "R4_FRAC": f"{100 * d['roofline_factor']['frac']:.1f}",
    "R4_SINGLEX": f"{d['single_graph']['iters_per_sec'] / cpu1:.1f}", "R4_SINGLE": f"{d['single_graph']['iters_per_sec']:,.0f}",
    "R4_TICKOPT": f"{t['ms_per_tick_optimize']:.2f}", "R4_TICKMARG": f"{t['ms_per_tick_marginals']:.2f}", "R4_TICKCPU": f"{t['cpu_baseline']['ms_per_tick']:.2f}", "R4_TICK": f"{t['ms_per_tick']:.2f}",
    "R4_LTICKOPT": f"{tl['ms_per_tick_optimize']:.2f}", "R4_LTICKMARG": f"{tl['ms_per_tick_marginals']:.2f}", "R4_LTICKCPU": f"{tl['cpu_baseline']['ms_per_tick']:.1f}", "R4_LTICK": f"{tl['ms_per_tick']:.2f}",
    "R4_PLANE": f"{d['plane_landmarks']['value']:,.0f}",
    "R4_FRONT": f"{f['batched']['planes_per_sec_kernels'] / 1e3:.1f}", "R4_PIPE": f"{f['pipelined']['planes_per_sec_incl_pcie_and_host'] / 1e3:.1f}",
    "R4_RANSACMS": f"{ri.get('ransac_kernel_ms_per_frame', 0):.3f}", "R4_ICPMS": f"{ri.get('icp_kernel_ms_per_frame', 0):.3f}", "R4_RIPLANES": f"{ri.get('planes_per_sec_kernels', 0):,.0f}",
    "R4_RBOX": f"{ri.get('cpu_baseline', {}).get('gpu_boxes_per_sec_kernels', 0):,.0f}", "R4_RCPU": f"{ri.get('cpu_baseline', {}).get('value', 0):,.0f}",
    "R4_CPU64": f"{d['cpu_baseline_multicore']['value']:.0f}", "R4_CPU1": f"{cpu1:.1f}",
}
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()
for k in sorted(v, key=len, reverse=True):
    s = s.replace(k, v[k])
left = re.findall(r"R4_[A-Z0-9]+", s)
open(p, "w").write(s)
print("filled", len(v), "placeholders; left:", left)
