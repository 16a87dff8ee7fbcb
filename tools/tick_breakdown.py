"""Aggregate the [timing] lines of a tick replay: SSLAM_TIMING=1 python tools/tick_legs.py 2> t.txt; python tools/tick_breakdown.py t.txt"""
import re, sys, collections
acc = collections.defaultdict(float); cnt = collections.Counter()
for line in open(sys.argv[1]):
    if not line.startswith("[timing]"): continue
    m = re.match(r"\[timing\] optimize: vertices (\d+) edges (\d+) \| structure ([\d.]+) upload ([\d.]+) LM ([\d.]+) \(iterations (\d+) trials (\d+)\) download ([\d.]+)", line)
    if m:
        for k, v in zip(("structure", "upload", "LM", "iterations", "trials", "download"), m.groups()[2:]): acc[k] += float(v)
        cnt["ticks"] += 1; continue
    m = re.match(r"\[timing\] LM chunk: (\d+) steps enqueued in ([\d.]+) ms, waited ([\d.]+)", line)
    if m: acc["steps"] += int(m.group(1)); acc["enqueue"] += float(m.group(2)); acc["wait"] += float(m.group(3)); cnt["chunks"] += 1; continue
    m = re.match(r"\[timing\] cholesky plan build ([\d.]+)", line)
    if m: acc["plan_build"] += float(m.group(1)); cnt["plans"] += 1; continue
    m = re.match(r"\[timing\] (.*?) ([\d.]+) ms", line)
    if m: acc["other: " + m.group(1)[:50]] += float(m.group(2))
n = max(cnt["ticks"], 1)
print("ticks", cnt["ticks"], "chunks/tick %.2f" % (cnt["chunks"] / n), "plans", cnt["plans"])
for k, v in sorted(acc.items()): print("  %-60s %.4f per tick" % (k, v / n))
