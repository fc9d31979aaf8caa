"""Per-step kernel time breakdown from a rocprofv3 kernel_trace.csv of bench.py (one step window: between the last
two gwc_fwd launches, so MIOpen find-mode / warm-up kernels do not pollute the totals).
    python tools/step_breakdown.py gpurun_out/prof/.../NNN_kernel_trace.csv [top] [--by-grid]
--by-grid: a second table with one row per (kernel, grid size, workgroup size) -- i.e. per launch SHAPE -- with its launch
count and AVERAGE duration, so that a kernel's time at one problem size (the roofline's 32->32 conv at 48x136x240, grid
256 x 512 threads per sample batch) can be read from the committed file instead of an average over all shapes."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
by_grid = "--by-grid" in sys.argv
argv = [a for a in sys.argv if a != "--by-grid"]
top = int(argv[2]) if len(argv) > 2 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "gwc_fwd" in r["Kernel_Name"] or "gwc_fused" in r["Kernel_Name"]]
# last pair of gwc_fwd launches with a whole step between them (kernel_roofline() launches gwc_fwd back to back at the end)
pairs = [(a, b) for a, b in zip(idx[:-1], idx[1:]) if b - a > 40]
# the last window with the MOST COMMON launch count: the first ones contain warm-up work (weight packing, MIOpen searches), the last
# one also spans the micro-benchmarks after the timed loop
mode = collections.Counter(p[1] - p[0] for p in pairs).most_common(1)[0][0]
a, b = [p for p in pairs if p[1] - p[0] == mode][-1]
win = rows[a:b]
tot, cnt = collections.Counter(), collections.Counter()
for r in win:
    n = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:64]
    tot[n] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[n] += 1
wall = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
print("steps found %d; window wall %.2f ms, kernel sum %.2f ms, launches %d" % (len(idx), wall / 1e6, sum(tot.values()) / 1e6, len(win)))
for n, t in tot.most_common(top):
    print("  %-66s %4d %8.3f ms" % (n, cnt[n], t / 1e6))

if by_grid:
    def short(r):
        return r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:64]

    def shape(r):
        g = "x".join(str(r.get("Grid_Size_" + a, r.get("Grid_Size", "?"))) for a in "XYZ")
        w = "x".join(str(r.get("Workgroup_Size_" + a, r.get("Workgroup_Size", "?"))) for a in "XYZ")
        return g + " / " + w
    t2, c2 = collections.Counter(), collections.Counter()
    for r in win:
        k = (short(r), shape(r))
        t2[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); c2[k] += 1
    print("\nper launch shape (grid threads X x Y x Z / workgroup), one steady-state step:")
    print("  %-58s %-26s %5s %10s %10s" % ("kernel", "grid / workgroup", "n", "avg us", "total ms"))
    for k, t in t2.most_common():
        print("  %-58s %-26s %5d %10.1f %10.3f" % (k[0][:58], k[1], c2[k], t / c2[k] / 1e3, t / 1e6))
