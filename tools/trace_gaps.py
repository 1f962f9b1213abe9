"""Launch timeline of the step from the rocprofv3 kernel trace of tools/profile_bench.sh's stats pass (gpurun_out/<tag>/stats):
per launch of one steady-state step (median over the last steps): kernel duration, and the GAP between the end of the previous
dispatch and the start of this one -- what a dependent kernel boundary costs after each producer (MI355X_MICROARCH.md price list,
row `boundary`: 1.45-1.9 us + dirty bytes / 6 TB/s; rocprofv3 7.x stamps a dispatch's start at the previous one's end when the
launches are back to back, so the boundary is inside `dur` and the gaps read 0).  Usage: python tools/trace_gaps.py gpurun_out/<tag>"""
import csv, glob, os, sys
import numpy as np
src = sys.argv[1]
f = max(glob.glob(os.path.join(src, 'stats', '*', '*_kernel_trace.csv')), key=os.path.getmtime)
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# a step starts at every clip_adam launch + 1
adam = [i for i, n in enumerate(names) if 'clip_adam' in n]
steps = [(adam[i] + 1, adam[i + 1] + 1) for i in range(len(adam) - 1)]
steps = [s for s in steps if s[1] - s[0] == steps[-1][1] - steps[-1][0]][-12:]
n = steps[-1][1] - steps[-1][0]
dur = np.zeros((len(steps), n)); gap = np.zeros((len(steps), n))
for k, (a, b) in enumerate(steps):
    for j in range(n):
        r = rows[a + j]
        dur[k, j] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
        gap[k, j] = (int(r['Start_Timestamp']) - int(rows[a + j - 1]['End_Timestamp'])) / 1e3
print(f'{len(steps)} steps of {n} launches; medians in us; step = {np.median(dur.sum(1) + gap.sum(1)):.1f} us '
      f'(kernels {np.median(dur.sum(1)):.1f} + gaps {np.median(gap.sum(1)):.1f})')
for j in range(n):
    r = rows[steps[-1][0] + j]
    print(f'{j + 1:3d} gap {np.median(gap[:, j]):6.2f}  dur {np.median(dur[:, j]):7.2f}  grid {str(r.get("Grid_Size", r.get("Grid_Size_X", "?"))):>8s}  {r["Kernel_Name"][:100]}')
