"""dev: per-tensor relative difference between two BLOCKLOOP_DUMP files of tests/dev/blockloop_errors.py"""
import sys, torch
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
rows = []
for k in a:
    d = float((a[k] - b[k]).abs().max()); s = float(a[k].abs().max()) + 1e-30
    rows.append((d / s, k))
rows.sort(reverse=True)
for r in rows[:25]:
    print(f"{r[0]:.2e} {r[1]}")
