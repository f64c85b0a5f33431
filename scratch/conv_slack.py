"""per conv closure of the train step: time against max(matrix time at the power ceiling, HBM time): where is slack left?
   usage (on the GPU box): MONOCON_HIP_PROFILE_DUMP=1 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-modes 2> dump.txt; python scratch/conv_slack.py dump.txt"""
import sys, collections
rows = []
for l in open(sys.argv[1]):
    if not l.startswith("prof "): continue
    p = l.split()
    rows.append((int(p[1]), p[2], int(p[4]), float(p[6]), float(p[8]), float(p[10])))
CEIL_TF, HBM = 1300.0, 4500.0      # executed fp16 TFLOP/s at the power ceiling; GB/s
out = []
for i, ph, kind, ms, gflop, mb in rows:
    if kind == 0 or ms <= 0: continue
    t_m = gflop * 3 / CEIL_TF          # ms
    t_h = mb / HBM                     # ms
    bound = max(t_m, t_h)
    out.append((ms - bound, i, ph, kind, ms, gflop, mb, t_m, t_h))
out.sort(reverse=True)
print("slack_ms  idx ph kind   ms   gflop     mb   t_mfma t_hbm")
for r in out[:40]: print("%7.3f %5d %s %d %7.3f %7.1f %7.1f %6.3f %6.3f" % r)
print("total slack of conv+wgrad closures: %.2f ms; total time %.2f ms" % (sum(r[0] for r in out), sum(r[4] for r in out)))
