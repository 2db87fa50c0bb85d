"""Direct-form and executed FLOP of the Winograd launches of ONE iteration, from the library's launch log
(SCDA_GAN_GRAPH=0 SCDA_WINO_LOG=<file> python bench.py --steps 1 --warmup 0 --no-cpu-baseline): the Winograd-eligible share of
F_iter that bench.py's `roofline.iteration_executed` uses (WINO_ELIGIBLE_TFLOP).
    python scripts/wino_iteration_flops.py <log> <iterations in the log>"""
import collections
import sys
log, iters = sys.argv[1], int(sys.argv[2])
agg = collections.OrderedDict()
for l in open(log):
    f = l.split()
    if not f:
        continue
    grp = ("detector" if int(f[3]) == 1 else "scda nets") + (" wgrad" if f[0] == "wgrad" else " fwd + dgrad")
    a = agg.setdefault(grp, [0, 0.0])
    a[0] += 1; a[1] += float(f[10])
for k, (n, fl) in agg.items():
    print("%-24s %6.1f launches / iteration   executed %.4f TFLOP   direct-form %.4f TFLOP" % (k, n / iters, fl / iters / 1e12, 2.25 * fl / iters / 1e12))
